// The reference's unit tests (test/test_filter.cpp, Catch2) restated against the C++ drop-in
// surface, i.e. against the GPU path: same cases, same properties.  Catch2 is not used (the
// vendored ext/catch.hpp belongs to the reference); this is a plain main() that prints one line
// per check and exits non-zero on failure.
//
// Tolerances: the reference's own, 1e-10 (test/test_filter.cpp:8) -- the stage-level functions of include/nle/filter.hpp
// run on fp64 device matrices (nle_*64, csrc/generic64.hip), so its assertions hold unchanged.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "nle/filter.hpp"

using nle::Mat;
using nle::Vec;

static int g_fail = 0, g_total = 0;
#define CHECK(cond)                                                         \
    do {                                                                    \
        ++g_total;                                                          \
        if (!(cond)) {                                                      \
            ++g_fail;                                                       \
            std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond);   \
        }                                                                   \
    } while (0)

static double fro(const Mat& m) {
    double s = 0;
    for (int j = 0; j < m.cols(); ++j)
        for (int i = 0; i < m.rows(); ++i) s += m(i, j) * m(i, j);
    return std::sqrt(s);
}
static Mat sub(const Mat& a, const Mat& b) {
    Mat c(a.rows(), a.cols());
    for (int j = 0; j < a.cols(); ++j)
        for (int i = 0; i < a.rows(); ++i) c(i, j) = a(i, j) - b(i, j);
    return c;
}
// Eigen's isApprox: ||a - b|| <= prec * min(||a||, ||b||)
static bool isApprox(const Mat& a, const Mat& b, double prec) {
    if (a.rows() != b.rows() || a.cols() != b.cols()) return false;
    return fro(sub(a, b)) <= prec * std::min(fro(a), fro(b));
}
static Mat diag(const Vec& d) {
    Mat m(d.size(), d.size());
    for (int i = 0; i < d.size(); ++i) m(i, i) = d(i);
    return m;
}
static Mat randomUnit(int r, int c, std::mt19937& g) {  // (Mat::Random + 1) / 2
    std::uniform_real_distribution<double> u(0.0, 1.0);
    Mat m(r, c);
    for (int j = 0; j < c; ++j)
        for (int i = 0; i < r; ++i) m(i, j) = u(g);
    return m;
}

static void check_balanced(const Mat& Wa, const Mat& Wab, double tol) {
    CHECK(isApprox(Wa, Wa.transpose(), tol));
    for (int a = 0; a < Wa.rows(); ++a) {  // rows of [Wa Wab] sum to 1
        double s = 0;
        for (int j = 0; j < Wa.cols(); ++j) s += Wa(a, j);
        for (int j = 0; j < Wab.cols(); ++j) s += Wab(a, j);
        CHECK(std::fabs(s - 1.0) <= tol);
    }
    for (int j = 0; j < Wa.cols(); ++j) {  // columns of [Wa; Wab^T] sum to 1
        double s = 0;
        for (int a = 0; a < Wa.rows(); ++a) s += Wa(a, j);
        for (int a = 0; a < Wab.cols(); ++a) s += Wab(j, a);
        CHECK(std::fabs(s - 1.0) <= tol);
    }
}

int main() {
    const double tol = 1e-10;
    std::mt19937 gen(12345);

    {  // "OpenCV and Eigen conversions", test/test_filter.cpp:10-40
        nle::Image ones(2, 5, nle::NLE_64F, 1);
        for (int i = 0; i < 10; ++i) ones.ptr<double>()[i] = 1.0;
        Vec v = nle::opencv2eigen(ones);
        CHECK(v.size() == 10);
        bool all1 = true;
        for (int i = 0; i < 10; ++i) all1 = all1 && v(i) == 1.0;
        CHECK(all1);
        nle::Image m(3, 3, nle::NLE_64F, 1);
        for (int i = 0; i < 9; ++i) m.ptr<double>()[i] = i + 1;
        Vec lv = nle::opencv2eigen(m);
        bool lin = true;
        for (int i = 0; i < 9; ++i) lin = lin && lv(i) == i + 1;
        CHECK(lin);  // row-major order
        nle::Image back = nle::eigen2opencv(lv, 3, 3);
        bool same = true;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) same = same && back.at<double>(r, c) == m.at<double>(r, c);
        CHECK(same);
    }
    {  // "Eigen Decomposition", :42-68
        Mat R(3, 3);
        const double vals[9] = {2, -1, 0, -1, 2, -1, 0, -1, 2};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R(i, j) = vals[3 * i + j];
        Mat U;
        Vec D;
        std::tie(U, D) = nle::eigenDecomposition(R, tol);
        CHECK(D.size() == 3);
        CHECK(std::fabs(D(0) - 3.41421356) < 1e-5 && std::fabs(D(1) - 2.0) < 1e-5 && std::fabs(D(2) - 0.58578644) < 1e-5);
        CHECK(isApprox(U * diag(D) * U.transpose(), R, tol));
        CHECK(isApprox(U.transpose() * U, Mat::Identity(3, 3), tol));
    }
    {  // "Sinkhorn", :70-123
        Mat Wa, Wab;
        std::tie(Wa, Wab) = nle::sinkhorn(Mat::Identity(2, 2), Vec::Ones(2), 10);
        CHECK(Wa.rows() == 2 && Wa.cols() == 2 && Wab.cols() == 0);
        check_balanced(Wa, Wab, tol);  // phi = I is exact in fp32

        // "Balanced random matrix" (:96-122) on ten draws instead of the reference's one (its Mat::Random values are
        // platform dependent): R is NOT symmetrised, eigenDecomposition reads its lower triangle and drops the negative
        // eigenvalues, so phi = U may have fewer than 5 columns and the "A block" is then the first q rows (:247)
        for (int draw = 0; draw < 10; ++draw) {
            Mat R = randomUnit(5, 5, gen);
            Mat U;
            Vec D;
            std::tie(U, D) = nle::eigenDecomposition(R, tol);
            std::tie(Wa, Wab) = nle::sinkhorn(U, D, 20);
            CHECK(Wa.rows() == U.cols() && Wab.cols() == 5 - U.cols());
            check_balanced(Wa, Wab, tol);
        }
    }
    {  // "Orthogonalize", :126-153
        const int p = 10, n = 100, k = 5;
        Mat Wa = randomUnit(p, p, gen);
        Mat WaT = Wa.transpose();
        for (int j = 0; j < p; ++j)
            for (int i = 0; i < p; ++i) Wa(i, j) = (Wa(i, j) + WaT(i, j)) / 2;
        Mat Wab = randomUnit(p, n - p, gen);
        Mat V;
        Vec S;
        std::tie(V, S) = nle::orthogonalize(Wa, Wab, k);
        CHECK(S.size() > 0);
        CHECK(V.cols() > 0);
        CHECK(S.size() == V.cols());
        CHECK(V.rows() == n);
        CHECK(isApprox(V.transpose() * V, Mat::Identity(V.cols(), V.cols()), tol));  // :148-152
    }
    {  // end to end through the class: train on a small synthetic plane, projector property
        const int H = 48, W = 64;
        nle::Image L(H, W, nle::NLE_64F, 1);
        for (int r = 0; r < H; ++r)
            for (int c = 0; c < W; ++c)
                L.at<double>(r, c) = std::floor(128 + 60 * std::sin(0.2 * r) * std::cos(0.15 * c) + 20 * ((r * 7 + c * 13) % 5) / 5.0);
        nle::NLEFilter f;
        f.verbose = false;
        f.trainFilter(L, 4, 5, 16.0, 30.0, 10, 8);
        Vec ev = f.eigvals();
        CHECK(ev.size() == 8);
        CHECK(ev(0) > 0.99 && ev(0) < 1.01);
        Mat V = f.eigvecs();
        CHECK(isApprox(V.transpose() * V, Mat::Identity(8, 8), 1e-4));
        nle::Image y1 = f.apply(L, Vec(8, 1.0));
        nle::Image y2 = f.apply(y1, Vec(8, 1.0));
        double num = 0, den = 0;
        for (size_t i = 0; i < y1.total(); ++i) {
            const double d = y2.ptr<double>()[i] - y1.ptr<double>()[i];
            num += d * d;
            den += y1.ptr<double>()[i] * y1.ptr<double>()[i];
        }
        CHECK(std::sqrt(num / den) < 1e-5);  // V V^T is a projector
        bool threw = false;
        try {
            f.apply(nle::Image(10, 10, nle::NLE_64F, 1), Vec(8, 1.0));
        } catch (const std::runtime_error&) {
            threw = true;
        }
        CHECK(threw);  // src/filter.cpp:447-449
        threw = false;
        try {
            nle::computeKernel(L, H + 1, 5, 16.0, 30.0);
        } catch (const std::runtime_error&) {
            threw = true;
        }
        CHECK(threw);  // src/filter.cpp:117-119
    }
    {  // colour wrapper: device kernels against the host restatement of cv::cvtColor on 8-bit images
        std::uniform_int_distribution<int> u8(0, 255);
        nle::Image img(61, 83, nle::NLE_8U, 3);
        for (size_t i = 0; i < img.total() * 3; ++i) img.ptr<unsigned char>()[i] = (unsigned char)u8(gen);
        nle::Image lab_h = nle::bgr2lab8(img), lab_d = nle::bgr2lab8_device(img);
        nle::Image bgr_h = nle::lab2bgr8(lab_h), bgr_d = nle::lab2bgr8_device(lab_h);
        size_t bad_lab = 0, bad_bgr = 0;
        int worst = 0;
        for (size_t i = 0; i < img.total() * 3; ++i) {
            const int d1 = std::abs((int)lab_h.ptr<unsigned char>()[i] - (int)lab_d.ptr<unsigned char>()[i]);
            const int d2 = std::abs((int)bgr_h.ptr<unsigned char>()[i] - (int)bgr_d.ptr<unsigned char>()[i]);
            bad_lab += d1 != 0;
            bad_bgr += d2 != 0;
            worst = std::max(worst, std::max(d1, d2));
        }
        CHECK(worst == 0);                                   // both directions are integer arithmetic on shared tables: exact
        CHECK(bad_lab == 0);
        CHECK(bad_bgr == 0);
    }
    {  // denoise wrapper: the bilateral prefilter, device kernel against the host restatement (same fp32 tables and
       // summation order: bit-identical), on a smooth-plus-noise plane with ragged sizes and radius > width
        std::uniform_int_distribution<int> noise(-12, 12);
        for (auto dims : {std::pair<int, int>{45, 70}, std::pair<int, int>{7, 5}}) {
            nle::Image pl(dims.first, dims.second, nle::NLE_8U, 1);
            for (int y = 0; y < pl.rows; ++y)
                for (int x = 0; x < pl.cols; ++x)
                    pl.at<unsigned char>(y, x) =
                        (unsigned char)std::min(255, std::max(0, 128 + (int)(70 * std::sin(0.13 * x + 0.07 * y)) + noise(gen)));
            for (auto sig : {std::pair<int, int>{10, 10}, std::pair<int, int>{25, 3}}) {
                nle::Image h = nle::bilateralFilter8(pl, sig.first, sig.second);
                nle::Image d = nle::bilateralFilter8_device(pl, sig.first, sig.second);
                size_t bad = 0;
                double moved = 0;
                for (size_t i = 0; i < pl.total(); ++i) {
                    bad += h.ptr<unsigned char>()[i] != d.ptr<unsigned char>()[i];
                    moved += std::abs((int)h.ptr<unsigned char>()[i] - (int)pl.ptr<unsigned char>()[i]);
                }
                CHECK(bad == 0);
                CHECK(moved > 0);  // the filter does something
            }
        }
        nle::Image flat(9, 11, nle::NLE_8U, 1);
        for (size_t i = 0; i < flat.total(); ++i) flat.ptr<unsigned char>()[i] = 77;
        nle::Image ff = nle::bilateralFilter8_device(flat, 10, 10);
        bool same = true;
        for (size_t i = 0; i < flat.total(); ++i) same = same && ff.ptr<unsigned char>()[i] == 77;
        CHECK(same);
    }
    {  // trainForDenoise / denoise: errors and shape (src/filter.cpp:351-357)
        std::uniform_int_distribution<int> u8(0, 255);
        nle::Image img(48, 64, nle::NLE_8U, 3);
        for (int y = 0; y < img.rows; ++y)
            for (int x = 0; x < img.cols; ++x)
                for (int ch = 0; ch < 3; ++ch)
                    img.ptr<unsigned char>(y)[3 * x + ch] =
                        (unsigned char)std::min(255, std::max(0, 120 + 40 * ch + (int)(60 * std::sin(0.2 * x + 0.1 * y + ch)) + u8(gen) % 9));
        nle::NLEFilter f;
        f.verbose = false;
        f.trainForDenoise(img, 4, 5, 20.0, 30.0, 10, 8, 10, 3);
        nle::Image out = f.denoise(img, 2.0, 10, 3);
        CHECK(out.rows == img.rows && out.cols == img.cols && out.channels() == 3);
        bool threw = false;
        try {
            nle::Image small(10, 10, nle::NLE_8U, 3);
            f.denoise(small, 2.0);
        } catch (const std::runtime_error& e) {
            threw = std::string(e.what()).find("different size") != std::string::npos;
        }
        CHECK(threw);
        threw = false;
        try {
            nle::Image grey(48, 64, nle::NLE_8U, 1);
            f.denoise(grey, 2.0);
        } catch (const std::runtime_error&) {
            threw = true;
        }
        CHECK(threw);
    }
    std::printf("%d checks, %d failed\n", g_total, g_fail);
    return g_fail == 0 ? 0 : 1;
}
