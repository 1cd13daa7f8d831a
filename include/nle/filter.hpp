// nle/filter.hpp -- C++ drop-in surface of the MI355X-native nonlocal filter.
//
// Same names, parameter order, defaults, return-tuple order and exceptions as the reference's
// include/filter.hpp:10-54 (lightalchemist/nonlocal-image-edit).  The reference's types come
// from Eigen and OpenCV, neither of which exists in this build; the types below are minimal
// stand-ins with the SAME memory layouts (column-major fp64 matrix = Eigen::MatrixXd, row-major
// image = continuous cv::Mat), so a build that has Eigen/OpenCV can map them without copies.
// Every function here is a thin host wrapper over the C ABI in include/nle.h (libnle_hip.so);
// all N-sized arithmetic runs in HIP kernels on the GPU, there is no CPU fallback.
#pragma once
#ifndef NLE_FILTER_HPP
#define NLE_FILTER_HPP

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <tuple>
#include <utility>
#include <vector>

struct nle_ctx;
struct nle_filter;

namespace nle {

using DType = double;      // include/filter.hpp:12
const double EPS = 1e-10;  // include/filter.hpp:14

struct Point {  // include/filter.hpp:15-18
    int row;
    int col;
};

// Eigen::VectorXd stand-in
class Vec {
public:
    Vec() = default;
    explicit Vec(int n, double v = 0.0) : d_((size_t)n, v) {}
    Vec(std::initializer_list<double> l) : d_(l) {}
    static Vec Ones(int n) { return Vec(n, 1.0); }
    int size() const { return (int)d_.size(); }
    int rows() const { return (int)d_.size(); }
    int cols() const { return 1; }
    double& operator()(int i) { return d_[(size_t)i]; }
    double operator()(int i) const { return d_[(size_t)i]; }
    double* data() { return d_.data(); }
    const double* data() const { return d_.data(); }
    Vec head(int n) const {
        Vec v(n);
        for (int i = 0; i < n; ++i) v(i) = d_[(size_t)i];
        return v;
    }

private:
    std::vector<double> d_;
};

// Eigen::MatrixXd stand-in: COLUMN-major, fp64
class Mat {
public:
    Mat() = default;
    Mat(int r, int c, double v = 0.0) : r_(r), c_(c), d_((size_t)r * c, v) {}
    static Mat Identity(int r, int c) {
        Mat m(r, c);
        for (int i = 0; i < r && i < c; ++i) m(i, i) = 1.0;
        return m;
    }
    int rows() const { return r_; }
    int cols() const { return c_; }
    double& operator()(int i, int j) { return d_[(size_t)j * r_ + i]; }
    double operator()(int i, int j) const { return d_[(size_t)j * r_ + i]; }
    double* data() { return d_.data(); }
    const double* data() const { return d_.data(); }
    Mat transpose() const {
        Mat t(c_, r_);
        for (int j = 0; j < c_; ++j)
            for (int i = 0; i < r_; ++i) t(j, i) = (*this)(i, j);
        return t;
    }
    Mat leftCols(int n) const {
        Mat m(r_, n);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < r_; ++i) m(i, j) = (*this)(i, j);
        return m;
    }

private:
    int r_ = 0, c_ = 0;
    std::vector<double> d_;
};

Mat operator*(const Mat& a, const Mat& b);  // small host products (tests, glue)

// Eigen::PermutationMatrix stand-in: indices()[i] = row-major pixel index of the i-th pixel in the
// reference's [selected; rest] order (src/filter.cpp:156-164)
struct Permutation {
    std::vector<int> idx;
    std::vector<int>& indices() { return idx; }
    const std::vector<int>& indices() const { return idx; }
    int size() const { return (int)idx.size(); }
};

// continuous cv::Mat stand-in: ROW-major, interleaved channels; depth CV_8U or CV_64F
enum { NLE_8U = 0, NLE_64F = 6 };
const int OPENCV_MAT_TYPE = NLE_64F;  // include/filter.hpp:13
class Image {
public:
    int rows = 0, cols = 0;
    Image() = default;
    Image(int r, int c, int depth, int channels = 1)
        : rows(r), cols(c), depth_(depth), ch_(channels),
          buf_((size_t)r * c * channels * (depth == NLE_64F ? 8 : 1)) {}
    int channels() const { return ch_; }
    int depth() const { return depth_; }
    size_t total() const { return (size_t)rows * cols; }
    bool empty() const { return rows == 0 || cols == 0; }
    template <typename T>
    T* ptr(int r = 0) {
        return reinterpret_cast<T*>(buf_.data()) + (size_t)r * cols * ch_;
    }
    template <typename T>
    const T* ptr(int r = 0) const {
        return reinterpret_cast<const T*>(buf_.data()) + (size_t)r * cols * ch_;
    }
    template <typename T>
    T& at(int r, int c) {
        return ptr<T>(r)[c];
    }
    template <typename T>
    const T& at(int r, int c) const {
        return ptr<T>(r)[c];
    }
    Image clone() const { return *this; }

private:
    int depth_ = NLE_8U, ch_ = 1;
    std::vector<unsigned char> buf_;
};

// include/utils.hpp:11-41
inline int to1DIndex(int row, int col, int ncols) { return row * ncols + col; }
inline std::pair<int, int> to2DCoords(int index, int ncols) { return std::make_pair(index / ncols, index % ncols); }
Image eigen2opencv(const Vec& v, int nrows, int ncols);
Vec opencv2eigen(const Image& mat);

// ---- the five free functions, include/filter.hpp:20-33 ----
std::tuple<Permutation, Mat, Mat> computeKernel(const Image& mat, int nRowSamples, int nColSamples, DType hx,
                                                DType hy);
std::pair<Mat, Vec> eigenDecomposition(const Mat& M, DType eps = EPS);
std::pair<Vec, Mat> nystromApproximation(const Mat& Ka, const Mat& Kab);
std::pair<Mat, Mat> sinkhorn(const Mat& phi, const Vec& eigvals, int maxIter = 10);
std::pair<Mat, Vec> orthogonalize(const Mat& Wa, const Mat& Wab, int nEigVectors = 5, DType eps = EPS);

// src/filter.cpp:334-347 (a global in the reference, not declared in its header)
Vec transformEigenValues(const Vec& eigvals, const std::vector<DType>& weights);

// 8-bit BGR <-> Lab as cv::cvtColor(COLOR_BGR2Lab / COLOR_Lab2BGR) documents it (host)
Image bgr2lab8(const Image& bgr);
Image lab2bgr8(const Image& lab);
// the same on the GPU (what NLEFilter uses); agree with the host forms except for isolated rounding ties
Image bgr2lab8_device(const Image& bgr);
Image lab2bgr8_device(const Image& lab);

// cv::bilateralFilter(src, dst, -1, sigmaColor, sigmaSpace, BORDER_DEFAULT) on one 8-bit channel, as the denoise
// wrapper calls it (src/filter.cpp:371,535): host form and device form (bit-identical: same fp32 tables, same
// summation order); see nle_bilateral8 in nle.h for what "as OpenCV documents it" covers
Image bilateralFilter8(const Image& plane, double sigmaColor, double sigmaSpace);
Image bilateralFilter8_device(const Image& plane, double sigmaColor, double sigmaSpace);

// include/filter.hpp:35-54.  The trained state (m_eigvecs N x K', m_eigvals) lives on the GPU.
class NLEFilter {
public:
    // copyable like the reference's class (its state is two Eigen members, include/filter.hpp:52-53): copies
    // share the trained, immutable device-side filter
    NLEFilter();
    ~NLEFilter();

    void trainForEnhancement(const Image& image, int nRowSamples, int nColSamples, DType hx, DType hy,
                             int nSinkhornIter = 10, int nEigenVectors = 5);
    Image enhance(const Image& I, const std::vector<DType>& weights) const;
    // include/filter.hpp:40-45: the filter is trained on the bilateral-filtered L channel; denoise shrinks the
    // eigenvalues to min(lambda, 1)^k on the a and b channels (src/filter.cpp:349-410, 521-538)
    void trainForDenoise(const Image& image, int nRowSamples, int nColSamples, DType hx, DType hy, int nSinkhornIter,
                         int nEigenVectors, int sigmaColor = 10, int sigmaSpace = 10);
    Image denoise(const Image& I, DType k, int sigmaColor = 10, int sigmaSpace = 10) const;

    // private in the reference (include/filter.hpp:47-50); public here so tests and other hosts can
    // drive the hot path on a luminance plane directly
    Image apply(const Image& channel, const Vec& transformedEigVals) const;
    void trainFilter(const Image& channel, int nRowSamples, int nColSamples, DType hx, DType hy, int nSinkhornIter,
                     int nEigenVectors);
    // train on an fp32 luminance plane that is already on the device
    void trainOnDevice(const float* d_lum, int rows, int cols, int nRowSamples, int nColSamples, DType hx, DType hy,
                       int nSinkhornIter, int nEigenVectors);
    // per-layer outputs (L planes, CV_64F) -- what the 1e-4 per-detail-layer bar compares
    std::vector<Image> applyLayers(const Image& channel, int nLayers) const;

    Vec eigvals() const;                 // m_eigvals
    Mat eigvecs() const;                 // m_eigvecs, downloaded (N x K')
    void timings(double ms[6]) const;    // nle_filter_timings
    // nle_filter_diag: {formulation, p, rank Ka, rank Wa, rank Q, K', chol(Ka), chol(Wa)}
    void diag(int info[8]) const;
    bool verbose = true;                 // the reference's stdout stage banners (:483-498,506)

private:
    nle_ctx* ctx_ = nullptr;
    std::shared_ptr<nle_filter> fh_;  // nle_filter_destroy when the last copy goes
    nle_filter* f_ = nullptr;         // == fh_.get()
    int rows_ = 0, cols_ = 0;
    // NLE_DEVICES=<dev>,<dev>,...: trainForEnhancement / enhance shard the image by row slabs over one context (and
    // one host thread per call) per listed device; rank r's filter is group_[r] and f_ == group_[0].get()
    std::vector<std::shared_ptr<nle_filter>> group_;
    void trainForEnhancementGroup(const Image& image, int nRowSamples, int nColSamples, DType hx, DType hy,
                                  int nSinkhornIter, int nEigenVectors);
    Image enhanceGroup(const Image& image, const std::vector<DType>& weights) const;
};

}  // namespace nle

#endif
