// Phase times of the host eigensolver (csrc/eigen_sym.cpp) next to the classic one-loop form.
// build: g++ -O3 -fopenmp-simd -pthread -Wno-psabi -o eig_phase_bench eig_phase_bench.cpp ; run: ./eig_phase_bench n ncols threads
#include "../../nonlocal-image-edit_amd/csrc/eigen_sym.cpp"
#include <chrono>
#include <cstdio>
#include <random>
using namespace nleh;
static double ms(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b){return std::chrono::duration<double,std::milli>(b-a).count();}
int main(int argc,char**argv){
  int n = argc>1?atoi(argv[1]):200; int ncols = argc>2?atoi(argv[2]):n; int nthreads = argc>3?atoi(argv[3]):1;
  std::mt19937_64 g(3); std::normal_distribution<double> nd;
  std::vector<double> X((size_t)n*n), A((size_t)n*n);
  for(auto&v:X) v=nd(g);
  for(int i=0;i<n;++i)for(int j=0;j<n;++j){double s=0;for(int kk=0;kk<n;++kk)s+=X[(size_t)kk*n+i]*X[(size_t)kk*n+j]*std::exp(-0.1*kk);A[(size_t)j*n+i]=s;}
  {
    std::vector<double> U((size_t)n*n), D(n);
    for(int rep=0;rep<3;++rep){
      auto t0=std::chrono::steady_clock::now();
      sym_eigen(A.data(),n,U.data(),D.data());
      auto t1=std::chrono::steady_clock::now();
      sym_eigen_top(A.data(),n,ncols,nthreads,U.data(),D.data());
      auto t2=std::chrono::steady_clock::now();
      sym_eigen_blocked(A.data(),n,U.data(),D.data());
      auto t3=std::chrono::steady_clock::now();
      printf("n=%d classic+blocked rotations %.3f ms\n",n,ms(t2,t3));
      printf("n=%d classic %.3f ms   three-phase (ncols=%d, threads=%d) %.3f ms\n",n,ms(t0,t1),ncols,nthreads,ms(t1,t2));
    }
  }
  for(int rep=0;rep<3;++rep){
    std::vector<double> V(A), d(n), e(n), hs(n), U((size_t)n*ncols);
    auto t0=std::chrono::steady_clock::now();
    tridiag_reduce(n, V.data(), d.data(), e.data(), hs.data());
    auto t1=std::chrono::steady_clock::now();
    std::vector<Sweep> sweeps; std::vector<double> cs, sn; cs.reserve((size_t)n*n); sn.reserve((size_t)n*n);
    ql_record(n, d.data(), e.data(), sweeps, cs, sn);
    auto t2=std::chrono::steady_clock::now();
    const int ldz = (n + 7) & ~7;
    std::vector<double> Z((size_t)ldz * n, 0.0);
    for (int i = 0; i < n; ++i) Z[(size_t)i * ldz + i] = 1.0;
    const int nvec = ldz / 8, nblocks = (nvec + 7) / 8;
    run_split(nblocks, nthreads, [&](int b) {
        apply_rotations_rows(ldz, Z.data(), b * 64, std::min(8, nvec - b * 8), sweeps.data(), sweeps.size(), cs.data(), sn.data());
    });
    auto t3=std::chrono::steady_clock::now();
    for (int j = 0; j < ncols; ++j) std::copy(Z.begin() + (size_t)j * ldz, Z.begin() + (size_t)j * ldz + n, U.begin() + (size_t)j * n);
    const int cparts = std::max(1, std::min(nthreads, (ncols + 3) / 4));
    run_split(cparts, nthreads, [&](int q) {
        const int j0 = (int)((long long)ncols * q / cparts), j1 = (int)((long long)ncols * (q + 1) / cparts);
        back_transform_cols(n, V.data(), hs.data(), U.data(), j0, j1);
    });
    auto t4=std::chrono::steady_clock::now();
    printf("n=%d reduce %.3f  ql_record %.3f (%zu sweeps, %zu rot)  rotations %.3f  backtransform %.3f\n", n, ms(t0,t1), ms(t1,t2), sweeps.size(), cs.size(), ms(t2,t3), ms(t3,t4));
  }
}
