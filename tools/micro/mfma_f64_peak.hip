// Achievable v_mfma_f64_16x16x4_f64 rate on gfx950 (no memory traffic): the ceiling for k_ghist_gemm / k_gram64 /
// k_project64.  build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    f64x4 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = f64x4{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
        a += 1e-9;
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 12345.678) out[blockIdx.x] = s;
}
template <int NACC>
void run(int waves_per_simd) {
    double* d;
    hipMalloc(&d, 1 << 20);
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 4000, grid = ncu * waves_per_simd;  // 256-thread blocks: one wave per SIMD each
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0, 2.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * NACC * 2.0 * 16 * 16 * 4;
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double cyc = ms * 1e-3 * khz * 1e3 / ((double)waves_per_simd * iters * NACC);
    printf("{\"independent_accumulators\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f, \"cycles_per_mfma_per_simd\": %.1f}\n",
           NACC, waves_per_simd, ms, flops / (ms * 1e-3) / 1e12, cyc);
    hipFree(d);
}
int main() {
    run<1>(1); run<4>(1); run<14>(1); run<14>(2); run<14>(3); run<8>(4);
    return 0;
}
