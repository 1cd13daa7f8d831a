// LDS atomic throughput on gfx950: the ceiling for the table kernels' histogram accumulation.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_bench lds_atomic_bench.hip ; run: ./lds_atomic_bench
// For each (type, address pattern) prints wave-instructions per CU per microsecond and cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kSlots = 256 * 11;  // the k_hist_pix<10> histogram: 256 levels x odd stride 11
constexpr int kIters = 2048, kPer = 10;

template <typename T>
__global__ __launch_bounds__(256) void k_atomics(const int* __restrict__ levels, int nlev, T* __restrict__ out) {
    __shared__ T sh[kSlots];
    for (int i = threadIdx.x; i < kSlots; i += 256) sh[i] = T(0);
    __syncthreads();
    const int tid = threadIdx.x;
    for (int it = 0; it < kIters; ++it) {
        const int x = levels[(it * 256 + tid) % nlev];  // L2-resident
        const T v = T(it + tid);
#pragma unroll
        for (int b = 0; b < kPer; ++b) atomicAdd(&sh[x * 11 + b], v);
    }
    __syncthreads();
    T s = T(0);
    for (int i = threadIdx.x; i < kSlots; i += 256) s += sh[i];
    if (s == T(12345)) out[blockIdx.x] = s;
}

template <typename T>
double run(const char* name, const char* pat, const int* d_lev, int nlev, int blocks_per_cu) {
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    T* d_out;
    CK(hipMalloc(&d_out, sizeof(T) * ncu * blocks_per_cu));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int grid = ncu * blocks_per_cu;
    hipLaunchKernelGGL(k_atomics<T>, dim3(grid), dim3(256), 0, 0, d_lev, nlev, d_out);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_atomics<T>, dim3(grid), dim3(256), 0, 0, d_lev, nlev, d_out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    int khz = 0;
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double wave_instr_per_cu = (double)blocks_per_cu * 4 * kIters * kPer;
    const double cyc = ms * 1e-3 * khz * 1e3 / wave_instr_per_cu;
    printf("{\"type\": \"%s\", \"pattern\": \"%s\", \"blocks_per_cu\": %d, \"ms\": %.4f, \"cycles_per_wave_instr\": %.2f, "
           "\"lane_atomics_per_s\": %.3e}\n",
           name, pat, blocks_per_cu, ms, cyc, (double)grid * 256 * kIters * kPer / (ms * 1e-3));
    CK(hipFree(d_out));
    return cyc;
}

int main() {
    const int nlev = 1 << 20;
    std::vector<int> h(nlev);
    int* d;
    CK(hipMalloc(&d, nlev * sizeof(int)));
    struct Pat { const char* name; int kind; } pats[] = {{"random256", 0}, {"lane_distinct", 1}, {"all_same", 2}, {"two_levels", 3}, {"runs_of_8", 4}};
    for (auto& p : pats) {
        srand(7);
        for (int i = 0; i < nlev; ++i) {
            switch (p.kind) {
                case 0: h[i] = rand() & 255; break;
                case 1: h[i] = i & 255; break;
                case 2: h[i] = 77; break;
                case 3: h[i] = (i & 1) ? 60 : 190; break;
                default: h[i] = ((i >> 3) * 37) & 255; break;
            }
        }
        CK(hipMemcpy(d, h.data(), nlev * sizeof(int), hipMemcpyHostToDevice));
        for (int bpc : {1, 2, 4}) {
            run<double>("f64", p.name, d, nlev, bpc);
            run<unsigned long long>("u64", p.name, d, nlev, bpc);
            run<float>("f32", p.name, d, nlev, bpc);
            run<unsigned int>("u32", p.name, d, nlev, bpc);
        }
    }
    return 0;
}
