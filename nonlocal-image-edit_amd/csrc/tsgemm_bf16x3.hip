// Nystrom extension Phi = K_AB^T (V_A Lambda^-1) (reference src/filter.cpp:275) on the bf16 matrix cores with SPLIT operands --
// BASELINE.json configs[2] names it ("bf16 MFMA Nystrom GEMM"), SURVEY.md Appendix C says why it cannot be plain bf16
// (operands rounded to 8 bits miss the 1e-4 per-layer bar by ~70x) and what recovers fp32 accuracy: every fp32 operand x is
// written as hi + mid + lo, three bf16 numbers carrying 24 bits between them, and the product keeps the six terms of weight
// >= 2^-16:   a b ~= ah bh + (ah bm + am bh) + (ah bl + al bh + am bm),   fp32 accumulation inside the MFMA.
// v_mfma_f32_32x32x16_bf16 does 16 384 multiply-adds in 32 cycles where the exact-fp32 v_mfma_f32_32x32x2_f32 of k_tsgemm
// does 2 048 in 64: six products still leave 16 / 6 ~ 2.7x.
//
// Opt-in (nle_ctx_set_nystrom_bf16x3 / NLE_NYSTROM_BF16X3=1), used by the materialised fp32 formulation
// (NLE_MODE_MATERIALISED) only; the default path never forms Phi.
//
// Tiling as k_tsgemm<NT, FUSED = true>: a 256-thread workgroup owns 128 pixels (wave = 32 rows) x NT 32-column tiles, the
// affinity A operand is generated in the lane that feeds it (lane l: row l & 31, k = k0 + 8 (l >> 5) + j, j < 8), the B
// operand comes pre-split (k_split_b3: [split][k / 8][column][8] bf16, so a lane's fragment is one 16-byte read) through a
// double-buffered LDS tile; the next tile's global loads are in flight under the current step's 6 NT MFMAs.
#include "kernels.h"

namespace nlek {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// round-to-nearest-even truncation of an fp32 to bf16 (finite inputs), as its 16 bits and as the fp32 it represents
__device__ __forceinline__ unsigned bf16_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = bf16_bits(x);
    const float r1 = x - __uint_as_float(h << 16);  // exact
    m = bf16_bits(r1);
    const float r2 = r1 - __uint_as_float(m << 16);  // exact
    l = bf16_bits(r2);
}
}  // namespace

// B (kd x ldb fp32, k-major) -> Bs[split s][kblk][col][8] bf16, kblk < ceil(kd / 8) (k >= kd: zeros), col < ldb
__global__ void k_split_b3(const float* __restrict__ B, int kd, int ldb, unsigned short* __restrict__ Bs) {
    const int kblks = (kd + 7) / 8;
    const size_t plane = (size_t)kblks * ldb * 8;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < plane; t += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(t & 7);
        const size_t q = t >> 3;
        const int col = (int)(q % ldb), kb = (int)(q / ldb);
        const int k = kb * 8 + j;
        const float x = k < kd ? B[(size_t)k * ldb + col] : 0.f;
        unsigned h, m, l;
        split3(x, h, m, l);
        Bs[t] = (unsigned short)h;
        Bs[plane + t] = (unsigned short)m;
        Bs[2 * plane + t] = (unsigned short)l;
    }
}

size_t ts_gemm_bf16x3_bsplit_elems(int kd, int ldb) { return (size_t)3 * ((kd + 7) / 8) * ldb * 8; }

hipError_t ts_gemm_bf16x3_split(hipStream_t s, const float* d_B, int kd, int ldb, unsigned short* d_Bs) {
    const size_t plane = (size_t)((kd + 7) / 8) * ldb * 8;
    hipLaunchKernelGGL(k_split_b3, dim3((unsigned)std::min<size_t>((plane + 255) / 256, 1024)), dim3(256), 0, s, d_B, kd, ldb, d_Bs);
    return hipGetLastError();
}

struct Ts3Args {
    const float* lum;
    GridSpec gs;
    const Sample4* samples;
    float nsw, npw;
    unsigned pix0;
    const unsigned short* Bs;  // split B
    int ldb, kd;
    float* C;
    int ldc;
    long long M;
    const float* cvec;
};

template <int NT>
__global__ __launch_bounds__(256) void k_tsgemm_bf16x3(Ts3Args a) {
    constexpr int PW = NT * 32;
    constexpr int NENT = 3 * 2 * PW;                  // 16-byte fragments of one 16-deep B tile
    constexpr int EPT = (NENT + 255) / 256;           // per thread
    __shared__ __attribute__((aligned(16))) uint4 sB[2][NENT];  // [buffer][split][kblk in step][column]
    __shared__ Sample4 sS[2][16];
    __shared__ float sScale[128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const long long m0 = (long long)blockIdx.x * 128;
    const int col0 = blockIdx.y * PW;
    const int kblks = (a.kd + 7) / 8;
    const size_t plane = (size_t)kblks * a.ldb;  // 16-byte entries per split

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;

    long long myrow = m0 + wave * 32 + l31;
    if (myrow >= a.M) myrow = a.M - 1;
    const unsigned gi = a.pix0 + (unsigned)myrow;
    const unsigned row = gi / (unsigned)a.gs.W;
    const float pr = (float)row, pc = (float)(gi - row * (unsigned)a.gs.W), px = a.lum[gi];

    const uint4* Bg = reinterpret_cast<const uint4*>(a.Bs);
    uint4 pre[EPT];
    Sample4 spre = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int k0) {  // the B tile and the 16 samples of step k0 into registers
        const int kb0 = k0 >> 3;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + 256 * i;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (e < NENT) {
                const int s = e / (2 * PW), rem = e - s * 2 * PW, kb = rem / PW, cc = rem - kb * PW;
                const int col = col0 + cc;
                if (kb0 + kb < kblks && col < a.ldb) v = Bg[(size_t)s * plane + (size_t)(kb0 + kb) * a.ldb + col];
            }
            pre[i] = v;
        }
        if (tid < 16) spre = (k0 + tid < a.kd) ? a.samples[k0 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + 256 * i;
            if (e < NENT) sB[buf][e] = pre[i];
        }
        if (tid < 16) sS[buf][tid] = spre;
    };
    fetch(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < a.kd; k0 += 16, buf ^= 1) {
        const bool more = k0 + 16 < a.kd;
        if (more) fetch(k0 + 16);
        // A fragments: 8 affinities of this lane's row, split three ways (a padded sample -- all zeros -- gives
        // exp2(finite) != 0, but its B rows are zero)
        unsigned ah[4], am[4], al[4];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            unsigned h0, m0_, l0, h1, m1, l1;
            split3(affinity_value(pr, pc, px, sS[buf][8 * half + j], a.nsw, a.npw), h0, m0_, l0);
            split3(affinity_value(pr, pc, px, sS[buf][8 * half + j + 1], a.nsw, a.npw), h1, m1, l1);
            ah[j >> 1] = h0 | (h1 << 16);
            am[j >> 1] = m0_ | (m1 << 16);
            al[j >> 1] = l0 | (l1 << 16);
        }
        const bf16x8 Ah = __builtin_bit_cast(bf16x8, make_uint4(ah[0], ah[1], ah[2], ah[3]));
        const bf16x8 Am = __builtin_bit_cast(bf16x8, make_uint4(am[0], am[1], am[2], am[3]));
        const bf16x8 Al = __builtin_bit_cast(bf16x8, make_uint4(al[0], al[1], al[2], al[3]));
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int cc = n * 32 + l31;
            const bf16x8 Bh = __builtin_bit_cast(bf16x8, sB[buf][0 * 2 * PW + half * PW + cc]);
            const bf16x8 Bm = __builtin_bit_cast(bf16x8, sB[buf][1 * 2 * PW + half * PW + cc]);
            const bf16x8 Bl = __builtin_bit_cast(bf16x8, sB[buf][2 * 2 * PW + half * PW + cc]);
            // smallest terms first
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[n], 0, 0, 0);
        }
        if (more) stage(buf ^ 1);  // (its last readers finished before the previous step's barrier)
        __syncthreads();
    }
    if (a.cvec != nullptr && tid < 128) sScale[tid] = (m0 + tid < a.M) ? a.cvec[m0 + tid] : 0.f;
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int col = col0 + n * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rl = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            const long long grow = m0 + rl;
            if (grow < a.M && col < a.ldc) {
                float v = acc[n][e];
                if (a.cvec != nullptr) v *= sScale[rl];
                a.C[(size_t)grow * a.ldc + col] = v;
            }
        }
    }
}

// C (M x ldc) = diag(c) K B with K = the affinity rows of pixels [pix0, pix0 + M) generated in registers, B given split
// (ts_gemm_bf16x3_split of the kd x ldb fp32 matrix); ldb >= ldc, both multiples of 4
hipError_t ts_gemm_bf16x3(hipStream_t s, const float* d_lum, GridSpec gs, const Sample4* d_samples, float nsw, float npw,
                          long long pix0, const unsigned short* d_Bs, int ldb, int kd, float* d_C, int ldc, long long M,
                          const float* d_c) {
    if (M <= 0) return hipSuccess;
    Ts3Args a;
    a.lum = d_lum;
    a.gs = gs;
    a.samples = d_samples;
    a.nsw = nsw;
    a.npw = npw;
    a.pix0 = (unsigned)pix0;
    a.Bs = d_Bs;
    a.ldb = ldb;
    a.kd = kd;
    a.C = d_C;
    a.ldc = ldc;
    a.M = M;
    a.cvec = d_c;
    const int ntiles = (ldc + 31) / 32;
    const int panels = (ntiles + 6) / 7;
    const int nt = (ntiles + panels - 1) / panels;
    const dim3 grid((unsigned)((M + 127) / 128), (unsigned)panels), block(256);
    switch (nt) {
#define NLE_T3(NTV) case NTV: hipLaunchKernelGGL((k_tsgemm_bf16x3<NTV>), grid, block, 0, s, a); break;
        NLE_T3(1) NLE_T3(2) NLE_T3(3) NLE_T3(4) NLE_T3(5) NLE_T3(6) NLE_T3(7)
#undef NLE_T3
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace nlek
