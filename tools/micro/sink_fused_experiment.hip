// EXPERIMENT, NOT BUILT INTO THE PRODUCT (round 3).  This is the text of a section that was appended to
// nonlocal-image-edit_amd/csrc/sorted.hip (it uses that file's internals: column_factors, combine_chunks, the chunk
// descriptors) and wired into sink_hist_tiled; kept for the record of what was measured.
//
// Result (cfg4, one MI355X, the whole tests/test_gpu_parity.py suite green on it, bitwise reproducible):
//     three-kernel half-iteration (k_hist_g + k_sorted_pass + k_hist_hh + k_z_reduce)   ~165 us   Sinkhorn 3.3 ms
//     this kernel, 512 threads, 209 VGPRs, 1 workgroup / CU                              ~400 us   Sinkhorn 8.3 ms
//     this kernel, 1024 threads, 32 rows / round, 128 VGPRs + 52 spilled                 ~430 us   Sinkhorn 8.9 ms
//     512 threads, 256 instead of 512 tasks                                                        Sinkhorn 10.1 ms
// Why: the plan for the bench image has 533 tasks / 5178 rounds (75 % of the chunk slots used: sparse level ranges hit the
// 16-rows-per-round limit of the MFMA tile long before 512 chunks); a round is a chain of dependent steps (er rows and
// chunk descriptors -> column indices -> G on the matrix cores -> pixel loop -> tree -> HH on the matrix cores) separated
// by workgroup barriers, and with the E table (32 KB), the tree buffer (45 KB) and > 128 registers there is ONE
// workgroup per CU, so nothing overlaps those steps: ~15-20 us per round against ~4 us per image row in k_sorted_pass,
// whose two workgroups per CU hide each other's combine / load phases and whose loads are requested a row ahead.  With
// next-round prefetching the estimate is ~7 us per round = ~140-160 us per half-iteration: no better than the three
// kernels, which run their 4 x 84 MB of table traffic at 3-5 TB/s.  Dropped.

// ------------------------------------------------------------------ one Sinkhorn half-iteration in ONE kernel
// The tiled pass (fused.hip: k_hist_g -> k_sorted_pass -> k_hist_hh) moves two tables of 256 nC doubles per image row
// through HBM per half-iteration (cfg4: 84 MB written by k_hist_g, read by the pass, 84 MB written by the pass, read by
// k_hist_hh) -- 4 x 84 MB against the 21 MB of column indices the pixel work itself needs.  Both tables are products with
// the SAME small operands (er: rows x nR, w o Ep: nR x 256 nC), so a workgroup that owns a LEVEL RANGE of 16 levels and a
// slab of image rows can make its part of g on the matrix cores just before its pixels use it, and contract its part of h
// with er just after, without either ever leaving the CU:
//     round (<= 16 rows, <= NTH chunks of the range's levels, the rows' chunks of a level range are contiguous)
//       G[16 rows][b][16 levels] = er_round (16 x nR) . (w o Ep)[nR][b][16 levels]        fp64 MFMA -> LDS
//       chunk threads: g row of their level from LDS, pixel loop, tree over a level's chunks      (as k_sorted_pass)
//       HH[a][b][16 levels] += er_round^T (nR x 16 rows) . H[16 rows][b][16 levels]        fp64 MFMA, in registers
//     task end: zpart[task][a, b] = sum over the 16 levels of Ep[x][a, b] HH[a][b][x]
// The g values are bit-identical to k_hist_g's (same products, same MFMA order); z is summed in another (fixed) order.
// The tasks (level range, row slab) are cut by k_sink_plan from the chunk counts, ~equal work each.
struct SinkTask {
    int x0, nx, r0, r1;  // levels [x0, x0 + nx), local rows [r0, r1)
};
constexpr int kFusedRows = 16;       // image rows of a round (the M / K extent of the two MFMA products)
constexpr int kFusedLevels = 16;     // levels of a task (one MFMA tile of columns per sample column b)
constexpr int kFusedTaskRows = 256;  // rows of a task at most (LDS tables)
constexpr int kFusedTasks = 1536;    // rows of the task table / partial-sum rows (k_z_reduce adds them all)

typedef double f64x4_s __attribute__((ext_vector_type(4)));

// One workgroup of 256 threads: chunk counts per level over the slab's rows, ranges of <= 16 levels starting at a
// non-empty level, row slabs per range in proportion to its share of the chunks.
__global__ __launch_bounds__(256) void k_sink_plan(const unsigned short* __restrict__ first, int nrows, int target,
                                                   SinkTask* __restrict__ tasks) {
    __shared__ unsigned int hist[kLevels];
    __shared__ int rx0[kLevels], rnx[kLevels], nranges, ntasks;
    __shared__ unsigned long long rmass[kLevels], total;
    const int tid = threadIdx.x;
    unsigned int h = 0;
    for (int r = 0; r < nrows; ++r) h += (unsigned)first[(size_t)r * 258 + tid + 1] - (unsigned)first[(size_t)r * 258 + tid];
    hist[tid] = h;
    __syncthreads();
    if (tid == 0) {
        int k = 0;
        unsigned long long tot = 0;
        for (int x = 0; x < kLevels;) {
            if (hist[x] == 0) {
                ++x;
                continue;
            }
            const int xe = min(kLevels, x + kFusedLevels);
            unsigned long long m = 0;
            for (int y = x; y < xe; ++y) m += hist[y];
            rx0[k] = x;
            rnx[k] = xe - x;
            rmass[k] = m;
            tot += m;
            ++k;
            x = xe;
        }
        nranges = k;
        total = tot;
        int nt = 0;
        const int min_slabs = (nrows + kFusedTaskRows - 1) / kFusedTaskRows;
        for (int i = 0; i < k; ++i) {
            long long want = tot ? (long long)((rmass[i] * (unsigned long long)target + tot / 2) / tot) : 1;
            int slabs = (int)max((long long)min_slabs, min((long long)nrows, max(1ll, want)));
            const int rows = (nrows + slabs - 1) / slabs;
            for (int r0 = 0; r0 < nrows && nt < kFusedTasks; r0 += rows) tasks[nt++] = SinkTask{rx0[i], rnx[i], r0, min(nrows, r0 + rows)};
        }
        ntasks = nt;
    }
    __syncthreads();
    for (int i = ntasks + tid; i < kFusedTasks; i += 256) tasks[i] = SinkTask{0, 0, 0, 0};
}

// KS: k-steps of the G product (ceil(nR / 4) <= KS), MAXB: blocks of four column indices per chunk (sorted_chp_max(W) / 4 <= MAXB)
template <int NC, bool REC, int NTH, int KS, int MAXB>
__global__ __launch_bounds__(NTH, 1) void k_sink_fused(int mode, const SinkTask* __restrict__ tasks,
                                                       const unsigned short* __restrict__ scol, const uint2* __restrict__ desc,
                                                       const unsigned short* __restrict__ first, GridSpec gs, int row0, int p,
                                                       int ldp, const double* __restrict__ Etab, const double* __restrict__ er,
                                                       const double* __restrict__ Ep, const double* __restrict__ w, double eps,
                                                       double* __restrict__ ybuf, const double* __restrict__ cvec,
                                                       const float* __restrict__ xvec, double kappa, double* __restrict__ zpart) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int SL = NC < 11 ? NC : 11, PS = SL | 1;
    constexpr int GS = NC * kFusedLevels;                         // doubles per row of G
    constexpr int NW = NTH / 64, TPW = (NC + NW - 1) / NW;        // waves, column tiles (one per b) per wave
    constexpr int FR = NTH / 32, MT = FR / 16;                    // image rows of a round, row tiles of the G product
    constexpr int USZ = NTH * PS > FR * GS ? NTH * PS : FR * GS;
    constexpr bool KEEP_E = NC <= 32;
    const int W = gs.W, nR = gs.nSelRows, ERS = nR | 1;
    const size_t pitch = sorted_row_pitch(W);
    double* sE = reinterpret_cast<double*>(smem_raw);
    double* sU = sE + ((W + 2) & ~1);  // G of the round, then the tree buffer
    double* sEr = sU + USZ;            // [16][ERS] the round's er rows (zero rows past the round's end)
    unsigned short* sIdx = reinterpret_cast<unsigned short*>(sEr + FR * 33);  // [16][16] thread of (row, level)'s first chunk
    unsigned short* sF0 = sIdx + FR * kFusedLevels;  // per task row: first chunk of the range,
    unsigned short* sCn = sF0 + kFusedTaskRows;              //               chunks of the range,
    unsigned short* sSt = sCn + kFusedTaskRows;              //               tree depth
    unsigned short* sRound = sSt + kFusedTaskRows;           // [rows + 1] first row of each round
    __shared__ int s_nrounds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const SinkTask tk = tasks[blockIdx.x];
    double* zout = zpart + (size_t)blockIdx.x * ldp;
    if (tk.nx <= 0) {  // padding of the task table
        for (int i = tid; i < ldp; i += NTH) zout[i] = 0.0;
        return;
    }
    const int x0 = tk.x0, nx = tk.nx, nrt = tk.r1 - tk.r0;
    const int cb0 = gs.colOff, cs = gs.colStep;
    const bool recip = mode == ROWPASS_RECIP, xmode = mode == ROWPASS_XVEC;
    const unsigned sEa = lds_addr(sE);
    const int ksteps = (nR + 3) >> 2;
    const bool two = nR > 16;
    for (int i = tid; i <= W; i += NTH) sE[i] = Etab[i];
    for (int i = tid; i < nrt; i += NTH) {
        const unsigned short* f = first + (size_t)(tk.r0 + i) * 258;
        const int f0 = f[x0], f1 = f[x0 + nx];
        sF0[i] = (unsigned short)f0;
        sCn[i] = (unsigned short)(f1 - f0);
        sSt[i] = f[257];
    }
    // B operands of the G product: lane (l15 = level, kq): a = 4 ks + kq, column tile = sample column b
    double bop[TPW][KS];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int b = wave + NW * ti;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int a = ks * 4 + kq;
            bop[ti][ks] = (recip && b < NC && l15 < nx && a < nR) ? w[a * NC + b] * Ep[(size_t)(x0 + l15) * p + a * NC + b] : 0.0;
        }
    }
    f64x4_s hacc[TPW][2];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) hacc[ti][0] = hacc[ti][1] = f64x4_s{0.0, 0.0, 0.0, 0.0};
    __syncthreads();
    if (tid == 0) {  // rounds: consecutive rows, at most 16, at most NTH chunks (a row has at most kT <= NTH)
        int k = 0, i = 0;
        while (i < nrt) {
            sRound[k++] = (unsigned short)i;
            int tot = 0, cnt = 0;
            while (i < nrt && cnt < FR && tot + (int)sCn[i] <= NTH) {
                tot += sCn[i];
                ++i;
                ++cnt;
            }
        }
        sRound[k] = (unsigned short)nrt;
        s_nrounds = k;
    }
    __syncthreads();
    const int nrounds = s_nrounds;
    for (int rd = 0; rd < nrounds; ++rd) {
        const int ra = sRound[rd], nr = (int)sRound[rd + 1] - ra;
        for (int i = tid; i < FR * ERS; i += NTH) {
            const int ri = i / ERS, a = i - ri * ERS;
            sEr[i] = (ri < nr && a < nR) ? er[(size_t)(tk.r0 + ra + ri) * nR + a] : 0.0;
        }
        if (tid < FR * kFusedLevels) sIdx[tid] = 0xffffu;
        int base = 0, myrow = 0, mychunk = 0, steps = 0;
        bool has = false;
        for (int i = 0; i < nr; ++i) {
            const int cn = sCn[ra + i];
            if (tid >= base && tid < base + cn) {
                has = true;
                myrow = i;
                mychunk = (int)sF0[ra + i] + tid - base;
            }
            base += cn;
            steps = max(steps, (int)sSt[ra + i]);
        }
        const int lrow = tk.r0 + ra + myrow;
        uint2 dsc = make_uint2(0u, 1u | (4u << 16));  // idle: len 0, m 1
        if (has) dsc = desc[(size_t)lrow * kT + mychunk];
        const int len = dsc_len(dsc), xl = dsc_level(dsc) - x0, j = dsc_j(dsc), m = dsc_m(dsc);
        uint2 idx[MAXB];
        {
            const uint2* slot = reinterpret_cast<const uint2*>(scol + (size_t)lrow * pitch + (size_t)mychunk * dsc_chp(dsc));
#pragma unroll
            for (int b = 0; b < MAXB; ++b) {
                uint2 v = make_uint2(0u, 0u);
                if (4 * b < len) v = slot[b];
                idx[b] = v;
            }
        }
        __syncthreads();  // sEr, sIdx
        double gv[NC];
#pragma unroll
        for (int b = 0; b < NC; ++b) gv[b] = 0.0;
        if (recip) {
#pragma unroll
            for (int ti = 0; ti < TPW; ++ti) {
                const int b = wave + NW * ti;
                if (b < NC) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        f64x4_s acc = f64x4_s{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks)
                            if (ks < ksteps) {
                                const int a = ks * 4 + kq;
                                const double aop = a < nR ? sEr[(mt * 16 + l15) * ERS + a] : 0.0;
                                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop[ti][ks], acc, 0, 0, 0);
                            }
#pragma unroll
                        for (int e = 0; e < 4; ++e) sU[(mt * 16 + kq + 4 * e) * GS + b * kFusedLevels + l15] = acc[e];
                    }
                }
            }
            __syncthreads();
            if (len > 0) {
#pragma unroll
                for (int b = 0; b < NC; ++b) gv[b] = sU[myrow * GS + b * kFusedLevels + xl];
            }
            __syncthreads();  // G is read before the tree buffer (same LDS) is written
        }
        double acc[NC];
#pragma unroll
        for (int b = 0; b < NC; ++b) acc[b] = 0.0;
        const double* cv_row = cvec ? cvec + (size_t)lrow * W : nullptr;
        const float* xv_row = xvec ? xvec + (size_t)(row0 + lrow) * W : nullptr;
        double* yb_row = ybuf ? ybuf + (size_t)lrow * W : nullptr;
        auto pixel = [&](const unsigned c8, const bool on) {
            double e[KEEP_E ? NC : 1];
            double y = 1.0;
            bool keep = on;
            if (recip) {
                double s0 = 0.0, s1 = 0.0;
                column_factors<NC, REC>(sEa, c8, cb0, cs, kappa, [&](const int b, const double ev) {
                    if constexpr (KEEP_E) e[b] = ev;
                    if (b & 1) s1 += ev * gv[b];
                    else s0 += ev * gv[b];
                });
                const double sm = s0 + s1;
                double r = __builtin_amdgcn_rcp(sm);
                r = fma(fma(-sm, r, 1.0), r, r);
                r = fma(fma(-sm, r, 1.0), r, r);
                y = r;
                keep = on && fabs(sm) >= eps;
            } else {
                if (xmode && on) y = cv_row[c8 >> 3] * (double)xv_row[c8 >> 3];
                if constexpr (KEEP_E) column_factors<NC, REC>(sEa, c8, cb0, cs, kappa, [&](const int b, const double ev) { e[b] = ev; });
            }
            y = keep ? y : 0.0;
            if (yb_row != nullptr && on) yb_row[c8 >> 3] = y;
            if constexpr (KEEP_E) {
#pragma unroll
                for (int b = 0; b < NC; ++b) acc[b] += e[b] * y;
            } else {
                column_factors<NC, REC>(sEa, c8, cb0, cs, kappa, [&](const int b, const double ev) { acc[b] += ev * y; });
            }
        };
        int wlen = len;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wlen = max(wlen, __shfl_xor(wlen, off));
        wlen = __builtin_amdgcn_readfirstlane(wlen);
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            if (4 * b >= wlen) break;
            pixel(idx[b].x & 0xffffu, 4 * b < len);
            NLE_PIXEL_FENCE();
            if (4 * b + 1 < wlen) {
                pixel(idx[b].x >> 16, 4 * b + 1 < len);
                NLE_PIXEL_FENCE();
            }
            if (4 * b + 2 < wlen) {
                pixel(idx[b].y & 0xffffu, 4 * b + 2 < len);
                NLE_PIXEL_FENCE();
            }
            if (4 * b + 3 < wlen) {
                pixel(idx[b].y >> 16, 4 * b + 3 < len);
                NLE_PIXEL_FENCE();
            }
        }
        if (len > 0 && j == 0) sIdx[myrow * kFusedLevels + xl] = (unsigned short)tid;
#pragma unroll
        for (int s0 = 0; s0 < NC; s0 += SL) {
            double v[SL];
#pragma unroll
            for (int i = 0; i < SL; ++i) v[i] = (s0 + i < NC) ? acc[s0 + i] : 0.0;
            combine_chunks<SL, PS>(v, sU, tid, len > 0, j, m, steps);
            // HH += er^T H for the sample columns of this slice: lane (l15 = level, kq): image rows 4 ks + kq
#pragma unroll
            for (int ti = 0; ti < TPW; ++ti) {
                const int b = wave + NW * ti;
                if (b >= s0 && b < s0 + SL && b < NC) {
#pragma unroll
                    for (int ks = 0; ks < FR / 4; ++ks) {
                        const int ri = ks * 4 + kq;
                        const unsigned t = sIdx[ri * kFusedLevels + l15];
                        const double bv = t != 0xffffu ? sU[t * PS + (b - s0)] : 0.0;
                        const double a0 = l15 < nR ? sEr[ri * ERS + l15] : 0.0;
                        hacc[ti][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv, hacc[ti][0], 0, 0, 0);
                        if (two) {
                            const double a1 = 16 + l15 < nR ? sEr[ri * ERS + 16 + l15] : 0.0;
                            hacc[ti][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, hacc[ti][1], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();  // before the next slice / round overwrites sU, sEr, sIdx
        }
    }
    // contract the task's 16 levels with Ep: lane (l15 = level, kq) holds rows a = kq + 4 e (+ 16) of column b
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int b = wave + NW * ti;
        if (b < NC) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                double v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int a = mt * 16 + kq + 4 * e;
                    v[e] = (l15 < nx && a < nR) ? hacc[ti][mt][e] * Ep[(size_t)(x0 + l15) * p + a * NC + b] : 0.0;
                }
#pragma unroll
                for (int off = 1; off < 16; off <<= 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], off);
                if (l15 == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int a = mt * 16 + kq + 4 * e;
                        if (a < nR) zout[a * NC + b] = v[e];
                    }
                }
            }
        }
    }
    for (int i = p + tid; i < ldp; i += NTH) zout[i] = 0.0;
}

int sink_fused_tasks() { return kFusedTasks; }
static size_t sink_fused_lds(int W, int nC, int nth) {
    const int SL = nC < 11 ? nC : 11, PS = SL | 1, fr = nth / 32;
    const size_t usz = std::max<size_t>((size_t)nth * PS, (size_t)fr * nC * kFusedLevels);
    return ((size_t)((W + 2) & ~1) + usz + (size_t)fr * 33) * sizeof(double) +
           ((size_t)fr * kFusedLevels + 3 * kFusedTaskRows + kFusedTaskRows + 2) * sizeof(unsigned short);
}
static int sink_fused_threads() {
    static const int v = [] {
        const char* e = std::getenv("NLE_FUSED_THREADS");
        return e && std::atoi(e) == 512 ? 512 : 1024;
    }();
    return v;
}
bool sink_fused_ok(GridSpec gs, int nrows_local) {
    static const bool off = std::getenv("NLE_NO_FUSED_SINKHORN") != nullptr;
    // the task table holds every task of the plan: <= 16 ranges x ceil(rows / 256) slabs at least, target + 16 otherwise
    const long long worst = 16ll * ((nrows_local + kFusedTaskRows - 1) / kFusedTaskRows) + 1024 + 16;
    return !off && nrows_local > 0 && worst <= kFusedTasks && gs.nSelCols >= 1 && gs.nSelCols <= 12 && gs.nSelRows <= 32 && gs.W <= sorted_max_width() &&
           sink_fused_lds(gs.W, gs.nSelCols, sink_fused_threads()) <= 160 * 1024 - 64;
}
hipError_t sink_plan(hipStream_t s, const unsigned short* d_first, int nrows_local, void* d_tasks) {
    int target = 512;
    if (const char* e = std::getenv("NLE_FUSED_TARGET")) target = std::min(1024, std::max(1, std::atoi(e)));
    hipLaunchKernelGGL(k_sink_plan, dim3(1), dim3(256), 0, s, d_first, nrows_local, target, static_cast<SinkTask*>(d_tasks));
    return hipGetLastError();
}
size_t sink_fused_task_bytes() { return sizeof(SinkTask) * kFusedTasks; }

// d_zpart: kFusedTasks x ldp doubles
hipError_t sink_fused(hipStream_t s, int mode, GridSpec gs, int row0, int nrows_local, int p, int ldp, const SortedRows& sr,
                      const double* d_er, const double* d_Ep, const double* d_w, double eps, double* d_ybuf,
                      const double* d_cvec, const float* d_xvec, double* d_zpart) {
    if (!sink_fused_ok(gs, nrows_local) || sr.tasks == nullptr) return hipErrorInvalidValue;
    const int nth = sink_fused_threads();
    const size_t shm = sink_fused_lds(gs.W, gs.nSelCols, nth);
    const bool small = gs.nSelRows <= 20 && sorted_chp_max(gs.W) <= 16;  // fewer registers: 5 k-steps, 4 index blocks
#define NLE_SF2(NCV, NTHV, KSV, MBV)                                                                                     \
    {                                                                                                                    \
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sink_fused<NCV, false, NTHV, KSV, MBV>),      \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                       \
        if (ea != hipSuccess) return ea;                                                                                 \
        hipLaunchKernelGGL((k_sink_fused<NCV, false, NTHV, KSV, MBV>), dim3((unsigned)kFusedTasks), dim3(NTHV), shm, s, mode, \
                           static_cast<const SinkTask*>(sr.tasks), sr.scol, sr.desc, sr.first, gs, row0, p, ldp, sr.E, d_er, \
                           d_Ep, d_w, eps, d_ybuf, d_cvec, d_xvec, sr.kappa, d_zpart);                                   \
    }
#define NLE_SF1(NCV, NTHV)                                                                                               \
    {                                                                                                                    \
        if (small) NLE_SF2(NCV, NTHV, 5, 4) else NLE_SF2(NCV, NTHV, 8, 8)                                                \
    }
#define NLE_SF(NCV)                                                                                                      \
    case NCV:                                                                                                            \
        if (nth == 512) NLE_SF1(NCV, 512) else NLE_SF1(NCV, 1024)                                                        \
        break;
    switch (gs.nSelCols) {
        NLE_SF(1) NLE_SF(2) NLE_SF(3) NLE_SF(4) NLE_SF(5) NLE_SF(6) NLE_SF(7) NLE_SF(8) NLE_SF(9) NLE_SF(10) NLE_SF(11) NLE_SF(12)
        default: return hipErrorInvalidValue;
    }
#undef NLE_SF
#undef NLE_SF1
#undef NLE_SF2
    return hipGetLastError();
}

}  // namespace nlek
