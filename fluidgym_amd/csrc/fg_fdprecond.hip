// Fast-diagonalisation preconditioner for the pressure CG (see simulation/fd_precond.py for the
// maths): z = M^-1 r with M = the constant-coefficient (A = 1) pressure operator, applied as
//     r^ = (Qx^T (x) Qz^T) r  ->  one tridiagonal solve along y per mode  ->  z = (Qx (x) Qz) u .
// The two basis changes are genuinely GEMM-shaped (tall-skinny field matrix times a small dense
// eigenbasis) and run on the matrix cores with the exact-fp32 MFMA v_mfma_f32_32x32x2_f32 (gfx950:
// 64 cycles / 4096 MACs per wave, bitwise an fmaf chain, cdna_hip_programming.md section 3); the
// tridiagonal sweep streams the per-mode LU factors precomputed on the host.
#include <mutex>

#include "fg_internal.h"
#include "fg_cg.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;


struct GemmArgs {
    const float* A; long lda; long strideA;   // [M x K] row-major, per batch (stride 0 = shared)
    const float* B; long ldb; long strideB;   // [K x N]
    const float* Bt; long ldbt;               // optional: the same matrix stored transposed [N x K] (shared, stride 0)
    float* C; long ldc; long strideC;         // [M x N]
    int M, N, K;
    const int32_t* flags;                     // batch b is skipped when flags[b] != 0
    const float* dot_with; long strideW;      // optional: acc[b] += sum C .* W  (W laid out like C)
    FgDacc* dot_acc; int dot_stride; int dot_ns;  // slotted accumulator: dot_acc[b * dot_stride + (block & (ns-1))]
    FgCgJudge judge;                          // optional (judge.acc != nullptr): residual verdict before any work, fg_cg.h
};

// C = A * B, fp32 in / fp32 accumulate on MFMA 32x32x2.  256 threads = 4 waves in a 2 x 2 arrangement,
// each wave owns TM x TN MFMA tiles of 32 x 32 (TM = TN = 2: 128 x 128 block tile, 64 accumulator
// registers; TM = TN = 1: 64 x 64 block tile for launches that would otherwise leave CUs idle).
template <int TM, int TN, int BK>
__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs g, int tiles_n, int tiles_m) {
    // BK = K-depth staged per barrier pair.  The small launches of the batched 2-D case are latency-bound
    // (one L2 round trip per K-step, only 8 MFMAs per wave to cover it), so they use BK = 64: 4x fewer steps,
    // 4x more bytes in flight per step.
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int LDA_S = BM + 4, LDB_S = BN + 4;  // LDS row pitch; +4 floats keeps 16-B alignment, breaks conflicts
    constexpr int PA = BM * BK / 256, PB = BN * BK / 256;  // floats staged per thread (8 or 4)
    // 1-D grid, logical tile id = (env, m-tile, n-tile) with n fastest; the XCD remap hands every XCD a contiguous
    // range of ids, so the n-tiles that re-read one A row block (and an env's tiles) share one L2 (PMC before:
    // 13 MB fetched per launch against 6.4 MB algorithmic, each n-tile pulling A through a different XCD)
    const unsigned id = fg_xcd_remap(blockIdx.x, gridDim.x);
    const int per_b = tiles_n * tiles_m;
    const int b = id / per_b;
    if (g.flags && g.flags[b] != 0) return;
    const int rem = id - b * per_b, tile_m = rem / tiles_n, tile_n = rem - tile_m * tiles_n;
    if (g.judge.acc && fg_cg_judge(g.judge, b, rem == 0 && threadIdx.x == 0)) return;
    __shared__ __attribute__((aligned(16))) float As[BK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB_S];
    const float* __restrict__ A = g.A + (size_t)b * g.strideA;
    const float* __restrict__ B = g.B + (size_t)b * g.strideB;
    float* __restrict__ C = g.C + (size_t)b * g.strideC;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32 * TM, wn = (wave & 1) * 32 * TN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // global -> register staging: A tile BM x 16 (thread: PA consecutive k of one row), B tile 16 x BN
    constexpr int A_TPR = BK / PA;  // threads per A row
    constexpr int B_TPR = BN / PB;  // threads per B row
    const int a_row = tid / A_TPR, a_k = (tid % A_TPR) * PA;
    const int b_k = tid / B_TPR, b_n = (tid % B_TPR) * PB;
    float ra[PA], rb[PB];
    // fast path: tile fully inside the matrices and rows 16-B aligned -> float4 loads
    const bool vec_ok = (m0 + BM <= g.M) && (n0 + BN <= g.N) && (g.K % BK == 0) && ((g.lda & 3) == 0) && ((g.ldb & 3) == 0) &&
                        ((reinterpret_cast<size_t>(A) & 15) == 0) && ((reinterpret_cast<size_t>(B) & 15) == 0);
    auto load_tiles = [&](int k0) {
        if (vec_ok) {
            const float4* pa = reinterpret_cast<const float4*>(A + (size_t)(m0 + a_row) * g.lda + k0 + a_k);
#pragma unroll
            for (int q = 0; q < PA; q += 4) {
                const float4 v = pa[q >> 2];
                ra[q] = v.x; ra[q + 1] = v.y; ra[q + 2] = v.z; ra[q + 3] = v.w;
            }
            const float4* pb = reinterpret_cast<const float4*>(B + (size_t)(k0 + b_k) * g.ldb + n0 + b_n);
#pragma unroll
            for (int q = 0; q < PB; q += 4) {
                const float4 v = pb[q >> 2];
                rb[q] = v.x; rb[q + 1] = v.y; rb[q + 2] = v.z; rb[q + 3] = v.w;
            }
            return;
        }
        const int gm = m0 + a_row;
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int gk = k0 + a_k + q;
            ra[q] = (gm < g.M && gk < g.K) ? A[(size_t)gm * g.lda + gk] : 0.f;
        }
        const int gk = k0 + b_k;
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int gn = n0 + b_n + q;
            rb[q] = (gk < g.K && gn < g.N) ? B[(size_t)gk * g.ldb + gn] : 0.f;
        }
    };
    load_tiles(0);
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int q = 0; q < PA; ++q) As[(a_k + q) * LDA_S + a_row] = ra[q];
#pragma unroll
        for (int q = 0; q < PB; q += 4)
            *reinterpret_cast<float4*>(&Bs[b_k * LDB_S + b_n + q]) = make_float4(rb[q], rb[q + 1], rb[q + 2], rb[q + 3]);
        __syncthreads();
        if (k0 + BK < g.K) load_tiles(k0 + BK);  // overlaps with the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int kr = kk + (lane >> 5);
            float a[TM], bb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[kr * LDA_S + wm + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < TN; ++j) bb[j] = Bs[kr * LDB_S + wn + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
    }
    // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float dot = 0.f;
    const float* __restrict__ W = g.dot_with ? g.dot_with + (size_t)b * g.strideW : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = n0 + wn + j * 32 + (lane & 31);
                if (row < g.M && col < g.N) {
                    const float v = acc[i][j][r];
                    C[(size_t)row * g.ldc + col] = v;
                    if (W) dot += v * W[(size_t)row * g.ldc + col];
                }
            }
    if (W) {
        __shared__ float lds[4];
        float part[1] = {dot};
        fg_block_sum<1>(part, lds);
        if (tid == 0)
            acc_add(g.dot_acc + (size_t)b * g.dot_stride + ((unsigned)rem & (unsigned)(g.dot_ns - 1)),
                      (double)part[0]);
    }
}

// Split-K variant for the batched 2-D / small 3-D launches: one 32 x 32 output tile per workgroup, its four waves
// each own a quarter of K.  No operand is shared between the waves of a workgroup, so there is no LDS staging
// and no barrier in the K loop: every wave issues ALL its global loads up front (one exposed memory round trip),
// runs its MFMAs, and the four partial tiles meet in LDS once.  Versus the 64 x 64 tile this gives 4x the
// workgroups (the launches carry 10-64 active envs: 160-512 big tiles left most of the 256 CUs with <= 1 wave
// per SIMD) at the price of 2x the L2 reads, which the XCD remap keeps local (an env's tiles share one L2).
// MFMA 32x32x2 operand layout: lane l feeds A[row l%32][k-slot l/32], B[k-slot l/32][col l%32]; the k order inside
// a wave's slab is free as long as A and B agree, so lane half h takes the h-th half of the slab CONTIGUOUSLY
// (float4 loads from a row-major A, and from B when its transpose is available).
template <bool BT>
__global__ __launch_bounds__(256) void k_gemm_sk(GemmArgs g, int tiles_n, int tiles_m) {
    const unsigned id = fg_xcd_remap(blockIdx.x, gridDim.x);
    const int per_b = tiles_n * tiles_m;
    const int b = id / per_b;
    if (g.flags && g.flags[b] != 0) return;
    const int rem = id - b * per_b, tm = rem / tiles_n, tn = rem - tm * tiles_n;
    if (g.judge.acc && fg_cg_judge(g.judge, b, rem == 0 && threadIdx.x == 0)) return;
    const int m0 = tm * 32, n0 = tn * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const float* __restrict__ A = g.A + (size_t)b * g.strideA;
    const float* __restrict__ B = BT ? g.Bt : g.B + (size_t)b * g.strideB;
    const long ldb = BT ? g.ldbt : g.ldb;
    const int arow = min(m0 + l31, g.M - 1), bcol = min(n0 + l31, g.N - 1);
    const bool a_ok = m0 + l31 < g.M;
    const int KS = ((g.K + 7) >> 3) << 1;  // K slab per wave (even)
    const int kbeg = wave * KS, kend = min(g.K, kbeg + KS);
    const bool aligned = ((g.lda & 3) == 0) && ((reinterpret_cast<size_t>(A) & 15) == 0) &&
                         (!BT || (((ldb & 3) == 0) && ((reinterpret_cast<size_t>(B) & 15) == 0)));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kc = kbeg; kc < kend; kc += 64) {
        const int L = min(64, kend - kc), Lh = (L + 1) >> 1;
        const int k0 = kc + h * Lh, kstop = h ? kc + L : kc + Lh;
        float a[32], bb[32];
        if (L == 64 && aligned && (kc & 3) == 0) {
            const float4* pa = reinterpret_cast<const float4*>(A + (size_t)arow * g.lda + k0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 v = pa[q];
                a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
            }
            if (BT) {
                const float4* pb = reinterpret_cast<const float4*>(B + (size_t)bcol * ldb + k0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 v = pb[q];
                    bb[4 * q] = v.x; bb[4 * q + 1] = v.y; bb[4 * q + 2] = v.z; bb[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 32; ++j) bb[j] = B[(size_t)(k0 + j) * ldb + bcol];
            }
            // keep ALL loads ahead of the MFMAs (left alone, the scheduler interleaves them 2-3 loads at a time to save
            // registers, which exposes a memory round trip every few MFMAs)
            __builtin_amdgcn_sched_barrier(0);
            if (!a_ok) {
#pragma unroll
                for (int j = 0; j < 32; ++j) a[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 32; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bb[j], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int k = min(k0 + j, g.K - 1);
                const bool ok = k0 + j < kstop;
                const float av = A[(size_t)arow * g.lda + k];
                const float bv = BT ? B[(size_t)bcol * ldb + k] : B[(size_t)k * ldb + bcol];
                a[j] = (ok && a_ok) ? av : 0.f;
                bb[j] = ok ? bv : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 32; ++j)
                if (j < Lh) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bb[j], acc, 0, 0, 0);
        }
    }
    // combine the four K-partials: red[wave][reg][lane], then thread (wave w, lane) finishes regs 4w .. 4w+3
    __shared__ float red[4][16][64];
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    float* __restrict__ C = g.C + (size_t)b * g.strideC;
    const float* __restrict__ W = g.dot_with ? g.dot_with + (size_t)b * g.strideW : nullptr;
    float dot = 0.f;
    const int col = n0 + l31;
    float wv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // dot operand first: its loads must not queue behind the stores below
        const int row = m0 + q + 8 * wave + 4 * h;
        wv[q] = (W && row < g.M && col < g.N) ? W[(size_t)row * g.ldc + col] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = wave * 4 + q;
        const float v = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
        const int row = m0 + q + 8 * wave + 4 * h;  // C/D layout: row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        if (row < g.M && col < g.N) {
            C[(size_t)row * g.ldc + col] = v;
            dot += v * wv[q];
        }
    }
    if (W) {
        __shared__ float lds[4];
        float part[1] = {dot};
        fg_block_sum<1>(part, lds);
        if (tid == 0)
            acc_add(g.dot_acc + (size_t)b * g.dot_stride + ((unsigned)rem & (unsigned)(g.dot_ns - 1)), (double)part[0]);
    }
}

// Thomas sweep along y for every mode: forward y_j = (b_j - l_j y_{j-1}) inv_j, backward
// x_j = y_j - c'_j x_{j+1}.  One thread per (env, z-mode, x-mode); x-mode is the fastest index so every
// step is a coalesced row access.  The intermediate y_j stay in LDS ([j][lane], conflict-free), so the global
// traffic is: read b, inv, c' once (prefetched TRI_CH rows ahead of the recurrence), write x once -- no global
// store sits between a load and its use (the in-place global variant spent ~170 ns per recurrence step).
constexpr int TRI_CH = 16;   // rows per register chunk
constexpr int TRI_NB = 4;    // chunk buffers: TRI_NB - 1 chunks (48 rows) of loads in flight ahead of the recurrence
// The sweep is a serial chain per mode, so its duration is (number of chunks) x max(memory round trip / chunks in
// flight, chunk arithmetic): with one chunk in flight it sat at ~0.7 us per 16 rows whatever the batch (17-21 us for
// ny = 128, B = 4 .. 128).  The chunk loop is unrolled by TRI_NB so the rotating buffers keep static register indices.
// Row indices are CLAMPED instead of branched on: every load is straight-line code and the compiler can keep the
// later chunks in flight with counted s_waitcnt (a branchy form compiled to 36 x vmcnt(0) drains).
__global__ __launch_bounds__(64) void k_tridiag_y(float* __restrict__ x, const float* __restrict__ inv,
                                                   const float* __restrict__ cp, const float* __restrict__ lower,
                                                   const int32_t* __restrict__ flags, int nx, int ny, int nz, FgCgLead lead) {
    extern __shared__ __attribute__((aligned(16))) float ybuf[];  // [ny padded to TRI_CH * TRI_NB][64]
    const int b = blockIdx.y;
    if (flags && flags[b] != 0) return;
    if (lead.judge.acc && fg_cg_lead(lead, b, blockIdx.x == 0)) return;
    const int lane = threadIdx.x;
    int t = blockIdx.x * 64 + lane;
    const bool live = t < nx * nz;
    if (!live) t = nx * nz - 1;
    const int a = t % nx, c = t / nx;
    const size_t col = (size_t)c * ny * nx + a;
    float* __restrict__ xb = x + (size_t)b * nx * ny * nz + col;
    const float* __restrict__ iv = inv + col;
    const float* __restrict__ cpp = cp + col;
    const int last = ny - 1;
    constexpr int SPAN = TRI_CH * TRI_NB;
    const int nyp = (ny + SPAN - 1) / SPAN * SPAN;
    float bx[TRI_NB][TRI_CH], bi[TRI_NB][TRI_CH], bl[TRI_NB][TRI_CH];
    // ---- forward elimination: y_j = (b_j - l_j y_{j-1}) inv_j ; y to LDS
#pragma unroll
    for (int u = 0; u < TRI_NB - 1; ++u)
#pragma unroll
        for (int q = 0; q < TRI_CH; ++q) {
            const int j = min(u * TRI_CH + q, last);
            bx[u][q] = xb[(size_t)j * nx]; bi[u][q] = iv[(size_t)j * nx]; bl[u][q] = lower[j];
        }
    float prev = 0.f;
    for (int j0 = 0; j0 < nyp; j0 += SPAN) {
#pragma unroll
        for (int u = 0; u < TRI_NB; ++u) {
            constexpr int dummy = 0; (void)dummy;
            const int jc = j0 + u * TRI_CH;              // chunk computed now (buffer u)
            const int jl = jc + (TRI_NB - 1) * TRI_CH;   // chunk loaded now (buffer (u + TRI_NB - 1) % TRI_NB)
#pragma unroll
            for (int q = 0; q < TRI_CH; ++q) {
                const int j = min(jl + q, last);
                bx[(u + TRI_NB - 1) % TRI_NB][q] = xb[(size_t)j * nx];
                bi[(u + TRI_NB - 1) % TRI_NB][q] = iv[(size_t)j * nx];
                bl[(u + TRI_NB - 1) % TRI_NB][q] = lower[j];
            }
#pragma unroll
            for (int q = 0; q < TRI_CH; ++q) {
                const float v = (bx[u][q] - bl[u][q] * prev) * bi[u][q];
                prev = (jc + q <= last) ? v : prev;      // rows past the end keep the last value (select, no branch)
                ybuf[(jc + q) * 64 + lane] = prev;
            }
        }
    }
    // ---- back substitution: x_last = y_last ; x_j = y_j - c'_j x_{j+1}   (prev = 0 makes the first row x = y)
    prev = 0.f;
#pragma unroll
    for (int u = 0; u < TRI_NB - 1; ++u)
#pragma unroll
        for (int q = 0; q < TRI_CH; ++q) bx[u][q] = cpp[(size_t)min(nyp - (u + 1) * TRI_CH + q, last) * nx];
    for (int j0 = nyp - SPAN; j0 >= 0; j0 -= SPAN) {
#pragma unroll
        for (int u = 0; u < TRI_NB; ++u) {
            const int jc = j0 + (TRI_NB - 1 - u) * TRI_CH;   // chunks go top-down: buffer u holds chunk jc
            const int jl = jc - (TRI_NB - 1) * TRI_CH;
#pragma unroll
            for (int q = 0; q < TRI_CH; ++q)
                bx[(u + TRI_NB - 1) % TRI_NB][q] = cpp[(size_t)min(max(jl + q, 0), last) * nx];
#pragma unroll
            for (int q = 0; q < TRI_CH; ++q) bi[0][q] = ybuf[(jc + q) * 64 + lane];
#pragma unroll
            for (int q = TRI_CH - 1; q >= 0; --q) {
                const int j = jc + q;
                const float v = bi[0][q] - bx[u][q] * prev;
                prev = (j <= last) ? v : prev;
                if (live && j <= last) xb[(size_t)j * nx] = prev;
            }
        }
    }
}

// Cooperative variant (whole column block in LDS).  Two things bound the streaming kernel above, both per wave and
// independent of the batch (16-19 us for ny = 128 at B = 4 .. 128): a wave keeps at most 64 loads in flight, and a lone
// wave issues about one instruction per 4-5 cycles, so ~20 instructions per row (clamps, selects, addresses) cost
// ~100 cycles per row.  Here the four waves of a workgroup share one 64-column block:
//   1. float4 loads along the mode index (a wave instruction covers 4 rows x 64 columns; 24 loads per wave for
//      ny = 128), folded on the way into LDS to  bs = b inv,  ms = l inv,  cs = c'  with zero padding rows;
//   2. wave 0 runs  y_j = bs_j - ms_j y_{j-1}  and  x_j = y_j - cs_j x_{j+1}  out of LDS in 16-row register chunks:
//      one dependent FMA and four instructions per row, no clamps or selects (the zero padding rows are neutral);
//   3. all four waves store the result with float4s.
// Measured 7.8 us at B = 64 (256 x 128), 4.5k / 6.1k / 1.5k clocks for the three phases.  Needs nx % 4 == 0 and
// 3 x roundup(ny, 16) x 256 B of LDS (ny <= 208).
// TWO = true (ny beyond 208: the 512 x 256 grid of `large_env`): only two arrays fit the 160 KB -- c' is staged into the ms region
// AFTER the forward sweep has used it up (one more load phase and barrier in the middle; the streaming kernel it replaces there took
// 32 us per application).
// FAC (round 6): the launch also MAKES the per-env factors of the row-mean operator it solves with -- k_fd_rowmean_factor's
// arithmetic (same row means from the assembly's per-tile sums in the same order, same pivots: identical bits), run by wave 0 inside
// its forward sweep, where the factor chain (fma -> rcp -> mul per row) and the solve chain (one fma per row) interleave.  The
// separate factorisation launch was 26 us of a 287 us PISO step for 0.6 MB fetched: one wave per CU on a 128-row dependent chain.
// The factors go to `inv_out` / `cp_out` / `lower_out` for the later applications of the PISO step (both correctors share 1/A).
// An env the verdict stops before its first application (x_0 already good) still gets its factors: the next solve needs them.
struct FgFacArgs {
    const float* row_part; int tiles_x;      // per-tile row sums of 1/A the assembly left ([B][ny][tiles_x])
    const float* lam; const float* hy; const float* rhy; const float* dt;
    float* inv_out; float* cp_out; float* lower_out;
};
template <bool TWO, int CB, bool FAC = false>
__global__ __launch_bounds__(256) void k_tridiag_y_lds(float* __restrict__ x, const float* __restrict__ inv,
                                                       const float* __restrict__ cp, const float* __restrict__ lower,
                                                       const int32_t* __restrict__ flags, int nx, int ny, int nz, FgCgLead lead,
                                                       long fac_stride, int lower_stride, FgFacArgs fa = FgFacArgs{}) {
    // fac_stride / lower_stride: 0 = the grid's factors (A = 1 operator, shared by all envs); N / ny = per-env factors of the
    // row-mean operator (k_fd_rowmean_factor below)
    // CB = columns per workgroup: 64 (one float4 row segment per 16 lanes) or 32 (half the LDS: at ny = 128 two to three workgroups per
    // CU instead of one -- round 5: the whole launch resident at once on the 2-D env grids)
    extern __shared__ __attribute__((aligned(16))) float tbuf[];  // bs[nyp][CB] | ms[nyp][CB] | cs[nyp][CB]  (TWO: cs shares ms)
    constexpr int CBL = CB / 4, RPW = 64 / CBL, UQ = 32 / RPW;    // lanes per row | rows per wave access | accesses per wave and round (32 rows)
    const int b = blockIdx.y;
    bool solve = true;      // FAC: an env may need its factors without a solve
    if (flags && flags[b] != 0) { if (!FAC || !(fa.dt == nullptr || fa.dt[b] > 0.f)) return; solve = false; }
    // fused CG (fg_fftcg.hip): the verdict on the residual this application preconditions, and the leader's bookkeeping, ride here
    if (solve && lead.judge.acc && fg_cg_lead(lead, b, blockIdx.x == 0)) { if (!FAC) return; solve = false; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int CH = 16;
    const int nyp = (ny + CH - 1) / CH * CH, last = ny - 1;
    float* bs = tbuf;
    float* ms = tbuf + (size_t)nyp * CB;
    float* cs = TWO ? ms : tbuf + (size_t)2 * nyp * CB;
    // FAC: the env's row coefficients behind the column arrays: e = a hy | d | l | u | z (k_fd_rowmean_factor), a = row mean of 1/A
    float* rc = tbuf + (size_t)(TWO ? 2 : 3) * nyp * CB;
    if (FAC) {
        float* sa = rc + 5 * nyp;
        const float rn = 1.f / (float)nx;
        for (int j = threadIdx.x; j < ny; j += 256) {
            const float* __restrict__ pp = fa.row_part + ((size_t)b * ny + j) * fa.tiles_x;
            float acc = 0.f;
            for (int t = 0; t < fa.tiles_x; ++t) acc += pp[t];
            sa[j] = acc * rn;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < nyp; j += 256) {
            float e = 0.f, d = 1.f, l = 0.f, u = 0.f, z = 0.f;      // padding rows: the identity
            if (j < ny) {
                const float aj = sa[j];
                const float cm = j > 0 ? 0.5f * (sa[j - 1] * fa.rhy[j - 1] + aj * fa.rhy[j]) : 0.f;
                const float cpl = j < ny - 1 ? 0.5f * (aj * fa.rhy[j] + sa[j + 1] * fa.rhy[j + 1]) : 0.f;
                l = cm; u = cpl; d = -(cm + cpl); e = aj * fa.hy[j];
                z = (j == ny - 1) ? -(cm + cpl) : 0.f;      // the mode constant along x: its last pivot is shifted (k_fd_rowmean_factor)
                if (blockIdx.x == 0) fa.lower_out[(size_t)b * ny + j] = cm;
            }
            rc[j] = e; rc[nyp + j] = d; rc[2 * nyp + j] = l; rc[3 * nyp + j] = u; rc[4 * nyp + j] = z;
        }
        // (the barrier behind the load phase below orders these writes before wave 0 reads them)
    }
    const int lc = 4 * (lane % CBL), rsub = lane / CBL;
    int t4 = blockIdx.x * CB + lc;
    const bool live = t4 < nx * nz;
    if (!live) t4 = nx * nz - 4;
    const int a4 = t4 % nx, c4 = t4 / nx;
    const size_t col4 = (size_t)c4 * ny * nx + a4;
    float* __restrict__ xb4 = x + (size_t)b * nx * ny * nz + col4;
    const float* __restrict__ iv4 = inv + (size_t)b * fac_stride + col4;
    const float* __restrict__ cp4 = cp + (size_t)b * fac_stride + col4;
    lower += (size_t)b * lower_stride;
    if (FAC) {
        // only the right-hand side is loaded: bs = b (the factors are made below)
        if (solve)
            for (int jb = wave * 32; jb < nyp; jb += 128) {
                float4 vx[UQ];
#pragma unroll
                for (int q = 0; q < UQ; ++q) vx[q] = *reinterpret_cast<const float4*>(xb4 + (size_t)min(jb + RPW * q + rsub, last) * nx);
#pragma unroll
                for (int q = 0; q < UQ; ++q) {
                    const int j = jb + RPW * q + rsub;
                    if (j < nyp) {
                        const float m = (j > last) ? 0.f : 1.f;
                        *reinterpret_cast<float4*>(bs + j * CB + lc) = make_float4(vx[q].x * m, vx[q].y * m, vx[q].z * m, vx[q].w * m);
                    }
                }
            }
    } else
    for (int jb = wave * 32; jb < nyp; jb += 128) {  // 32 rows per wave per round
        float4 vx[UQ], vi[UQ], vc[UQ];
        float vl[UQ];
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            const int j = min(jb + RPW * q + rsub, last);
            vx[q] = *reinterpret_cast<const float4*>(xb4 + (size_t)j * nx);
            vi[q] = *reinterpret_cast<const float4*>(iv4 + (size_t)j * nx);
            if (!TWO) vc[q] = *reinterpret_cast<const float4*>(cp4 + (size_t)j * nx);
            vl[q] = lower[j];
        }
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            const int j = jb + RPW * q + rsub;
            if (j < nyp) {
                const float m = (j > last) ? 0.f : 1.f;
                const float l = vl[q] * m;
                const int o = j * CB + lc;
                *reinterpret_cast<float4*>(bs + o) =
                    make_float4(vx[q].x * vi[q].x * m, vx[q].y * vi[q].y * m, vx[q].z * vi[q].z * m, vx[q].w * vi[q].w * m);
                *reinterpret_cast<float4*>(ms + o) = make_float4(l * vi[q].x, l * vi[q].y, l * vi[q].z, l * vi[q].w);
                if (!TWO) *reinterpret_cast<float4*>(cs + o) = make_float4(vc[q].x * m, vc[q].y * m, vc[q].z * m, vc[q].w * m);
            }
        }
    }
    __syncthreads();
    if (FAC) {
        if (wave == 0 && lane < CB) {
            // factor chain and solve chain side by side: p_j = 1 / (d_j - l_j c'_{j-1}), c'_j = u_j p_j (k_fd_rowmean_factor, same
            // operations); y_j = b_j p_j - (l_j p_j) y_{j-1} (the folded form of the load phase above: bs = b inv, ms = l inv)
            const int col = (blockIdx.x * CB + lane) % nx;
            const float lm = fa.lam[col];
            const float zm = (col == 0) ? 1.f : 0.f;
            float prev = 0.f, cprev = 0.f;
            float* px = bs + lane;
            float* pm = ms + lane;                          // !TWO: p (to be stored as inv); TWO: c'
            float* pc = cs + lane;                          // c' of the back sweep
            float* iv = fa.inv_out + (size_t)b * fac_stride + (size_t)(blockIdx.x * CB + lane);      // TWO: p goes straight to memory
            for (int j0 = 0; j0 < nyp; j0 += CH, px += CH * CB, pm += CH * CB, pc += CH * CB) {
                float ax[CH], e8[CH], d8[CH], l8[CH], u8[CH], z8[CH], pv[CH], cv[CH];
#pragma unroll
                for (int q = 0; q < CH; ++q) ax[q] = solve ? px[q * CB] : 0.f;
                auto ldc = [&](const float* a, float (&o)[CH]) {
#pragma unroll
                    for (int q4 = 0; q4 < CH / 4; ++q4) {
                        const float4 v = *reinterpret_cast<const float4*>(a + j0 + 4 * q4);
                        o[4 * q4] = v.x; o[4 * q4 + 1] = v.y; o[4 * q4 + 2] = v.z; o[4 * q4 + 3] = v.w;
                    }
                };
                ldc(rc, e8); ldc(rc + nyp, d8); ldc(rc + 2 * nyp, l8); ldc(rc + 3 * nyp, u8); ldc(rc + 4 * nyp, z8);
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    const float d = fmaf(zm, z8[q], fmaf(e8[q], lm, d8[q]));
                    pv[q] = __builtin_amdgcn_rcpf(fmaf(-l8[q], cprev, d));
                    cprev = u8[q] * pv[q];
                    cv[q] = cprev;
                    const float m = (j0 + q > last) ? 0.f : 1.f;
                    prev = fmaf(-((l8[q] * m) * pv[q]), prev, ax[q] * pv[q]);
                    ax[q] = prev;
                }
#pragma unroll
                for (int q = 0; q < CH; ++q) {
                    px[q * CB] = ax[q];
                    const float m = (j0 + q > last) ? 0.f : 1.f;
                    if (TWO) { pm[q * CB] = cv[q] * m; if (j0 + q <= last) iv[(size_t)(j0 + q) * nx] = pv[q]; }
                    else { pm[q * CB] = pv[q]; pc[q * CB] = cv[q] * m; }
                }
            }
        }
    } else
    if (wave == 0 && lane < CB) {
        float prev = 0.f;
        float* px = bs + lane;
        const float* pm = ms + lane;
        for (int j0 = 0; j0 < nyp; j0 += CH, px += CH * CB, pm += CH * CB) {
            float ax[CH], am[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) { ax[q] = px[q * CB]; am[q] = pm[q * CB]; }
#pragma unroll
            for (int q = 0; q < CH; ++q) { prev = fmaf(-am[q], prev, ax[q]); ax[q] = prev; }
#pragma unroll
            for (int q = 0; q < CH; ++q) px[q * CB] = ax[q];
        }
    }
    if (TWO && !FAC) {
        __syncthreads();       // the forward sweep is through with ms: c' takes its place
        for (int jb = wave * 32; jb < nyp; jb += 128) {
            float4 vc[UQ];
#pragma unroll
            for (int q = 0; q < UQ; ++q) vc[q] = *reinterpret_cast<const float4*>(cp4 + (size_t)min(jb + RPW * q + rsub, last) * nx);
#pragma unroll
            for (int q = 0; q < UQ; ++q) {
                const int j = jb + RPW * q + rsub;
                if (j < nyp) {
                    const float m = (j > last) ? 0.f : 1.f;
                    *reinterpret_cast<float4*>(cs + j * CB + lc) = make_float4(vc[q].x * m, vc[q].y * m, vc[q].z * m, vc[q].w * m);
                }
            }
        }
        __syncthreads();
    }
    if (solve && wave == 0 && lane < CB) {
        float prev = 0.f;
        float* px = bs + (size_t)(nyp - CH) * CB + lane;
        const float* pc = cs + (size_t)(nyp - CH) * CB + lane;
        for (int j0 = nyp - CH; j0 >= 0; j0 -= CH, px -= CH * CB, pc -= CH * CB) {
            float ax[CH], ac[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) { ax[q] = px[q * CB]; ac[q] = pc[q * CB]; }
#pragma unroll
            for (int q = CH - 1; q >= 0; --q) { prev = fmaf(-ac[q], prev, ax[q]); ax[q] = prev; }
#pragma unroll
            for (int q = 0; q < CH; ++q) px[q * CB] = ax[q];
        }
    }
    __syncthreads();
    if (live)
        for (int jb = wave * 32; jb < ny; jb += 128) {
#pragma unroll
            for (int q = 0; q < UQ; ++q) {
                const int j = jb + RPW * q + rsub;
                if (j <= last) {
                    if (solve) *reinterpret_cast<float4*>(xb4 + (size_t)j * nx) = *reinterpret_cast<const float4*>(bs + j * CB + lc);
                    if (FAC) {      // the factors wave 0 left in LDS, for the later applications of this 1/A
                        float* io = fa.inv_out + (size_t)b * fac_stride + col4 + (size_t)j * nx;
                        float* co = fa.cp_out + (size_t)b * fac_stride + col4 + (size_t)j * nx;
                        if (!TWO) *reinterpret_cast<float4*>(io) = *reinterpret_cast<const float4*>(ms + j * CB + lc);
                        *reinterpret_cast<float4*>(co) = *reinterpret_cast<const float4*>(cs + j * CB + lc);
                    }
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Row-mean preconditioner (round 5).  M = the pressure operator with the coefficient 1/A replaced by its MEAN ALONG x per row and
// env, a_j = mean_i (1/A)_{ij}: still separable -- the x basis is the grid's (cosine / Fourier, eigenvalues lam_a), and mode a of env
// b solves the tridiagonal system
//     (a_j hy_j lam_a - c_{j-1/2} - c_{j+1/2}) u_j + c_{j-1/2} u_{j-1} + c_{j+1/2} u_{j+1} = r^_j,   c_{j+1/2} = (a_j / hy_j + a_{j+1} / hy_{j+1}) / 2
// (the A = 1 operator of simulation/fd_precond.py with a_j in place of 1).  It follows the part of 1/A that varies ACROSS the channel
// -- the mean profile in the channel stand-ins, the 40 : 1 wall refinement in RBC -- exactly, which the A = 1 operator cannot:
// measured on the oracle's matrices of the stirred 256 x 128 channel (profiles/scratch/precond_rowmean_exp.py) the residual after
// ONE iteration is 7-8e-6 against 1.2-1.3e-5, i.e. under the reference's tolerance of 1e-5 instead of just over it -- one CG iteration
// per solve instead of two.  The factors depend on the env's A: one factorisation per PISO step (both correctors share A).
// One wave per 64 modes and env; the row means are taken by the same wave first (the env's 1/A rows from L2).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int ROW_MAX_NY = 320;
__global__ __launch_bounds__(64) void k_fd_rowmean_factor(const float* __restrict__ rA, const float* __restrict__ lam, const float* __restrict__ hy,
                                                          const float* __restrict__ rhy, const float* __restrict__ dt, float* __restrict__ inv,
                                                          float* __restrict__ cp, float* __restrict__ lower_out, int nx, int ny,
                                                          const float* __restrict__ row_part, int tiles_x) {
    __shared__ __attribute__((aligned(16))) float se[ROW_MAX_NY], sd[ROW_MAX_NY], sl[ROW_MAX_NY], su[ROW_MAX_NY], sz[ROW_MAX_NY];
    __shared__ float sa[ROW_MAX_NY];
    const int b = blockIdx.y, lane = threadIdx.x;
    if (dt && !(dt[b] > 0.f)) return;
    const size_t N = (size_t)nx * ny;
    const float* __restrict__ ra = rA + (size_t)b * N;
    // a_j = mean over x of 1/A: from the per-tile row sums the assembly left (k_adv_build, fixed order), or -- 1/A fields that did
    // not come from it -- taken here (float4 loads, eight rows in flight per round: a lone wave streaming the env's field, slow)
    const float rn = 1.f / (float)nx;
    if (row_part) {
        for (int j = lane; j < ny; j += 64) {
            const float* __restrict__ pp = row_part + ((size_t)b * ny + j) * tiles_x;
            float acc = 0.f;
            for (int t = 0; t < tiles_x; ++t) acc += pp[t];
            sa[j] = acc * rn;
        }
    } else
    for (int j0 = 0; j0 < ny; j0 += 8) {
        float part[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = min(j0 + q, ny - 1);
            float acc = 0.f;
            for (int i = 4 * lane; i < nx; i += 256) {
                const float4 v = *reinterpret_cast<const float4*>(ra + (size_t)j * nx + i);
                acc += (v.x + v.y) + (v.z + v.w);
            }
            part[q] = acc;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float tot = fg_wave_sum(part[q]);
            if (lane == 0 && j0 + q < ny) sa[j0 + q] = tot * rn;
        }
    }
    __syncthreads();
    const int nyp = (ny + 7) & ~7;
    for (int j = lane; j < nyp; j += 64) {
        if (j < ny) {
            const float aj = sa[j];
            const float cm = j > 0 ? 0.5f * (sa[j - 1] * rhy[j - 1] + aj * rhy[j]) : 0.f;
            const float cpl = j < ny - 1 ? 0.5f * (aj * rhy[j] + sa[j + 1] * rhy[j + 1]) : 0.f;
            sl[j] = cm; su[j] = cpl; sd[j] = -(cm + cpl); se[j] = aj * hy[j];
            // the mode that is constant along x meets the singular Neumann operator along y: its last pivot is shifted by the row's own
            // diagonal (fd_precond.py: a rank-one change along the null space, which CG never sees)
            sz[j] = (j == ny - 1) ? -(cm + cpl) : 0.f;
            if (blockIdx.x == 0) lower_out[(size_t)b * ny + j] = cm;
        } else { sl[j] = 0.f; su[j] = 0.f; sd[j] = 1.f; se[j] = 0.f; sz[j] = 0.f; }      // padding rows: the identity
    }
    __syncthreads();
    const int col = blockIdx.x * 64 + lane;                  // (nx is a multiple of 64: every lane owns a mode)
    const float lm = lam[col];
    const float zm = (col == 0) ? 1.f : 0.f;
    float* __restrict__ iv = inv + (size_t)b * N + col;
    float* __restrict__ cq = cp + (size_t)b * N + col;
    float cprev = 0.f;
    // eight rows per trip: their coefficients come out of LDS in one batch, then the dependent chain fma -> rcp -> mul runs from
    // registers (a first version read LDS and branched on `live` inside every step: one exposed LDS round trip per row, 17 us for 128 rows)
    for (int j0 = 0; j0 < nyp; j0 += 8) {
        float e8[8], d8[8], l8[8], u8[8], z8[8], pv[8], cv[8];
        {   // (128-bit LDS reads: ten per trip instead of forty)
            auto ld8 = [&](const float* a, float (&o)[8]) {
                const float4 x0 = *reinterpret_cast<const float4*>(a + j0), x1 = *reinterpret_cast<const float4*>(a + j0 + 4);
                o[0] = x0.x; o[1] = x0.y; o[2] = x0.z; o[3] = x0.w; o[4] = x1.x; o[5] = x1.y; o[6] = x1.z; o[7] = x1.w;
            };
            ld8(se, e8); ld8(sd, d8); ld8(sl, l8); ld8(su, u8); ld8(sz, z8);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float d = fmaf(zm, z8[q], fmaf(e8[q], lm, d8[q]));
            pv[q] = __builtin_amdgcn_rcpf(fmaf(-l8[q], cprev, d));
            cprev = u8[q] * pv[q];
            cv[q] = cprev;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (j0 + q < ny) { iv[(size_t)(j0 + q) * nx] = pv[q]; cq[(size_t)(j0 + q) * nx] = cv[q]; }
    }
}

}  // namespace

// dynamic LDS above 64 KB needs an explicit opt-in per kernel (once per process and size class)
static bool tridiag_lds_ready(size_t bytes, bool fac = false) {
    static std::mutex mu;
    static size_t granted_dev[64][2] = {{0}};      // per device (the attribute belongs to the device current when it is set) and kernel family
    static bool failed_dev[64][2] = {{false}};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return false; }
    std::lock_guard<std::mutex> lock(mu);
    size_t& granted = granted_dev[dev][fac ? 1 : 0];
    bool& failed = failed_dev[dev][fac ? 1 : 0];
    if (bytes <= granted) return true;
    if (failed) return false;
    const void* plain[2] = {reinterpret_cast<const void*>(k_tridiag_y_lds<false, 64>), reinterpret_cast<const void*>(k_tridiag_y_lds<true, 64>)};
    const void* facs[2] = {reinterpret_cast<const void*>(k_tridiag_y_lds<false, 64, true>), reinterpret_cast<const void*>(k_tridiag_y_lds<true, 64, true>)};
    for (const void* f : (fac ? facs : plain))
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            (void)hipGetLastError();
            failed = true;
            return false;
        }
    granted = bytes;
    return true;
}

static int launch_gemm(const fg_state* s, const GemmArgs& g, int batch, int expect_active, hipStream_t st) {
    // algorithmic traffic per env: A read + C written (+ the dot operand); the transform matrix stays in L2
    const double bytes = 4.0 * ((double)g.M * g.K + (double)g.M * g.N + (g.dot_with ? (double)g.M * g.N : 0.0));
    const long big_blocks = (long)((g.N + 127) / 128) * ((g.M + 127) / 128) * batch;
    // The kernel form is chosen from the BATCH, not from how many envs are still iterating: the two forms sum over k in different
    // orders, and `expect_active` comes from the host's convergence polls, whose schedule depends on the solver's history -- a
    // replayed state (get_state -> set_state -> step) would pick another form and round differently.  (Split-K is ~10 % faster
    // once only a few envs of a large batch are left: B = 16 of 64 live 8.8 vs 10.1 us.)
    (void)expect_active;
    const long live_tiles = (long)((g.N + 63) / 64) * ((g.M + 63) / 64) * batch;
    const bool tiled = big_blocks >= 512 || live_tiles >= 384;
    const int slot = fg_prof_slot(s, tiled ? FG_PK_GEMM : FG_PK_GEMM_SK, g.flags, batch, bytes, 2.0 * g.M * g.N * g.K, st);
    if (g.M <= 64 && g.K == 64 && (long)((g.N + 127) / 128) * batch >= 256) {
        // the z transform of a 64-plane grid (TCF 128 x 64 x 64: M = K = nz = 64, N = ny nx): a 64 x 128 tile with the WHOLE K in one
        // stage -- the 128 x 128 tile below spent half its MFMAs on padding rows and four barrier pairs on a K of 64 (round 4:
        // 21 us per transform at 24 TFLOP/s; the transform moves 34 MB per launch, i.e. ~7 us of traffic)
        const int tn = (g.N + 127) / 128;
        FG_LAUNCH_P(s, slot, (k_gemm_f32<1, 2, 64>), dim3((unsigned)(tn * batch)), dim3(256), 0, st, g, tn, 1);
    } else if (big_blocks >= 512) {  // >= 2 workgroups per CU with the 128 x 128 tile
        const int tn = (g.N + 127) / 128, tm = (g.M + 127) / 128;
        FG_LAUNCH_P(s, slot, (k_gemm_f32<2, 2, 16>), dim3((unsigned)(tn * tm * batch)), dim3(256), 0, st, g, tn, tm);
    } else if (live_tiles >= 384) {
        // enough live 64 x 64 tiles for ~1.5 workgroups per CU: the LDS-staged tile reads each operand half as often
        // (measured at 256 x 128, all envs live: B = 64 17.5 us vs 27 us split-K; B = 16 10.1 vs 8.8; B = 4 9.6 vs 6.5)
        const int tn = (g.N + 63) / 64, tm = (g.M + 63) / 64;
        FG_LAUNCH_P(s, slot, (k_gemm_f32<1, 1, 64>), dim3((unsigned)(tn * tm * batch)), dim3(256), 0, st, g, tn, tm);
    } else {
        const int tn = (g.N + 31) / 32, tm = (g.M + 31) / 32;
        dim3 grid((unsigned)(tn * tm * batch));
        if (g.Bt) FG_LAUNCH_P(s, slot, (k_gemm_sk<true>), grid, dim3(256), 0, st, g, tn, tm);
        else FG_LAUNCH_P(s, slot, (k_gemm_sk<false>), grid, dim3(256), 0, st, g, tn, tm);
    }
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

// factors of the row-mean operator for the envs' current 1/A (rA [B][N]); 2-D grids with a fast x transform only
bool fg_fd_rowmean_ok(const fg_state* s) {
    return s->fd_rowmean && s->fd_dct_x != 0 && s->grid.dims == 2 && s->grid.ny <= ROW_MAX_NY && (s->grid.nx & 63) == 0 && s->fd_lam_x != nullptr;
}
int fg_fd_rowmean_factor(fg_state* s, const float* rA, const float* dt, hipStream_t st, const float* row_part, int tiles_x) {
    const FgGrid& G = s->grid;
    const size_t BN = (size_t)G.B * G.n;
    if (!s->fd_row_inv) {
        FG_HIP_CHECK(hipMalloc(&s->fd_row_inv, sizeof(float) * BN));
        FG_HIP_CHECK(hipMalloc(&s->fd_row_cp, sizeof(float) * BN));
        FG_HIP_CHECK(hipMalloc(&s->fd_row_lower, sizeof(float) * (size_t)G.B * G.ny));
    }
    hipLaunchKernelGGL(k_fd_rowmean_factor, dim3((G.nx + 63) / 64, G.B), dim3(64), 0, st, rA, (const float*)s->fd_lam_x, G.h[1], G.rh[1], dt,
                       s->fd_row_inv, s->fd_row_cp, s->fd_row_lower, G.nx, G.ny, row_part, tiles_x);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

// per-mode Thomas solve along y of the transformed field `cur` (in place), all envs with flags == 0; lead (optional): the fused CG's
// verdict / leader bookkeeping taken by this launch (FgCgLead, fg_cg.h)
// the tridiagonal launches that can make the row-mean factors themselves (k_tridiag_y_lds<.., 64, FAC>): LDS for the column arrays
// plus the env's six row arrays
bool fg_fd_tridiag_can_factor(const fg_state* s) {
    const FgGrid& G = s->grid;
    if (!s->fd_facfuse || G.dims != 2 || G.nz != 1 || (G.nx & 63) != 0) return false;
    const size_t nyp = (size_t)(G.ny + 15) / 16 * 16;
    const size_t lds_three = (size_t)3 * nyp * 64 * sizeof(float) + 6 * nyp * sizeof(float), lds_two = (size_t)2 * nyp * 64 * sizeof(float) + 6 * nyp * sizeof(float);
    return lds_three <= 160 * 1024 || lds_two <= 160 * 1024;
}

// per-mode Thomas solve along y of the transformed field `cur` (in place), all envs with flags == 0; lead (optional): the fused CG's
// verdict / leader bookkeeping taken by this launch (FgCgLead, fg_cg.h); factor_from (optional): the launch also makes the per-env
// row-mean factors from these per-tile row sums of 1/A (fg_fd_tridiag_can_factor)
int fg_fd_tridiag(fg_state* s, float* cur, hipStream_t st, const FgCgLead* lead, bool use_rowmean, const float* factor_from, const float* factor_dt) {
    const FgGrid& G = s->grid;
    const int nx = G.nx, ny = G.ny, nz = G.nz, B = G.B;
    const long N = G.n;
    FgCgLead ld = {};
    if (lead) ld = *lead;
    // per env: x read + written; inv + c' read -- the grid's factors are shared by all envs (they come from L2 after the first env: not
    // counted), the row-mean operator's are the env's own (8 more bytes per cell); the factoring launch writes them instead (a kind of its
    // own in the live profile: it also runs the factor chain)
    dim3 grid((nx * nz + 63) / 64, B);
    const bool own_factors = factor_from || (use_rowmean && s->fd_row_inv && (nx & 3) == 0 && (size_t)2 * ((ny + 15) / 16 * 16) * 64 * sizeof(float) <= 160 * 1024);
    const int slot = fg_prof_slot(s, factor_from ? FG_PK_TRIDIAG_FAC : FG_PK_TRIDIAG, s->flags, B, (own_factors ? 16.0 : 8.0) * N, (factor_from ? 9.0 : 5.0) * N, st);
    const size_t lds_coop = (size_t)3 * ((ny + 15) / 16 * 16) * 64 * sizeof(float);
    const size_t lds_two = lds_coop / 3 * 2;      // two arrays: c' staged after the forward sweep (ny up to 320)
    if (factor_from) {
        const size_t rows = (size_t)6 * ((ny + 15) / 16 * 16) * sizeof(float);
        if (!s->fd_row_inv) {
            const size_t BN = (size_t)B * N;
            FG_HIP_CHECK(hipMalloc(&s->fd_row_inv, sizeof(float) * BN));
            FG_HIP_CHECK(hipMalloc(&s->fd_row_cp, sizeof(float) * BN));
            FG_HIP_CHECK(hipMalloc(&s->fd_row_lower, sizeof(float) * (size_t)B * ny));
        }
        FgFacArgs fa;
        fa.row_part = factor_from; fa.tiles_x = (nx + 63) / 64; fa.lam = s->fd_lam_x; fa.hy = G.h[1]; fa.rhy = G.rh[1]; fa.dt = factor_dt;
        fa.inv_out = s->fd_row_inv; fa.cp_out = s->fd_row_cp; fa.lower_out = s->fd_row_lower;
        if (lds_coop + rows <= 160 * 1024 && tridiag_lds_ready(lds_coop + rows, true))
            FG_LAUNCH_P(s, slot, (k_tridiag_y_lds<false, 64, true>), grid, dim3(256), lds_coop + rows, st, cur, (const float*)nullptr, (const float*)nullptr,
                        (const float*)nullptr, s->flags, nx, ny, nz, ld, N, ny, fa);
        else if (lds_two + rows <= 160 * 1024 && tridiag_lds_ready(lds_two + rows, true))
            FG_LAUNCH_P(s, slot, (k_tridiag_y_lds<true, 64, true>), grid, dim3(256), lds_two + rows, st, cur, (const float*)nullptr, (const float*)nullptr,
                        (const float*)nullptr, s->flags, nx, ny, nz, ld, N, ny, fa);
        else { fg_set_error("fg_fd_tridiag: no LDS for the factoring launch (fg_fd_tridiag_can_factor said otherwise)"); return FG_ERR_HIP; }
        FG_HIP_CHECK(hipGetLastError());
        return FG_OK;
    }
    // per-env factors of the row-mean operator (fg_fd_rowmean_factor) when the caller asked for them and the LDS kernels run
    const bool rowf = use_rowmean && s->fd_row_inv && (nx & 3) == 0 && lds_two <= 160 * 1024;
    const float* f_inv = rowf ? s->fd_row_inv : s->fd_inv;
    const float* f_cp = rowf ? s->fd_row_cp : s->fd_cp;
    const float* f_lower = rowf ? s->fd_row_lower : s->fd_lower;
    const long fstride = rowf ? N : 0;
    const int lstride = rowf ? ny : 0;
    if ((nx & 3) == 0 && lds_coop <= 160 * 1024 && tridiag_lds_ready(lds_coop)) {
        FG_LAUNCH_P(s, slot, (k_tridiag_y_lds<false, 64>), grid, dim3(256), lds_coop, st, cur, f_inv, f_cp, f_lower,
                    s->flags, nx, ny, nz, ld, fstride, lstride, FgFacArgs{});
    } else if ((nx & 3) == 0 && lds_two <= 160 * 1024 && tridiag_lds_ready(lds_two)) {
        FG_LAUNCH_P(s, slot, (k_tridiag_y_lds<true, 64>), grid, dim3(256), lds_two, st, cur, f_inv, f_cp, f_lower,
                    s->flags, nx, ny, nz, ld, fstride, lstride, FgFacArgs{});
    } else {
        FG_LAUNCH_P(s, slot, k_tridiag_y, grid, dim3(64), (size_t)((ny + 63) / 64 * 64) * 64 * sizeof(float), st, cur,
                    s->fd_inv, s->fd_cp, s->fd_lower, s->flags, nx, ny, nz, ld);
    }
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

// z = M^-1 r for all envs with flags == 0; optionally rz_acc[b * rz_stride] += r . z
int fg_fd_apply(fg_state* s, const float* r, float* z, FgDacc* rz_acc, int rz_stride, int rz_ns, int expect_active,
                hipStream_t st, const FgCgJudge* judge) {
    if (expect_active <= 0 || expect_active > s->grid.B) expect_active = s->grid.B;
    const FgGrid& G = s->grid;
    const int nx = G.nx, ny = G.ny, nz = G.nz, B = G.B;
    const long N = G.n;
    float* t1 = s->w[3];
    float* t2 = s->w[4];
    GemmArgs g = {};
    g.flags = s->flags; g.dot_with = nullptr; g.strideW = 0; g.dot_acc = nullptr; g.dot_stride = 0; g.dot_ns = 1;
    g.judge = FgCgJudge{};
    if (s->fd_dct_x) {
        // uniform FIXED x axis: the basis is the DCT-II basis, applied as an FFT per row (fg_fdfft.hip)
        if (int rc = fg_fd_dct_forward(s, r, t1, st, 0, judge)) return rc;
    } else {
        if (judge) g.judge = *judge;     // (the first kernel of the application only)
        // forward x: t1[rows, a] = sum_i r[rows, i] Qx[i, a]
        g.A = r; g.lda = nx; g.strideA = N;
        g.B = s->fd_Qx; g.ldb = nx; g.strideB = 0; g.Bt = s->fd_QxT; g.ldbt = nx;
        g.C = t1; g.ldc = nx; g.strideC = N;
        g.M = ny * nz; g.N = nx; g.K = nx;
        if (int rc = launch_gemm(s, g, B, expect_active, st)) return rc;
        g.judge = FgCgJudge{};
    }
    float* cur = t1;
    if (G.dims == 3) {
        // forward z: t2[c, m] = sum_k QzT[c, k] t1[k, m]   (m over ny*nx)
        g.A = s->fd_QzT; g.lda = nz; g.strideA = 0;
        g.B = t1; g.ldb = (long)ny * nx; g.strideB = N; g.Bt = nullptr; g.ldbt = 0;
        g.C = t2; g.ldc = (long)ny * nx; g.strideC = N;
        g.M = nz; g.N = ny * nx; g.K = nz;
        if (int rc = launch_gemm(s, g, B, expect_active, st)) return rc;
        cur = t2;
    }
    if (int rc = fg_fd_tridiag(s, cur, st, nullptr, false)) return rc;
    if (G.dims == 3) {
        // inverse z: t1[k, m] = sum_c Qz[k, c] t2[c, m]
        g.A = s->fd_Qz; g.lda = nz; g.strideA = 0;
        g.B = t2; g.ldb = (long)ny * nx; g.strideB = N; g.Bt = nullptr; g.ldbt = 0;
        g.C = t1; g.ldc = (long)ny * nx; g.strideC = N;
        g.M = nz; g.N = ny * nx; g.K = nz;
        if (int rc = launch_gemm(s, g, B, expect_active, st)) return rc;
        cur = t1;
    }
    if (s->fd_dct_x) return fg_fd_dct_inverse(s, cur, z, r, rz_acc, rz_stride, rz_ns, st);
    // inverse x: z[rows, i] = sum_a cur[rows, a] QxT[a, i], fused r.z
    g.A = cur; g.lda = nx; g.strideA = N;
    g.B = s->fd_QxT; g.ldb = nx; g.strideB = 0; g.Bt = s->fd_Qx; g.ldbt = nx;
    g.C = z; g.ldc = nx; g.strideC = N;
    g.M = ny * nz; g.N = nx; g.K = nx;
    g.dot_with = rz_acc ? r : nullptr; g.strideW = N; g.dot_acc = rz_acc; g.dot_stride = rz_stride; g.dot_ns = rz_ns;
    if (int rc = launch_gemm(s, g, B, expect_active, st)) return rc;
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Separable Helmholtz preconditioner of the advection-diffusion solves:  M = I/dt - nu (Dxx + Dyy + Dzz), the matrix C of
// k_adv_build without its advective part.  On grids whose transform axes (x, and z in 3-D) are PERIODIC and uniform the
// eigenvectors of Dxx / Dzz are the ones the pressure preconditioner already holds (fd_Qx / fd_Qz; a periodic axis has no
// boundary rows, so velocity and pressure share the 1-D operator), and M decouples into one tridiagonal system along y per mode
// and env (coefficients: k_helm_coeffs, factorised per solve by fg_helm_factor, fg_linepre.hip).  z = M^-1 r for the nc systems
// of every env: basis change along x (and z) with the fp32 MFMA GEMMs above, per-mode Thomas solve, basis change back.
// BiCGStab on C M^-1 then only has the advective part left to iterate on: RBC 512 x 128 at CFL 0.8 -- 2 iterations against 36-44
// (measured with SciPy on the oracle's matrix before the kernels were written; 4 at three times that CFL).  The reference's
// preconditioner for these solves is cuSPARSE ILU(0) (bicgstab_solver_kernel.cu:191-226), off by default (preconditionBiCG).
// ---------------------------------------------------------------------------------------------------------------------------
extern "C" int fg_set_fd_helmholtz(fg_handle s, const float* lam_host) {
    FG_REQUIRE(s && lam_host, FG_ERR_INVALID_ARG, "fg_set_fd_helmholtz: null argument");
    FG_REQUIRE(s->fd_Qx && s->fd_dct_x != 1, FG_ERR_INVALID_ARG, "fg_set_fd_helmholtz: call fg_set_fd_preconditioner first (periodic x basis)");
    FG_REQUIRE(!s->grid.fixed[0] && !s->grid.fixed[1] && s->grid.fixed[2] && s->grid.fixed[3] && (s->grid.dims == 2 || (!s->grid.fixed[4] && !s->grid.fixed[5])),
               FG_ERR_UNSUPPORTED, "fg_set_fd_helmholtz: needs PERIODIC transform axes (x, z) and FIXED y faces");
    const size_t count = (size_t)s->grid.nx * s->grid.nz;
    if (!s->fd_lam) FG_HIP_CHECK(hipMalloc(&s->fd_lam, sizeof(float) * count));
    FG_HIP_CHECK(hipMemcpy(s->fd_lam, lam_host, sizeof(float) * count, hipMemcpyHostToDevice));
    return fg_helm_alloc(s);
}

int fg_fd_helmholtz_apply(fg_state* s, int nc, const float* r, float* z, hipStream_t st) {
    const FgGrid& G = s->grid;
    const int nx = G.nx, ny = G.ny, nz = G.nz, nsys = G.B * nc;
    const long N = G.n;
    float* t1 = s->w[7];          // free in the five-kernel BiCGStab
    float* t2 = s->helm_tmp;
    GemmArgs g = {};
    g.flags = s->flags; g.dot_with = nullptr; g.strideW = 0; g.dot_acc = nullptr; g.dot_stride = 0; g.dot_ns = 1;
    // forward x: t1[rows, a] = sum_i r[rows, i] Qx[i, a]  (periodic x marked as a fast-transform axis: one real FFT per row)
    if (s->fd_dct_x == 2) {
        if (int rc = fg_fd_dct_forward(s, r, t1, st, nsys)) return rc;
    } else {
        g.A = r; g.lda = nx; g.strideA = N;
        g.B = s->fd_Qx; g.ldb = nx; g.strideB = 0; g.Bt = s->fd_QxT; g.ldbt = nx;
        g.C = t1; g.ldc = nx; g.strideC = N;
        g.M = ny * nz; g.N = nx; g.K = nx;
        if (int rc = launch_gemm(s, g, nsys, nsys, st)) return rc;
    }
    float* cur = t1;
    if (G.dims == 3) {
        g.A = s->fd_QzT; g.lda = nz; g.strideA = 0;
        g.B = t1; g.ldb = (long)ny * nx; g.strideB = N; g.Bt = nullptr; g.ldbt = 0;
        g.C = t2; g.ldc = (long)ny * nx; g.strideC = N;
        g.M = nz; g.N = ny * nx; g.K = nz;
        if (int rc = launch_gemm(s, g, nsys, nsys, st)) return rc;
        cur = t2;
    }
    if (int rc = fg_helm_apply(s, nc, cur, cur, st)) return rc;   // per-mode Thomas solve, in place
    if (G.dims == 3) {
        g.A = s->fd_Qz; g.lda = nz; g.strideA = 0;
        g.B = t2; g.ldb = (long)ny * nx; g.strideB = N; g.Bt = nullptr; g.ldbt = 0;
        g.C = t1; g.ldc = (long)ny * nx; g.strideC = N;
        g.M = nz; g.N = ny * nx; g.K = nz;
        if (int rc = launch_gemm(s, g, nsys, nsys, st)) return rc;
        cur = t1;
    }
    // inverse x: z[rows, i] = sum_a cur[rows, a] QxT[a, i]
    if (s->fd_dct_x == 2) return fg_fd_dct_inverse(s, cur, z, nullptr, nullptr, 0, 1, st, nsys);
    g.A = cur; g.lda = nx; g.strideA = N;
    g.B = s->fd_QxT; g.ldb = nx; g.strideB = 0; g.Bt = s->fd_Qx; g.ldbt = nx;
    g.C = z; g.ldc = nx; g.strideC = N;
    g.M = ny * nz; g.N = nx; g.K = nx;
    if (int rc = launch_gemm(s, g, nsys, nsys, st)) return rc;
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}
