// Fast-diagonalisation preconditioner for the pressure CG (see simulation/fd_precond.py for the
// maths): z = M^-1 r with M = the constant-coefficient (A = 1) pressure operator, applied as
//     r^ = (Qx^T (x) Qz^T) r  ->  one tridiagonal solve along y per mode  ->  z = (Qx (x) Qz) u .
// The two basis changes are genuinely GEMM-shaped (tall-skinny field matrix times a small dense
// eigenbasis) and run on the matrix cores with the exact-fp32 MFMA v_mfma_f32_32x32x2_f32 (gfx950:
// 64 cycles / 4096 MACs per wave, bitwise an fmaf chain, cdna_hip_programming.md section 3); the
// tridiagonal sweep streams the per-mode LU factors precomputed on the host.
#include "fg_internal.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;


struct GemmArgs {
    const float* A; long lda; long strideA;   // [M x K] row-major, per batch (stride 0 = shared)
    const float* B; long ldb; long strideB;   // [K x N]
    float* C; long ldc; long strideC;         // [M x N]
    int M, N, K;
    const int32_t* flags;                     // batch b is skipped when flags[b] != 0
    const float* dot_with; long strideW;      // optional: acc[b] += sum C .* W  (W laid out like C)
    double* dot_acc; int dot_stride; int dot_ns;  // slotted accumulator: dot_acc[b * dot_stride + (block & (ns-1))]
};

// C = A * B, fp32 in / fp32 accumulate on MFMA 32x32x2.  256 threads = 4 waves in a 2 x 2 arrangement,
// each wave owns TM x TN MFMA tiles of 32 x 32 (TM = TN = 2: 128 x 128 block tile, 64 accumulator
// registers; TM = TN = 1: 64 x 64 block tile for launches that would otherwise leave CUs idle).
template <int TM, int TN, int BK>
__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs g) {
    // BK = K-depth staged per barrier pair.  The small launches of the batched 2-D case are latency-bound
    // (one L2 round trip per K-step, only 8 MFMAs per wave to cover it), so they use BK = 64: 4x fewer steps,
    // 4x more bytes in flight per step.
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int LDA_S = BM + 4, LDB_S = BN + 4;  // LDS row pitch; +4 floats keeps 16-B alignment, breaks conflicts
    constexpr int PA = BM * BK / 256, PB = BN * BK / 256;  // floats staged per thread (8 or 4)
    const int b = blockIdx.z;
    if (g.flags && g.flags[b] != 0) return;
    __shared__ __attribute__((aligned(16))) float As[BK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB_S];
    const float* __restrict__ A = g.A + (size_t)b * g.strideA;
    const float* __restrict__ B = g.B + (size_t)b * g.strideB;
    float* __restrict__ C = g.C + (size_t)b * g.strideC;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
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
            atomicAdd(g.dot_acc + (size_t)b * g.dot_stride + ((blockIdx.x + blockIdx.y * gridDim.x) & (unsigned)(g.dot_ns - 1)),
                      (double)part[0]);
    }
}

// Thomas sweep along y for every mode: forward y_j = (b_j - l_j y_{j-1}) inv_j, backward
// x_j = y_j - c'_j x_{j+1}.  One thread per (env, z-mode, x-mode); x-mode is the fastest index so every
// step is a coalesced row access.  The intermediate y_j stay in LDS ([j][lane], conflict-free), so the global
// traffic is: read b, inv, c' once (prefetched TRI_CH rows ahead of the recurrence), write x once -- no global
// store sits between a load and its use (the in-place global variant spent ~170 ns per recurrence step).
constexpr int TRI_CH = 16;
// Row indices are CLAMPED instead of branched on: every load of a chunk is straight-line code, so the compiler
// can keep the next chunk in flight with counted s_waitcnt (the branchy form compiled to 36 x vmcnt(0) drains).
__global__ __launch_bounds__(64) void k_tridiag_y(float* __restrict__ x, const float* __restrict__ inv,
                                                   const float* __restrict__ cp, const float* __restrict__ lower,
                                                   const int32_t* __restrict__ flags, int nx, int ny, int nz) {
    extern __shared__ __attribute__((aligned(16))) float ybuf[];  // [ny_padded][64]
    const int b = blockIdx.y;
    if (flags && flags[b] != 0) return;
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
    float cx[TRI_CH], ci[TRI_CH], cl[TRI_CH], nxv[TRI_CH], niv[TRI_CH], nlv[TRI_CH];
    // ---- forward elimination: y_j = (b_j - l_j y_{j-1}) inv_j ; y to LDS
#pragma unroll
    for (int q = 0; q < TRI_CH; ++q) {
        const int j = min(q, last);
        cx[q] = xb[(size_t)j * nx]; ci[q] = iv[(size_t)j * nx]; cl[q] = lower[j];
    }
    float prev = 0.f;
    for (int j0 = 0; j0 < ny; j0 += TRI_CH) {
#pragma unroll
        for (int q = 0; q < TRI_CH; ++q) {
            const int j = min(j0 + TRI_CH + q, last);
            nxv[q] = xb[(size_t)j * nx]; niv[q] = iv[(size_t)j * nx]; nlv[q] = lower[j];
        }
#pragma unroll
        for (int q = 0; q < TRI_CH; ++q) {
            const float v = (cx[q] - cl[q] * prev) * ci[q];
            prev = (j0 + q <= last) ? v : prev;     // rows past the end keep the last value (select, no branch)
            ybuf[(j0 + q) * 64 + lane] = prev;      // LDS is padded to a multiple of TRI_CH rows
        }
#pragma unroll
        for (int q = 0; q < TRI_CH; ++q) { cx[q] = nxv[q]; ci[q] = niv[q]; cl[q] = nlv[q]; }
    }
    // ---- back substitution: x_last = y_last ; x_j = y_j - c'_j x_{j+1}
    // (c'_last = 0 in the factor table, so starting one row early with prev = 0 reproduces x_last = y_last)
    prev = 0.f;
    const int top = ((ny + TRI_CH - 1) / TRI_CH) * TRI_CH - TRI_CH;  // first row of the last chunk
#pragma unroll
    for (int q = 0; q < TRI_CH; ++q) ci[q] = cpp[(size_t)min(top + q, last) * nx];
    for (int j0 = top; j0 >= 0; j0 -= TRI_CH) {
#pragma unroll
        for (int q = 0; q < TRI_CH; ++q) niv[q] = cpp[(size_t)max(j0 - TRI_CH + q, 0) * nx];
#pragma unroll
        for (int q = 0; q < TRI_CH; ++q) cx[q] = ybuf[(j0 + q) * 64 + lane];
#pragma unroll
        for (int q = TRI_CH - 1; q >= 0; --q) {
            const int j = j0 + q;
            const float v = cx[q] - ci[q] * prev;
            prev = (j <= last) ? v : prev;
            if (live && j <= last) xb[(size_t)j * nx] = prev;
        }
#pragma unroll
        for (int q = 0; q < TRI_CH; ++q) ci[q] = niv[q];
    }
}

}  // namespace

static int launch_gemm(const GemmArgs& g, int batch, hipStream_t st) {
    const long big_blocks = (long)((g.N + 127) / 128) * ((g.M + 127) / 128) * batch;
    if (big_blocks >= 512) {  // >= 2 workgroups per CU with the 128 x 128 tile
        dim3 grid((g.N + 127) / 128, (g.M + 127) / 128, batch);
        hipLaunchKernelGGL((k_gemm_f32<2, 2, 16>), grid, dim3(256), 0, st, g);
    } else {
        dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, batch);
        hipLaunchKernelGGL((k_gemm_f32<1, 1, 64>), grid, dim3(256), 0, st, g);
    }
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

// z = M^-1 r for all envs with flags == 0; optionally rz_acc[b * rz_stride] += r . z
int fg_fd_apply(fg_state* s, const float* r, float* z, double* rz_acc, int rz_stride, int rz_ns, hipStream_t st) {
    const FgGrid& G = s->grid;
    const int nx = G.nx, ny = G.ny, nz = G.nz, B = G.B;
    const long N = G.n;
    float* t1 = s->w[3];
    float* t2 = s->w[4];
    GemmArgs g;
    // forward x: t1[rows, a] = sum_i r[rows, i] Qx[i, a]
    g.A = r; g.lda = nx; g.strideA = N;
    g.B = s->fd_Qx; g.ldb = nx; g.strideB = 0;
    g.C = t1; g.ldc = nx; g.strideC = N;
    g.M = ny * nz; g.N = nx; g.K = nx;
    g.flags = s->flags; g.dot_with = nullptr; g.strideW = 0; g.dot_acc = nullptr; g.dot_stride = 0; g.dot_ns = 1;
    if (int rc = launch_gemm(g, B, st)) return rc;
    float* cur = t1;
    if (G.dims == 3) {
        // forward z: t2[c, m] = sum_k QzT[c, k] t1[k, m]   (m over ny*nx)
        g.A = s->fd_QzT; g.lda = nz; g.strideA = 0;
        g.B = t1; g.ldb = (long)ny * nx; g.strideB = N;
        g.C = t2; g.ldc = (long)ny * nx; g.strideC = N;
        g.M = nz; g.N = ny * nx; g.K = nz;
        if (int rc = launch_gemm(g, B, st)) return rc;
        cur = t2;
    }
    {
        // one wave per workgroup: at batch 64 x 256 modes this spreads the sweep over 256 CUs instead of 64 (the
        // sweep is bound by per-CU load/store throughput, not by arithmetic)
        dim3 grid((nx * nz + 63) / 64, B);
        hipLaunchKernelGGL(k_tridiag_y, grid, dim3(64), (size_t)((ny + 15) / 16 * 16) * 64 * sizeof(float), st, cur, s->fd_inv, s->fd_cp, s->fd_lower, s->flags, nx, ny, nz);
    }
    if (G.dims == 3) {
        // inverse z: t1[k, m] = sum_c Qz[k, c] t2[c, m]
        g.A = s->fd_Qz; g.lda = nz; g.strideA = 0;
        g.B = t2; g.ldb = (long)ny * nx; g.strideB = N;
        g.C = t1; g.ldc = (long)ny * nx; g.strideC = N;
        g.M = nz; g.N = ny * nx; g.K = nz;
        if (int rc = launch_gemm(g, B, st)) return rc;
        cur = t1;
    }
    // inverse x: z[rows, i] = sum_a cur[rows, a] QxT[a, i], fused r.z
    g.A = cur; g.lda = nx; g.strideA = N;
    g.B = s->fd_QxT; g.ldb = nx; g.strideB = 0;
    g.C = z; g.ldc = nx; g.strideC = N;
    g.M = ny * nz; g.N = nx; g.K = nx;
    g.dot_with = rz_acc ? r : nullptr; g.strideW = N; g.dot_acc = rz_acc; g.dot_stride = rz_stride; g.dot_ns = rz_ns;
    if (int rc = launch_gemm(g, B, st)) return rc;
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}
