// Fast cosine transforms for the fast-diagonalisation preconditioner (uniform FIXED x axis).
//
// On an axis with uniform cell width h and FIXED (pressure-Neumann) ends the generalised eigenvectors of the 1-D
// operator are the orthonormal DCT-II basis scaled by 1/sqrt(h) (simulation/fd_precond.py checks this against
// numpy's eigh), so the two basis changes of M^-1 are a DCT-II and a DCT-III of every grid row -- O(n log n) instead
// of the dense n x n GEMM of fg_fdprecond.hip (131 kflop per 256-cell row there, ~7 kflop here; both memory-bound
// after that).  Makhoul's mapping onto ONE n-point complex FFT per row:
//     v_j = x_{2j},  v_{n-1-j} = x_{2j+1};   V = FFT_n(v);   X_k = s_k Re(e^{-i pi k / 2n} V_k)
// and backwards  V_k = (Y_k - i Y_{n-k}) e^{+i pi k / 2n},  v = FFT^-1(V),  x_{2j} = v_j, x_{2j+1} = v_{n-1-j}.
// One wave per PAIR of rows (two real rows per complex transform), four waves per workgroup, radix-4 (+ one radix-2) Stockham stages ping-ponging between two LDS buffers;
// twiddles come from a table built in double precision on the host (fg_set_fd_fast_transform), staged in LDS per workgroup.
#include <math.h>

#include <vector>

#include "fg_internal.h"
#include "fg_cg.h"

namespace {

struct DctArgs {
    const float* src; float* dst;           // [B, rows, n] contiguous rows
    const float2* tw;                       // [n]    (cos, sin)(2 pi j / n)
    const float2* rot;                      // [n]    (cos, sin)(pi k / 2n)
    float scale0, scale;                    // forward: s_k / sqrt(h); inverse: 1 / (s_k n sqrt(h))   (k = 0 | k > 0)
    const int32_t* flags;                   // env b skipped when flags[b] != 0
    const float* dot_with; FgDacc* dot_acc; int dot_stride, dot_ns;   // inverse only: acc[b] += sum dst .* dot_with
    long env_stride; int rows;
    FgCgJudge judge;                        // forward only, optional (judge.acc != nullptr): fg_cg.h
};

// PERIODIC = false: the cosine transforms above (FIXED axis).  PERIODIC = true: the real Fourier basis of a periodic uniform axis,
//   q_0 = 1/sqrt(n),  q_k = sqrt(2/n) cos(2 pi k i / n),  q_{n-k} = sqrt(2/n) sin(2 pi k i / n)  (0 < k < n/2),  q_{n/2} = (-1)^i / sqrt(n),
// all / sqrt(h); mode m has the eigenvalue of k = min(m, n - m).  Forward: V = FFT_n(x), X_k = s_k Re V_k, X_{n-k} = -s_k Im V_k.
// Inverse: W_0 = g_0 Y_0, W_{n/2} = g_0 Y_{n/2}, W_k = g/2 (Y_k - i Y_{n-k}), W_{n-k} = conj(W_k); x = Re FFT^-1_n(W) (unnormalised).
// TWO rows per complex FFT: the rows are real, so rows a and b travel as z = a + i b through one transform and come apart by the
// symmetry of a real sequence's spectrum, V^a_k = (Z_k + conj Z_{N-k}) / 2, V^b_k = (Z_k - conj Z_{N-k}) / 2i; backwards the
// two Hermitian spectra are added as W^a + i W^b and the rows are the real and imaginary part of the result.  Half the butterflies,
// LDS traffic and wave barriers per row of the one-row-per-transform form of rounds 1-3 (identities checked in NumPy for all four
// modes).  Measured: it leaves the kernel's duration where it was (RBC 512-point rows: 10-11.6 us, half the workgroups) -- a build
// without the butterflies (-DFG_DCT_KNOCK) still takes 7 of the 9 us of the 256-point kernel: load, staging barrier, store and
// launch are the floor at these sizes, not the transform.
template <int N, bool INVERSE, bool PERIODIC = false>
__global__ __launch_bounds__(256) void k_dct_rows(DctArgs a) {
    constexpr int EPL = N / 64;             // elements per lane and row
    __shared__ float2 buf[2][4][N];
    __shared__ float2 twl[N];                // W^k = (cos, sin)(2 pi k / N), k < N
    __shared__ float red[4];
    const int b = blockIdx.y;
    if (a.flags && a.flags[b] != 0) return;
    if constexpr (!INVERSE) {
        if (a.judge.acc && fg_cg_judge(a.judge, b, blockIdx.x == 0 && threadIdx.x == 0)) return;
    }
    for (int k = threadIdx.x; k < N; k += 256) twl[k] = a.tw[k];   // visible after the barrier that follows the row staging
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = 2 * (blockIdx.x * 4 + wave), row1 = row0 + 1;
    const bool live0 = row0 < a.rows, live1 = row1 < a.rows;
    const size_t off0 = (size_t)b * a.env_stride + (size_t)(live0 ? row0 : 0) * N + lane * EPL;
    const size_t off1 = (size_t)b * a.env_stride + (size_t)(live1 ? row1 : 0) * N + lane * EPL;
    float2* x = buf[0][wave];
    float2* y = buf[1][wave];
    float xa[EPL], xb[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { xa[e] = live0 ? a.src[off0 + e] : 0.f; xb[e] = live1 ? a.src[off1 + e] : 0.f; }
    if (!INVERSE) {
        // v_j = x_2j, v_{n-1-j} = x_{2j+1}  (periodic: v = x); row a in the real, row b in the imaginary part
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = lane * EPL + e;
            const int j = PERIODIC ? i : ((i & 1) ? N - 1 - (i >> 1) : (i >> 1));
            x[j] = make_float2(xa[e], xb[e]);
        }
    } else {
        // stage both rows so that Y_{n-k} is reachable (the second buffer holds 2 N floats), then Z_k = W^a_k + i W^b_k with
        //   cosine basis:  W_k = g_k (Y_k - i Y_{n-k}) e^{+i theta_k},  Y_n := 0
        //   Fourier basis: W_0 = g_0 Y_0, W_{n/2} = g_0 Y_{n/2}, W_k = g/2 (Y_k - i Y_{n-k}), W_{n-k} = conj W_k
        float* sa = reinterpret_cast<float*>(y);
        float* sb = sa + N;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { sa[lane * EPL + e] = xa[e]; sb[lane * EPL + e] = xb[e]; }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int k = lane * EPL + e;
            float2 wa, wb;
            if (PERIODIC) {
                const int km = (N - k) & (N - 1);
                const float ya = sa[k], yam = sa[km], yb = sb[k], ybm = sb[km];
                if (k == 0 || k == N / 2) { wa = make_float2(a.scale0 * ya, 0.f); wb = make_float2(a.scale0 * yb, 0.f); }
                else if (k < N / 2) { wa = make_float2(0.5f * a.scale * ya, -0.5f * a.scale * yam); wb = make_float2(0.5f * a.scale * yb, -0.5f * a.scale * ybm); }
                else { wa = make_float2(0.5f * a.scale * yam, 0.5f * a.scale * ya); wb = make_float2(0.5f * a.scale * ybm, 0.5f * a.scale * yb); }
            } else {
                const float ya = sa[k], yam = (k == 0) ? 0.f : sa[N - k], yb = sb[k], ybm = (k == 0) ? 0.f : sb[N - k];
                const float2 r = a.rot[k];
                const float g = (k == 0) ? a.scale0 : a.scale;
                wa = make_float2(g * (ya * r.x + yam * r.y), g * (ya * r.y - yam * r.x));
                wb = make_float2(g * (yb * r.x + ybm * r.y), g * (yb * r.y - ybm * r.x));
            }
            x[k] = make_float2(wa.x - wb.y, wa.y + wb.x);
        }
    }
    __syncthreads();
    // Stockham autosort, radix 4 while the remaining length allows it, then one radix-2 stage (N = 128, 512): half the stages,
    // LDS traffic and barriers of the radix-2 form (13 -> 8 us per transform of 64 x 128 rows), twiddles W^k from an LDS table.
    // Butterfly t of a stage with stride s: q = t mod s, ps = t - q; inputs x[t + j N/4], outputs y[4 ps + q + j s] with
    //   b1 = (a0 + rot a1 - a2 - rot a3) W^ps,  b2 = (a0 - a1 + a2 - a3) W^2ps,  b3 = (a0 - rot a1 - a2 + rot a3) W^3ps,
    // rot = -i (forward) / +i (inverse)  (formulas checked against numpy.fft for all four lengths).
    // every wave works on its own row: between stages only the lanes of ONE wave exchange data through LDS, whose operations a wave
    // issues in order -- a wave-level barrier (no instruction, a scheduling fence for the compiler) replaces the block barrier and
    // lets the four waves of a workgroup drift apart
    auto wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto cmul = [](float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); };
    auto twid = [&](int k) { float2 w = twl[k & (N - 1)]; if (!INVERSE) w.y = -w.y; return w; };
    int sft = 0;   // log2 of the stride s
    int rem = N;
#pragma unroll
    for (int st = 0; st < 5; ++st) {
#ifdef FG_DCT_KNOCK
        break;   // timing floor of the kernel without its butterflies (profiling build only)
#endif
        if (rem % 4 != 0) break;
        const int sm = (1 << sft) - 1;
        for (int t = lane; t < N / 4; t += 64) {
            const int q = t & sm, ps = t - q;
            const float2 a0 = x[t], a1 = x[t + N / 4], a2 = x[t + N / 2], a3 = x[t + 3 * N / 4];
            // rot * a = (-+) i a:  forward rot a = (a.y, -a.x), inverse rot a = (-a.y, a.x)
            const float2 r1 = INVERSE ? make_float2(-a1.y, a1.x) : make_float2(a1.y, -a1.x);
            const float2 r3 = INVERSE ? make_float2(-a3.y, a3.x) : make_float2(a3.y, -a3.x);
            const float2 b0 = make_float2(a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y);
            const float2 c1 = make_float2(a0.x + r1.x - a2.x - r3.x, a0.y + r1.y - a2.y - r3.y);
            const float2 c2 = make_float2(a0.x - a1.x + a2.x - a3.x, a0.y - a1.y + a2.y - a3.y);
            const float2 c3 = make_float2(a0.x - r1.x - a2.x + r3.x, a0.y - r1.y - a2.y + r3.y);
            float2* o = y + 4 * ps + q;
            o[0] = b0;
            o[1 << sft] = cmul(c1, twid(ps));
            o[2 << sft] = cmul(c2, twid(2 * ps));
            o[3 << sft] = cmul(c3, twid(3 * ps));
        }
        wave_sync();
        float2* tmp = x; x = y; y = tmp;
        sft += 2; rem /= 4;
    }
    if (rem == 2) {
        const int sm = (1 << sft) - 1;
        for (int t = lane; t < N / 2; t += 64) {
            const int q = t & sm, ps = t - q;
            const float2 u = x[t], v = x[t + N / 2];
            y[2 * ps + q] = make_float2(u.x + v.x, u.y + v.y);
            y[2 * ps + q + (1 << sft)] = cmul(make_float2(u.x - v.x, u.y - v.y), twid(ps));
        }
        wave_sync();
        float2* tmp = x; x = y; y = tmp;
    }
    float oa[EPL], ob[EPL];
    if (!INVERSE) {
        // split the spectrum of z = a + i b:  V^a = (Z_k + conj Z_{n-k}) / 2,  V^b = (Z_k - conj Z_{n-k}) / 2i
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int m = lane * EPL + e;
            const int k = PERIODIC ? (m <= N / 2 ? m : N - m) : m;
            const float2 Z = x[k], M = x[(N - k) & (N - 1)];
            const float2 Va = make_float2(0.5f * (Z.x + M.x), 0.5f * (Z.y - M.y));
            const float2 Vb = make_float2(0.5f * (Z.y + M.y), -0.5f * (Z.x - M.x));
            if (PERIODIC) {   // X_k = s_k Re V_k, X_{n-k} = -s_k Im V_k
                oa[e] = (m == 0 || m == N / 2) ? a.scale0 * Va.x : (m < N / 2 ? a.scale * Va.x : -a.scale * Va.y);
                ob[e] = (m == 0 || m == N / 2) ? a.scale0 * Vb.x : (m < N / 2 ? a.scale * Vb.x : -a.scale * Vb.y);
            } else {          // X_k = s_k Re(e^{-i theta_k} V_k)
                const float2 r = a.rot[k];
                const float g = (k == 0) ? a.scale0 : a.scale;
                oa[e] = g * (r.x * Va.x + r.y * Va.y);
                ob[e] = g * (r.x * Vb.x + r.y * Vb.y);
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = lane * EPL + e;
            const float2 v = x[PERIODIC ? i : ((i & 1) ? N - 1 - (i >> 1) : (i >> 1))];
            oa[e] = v.x; ob[e] = v.y;
        }
    }
    float dot = 0.f;
    if (live0) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) a.dst[off0 + e] = oa[e];
        if (INVERSE && a.dot_with) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) dot += oa[e] * a.dot_with[off0 + e];
        }
    }
    if (live1) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) a.dst[off1 + e] = ob[e];
        if (INVERSE && a.dot_with) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) dot += ob[e] * a.dot_with[off1 + e];
        }
    }
    if (INVERSE && a.dot_with) {
        float part[1] = {dot};
        fg_block_sum<1>(part, red);
        if (threadIdx.x == 0)
            acc_add(a.dot_acc + (size_t)b * a.dot_stride + (blockIdx.x & (unsigned)(a.dot_ns - 1)), (double)part[0]);
    }
}

template <bool INVERSE, bool PERIODIC>
int launch_dct_mode(const fg_state* s, int n, const DctArgs& a, int slot, int batch, hipStream_t st) {
    const dim3 grid(((a.rows + 1) / 2 + 3) / 4, batch);   // a wave takes a PAIR of rows, four waves per workgroup
    switch (n) {
        case 64: FG_LAUNCH_P(s, slot, (k_dct_rows<64, INVERSE, PERIODIC>), grid, dim3(256), 0, st, a); break;
        case 128: FG_LAUNCH_P(s, slot, (k_dct_rows<128, INVERSE, PERIODIC>), grid, dim3(256), 0, st, a); break;
        case 256: FG_LAUNCH_P(s, slot, (k_dct_rows<256, INVERSE, PERIODIC>), grid, dim3(256), 0, st, a); break;
        case 512: FG_LAUNCH_P(s, slot, (k_dct_rows<512, INVERSE, PERIODIC>), grid, dim3(256), 0, st, a); break;
        default: fg_set_error("fast transform: unsupported length"); return FG_ERR_UNSUPPORTED;
    }
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}
template <bool INVERSE>
int launch_dct(const fg_state* s, int n, const DctArgs& a, int slot, int batch, hipStream_t st) {
    return s->fd_dct_x == 2 ? launch_dct_mode<INVERSE, true>(s, n, a, slot, batch, st) : launch_dct_mode<INVERSE, false>(s, n, a, slot, batch, st);
}

}  // namespace

bool fg_fd_dct_supported(int n) { return n == 64 || n == 128 || n == 256 || n == 512; }

// x-axis transforms of fg_fd_apply when the axis is marked as DCT (fd_dct_x): rows = ny * nz per env, length nx.
int fg_fd_dct_forward(fg_state* s, const float* r, float* out, hipStream_t st, int batch, const FgCgJudge* judge) {
    const FgGrid& G = s->grid;
    if (batch <= 0) batch = G.B;
    DctArgs a = {};
    a.src = r; a.dst = out; a.tw = s->fd_dct_tw; a.rot = s->fd_dct_rot;
    a.scale0 = s->fd_dct_fwd[0]; a.scale = s->fd_dct_fwd[1];
    a.flags = s->flags; a.env_stride = G.n; a.rows = G.ny * G.nz;
    if (judge) a.judge = *judge;
    // per env: the row read + written; ~5 n log2 n flops per row
    const int slot = fg_prof_slot(s, FG_PK_DCT, s->flags, batch, 8.0 * G.n, 5.0 * G.n * log2((double)G.nx), st);
    return launch_dct<false>(s, G.nx, a, slot, batch, st);
}
int fg_fd_dct_inverse(fg_state* s, const float* u, float* z, const float* dot_with, FgDacc* dot_acc, int dot_stride,
                      int dot_ns, hipStream_t st, int batch) {
    const FgGrid& G = s->grid;
    if (batch <= 0) batch = G.B;
    DctArgs a = {};
    a.src = u; a.dst = z; a.tw = s->fd_dct_tw; a.rot = s->fd_dct_rot;
    a.scale0 = s->fd_dct_inv[0]; a.scale = s->fd_dct_inv[1];
    a.flags = s->flags; a.env_stride = G.n; a.rows = G.ny * G.nz;
    a.dot_with = dot_acc ? dot_with : nullptr; a.dot_acc = dot_acc; a.dot_stride = dot_stride; a.dot_ns = dot_ns;
    const int slot = fg_prof_slot(s, FG_PK_DCT, s->flags, batch, (dot_acc ? 12.0 : 8.0) * G.n, 5.0 * G.n * log2((double)G.nx), st);
    return launch_dct<true>(s, G.nx, a, slot, batch, st);
}

extern "C" int fg_set_fd_fast_transform(fg_handle s, int axis, float cell_width) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    FG_REQUIRE(axis == 0, FG_ERR_UNSUPPORTED, "fast transforms are built for the x axis only");
    const int n = s->grid.nx;
    FG_REQUIRE(fg_fd_dct_supported(n), FG_ERR_UNSUPPORTED, "fast transform needs nx in {64, 128, 256, 512}");
    FG_REQUIRE(cell_width > 0.f && (s->grid.fixed[0] != 0) == (s->grid.fixed[1] != 0), FG_ERR_INVALID_ARG,
               "fast transform needs a uniform positive cell width");
    const bool periodic = !s->grid.fixed[0];   // FIXED x: cosine basis (DCT-II / III); PERIODIC x: real Fourier basis
    std::vector<float2> tw(n), rot(n);
    for (int j = 0; j < n; ++j) tw[j] = make_float2((float)cos(2.0 * M_PI * j / n), (float)sin(2.0 * M_PI * j / n));
    for (int k = 0; k < n; ++k) rot[k] = make_float2((float)cos(M_PI * k / (2.0 * n)), (float)sin(M_PI * k / (2.0 * n)));
    if (s->fd_dct_tw) { (void)hipFree(s->fd_dct_tw); (void)hipFree(s->fd_dct_rot); }
    FG_HIP_CHECK(hipMalloc(&s->fd_dct_tw, sizeof(float2) * n));
    FG_HIP_CHECK(hipMalloc(&s->fd_dct_rot, sizeof(float2) * n));
    FG_HIP_CHECK(hipMemcpy(s->fd_dct_tw, tw.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->fd_dct_rot, rot.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
    const double rs = 1.0 / sqrt((double)cell_width);
    s->fd_dct_fwd[0] = (float)(sqrt(1.0 / n) * rs); s->fd_dct_fwd[1] = (float)(sqrt(2.0 / n) * rs);
    if (periodic) {   // x = Re FFT^-1(W) with the basis factors inside W: g_0 = rs / sqrt(n), g = rs sqrt(2 / n)
        s->fd_dct_inv[0] = (float)(rs * sqrt(1.0 / n)); s->fd_dct_inv[1] = (float)(rs * sqrt(2.0 / n));
    } else {
        s->fd_dct_inv[0] = (float)(rs / (sqrt(1.0 / n) * n)); s->fd_dct_inv[1] = (float)(rs / (sqrt(2.0 / n) * n));
    }
    s->fd_dct_x = periodic ? 2 : 1;
    // eigenvalues of the 1-D operator in the order the transform produces the modes (fd_precond.cosine_basis / fourier_basis): the
    // row-mean preconditioner synthesises its per-env tridiagonal systems from them (fg_fdprecond.hip k_fd_rowmean_factor)
    std::vector<float> lam(n);
    for (int m = 0; m < n; ++m) {
        const int k = periodic ? (m <= n / 2 ? m : n - m) : m;
        const double ang = periodic ? 2.0 * M_PI * k / n : M_PI * k / n;
        lam[m] = (float)(-(2.0 - 2.0 * cos(ang)) / ((double)cell_width * (double)cell_width));
    }
    if (s->fd_lam_x) (void)hipFree(s->fd_lam_x);
    FG_HIP_CHECK(hipMalloc(&s->fd_lam_x, sizeof(float) * n));
    FG_HIP_CHECK(hipMemcpy(s->fd_lam_x, lam.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    return FG_OK;
}
