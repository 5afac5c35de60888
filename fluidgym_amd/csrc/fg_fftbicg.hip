// Fused row kernels of the Helmholtz-preconditioned BiCGStab on 2-D grids with a periodic, uniform x axis (RBC 512 x 128 = BASELINE
// config 3: the passive-scalar and velocity systems of rbc_env_base.py:285-329).  Replaces bicgstabSolveGPU with its preconditioned
// branch (bicgstab_solver_kernel.cu:63-411; preconditioner :191-226, 288-293) on the matrix of PISO_build_matrix
// (PISO_multiblock_cuda_kernel.cu:3616-3880); right preconditioning by M = I/dt - nu Laplacian (fg_fd_helmholtz_apply, fg_fdprecond.hip).
//
// Rounds 3-4 ran an iteration as ELEVEN launches: k_bicg_p | forward transform, y-line solve, inverse transform | k_bicg_v |
// k_bicg_s | forward, line, inverse | k_bicg_t | k_bicg_x -- 117 launches of ~9 us per PISO step of the RBC env.  The transforms are
// row-local, so the vector updates ride in the loaders of the forward transforms and the matrix is applied to the rows a workgroup
// has just transformed back (one halo row above and below, transformed again by a fifth wave), and rho_{i+1} = rw.s - omega rw.t
// comes from dot products of the t kernel as in the two-kernel form (fg_bicgstab.hip) -- SIX launches per iteration:
//
//   FS(i):  alpha_i = rho_i / rw.v_i;  s = r - alpha v;  s.s, rw.s;  u = Qx^T s                       (k_fbicg_fwd<KIND_S>)
//   L:      [s converged?]  per-mode Thomas solve of the Helmholtz operator                          (k_line_apply_y, fg_linepre.hip)
//   IT(i):  s^ = Qx u;  t = C s^;  t.s, t.t, rw.t                                                     (k_fbicg_inv<KIND_T>)
//   FP(i+1): omega, beta;  x += alpha p^ + omega s^;  r = s - omega t;  r.r;  p = r + beta (p - omega v);  u = Qx^T p   (k_fbicg_fwd<KIND_P>)
//   L
//   IV(i+1): p^ = Qx u;  v = C p^;  rw.v                                                              (k_fbicg_inv<KIND_V>)
//
// Decisions (convergence on r / on s, breakdown restarts, the per-system scalars) are those of the two-kernel form: fg_bicgf_decide_a
// / _b of fg_bicg.h on the same accumulator names, taken by every workgroup from the same words.  s lives in the r buffer (as in the
// five-kernel form); p and v need no second buffer (they are only used row-locally here).
#include "fg_internal.h"
#include "fg_bicg.h"
#include "fg_fftrow.h"
#include "fg_fftbicg.h"

#if !FG_F64
namespace {

using fgfft::Map;

struct FbicgArgs {
    FgGrid g; BicgPtrs q;
    float* t1;                 // [nsys][N] transformed field, in / out of the line kernel
    const float2* tw;
    float fs0, fs, is0, is;    // forward / inverse scales of the real Fourier basis
    int it, fold, rows;
};

__device__ __forceinline__ float pick3(const float (&v)[3], int c) { return c == 0 ? v[0] : (c == 1 ? v[1] : v[2]); }

// ---------------------------------------------------------------------------------------------------------------------------
// Forward kernels.  One wave per PAIR of rows of ONE system, four waves per workgroup; grid (ceil(rows / 8), nsys).
// ---------------------------------------------------------------------------------------------------------------------------
template <int N, int KIND>     // KIND 0: FS (s = r - alpha v)   1: FP (x, r, p updates)
__global__ __launch_bounds__(256) void k_fbicg_fwd(FbicgArgs a) {
    constexpr int EPL = N / 64;
    __shared__ __attribute__((aligned(16))) float2 buf[2][4][N];
    __shared__ __attribute__((aligned(16))) float2 twl[N];
    __shared__ float red[8];
    const BicgPtrs& q = a.q;
    const int sys = blockIdx.y, b = sys / q.nc, comp = sys - b * q.nc;
    const bool leader = threadIdx.x == 0 && blockIdx.x == 0 && comp == 0;
    const int it = a.it, e = it & 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = 2 * (blockIdx.x * 4 + wave), row1 = row0 + 1;
    const bool live0 = row0 < a.rows, live1 = row1 < a.rows;
    const size_t vb = (size_t)sys * a.g.n;
    const size_t o0 = vb + (size_t)(live0 ? row0 : 0) * N, o1 = vb + (size_t)(live1 ? row1 : 0) * N;
    float fa[EPL], fb[EPL];        // the rows to transform
    float part[2] = {0.f, 0.f};
    bool add[2] = {false, false};
    int name[2] = {0, 0};
    if constexpr (KIND == 0) {
        const BicgDecB D = fg_bicgf_decide_b(a.g, q, b, it, leader);
        const bool work = comp == 0 ? D.work[0] : (comp == 1 ? D.work[1] : D.work[2]);
        if (!work) return;
        const float alpha = pick3(D.alpha, comp);
        const float* __restrict__ r_src = (a.fold && it == 0) ? q.rhs : q.r;      // folded start: r_0 is the right-hand side
        float va[EPL], vb_[EPL], wa[EPL], wb[EPL];
        fgfft::load_row<N>(r_src + o0, lane, fa); fgfft::load_row<N>(r_src + o1, lane, fb);
        fgfft::load_row<N>(q.v + o0, lane, va); fgfft::load_row<N>(q.v + o1, lane, vb_);
        fgfft::load_row<N>(q.rw + o0, lane, wa); fgfft::load_row<N>(q.rw + o1, lane, wb);
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            fa[k] -= alpha * va[k]; fb[k] -= alpha * vb_[k];
            if (!live0) fa[k] = 0.f;
            if (!live1) fb[k] = 0.f;
            part[0] += fa[k] * fa[k] + fb[k] * fb[k];
            part[1] += (live0 ? wa[k] * fa[k] : 0.f) + (live1 ? wb[k] * fb[k] : 0.f);
        }
        if (live0) fgfft::store_row<N>(q.r + o0, lane, fa);
        if (live1) fgfft::store_row<N>(q.r + o1, lane, fb);
        add[0] = add[1] = true; name[0] = F_SS + e; name[1] = F_RS + e;
    } else {
        const BicgDecA D = fg_bicgf_decide_a(a.g, q, b, it, leader, a.fold != 0);
        const int mode = comp == 0 ? D.mode[0] : (comp == 1 ? D.mode[1] : D.mode[2]);
        if (mode == 0) return;
        const float al = pick3(D.alpha, comp), om = pick3(D.omega, comp), be = pick3(D.beta, comp);
        const bool restart = comp == 0 ? D.restart[0] : (comp == 1 ? D.restart[1] : D.restart[2]);
        if (mode == 2) {      // converged on s: x += alpha p^ and the system is done (bicgstab_solver_kernel.cu:305-329)
            float xa[EPL], xb[EPL], pa[EPL], pb[EPL];
            fgfft::load_row<N>(q.x + o0, lane, xa); fgfft::load_row<N>(q.x + o1, lane, xb);
            fgfft::load_row<N>(q.mp + o0, lane, pa); fgfft::load_row<N>(q.mp + o1, lane, pb);
#pragma unroll
            for (int k = 0; k < EPL; ++k) { xa[k] += al * pa[k]; xb[k] += al * pb[k]; }
            if (live0) fgfft::store_row<N>(q.x + o0, lane, xa);
            if (live1) fgfft::store_row<N>(q.x + o1, lane, xb);
            return;
        }
        if (mode == 3) {      // first iteration: p_0 = r_0 (the right-hand side itself when the start vector is zero)
            const float* __restrict__ p0 = a.fold ? q.rhs : q.p;
            fgfft::load_row<N>(p0 + o0, lane, fa); fgfft::load_row<N>(p0 + o1, lane, fb);
#pragma unroll
            for (int k = 0; k < EPL; ++k) { if (!live0) fa[k] = 0.f; if (!live1) fb[k] = 0.f; }
            if (a.fold) {     // rw = r_0 = rhs, x_0 = 0, p_0 stored, r.r: what the init kernel does otherwise
                float z0[EPL];
#pragma unroll
                for (int k = 0; k < EPL; ++k) { z0[k] = 0.f; part[0] += fa[k] * fa[k] + fb[k] * fb[k]; }
                if (live0) { fgfft::store_row<N>(q.rw + o0, lane, fa); fgfft::store_row<N>(q.p + o0, lane, fa); fgfft::store_row<N>(q.x + o0, lane, z0); }
                if (live1) { fgfft::store_row<N>(q.rw + o1, lane, fb); fgfft::store_row<N>(q.p + o1, lane, fb); fgfft::store_row<N>(q.x + o1, lane, z0); }
                add[0] = true; name[0] = F_RR + e;
            }
        } else {
            float xa[EPL], xb[EPL], ra[EPL], rb[EPL];
            {
                float pa[EPL], pb[EPL], sa[EPL], sb[EPL], ta[EPL], tb[EPL];
                fgfft::load_row<N>(q.x + o0, lane, xa); fgfft::load_row<N>(q.x + o1, lane, xb);
                fgfft::load_row<N>(q.mp + o0, lane, pa); fgfft::load_row<N>(q.mp + o1, lane, pb);
                fgfft::load_row<N>(q.ms + o0, lane, sa); fgfft::load_row<N>(q.ms + o1, lane, sb);
                fgfft::load_row<N>(q.r + o0, lane, ra); fgfft::load_row<N>(q.r + o1, lane, rb);      // s
                fgfft::load_row<N>(q.t + o0, lane, ta); fgfft::load_row<N>(q.t + o1, lane, tb);
#pragma unroll
                for (int k = 0; k < EPL; ++k) {
                    xa[k] += al * pa[k] + om * sa[k]; xb[k] += al * pb[k] + om * sb[k];
                    ra[k] -= om * ta[k]; rb[k] -= om * tb[k];
                    if (!live0) ra[k] = 0.f;
                    if (!live1) rb[k] = 0.f;
                    part[0] += ra[k] * ra[k] + rb[k] * rb[k];
                }
            }
            if (live0) { fgfft::store_row<N>(q.x + o0, lane, xa); fgfft::store_row<N>(q.r + o0, lane, ra); }
            if (live1) { fgfft::store_row<N>(q.x + o1, lane, xb); fgfft::store_row<N>(q.r + o1, lane, rb); }
            if (restart) {    // breakdown restart: rw = p = r, rho = r.r (the decision parked a NaN in the rho slot)
#pragma unroll
                for (int k = 0; k < EPL; ++k) { fa[k] = ra[k]; fb[k] = rb[k]; }
                if (live0) fgfft::store_row<N>(q.rw + o0, lane, ra);
                if (live1) fgfft::store_row<N>(q.rw + o1, lane, rb);
            } else {
                float pa[EPL], pb[EPL], va[EPL], vb_[EPL];
                fgfft::load_row<N>(q.p + o0, lane, pa); fgfft::load_row<N>(q.p + o1, lane, pb);
                fgfft::load_row<N>(q.v + o0, lane, va); fgfft::load_row<N>(q.v + o1, lane, vb_);
#pragma unroll
                for (int k = 0; k < EPL; ++k) {
                    fa[k] = live0 ? ra[k] + be * (pa[k] - om * va[k]) : 0.f;
                    fb[k] = live1 ? rb[k] + be * (pb[k] - om * vb_[k]) : 0.f;
                }
            }
            if (live0) fgfft::store_row<N>(q.p + o0, lane, fa);
            if (live1) fgfft::store_row<N>(q.p + o1, lane, fb);
            add[0] = true; name[0] = F_RR + e;
        }
    }
    for (int k = threadIdx.x; k < N; k += 256) twl[k] = a.tw[k];
    __syncthreads();
    float oa[EPL], ob[EPL];
    fgfft::forward_rows<N, true>(fa, fb, oa, ob, buf[0][wave], buf[1][wave], twl, nullptr, fgfft::Scales{a.fs0, a.fs}, lane);
    if (live0) fgfft::store_row<N>(a.t1 + o0, lane, oa);
    if (live1) fgfft::store_row<N>(a.t1 + o1, lane, ob);
    const float tot = fg_block_sum_lanes<2>(part, red);
    if (threadIdx.x < 2 && add[threadIdx.x]) acc_add(q.acc + ((size_t)sys * FG_ACC_DOUBLES + name[threadIdx.x]), (double)tot);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Inverse kernels: p^ / s^ = Qx u of eight rows (+ the halo pair), the advection-diffusion matrix on them, the dot products.
// Five waves; grid (ceil(rows / 8), nsys).  The matrix rows are fg_spmv's (fg_bicgstab.hip): y = d x_c + sum_f o_f x_{N_f}, the
// off-diagonal of a prescribed face being zero in the assembled matrix (k_adv_build).
// ---------------------------------------------------------------------------------------------------------------------------
template <int N, int KIND>     // KIND 0: IT (t = C s^; t.s, t.t, rw.t)   1: IV (v = C p^; rw.v)
__global__ __launch_bounds__(320) void k_fbicg_inv(FbicgArgs a) {
    constexpr int EPL = N / 64;
    using M = Map<N>;
    constexpr int VW = M::VW;
    __shared__ __attribute__((aligned(16))) float2 buf[2][5][N];
    __shared__ __attribute__((aligned(16))) float2 twl[N];
    __shared__ float red[15];
    const BicgPtrs& q = a.q;
    const int sys = blockIdx.y, b = sys / q.nc;
    if (flag_ld(q.flags + sys) != 0) return;      // (stable in this launch: flags are written by the forward kernels and the line kernel)
    const int it = a.it, e = it & 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < N; k += 320) twl[k] = a.tw[k];
    const int j0 = blockIdx.x * 8;
    const int rowa = wave < 4 ? j0 + 2 * wave : j0 - 1, rowb = wave < 4 ? rowa + 1 : j0 + 8;
    const bool livea = rowa >= 0 && rowa < a.rows, liveb = rowb < a.rows;
    const size_t vb = (size_t)sys * a.g.n;
    const size_t oa = vb + (size_t)(livea ? rowa : 0) * N, ob = vb + (size_t)(liveb ? rowb : 0) * N;
    float ua[EPL], ub[EPL];
    fgfft::load_row<N>(a.t1 + oa, lane, ua); fgfft::load_row<N>(a.t1 + ob, lane, ub);
#pragma unroll
    for (int k = 0; k < EPL; ++k) { if (!livea) ua[k] = 0.f; if (!liveb) ub[k] = 0.f; }
    __syncthreads();
    float2* spare;
    float* zr = fgfft::inverse_rows<N, true>(ua, ub, buf[0][wave], buf[1][wave], twl, nullptr, fgfft::Scales{a.is0, a.is}, lane, &spare);
    __syncthreads();
    constexpr int NP = (KIND == 0) ? 3 : 1;
    float part[3] = {0.f, 0.f, 0.f};
    if (wave < 4) {
        const float* zup = (wave == 0) ? zr + 4 * 2 * N : zr - 2 * N + N;
        const float* zdn = (wave == 3) ? zr + 1 * 2 * N + N : zr + 2 * N;
        const size_t mb = (size_t)b * a.g.n, NN = (size_t)a.g.n;
        const float* __restrict__ offb = q.off + (size_t)b * 4 * NN;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const bool live = half ? liveb : livea;
            if (!live) continue;      // (wave-uniform)
            const int j = half ? rowb : rowa;
            const size_t ro = (size_t)j * N;
            const float* zc_ = zr + half * N;
            const float* zu_ = half ? zr : zup;
            const float* zd_ = half ? zdn : zr + N;
            float xo[EPL], yo[EPL], sv[EPL], wv[EPL];
            fgfft::load_row<N>(q.rw + vb + ro, lane, wv);
            if (KIND == 0) fgfft::load_row<N>(q.r + vb + ro, lane, sv);
#pragma unroll
            for (int g = 0; g < M::NG; ++g) {
                const int i0 = g * 64 * VW + lane * VW;
                float d[VW], o0[VW], o1[VW], o2[VW], o3[VW], xc[VW], xu[VW], xd[VW];
                fgfft::ldv<VW>(q.diag + mb + ro + i0, d);
                fgfft::ldv<VW>(offb + 0 * NN + ro + i0, o0); fgfft::ldv<VW>(offb + 1 * NN + ro + i0, o1);
                fgfft::ldv<VW>(offb + 2 * NN + ro + i0, o2); fgfft::ldv<VW>(offb + 3 * NN + ro + i0, o3);
                fgfft::ldv<VW>(zc_ + i0, xc); fgfft::ldv<VW>(zu_ + i0, xu); fgfft::ldv<VW>(zd_ + i0, xd);
                const float xl = zc_[(i0 == 0) ? N - 1 : i0 - 1], xr = zc_[(i0 + VW == N) ? 0 : i0 + VW];      // periodic x
#pragma unroll
                for (int k = 0; k < VW; ++k) {
                    const float x_l = (k == 0) ? xl : xc[k > 0 ? k - 1 : 0], x_r = (k == VW - 1) ? xr : xc[k < VW - 1 ? k + 1 : VW - 1];
                    xo[g * VW + k] = xc[k];
                    yo[g * VW + k] = d[k] * xc[k] + o0[k] * x_l + o1[k] * x_r + o2[k] * xu[k] + o3[k] * xd[k];
                }
            }
#pragma unroll
            for (int k = 0; k < EPL; ++k) {
                if (KIND == 0) { part[0] += yo[k] * sv[k]; part[1] += yo[k] * yo[k]; part[2] += wv[k] * yo[k]; }
                else part[0] += wv[k] * yo[k];
            }
            fgfft::store_row<N>((KIND == 0 ? const_cast<float*>(q.ms) : const_cast<float*>(q.mp)) + vb + ro, lane, xo);
            fgfft::store_row<N>((KIND == 0 ? q.t : q.v) + vb + ro, lane, yo);
        }
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const float s1 = fg_wave_sum(part[k]);
        if (lane == 0) red[k * 5 + wave] = s1;
    }
    __syncthreads();
    if (threadIdx.x < NP) {
        const int k = threadIdx.x;
        const float tot = ((red[k * 5] + red[k * 5 + 1]) + (red[k * 5 + 2] + red[k * 5 + 3])) + red[k * 5 + 4];
        const int nm = (KIND == 0) ? (k == 0 ? F_TS : (k == 1 ? F_TT : F_RT)) + e : F_RV + e;
        acc_add(q.acc + ((size_t)sys * FG_ACC_DOUBLES + nm), (double)tot);
    }
}

template <int KIND>
int launch_fwd(const fg_state* s, int slot, int n, const FbicgArgs& a, dim3 grid, hipStream_t st) {
    switch (n) {
        case 64: FG_LAUNCH_P(s, slot, (k_fbicg_fwd<64, KIND>), grid, dim3(256), 0, st, a); break;
        case 128: FG_LAUNCH_P(s, slot, (k_fbicg_fwd<128, KIND>), grid, dim3(256), 0, st, a); break;
        case 256: FG_LAUNCH_P(s, slot, (k_fbicg_fwd<256, KIND>), grid, dim3(256), 0, st, a); break;
        case 512: FG_LAUNCH_P(s, slot, (k_fbicg_fwd<512, KIND>), grid, dim3(256), 0, st, a); break;
        default: fg_set_error("fused BiCGStab: unsupported row length"); return FG_ERR_UNSUPPORTED;
    }
    return FG_OK;
}
template <int KIND>
int launch_inv(const fg_state* s, int slot, int n, const FbicgArgs& a, dim3 grid, hipStream_t st) {
    switch (n) {
        case 64: FG_LAUNCH_P(s, slot, (k_fbicg_inv<64, KIND>), grid, dim3(320), 0, st, a); break;
        case 128: FG_LAUNCH_P(s, slot, (k_fbicg_inv<128, KIND>), grid, dim3(320), 0, st, a); break;
        case 256: FG_LAUNCH_P(s, slot, (k_fbicg_inv<256, KIND>), grid, dim3(320), 0, st, a); break;
        case 512: FG_LAUNCH_P(s, slot, (k_fbicg_inv<512, KIND>), grid, dim3(320), 0, st, a); break;
        default: fg_set_error("fused BiCGStab: unsupported row length"); return FG_ERR_UNSUPPORTED;
    }
    return FG_OK;
}

FbicgArgs make_args(const fg_state* s, const BicgPtrs& q, int it, int fold) {
    FbicgArgs a = {};
    a.g = s->grid; a.q = q; a.t1 = s->w[7]; a.tw = s->fd_dct_tw;
    a.fs0 = s->fd_dct_fwd[0]; a.fs = s->fd_dct_fwd[1]; a.is0 = s->fd_dct_inv[0]; a.is = s->fd_dct_inv[1];
    a.it = it; a.fold = fold; a.rows = s->grid.ny;
    return a;
}

}  // namespace

bool fg_fbicg_ok(const fg_state* s) {
    const FgGrid& G = s->grid;
    return s->bicg_pfused && s->fd_dct_x == 2 && G.dims == 2 && fg_fd_dct_supported(G.nx) && G.fixed[2] && G.fixed[3] && G.ny >= 3;
}

// kind: 0 = FS(it), 1 = FP(it)
int fg_fbicg_forward(fg_state* s, const BicgPtrs& q, int kind, int it, int fold, hipStream_t st) {
    const FgGrid& G = s->grid;
    const FbicgArgs a = make_args(s, q, it, fold);
    const int nsys = G.B * q.nc;
    const dim3 grid((G.ny + 7) / 8, nsys);
    // per system: FS r, v, rw read, s, u written (20 B / cell); FP x, p^, s^, s, t, p, v read, x, r, p, u written (44)
    const int slot = fg_prof_slot(s, FG_PK_FBICG_FWD, q.flags, nsys, (kind == 0 ? 20.0 : 44.0) * G.n, (8.0 + 5.0 * log2((double)G.nx)) * G.n, st);
    if (int rc = (kind == 0 ? launch_fwd<0>(s, slot, G.nx, a, grid, st) : launch_fwd<1>(s, slot, G.nx, a, grid, st))) return rc;
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}
// kind: 0 = IT(it), 1 = IV(it)
int fg_fbicg_inverse(fg_state* s, const BicgPtrs& q, int kind, int it, hipStream_t st) {
    const FgGrid& G = s->grid;
    const FbicgArgs a = make_args(s, q, it, 0);
    const int nsys = G.B * q.nc;
    const dim3 grid((G.ny + 7) / 8, nsys);
    // per system: u read, p^ / s^ and v / t written, rw (+ s) read, the five matrix fields of the env (20 B per cell and WORKGROUP:
    // a workgroup serves one system here)
    const int slot = fg_prof_slot(s, FG_PK_FBICG_INV, q.flags, nsys, (kind == 0 ? 40.0 : 36.0) * G.n, (12.0 + 5.0 * log2((double)G.nx)) * G.n, st);
    if (int rc = (kind == 0 ? launch_inv<0>(s, slot, G.nx, a, grid, st) : launch_inv<1>(s, slot, G.nx, a, grid, st))) return rc;
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}
#endif
