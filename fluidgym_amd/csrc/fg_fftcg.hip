// Fused row kernels of the fast-diagonalisation preconditioned pressure CG on 2-D grids whose x axis carries a fast transform
// (uniform FIXED x: cosine basis, the channel stand-ins of BASELINE configs 2 / 5; uniform periodic x: real Fourier basis, RBC =
// config 3).  Replaces cgSolveGPU (cg_solver_kernel.cu:129-471) on the pressure system of PISO_build_pressure_matrix
// (PISO_multiblock_cuda_kernel.cu:4812-4978), at the tolerance / RMS criterion of the reference (cg_solver_kernel.cu:100-106).
//
// Rounds 1-4 ran an iteration of the preconditioned CG as FIVE launches (k_cg_ap, k_cg_update, k_dct_rows forward, k_tridiag_y_lds,
// k_dct_rows inverse), every one of them 8-12 us at 2 M cells, i.e. launch-latency-sized: 20 of the 49 launches of a PISO step of the
// headline workload and 0.2-0.35 of the HBM figure each.  Here an iteration is THREE launches.  The two row transforms are
// row-local, so everything else that is row-local rides in them, and the one stencil of the iteration is applied to the rows a
// workgroup has just transformed back (plus one halo row above and below, transformed again by a fifth wave):
//
//   I'(k):  z_k = Qx (T^-1 u)            inverse transform of the rows the tridiagonal kernel left        (k_fcg_inv_apply)
//           w_k = P z_k                  matrix-free pressure operator on the rows in LDS
//           gamma_k = r_k . z_k,  delta_k = z_k . w_k
//   F'(k):  beta_k = gamma_k / gamma_{k-1},  alpha_k = gamma_k / (delta_k - beta_k gamma_k / alpha_{k-1})  (k_fcg_update_fwd)
//           p_k = z_k + beta_k p_{k-1},  s_k = w_k + beta_k s_{k-1}          (s = P p by linearity: no second stencil)
//           x_{k+1} = x_k + alpha_k p_k,  r_{k+1} = r_k - alpha_k s_k,  rr_{k+1} = r.r,  sum(x_{k+1})
//           u = Qx^T r_{k+1}             forward transform of the rows just updated
//   L(k+1): verdict on rr_{k+1}, per-mode Thomas solve along y                                             (k_tridiag_y_lds, fg_fdprecond.hip)
//
// This is the Chronopoulos-Gear arrangement of the preconditioned CG (one reduction phase per iteration: p.Pp is obtained from
// z.Pz by the recurrence above instead of a dot product of its own); in exact arithmetic the iterates are those of the classic
// recurrence, and like it the loop is restarted from the true residual every `reset_steps` iterations (cg_solver_kernel.cu:281-302).
// Same 72 B per cell and iteration as the five kernels, three launches.  sum(x) rides along so that the mean removal of the
// pressure (PISOtorch_simulation.py:1922-1925) needs no pass of its own: the corrector subtracts it where it copies the pressure
// to the block (k_correct, fg_piso.hip).
#include "fg_internal.h"
#include "fg_cg.h"
#include "fg_fftrow.h"
#include "fg_fftcg.h"

#if !FG_F64
namespace {

using fgfft::Map;

// ---------------------------------------------------------------------------------------------------------------------------
// F'(it): the vector updates of iteration `it` and the forward transform of the new residual.  One wave per PAIR of rows, four
// waves per workgroup; grid (ceil(rows / 8), B).
// ---------------------------------------------------------------------------------------------------------------------------
template <int N, bool PERIODIC>
__global__ __launch_bounds__(256) void k_fcg_update_fwd(FcgUpdArgs a) {
    constexpr int EPL = N / 64;
    __shared__ __attribute__((aligned(16))) float2 buf[2][4][N];
    __shared__ __attribute__((aligned(16))) float2 twl[N];
    __shared__ float red[8];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (a.flags[b] != 0) {
        if (a.lazy && a.lazy[b] == 1) {
            // the first iterate of this env met the tolerance (k_fcg_check0): x_1 = x_0 + alpha z is all that is left to write
            const float alpha = (float)a.alpha[b * 2];
            const int row0 = 2 * (blockIdx.x * 4 + wave);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int row = row0 + half;
                if (row >= a.rows) continue;
                const size_t o = (size_t)b * a.env_stride + (size_t)row * N;
                float zv[EPL], xv[EPL];
                fgfft::load_row<N>(a.z + o, lane, zv);
                if (a.x_zero) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) xv[e] = 0.f;
                } else {
                    fgfft::load_row<N>(a.x + o, lane, xv);
                }
#pragma unroll
                for (int e = 0; e < EPL; ++e) xv[e] += alpha * zv[e];
                fgfft::store_row<N>(a.x + o, lane, xv);
            }
        } else if (a.x_zero && (a.flags[b] == 1 || a.flags[b] == 2)) {
            // the start vector of this env met the tolerance (or its right-hand side is not finite): x_0 = 0 is its result
            float z0[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) z0[e] = 0.f;
            const int r0 = 2 * (blockIdx.x * 4 + wave);
            if (r0 < a.rows) fgfft::store_row<N>(a.x + (size_t)b * a.env_stride + (size_t)r0 * N, lane, z0);
            if (r0 + 1 < a.rows) fgfft::store_row<N>(a.x + (size_t)b * a.env_stride + (size_t)(r0 + 1) * N, lane, z0);
        }
        return;
    }
    for (int k = threadIdx.x; k < N; k += 256) twl[k] = a.tw[k];
    // scalars of the iteration, derived by every workgroup from the same accumulator words
    const int it = a.it;
    const double g_new = fg_acc_total(fg_acc_ptr(a.acc, b, FCG_GAMMA + it % 3), a.ns);
    const double d_new = fg_acc_total(fg_acc_ptr(a.acc, b, FCG_DELTA + (it & 1)), a.ns);
    double beta = 0.0, denom = d_new;
    if (!a.first) {
        const double g_old = fg_acc_total(fg_acc_ptr(a.acc, b, FCG_GAMMA + (it + 2) % 3), a.ns);
        const double al_old = a.alpha[b * 2 + ((it + 1) & 1)];
        beta = g_new / g_old;
        denom = d_new - beta * g_new / al_old;
    }
    const double alpha_d = g_new / denom;
    const float alpha = (float)alpha_d, betaf = (float)beta;
    const bool save = a.best.save_at[b] == it;      // the leader of L(it) found x_it worth keeping (FgBest)
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        fg_acc_zero(fg_acc_ptr(a.acc, b, FCG_GAMMA + (it + 1) % 3), a.ns);     // filled by I'(it + 1); last read by F'(it - 1)
        fg_acc_zero(fg_acc_ptr(a.acc, b, FCG_DELTA + ((it + 1) & 1)), a.ns);
        if (threadIdx.x == 0) {
            a.alpha[b * 2 + (it & 1)] = alpha_d;
            acc_st(a.xsum + (b * 2 + ((it + 1) & 1)), 0.0);
        }
    }
    const int row0 = 2 * (blockIdx.x * 4 + wave), row1 = row0 + 1;
    const bool live0 = row0 < a.rows, live1 = row1 < a.rows;
    const size_t o0 = (size_t)b * a.env_stride + (size_t)(live0 ? row0 : 0) * N;
    const size_t o1 = (size_t)b * a.env_stride + (size_t)(live1 ? row1 : 0) * N;
    float pa[EPL], pb[EPL], sa[EPL], sb[EPL], xa[EPL], xb[EPL], ra[EPL], rb[EPL];
    {
        float za[EPL], zb[EPL], wa[EPL], wb[EPL];
        fgfft::load_row<N>(a.z + o0, lane, za); fgfft::load_row<N>(a.z + o1, lane, zb);
        fgfft::load_row<N>(a.w + o0, lane, wa); fgfft::load_row<N>(a.w + o1, lane, wb);
        if (!a.first) {
            fgfft::load_row<N>(a.p + o0, lane, pa); fgfft::load_row<N>(a.p + o1, lane, pb);
            fgfft::load_row<N>(a.s + o0, lane, sa); fgfft::load_row<N>(a.s + o1, lane, sb);
        }
        if (a.x_zero) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) { xa[e] = 0.f; xb[e] = 0.f; }
        } else {
            fgfft::load_row<N>(a.x + o0, lane, xa); fgfft::load_row<N>(a.x + o1, lane, xb);
        }
        const float* rsrc = a.r0 ? a.r0 : a.r;
        fgfft::load_row<N>(rsrc + o0, lane, ra); fgfft::load_row<N>(rsrc + o1, lane, rb);
        if (save) {
            if (live0) fgfft::store_row<N>(a.best.best_x + o0, lane, xa);
            if (live1) fgfft::store_row<N>(a.best.best_x + o1, lane, xb);
        }
        if (a.first) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) { pa[e] = za[e]; pb[e] = zb[e]; sa[e] = wa[e]; sb[e] = wb[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                pa[e] = za[e] + betaf * pa[e]; pb[e] = zb[e] + betaf * pb[e];
                sa[e] = wa[e] + betaf * sa[e]; sb[e] = wb[e] + betaf * sb[e];
            }
        }
    }
    float part[2] = {0.f, 0.f};     // r.r | sum x
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        xa[e] += alpha * pa[e]; ra[e] -= alpha * sa[e];
        xb[e] += alpha * pb[e]; rb[e] -= alpha * sb[e];
        if (!live0) { xa[e] = 0.f; ra[e] = 0.f; }
        if (!live1) { xb[e] = 0.f; rb[e] = 0.f; }
        part[0] += ra[e] * ra[e] + rb[e] * rb[e];
        part[1] += xa[e] + xb[e];
    }
    if (live0) {
        if (!a.first) { fgfft::store_row<N>(a.p + o0, lane, pa); fgfft::store_row<N>(a.s + o0, lane, sa); }
        fgfft::store_row<N>(a.x + o0, lane, xa); fgfft::store_row<N>(a.r + o0, lane, ra);
    }
    if (live1) {
        if (!a.first) { fgfft::store_row<N>(a.p + o1, lane, pb); fgfft::store_row<N>(a.s + o1, lane, sb); }
        fgfft::store_row<N>(a.x + o1, lane, xb); fgfft::store_row<N>(a.r + o1, lane, rb);
    }
    __syncthreads();      // twiddle table staged
    float oa[EPL], ob[EPL];
    fgfft::forward_rows<N, PERIODIC>(ra, rb, oa, ob, buf[0][wave], buf[1][wave], twl, a.rot, fgfft::Scales{a.fs0, a.fs}, lane);
    if (live0) fgfft::store_row<N>(a.t1 + o0, lane, oa);
    if (live1) fgfft::store_row<N>(a.t1 + o1, lane, ob);
    const float tot = fg_block_sum_lanes<2>(part, red);
    if (threadIdx.x == 0) fg_acc_add(fg_acc_ptr(a.acc, b, (it + 1) % 3), a.ns, blockIdx.x, (double)tot);
    if (threadIdx.x == 1) acc_add(a.xsum + (b * 2 + (it & 1)), (double)tot);
}

// ---------------------------------------------------------------------------------------------------------------------------
// I'(it): inverse transform of eight rows (+ one halo row above and below), the pressure operator on them, the two dot products.
// Five waves per workgroup: waves 0-3 own rows j0 + 2w, j0 + 2w + 1; wave 4 the halo pair (j0 - 1, j0 + 8).  grid (ceil(rows / 8), B).
// The operator is fg_poisson_coef / fg_apply_elem of fg_poisson.hip term by term (PISO_build_pressure_matrix,
// PISO_multiblock_cuda_kernel.cu:4842-4889): off_f = (alpha_a(c) rA_c + alpha_a(N) rA_N) / 2, no entry at a prescribed face.
// ---------------------------------------------------------------------------------------------------------------------------
template <int N, bool PERIODIC>
__global__ __launch_bounds__(320) void k_fcg_inv_apply(FcgInvArgs a) {
    constexpr int EPL = N / 64;
    using M = Map<N>;
    constexpr int VW = M::VW;
    __shared__ __attribute__((aligned(16))) float2 buf[2][5][N];
    __shared__ __attribute__((aligned(16))) float2 twl[N];
    __shared__ float red[30];
    const int b = blockIdx.y;
    if (a.flags[b] != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < N; k += 320) twl[k] = a.tw[k];
    const int j0 = blockIdx.x * 8;
    const int rowa = wave < 4 ? j0 + 2 * wave : j0 - 1, rowb = wave < 4 ? rowa + 1 : j0 + 8;
    const bool livea = rowa >= 0 && rowa < a.rows, liveb = rowb < a.rows;
    const size_t oa = (size_t)b * a.env_stride + (size_t)(livea ? rowa : 0) * N;
    const size_t ob = (size_t)b * a.env_stride + (size_t)(liveb ? rowb : 0) * N;
    float ua[EPL], ub[EPL], ca[EPL], cb[EPL];
    fgfft::load_row<N>(a.u + oa, lane, ua); fgfft::load_row<N>(a.u + ob, lane, ub);
    fgfft::load_row<N>(a.rA + oa, lane, ca); fgfft::load_row<N>(a.rA + ob, lane, cb);
    float ra[EPL], rb[EPL];
    if (wave < 4) { fgfft::load_row<N>(a.r + oa, lane, ra); fgfft::load_row<N>(a.r + ob, lane, rb); }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        if (!livea) { ua[e] = 0.f; ca[e] = 0.f; }
        if (!liveb) { ub[e] = 0.f; cb[e] = 0.f; }
    }
    __syncthreads();      // twiddle table staged
    float2* spare;
    float* zr = fgfft::inverse_rows<N, PERIODIC>(ua, ub, buf[0][wave], buf[1][wave], twl, a.rot, fgfft::Scales{a.is0, a.is}, lane, &spare);
    float* cr = reinterpret_cast<float*>(spare);      // rA rows next to the z rows: [row a | row b], natural order
#pragma unroll
    for (int q = 0; q < M::NG; ++q) {
        fgfft::stv<VW>(cr + q * 64 * VW + lane * VW, &ca[q * VW]);
        fgfft::stv<VW>(cr + N + q * 64 * VW + lane * VW, &cb[q * VW]);
    }
    __syncthreads();
    float part[2] = {0.f, 0.f};     // r.z | z.Pz
    float ext[4] = {0.f, 0.f, 0.f, 0.f};      // d.w | w.w | sum z | d.d with d = r - w (I'(0) only: FcgInvArgs::extras)
    if (wave < 4) {
        // every wave's result landed in the same one of its two buffers: wave w' has its rows at zr + (w' - wave) * 2N floats
        const float* zup = (wave == 0) ? zr + 4 * 2 * N : zr - 2 * N + N;           // row above row a: halo wave's row a | wave - 1's row b
        const float* cup = (wave == 0) ? cr + 4 * 2 * N : cr - 2 * N + N;
        const float* zdn = (wave == 3) ? zr + 1 * 2 * N + N : zr + 2 * N;           // row below row b: halo wave's row b | wave + 1's row a
        const float* cdn = (wave == 3) ? cr + 1 * 2 * N + N : cr + 2 * N;
        float za[EPL], zb[EPL], wa[EPL], wb[EPL];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int j = half ? rowb : rowa;
            const bool live = half ? liveb : livea;
            const float* zc_ = zr + half * N;  const float* cc_ = cr + half * N;
            const float* zu_ = half ? zr : zup; const float* cu_ = half ? cr : cup;
            const float* zd_ = half ? zdn : zr + N; const float* cd_ = half ? cdn : cr + N;
            const int jc = live ? j : 0;
            const float hy = a.hy[jc], rhy = a.rhy[jc];
            const float rhy_m = a.rhy[jc == 0 ? a.rows - 1 : jc - 1], rhy_p = a.rhy[jc == a.rows - 1 ? 0 : jc + 1];
            const float mym = (jc == 0) ? 0.f : 1.f, myp = (jc == a.rows - 1) ? 0.f : 1.f;     // y faces are FIXED (the tridiagonal axis)
#pragma unroll
            for (int q = 0; q < M::NG; ++q) {
                const int i0 = q * 64 * VW + lane * VW;
                float zc[VW], cc[VW], zu[VW], cu[VW], zd[VW], cd[VW], hx[VW], rhx[VW];
                fgfft::ldv<VW>(zc_ + i0, zc); fgfft::ldv<VW>(cc_ + i0, cc);
                fgfft::ldv<VW>(zu_ + i0, zu); fgfft::ldv<VW>(cu_ + i0, cu);
                fgfft::ldv<VW>(zd_ + i0, zd); fgfft::ldv<VW>(cd_ + i0, cd);
                fgfft::ldv<VW>(a.hx + i0, hx); fgfft::ldv<VW>(a.rhx + i0, rhx);
                const int il = (i0 == 0) ? N - 1 : i0 - 1, ir = (i0 + VW == N) ? 0 : i0 + VW;
                const float zl = zc_[il], cl = cc_[il], zrr = zc_[ir], crr = cc_[ir];
                const float rhx_l = a.rhx[il], rhx_r = a.rhx[ir];
                const float mxm = (i0 == 0 && a.fixed_x) ? 0.f : 1.f, mxp = (i0 + VW == N && a.fixed_x) ? 0.f : 1.f;
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    const float z_l = (e == 0) ? zl : zc[e > 0 ? e - 1 : 0], z_r = (e == VW - 1) ? zrr : zc[e < VW - 1 ? e + 1 : VW - 1];
                    const float c_l = (e == 0) ? cl : cc[e > 0 ? e - 1 : 0], c_r = (e == VW - 1) ? crr : cc[e < VW - 1 ? e + 1 : VW - 1];
                    const float rh_lo = (e == 0) ? rhx_l : rhx[e > 0 ? e - 1 : 0], rh_hi = (e == VW - 1) ? rhx_r : rhx[e < VW - 1 ? e + 1 : VW - 1];
                    const float ml = (e == 0) ? mxm : 1.f, mh = (e == VW - 1) ? mxp : 1.f;
                    const float ayz = hy;                       // (hz = 1 in 2-D)
                    const float apx = ayz * rhx[e] * cc[e];
                    const float kxm = ml * 0.5f * (apx + ayz * rh_lo * c_l);
                    const float kxp = mh * 0.5f * (apx + ayz * rh_hi * c_r);
                    const float axz = hx[e];
                    const float apy = axz * rhy * cc[e];
                    const float kym = mym * 0.5f * (apy + axz * rhy_m * cu[e]);
                    const float kyp = myp * 0.5f * (apy + axz * rhy_p * cd[e]);
                    const float wv = kxm * (z_l - zc[e]) + kxp * (z_r - zc[e]) + kym * (zu[e] - zc[e]) + kyp * (zd[e] - zc[e]);
                    if (half) { zb[q * VW + e] = zc[e]; wb[q * VW + e] = wv; }
                    else { za[q * VW + e] = zc[e]; wa[q * VW + e] = wv; }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            if (livea) { part[0] += ra[e] * za[e]; part[1] += za[e] * wa[e]; }
            if (liveb) { part[0] += rb[e] * zb[e]; part[1] += zb[e] * wb[e]; }
        }
        if (a.extras) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const float da = ra[e] - wa[e], db = rb[e] - wb[e];
                if (livea) { ext[0] += da * wa[e]; ext[1] += wa[e] * wa[e]; ext[2] += za[e]; ext[3] += da * da; }
                if (liveb) { ext[0] += db * wb[e]; ext[1] += wb[e] * wb[e]; ext[2] += zb[e]; ext[3] += db * db; }
            }
        }
        if (livea) { fgfft::store_row<N>(a.z + oa, lane, za); fgfft::store_row<N>(a.w + oa, lane, wa); }
        if (liveb) { fgfft::store_row<N>(a.z + ob, lane, zb); fgfft::store_row<N>(a.w + ob, lane, wb); }
    }
    // workgroup sums over five waves
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float sv = fg_wave_sum(part[q]);
        if (lane == 0) red[q * 5 + wave] = sv;
    }
    if (a.extras) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float sv = fg_wave_sum(ext[q]);
            if (lane == 0) red[10 + q * 5 + wave] = sv;
        }
    }
    __syncthreads();
    if (a.extras && threadIdx.x >= 64 && threadIdx.x < 68) {
        const int q = threadIdx.x - 64;
        const float* rq = red + 10 + q * 5;
        const float tot = ((rq[0] + rq[1]) + (rq[2] + rq[3])) + rq[4];
        if (q == 2) acc_add(a.xsum + (b * 2 + 1), (double)tot);
        else fg_acc_add(fg_acc_ptr(a.acc, b, q == 0 ? FCG_GAMMA + 2 : (q == 1 ? FCG_DELTA + 1 : 2)), a.ns, blockIdx.x, (double)tot);
    }
    if (threadIdx.x < 2) {
        const int q = threadIdx.x;
        const float tot = ((red[q * 5] + red[q * 5 + 1]) + (red[q * 5 + 2] + red[q * 5 + 3])) + red[q * 5 + 4];
        fg_acc_add(fg_acc_ptr(a.acc, b, q == 0 ? FCG_GAMMA + a.it % 3 : FCG_DELTA + (a.it & 1)), a.ns, blockIdx.x, (double)tot);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// C(0): the verdict on the FIRST iterate, before any vector is updated.  The first step of the recurrence is x_1 = x_0 + alpha z_0,
// r_1 = r_0 - alpha w_0 with alpha = gamma_0 / delta_0.  With d = r_0 - w_0 (small: the preconditioner is close to the inverse, so
// w_0 = P M^-1 r_0 is close to r_0 and alpha to 1) r_1 = d + (1 - alpha) w_0 and |r_1|^2 = d.d + 2 (1 - alpha) d.w + (1 - alpha)^2 w.w
// is a sum of three SMALL terms that I'(0) accumulates -- r.r - 2 alpha r.w + alpha^2 w.w cancels to the rounding of its fp32
// partial sums, 3e-4 |r_0|, and reported 1.5e-4 for an env whose first residual is 3e-6 -- with alpha rounded to fp32 as the update
// applies it.  An env that
// meets the tolerance (the RMS criterion of cg_solver_kernel.cu:100-106 on that number) is marked: its result is x_0 + alpha z_0,
// which F'(0) writes without touching r -- or which nobody writes at all when EVERY env of the batch ends here (the PISO pressure
// systems of the channel family: 99 % of the solves) and the caller reads the pressure as alpha z (FgLazyRef).  One wave per env.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int CHECK_WAVES = 16;      // envs per workgroup of the verdict kernels: their result words leave as one 256-byte run
__global__ __launch_bounds__(64 * CHECK_WAVES) void k_fcg_check0(FgDacc* __restrict__ acc, int32_t* __restrict__ flags, fg_solve_info* __restrict__ info,
                                                                 fg_solve_info* __restrict__ mirror, double* __restrict__ alpha, int32_t* __restrict__ lazy,
                                                                 float tol, int n, int B, int ns, FgPollOut poll) {
    __shared__ uint32_t stage[CHECK_WAVES * 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = blockIdx.x * CHECK_WAVES, b = b0 + wave;      // one wave per env (fg_acc_total is a wave's shuffle tree)
    const bool valid = b < B;
    int lz = 0;
    if (valid && flag_ld(flags + (b)) == 0) {
        const double g = fg_acc_total(fg_acc_ptr(acc, b, FCG_GAMMA), ns), d = fg_acc_total(fg_acc_ptr(acc, b, FCG_DELTA), ns);
        const double dd = fg_acc_total(fg_acc_ptr(acc, b, 2), ns);
        const double dw = fg_acc_total(fg_acc_ptr(acc, b, FCG_GAMMA + 2), ns), ww = fg_acc_total(fg_acc_ptr(acc, b, FCG_DELTA + 1), ns);
        const double alpha_d = g / d;
        const double e1 = 1.0 - (double)(float)alpha_d;
        double rr1 = dd + 2.0 * e1 * dw + e1 * e1 * ww;
        if (rr1 < 0.0) rr1 = 0.0;      // (a NaN stays one)
        const float crit = (float)sqrt(rr1 / (double)n);
        if (lane == 0) {
            info[b].final_residual = crit;
            info[b].used_iterations = 0;
            if (!(crit >= tol)) {
                const bool finite = isfinite(crit);
                flag_st(flags + (b), finite ? 1 : 2);
                info[b].converged = finite ? 1 : 0;
                info[b].is_finite = finite ? 1 : 0;
                if (finite) { alpha[b * 2] = alpha_d; lz = 1; }
            }
        }
    }
    if (valid && lane == 0) lazy[b] = lz;
    if (!mirror) return;
    if (poll.gran) {      // (the verdicts travel in the words the host spins on: no release, no L2 write-back -- FgPollOut, fg_internal.h)
        uint32_t w[2] = {0u, 0u};
        const bool writer = valid && lane == 0;
        if (writer) { const fg_solve_info v = info[b]; w[0] = __float_as_uint(v.final_residual); w[1] = fg_info_word(v); }
        fg_poll_publish_records<2>(poll, b0, min(CHECK_WAVES, B - b0), wave, w, writer, stage);
    } else if (valid && lane == 0) { mirror[b] = info[b]; fg_poll_publish(poll, b); }
}

// ---------------------------------------------------------------------------------------------------------------------------
// D(0): the right-hand side of the pressure system AND the start of its solve in one row kernel -- b = div of the fluxes of h
// (k_div of fg_piso.hip term by term: k_computePressureRHSdivergenceFromFlux, PISO_multiblock_cuda_kernel.cu:5389-5434; prescribed
// faces take the boundary-velocity flux, computeFluxesNDLoop :1567-1645), r_0 = b, x_0 = 0, r.r, u = Qx^T r_0.  Replaces k_div +
// the stand-alone forward transform in front of the first tridiagonal solve.  The divergence only needs h (written by k_h in the
// launch before): the x neighbours of a row sit in the row, the y neighbours are two more row loads.  grid (ceil(rows / 8), B).
// ---------------------------------------------------------------------------------------------------------------------------
template <int N, bool PERIODIC>
__global__ __launch_bounds__(256) void k_fcg_div_fwd(FcgDivArgs a) {
    constexpr int EPL = N / 64;
    using M = Map<N>;
    constexpr int VW = M::VW;
    __shared__ __attribute__((aligned(16))) float2 buf[2][4][N];
    __shared__ __attribute__((aligned(16))) float2 twl[N];
    __shared__ float red[4];
    const int b = blockIdx.y;
    if (a.dt && !(a.dt[b] > 0.f)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < N; k += 256) twl[k] = a.tw[k];
    const int row0 = 2 * (blockIdx.x * 4 + wave);
    const size_t hb0 = ((size_t)b * 2 + 0) * a.n, hb1 = ((size_t)b * 2 + 1) * a.n;
    // x component of h for the wave's two rows, staged so that the row neighbours are reachable
    float* sx = reinterpret_cast<float*>(buf[1][wave]);      // [row a | row b], natural order
    float da[EPL], db[EPL];
    float part[1] = {0.f};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int j = row0 + half;
        const bool live = j < a.rows;
        const int jc = live ? j : 0;
        float h0[EPL];
        fgfft::load_row<N>(a.h + hb0 + (size_t)jc * N, lane, h0);
#pragma unroll
        for (int q = 0; q < M::NG; ++q) fgfft::stv<VW>(sx + half * N + q * 64 * VW + lane * VW, &h0[q * VW]);
    }
    fgfft::wave_sync();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int j = row0 + half;
        const bool live = j < a.rows;
        const int jc = live ? j : 0;
        const float hy = a.hy[jc];
        const bool at_lo = jc == 0, at_hi = jc == a.rows - 1;      // y faces are FIXED (the tridiagonal axis)
        const float* sr = sx + half * N;
        float* out = half ? db : da;
#pragma unroll
        for (int q = 0; q < M::NG; ++q) {
            const int i0 = q * 64 * VW + lane * VW;
            float hc[VW], v1[VW], vu[VW], vd[VW], hx[VW], bl[VW], bh[VW];
            fgfft::ldv<VW>(sr + i0, hc);
            fgfft::ldv<VW>(a.h + hb1 + (size_t)jc * N + i0, v1);
            fgfft::ldv<VW>(a.h + hb1 + (size_t)(at_lo ? jc : jc - 1) * N + i0, vu);
            fgfft::ldv<VW>(a.h + hb1 + (size_t)(at_hi ? jc : jc + 1) * N + i0, vd);
            fgfft::ldv<VW>(a.hx + i0, hx);
            if (at_lo) fgfft::ldv<VW>(a.bvel[2] + ((size_t)b * 2 + 1) * N + i0, bl);
            if (at_hi) fgfft::ldv<VW>(a.bvel[3] + ((size_t)b * 2 + 1) * N + i0, bh);
            const float hl = sr[(i0 == 0) ? N - 1 : i0 - 1], hr = sr[(i0 + VW == N) ? 0 : i0 + VW];
            const bool xlo = (i0 == 0) && a.fixed_x, xhi = (i0 + VW == N) && a.fixed_x;
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const float h_l = (e == 0) ? hl : hc[e > 0 ? e - 1 : 0], h_r = (e == VW - 1) ? hr : hc[e < VW - 1 ? e + 1 : VW - 1];
                float acc = 0.f;
                {   // x faces: area = hy (hz = 1)
                    const float area = hy;
                    const float F_hi = (e == VW - 1 && xhi) ? a.bvel[1][((size_t)b * 2 + 0) * a.rows + jc] * area : 0.5f * (hc[e] + h_r) * area;
                    const float F_lo = (e == 0 && xlo) ? a.bvel[0][((size_t)b * 2 + 0) * a.rows + jc] * area : 0.5f * (hc[e] + h_l) * area;
                    acc += F_hi - F_lo;
                }
                {   // y faces: area = hx
                    const float area = hx[e];
                    const float F_hi = at_hi ? bh[e] * area : 0.5f * (v1[e] + vd[e]) * area;
                    const float F_lo = at_lo ? bl[e] * area : 0.5f * (v1[e] + vu[e]) * area;
                    acc += F_hi - F_lo;
                }
                out[q * VW + e] = live ? acc : 0.f;
            }
        }
        if (live) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) part[0] += out[e] * out[e];
            const size_t o = (size_t)b * a.n + (size_t)j * N;
            fgfft::store_row<N>(a.div + o, lane, half ? db : da);
            // (r_0 = b and x_0 = 0 are not stored: the first update kernel of the solve reads b and knows the zero, FcgUpdArgs::r0)
        }
    }
    __syncthreads();      // twiddle table staged (and every lane is through with the staged rows)
    float oa[EPL], ob[EPL];
    fgfft::forward_rows<N, PERIODIC>(da, db, oa, ob, buf[0][wave], buf[1][wave], twl, a.rot, fgfft::Scales{a.fs0, a.fs}, lane);
    if (row0 < a.rows) fgfft::store_row<N>(a.t1 + (size_t)b * a.n + (size_t)row0 * N, lane, oa);
    if (row0 + 1 < a.rows) fgfft::store_row<N>(a.t1 + (size_t)b * a.n + (size_t)(row0 + 1) * N, lane, ob);
    fg_block_sum<1>(part, red);
    if (threadIdx.x == 0) fg_acc_add(fg_acc_ptr(a.acc, b, 0), a.ns, blockIdx.x, (double)part[0]);
}

template <bool PERIODIC>
int launch_upd(const fg_state* s, int slot, int n, const FcgUpdArgs& a, dim3 grid, hipStream_t st) {
    switch (n) {
        case 64: FG_LAUNCH_P(s, slot, (k_fcg_update_fwd<64, PERIODIC>), grid, dim3(256), 0, st, a); break;
        case 128: FG_LAUNCH_P(s, slot, (k_fcg_update_fwd<128, PERIODIC>), grid, dim3(256), 0, st, a); break;
        case 256: FG_LAUNCH_P(s, slot, (k_fcg_update_fwd<256, PERIODIC>), grid, dim3(256), 0, st, a); break;
        case 512: FG_LAUNCH_P(s, slot, (k_fcg_update_fwd<512, PERIODIC>), grid, dim3(256), 0, st, a); break;
        default: fg_set_error("fused CG: unsupported row length"); return FG_ERR_UNSUPPORTED;
    }
    return FG_OK;
}
template <bool PERIODIC>
int launch_inv(const fg_state* s, int slot, int n, const FcgInvArgs& a, dim3 grid, hipStream_t st) {
    switch (n) {
        case 64: FG_LAUNCH_P(s, slot, (k_fcg_inv_apply<64, PERIODIC>), grid, dim3(320), 0, st, a); break;
        case 128: FG_LAUNCH_P(s, slot, (k_fcg_inv_apply<128, PERIODIC>), grid, dim3(320), 0, st, a); break;
        case 256: FG_LAUNCH_P(s, slot, (k_fcg_inv_apply<256, PERIODIC>), grid, dim3(320), 0, st, a); break;
        case 512: FG_LAUNCH_P(s, slot, (k_fcg_inv_apply<512, PERIODIC>), grid, dim3(320), 0, st, a); break;
        default: fg_set_error("fused CG: unsupported row length"); return FG_ERR_UNSUPPORTED;
    }
    return FG_OK;
}

template <bool PERIODIC>
int launch_div(const fg_state* s, int n, const FcgDivArgs& a, dim3 grid, hipStream_t st) {
    // live profile (kind k_div): h (2) read, div, r, u written + the halo rows: 28 B per cell as the stand-alone divergence kernel it replaces
    const int slot = fg_prof_slot(s, FG_PK_DIV, nullptr, s->grid.B, 28.0 * (double)s->grid.n, (8.0 + 5.0 * log2((double)n)) * (double)s->grid.n, st);
    switch (n) {
        case 64: FG_LAUNCH_P(s, slot, (k_fcg_div_fwd<64, PERIODIC>), grid, dim3(256), 0, st, a); break;
        case 128: FG_LAUNCH_P(s, slot, (k_fcg_div_fwd<128, PERIODIC>), grid, dim3(256), 0, st, a); break;
        case 256: FG_LAUNCH_P(s, slot, (k_fcg_div_fwd<256, PERIODIC>), grid, dim3(256), 0, st, a); break;
        case 512: FG_LAUNCH_P(s, slot, (k_fcg_div_fwd<512, PERIODIC>), grid, dim3(256), 0, st, a); break;
        default: fg_set_error("fused CG: unsupported row length"); return FG_ERR_UNSUPPORTED;
    }
    return FG_OK;
}

}  // namespace

bool fg_fcg_ok(const fg_state* s) {
    const FgGrid& G = s->grid;
    return s->cg_fused && s->fd_dct_x != 0 && G.dims == 2 && fg_fd_dct_supported(G.nx) && G.fixed[2] && G.fixed[3] && G.ny >= 3 &&
           s->fcg_alpha != nullptr;
}

int fg_fcg_update_fwd(fg_state* s, const FcgVectors& v, int it, int first, int ns, hipStream_t st, const fg_real* r0) {
    const FgGrid& G = s->grid;
    FcgUpdArgs a = {};
    a.r0 = r0; a.x_zero = r0 ? 1 : 0;
    a.lazy = (it == 0 && s->fcg_check0_ran) ? s->fcg_lazy : nullptr;
    a.z = v.z; a.w = v.w; a.p = v.p; a.s = v.s; a.x = v.x; a.r = v.r; a.t1 = v.t1;
    a.tw = s->fd_dct_tw; a.rot = s->fd_dct_rot; a.fs0 = s->fd_dct_fwd[0]; a.fs = s->fd_dct_fwd[1];
    a.flags = s->flags; a.acc = s->cg_acc; a.alpha = s->fcg_alpha; a.xsum = s->fcg_xsum; a.best = s->cg_best;
    a.ns = ns; a.rows = G.ny; a.it = it; a.first = first; a.env_stride = G.n;
    const dim3 grid((G.ny + 7) / 8, G.B);
    // per env: z, w, x, r (+ p, s) read; x, r, t1 (+ p, s) written
    const int slot = fg_prof_slot(s, FG_PK_FCG_UPD, s->flags, G.B, (first ? (r0 ? 24.0 : 28.0) : 44.0) * G.n, (10.0 + 5.0 * log2((double)G.nx)) * G.n, st);
    if (int rc = (s->fd_dct_x == 2 ? launch_upd<true>(s, slot, G.nx, a, grid, st) : launch_upd<false>(s, slot, G.nx, a, grid, st))) return rc;
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_fcg_inv_apply(fg_state* s, const FcgVectors& v, const fg_real* rA, int it, int ns, hipStream_t st, int extras) {
    const FgGrid& G = s->grid;
    FcgInvArgs a = {};
    a.u = v.t1; a.r = v.r; a.rA = rA; a.z = v.z; a.w = v.w;
    a.tw = s->fd_dct_tw; a.rot = s->fd_dct_rot; a.is0 = s->fd_dct_inv[0]; a.is = s->fd_dct_inv[1];
    a.flags = s->flags; a.acc = s->cg_acc; a.ns = ns; a.rows = G.ny; a.it = it; a.env_stride = G.n;
    a.xsum = s->fcg_xsum; a.extras = (extras && it == 0) ? 1 : 0;
    a.hy = G.h[1]; a.rhy = G.rh[1]; a.hx = G.h[0]; a.rhx = G.rh[0]; a.fixed_x = G.fixed[0];
    const dim3 grid((G.ny + 7) / 8, G.B);
    // per env: u, rA, r read; z, w written (the halo rows come from L2)
    const int slot = fg_prof_slot(s, FG_PK_FCG_INV, s->flags, G.B, 20.0 * G.n, (14.0 + 5.0 * log2((double)G.nx)) * G.n, st);
    if (int rc = (s->fd_dct_x == 2 ? launch_inv<true>(s, slot, G.nx, a, grid, st) : launch_inv<false>(s, slot, G.nx, a, grid, st))) return rc;
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}
int fg_fcg_check0(fg_state* s, fg_real tol, int ns, hipStream_t st, FgPollOut poll) {
    const FgGrid& G = s->grid;
    hipLaunchKernelGGL(k_fcg_check0, dim3((G.B + CHECK_WAVES - 1) / CHECK_WAVES), dim3(64 * CHECK_WAVES), 0, st, s->cg_acc, s->flags, s->info_dev, s->info_pinned, s->fcg_alpha, s->fcg_lazy,
                       tol, G.n, G.B, ns, poll);
    FG_HIP_CHECK(hipGetLastError());
    s->fcg_check0_ran = 1;
    return FG_OK;
}
// right-hand side + start of the solve + first forward transform (k_fcg_div_fwd): div, r = w[0], x = p_result, u = w[3], r.r in ring entry 0
int fg_fcg_div_fwd(fg_state* s, const FgBounds& bnd, const fg_real* dt, const fg_real* hvec, fg_real* div, int ns, hipStream_t st) {
    const FgGrid& G = s->grid;
    FcgDivArgs a = {};
    a.h = hvec; a.div = div; a.r = s->w[0]; a.x = s->p_result; a.t1 = s->w[3];
    for (int f = 0; f < 4; ++f) a.bvel[f] = bnd.vel[f];
    a.tw = s->fd_dct_tw; a.rot = s->fd_dct_rot; a.fs0 = s->fd_dct_fwd[0]; a.fs = s->fd_dct_fwd[1];
    a.dt = dt; a.acc = s->cg_acc; a.ns = ns; a.rows = G.ny; a.n = G.n;
    a.hx = G.h[0]; a.hy = G.h[1]; a.fixed_x = G.fixed[0];
    const dim3 grid((G.ny + 7) / 8, G.B);
    if (int rc = (s->fd_dct_x == 2 ? launch_div<true>(s, G.nx, a, grid, st) : launch_div<false>(s, G.nx, a, grid, st))) return rc;
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}
#endif
