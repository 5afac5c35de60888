// Pressure-Poisson kernels for gfx950: matrix-free operator, Jacobi, red-black Gauss-Seidel and a
// device-resident batched conjugate-gradient solver (2 kernels / iteration, no host scalar reads).
//
// Operator (PISO_build_pressure_matrix, PISO_multiblock_cuda_kernel.cu:4842-4889):
//     (P x)_c = sum_f off_f (x_{N_f} - x_c),   off_f = (alpha_a(c) rA_c + alpha_a(N) rA_N) / 2,
// no entry at a prescribed (FIXED) face; alpha_a = J / h_a^2.  P is negative semi-definite; like the
// reference's CG (cg_solver_kernel.cu:250-442) we iterate on P itself (alpha, pAp < 0).
//
// Algorithmic HBM bytes per cell per launch (fp32): apply 12 (x, rA, y); Jacobi 20 (x, b, rA, xnew
// + x again is the same stream) -> 16; CG kernel 1 (r, p, rA -> p, Ap) 20; CG kernel 2 (p, Ap, x, r
// -> x, r) 24.  See DESIGN.md "roofline".
#include <math.h>
#include <cmath>
#include <vector>

#include "fg_internal.h"
#include "fg_cg.h"
#include "fg_rung64.h"
#include "fg_fftcg.h"

namespace {

// face coefficients of the thread's cells from rA (+ neighbours) and rectilinear metrics
template <int DIMS, int VEC>
struct FgCoef {
    fg_real xm[VEC], xp[VEC], ym[VEC], yp[VEC], zm[VEC], zp[VEC];
};

template <int DIMS, int VEC>
__device__ __forceinline__ FgCoef<DIMS, VEC> fg_poisson_coef(const FgCtx<DIMS, VEC>& c, const FgMetric<DIMS, VEC>& m,
                                                             const FgNbr<DIMS, VEC>& rA) {
    FgCoef<DIMS, VEC> k;
    const fg_real ayz = m.hy * m.hz;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const fg_real rh_lo = (e == 0) ? m.rhx_m : m.rhx[e > 0 ? e - 1 : 0];
        const fg_real rh_hi = (e == VEC - 1) ? m.rhx_p : m.rhx[e < VEC - 1 ? e + 1 : VEC - 1];
        const fg_real ml = (e == 0) ? c.mxm : 1.f, mh = (e == VEC - 1) ? c.mxp : 1.f;
        const fg_real apx = ayz * m.rhx[e] * rA.c.v[e];
        k.xm[e] = ml * 0.5f * (apx + ayz * rh_lo * rA.xm.v[e]);
        k.xp[e] = mh * 0.5f * (apx + ayz * rh_hi * rA.xp.v[e]);
        const fg_real axz = m.hx[e] * m.hz;
        const fg_real apy = axz * m.rhy * rA.c.v[e];
        k.ym[e] = c.mym * 0.5f * (apy + axz * m.rhy_m * rA.ym.v[e]);
        k.yp[e] = c.myp * 0.5f * (apy + axz * m.rhy_p * rA.yp.v[e]);
        if constexpr (DIMS == 3) {
            const fg_real axy = m.hx[e] * m.hy;
            const fg_real apz = axy * m.rhz * rA.c.v[e];
            k.zm[e] = c.mzm * 0.5f * (apz + axy * m.rhz_m * rA.zm.v[e]);
            k.zp[e] = c.mzp * 0.5f * (apz + axy * m.rhz_p * rA.zp.v[e]);
        } else {
            k.zm[e] = k.zp[e] = 0.f;
        }
    }
    return k;
}

template <int DIMS, int VEC>
__device__ __forceinline__ fg_real fg_apply_elem(const FgCoef<DIMS, VEC>& k, const FgNbr<DIMS, VEC>& x, int e) {
    fg_real y = k.xm[e] * (x.xm.v[e] - x.c.v[e]) + k.xp[e] * (x.xp.v[e] - x.c.v[e]) +
              k.ym[e] * (x.ym.v[e] - x.c.v[e]) + k.yp[e] * (x.yp.v[e] - x.c.v[e]);
    if constexpr (DIMS == 3) y += k.zm[e] * (x.zm.v[e] - x.c.v[e]) + k.zp[e] * (x.zp.v[e] - x.c.v[e]);
    return y;
}

// y = P x
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_poisson_apply(FgGrid g, const fg_real* __restrict__ rA_,
                                                             const fg_real* __restrict__ x_, fg_real* __restrict__ y_,
                                                             int tiles_x, int tiles_y, int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    if (!c.valid) return;
    const size_t base = (size_t)c.b * g.n;
    const FgMetric<DIMS, VEC> m = fg_metrics<DIMS, VEC>(g, c);
    const FgNbr<DIMS, VEC> rA = fg_gather<DIMS, VEC>(rA_ + base, c);
    const FgNbr<DIMS, VEC> x = fg_gather<DIMS, VEC>(x_ + base, c);
    const FgCoef<DIMS, VEC> k = fg_poisson_coef<DIMS, VEC>(c, m, rA);
    FgVec<VEC> y;
#pragma unroll
    for (int e = 0; e < VEC; ++e) y.v[e] = fg_apply_elem<DIMS, VEC>(k, x, e);
    fg_store<VEC>(y_ + base + c.idx, y);
}

// damped Jacobi sweep: xnew = x + omega (b - P x) / diag,  diag = -sum_f off_f
// RBGS (COLOR >= 0): same update, in place, only for cells with (i+j+k)&1 == COLOR
template <int DIMS, int VEC, bool RB>
__global__ __launch_bounds__(FG_BLOCK) void k_poisson_relax(FgGrid g, const fg_real* __restrict__ rA_,
                                                             const fg_real* __restrict__ b_, const fg_real* x_,
                                                             fg_real* xnew_, fg_real omega, int color,
                                                             int tiles_x, int tiles_y, int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    if (!c.valid) return;
    const size_t base = (size_t)c.b * g.n;
    const FgMetric<DIMS, VEC> m = fg_metrics<DIMS, VEC>(g, c);
    const FgNbr<DIMS, VEC> rA = fg_gather<DIMS, VEC>(rA_ + base, c);
    const FgNbr<DIMS, VEC> x = fg_gather<DIMS, VEC>(x_ + base, c);
    const FgVec<VEC> b = fg_load<VEC>(b_ + base + c.idx);
    const FgCoef<DIMS, VEC> k = fg_poisson_coef<DIMS, VEC>(c, m, rA);
    FgVec<VEC> out;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const fg_real diag = -(k.xm[e] + k.xp[e] + k.ym[e] + k.yp[e] + k.zm[e] + k.zp[e]);
        const fg_real res = b.v[e] - fg_apply_elem<DIMS, VEC>(k, x, e);
        fg_real v = x.c.v[e] + omega * res / diag;
        if constexpr (RB) {
            if (((c.i0 + e + c.j + c.k) & 1) != color) v = x.c.v[e];
        }
        out.v[e] = v;
    }
    fg_store<VEC>(xnew_ + base + c.idx, out);
}

// ---------------------------------------------------------------------------------------------
// Batched CG.  Per-env reduction accumulators (fp64).  To keep same-address atomic traffic off the
// critical path on big grids (256^3 = 16384 workgroups per kernel) every accumulator is spread over
// `ns` slots (power of two <= FG_CG_SLOTS); workgroup w adds into slot w & (ns-1) and every wave of a
// consuming kernel re-sums the ns slots itself with one coalesced load + wave64 shuffle reduction:
//   acc[b][0..2][slot]  rr ring   (rr_i in name i % 3)
//   acc[b][3..4][slot]  pAp ring  (pAp_i in name 3 + i % 2)
// Every workgroup derives alpha / beta / the convergence decision itself from the accumulators, so
// no scalar ever travels to the host inside the loop (the reference reads 3-4 scalars per
// iteration through cublasTdot/nrm2, cg_solver_kernel.cu:277,317,332,431).
// flags[b]: 0 running, 1 converged, 2 non-finite residual, 3 inactive env.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ fg_real fg_rms(double rr, int n) { return (fg_real)sqrt(rr / (double)n); }
// (FG_CG_SLOTS / FG_CG_NAMES, fg_acc_ptr / fg_acc_total / fg_acc_zero / fg_acc_add: fg_cg.h)

// r = b - P x (or r = b when x0 == 0), accumulates rr into ring name `name`
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_cg_residual(FgGrid g, const fg_real* __restrict__ rA_,
                                                           const fg_real* __restrict__ b_, fg_real* __restrict__ x_,
                                                           fg_real* __restrict__ r_, FgDacc* __restrict__ acc,
                                                           const int32_t* __restrict__ flags, int use_x0, int name,
                                                           int ns, int tiles_x, int tiles_y, int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    if (flag_ld(flags + (c.b)) != 0) return;
    __shared__ fg_real lds[4];
    const size_t base = (size_t)c.b * g.n;
    fg_real part[1] = {0.f};
    if (c.valid) {
        FgVec<VEC> r = fg_load<VEC>(b_ + base + c.idx);
        if (use_x0) {
            const FgMetric<DIMS, VEC> m = fg_metrics<DIMS, VEC>(g, c);
            const FgNbr<DIMS, VEC> rA = fg_gather<DIMS, VEC>(rA_ + base, c);
            const FgNbr<DIMS, VEC> x = fg_gather<DIMS, VEC>(x_ + base, c);
            const FgCoef<DIMS, VEC> k = fg_poisson_coef<DIMS, VEC>(c, m, rA);
#pragma unroll
            for (int e = 0; e < VEC; ++e) r.v[e] -= fg_apply_elem<DIMS, VEC>(k, x, e);
        } else {
            FgVec<VEC> z;
#pragma unroll
            for (int e = 0; e < VEC; ++e) z.v[e] = 0.f;
            fg_store<VEC>(x_ + base + c.idx, z);
        }
        fg_store<VEC>(r_ + base + c.idx, r);
#pragma unroll
        for (int e = 0; e < VEC; ++e) part[0] += r.v[e] * r.v[e];
    }
    fg_block_sum<1>(part, lds);
    if (threadIdx.x == 0) {
        const unsigned tile = fg_xcd_remap(blockIdx.x, gridDim.x) % tiles;
        fg_acc_add(fg_acc_ptr(acc, c.b, name), ns, tile, (double)part[0]);
    }
}

// CG kernel 1 of iteration `it`:  p = r + beta p ; Ap = P p ; pAp += p.Ap
// The p update is fused into the operator by recomputing it on the halo (r and p are gathered
// with neighbours), which removes one full pass over p per iteration.  Because the halo cells
// belong to other workgroups that rewrite p in this same launch, p is double-buffered:
// pin_ = p of iteration it-1 (read with halo), pout_ = p of iteration it (written, centre only).
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_cg_ap(FgGrid g, const fg_real* __restrict__ rA_,
                                                     const fg_real* __restrict__ z_, const fg_real* __restrict__ pin_,
                                                     fg_real* __restrict__ pout_, fg_real* __restrict__ Ap_,
                                                     FgDacc* __restrict__ acc, int32_t* __restrict__ flags,
                                                     fg_solve_info* __restrict__ info, int32_t* __restrict__ prof_active,
                                                     FgBest best, fg_real tol, int it, int first, int ns, int num_base,
                                                     int tiles_x, int tiles_y, int tiles) {
    // z_ = preconditioned residual (= r when num_base == 0); beta = num_it / num_{it-1} with the numerator
    // ring num_base (0: r.r, 5: r.z).  Convergence is always judged on the r.r ring (RMS residual).
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    if (flag_ld(flags + (c.b)) != 0) return;
    const double rr_new = fg_acc_total(fg_acc_ptr(acc, c.b, it % 3), ns);
    const fg_real crit = fg_rms(rr_new, g.n);
    const unsigned tile = fg_xcd_remap(blockIdx.x, gridDim.x) % tiles;
    const bool lead_block = (tile == 0);
    if (!(crit >= tol)) {  // converged (crit < tol) or NaN
        if (lead_block && threadIdx.x == 0) {
            const bool finite = isfinite(crit);
            flag_st(flags + (c.b), finite ? 1 : 2);
            info[c.b].final_residual = crit;
            info[c.b].used_iterations = it - 1;
            info[c.b].converged = finite ? 1 : 0;
            info[c.b].is_finite = finite ? 1 : 0;
        }
        return;
    }
    const double num_new = (num_base == 0) ? rr_new : fg_acc_total(fg_acc_ptr(acc, c.b, num_base + it % 3), ns);
    const double num_old = first ? 1.0 : fg_acc_total(fg_acc_ptr(acc, c.b, num_base + (it + 2) % 3), ns);
    if (lead_block && threadIdx.x < 64) {
        fg_acc_zero(fg_acc_ptr(acc, c.b, (it + 1) % 3), ns);  // rr ring entry of the next iteration
        if (num_base) fg_acc_zero(fg_acc_ptr(acc, c.b, num_base + (it + 1) % 3), ns);
        if (threadIdx.x == 0) {
            info[c.b].final_residual = crit;
            info[c.b].used_iterations = it - 1;
            if (prof_active) atomicAdd(prof_active, 1);
            fg_best_decide(best, c.b, crit, it);
        }
    }
    const fg_real beta = first ? 0.f : (fg_real)(num_new / num_old);
    __shared__ fg_real lds[4];
    const size_t base = (size_t)c.b * g.n;
    fg_real part[1] = {0.f};
    if (c.valid) {
        const FgMetric<DIMS, VEC> m = fg_metrics<DIMS, VEC>(g, c);
        const FgNbr<DIMS, VEC> rA = fg_gather<DIMS, VEC>(rA_ + base, c);
        FgNbr<DIMS, VEC> p = fg_gather<DIMS, VEC>(z_ + base, c);
        if (!first) {
            const FgNbr<DIMS, VEC> po = fg_gather<DIMS, VEC>(pin_ + base, c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                p.c.v[e] += beta * po.c.v[e];
                p.xm.v[e] += beta * po.xm.v[e];
                p.xp.v[e] += beta * po.xp.v[e];
                p.ym.v[e] += beta * po.ym.v[e];
                p.yp.v[e] += beta * po.yp.v[e];
                if constexpr (DIMS == 3) {
                    p.zm.v[e] += beta * po.zm.v[e];
                    p.zp.v[e] += beta * po.zp.v[e];
                }
            }
        }
        const FgCoef<DIMS, VEC> k = fg_poisson_coef<DIMS, VEC>(c, m, rA);
        FgVec<VEC> Ap;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            Ap.v[e] = fg_apply_elem<DIMS, VEC>(k, p, e);
            part[0] += p.c.v[e] * Ap.v[e];
        }
        fg_store<VEC>(Ap_ + base + c.idx, Ap);
        fg_store<VEC>(pout_ + base + c.idx, p.c);
    }
    fg_block_sum<1>(part, lds);
    if (threadIdx.x == 0) fg_acc_add(fg_acc_ptr(acc, c.b, 3 + (it & 1)), ns, tile, (double)part[0]);
}

// CG kernel 2:  alpha = rr / pAp ; x += alpha p ; r -= alpha Ap ; rr_next += r.r
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_cg_update(FgGrid g, const fg_real* __restrict__ p_,
                                                         const fg_real* __restrict__ Ap_, fg_real* __restrict__ x_,
                                                         fg_real* __restrict__ r_, FgDacc* __restrict__ acc,
                                                         const int32_t* __restrict__ flags, FgBest best, fg_real tol, int it,
                                                         int ns, int num_base, FgDacc* __restrict__ xsum, int tiles_x, int tiles_y, int tiles) {
    // xsum (optional): sum(x_{it+1}) per env into xsum[2 b + (it & 1)], so that the mean removal of the pressure needs no pass of its
    // own (FgMeanRef, fg_internal.h: k_correct subtracts it where it copies the pressure to the block) -- as the fused CG's update does
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    if (flag_ld(flags + (c.b)) != 0) return;
    const double rr = fg_acc_total(fg_acc_ptr(acc, c.b, num_base + it % 3), ns);  // r.r or r.z
    // best-iterate tracking (FgBest): the leader of k_cg_ap decided whether x_it (x on entry) is worth keeping
    const bool save = best.save_at[c.b] == it;
    const double pAp = fg_acc_total(fg_acc_ptr(acc, c.b, 3 + (it & 1)), ns);
    const fg_real alpha = (fg_real)(rr / pAp);
    const unsigned tile = fg_xcd_remap(blockIdx.x, gridDim.x) % tiles;
    if (tile == 0 && threadIdx.x < 64) {
        fg_acc_zero(fg_acc_ptr(acc, c.b, 3 + ((it + 1) & 1)), ns);  // next pAp
        if (xsum && threadIdx.x == 0) acc_st(xsum + (2 * c.b + ((it + 1) & 1)), 0.0);      // filled by the update of iteration it + 1
    }
    __shared__ fg_real lds[8];
    const size_t base = (size_t)c.b * g.n;
    fg_real part[2] = {0.f, 0.f};
    if (c.valid) {
        const FgVec<VEC> p = fg_load<VEC>(p_ + base + c.idx);
        const FgVec<VEC> Ap = fg_load<VEC>(Ap_ + base + c.idx);
        FgVec<VEC> x = fg_load<VEC>(x_ + base + c.idx);
        FgVec<VEC> r = fg_load<VEC>(r_ + base + c.idx);
        if (save) fg_store<VEC>(best.best_x + base + c.idx, x);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            x.v[e] += alpha * p.v[e];
            r.v[e] -= alpha * Ap.v[e];
            part[0] += r.v[e] * r.v[e];
            part[1] += x.v[e];
        }
        fg_store<VEC>(x_ + base + c.idx, x);
        fg_store<VEC>(r_ + base + c.idx, r);
    }
    if (xsum) {
        const fg_real tot = fg_block_sum_lanes<2>(part, lds);      // thread q holds value q (same per-wave sums and order as fg_block_sum)
        if (threadIdx.x == 0) fg_acc_add(fg_acc_ptr(acc, c.b, (it + 1) % 3), ns, tile, (double)tot);
        if (threadIdx.x == 1) acc_add(xsum + (2 * c.b + (it & 1)), (double)tot);
    } else {
        fg_block_sum<1>(*reinterpret_cast<fg_real (*)[1]>(part), lds);
        if (threadIdx.x == 0) fg_acc_add(fg_acc_ptr(acc, c.b, (it + 1) % 3), ns, tile, (double)part[0]);
    }
}

// Bookkeeping after the last launched iteration `it` (evaluates rr_{it+1}); one wave per env.  `mirror` (optional) is the
// host-pinned copy of info: thread 0 of every env writes its entry there, so a convergence poll is a stream
// synchronise without a device-to-host copy (the copy kernel + its launch cost ~6 us per poll, 4-5 polls per PISO step).
constexpr int CG_CHECK_WAVES = 16;      // envs per workgroup: their result words leave as one 256-byte run (FgPollOut, fg_internal.h)
__global__ __launch_bounds__(64 * CG_CHECK_WAVES) void k_cg_check(FgDacc* __restrict__ acc, int32_t* __restrict__ flags, fg_solve_info* __restrict__ info,
                                                                  fg_solve_info* __restrict__ mirror, fg_real tol, int it, int n, int B, int final_pass, int ns,
                                                                  FgPollOut poll = FgPollOut{nullptr, 0}) {
    __shared__ uint32_t stage[CG_CHECK_WAVES * 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = blockIdx.x * CG_CHECK_WAVES, b = b0 + wave;      // one wave per env (fg_acc_total is a wave's shuffle tree)
    const bool valid = b < B;
    if (valid && flag_ld(flags + (b)) == 0) {
        const fg_real crit = fg_rms(fg_acc_total(fg_acc_ptr(acc, b, (it + 1) % 3), ns), n);
        if (lane == 0) {
            info[b].final_residual = crit;
            info[b].used_iterations = it;
            if (!(crit >= tol)) {
                const bool finite = isfinite(crit);
                flag_st(flags + (b), finite ? 1 : 2);
                info[b].converged = finite ? 1 : 0;
                info[b].is_finite = finite ? 1 : 0;
            } else if (final_pass) {
                info[b].converged = 0;
                info[b].is_finite = 1;
            }
        }
    }
    if (!mirror) return;
    const bool writer = valid && lane == 0;
    if (poll.gran) {
        uint32_t w[2] = {0u, 0u};
        if (writer) { const fg_solve_info v = info[b]; w[0] = __float_as_uint((float)v.final_residual); w[1] = fg_info_word(v); }
        fg_poll_publish_records<2>(poll, b0, min(CG_CHECK_WAVES, B - b0), wave, w, writer, stage);
    } else if (writer) {
        mirror[b] = info[b];
        fg_poll_publish(poll, b);      // (after the entry: the host spins on this word instead of synchronising the stream)
    }
}

__global__ void k_cg_begin(const fg_real* __restrict__ dt, FgCgBegin q, int B) {
    if ((int)blockIdx.x < B) fg_cg_begin_env(q, dt, blockIdx.x);
}

__global__ void k_zero_name(FgDacc* __restrict__ acc, int name, int B) {
    const int b = blockIdx.x;
    if (b < B && threadIdx.x < FG_CG_SLOTS) acc_st(fg_acc_ptr(acc, b, name) + threadIdx.x, 0.0);
}

// Unconverged envs get the best iterate seen back (returnBestResult): x = best_x where its residual beats the final one.
__global__ __launch_bounds__(FG_BLOCK) void k_cg_restore_best(fg_real* __restrict__ x, fg_solve_info* __restrict__ info,
                                                               fg_solve_info* __restrict__ mirror, FgBest best, int n) {
    const int b = blockIdx.y;
    const fg_solve_info I = info[b];
    const fg_real have = best.saved_crit[b];
    const bool worse = !(I.final_residual <= have);   // also true for a NaN final residual
    if (I.converged || !worse || !isfinite(have)) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        x[(size_t)b * n + i] = best.best_x[(size_t)b * n + i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        mirror[b].final_residual = have;   // info itself is read by the other workgroups of this launch: leave it alone
        mirror[b].is_finite = 1;
    }
}

}  // namespace


int fg_poisson_apply_launch(const fg_state* s, const fg_real* rA, const fg_real* x, fg_real* y, hipStream_t st) {
    int zc;
    if (fg_zmarch_ok(s, &zc)) return fg_zmarch_apply(s, rA, x, y, zc, st);
    FG_DISPATCH(s, {
        const FgLaunch L = fg_launch_geometry<DIMS, VEC>(s->grid);
        hipLaunchKernelGGL((k_poisson_apply<DIMS, VEC>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, rA, x, y, L.tiles_x,
                           L.tiles_y, L.tiles);
    });
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_poisson_jacobi_launch(const fg_state* s, const fg_real* rA, const fg_real* b, const fg_real* x, fg_real* xnew,
                             fg_real omega, hipStream_t st) {
    int zc;
    if (fg_zmarch_ok(s, &zc)) return fg_zmarch_relax(s, rA, b, x, xnew, omega, -1, zc, st);
    FG_DISPATCH(s, {
        const FgLaunch L = fg_launch_geometry<DIMS, VEC>(s->grid);
        hipLaunchKernelGGL((k_poisson_relax<DIMS, VEC, false>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, rA, b, x, xnew,
                           omega, 0, L.tiles_x, L.tiles_y, L.tiles);
    });
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_poisson_rbgs_launch(const fg_state* s, const fg_real* rA, const fg_real* b, fg_real* x, fg_real omega, int color,
                           hipStream_t st) {
    int zc;
    if (fg_zmarch_ok(s, &zc)) return fg_zmarch_relax(s, rA, b, x, x, omega, color, zc, st);
    FG_DISPATCH(s, {
        const FgLaunch L = fg_launch_geometry<DIMS, VEC>(s->grid);
        hipLaunchKernelGGL((k_poisson_relax<DIMS, VEC, true>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, rA, b, x, x,
                           omega, color, L.tiles_x, L.tiles_y, L.tiles);
    });
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

// Host driver of the batched CG.  p is double-buffered: a.p is buffer 0, s->w[6] buffer 1.
int fg_fd_rowmean_prefactor(fg_state* s, const fg_real* dt, hipStream_t st) {
#if !FG_F64
    if (!(s->fd_Qx && fg_fcg_ok(s) && fg_fd_rowmean_ok(s)) || s->fd_row_epoch == s->rA_epoch || dt == nullptr) return FG_OK;
    const bool parts = s->fd_row_part && s->fd_row_part_epoch == s->rA_epoch;      // the assembly left the row sums
    if (parts && fg_fd_tridiag_can_factor(s)) return FG_OK;      // round 6: the first tridiagonal solve makes the factors itself
    if (int rc = fg_fd_rowmean_factor(s, s->rA, dt, st, parts ? s->fd_row_part : nullptr, (s->grid.nx + 63) / 64)) return rc;
    s->fd_row_epoch = s->rA_epoch;
#else
    (void)s; (void)dt; (void)st;
#endif
    return FG_OK;
}

int fg_cg_slots(const fg_state* s) {   // ~cg_wgs_per_slot workgroups per accumulator slot, power of two
    int tiles_per_env = 1;
    FG_DISPATCH(s, { tiles_per_env = fg_launch_geometry<DIMS, VEC>(s->grid).tiles; });
    int ns = 1;
    while (ns < FG_CG_SLOTS && tiles_per_env / ns > s->cg_wgs_per_slot) ns *= 2;
    return ns;
}

int fg_cg_solve(fg_state* s, const FgCgArgs& a, fg_solve_info* info_host, hipStream_t st) {
    const int B = s->grid.B, n = s->grid.n;
    const dim3 sg(B), sb(64);
    fg_real* pbuf[2] = {a.p, s->w[6]};
    int tiles_per_env = 1;
    FG_DISPATCH(s, { tiles_per_env = fg_launch_geometry<DIMS, VEC>(s->grid).tiles; });
    int zc = 0;
    const bool zmarch = fg_zmarch_ok(s, &zc);
    const int ns = fg_cg_slots(s);
    (void)tiles_per_env;
    bool start_ready = false;      // k_div started this solve (FgCgStart): no k_cg_residual launch
    bool start_fwd = false;
    {   // state already prepared by the k_div that built this right-hand side (FgCgBegin, fg_cg.h)?
        const bool ready = s->cg_ready_ns == ns && s->cg_ready_best == s->cg_return_best && s->cg_ready_dt == a.dt;
        start_ready = ready && s->cg_start_ready && !a.use_x0 && a.b == s->div && a.r == s->w[0] && a.x == s->p_result;
        start_fwd = start_ready && s->cg_start_ready == 2;      // ... and w[3] holds Qx^T r_0 (k_fcg_div_fwd)
        (void)start_fwd;
        s->cg_ready_ns = 0; s->bicg_ready_nc = 0; s->cg_start_ready = 0;
        if (!ready) {
            FgCgBegin q;
            q.acc = s->cg_acc; q.flags = s->flags; q.info = s->info_dev; q.mean_sums = s->acc; q.best = s->cg_best;
            q.track_best = s->cg_return_best; q.ns = ns; q.xsum = s->fcg_xsum;
            hipLaunchKernelGGL(k_cg_begin, sg, sb, 0, st, a.dt, q, B);
        }
    }
    if (!start_ready)
    FG_DISPATCH(s, {
        const FgLaunch L = fg_launch_geometry<DIMS, VEC>(s->grid);
        hipLaunchKernelGGL((k_cg_residual<DIMS, VEC>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, a.rA, a.b, a.x, a.r,
                           s->cg_acc, s->flags, a.use_x0, 0, ns, L.tiles_x, L.tiles_y, L.tiles);
    });
    const int check_every = a.check_every > 0 ? a.check_every : 16;
    const int nb = a.precond ? 5 : 0;
    const int acc_stride = FG_CG_NAMES * FG_CG_SLOTS;
    fg_real* zvec = a.precond ? s->w[5] : a.r;
    FgCgJudge judge;
    judge.acc = s->cg_acc; judge.flags = s->flags; judge.info = s->info_dev; judge.tol = a.tol; judge.it = -1; judge.n = n; judge.ns = ns;
    s->fcg_mean_ready = 0;
    s->fcg_check0_ran = 0; s->fcg_lazy_on = 0; s->fcg_spec_done = 0;
#if !FG_F64
    const bool fused = a.precond && s->fd_Qx && fg_fcg_ok(s);
#else
    const bool fused = false;
#endif
    if (a.precond && !fused) {
        if (!s->fd_Qx) { fg_set_error("preconditioned CG requested but fg_set_fd_preconditioner was not called"); return FG_ERR_INVALID_ARG; }
        // z0 = M^-1 r0, r0.z0; the residual check of x0 (flags for already-converged envs) is taken by the first kernel of the
        // application (FgCgJudge, fg_cg.h) -- as is the one after every iteration below: no k_cg_check launch outside the polls
        judge.it = -1;
        if (int rc = fg_fd_apply(s, a.r, zvec, s->cg_acc + (size_t)(nb + 0) * FG_CG_SLOTS, acc_stride, ns, B, st, &judge)) return rc;
    }
    bool done = false, info_fresh = false;
    int active_est = (B + 3) / 4;  // envs expected to still iterate after the first iteration, refreshed by every poll
    int next_poll = a.precond ? (s->pred_cg[a.kind & 3] + 1 > 1 ? s->pred_cg[a.kind & 3] + 1 : 1) : check_every;
    // sum(x) rides in the update kernels when the solve starts from zero inside the PISO step (a.x = pressureResult): the mean removal
    // then needs no pass of its own (fcg_mean_ready, consumed by fg_piso_step's last corrector)
    FgDacc* mean_sums = (!a.use_x0 && a.x == s->p_result && s->fcg_xsum) ? s->fcg_xsum : nullptr;
    int it = 0;
#if !FG_F64
    if (fused) {
        // Three launches per iteration (fg_fftcg.hip): F'(it) = vector updates + forward transform, L = verdict + per-mode Thomas solve,
        // I'(it + 1) = inverse transform + operator + dot products.  Polls sit behind F' exactly where they sat behind k_cg_update.
        FcgVectors v;
        v.x = a.x; v.r = a.r; v.t1 = s->w[3]; v.z = zvec; v.w = s->w[4]; v.p = a.p; v.s = a.Ap;
        FgCgLead lead;
        lead.best = s->cg_best;
        // preconditioner of this solve: the row-mean operator (per-env factors, made once per 1/A field: both correctors of a PISO
        // step share them) where it applies, the grid's A = 1 operator otherwise
        const bool rowm = fg_fd_rowmean_ok(s);
        const float* factor_from = nullptr;      // != nullptr: the first tridiagonal launch of this solve makes the factors (round 6)
        if (rowm && !(a.rA == s->rA && s->fd_row_epoch == s->rA_epoch)) {
            const bool parts = a.rA == s->rA && s->fd_row_part && s->fd_row_part_epoch == s->rA_epoch;      // the assembly left the row sums
            if (parts && fg_fd_tridiag_can_factor(s)) factor_from = s->fd_row_part;
            else if (int rc = fg_fd_rowmean_factor(s, a.rA, a.dt, st, parts ? s->fd_row_part : nullptr, (s->grid.nx + 63) / 64)) return rc;
            s->fd_row_epoch = (a.rA == s->rA) ? s->rA_epoch : -1;
        }
        judge.it = -1;
        if (!(start_ready && start_fwd))      // (k_fcg_div_fwd already transformed r_0: the verdict on x_0 is the tridiagonal kernel's alone)
            if (int rc = fg_fd_dct_forward(s, a.r, v.t1, st, 0, &judge)) return rc;      // u = Qx^T r_0 (the verdict on x_0 rides here)
        lead.judge = judge;
        if (int rc = fg_fd_tridiag(s, v.t1, st, &lead, rowm, factor_from, a.dt)) return rc;
        // a solve started by k_fcg_div_fwd never stored r_0 = b and x_0 = 0: until the first update has written r and x, r is the
        // right-hand side itself and x is known
        const fg_real* r0 = (start_ready && start_fwd) ? a.b : nullptr;
        {
            FcgVectors v0 = v;
            if (r0) v0.r = const_cast<fg_real*>(r0);      // (read only by the inverse kernel)
            if (int rc = fg_fcg_inv_apply(s, v0, a.rA, 0, ns, st, s->fcg_first)) return rc;
        }
        if (s->fcg_first && a.max_iterations >= 1) {
            // C(0): the first iterate is judged from I'(0)'s dot products (k_fcg_check0).  Polled when the previous solve of this kind
            // ended after one iteration: if every env ends here nothing is left to do but write x_1 = alpha z -- or not even that
            // (lazy_ok: the corrector reads alpha z itself)
            const bool poll0 = next_poll <= 1;
            const FgPollOut po = poll0 ? fg_poll_next(&s->poll) : FgPollOut{nullptr, 0};
            if (poll0) fg_prof_prefetch(s, st);
            if (int rc = fg_fcg_check0(s, a.tol, ns, st, po)) return rc;
            bool spec = false;
            if (poll0 && s->fcg_spec && s->fcg_spec_fn && a.lazy_ok && r0 && mean_sums) {
                // the corrector in its unstored-pressure form, behind the verdict kernel: it runs while the host waits (fg_internal.h)
                s->fcg_lazy_z = v.z;
                if (int rc = s->fcg_spec_fn(s->fcg_spec_ctx)) return rc;
                spec = true;
            }
            if (poll0) {
                fg_htrace("cg_check_launched");
                if (int rc = fg_poll_wait_infos(&s->poll, po, 0, B, s->info_pinned, st)) return rc;
                fg_htrace("cg_poll_done");
                info_fresh = true;
                s->fcg_first_polls += 1;
                bool all = true, all_ok = true;
                for (int b = 0; b < B; ++b) {
                    all = all && (s->info_pinned[b].converged || !s->info_pinned[b].is_finite);
                    all_ok = all_ok && s->info_pinned[b].converged && s->info_pinned[b].is_finite;
                }
                if (all) {
                    if (a.lazy_ok && r0 && all_ok && mean_sums) {
                        s->fcg_lazy_on = 1; s->fcg_lazy_z = v.z; s->fcg_unstored += 1; s->fcg_spec_done = spec ? 1 : 0;
                    } else {
                        if (int rc = fg_fcg_update_fwd(s, v, 0, 1, ns, st, r0)) return rc;      // (every env takes the short path)
                    }
                    done = true;
                } else {
                    next_poll = 2;
                }
            }
        }
        int first = 1;
        // restart period of THIS recurrence: s = P p is carried by a recurrence of its own here (s = w + beta s), so r and the true
        // residual drift apart faster than in the classic form once a solve stagnates at fp32 round-off (a solve asked for more than
        // fp32 can give then "improves" only in its recurrence: the best-iterate bookkeeping kept an iterate whose true residual was
        // 1.7e-4 where the classic form keeps 6e-5, tests/test_gpu_parity.py::test_fd_preconditioner_fast_cosine_transform).  The
        // true residual is therefore recomputed every ten iterations at the latest; solves of the envs take one to three.
        const int fused_reset = (a.reset_steps > 0 && a.reset_steps < 10) ? a.reset_steps : 10;
        for (; it < a.max_iterations && !done; ++it) {
            if (int rc = fg_fcg_update_fwd(s, v, it, first, ns, st, it == 0 ? r0 : nullptr)) return rc;      // (first is 1 again after a restart: x and r exist by then)
            if (first) {      // p_it = z_it, s_it = w_it: the buffers change roles instead of being copied
                fg_real* t = v.p; v.p = v.z; v.z = t;
                t = v.s; v.s = v.w; v.w = t;
                first = 0;
            }
            const bool poll = (it + 1 >= next_poll || it + 1 == a.max_iterations);
            if (poll) {
                next_poll = it + 1 + check_every;
                const int final_pass = (it + 1 == a.max_iterations);
                fg_prof_prefetch(s, st);
                const FgPollOut po = fg_poll_next(&s->poll);
                hipLaunchKernelGGL(k_cg_check, dim3((B + CG_CHECK_WAVES - 1) / CG_CHECK_WAVES), dim3(64 * CG_CHECK_WAVES), 0, st, s->cg_acc, s->flags, s->info_dev, s->info_pinned, a.tol, it,
                                   n, B, final_pass, ns, po);
                fg_htrace("cg_check_launched");
                if (int rc = fg_poll_wait_infos(&s->poll, po, 0, B, s->info_pinned, st)) return rc;
                fg_htrace("cg_poll_done");
                info_fresh = true;
                done = true;
                for (int b = 0; b < B; ++b) done = done && (s->info_pinned[b].converged || !s->info_pinned[b].is_finite);
                if (done) break;
            }
            if (it + 1 < a.max_iterations) {
                judge.it = it;
                if ((it + 2) % fused_reset == 0) {
                    // residual restart (cg_solver_kernel.cu:281-302): r = b - P x, then the recurrence starts over (beta = 0)
                    hipLaunchKernelGGL(k_zero_name, sg, sb, 0, st, s->cg_acc, (it + 1) % 3, B);
                    FG_DISPATCH(s, {
                        const FgLaunch L = fg_launch_geometry<DIMS, VEC>(s->grid);
                        hipLaunchKernelGGL((k_cg_residual<DIMS, VEC>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, a.rA, a.b, a.x,
                                           a.r, s->cg_acc, s->flags, 1, (it + 1) % 3, ns, L.tiles_x, L.tiles_y, L.tiles);
                    });
                    if (int rc = fg_fd_dct_forward(s, a.r, v.t1, st, 0, nullptr)) return rc;
                    first = 1;
                }
                lead.judge = judge;
                if (int rc = fg_fd_tridiag(s, v.t1, st, &lead, rowm)) return rc;
                if (int rc = fg_fcg_inv_apply(s, v, a.rA, it + 1, ns, st)) return rc;
            }
        }
        s->fcg_mean_ready = a.use_x0 ? 0 : 1;     // (sum(x) of a solve that takes no iteration is only known for x_0 = 0)
    } else
#endif
    for (; it < a.max_iterations && !done; ++it) {
        int first = (it == 0);
        if (a.reset_steps > 0 && it > 0 && (it + 1) % a.reset_steps == 0) {
            // residual restart (cg_solver_kernel.cu:281-302): r = b - P x, p = r
            hipLaunchKernelGGL(k_zero_name, sg, sb, 0, st, s->cg_acc, it % 3, B);
            FG_DISPATCH(s, {
                const FgLaunch L = fg_launch_geometry<DIMS, VEC>(s->grid);
                hipLaunchKernelGGL((k_cg_residual<DIMS, VEC>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, a.rA, a.b, a.x,
                                   a.r, s->cg_acc, s->flags, 1, it % 3, ns, L.tiles_x, L.tiles_y, L.tiles);
            });
            first = 1;
        }
        // p double buffer: read p_{it-1} from pbuf[(it+1)&1], write p_it to pbuf[it&1]
        const fg_real* p_in = pbuf[(it + 1) & 1];
        fg_real* p_out = pbuf[it & 1];
        // algorithmic bytes per env of k_cg_ap: z, rA read + p, Ap written (+ p_in read unless first); 2d+1-point
        // stencil = (4d + 1) flops + 2 (p update) + 2 (dot) per cell.  k_cg_update: x, r, p, Ap read + x, r written.
        const int slot_ap = fg_prof_slot(s, FG_PK_CG_AP, FG_PROF_SELF, B, (double)n * (first ? 16.0 : 20.0),
                                         (double)n * (4.0 * s->grid.dims + 5.0), st);
        if (zmarch) {
            if (int rc = fg_zmarch_cg_ap(s, a.rA, zvec, p_in, p_out, a.Ap, s->cg_acc, s->flags, s->info_dev, slot_ap,
                                         a.tol, it, first, ns, nb, zc, st))
                return rc;
        } else {
            FG_DISPATCH(s, {
                const FgLaunch L = fg_launch_geometry<DIMS, VEC>(s->grid);
                FG_LAUNCH_P(s, slot_ap, (k_cg_ap<DIMS, VEC>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, a.rA, zvec, p_in,
                            p_out, a.Ap, s->cg_acc, s->flags, s->info_dev,
                            slot_ap >= 0 ? s->prof.active_dev + slot_ap : nullptr, s->cg_best, a.tol, it, first, ns,
                            nb, L.tiles_x, L.tiles_y, L.tiles);
            });
        }
        const int slot_up = fg_prof_slot(s, FG_PK_CG_UPDATE, s->flags, B, (double)n * 24.0, (double)n * 6.0, st);
        FG_DISPATCH(s, {
            const FgLaunch L = fg_launch_geometry<DIMS, VEC>(s->grid);
            FG_LAUNCH_P(s, slot_up, (k_cg_update<DIMS, VEC>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, p_out, a.Ap, a.x,
                        a.r, s->cg_acc, s->flags, s->cg_best, a.tol, it, ns, nb, mean_sums, L.tiles_x, L.tiles_y, L.tiles);
        });
        const bool poll = (it + 1 >= next_poll || it + 1 == a.max_iterations);
        if (poll) next_poll = it + 1 + check_every;
        if (poll) {
            const int final_pass = (it + 1 == a.max_iterations);
            fg_prof_prefetch(s, st);       // (in front of the polled kernel: its completion then covers the copy)
            const FgPollOut po = fg_poll_next(&s->poll);
            hipLaunchKernelGGL(k_cg_check, dim3((B + CG_CHECK_WAVES - 1) / CG_CHECK_WAVES), dim3(64 * CG_CHECK_WAVES), 0, st, s->cg_acc, s->flags, s->info_dev, s->info_pinned, a.tol, it,
                               n, B, final_pass, ns, po);
            // one read-back serves the poll and the result: k_cg_check above mirrored info (converged / is_finite of every
            // env) into the pinned host copy, and nothing is launched between the last poll and the end of the solve.
            // The poll comes BEFORE the preconditioner of the next iteration: polls are scheduled where the previous solve
            // finished, so they usually end the solve, and three kernels of M^-1 that would find every env converged
            // cost more than the idle round trip of a poll that does not.
            if (int rc = fg_poll_wait_infos(&s->poll, po, 0, B, s->info_pinned, st)) return rc;
            info_fresh = true;
            done = true;
            active_est = 0;
            for (int b = 0; b < B; ++b) {
                const bool fin = s->info_pinned[b].converged || !s->info_pinned[b].is_finite;
                done = done && fin;
                active_est += !fin;
            }
            if (active_est < 1) active_est = 1;
            if (done) break;
        }
        if (a.precond && it + 1 < a.max_iterations) {
            // z = M^-1 r and r.z of the next iteration (envs that just converged are skipped via flags)
            judge.it = it;
            if (int rc = fg_fd_apply(s, a.r, zvec, s->cg_acc + (size_t)(nb + (it + 1) % 3) * FG_CG_SLOTS, acc_stride, ns, active_est, st, &judge))
                return rc;
        }
    }
    if (!fused && mean_sums) s->fcg_mean_ready = 1;      // (a residual restart recomputes r only: the sums of x stay valid)
    if (!info_fresh) {
        FG_HIP_CHECK(hipMemcpyAsync(s->info_pinned, s->info_dev, sizeof(fg_solve_info) * B, hipMemcpyDeviceToHost, st));
        FG_HIP_CHECK(hipStreamSynchronize(st));
    }
    if (int prc = fg_prof_collect(s, st)) return prc;
    bool failed = false;
    for (int b = 0; b < B; ++b) failed = failed || !s->info_pinned[b].converged;
    if (failed) {  // rare path: hand back the best iterate instead of the last one (results land in the pinned mirror)
        s->fcg_mean_ready = 0;     // (x may be replaced: its sum is no longer the one the update kernels left)
        hipLaunchKernelGGL(k_cg_restore_best, dim3(32, B), dim3(FG_BLOCK), 0, st, a.x, s->info_dev, s->info_pinned, s->cg_best, n);
        FG_HIP_CHECK(hipStreamSynchronize(st));
    }
    int rc = FG_OK;
    if (a.precond) {
        int used_max = 0;
        for (int b = 0; b < B; ++b) used_max = s->info_pinned[b].used_iterations > used_max ? s->info_pinned[b].used_iterations : used_max;
        s->pred_cg[a.kind & 3] = used_max;
    }
    for (int b = 0; b < B; ++b) {
        if (info_host) info_host[b] = s->info_pinned[b];
        if (!s->info_pinned[b].is_finite) rc = FG_ERR_NOT_FINITE;
        else if (!s->info_pinned[b].converged && rc == FG_OK) rc = FG_ERR_NOT_CONVERGED;
    }
    FG_HIP_CHECK(hipGetLastError());
    return rc;
}

#if !FG_F64
// ---------------------------------------------------------------------------------------------------------------------------
// fp64 repeat of a pressure solve that ended non-finite (fg_rung64.h; the reference's solver_double_fallback on a solve that runs with
// returnBestResult, PISOtorch_diff.py:410-445): plain CG in double on the fp32 matrix entries, from zero, same criterion / cap.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k64_papply(FgGrid g, const float* __restrict__ rA_, const double* __restrict__ x, double* __restrict__ y,
                                                       int tiles_x, int tiles_y, int tiles) {
    const FgCtx<DIMS, 1> c = fg_make_ctx<DIMS, 1>(g, tiles_x, tiles_y, tiles);
    if (!c.valid) return;
    const FgMetric<DIMS, 1> m = fg_metrics<DIMS, 1>(g, c);
    const FgNbr<DIMS, 1> rA = fg_gather<DIMS, 1>(rA_, c);
    const FgCoef<DIMS, 1> k = fg_poisson_coef<DIMS, 1>(c, m, rA);      // the matrix entries as the fp32 solver forms them, promoted below
    const R64Nbr<DIMS> X = r64_gather<DIMS>(x, c);
    double v = (double)k.xm[0] * (X.xm - X.c) + (double)k.xp[0] * (X.xp - X.c) + (double)k.ym[0] * (X.ym - X.c) + (double)k.yp[0] * (X.yp - X.c);
    if constexpr (DIMS == 3) v += (double)k.zm[0] * (X.zm - X.c) + (double)k.zp[0] * (X.zp - X.c);
    y[c.idx] = v;
}
}  // namespace

int fg_rung64_cg(fg_state* s, const FgCgArgs& a, fg_solve_info* info, bool all_envs, hipStream_t st) {
    R64 w;
    if (int rc = w.init(s, st)) return rc;
    const int B = s->grid.B, n = s->grid.n, d = s->grid.dims;
    FgGrid g1 = s->grid; g1.B = 1;
    double *r = w.v[0], *p = w.v[1], *Ap = w.v[2], *x = w.v[3], *bb = w.v[4];
    auto apply = [&](int b, const double* in, double* out) {
        const float* rA = a.rA + (size_t)b * n;
        if (d == 2) { FgLaunch L = fg_launch_geometry<2, 1>(g1); hipLaunchKernelGGL((k64_papply<2>), L.grid, dim3(FG_BLOCK), 0, st, g1, rA, in, out, L.tiles_x, L.tiles_y, L.tiles); }
        else { FgLaunch L = fg_launch_geometry<3, 1>(g1); hipLaunchKernelGGL((k64_papply<3>), L.grid, dim3(FG_BLOCK), 0, st, g1, rA, in, out, L.tiles_x, L.tiles_y, L.tiles); }
    };
    auto rms = [&](double q) { return sqrt(q / (double)n); };
    std::vector<float> dt_host;
    if (a.dt) { dt_host.resize(B); FG_HIP_CHECK(hipMemcpyAsync(dt_host.data(), a.dt, sizeof(float) * B, hipMemcpyDeviceToHost, st)); FG_HIP_CHECK(hipStreamSynchronize(st)); }
    int rc_all = FG_OK;
    for (int b = 0; b < B; ++b) {
        if (a.dt && !(dt_host[b] > 0.f)) continue;
        fg_solve_info& I = info[b];
        if (!all_envs && I.is_finite) continue;
        w.load(bb, a.b + (size_t)b * n);
        w.zero(x);
        w.axpby(r, 1.0, bb, 0.0, bb);
        w.axpby(p, 1.0, r, 0.0, r);
        double rr = w.dot(r, r);
        I.used_iterations = -1; I.converged = 0; I.is_finite = 1; I.final_residual = (float)rms(rr);
        for (int it = 0; it < a.max_iterations; ++it) {
            if (!(rms(rr) >= (double)a.tol)) { I.converged = std::isfinite(rr) ? 1 : 0; break; }
            if (a.reset_steps > 0 && it > 0 && (it + 1) % a.reset_steps == 0) {      // residual restart (cg_solver_kernel.cu:281-302)
                apply(b, x, Ap);
                w.axpby(r, 1.0, bb, -1.0, Ap);
                w.axpby(p, 1.0, r, 0.0, r);
                rr = w.dot(r, r);
            }
            apply(b, p, Ap);
            const double pAp = w.dot(p, Ap);
            const double alpha = rr / pAp;
            w.axpby(x, 1.0, x, alpha, p);
            w.axpby(r, 1.0, r, -alpha, Ap);
            const double rr_new = w.dot(r, r);
            w.axpby(p, rr_new / rr, p, 1.0, r);
            rr = rr_new;
            I.used_iterations = it; I.final_residual = (float)rms(rr);
            if (!std::isfinite(rr)) { I.is_finite = 0; break; }
            if (it + 1 == a.max_iterations && !(rms(rr) >= (double)a.tol)) I.converged = 1;
        }
        if (w.err) return w.err;
        w.store(a.x + (size_t)b * n, x);
        if (!I.is_finite) rc_all = FG_ERR_NOT_FINITE;
        else if (!I.converged && rc_all == FG_OK) rc_all = FG_ERR_NOT_CONVERGED;
    }
    FG_HIP_CHECK(hipStreamSynchronize(st));
    FG_HIP_CHECK(hipGetLastError());
    return rc_all;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Opt-in mixed-precision refinement of the pressure solve (fg_set_pressure_refinement; round 6).  The fp32 CG ends
// where its residual cannot be resolved any further -- an ABSOLUTE RMS tolerance of 1e-7 on a 40 : 1 wall-refined grid leaves the
// projected velocity 5-7e-4 of the forcing scale away from the fp64 answer (tests/test_gpu_config3.py) although every kernel is right
// to 1e-8 (the fp64 twins).  Here the iterate is kept in fp64: r = b - P x is formed in fp64 with the fp32 matrix entries promoted
// (k64_papply: the matrix the fp32 solver sees, as the reference's fp64 fallback promotes its CSR values, PISOtorch_diff.py:418-445),
// the CORRECTION P d = r / |r| is solved by the same fp32 solver (fused FD-preconditioned CG) to a relative tolerance, x += |r| d --
// the pattern of the multi-block path's refined BiCGStab (fg_mb_krylov.hip).  Per env, a few launches and two host reads per outer
// iteration: an accuracy mode, not a fast path.  The mean of r is removed (P 1 = 1^T P = 0 exactly: what is left of mean(b) is the
// right-hand side's own rounding and no iterate can reduce it).
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
__global__ void k64_fill(double* __restrict__ y, double v, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = v;
}
}  // namespace
int fg_refine64_pressure(fg_state* s, const FgCgArgs& a0, fg_solve_info* info_host, hipStream_t st) {
    R64 w;
    if (int rc = w.init(s, st)) return rc;
    const int B = s->grid.B, n = s->grid.n, d = s->grid.dims;
    FG_REQUIRE(d >= 2, FG_ERR_UNSUPPORTED, "pressure refinement: 2-D / 3-D grids");
    if (!s->ref64_x) FG_HIP_CHECK(hipMalloc(&s->ref64_x, sizeof(double) * (size_t)B * n));
    float* rhs32 = s->w[7];                       // (free during a pressure solve; B d n floats: room for both)
    float* dx32 = s->w[7] + (size_t)B * n;
    FgGrid g1 = s->grid; g1.B = 1;
    double *r = w.v[0], *Ap = w.v[1], *bb = w.v[2], *ones = w.v[3], *tmp = w.v[4];
    auto apply = [&](int b, const double* in, double* out) {
        const float* rA = a0.rA + (size_t)b * n;
        if (d == 2) { FgLaunch L = fg_launch_geometry<2, 1>(g1); hipLaunchKernelGGL((k64_papply<2>), L.grid, dim3(FG_BLOCK), 0, st, g1, rA, in, out, L.tiles_x, L.tiles_y, L.tiles); }
        else { FgLaunch L = fg_launch_geometry<3, 1>(g1); hipLaunchKernelGGL((k64_papply<3>), L.grid, dim3(FG_BLOCK), 0, st, g1, rA, in, out, L.tiles_x, L.tiles_y, L.tiles); }
    };
    hipLaunchKernelGGL(k64_fill, w.grid(), dim3(256), 0, st, ones, 1.0, n);
    std::vector<float> dt_host;
    if (a0.dt) { dt_host.resize(B); FG_HIP_CHECK(hipMemcpyAsync(dt_host.data(), a0.dt, sizeof(float) * B, hipMemcpyDeviceToHost, st)); FG_HIP_CHECK(hipStreamSynchronize(st)); }
    auto active = [&](int b) { return !(a0.dt && !(dt_host[b] > 0.f)); };
    for (int b = 0; b < B; ++b) if (active(b)) w.load(s->ref64_x + (size_t)b * n, a0.x + (size_t)b * n);
    std::vector<double> scale(B, 0.0), res(B, 0.0);
    int outer = 0;
    for (;; ++outer) {
        double worst = 0.0;
        for (int b = 0; b < B; ++b) {
            scale[b] = 0.0;
            if (!active(b)) continue;
            double* x64 = s->ref64_x + (size_t)b * n;
            w.load(bb, a0.b + (size_t)b * n);
            apply(b, x64, Ap);
            w.axpby(r, 1.0, bb, -1.0, Ap);
            const double mean = w.dot(r, ones) / (double)n;
            w.axpby(r, 1.0, r, -mean, ones);
            const double rms = sqrt(w.dot(r, r) / (double)n);
            if (w.err) return w.err;
            res[b] = rms;
            if (std::isfinite(rms) && rms >= (double)s->ref64_tol && outer < s->ref64_outer) {
                scale[b] = rms;
                w.axpby(r, 1.0 / rms, r, 0.0, r);
                w.store(rhs32 + (size_t)b * n, r);
            } else {
                FG_HIP_CHECK(hipMemsetAsync(rhs32 + (size_t)b * n, 0, sizeof(float) * (size_t)n, st));
            }
            worst = (std::isfinite(rms) && rms > worst) ? rms : worst;
        }
        bool any = false;
        for (int b = 0; b < B; ++b) any = any || scale[b] > 0.0;
        if (!any) break;
        // the correction: the fp32 solver on the unit-RMS residuals, from zero, to a relative tolerance
        FgCgArgs c = a0;
        c.b = rhs32; c.x = dx32; c.use_x0 = 0; c.tol = s->ref64_inner; c.lazy_ok = 0;
        const int rc = fg_cg_solve(s, c, nullptr, st);
        if (rc != FG_OK && rc != FG_ERR_NOT_CONVERGED) return rc;
        for (int b = 0; b < B; ++b) {
            if (!(scale[b] > 0.0)) continue;
            double* x64 = s->ref64_x + (size_t)b * n;
            w.load(tmp, dx32 + (size_t)b * n);
            w.axpby(x64, 1.0, x64, scale[b], tmp);
        }
        s->ref64_corrections += 1;
        (void)worst;
    }
    int rc_all = FG_OK;
    for (int b = 0; b < B; ++b) {
        if (!active(b)) continue;
        w.store(a0.x + (size_t)b * n, s->ref64_x + (size_t)b * n);
        fg_solve_info& I = s->info_pinned[b];
        I.final_residual = (float)res[b];
        I.is_finite = std::isfinite(res[b]) ? 1 : 0;
        I.converged = (std::isfinite(res[b]) && res[b] < (double)s->ref64_tol) ? 1 : 0;
        if (info_host) info_host[b] = I;
        if (!I.is_finite) rc_all = FG_ERR_NOT_FINITE;
        else if (!I.converged && rc_all == FG_OK) rc_all = FG_ERR_NOT_CONVERGED;
    }
    FG_HIP_CHECK(hipStreamSynchronize(st));
    FG_HIP_CHECK(hipGetLastError());
    return rc_all;
}
#endif
