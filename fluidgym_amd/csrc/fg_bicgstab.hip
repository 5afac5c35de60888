// Batched, multi-RHS BiCGStab on the stencil-form advection-diffusion matrix, gfx950.
//
// Replaces bicgstabSolveGPU (bicgstab_solver_kernel.cu:63-411): same un-preconditioned recurrence
// and the same RMS-residual criterion (cg_solver_kernel.cu:100-106), but all B*nc systems (env x
// velocity component, one shared matrix per env) advance in lock-step inside 5 kernels per
// iteration, with the scalars (rho, alpha, omega, ...) living in device accumulators.  The
// reference loops the right-hand sides sequentially and reads every scalar back to the host.
//
// accumulators per system (fp64): 0,1 rho ring | 2 rw.v | 3 s.s | 4 t.s | 5 t.t | 6 r.r
// derived scalars per system (fp32, `sc`): 0 alpha | 1 omega
#include <math.h>
#include <cmath>
#include <vector>

#include "fg_internal.h"
#include "fg_bicg.h"
#include "fg_rung64.h"
#include "fg_fftbicg.h"

namespace {

constexpr int A_RHO = 0, A_RV = 2, A_SS = 3, A_TS = 4, A_TT = 5, A_RR = 6;
// A_RHOE + (it & 1): the rho iteration `it` actually uses -- rw.r, or r.r after a breakdown restart (k_bicg_p; the guard of
// fg_mb_step.hip MB_BETA: an exact rho = 0 or rw.v = 0 at the fp32 rounding level must not turn into inf / NaN)
constexpr int A_RHOE = 10;

template <int DIMS, int VEC>
__device__ __forceinline__ FgVec<VEC> fg_spmv(const fg_real* __restrict__ diag, const fg_real* __restrict__ off,
                                              const fg_real* __restrict__ x, const FgCtx<DIMS, VEC>& c, size_t N) {
    // diag/off already offset to the env; x offset to the system
    const FgNbr<DIMS, VEC> X = fg_gather<DIMS, VEC>(x, c);
    const FgVec<VEC> d = fg_load<VEC>(diag + c.idx);
    FgVec<VEC> o[2 * DIMS];
#pragma unroll
    for (int f = 0; f < 2 * DIMS; ++f) o[f] = fg_load<VEC>(off + f * N + c.idx);
    FgVec<VEC> y;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        fg_real v = d.v[e] * X.c.v[e] + o[0].v[e] * X.xm.v[e] + o[1].v[e] * X.xp.v[e] + o[2].v[e] * X.ym.v[e] +
                  o[3].v[e] * X.yp.v[e];
        if constexpr (DIMS == 3) v += o[4].v[e] * X.zm.v[e] + o[5].v[e] * X.zp.v[e];
        y.v[e] = v;
    }
    return y;
}

struct SysCtx {
    int sys;       // b * nc + comp
    int comp;
    bool leader;
};

template <int DIMS, int VEC>
__device__ __forceinline__ SysCtx fg_sys(const FgCtx<DIMS, VEC>& c, int nc, int tiles) {
    SysCtx s;
    s.comp = blockIdx.y;
    s.sys = c.b * nc + s.comp;
    s.leader = (threadIdx.x == 0) && ((fg_xcd_remap(blockIdx.x, gridDim.x) % tiles) == 0);
    return s;
}

// SpMV with the matrix held in registers: the (1 + 2 DIMS) coefficient fields belong to the env, not to the system,
// so one workgroup applies them to ALL nc right-hand sides of its tile.  With one workgroup per (tile, component) the
// two/three component workgroups usually sat on different XCDs and each pulled the matrix through its own L2
// (rocprofv3 FETCH_SIZE of k_bicg_v: 86 MB per launch against 56 MB of algorithmic reads).
template <int DIMS, int VEC>
struct FgStencilRow {
    FgVec<VEC> d, o[2 * DIMS];
};
template <int DIMS, int VEC>
__device__ __forceinline__ FgStencilRow<DIMS, VEC> fg_load_row(const fg_real* __restrict__ diag,
                                                              const fg_real* __restrict__ off,
                                                              const FgCtx<DIMS, VEC>& c, size_t N) {
    FgStencilRow<DIMS, VEC> m;
    m.d = fg_load<VEC>(diag + c.idx);
#pragma unroll
    for (int f = 0; f < 2 * DIMS; ++f) m.o[f] = fg_load<VEC>(off + f * N + c.idx);
    return m;
}
template <int DIMS, int VEC>
__device__ __forceinline__ FgVec<VEC> fg_apply_row(const FgStencilRow<DIMS, VEC>& m, const fg_real* __restrict__ x,
                                                   const FgCtx<DIMS, VEC>& c) {
    const FgNbr<DIMS, VEC> X = fg_gather<DIMS, VEC>(x, c);
    FgVec<VEC> y;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        fg_real v = m.d.v[e] * X.c.v[e] + m.o[0].v[e] * X.xm.v[e] + m.o[1].v[e] * X.xp.v[e] + m.o[2].v[e] * X.ym.v[e] +
                  m.o[3].v[e] * X.yp.v[e];
        if constexpr (DIMS == 3) v += m.o[4].v[e] * X.zm.v[e] + m.o[5].v[e] * X.zp.v[e];
        y.v[e] = v;
    }
    return y;
}

// init: r = rhs - C x0 ; rw = r ; p = r ; rho_0 = rr = r.r        (grid.y = 1: loops over the nc systems of the env)
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_bicg_init(FgGrid g, BicgPtrs q, int use_x0, int tiles_x, int tiles_y,
                                                         int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    const size_t N = g.n;
    bool any = false;
    for (int comp = 0; comp < q.nc; ++comp) any = any || (flag_ld(q.flags + (c.b * q.nc + comp)) == 0);
    if (!any) return;
    FgStencilRow<DIMS, VEC> m;
    if (use_x0 && c.valid) m = fg_load_row<DIMS, VEC>(q.diag + (size_t)c.b * N, q.off + (size_t)c.b * 2 * DIMS * N, c, N);
    __shared__ fg_real lds[4];
    for (int comp = 0; comp < q.nc; ++comp) {
        const int sys = c.b * q.nc + comp;
        if (flag_ld(q.flags + (sys)) != 0) continue;
        const size_t vb = (size_t)sys * N;
        fg_real part[1] = {0.f};
        if (c.valid) {
            FgVec<VEC> r = fg_load<VEC>(q.rhs + vb + c.idx);
            if (use_x0) {
                const FgVec<VEC> y = fg_apply_row<DIMS, VEC>(m, q.x + vb, c);
#pragma unroll
                for (int e = 0; e < VEC; ++e) r.v[e] -= y.v[e];
            } else {
                FgVec<VEC> z;
#pragma unroll
                for (int e = 0; e < VEC; ++e) z.v[e] = 0.f;
                fg_store<VEC>(q.x + vb + c.idx, z);
            }
            fg_store<VEC>(q.r + vb + c.idx, r);
            fg_store<VEC>(q.rw + vb + c.idx, r);
            fg_store<VEC>(q.p + vb + c.idx, r);
#pragma unroll
            for (int e = 0; e < VEC; ++e) part[0] += r.v[e] * r.v[e];
        }
        fg_block_sum<1>(part, lds);
        if (threadIdx.x == 0) {
            FgDacc* a = q.acc + (size_t)sys * FG_ACC_DOUBLES;
            acc_add(a + A_RHO, (double)part[0]);
            acc_add(a + A_RR, (double)part[0]);
        }
        __syncthreads();
    }
}

// Kp_i: convergence check on r of iteration i-1, then p = r + beta (p - omega v)   (i > 0)
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_bicg_p(FgGrid g, BicgPtrs q, int it, int tiles_x, int tiles_y,
                                                      int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    const SysCtx s = fg_sys<DIMS, VEC>(c, q.nc, tiles);
    const int f = flag_ld(q.flags + (s.sys));
    if (f == 4) {  // converged on s in the previous iteration: K5 has applied x += alpha p, finalise
        if (s.leader) flag_st(q.flags + (s.sys), 1);
        return;
    }
    if (f != 0) return;
    FgDacc* a = q.acc + (size_t)s.sys * FG_ACC_DOUBLES;
    const fg_real crit = fg_rms(acc_ld(a + (A_RR)), g.n);
    if (!(crit >= q.tol)) {
        if (s.leader) fg_mark(q.flags, q.info, s.sys, crit, it == 0 ? -1 : it);
        return;
    }
    if (s.leader) {
        acc_st(a + (A_SS), 0.0); acc_st(a + (A_TS), 0.0); acc_st(a + (A_TT), 0.0);
        q.info[s.sys].final_residual = crit;
        q.info[s.sys].used_iterations = it - 1;
    }
    const fg_real alpha = sc_ld(q.sc + (s.sys * 2 + 0)), omega = sc_ld(q.sc + (s.sys * 2 + 1));
    const double rho_now = acc_ld(a + (A_RHO + (it & 1)));
    const fg_real beta = it == 0 ? 0.f : (fg_real)(rho_now / acc_ld(a + (A_RHOE + ((it + 1) & 1)))) * (alpha / omega);
    const bool restart = it > 0 && !isfinite(beta);   // rho of the previous iteration exactly 0, or omega 0: rw = p = r, rho = r.r
    if (s.leader) acc_st(a + (A_RHOE + (it & 1)), restart ? acc_ld(a + (A_RR)) : rho_now);
    if (it == 0 || !c.valid) return;
    const size_t vb = (size_t)s.sys * g.n;
    const FgVec<VEC> r = fg_load<VEC>(q.r + vb + c.idx);
    if (restart) {
        fg_store<VEC>(q.rw + vb + c.idx, r);
        fg_store<VEC>(q.p + vb + c.idx, r);
        return;
    }
    const FgVec<VEC> v = fg_load<VEC>(q.v + vb + c.idx);
    FgVec<VEC> p = fg_load<VEC>(q.p + vb + c.idx);
#pragma unroll
    for (int e = 0; e < VEC; ++e) p.v[e] = r.v[e] + beta * (p.v[e] - omega * v.v[e]);
    fg_store<VEC>(q.p + vb + c.idx, p);
}

// K2_i: v = C p ; rv += rw.v        (grid.y = 1: loops over the nc systems of the env)
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_bicg_v(FgGrid g, BicgPtrs q, int it, int tiles_x, int tiles_y,
                                                      int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    const size_t N = g.n;
    bool any = false;
    for (int comp = 0; comp < q.nc; ++comp) any = any || (flag_ld(q.flags + (c.b * q.nc + comp)) == 0);
    if (!any) return;
    FgStencilRow<DIMS, VEC> m;
    if (c.valid) m = fg_load_row<DIMS, VEC>(q.diag + (size_t)c.b * N, q.off + (size_t)c.b * 2 * DIMS * N, c, N);
    __shared__ fg_real lds[4];
    for (int comp = 0; comp < q.nc; ++comp) {
        const int sys = c.b * q.nc + comp;
        if (flag_ld(q.flags + (sys)) != 0) continue;  // uniform over the workgroup
        const size_t vb = (size_t)sys * N;
        fg_real part[1] = {0.f};
        if (c.valid) {
            const FgVec<VEC> y = fg_apply_row<DIMS, VEC>(m, (q.mp ? q.mp : q.p) + vb, c);
            const FgVec<VEC> rw = fg_load<VEC>(q.rw + vb + c.idx);
            fg_store<VEC>(q.v + vb + c.idx, y);
#pragma unroll
            for (int e = 0; e < VEC; ++e) part[0] += rw.v[e] * y.v[e];
        }
        fg_block_sum<1>(part, lds);
        if (threadIdx.x == 0) acc_add(q.acc + (size_t)sys * FG_ACC_DOUBLES + A_RV, (double)part[0]);
        __syncthreads();  // lds reused by the next component
    }
}

// K3_i: alpha = rho_i / rv ; s = r - alpha v (stored in r) ; ss += s.s
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_bicg_s(FgGrid g, BicgPtrs q, int it, int tiles_x, int tiles_y,
                                                      int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    const SysCtx s = fg_sys<DIMS, VEC>(c, q.nc, tiles);
    if (flag_ld(q.flags + (s.sys)) != 0) return;
    FgDacc* a = q.acc + (size_t)s.sys * FG_ACC_DOUBLES;
    const fg_real alpha_raw = (fg_real)(acc_ld(a + (A_RHOE + (it & 1))) / acc_ld(a + (A_RV)));
    const fg_real alpha = isfinite(alpha_raw) ? alpha_raw : 0.f;   // rw.v == 0 exactly: the iteration keeps its minimal-residual half
    if (s.leader) {
        sc_st(q.sc + (s.sys * 2 + 0), alpha);
        acc_st(a + (A_RHO + ((it + 1) & 1)), 0.0);  // rho slot of the next iteration
        acc_st(a + (A_RR), 0.0);                    // read by Kp_i / K2_i, re-accumulated by K5_i
    }
    const size_t vb = (size_t)s.sys * g.n;
    __shared__ fg_real lds[4];
    fg_real part[1] = {0.f};
    if (c.valid) {
        FgVec<VEC> r = fg_load<VEC>(q.r + vb + c.idx);
        const FgVec<VEC> v = fg_load<VEC>(q.v + vb + c.idx);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            r.v[e] -= alpha * v.v[e];
            part[0] += r.v[e] * r.v[e];
        }
        fg_store<VEC>(q.r + vb + c.idx, r);
    }
    fg_block_sum<1>(part, lds);
    if (threadIdx.x == 0) acc_add(a + A_SS, (double)part[0]);
}

// K4_i: t = C s ; ts += t.s ; tt += t.t        (skipped when ||s|| already meets the tolerance; grid.y = 1)
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_bicg_t(FgGrid g, BicgPtrs q, int it, int tiles_x, int tiles_y,
                                                      int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    const size_t N = g.n;
    const bool leader = (threadIdx.x == 0) && ((fg_xcd_remap(blockIdx.x, gridDim.x) % tiles) == 0);
    bool work[3] = {false, false, false};
    bool any = false;
    for (int comp = 0; comp < q.nc; ++comp) {
        const int sys = c.b * q.nc + comp;
        if (flag_ld(q.flags + (sys)) != 0) continue;
        FgDacc* a = q.acc + (size_t)sys * FG_ACC_DOUBLES;
        const fg_real crit_s = fg_rms(acc_ld(a + (A_SS)), g.n);
        if (!(crit_s >= q.tol)) {
            // converged on s (bicgstab_solver_kernel.cu:305-329): flag 4 = "K5 applies x += alpha p, then done".
            // Nothing in THIS launch depends on the flag value written here (every workgroup of the env takes this
            // branch from the same accumulator value, and reads flags only above).
            if (leader) fg_mark(q.flags, q.info, sys, crit_s, it, 4);
            continue;
        }
        work[comp] = true;
        any = true;
    }
    // every workgroup must have read flags before a leader of the same env may overwrite them: flags of an env are
    // only written by that env's leader workgroup, after its own reads; other workgroups read either value and then
    // decide by the accumulator, which is stable -> same decision.  (Same protocol as before, per system.)
    if (!any) return;
    FgStencilRow<DIMS, VEC> m;
    if (c.valid) m = fg_load_row<DIMS, VEC>(q.diag + (size_t)c.b * N, q.off + (size_t)c.b * 2 * DIMS * N, c, N);
    __shared__ fg_real lds[8];
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
        if (comp >= q.nc || !work[comp]) continue;
        const int sys = c.b * q.nc + comp;
        const size_t vb = (size_t)sys * N;
        fg_real part[2] = {0.f, 0.f};
        if (c.valid) {
            const FgVec<VEC> t = fg_apply_row<DIMS, VEC>(m, (q.ms ? q.ms : q.r) + vb, c);
            const FgVec<VEC> sv = fg_load<VEC>(q.r + vb + c.idx);
            fg_store<VEC>(q.t + vb + c.idx, t);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                part[0] += t.v[e] * sv.v[e];
                part[1] += t.v[e] * t.v[e];
            }
        }
        fg_block_sum<2>(part, lds);
        if (threadIdx.x == 0) {
            FgDacc* a = q.acc + (size_t)sys * FG_ACC_DOUBLES;
            acc_add(a + A_TS, (double)part[0]);
            acc_add(a + A_TT, (double)part[1]);
        }
        __syncthreads();
    }
}

// K5_i: x += alpha p (+ omega s) ; r = s - omega t ; rr += r.r ; rho_{i+1} += rw.r
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_bicg_x(FgGrid g, BicgPtrs q, int it, int tiles_x, int tiles_y,
                                                      int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    const SysCtx s = fg_sys<DIMS, VEC>(c, q.nc, tiles);
    const int f = flag_ld(q.flags + (s.sys));  // stable during this launch: K5 never writes flags
    if (f != 0 && f != 4) return;
    FgDacc* a = q.acc + (size_t)s.sys * FG_ACC_DOUBLES;
    const fg_real alpha = sc_ld(q.sc + (s.sys * 2 + 0));
    const bool half = (f == 4);
    const fg_real omega_raw = half ? 0.f : (fg_real)(acc_ld(a + (A_TS)) / acc_ld(a + (A_TT)));
    const fg_real omega = isfinite(omega_raw) ? omega_raw : 0.f;
    if (s.leader) {
        sc_st(q.sc + (s.sys * 2 + 1), omega);
        acc_st(a + (A_RV), 0.0);
    }
    const size_t vb = (size_t)s.sys * g.n;
    __shared__ fg_real lds[8];
    fg_real part[2] = {0.f, 0.f};
    if (c.valid) {
        FgVec<VEC> x = fg_load<VEC>(q.x + vb + c.idx);
        const FgVec<VEC> p = fg_load<VEC>((q.mp ? q.mp : q.p) + vb + c.idx);
        FgVec<VEC> r = fg_load<VEC>(q.r + vb + c.idx);
        if (half) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) x.v[e] += alpha * p.v[e];
            fg_store<VEC>(q.x + vb + c.idx, x);
        } else {
            const FgVec<VEC> t = fg_load<VEC>(q.t + vb + c.idx);
            const FgVec<VEC> rw = fg_load<VEC>(q.rw + vb + c.idx);
            const FgVec<VEC> sd = q.ms ? fg_load<VEC>(q.ms + vb + c.idx) : r;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                x.v[e] += alpha * p.v[e] + omega * sd.v[e];
                r.v[e] -= omega * t.v[e];
                part[0] += r.v[e] * r.v[e];
                part[1] += rw.v[e] * r.v[e];
            }
            fg_store<VEC>(q.x + vb + c.idx, x);
            fg_store<VEC>(q.r + vb + c.idx, r);
        }
    }
    if (half) return;
    fg_block_sum<2>(part, lds);
    if (threadIdx.x == 0) {
        acc_add(a + A_RR, (double)part[0]);
        acc_add(a + A_RHO + ((it + 1) & 1), (double)part[1]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Two-kernel form of the same recurrence (default; FG_BICG_FUSED=0 at fg_create keeps the five kernels above).
//
// The five-kernel iteration has five global reductions and moves 96 B per cell and system; 87 launches of ~9 us made up a PISO
// step of the headline workload.  Iteration i of BiCGStab,
//     v_i = C p_i,  alpha_i = rho_i / rw.v_i,  s_i = r_i - alpha_i v_i,  t_i = C s_i,  omega_i = t_i.s_i / t_i.t_i,
//     x_{i+1} = x_i + alpha_i p_i + omega_i s_i,  r_{i+1} = s_i - omega_i t_i,  rho_{i+1} = rw.r_{i+1},
//     p_{i+1} = r_{i+1} + (rho_{i+1} / rho_i)(alpha_i / omega_i)(p_i - omega_i v_i),
// needs only two of them once rho_{i+1} = rw.s_i - omega_i rw.t_i is taken from dot products of the t kernel:
//   k_bicgf_b(i):  [r_i converged?]  s_i, t_i = C s_i   with s at the neighbours recomputed from r, v;   s.s, t.s, t.t, rw.s, rw.t
//   k_bicgf_a(i+1): [s_i converged? -> x += alpha p]  x_{i+1}, r_{i+1}, p_{i+1}, v_{i+1} = C p_{i+1}  with p_{i+1} at the
//                  neighbours recomputed from s, t, p, v;                                                      rw.v, r.r
// 80 B per cell and system, two launches.  s, t, r and the p / v pairs live in separate buffers (the neighbours' old values must
// survive a launch).  Same criterion (RMS residual), same iteration count bookkeeping, same breakdown guards as above: a
// non-finite beta restarts the recurrence (rw = p = r, rho = r.r -- r.r completes in the launch that takes the decision, so the
// decision is parked as a NaN in the rho slot and k_bicgf_b substitutes r.r), rw.v = 0 or t.t = 0 give alpha / omega = 0.
// Accumulators (FgDacc, indexed with the parity e of the iteration that fills them so that a leader can reset the set nobody
// reads in its launch): F_RV + e, F_RR + e by k_bicgf_a | F_SS.. F_RT + e by k_bicgf_b | F_RHOE + e the rho iteration e uses.
// ---------------------------------------------------------------------------------------------------------------------------
template <int DIMS, int VEC>
__device__ __forceinline__ FgVec<VEC> fg_apply_nbr(const FgStencilRow<DIMS, VEC>& m, const FgNbr<DIMS, VEC>& X) {
    FgVec<VEC> y;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        fg_real v = m.d.v[e] * X.c.v[e] + m.o[0].v[e] * X.xm.v[e] + m.o[1].v[e] * X.xp.v[e] + m.o[2].v[e] * X.ym.v[e] +
                  m.o[3].v[e] * X.yp.v[e];
        if constexpr (DIMS == 3) v += m.o[4].v[e] * X.zm.v[e] + m.o[5].v[e] * X.zp.v[e];
        y.v[e] = v;
    }
    return y;
}
// out = a + ca * b at the cell and all its neighbours
template <int DIMS, int VEC>
__device__ __forceinline__ void fg_nbr_axpy(FgNbr<DIMS, VEC>& a, fg_real cb, const FgNbr<DIMS, VEC>& b) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        a.c.v[e] += cb * b.c.v[e]; a.xm.v[e] += cb * b.xm.v[e]; a.xp.v[e] += cb * b.xp.v[e];
        a.ym.v[e] += cb * b.ym.v[e]; a.yp.v[e] += cb * b.yp.v[e];
        if constexpr (DIMS == 3) { a.zm.v[e] += cb * b.zm.v[e]; a.zp.v[e] += cb * b.zp.v[e]; }
    }
}
template <int DIMS, int VEC>
__device__ __forceinline__ void fg_nbr_scale_add(FgNbr<DIMS, VEC>& a, fg_real ca, const FgNbr<DIMS, VEC>& b) {  // a = b + ca * a
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        a.c.v[e] = b.c.v[e] + ca * a.c.v[e]; a.xm.v[e] = b.xm.v[e] + ca * a.xm.v[e]; a.xp.v[e] = b.xp.v[e] + ca * a.xp.v[e];
        a.ym.v[e] = b.ym.v[e] + ca * a.ym.v[e]; a.yp.v[e] = b.yp.v[e] + ca * a.yp.v[e];
        if constexpr (DIMS == 3) { a.zm.v[e] = b.zm.v[e] + ca * a.zm.v[e]; a.zp.v[e] = b.zp.v[e] + ca * a.zp.v[e]; }
    }
}

// init: r = rhs - C x0 ; rw = r ; p_0 = r ; rr_0 = r.r        (grid.y = 1: loops over the nc systems of the env)
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_bicgf_init(FgGrid g, BicgPtrs q, BicgFused w, int use_x0, int tiles_x, int tiles_y,
                                                          int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    const size_t N = g.n;
    bool any = false;
    for (int comp = 0; comp < q.nc; ++comp) any = any || (flag_ld(q.flags + (c.b * q.nc + comp)) == 0);
    if (!any) return;
    FgStencilRow<DIMS, VEC> m;
    if (use_x0 && c.valid) m = fg_load_row<DIMS, VEC>(q.diag + (size_t)c.b * N, q.off + (size_t)c.b * 2 * DIMS * N, c, N);
    __shared__ fg_real lds[4];
    for (int comp = 0; comp < q.nc; ++comp) {
        const int sys = c.b * q.nc + comp;
        if (flag_ld(q.flags + (sys)) != 0) continue;
        const size_t vb = (size_t)sys * N;
        fg_real part[1] = {0.f};
        if (c.valid) {
            FgVec<VEC> r = fg_load<VEC>(q.rhs + vb + c.idx);
            if (use_x0) {
                const FgVec<VEC> y = fg_apply_row<DIMS, VEC>(m, q.x + vb, c);
#pragma unroll
                for (int e = 0; e < VEC; ++e) r.v[e] -= y.v[e];
            } else {
                FgVec<VEC> z;
#pragma unroll
                for (int e = 0; e < VEC; ++e) z.v[e] = 0.f;
                fg_store<VEC>(q.x + vb + c.idx, z);
            }
            fg_store<VEC>(q.r + vb + c.idx, r);
            fg_store<VEC>(q.rw + vb + c.idx, r);
            fg_store<VEC>(w.p[0] + vb + c.idx, r);
#pragma unroll
            for (int e = 0; e < VEC; ++e) part[0] += r.v[e] * r.v[e];
        }
        fg_block_sum<1>(part, lds);
        if (threadIdx.x == 0) acc_add(q.acc + (size_t)sys * FG_ACC_DOUBLES + F_RR, (double)part[0]);
        __syncthreads();
    }
}

// k_bicgf_a(it): finish iteration it - 1 (x, r, p) and start iteration it (v = C p, rw.v, r.r)      (grid.y = 1)
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) __attribute__((amdgpu_waves_per_eu(DIMS == 2 ? 4 : 3))) void k_bicgf_a(FgGrid g, BicgPtrs q, BicgFused w, int it, int tiles_x, int tiles_y,
                                                       int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    const size_t N = g.n;
    const bool leader = (threadIdx.x == 0) && ((fg_xcd_remap(blockIdx.x, gridDim.x) % tiles) == 0);
    const int e = it & 1, pe = e ^ 1;
    // per-system decisions, taken by every workgroup of the env from the same accumulator words (fg_bicg.h: shared with the
    // z-marching kernels).  mode: 0 skip | 1 full update | 2 converged on s: x += alpha p only | 3 first iteration: v = C p
    const bool fold = w.fold0 != 0;
    const BicgDecA D = fg_bicgf_decide_a(g, q, c.b, it, leader, fold);
    const int (&mode)[3] = D.mode;
    const fg_real (&alpha)[3] = D.alpha; const fg_real (&omega)[3] = D.omega; const fg_real (&beta)[3] = D.beta;
    const bool (&restart)[3] = D.restart;
    const bool any = D.any;
    // p of the previous iteration; with a folded start (zero start vector: r_0 = p_0 = rw = rhs, no init kernel) p_0 IS the right-hand side
    const fg_real* __restrict__ p_prev = (fold && it <= 1) ? q.rhs : (it == 0 ? w.p[0] : w.p[pe]);
    if (!any) return;
    FgStencilRow<DIMS, VEC> m;
    if (c.valid) m = fg_load_row<DIMS, VEC>(q.diag + (size_t)c.b * N, q.off + (size_t)c.b * 2 * DIMS * N, c, N);
    // dot-product partials of all components, reduced across the workgroup ONCE after the loop (a reduction per component put a
    // barrier between the components and kept the loads of one from overlapping the arithmetic of the other)
    __shared__ fg_real lds[24];
    fg_real part[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // [comp][rw.v | r.r]
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
        if (comp >= q.nc || mode[comp] == 0) continue;
        const int sys = c.b * q.nc + comp;
        const size_t vb = (size_t)sys * N;
        if (mode[comp] == 2) {
            if (c.valid) {
                FgVec<VEC> x = fg_load<VEC>(q.x + vb + c.idx);
                const FgVec<VEC> p = fg_load<VEC>(p_prev + vb + c.idx);
#pragma unroll
                for (int k = 0; k < VEC; ++k) x.v[k] += alpha[comp] * p.v[k];
                fg_store<VEC>(q.x + vb + c.idx, x);
            }
            continue;
        }
        if (c.valid) {
            FgNbr<DIMS, VEC> P;   // p of iteration `it` at the cell and its neighbours
            FgVec<VEC> rwv;
            if (!(mode[comp] == 3 && fold)) rwv = fg_load<VEC>(q.rw + vb + c.idx);
            if (mode[comp] == 3) {
                P = fg_gather<DIMS, VEC>(p_prev + vb, c);
                if (fold) {   // rw = r_0 = rhs, x_0 = 0, r.r (what the init kernel would have done)
                    rwv = P.c;
                    fg_store<VEC>(q.rw + vb + c.idx, P.c);
                    FgVec<VEC> z0;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) { z0.v[k] = 0; part[2 * comp + 1] += P.c.v[k] * P.c.v[k]; }
                    fg_store<VEC>(q.x + vb + c.idx, z0);
                }
            } else {
                const fg_real al = alpha[comp], om = omega[comp], be = beta[comp];
                // r_{it} = s - omega t at the cell and its neighbours
                FgNbr<DIMS, VEC> R = fg_gather<DIMS, VEC>(w.s + vb, c);
                {
                    const FgNbr<DIMS, VEC> T = fg_gather<DIMS, VEC>(q.t + vb, c);
                    FgVec<VEC> x = fg_load<VEC>(q.x + vb + c.idx);
                    const FgVec<VEC> pold = fg_load<VEC>(p_prev + vb + c.idx);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) x.v[k] += al * pold.v[k] + om * R.c.v[k];
                    fg_store<VEC>(q.x + vb + c.idx, x);
                    fg_nbr_axpy<DIMS, VEC>(R, -om, T);
                }
                fg_store<VEC>(q.r + vb + c.idx, R.c);
#pragma unroll
                for (int k = 0; k < VEC; ++k) part[2 * comp + 1] += R.c.v[k] * R.c.v[k];
                if (restart[comp]) {
                    P = R;
                    rwv = R.c;
                    fg_store<VEC>(q.rw + vb + c.idx, rwv);
                } else {
                    // p_{it} = r + beta (p - omega v)
                    P = fg_gather<DIMS, VEC>(p_prev + vb, c);
                    const FgNbr<DIMS, VEC> V = fg_gather<DIMS, VEC>(w.v[pe] + vb, c);
                    fg_nbr_axpy<DIMS, VEC>(P, -om, V);
                    fg_nbr_scale_add<DIMS, VEC>(P, be, R);
                }
                fg_store<VEC>(w.p[e] + vb + c.idx, P.c);
            }
            const FgVec<VEC> y = fg_apply_nbr<DIMS, VEC>(m, P);
            fg_store<VEC>(w.v[e] + vb + c.idx, y);
#pragma unroll
            for (int k = 0; k < VEC; ++k) part[2 * comp] += rwv.v[k] * y.v[k];
        }
    }
    const fg_real tot = fg_block_sum_lanes<6>(part, lds);      // thread t < 6 holds value t: [comp][rw.v | r.r]
    if (threadIdx.x < 6) {
        const int comp = threadIdx.x >> 1, kind = threadIdx.x & 1;
        const int md = comp == 0 ? mode[0] : (comp == 1 ? mode[1] : mode[2]);
        if (comp < q.nc && (md == 1 || (md == 3 && (kind == 0 || fold)))) {
            FgDacc* a = q.acc + (size_t)(c.b * q.nc + comp) * FG_ACC_DOUBLES;
            acc_add(a + ((kind ? F_RR : F_RV) + e), (double)tot);
        }
    }
}

// k_bicgf_b(it): convergence test on r_it, alpha, s = r - alpha v, t = C s, the five dot products      (grid.y = 1)
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_bicgf_b(FgGrid g, BicgPtrs q, BicgFused w, int it, int tiles_x, int tiles_y,
                                                       int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    const size_t N = g.n;
    const bool leader = (threadIdx.x == 0) && ((fg_xcd_remap(blockIdx.x, gridDim.x) % tiles) == 0);
    const int e = it & 1;
    const BicgDecB D = fg_bicgf_decide_b(g, q, c.b, it, leader);     // (fg_bicg.h: shared with the z-marching kernels)
    const bool (&work)[3] = D.work;
    const fg_real (&alpha)[3] = D.alpha;
    const bool any = D.any;
    const fg_real* __restrict__ r_src = (w.fold0 && it == 0) ? q.rhs : q.r;      // folded start: r_0 is the right-hand side
    if (!any) return;
    FgStencilRow<DIMS, VEC> m;
    if (c.valid) m = fg_load_row<DIMS, VEC>(q.diag + (size_t)c.b * N, q.off + (size_t)c.b * 2 * DIMS * N, c, N);
    __shared__ fg_real lds[60];
    fg_real part[15];   // [comp][s.s | t.s | t.t | rw.s | rw.t], one workgroup reduction after the loop
#pragma unroll
    for (int k = 0; k < 15; ++k) part[k] = 0.f;
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
        if (comp >= q.nc || !work[comp]) continue;
        const int sys = c.b * q.nc + comp;
        const size_t vb = (size_t)sys * N;
        if (c.valid) {
            FgNbr<DIMS, VEC> S = fg_gather<DIMS, VEC>(r_src + vb, c);
            {
                const FgNbr<DIMS, VEC> V = fg_gather<DIMS, VEC>(w.v[e] + vb, c);
                fg_nbr_axpy<DIMS, VEC>(S, -alpha[comp], V);
            }
            const FgVec<VEC> t = fg_apply_nbr<DIMS, VEC>(m, S);
            const FgVec<VEC> rwv = fg_load<VEC>(q.rw + vb + c.idx);
            fg_store<VEC>(w.s + vb + c.idx, S.c);
            fg_store<VEC>(q.t + vb + c.idx, t);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                part[5 * comp + 0] += S.c.v[k] * S.c.v[k];
                part[5 * comp + 1] += t.v[k] * S.c.v[k];
                part[5 * comp + 2] += t.v[k] * t.v[k];
                part[5 * comp + 3] += rwv.v[k] * S.c.v[k];
                part[5 * comp + 4] += rwv.v[k] * t.v[k];
            }
        }
    }
    // (-DFG_KNOCK_B=1: no accumulator atomics, =2: no workgroup reduction either -- timing knock-outs, profiles/micro_bicg2d.py)
#if defined(FG_KNOCK_B) && FG_KNOCK_B >= 2
    if (part[0] == 12345.678f) q.sc[0] = part[1] + part[2] + part[3] + part[4] + part[5] + part[6] + part[7] + part[8] + part[9];
    return;
#endif
    const fg_real tot = fg_block_sum_lanes<15>(part, lds);     // thread t < 15 holds value t: [comp][s.s | t.s | t.t | rw.s | rw.t]
#if defined(FG_KNOCK_B) && FG_KNOCK_B == 1
    if (threadIdx.x == 0 && tot == 12345.678f) q.sc[0] = tot;
    return;
#endif
    if (threadIdx.x < 15) {
        const int comp = threadIdx.x / 5, kind = threadIdx.x - 5 * comp;
        const bool wk = comp == 0 ? work[0] : (comp == 1 ? work[1] : work[2]);
        if (comp < q.nc && wk) {
            FgDacc* a = q.acc + (size_t)(c.b * q.nc + comp) * FG_ACC_DOUBLES;
            acc_add(a + (F_SS + 2 * kind + e), (double)tot);      // F_SS, F_TS, F_TT, F_RS, F_RT are two apart
        }
    }
}

// poll after k_bicgf_a(it + 1), i.e. after iteration `it` completed: judges r_{it+1} exactly as k_bicg_check does (the next
// k_bicgf_b would come to the same verdict from the same accumulator) and mirrors info for the host
__global__ void k_bicgf_check(FgDacc* __restrict__ acc, int32_t* __restrict__ flags, fg_solve_info* __restrict__ info,
                              fg_solve_info* __restrict__ mirror, fg_real tol, int it, int n, int nsys, int final_pass, FgPollOut poll,
                              int sys0 = 0) {
    // systems sys0 .. sys0 + nsys - 1 (a sub-batch of envs: the systems behind it have not started and must not be judged)
    __shared__ uint32_t stage[2 * 64];
    const int first = sys0 + blockIdx.x * blockDim.x, s = first + threadIdx.x;
    const bool valid = s < sys0 + nsys;
    if (valid && flag_ld(flags + (s)) == 4) flag_st(flags + (s), 1);
    if (valid && flag_ld(flags + (s)) == 0) {
        const fg_real crit = (fg_real)sqrt(acc_ld(acc + ((size_t)s * FG_ACC_DOUBLES + F_RR + ((it + 1) & 1))) / (double)n);
        info[s].final_residual = crit;
        info[s].used_iterations = it + 1;
        if (!(crit >= tol)) {
            const bool finite = isfinite(crit);
            flag_st(flags + (s), finite ? 1 : 2);
            info[s].converged = finite ? 1 : 0;
            info[s].is_finite = finite ? 1 : 0;
        } else if (final_pass) {
            info[s].converged = 0;
        }
    }
    fg_poll_publish_infos(poll, mirror, info, s, valid, first, min((int)blockDim.x, sys0 + nsys - first), stage);
}

__global__ void k_bicg_begin(const fg_real* __restrict__ dt, FgDacc* __restrict__ acc, fg_real* __restrict__ sc,
                             int32_t* __restrict__ flags, fg_solve_info* __restrict__ info, int nsys, int nc) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    const FgBicgBegin q = {acc, sc, flags, info, nc};
    fg_bicg_begin_sys(q, dt, s);     // (fg_internal.h: shared with k_adv_build, which prepares the solve that follows an assembly)
}

__global__ void k_bicg_check(FgDacc* __restrict__ acc, int32_t* __restrict__ flags, fg_solve_info* __restrict__ info,
                             fg_solve_info* __restrict__ mirror, fg_real tol, int it, int n, int nsys, int final_pass, FgPollOut poll) {
    __shared__ uint32_t stage[2 * 64];
    const int first = blockIdx.x * blockDim.x, s = first + threadIdx.x;
    const bool valid = s < nsys;
    if (valid && flag_ld(flags + (s)) == 4) flag_st(flags + (s), 1);
    if (valid && flag_ld(flags + (s)) == 0) {
        const fg_real crit = (fg_real)sqrt(acc_ld(acc + ((size_t)s * FG_ACC_DOUBLES + A_RR)) / (double)n);
        info[s].final_residual = crit;
        info[s].used_iterations = it + 1;
        if (!(crit >= tol)) {
            const bool finite = isfinite(crit);
            flag_st(flags + (s), finite ? 1 : 2);
            info[s].converged = finite ? 1 : 0;
            info[s].is_finite = finite ? 1 : 0;
        } else if (final_pass) {
            info[s].converged = 0;
        }
    }
    fg_poll_publish_infos(poll, mirror, info, s, valid, first, min((int)blockDim.x, nsys - first), stage);
}

}  // namespace

static int bicgstab_krylov(fg_state* s, const FgBicgArgs& a, fg_solve_info* info_host, hipStream_t st, bool begun);

int fg_bicgstab_solve(fg_state* s, const FgBicgArgs& a, fg_solve_info* info_host, hipStream_t st) {
#if !FG_F64
    // velocity systems of the uniform 2-D grids: Jacobi sweeps first (fg_jacobi.hip); what they do not settle goes to BiCGStab from a
    // cleared start vector, with the solve state prepared afresh
    if (fg_jacobi_ok(s, a)) {
        const int nsys = s->grid.B * a.nc;
        const bool ready = s->bicg_ready_nc == a.nc && s->bicg_ready_dt == a.dt;
        s->bicg_ready_nc = 0; s->cg_ready_ns = 0;
        if (!ready)
            hipLaunchKernelGGL(k_bicg_begin, dim3((nsys + 63) / 64), dim3(64), 0, st, a.dt, s->acc, s->scratch_B + 4 * s->grid.B, s->flags, s->info_dev, nsys, a.nc);
        int outcome = 0;
        if (int rc = fg_jacobi_solve(s, a, info_host, st, &outcome)) return rc;
        if (outcome == 5) {      // (kernels launched behind the check on speculation overwrote the solve's state: the solve again, without them)
            s->jac_spec_missed += 1;
            int (*const fn)(void*) = s->jac_spec_fn;
            s->jac_spec_fn = nullptr;
            hipLaunchKernelGGL(k_bicg_begin, dim3((nsys + 63) / 64), dim3(64), 0, st, a.dt, s->acc, s->scratch_B + 4 * s->grid.B, s->flags, s->info_dev, nsys, a.nc);
            outcome = 0;
            const int rc = fg_jacobi_solve(s, a, info_host, st, &outcome);
            s->jac_spec_fn = fn;
            if (rc) return rc;
        }
        if (outcome == 1) { s->jac_solves += 1; return FG_OK; }
        if (outcome == 0) return bicgstab_krylov(s, a, info_host, st, true);   // (not tried: the prepared state is untouched)
        s->jac_fallbacks += 1;
        FgBicgArgs a2 = a;
        a2.use_x0 = 0;
        return bicgstab_krylov(s, a2, info_host, st, false);
    }
    // the Helmholtz-preconditioned family (wall-refined 2-D grids: RBC): line sweeps first (fg_linepre.hip); what they do not settle
    // goes to the preconditioned BiCGStab from a cleared start vector, with the solve state prepared afresh
    if (fg_linesweep_ok(s, a)) {
        const int nsys = s->grid.B * a.nc;
        const bool ready = s->bicg_ready_nc == a.nc && s->bicg_ready_dt == a.dt;
        s->bicg_ready_nc = 0; s->cg_ready_ns = 0;
        if (!ready)
            hipLaunchKernelGGL(k_bicg_begin, dim3((nsys + 63) / 64), dim3(64), 0, st, a.dt, s->acc, s->scratch_B + 4 * s->grid.B, s->flags, s->info_dev, nsys, a.nc);
        int outcome = 0;
        if (int rc = fg_linesweep_solve(s, a, info_host, st, &outcome)) return rc;
        if (outcome == 1) { s->jac_solves += 1; return FG_OK; }
        if (outcome == 0) return bicgstab_krylov(s, a, info_host, st, true);
        s->jac_fallbacks += 1;
        FgBicgArgs a2 = a;
        a2.use_x0 = 0;
        return bicgstab_krylov(s, a2, info_host, st, false);
    }
#endif
    return bicgstab_krylov(s, a, info_host, st, false);
}

static int bicgstab_krylov(fg_state* s, const FgBicgArgs& a, fg_solve_info* info_host, hipStream_t st, bool begun) {
    const int B = s->grid.B, n = s->grid.n, nsys = B * a.nc;
    BicgPtrs q;
    q.diag = a.diag; q.off = a.off; q.rhs = a.rhs; q.x = a.x;
    q.r = s->w[0]; q.rw = s->w[1]; q.p = s->w[2]; q.v = s->w[3]; q.t = s->w[4];
    q.acc = s->acc; q.sc = s->scratch_B + 4 * B;  // scratch_B holds 4*B floats of env scalars first
    q.flags = s->flags; q.info = s->info_dev; q.nc = a.nc; q.tol = a.tol;
    q.mp = nullptr; q.ms = nullptr;
    if (a.precond) {
        if (int rc = (a.precond == 3 ? fg_ilu_alloc(s) : fg_line_alloc(s))) return rc;
        q.mp = s->w[5]; q.ms = s->w[6];   // free during a BiCGStab solve (the CG's z and second p buffer)
    }
    const dim3 sg((nsys + 63) / 64), sb(64);
    {   // state already prepared by the k_adv_build that assembled this system (FgBicgBegin, fg_internal.h)?
        const bool ready = begun || (s->bicg_ready_nc == a.nc && s->bicg_ready_dt == a.dt);
        s->bicg_ready_nc = 0; s->cg_ready_ns = 0;
        if (!ready) hipLaunchKernelGGL(k_bicg_begin, sg, sb, 0, st, a.dt, q.acc, q.sc, q.flags, q.info, nsys, a.nc);
    }
    if (a.precond == 2) {
        FG_REQUIRE(s->fd_lam != nullptr, FG_ERR_INVALID_ARG, "Helmholtz preconditioner requested but fg_set_fd_helmholtz was not called");
        if (int rc = fg_helm_factor(s, a.dt, a.nu, a.wall_lo, a.wall_hi, a.nc, st, a.kind)) return rc;
    } else if (a.precond == 3) {
        if (int rc = fg_ilu_factor(s, a.diag, a.off, st)) return rc;
    } else if (a.precond) {
        if (int rc = fg_line_factor(s, a.diag, a.off, a.nc, st)) return rc;
    }
    auto precondition = [&](const fg_real* in, fg_real* out) -> int {
        if (a.precond == 3) return fg_ilu_apply(s, a.diag, a.off, a.nc, in, out, st);
        return a.precond == 2 ? fg_fd_helmholtz_apply(s, a.nc, in, out, st) : fg_line_apply(s, a.diag, a.off, a.nc, in, out, st);
    };

    FgGrid gsub = s->grid;     // the envs a launch covers: all of them, or a sub-batch (b0, B) in the two-kernel 2-D form below
#define FG_BICG_LAUNCH_Y(NY, SLOT, KERNEL, ...)                                                                          \
    do {                                                                                                     \
        if (s->grid.dims == 2) {                                                                             \
            if (s->vec == 4) {                                                                               \
                FgLaunch L = fg_launch_geometry<2, 4>(gsub); L.grid.y = (NY);                             \
                FG_LAUNCH_P(s, SLOT, (KERNEL<2, 4>), L.grid, dim3(FG_BLOCK), 0, st, gsub, q, __VA_ARGS__, L.tiles_x, L.tiles_y, L.tiles); \
            } else {                                                                                         \
                FgLaunch L = fg_launch_geometry<2, 1>(gsub); L.grid.y = (NY);                             \
                FG_LAUNCH_P(s, SLOT, (KERNEL<2, 1>), L.grid, dim3(FG_BLOCK), 0, st, gsub, q, __VA_ARGS__, L.tiles_x, L.tiles_y, L.tiles); \
            }                                                                                                \
        } else {                                                                                             \
            if (s->vec == 4) {                                                                               \
                FgLaunch L = fg_launch_geometry<3, 4>(gsub); L.grid.y = (NY);                             \
                FG_LAUNCH_P(s, SLOT, (KERNEL<3, 4>), L.grid, dim3(FG_BLOCK), 0, st, gsub, q, __VA_ARGS__, L.tiles_x, L.tiles_y, L.tiles); \
            } else {                                                                                         \
                FgLaunch L = fg_launch_geometry<3, 1>(gsub); L.grid.y = (NY);                             \
                FG_LAUNCH_P(s, SLOT, (KERNEL<3, 1>), L.grid, dim3(FG_BLOCK), 0, st, gsub, q, __VA_ARGS__, L.tiles_x, L.tiles_y, L.tiles); \
            }                                                                                                \
        }                                                                                                    \
    } while (0)

#define FG_BICG_LAUNCH(SLOT, KERNEL, ...) FG_BICG_LAUNCH_Y(a.nc, SLOT, KERNEL, __VA_ARGS__)
    bool done = false, info_fresh = false;
    // first convergence poll where the previous solve finished (kernels of converged systems exit at once,
    // so over-launching costs ~2 us per kernel while every poll costs a stream sync), then every 2 iterations
    int next_poll = s->pred_bicg[a.kind & 3] > 1 ? s->pred_bicg[a.kind & 3] : 1;
    const double cells = (double)n, mat = 4.0 * (1 + 2 * s->grid.dims) / a.nc, fl = 2.0 * (1 + 2 * s->grid.dims);
    // (3-D: the neighbour recomputation of four fields across six faces costs more than the two saved passes -- measured on TCF
    //  128 x 64 x 64: 299 us per iteration against 260 us -- so the five kernels stay there)
    int zc3 = 0;
#if FG_F64
    const bool zmarch3 = false; (void)zc3;
#else
    // 3-D: the two-kernel form as z-marching LDS-ring kernels (fg_bicgstab3d.hip) when the grid fits its tiles and fills the chip
    const bool zmarch3 = s->bicg_fused && !a.precond && s->grid.dims == 3 && fg_bicg3_ok(s, a.nc, &zc3);
#endif
    if (s->bicg_fused && !a.precond && (s->grid.dims == 2 || s->bicg_fused >= 2 || zmarch3)) {
        // two-kernel iteration (k_bicgf_a / k_bicgf_b above): per system and cell, a reads x, p, s, t, v, rw + the matrix and writes
        // x, r, p, v (40 + mat B); b reads r, v, rw + the matrix and writes s, t (20 + mat B)
        BicgFused w;
        w.s = s->w[7]; w.p[0] = s->w[2]; w.p[1] = s->w[5]; w.v[0] = s->w[3]; w.v[1] = s->w[6];
        w.fold0 = a.use_x0 ? 0 : 1;      // zero start vector: r_0 = p_0 = rw = rhs, no init kernel (fg_bicg.h BicgFused::fold0)
#if !FG_F64
        const bool za = zmarch3 && (s->bicg3_mix & 1), zb = zmarch3 && (s->bicg3_mix & 2);   // (FG_BICG3_MIX: the two forms share buffers and accumulators)
        // start vector zero on the z-marching kernels: r_0 = p_0 = rw = rhs, no init kernel (fg_bicg.h BicgFused::fold0; FG_BICG3_MIX & 4 keeps it)
        w.fold0 = (!a.use_x0 && !(s->bicg3_mix & 4) && (za == zb)) ? 1 : 0;     // (brick and z-marching kernels alike; not when the two are mixed)
#endif
        // Sub-batches of envs (2-D brick kernels): when the working set of the solve -- nine vectors per system, the matrix, the
        // right-hand sides -- is far beyond the 256 MB Infinity Cache, the envs are solved in groups whose set fits it (the systems are
        // independent), so that the iterations of a group stream from the cache instead of HBM: `large_env` (512 x 256 x 64, 839 MB)
        // runs its launches at the headline's size and efficiency (k_bicgf_a 0.56 -> ~0.68 of the HBM figure) -- FG_BICG_SUB=0 / N
        int nb_sub = B;
#if !FG_F64
        if (s->grid.dims == 2 && !za && !zb) {
#else
        if (s->grid.dims == 2) {
#endif
            const double per_env = (double)n * sizeof(fg_real) * (a.nc * 9.0 + (1 + 2 * s->grid.dims) + a.nc);
            if (s->bicg_sub > 0) nb_sub = s->bicg_sub < B ? s->bicg_sub : B;
            else if (s->bicg_sub < 0 && per_env * B > 400e6) {
                nb_sub = (int)(220e6 / per_env);
                const int tiles_env = (int)((n / (s->vec == 4 ? 4 : 1) + FG_BLOCK - 1) / FG_BLOCK);
                while (nb_sub < B && (long)nb_sub * tiles_env < 1024) ++nb_sub;     // keep the chip filled
                if (nb_sub < 1) nb_sub = 1;
                if (nb_sub >= B) nb_sub = B;
                else nb_sub = (B + (B + nb_sub - 1) / nb_sub - 1) / ((B + nb_sub - 1) / nb_sub);   // even groups
            }
        }
        const int pred0 = next_poll;
        int started_b0 = -1;      // group whose first kernels were launched ahead, behind the previous group's poll (see below)
        for (int b0 = 0; b0 < B; b0 += nb_sub) {
        gsub.b0 = b0; gsub.B = (b0 + nb_sub <= B) ? nb_sub : B - b0;
        const int sys0 = b0 * a.nc, nsys_sub = gsub.B * a.nc;
        done = false; next_poll = pred0;
        if (started_b0 != b0) {
        if (!w.fold0) FG_BICG_LAUNCH_Y(1, -1, k_bicgf_init, w, a.use_x0);
#if !FG_F64
        if (za) { if (int rc = fg_bicg3_launch_a(s, q, w, 0, zc3, -1, st)) return rc; }
        else
#endif
        FG_BICG_LAUNCH_Y(1, -1, k_bicgf_a, w, 0);
        }
        for (int it = 0; it < a.max_iterations && !done; ++it) {
#if !FG_F64
            if (zb) { if (int rc = fg_bicg3_launch_b(s, q, w, it, zc3, fg_prof_slot(s, FG_PK_BICGF_B, q.flags + sys0, nsys_sub, cells * (20.0 + mat), cells * (fl + 12.0), st), st)) return rc; }
            else
#endif
            FG_BICG_LAUNCH_Y(1, fg_prof_slot(s, FG_PK_BICGF_B, q.flags + sys0, nsys_sub, cells * (20.0 + mat), cells * (fl + 12.0), st), k_bicgf_b, w, it);
#if !FG_F64
            if (za) { if (int rc = fg_bicg3_launch_a(s, q, w, it + 1, zc3, fg_prof_slot(s, FG_PK_BICGF_A, q.flags + sys0, nsys_sub, cells * (40.0 + mat), cells * (fl + 14.0), st), st)) return rc; }
            else
#endif
            FG_BICG_LAUNCH_Y(1, fg_prof_slot(s, FG_PK_BICGF_A, q.flags + sys0, nsys_sub, cells * (40.0 + mat), cells * (fl + 14.0), st), k_bicgf_a, w, it + 1);
            if (it + 1 >= next_poll || it + 1 == a.max_iterations) {
                next_poll = it + 1 + 2;
                const int final_pass = (it + 1 == a.max_iterations);
                fg_prof_prefetch(s, st);       // (in front of the polled kernel: its completion then covers the copy)
                const FgPollOut po = fg_poll_next(&s->poll);
                hipLaunchKernelGGL(k_bicgf_check, dim3((nsys_sub + 63) / 64), sb, 0, st, q.acc, q.flags, q.info, s->info_pinned, a.tol, it, n, nsys_sub,
                                   final_pass, po, sys0);
                if (nb_sub < B && b0 + nb_sub < B && started_b0 != b0 + nb_sub) {
                    // the next group's first kernels go out behind this group's check, so that the GPU has work during the host's round
                    // trip (polls are placed where solves usually end); the groups share nothing, whatever this poll says
                    const FgGrid keep = gsub;
                    gsub.b0 = b0 + nb_sub; gsub.B = (gsub.b0 + nb_sub <= B) ? nb_sub : B - gsub.b0;
                    if (!w.fold0) FG_BICG_LAUNCH_Y(1, -1, k_bicgf_init, w, a.use_x0);
                    FG_BICG_LAUNCH_Y(1, -1, k_bicgf_a, w, 0);
                    gsub = keep;
                    started_b0 = b0 + nb_sub;
                }
                if (int rc = fg_poll_wait_infos(&s->poll, po, sys0, nsys_sub, s->info_pinned, st)) return rc;
                info_fresh = true;
                done = true;
                for (int i = sys0; i < sys0 + nsys_sub; ++i) done = done && (s->info_pinned[i].converged || !s->info_pinned[i].is_finite);
            }
        }
        }   // sub-batches
        gsub = s->grid;
    }
#if !FG_F64
    else if (a.precond == 2 && fg_fbicg_ok(s)) {
        // Helmholtz-preconditioned iteration in SIX launches (fg_fftbicg.hip): the vector updates ride in the forward transforms, the
        // matrix is applied by the inverse transforms, decisions and accumulators are those of the two-kernel form (fg_bicg.h)
        BicgFused w;
        w.s = q.r; w.p[0] = q.p; w.p[1] = q.p; w.v[0] = q.v; w.v[1] = q.v;
        w.fold0 = a.use_x0 ? 0 : 1;
        float* t1 = s->w[7];
        if (!w.fold0) FG_BICG_LAUNCH_Y(1, -1, k_bicgf_init, w, a.use_x0);      // r = rhs - C x0, rw = p_0 = r, r.r
        if (int rc = fg_fbicg_forward(s, q, 1, 0, w.fold0, st)) return rc;
        if (int rc = fg_helm_apply(s, a.nc, t1, t1, st)) return rc;
        if (int rc = fg_fbicg_inverse(s, q, 1, 0, st)) return rc;
        for (int it = 0; it < a.max_iterations && !done; ++it) {
            if (int rc = fg_fbicg_forward(s, q, 0, it, w.fold0, st)) return rc;
            if (int rc = fg_helm_apply(s, a.nc, t1, t1, st)) return rc;
            if (int rc = fg_fbicg_inverse(s, q, 0, it, st)) return rc;
            if (int rc = fg_fbicg_forward(s, q, 1, it + 1, w.fold0, st)) return rc;
            if (it + 1 >= next_poll || it + 1 == a.max_iterations) {
                next_poll = it + 1 + 2;
                const int final_pass = (it + 1 == a.max_iterations);
                fg_prof_prefetch(s, st);
                const FgPollOut po = fg_poll_next(&s->poll);
                hipLaunchKernelGGL(k_bicgf_check, sg, sb, 0, st, q.acc, q.flags, q.info, s->info_pinned, a.tol, it, n, nsys, final_pass, po, 0);
                if (int rc = fg_poll_wait_infos(&s->poll, po, 0, nsys, s->info_pinned, st)) return rc;
                info_fresh = true;
                done = true;
                for (int i = 0; i < nsys; ++i) done = done && (s->info_pinned[i].converged || !s->info_pinned[i].is_finite);
                if (done) break;
            }
            if (it + 1 < a.max_iterations) {
                if (int rc = fg_helm_apply(s, a.nc, t1, t1, st)) return rc;
                if (int rc = fg_fbicg_inverse(s, q, 1, it + 1, st)) return rc;
            }
        }
    }
#endif
    else {
    FG_BICG_LAUNCH_Y(1, -1, k_bicg_init, a.use_x0);
    for (int it = 0; it < a.max_iterations && !done; ++it) {
        // algorithmic bytes per system and cell: Kp r,v,p -> p (16; the first iteration only checks) | Kv p,rw -> v + the
        // (1 + 2d) matrix fields shared by the nc right-hand sides | Ks r,v -> s (12) | Kt s -> t + matrix |
        // Kx x,p,s,t,rw -> x,r (28)
        if (it > 0) FG_BICG_LAUNCH(fg_prof_slot(s, FG_PK_BICG_P, q.flags, nsys, cells * 16.0, cells * 4.0, st), k_bicg_p, it);
        else FG_BICG_LAUNCH(-1, k_bicg_p, it);
        if (a.precond)
            if (int rc = precondition(q.p, s->w[5])) return rc;
        FG_BICG_LAUNCH_Y(1, fg_prof_slot(s, FG_PK_BICG_V, q.flags, nsys, cells * (12.0 + mat), cells * (fl + 2.0), st), k_bicg_v, it);
        FG_BICG_LAUNCH(fg_prof_slot(s, FG_PK_BICG_S, q.flags, nsys, cells * 12.0, cells * 4.0, st), k_bicg_s, it);
        if (a.precond)
            if (int rc = precondition(q.r, s->w[6])) return rc;
        FG_BICG_LAUNCH_Y(1, fg_prof_slot(s, FG_PK_BICG_T, q.flags, nsys, cells * (8.0 + mat), cells * (fl + 4.0), st), k_bicg_t, it);
        FG_BICG_LAUNCH(fg_prof_slot(s, FG_PK_BICG_X, q.flags, nsys, cells * 28.0, cells * 10.0, st), k_bicg_x, it);
        if (it + 1 >= next_poll || it + 1 == a.max_iterations) {
            next_poll = it + 1 + 2;
            const int final_pass = (it + 1 == a.max_iterations);
            // one read-back serves the poll and the result (nothing is launched after the last poll)
            fg_prof_prefetch(s, st);
            const FgPollOut po = fg_poll_next(&s->poll);
            hipLaunchKernelGGL(k_bicg_check, sg, sb, 0, st, q.acc, q.flags, q.info, s->info_pinned, a.tol, it, n, nsys, final_pass, po);
            if (int rc = fg_poll_wait_infos(&s->poll, po, 0, nsys, s->info_pinned, st)) return rc;
            info_fresh = true;
            done = true;
            for (int i = 0; i < nsys; ++i) done = done && (s->info_pinned[i].converged || !s->info_pinned[i].is_finite);
        }
    }
    }
#undef FG_BICG_LAUNCH
#undef FG_BICG_LAUNCH_Y
    if (!info_fresh) {
        FG_HIP_CHECK(hipMemcpyAsync(s->info_pinned, s->info_dev, sizeof(fg_solve_info) * nsys, hipMemcpyDeviceToHost, st));
        FG_HIP_CHECK(hipStreamSynchronize(st));
    }
    if (int prc = fg_prof_collect(s, st)) return prc;
    int rc = FG_OK;
    int used_max = 0;
    for (int i = 0; i < nsys; ++i) used_max = s->info_pinned[i].used_iterations > used_max ? s->info_pinned[i].used_iterations : used_max;
    s->pred_bicg[a.kind & 3] = used_max;
    for (int i = 0; i < nsys; ++i) {
        if (info_host) info_host[i] = s->info_pinned[i];
        if (!s->info_pinned[i].is_finite) rc = FG_ERR_NOT_FINITE;
        else if (!s->info_pinned[i].converged && rc == FG_OK) rc = FG_ERR_NOT_CONVERGED;
    }
    FG_HIP_CHECK(hipGetLastError());
    return rc;
}

#if !FG_F64
// ---------------------------------------------------------------------------------------------------------------------------
// fp64 repeat of a failed advection-diffusion solve (fg_rung64.h; the reference's solver_double_fallback, PISOtorch_diff.py:418-445)
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
// y = C x for ONE system in double: the fp32 matrix entries promoted (csrMat.toType(dp)); grid of one env
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k64_adv_apply(FgGrid g, const float* __restrict__ diag, const float* __restrict__ off,
                                                          const double* __restrict__ x, double* __restrict__ y, int tiles_x, int tiles_y, int tiles) {
    const FgCtx<DIMS, 1> c = fg_make_ctx<DIMS, 1>(g, tiles_x, tiles_y, tiles);
    if (!c.valid) return;
    const FgStencilRow<DIMS, 1> m = fg_load_row<DIMS, 1>(diag, off, c, (size_t)g.n);
    const R64Nbr<DIMS> X = r64_gather<DIMS>(x, c);
    double v = (double)m.d.v[0] * X.c + (double)m.o[0].v[0] * X.xm + (double)m.o[1].v[0] * X.xp + (double)m.o[2].v[0] * X.ym +
               (double)m.o[3].v[0] * X.yp;
    if constexpr (DIMS == 3) v += (double)m.o[4].v[0] * X.zm + (double)m.o[5].v[0] * X.zp;
    y[c.idx] = v;
}
}  // namespace

int fg_rung64_bicgstab(fg_state* s, const FgBicgArgs& a, fg_solve_info* info, bool all_systems, hipStream_t st) {
    R64 w;
    if (int rc = w.init(s, st)) return rc;
    const int B = s->grid.B, n = s->grid.n, d = s->grid.dims;
    FgGrid g1 = s->grid; g1.B = 1;
    double *r = w.v[0], *rw = w.v[1], *p = w.v[2], *v = w.v[3], *sv = w.v[4], *t = w.v[5], *x = w.v[6];
    auto apply = [&](int b, const double* in, double* out) {
        const float* diag = a.diag + (size_t)b * n;
        const float* off = a.off + (size_t)b * 2 * d * n;
        if (d == 2) { FgLaunch L = fg_launch_geometry<2, 1>(g1); hipLaunchKernelGGL((k64_adv_apply<2>), L.grid, dim3(FG_BLOCK), 0, st, g1, diag, off, in, out, L.tiles_x, L.tiles_y, L.tiles); }
        else { FgLaunch L = fg_launch_geometry<3, 1>(g1); hipLaunchKernelGGL((k64_adv_apply<3>), L.grid, dim3(FG_BLOCK), 0, st, g1, diag, off, in, out, L.tiles_x, L.tiles_y, L.tiles); }
    };
    auto rms = [&](double q) { return sqrt(q / (double)n); };
    std::vector<float> dt_host;
    if (a.dt) { dt_host.resize(B); FG_HIP_CHECK(hipMemcpyAsync(dt_host.data(), a.dt, sizeof(float) * B, hipMemcpyDeviceToHost, st)); FG_HIP_CHECK(hipStreamSynchronize(st)); }
    int rc_all = FG_OK;
    for (int b = 0; b < B; ++b) {
        if (a.dt && !(dt_host[b] > 0.f)) continue;      // masked env
        bool failed = all_systems;
        for (int comp = 0; comp < a.nc; ++comp) failed = failed || !info[b * a.nc + comp].converged;
        if (!failed) continue;
        for (int comp = 0; comp < a.nc; ++comp) {       // the reference repeats the whole SolveLinear call: every right-hand side of the env
            const int sys = b * a.nc + comp;
            const size_t vb = (size_t)sys * n;
            fg_solve_info& I = info[sys];
            w.load(r, a.rhs + vb);                      // x_0 = 0 ("do not start with a possibly corrupted result tensor"): r_0 = rhs
            w.zero(x);
            w.axpby(rw, 1.0, r, 0.0, r);
            w.axpby(p, 1.0, r, 0.0, r);
            double rho = w.dot(r, r), rr = rho, alpha = 1.0, omega = 1.0;
            I.used_iterations = -1; I.converged = 0; I.is_finite = 1; I.final_residual = (float)rms(rr);
            for (int it = 0; it < a.max_iterations; ++it) {
                if (!(rms(rr) >= (double)a.tol)) { I.converged = std::isfinite(rr) ? 1 : 0; I.is_finite = std::isfinite(rr) ? 1 : 0; break; }
                apply(b, p, v);
                const double rv = w.dot(rw, v);
                alpha = rho / rv;
                if (!std::isfinite(alpha)) alpha = 0.0;
                w.axpby(sv, 1.0, r, -alpha, v);
                const double ss = w.dot(sv, sv);
                if (!(rms(ss) >= (double)a.tol)) {      // converged on s (bicgstab_solver_kernel.cu:305-329)
                    w.axpby(x, 1.0, x, alpha, p);
                    I.used_iterations = it; I.final_residual = (float)rms(ss);
                    I.converged = std::isfinite(ss) ? 1 : 0; I.is_finite = std::isfinite(ss) ? 1 : 0;
                    rr = ss;
                    break;
                }
                apply(b, sv, t);
                const double ts = w.dot(t, sv), tt = w.dot(t, t);
                omega = ts / tt;
                if (!std::isfinite(omega)) omega = 0.0;
                w.axpby(x, 1.0, x, alpha, p);
                w.axpby(x, 1.0, x, omega, sv);
                w.axpby(r, 1.0, sv, -omega, t);
                rr = w.dot(r, r);
                const double rho_new = w.dot(rw, r);
                double beta = (rho_new / rho) * (alpha / omega);
                I.used_iterations = it + 1; I.final_residual = (float)rms(rr);
                if (!std::isfinite(beta)) {             // breakdown: restart the recurrence (rw = p = r, rho = r.r)
                    w.axpby(rw, 1.0, r, 0.0, r); w.axpby(p, 1.0, r, 0.0, r); rho = rr;
                } else {
                    w.axpby(p, 1.0, p, -omega, v);
                    w.axpby(p, beta, p, 1.0, r);
                    rho = rho_new;
                }
                if (it + 1 == a.max_iterations && !(rms(rr) >= (double)a.tol)) { I.converged = 1; }
                if (!std::isfinite(rr)) { I.is_finite = 0; break; }
            }
            if (w.err) return w.err;
            w.store(a.x + vb, x);
            if (!I.is_finite) rc_all = FG_ERR_NOT_FINITE;
            else if (!I.converged && rc_all == FG_OK) rc_all = FG_ERR_NOT_CONVERGED;
        }
    }
    FG_HIP_CHECK(hipStreamSynchronize(st));
    FG_HIP_CHECK(hipGetLastError());
    return rc_all;
}
#endif
