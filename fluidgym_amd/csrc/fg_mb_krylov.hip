// Multi-block path: Krylov solvers on the ELL matrix of the connected-block mesh -- BiCGStab (one-cell, four-cell and fused kernels,
// fp64 iterative refinement, convergence verification), the multilevel preconditioner in kernel form, the chunked CG, ILU(0) --
// and their host drivers (mb_bicgstab, mb_pressure_bicgstab, mb_cg).  Split out of fg_mb_step.hip in round 4.  Replaces
// bicgstabSolveGPU / cgSolveGPU (bicgstab_solver_kernel.cu:63-411, cg_solver_kernel.cu:129-471) on the matrices of
// PISO_multiblock_cuda_kernel.cu:3616-3880, 4812-4978.  gfx950 / wave64 only.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "fg_mb.h"
#include "fg_mb_solve.h"

namespace {

struct MbGraphKey { MbSolve q; int vec4, project_mean; hipStream_t stream; };

template <int DIMS>
__device__ __forceinline__ mb_real mb_spmv(const MbDev& D, const MbSolve& q, int b, const mb_real* __restrict__ x, int i) {
    constexpr int F = 2 * DIMS;
    mb_real y = q.diag[(size_t)b * D.N + i] * x[i];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int n = D.nbr[(size_t)f * D.N + i];
        if (n >= 0) y += q.off[((size_t)b * F + f) * D.N + i] * x[n];
    }
    return y;
}

#define MB_SYS                                          \
    const int i = blockIdx.x * FG_BLOCK + threadIdx.x;  \
    const int sys = q.sys_map ? q.sys_map[blockIdx.y] : (int)blockIdx.y;   /* (compacted launches: MbSolve::sys_map) */ \
    const int b = sys / q.nc;                           \
    const int N = D.N;                                  \
    const bool valid = i < N;                           \
    const bool leader = (blockIdx.x == 0 && threadIdx.x == 0); \
    const size_t vb = (size_t)sys * N;                  \
    FgDacc* a = q.acc + (size_t)sys * MB_ACC;           \
    __shared__ mb_real lds[16];                           \
    (void)b; (void)leader; (void)a; (void)lds; (void)valid;


__global__ void k_mbs_begin(const mb_real* __restrict__ dt, MbSolve q, int nsys) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    for (int k = 0; k < MB_ACC; ++k) acc_st(q.acc + ((size_t)s * MB_ACC + k), 0.0);
    sc_st(q.sc + (s * 2), 1.f); sc_st(q.sc + (s * 2 + 1), 1.f);
    const bool active = mb_active(dt, s / q.nc);
    flag_st(q.flags + (s), active ? 0 : 3);
    q.info[s].final_residual = 0.f;
    q.info[s].used_iterations = -1;
    q.info[s].converged = active ? 0 : 1;
    q.info[s].is_finite = 1;
}

// r = rhs - M x0 (x0 = 0 unless use_x0); rw = p = r; rho0 = rr = r.r
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbs_init(MbDev D, MbSolve q, int use_x0, int sum_slot, int defer_rho) {
    MB_SYS
    if (flag_ld(q.flags + (sys)) != 0) return;
    mb_real r = 0.f;
    if (valid) {
        r = q.rhs[vb + i];
        if (use_x0) r -= mb_spmv<DIMS>(D, q, b, q.x + vb, i);
        else q.x[vb + i] = 0.f;
        q.r[vb + i] = r;
        if (q.rw) q.rw[vb + i] = r;
        q.p[vb + i] = r;
    }
    const mb_real s = mb_block_sum(r * r, lds);
    const mb_real s1 = sum_slot >= 0 ? mb_block_sum(valid ? r * (q.project ? 1.f : D.yproj[i]) : 0.f, lds) : 0.f;
    if (threadIdx.x == 0) {
        if (!defer_rho) { acc_add(a + A_RHO, (double)s); acc_add(a + A_RR, (double)s); }
        if (sum_slot >= 0) acc_add(a + sum_slot, (double)s1);
    }
}

// restart of the BiCGStab recurrence from the current iterate (the reference's residualResetSteps, bicgstab_solver_kernel.cu):
// accumulators and scalars of the systems still iterating are reset, then k_mbs_init recomputes r = b - A x and r^ = p = r.
// In fp32 the recurrence of the nearly singular, non-symmetric pressure systems drifts and finally diverges without it.
__global__ void k_mbb_restart(MbSolve q, int nsys) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys || flag_ld(q.flags + (s)) != 0) return;
    for (int k = 0; k < MB_ACC; ++k) acc_st(q.acc + ((size_t)s * MB_ACC + k), 0.0);
    sc_st(q.sc + (s * 2), 1.f); sc_st(q.sc + (s * 2 + 1), 1.f);
}

// ---- fp64 iterative refinement around the fp32 BiCGStab (pressure_use_bicgstab = 2).  On the nearly singular pressure
// systems of strongly non-orthogonal meshes the solution carries a large near-null component, and fp32 round-off of P x
// (and of x itself) puts the TRUE residual at 1e-5 while the recurrence residual keeps falling.  The reference answers
// that with an fp64 solve (solver_double_fallback, PISOtorch_diff.py:266-371); here the iterate is kept in fp64 (x64), every
// restart folds the fp32 correction into it and recomputes r = b - P x64 in fp64, and the fp32 solver always works on a
// correction that starts from zero -- its round-off scales with the correction, not with the solution.
__global__ __launch_bounds__(FG_BLOCK) void k_mbr_fold(int N, MbSolve q, double* __restrict__ x64, int mode) {
    const int i = blockIdx.x * FG_BLOCK + threadIdx.x, sys = blockIdx.y;
    if (i >= N || flag_ld(q.flags + (sys)) == 3) return;      // inactive envs untouched; converged ones fold their last correction once
    const size_t k = (size_t)sys * N + i;
    if (mode == 0) { x64[k] = 0.0; return; }                        // cold start
    if (mode == 1) { x64[k] = (double)q.x[k]; q.x[k] = 0.f; return; }  // warm start from the caller's x
    if (mode == 2) { if (flag_ld(q.flags + (sys)) == 0) { x64[k] += (double)q.x[k]; q.x[k] = 0.f; } return; }  // restart
    q.x[k] = (mb_real)(x64[k] + (double)q.x[k]);                       // mode 3: hand back the sum
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbr_residual(MbDev D, MbSolve q, const double* __restrict__ x64, int sum_slot, int defer_rho) {
    MB_SYS
    if (flag_ld(q.flags + (sys)) != 0) return;
    mb_real r = 0.f;
    if (valid) {
        constexpr int F = 2 * DIMS;
        const double* x = x64 + vb;
        double y = (double)q.diag[(size_t)b * N + i] * x[i];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int n = D.nbr[(size_t)f * N + i];
            if (n >= 0) y += (double)q.off[((size_t)b * F + f) * N + i] * x[n];
        }
        r = (mb_real)((double)q.rhs[vb + i] - y);
        q.x[vb + i] = 0.f;
        q.r[vb + i] = r; q.rw[vb + i] = r; q.p[vb + i] = r;
    }
    const mb_real s = mb_block_sum(r * r, lds);
    const mb_real s1 = sum_slot >= 0 ? mb_block_sum(valid ? r : 0.f, lds) : 0.f;
    if (threadIdx.x == 0) {
        if (!defer_rho) { acc_add(a + A_RHO, (double)s); acc_add(a + A_RR, (double)s); }
        if (sum_slot >= 0) acc_add(a + sum_slot, (double)s1);
    }
}

// best iterate of the refined BiCGStab: at every refinement point the TRUE residual is known (A_RR after k_mbr_residual /
// k_mbb_project_init), so the fp64 iterate with the smallest one is kept and handed back when a solve ends unconverged --
// BiCGStab's residual is far from monotone (spikes of two orders of magnitude on these systems) and without this an
// unconverged solve returned whatever the last iteration happened to hold.
__global__ void k_mbr_best_decide(MbSolve q, mb_real* __restrict__ best_res, int32_t* __restrict__ keep, int n, int nsys, int first) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    keep[s] = 0;
    if (flag_ld(q.flags + (s)) != 0) return;
    const mb_real crit = (mb_real)sqrt(acc_ld(q.acc + ((size_t)s * MB_ACC + A_RR)) / (double)n);
    if (first) best_res[s] = 3.0e38f;
    if (isfinite(crit) && crit < best_res[s]) { best_res[s] = crit; keep[s] = 1; }
}
__global__ __launch_bounds__(FG_BLOCK) void k_mbr_best_copy(int N, const int32_t* __restrict__ keep, const double* __restrict__ src,
                                                             double* __restrict__ dst) {
    const int i = blockIdx.x * FG_BLOCK + threadIdx.x, sys = blockIdx.y;
    if (i >= N || !keep[sys]) return;
    dst[(size_t)sys * N + i] = src[(size_t)sys * N + i];
}
// systems that ended without converging (or non-finite): x64 <- kept iterate, fp32 correction dropped, residual reported
__global__ __launch_bounds__(FG_BLOCK) void k_mbr_best_restore(int N, MbSolve q, double* __restrict__ x64, const double* __restrict__ best,
                                                                const mb_real* __restrict__ best_res) {
    const int i = blockIdx.x * FG_BLOCK + threadIdx.x, sys = blockIdx.y;
    if (flag_ld(q.flags + (sys)) == 3 || flag_ld(q.flags + (sys)) == 0 || (q.info[sys].converged && q.info[sys].is_finite)) return;
    if (!(best_res[sys] < 3.0e38f)) return;   // nothing kept (non-finite from the start): leave it to the caller's fallback
    if (i < N) { x64[(size_t)sys * N + i] = best[(size_t)sys * N + i]; q.x[(size_t)sys * N + i] = 0.f; }
    if (i == 0) { q.info[sys].final_residual = best_res[sys]; q.info[sys].is_finite = 1; }
}

// second half of the start of a projected BiCGStab solve: r <- r - mean r, rw = p = r, rho0 = rr = |r|^2
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_project_init(int N, MbSolve q) {
    const int i = blockIdx.x * FG_BLOCK + threadIdx.x, sys = blockIdx.y;
    if (flag_ld(q.flags + (sys)) != 0) return;
    FgDacc* a = q.acc + (size_t)sys * MB_ACC;
    const size_t vb = (size_t)sys * N;
    __shared__ mb_real lds[4];
    const mb_real m = (mb_real)(acc_ld(a + (A_ST)) / (double)N);  // k_mbs_init left sum r in A_ST
    mb_real r = 0.f;
    if (i < N) {
        r = q.r[vb + i] - m;
        q.r[vb + i] = r; q.rw[vb + i] = r; q.p[vb + i] = r;
    }
    const mb_real s = mb_block_sum(r * r, lds);
    if (threadIdx.x == 0) { acc_add(a + A_RHO, (double)s); acc_add(a + A_RR, (double)s); }
}

// ---- BiCGStab (same five-kernel recurrence as fg_bicgstab.hip)
// beta of iteration `it` and the breakdown guard.  In fp32 the last iterations of a solve whose tolerance sits at the rounding
// level of its right-hand side run on sums (rw.r, rw.v) that cancel to their rounding lattice -- which contains 0: an exact
// rho = 0 or rw.v = 0 turned alpha / beta into inf or NaN about once in 10^5-10^6 system solves (deterministic for
// given inputs, profiles/bicg_history.py).  rw.v == 0 makes k_mbb_s take alpha = 0 (the iteration degenerates to its minimal-
// residual half); a non-finite beta (rho of the previous iteration 0, omega 0) restarts the recurrence HERE from the current
// residual: rw = p = r, rho = r.r.  Every workgroup of a system decides from the same words, so the decision is uniform.
#define MB_BETA                                                                                                                   \
    const mb_real alpha = sc_ld(q.sc + (sys * 2)), omega = sc_ld(q.sc + (sys * 2 + 1));                                             \
    const double rho_now = acc_ld(a + (A_RHO + (it & 1)));                                                                         \
    const mb_real beta = it == 0 ? 0.f : (mb_real)(rho_now / acc_ld(a + (A_RHOE + ((it + 1) & 1)))) * (alpha / omega);                \
    const bool restart = it > 0 && !isfinite(beta);                                                                                \
    if (leader) acc_st(a + (A_RHOE + (it & 1)), restart ? acc_ld(a + (A_RR)) : rho_now);
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_p(MbDev D, MbSolve q, int it) {
    MB_SYS
    const int f = flag_ld(q.flags + (sys));
    if (f == 4) { if (leader) flag_st(q.flags + (sys), 1); return; }
    if (f != 0) return;
    const mb_real crit = mb_rms(acc_ld(a + (A_RR)), N);
    if (!(crit >= q.tol)) { if (leader) mb_mark(q, sys, crit, (it == 0 && q.it_base == 0) ? -1 : it + q.it_base); return; }
    if (leader) {
        acc_st(a + (A_SS), 0.0); acc_st(a + (A_TS), 0.0); acc_st(a + (A_TT), 0.0); acc_st(a + (A_ST), 0.0);
        q.info[sys].final_residual = crit;
        q.info[sys].used_iterations = it + q.it_base - 1;
    }
    MB_BETA
    if (it == 0 || !valid) return;
    const mb_real mv = q.project ? (mb_real)(acc_ld(a + (A_SV + 2 * ((it + 1) & 1))) / (double)N) : 0.f;  // sum v of the previous iteration (slots 7 / 9 alternate)
    if (restart) { const mb_real r = q.r[vb + i]; q.rw[vb + i] = r; q.p[vb + i] = r; }
    else q.p[vb + i] = q.r[vb + i] + beta * (q.p[vb + i] - omega * (q.v[vb + i] - mv));
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_v(MbDev D, MbSolve q, int it) {
    MB_SYS
    if (flag_ld(q.flags + (sys)) != 0) return;
    mb_real part = 0.f, psum = 0.f;
    if (valid) {
        const mb_real y = mb_spmv<DIMS>(D, q, b, (q.mp ? q.mp : q.p) + vb, i);
        q.v[vb + i] = y;
        part = q.rw[vb + i] * y;  // rw is mean-free: rw . (v - mean v) = rw . v
        psum = y;
    }
    part = mb_block_sum(part, lds);
    if (q.project) psum = mb_block_sum(psum, lds);
    { const int sl[2] = {A_RV, A_SV + 2 * (it & 1)}; const mb_real vv[2] = {part, psum}; const bool on[2] = {true, (bool)q.project}; mb_acc_tail<2>(a, sl, vv, on); }
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_s(MbDev D, MbSolve q, int it) {
    MB_SYS
    if (flag_ld(q.flags + (sys)) != 0) return;
    const mb_real alpha_raw = (mb_real)(acc_ld(a + (A_RHOE + (it & 1))) / acc_ld(a + (A_RV)));
    const mb_real alpha = isfinite(alpha_raw) ? alpha_raw : 0.f;   // rw.v == 0: see MB_BETA
    if (leader) { sc_st(q.sc + (sys * 2), alpha); acc_st(a + (A_RHO + ((it + 1) & 1)), 0.0); acc_st(a + (A_RR), 0.0); acc_st(a + (A_SV + 2 * ((it + 1) & 1)), 0.0); }
    const mb_real mv = q.project ? (mb_real)(acc_ld(a + (A_SV + 2 * (it & 1))) / (double)N) : 0.f;
    mb_real part = 0.f;
    if (valid) {
        const mb_real r = q.r[vb + i] - alpha * (q.v[vb + i] - mv);
        q.r[vb + i] = r;
        part = r * r;
    }
    part = mb_block_sum(part, lds);
    if (threadIdx.x == 0) acc_add(a + A_SS, (double)part);
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_t(MbDev D, MbSolve q, int it) {
    MB_SYS
    if (flag_ld(q.flags + (sys)) != 0) return;
    const mb_real crit_s = mb_rms(acc_ld(a + (A_SS)), N);
    if (!(crit_s >= q.tol)) {  // converged on s (bicgstab_solver_kernel.cu:305-329): k_mbb_x applies x += alpha p
        if (leader) mb_mark(q, sys, crit_s, it, 4);
        return;
    }
    mb_real pt = 0.f, ptt = 0.f, pst = 0.f;
    if (valid) {
        const mb_real t = mb_spmv<DIMS>(D, q, b, (q.ms ? q.ms : q.r) + vb, i);
        q.t[vb + i] = t;
        pt = t * q.r[vb + i];  // s is mean-free: (t - mean t) . s = t . s
        ptt = t * t;
        pst = t;
    }
    pt = mb_block_sum(pt, lds);
    ptt = mb_block_sum(ptt, lds);
    if (q.project) pst = mb_block_sum(pst, lds);
    { const int sl[3] = {A_TS, A_TT, A_ST}; const mb_real vv[3] = {pt, ptt, pst}; const bool on[3] = {true, true, (bool)q.project}; mb_acc_tail<3>(a, sl, vv, on); }
}
// "converged on s" (x += alpha p only, bicgstab_solver_kernel.cu:305-329).  With separate s and t kernels the t kernel has decided
// (flag 4).  With the fused kernel the decision is taken HERE from the complete s.s -- every workgroup of the system computes the
// same value; the leader publishes flag 4, which k_mbb_p of the next iteration (or k_mbs_check) turns into "converged".  A
// workgroup that starts after the leader's store reads 4 instead of 0 and decides the same from s.s.  Non-finite s.s: flagged,
// nothing is added to x.
#define MB_HALF                                                                                                           \
    bool half = (f == 4);                                                                                                 \
    if (q.sbuf) {                                                                                                         \
        const mb_real crit_s = mb_rms(acc_ld(a + (A_SS)), N);                                                               \
        half = !(crit_s >= q.tol);                                                                                        \
        if (half) {                                                                                                       \
            const bool fin = isfinite(crit_s);                                                                            \
            if (leader) {                                                                                                 \
                q.info[sys].final_residual = crit_s; q.info[sys].used_iterations = it + q.it_base;                        \
                q.info[sys].converged = fin ? 1 : 0; q.info[sys].is_finite = fin ? 1 : 0;                                 \
                flag_st(q.flags + (sys), fin ? 4 : 2);                                                                    \
            }                                                                                                             \
            if (!fin) return;                                                                                             \
        }                                                                                                                 \
    }
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_x(MbDev D, MbSolve q, int it) {
    MB_SYS
    const int f = flag_ld(q.flags + (sys));
    if (f != 0 && f != 4) return;
    const mb_real alpha = sc_ld(q.sc + (sys * 2));
    MB_HALF
    const double st = q.project ? acc_ld(a + (A_ST)) : 0.0;
    const mb_real mt = (mb_real)(st / (double)N);
    const mb_real omega_raw = half ? 0.f : (mb_real)(acc_ld(a + (A_TS)) / (acc_ld(a + (A_TT)) - st * st / (double)N));
    const mb_real omega = isfinite(omega_raw) ? omega_raw : 0.f;
    if (leader) { sc_st(q.sc + (sys * 2 + 1), omega); acc_st(a + (A_RV), 0.0); }
    mb_real prr = 0.f, prho = 0.f;
    if (valid) {
        const mb_real pd = (q.mp ? q.mp : q.p)[vb + i];
        if (half) {
            q.x[vb + i] += alpha * pd;
        } else {
            const mb_real sv = (q.sbuf ? q.sbuf : q.r)[vb + i];
            q.x[vb + i] += alpha * pd + omega * (q.ms ? q.ms[vb + i] : sv);
            const mb_real r = sv - omega * (q.t[vb + i] - mt);
            q.r[vb + i] = r;
            prr = r * r;
            prho = q.rw[vb + i] * r;
        }
    }
    if (half) return;
    mb_real sums[2] = {prr, prho};
    mb_block_sums<2>(sums, lds);
    { const int sl[2] = {A_RR, A_RHO + ((it + 1) & 1)}; const mb_real vv[2] = {sums[0], sums[1]}; const bool on[2] = {true, true}; mb_acc_tail<2>(a, sl, vv, on); }
}

// ---- the same five kernels with four consecutive cells per thread (N % 4 == 0): 128-bit loads / stores of the cell's own
// data, the -x / +x neighbours of the stencil from the thread's own cells or a lane shuffle (as in k_mbc_ap4).  At
// 16 envs x 46.7 k cells the one-cell kernels took 51 us per iteration, 2-3x what their bytes need.
#define MB_SYS4                                                   \
    const int i = (blockIdx.x * FG_BLOCK + threadIdx.x) * 4;      \
    const int sys = q.sys_map ? q.sys_map[blockIdx.y] : (int)blockIdx.y;   \
    const int b = sys / q.nc;                                     \
    const int N = D.N;                                            \
    const bool valid = i < N;                                     \
    const bool leader = (blockIdx.x == 0 && threadIdx.x == 0);    \
    const size_t vb = (size_t)sys * N;                            \
    FgDacc* a = q.acc + (size_t)sys * MB_ACC;                     \
    __shared__ mb_real lds[16];                                     \
    (void)b; (void)leader; (void)a; (void)lds; (void)valid;
__device__ __forceinline__ float4 ld4(const mb_real* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(mb_real* p, mb_real a, mb_real b, mb_real c, mb_real d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }
// y = A x for the thread's four cells: xi = the vector at the own cells, gather(n) = the vector at any other cell
template <int DIMS, typename G>
__device__ __forceinline__ void mb_spmv4_core(const MbDev& D, const MbSolve& q, int b, int i, const mb_real xi[4], G gather, mb_real y[4]) {
    constexpr int F = 2 * DIMS;
    const int N = D.N;
    const float4 d4 = ld4(q.diag + (size_t)b * N + i);
    y[0] = d4.x * xi[0]; y[1] = d4.y * xi[1]; y[2] = d4.z * xi[2]; y[3] = d4.w * xi[3];
    const int lane = threadIdx.x & 63;
    const mb_real from_prev = __shfl_up(xi[3], 1), from_next = __shfl_down(xi[0], 1);
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int4 n4 = *reinterpret_cast<const int4*>(D.nbr + (size_t)f * N + i);
        const float4 o4 = ld4(q.off + ((size_t)b * F + f) * N + i);
        const int nn[4] = {n4.x, n4.y, n4.z, n4.w};
        const mb_real oo[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = nn[e];
            if (n < 0) continue;  // prescribed face: no matrix entry (as mb_spmv; 0 * x would turn a non-finite x into NaN)
            mb_real xn;
            if (f == 0 && n == i + e - 1 && (e > 0 || lane > 0)) xn = e > 0 ? xi[e > 0 ? e - 1 : 0] : from_prev;
            else if (f == 1 && n == i + e + 1 && (e < 3 || lane < 63)) xn = e < 3 ? xi[e < 3 ? e + 1 : 3] : from_next;
            else xn = gather(n);
            y[e] += oo[e] * xn;
        }
    }
}
template <int DIMS>
__device__ __forceinline__ void mb_spmv4(const MbDev& D, const MbSolve& q, int b, const mb_real* __restrict__ x, int i, mb_real y[4]) {
    const float4 x4 = ld4(x + i);
    const mb_real xi[4] = {x4.x, x4.y, x4.z, x4.w};
    mb_spmv4_core<DIMS>(D, q, b, i, xi, [x](int n) { return x[n]; }, y);
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_p4(MbDev D, MbSolve q, int it) {
    MB_SYS4
    const int f = flag_ld(q.flags + (sys));
    if (f == 4) { if (leader) flag_st(q.flags + (sys), 1); return; }
    if (f != 0) return;
    const mb_real crit = mb_rms(acc_ld(a + (A_RR)), N);
    if (!(crit >= q.tol)) { if (leader) mb_mark(q, sys, crit, (it == 0 && q.it_base == 0) ? -1 : it + q.it_base); return; }
    if (leader) {
        acc_st(a + (A_SS), 0.0); acc_st(a + (A_TS), 0.0); acc_st(a + (A_TT), 0.0); acc_st(a + (A_ST), 0.0);
        q.info[sys].final_residual = crit;
        q.info[sys].used_iterations = it + q.it_base - 1;
    }
    MB_BETA
    if (it == 0 || !valid) return;
    const mb_real mv = q.project ? (mb_real)(acc_ld(a + (A_SV + 2 * ((it + 1) & 1))) / (double)N) : 0.f;
    const float4 r = ld4(q.r + vb + i);
    if (restart) { st4(q.rw + vb + i, r.x, r.y, r.z, r.w); st4(q.p + vb + i, r.x, r.y, r.z, r.w); return; }
    const float4 p = ld4(q.p + vb + i), v = ld4(q.v + vb + i);
    st4(q.p + vb + i, r.x + beta * (p.x - omega * (v.x - mv)), r.y + beta * (p.y - omega * (v.y - mv)),
        r.z + beta * (p.z - omega * (v.z - mv)), r.w + beta * (p.w - omega * (v.w - mv)));
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_v4(MbDev D, MbSolve q, int it) {
    MB_SYS4
    if (flag_ld(q.flags + (sys)) != 0) return;
    mb_real part = 0.f, psum = 0.f;
    if (valid) {
        mb_real y[4];
        mb_spmv4<DIMS>(D, q, b, (q.mp ? q.mp : q.p) + vb, i, y);
        st4(q.v + vb + i, y[0], y[1], y[2], y[3]);
        const float4 w = ld4(q.rw + vb + i);
        part = w.x * y[0] + w.y * y[1] + w.z * y[2] + w.w * y[3];
        psum = y[0] + y[1] + y[2] + y[3];
    }
    part = mb_block_sum(part, lds);
    if (q.project) psum = mb_block_sum(psum, lds);
    { const int sl[2] = {A_RV, A_SV + 2 * (it & 1)}; const mb_real vv[2] = {part, psum}; const bool on[2] = {true, (bool)q.project}; mb_acc_tail<2>(a, sl, vv, on); }
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_s4(MbDev D, MbSolve q, int it) {
    MB_SYS4
    if (flag_ld(q.flags + (sys)) != 0) return;
    const mb_real alpha_raw = (mb_real)(acc_ld(a + (A_RHOE + (it & 1))) / acc_ld(a + (A_RV)));
    const mb_real alpha = isfinite(alpha_raw) ? alpha_raw : 0.f;   // rw.v == 0: see MB_BETA
    if (leader) { sc_st(q.sc + (sys * 2), alpha); acc_st(a + (A_RHO + ((it + 1) & 1)), 0.0); acc_st(a + (A_RR), 0.0); acc_st(a + (A_SV + 2 * ((it + 1) & 1)), 0.0); }
    const mb_real mv = q.project ? (mb_real)(acc_ld(a + (A_SV + 2 * (it & 1))) / (double)N) : 0.f;
    mb_real part = 0.f;
    if (valid) {
        const float4 r = ld4(q.r + vb + i), v = ld4(q.v + vb + i);
        const mb_real s0 = r.x - alpha * (v.x - mv), s1 = r.y - alpha * (v.y - mv), s2 = r.z - alpha * (v.z - mv), s3 = r.w - alpha * (v.w - mv);
        st4(q.r + vb + i, s0, s1, s2, s3);
        part = s0 * s0 + s1 * s1 + s2 * s2 + s3 * s3;
    }
    part = mb_block_sum(part, lds);
    if (threadIdx.x == 0) acc_add(a + A_SS, (double)part);
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_t4(MbDev D, MbSolve q, int it) {
    MB_SYS4
    if (flag_ld(q.flags + (sys)) != 0) return;
    const mb_real crit_s = mb_rms(acc_ld(a + (A_SS)), N);
    if (!(crit_s >= q.tol)) {
        if (leader) mb_mark(q, sys, crit_s, it, 4);
        return;
    }
    mb_real pt = 0.f, ptt = 0.f, pst = 0.f;
    if (valid) {
        mb_real t[4];
        mb_spmv4<DIMS>(D, q, b, (q.ms ? q.ms : q.r) + vb, i, t);
        st4(q.t + vb + i, t[0], t[1], t[2], t[3]);
        const float4 sv = ld4(q.r + vb + i);
        pt = t[0] * sv.x + t[1] * sv.y + t[2] * sv.z + t[3] * sv.w;
        ptt = t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
        pst = t[0] + t[1] + t[2] + t[3];
    }
    pt = mb_block_sum(pt, lds);
    ptt = mb_block_sum(ptt, lds);
    if (q.project) pst = mb_block_sum(pst, lds);
    { const int sl[3] = {A_TS, A_TT, A_ST}; const mb_real vv[3] = {pt, ptt, pst}; const bool on[3] = {true, true, (bool)q.project}; mb_acc_tail<3>(a, sl, vv, on); }
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_x4(MbDev D, MbSolve q, int it) {
    MB_SYS4
    const int f = flag_ld(q.flags + (sys));
    if (f != 0 && f != 4) return;
    const mb_real alpha = sc_ld(q.sc + (sys * 2));
    MB_HALF
    const double st = q.project ? acc_ld(a + (A_ST)) : 0.0;
    const mb_real mt = (mb_real)(st / (double)N);
    const mb_real omega_raw = half ? 0.f : (mb_real)(acc_ld(a + (A_TS)) / (acc_ld(a + (A_TT)) - st * st / (double)N));
    const mb_real omega = isfinite(omega_raw) ? omega_raw : 0.f;
    if (leader) { sc_st(q.sc + (sys * 2 + 1), omega); acc_st(a + (A_RV), 0.0); }
    mb_real prr = 0.f, prho = 0.f;
    if (valid) {
        const float4 x = ld4(q.x + vb + i), p = ld4((q.mp ? q.mp : q.p) + vb + i);
        if (half) {
            st4(q.x + vb + i, x.x + alpha * p.x, x.y + alpha * p.y, x.z + alpha * p.z, x.w + alpha * p.w);
        } else {
            const float4 sv = ld4((q.sbuf ? q.sbuf : q.r) + vb + i), t = ld4(q.t + vb + i), w = ld4(q.rw + vb + i);
            const float4 sd = q.ms ? ld4(q.ms + vb + i) : sv;
            st4(q.x + vb + i, x.x + alpha * p.x + omega * sd.x, x.y + alpha * p.y + omega * sd.y, x.z + alpha * p.z + omega * sd.z,
                x.w + alpha * p.w + omega * sd.w);
            const mb_real r0 = sv.x - omega * (t.x - mt), r1 = sv.y - omega * (t.y - mt), r2 = sv.z - omega * (t.z - mt), r3 = sv.w - omega * (t.w - mt);
            st4(q.r + vb + i, r0, r1, r2, r3);
            prr = r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3;
            prho = w.x * r0 + w.y * r1 + w.z * r2 + w.w * r3;
        }
    }
    if (half) return;
    mb_real sums[2] = {prr, prho};
    mb_block_sums<2>(sums, lds);
    { const int sl[2] = {A_RR, A_RHO + ((it + 1) & 1)}; const mb_real vv[2] = {sums[0], sums[1]}; const bool on[2] = {true, true}; mb_acc_tail<2>(a, sl, vv, on); }
}

// ---- p and v in one launch: p_new = r + beta (p - omega (v - mean v)) for the own cell and, recomputed from r, p, v of the
// previous iteration, for its neighbours; v_new = A p_new.  p and v ping-pong between two buffers each (the neighbours' old
// values must survive the launch); the convergence test on r, the breakdown restart and the leader's bookkeeping are k_mbb_p's.
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_pv(MbDev D, MbSolve q, int it) {
    MB_SYS
    constexpr int F = 2 * DIMS;
    const int f = flag_ld(q.flags + (sys));
    if (f == 4) { if (leader) flag_st(q.flags + (sys), 1); return; }
    if (f != 0) return;
    const mb_real crit = mb_rms(acc_ld(a + (A_RR)), N);
    if (!(crit >= q.tol)) { if (leader) mb_mark(q, sys, crit, (it == 0 && q.it_base == 0) ? -1 : it + q.it_base); return; }
    if (leader) {
        acc_st(a + (A_SS), 0.0); acc_st(a + (A_TS), 0.0); acc_st(a + (A_TT), 0.0); acc_st(a + (A_ST), 0.0);
        q.info[sys].final_residual = crit;
        q.info[sys].used_iterations = it + q.it_base - 1;
    }
    MB_BETA
    const mb_real mv = (q.project && it > 0) ? (mb_real)(acc_ld(a + (A_SV + 2 * ((it + 1) & 1))) / (double)N) : 0.f;
    mb_real part = 0.f, psum = 0.f;
    if (valid) {
        const mb_real* __restrict__ r = q.r + vb;
        const mb_real* __restrict__ pp = q.p_prev + vb;
        const mb_real* __restrict__ vp = q.v_prev + vb;
        // it == 0: p = r was laid down by the initialisation in the CURRENT p buffer
        auto pnew = [&](int c) -> mb_real {
            if (it == 0) return q.p[vb + c];
            if (restart) return r[c];
            return r[c] + beta * (pp[c] - omega * (vp[c] - mv));
        };
        const mb_real pc = pnew(i);
        mb_real y = q.diag[(size_t)b * N + i] * pc;
#pragma unroll
        for (int ff = 0; ff < F; ++ff) {
            const int n = D.nbr[(size_t)ff * N + i];
            if (n >= 0) y += q.off[((size_t)b * F + ff) * N + i] * pnew(n);
        }
        mb_real rwv = q.rw[vb + i];
        if (restart) { rwv = r[i]; q.rw[vb + i] = rwv; }
        if (it > 0) q.p[vb + i] = pc;
        q.v[vb + i] = y;
        part = rwv * y;
        psum = y;
    }
    mb_real sums[2] = {part, psum};
    mb_block_sums<2>(sums, lds);
    { const int sl[2] = {A_RV, A_SV + 2 * (it & 1)}; const mb_real vv[2] = {sums[0], sums[1]}; const bool on[2] = {true, (bool)q.project}; mb_acc_tail<2>(a, sl, vv, on); }
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_pv4(MbDev D, MbSolve q, int it) {
    MB_SYS4
    const int f = flag_ld(q.flags + (sys));
    if (f == 4) { if (leader) flag_st(q.flags + (sys), 1); return; }
    if (f != 0) return;
    const mb_real crit = mb_rms(acc_ld(a + (A_RR)), N);
    if (!(crit >= q.tol)) { if (leader) mb_mark(q, sys, crit, (it == 0 && q.it_base == 0) ? -1 : it + q.it_base); return; }
    if (leader) {
        acc_st(a + (A_SS), 0.0); acc_st(a + (A_TS), 0.0); acc_st(a + (A_TT), 0.0); acc_st(a + (A_ST), 0.0);
        q.info[sys].final_residual = crit;
        q.info[sys].used_iterations = it + q.it_base - 1;
    }
    MB_BETA
    const mb_real mv = (q.project && it > 0) ? (mb_real)(acc_ld(a + (A_SV + 2 * ((it + 1) & 1))) / (double)N) : 0.f;
    mb_real part = 0.f, psum = 0.f;
    if (valid) {
        const mb_real* __restrict__ r = q.r + vb;
        const mb_real* __restrict__ pp = q.p_prev + vb;
        const mb_real* __restrict__ vp = q.v_prev + vb;
        const mb_real* __restrict__ pcur = q.p + vb;
        mb_real pc[4];
        if (it == 0) {
            const float4 p4 = ld4(pcur + i);
            pc[0] = p4.x; pc[1] = p4.y; pc[2] = p4.z; pc[3] = p4.w;
        } else {
            const float4 r4 = ld4(r + i);
            if (restart) { pc[0] = r4.x; pc[1] = r4.y; pc[2] = r4.z; pc[3] = r4.w; }
            else {
                const float4 p4 = ld4(pp + i), v4 = ld4(vp + i);
                pc[0] = r4.x + beta * (p4.x - omega * (v4.x - mv)); pc[1] = r4.y + beta * (p4.y - omega * (v4.y - mv));
                pc[2] = r4.z + beta * (p4.z - omega * (v4.z - mv)); pc[3] = r4.w + beta * (p4.w - omega * (v4.w - mv));
            }
        }
        mb_real y[4];
        mb_spmv4_core<DIMS>(D, q, b, i, pc, [=](int n) -> mb_real {
            if (it == 0) return pcur[n];
            if (restart) return r[n];
            return r[n] + beta * (pp[n] - omega * (vp[n] - mv));
        }, y);
        float4 w = ld4(q.rw + vb + i);
        if (restart) { w = make_float4(pc[0], pc[1], pc[2], pc[3]); st4(q.rw + vb + i, w.x, w.y, w.z, w.w); }
        if (it > 0) st4(q.p + vb + i, pc[0], pc[1], pc[2], pc[3]);
        st4(q.v + vb + i, y[0], y[1], y[2], y[3]);
        part = w.x * y[0] + w.y * y[1] + w.z * y[2] + w.w * y[3];
        psum = y[0] + y[1] + y[2] + y[3];
    }
    mb_real sums[2] = {part, psum};
    mb_block_sums<2>(sums, lds);
    { const int sl[2] = {A_RV, A_SV + 2 * (it & 1)}; const mb_real vv[2] = {sums[0], sums[1]}; const bool on[2] = {true, (bool)q.project}; mb_acc_tail<2>(a, sl, vv, on); }
}

// ---- s and t in one launch (five kernels per iteration -> four): s = r - alpha (v - mean v) for the own cell and, recomputed
// from r and v, for its neighbours; t = A s; s goes to its own buffer (q.sbuf) because the neighbours' r must survive the launch.
// The convergence-on-s test (bicgstab_solver_kernel.cu:305-329) needs the complete s.s and moves into k_mbb_x.  At 16 x 46.7 k
// cells every one of these kernels is launch-bound (5-9 us); not used with the right-preconditioned recurrence (t = A M s).
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_st(MbDev D, MbSolve q, int it) {
    MB_SYS
    constexpr int F = 2 * DIMS;
    if (flag_ld(q.flags + (sys)) != 0) return;
    const mb_real alpha_raw = (mb_real)(acc_ld(a + (A_RHOE + (it & 1))) / acc_ld(a + (A_RV)));
    const mb_real alpha = isfinite(alpha_raw) ? alpha_raw : 0.f;   // rw.v == 0: see MB_BETA
    if (leader) { sc_st(q.sc + (sys * 2), alpha); acc_st(a + (A_RHO + ((it + 1) & 1)), 0.0); acc_st(a + (A_RR), 0.0); acc_st(a + (A_SV + 2 * ((it + 1) & 1)), 0.0); }
    const mb_real mv = q.project ? (mb_real)(acc_ld(a + (A_SV + 2 * (it & 1))) / (double)N) : 0.f;
    mb_real pss = 0.f, pts = 0.f, ptt = 0.f, pst = 0.f;
    if (valid) {
        const mb_real* __restrict__ r = q.r + vb;
        const mb_real* __restrict__ v = q.v + vb;
        const mb_real sv = r[i] - alpha * (v[i] - mv);
        mb_real t = q.diag[(size_t)b * N + i] * sv;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int n = D.nbr[(size_t)f * N + i];
            if (n >= 0) t += q.off[((size_t)b * F + f) * N + i] * (r[n] - alpha * (v[n] - mv));
        }
        q.sbuf[vb + i] = sv;
        q.t[vb + i] = t;
        pss = sv * sv; pts = t * sv; ptt = t * t; pst = t;
    }
    mb_real sums[4] = {pss, pts, ptt, pst};
    mb_block_sums<4>(sums, lds);
    { const int sl[4] = {A_SS, A_TS, A_TT, A_ST}; const mb_real vv[4] = {sums[0], sums[1], sums[2], sums[3]}; const bool on[4] = {true, true, true, (bool)q.project}; mb_acc_tail<4>(a, sl, vv, on); }
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbb_st4(MbDev D, MbSolve q, int it) {
    MB_SYS4
    if (flag_ld(q.flags + (sys)) != 0) return;
    const mb_real alpha_raw = (mb_real)(acc_ld(a + (A_RHOE + (it & 1))) / acc_ld(a + (A_RV)));
    const mb_real alpha = isfinite(alpha_raw) ? alpha_raw : 0.f;
    if (leader) { sc_st(q.sc + (sys * 2), alpha); acc_st(a + (A_RHO + ((it + 1) & 1)), 0.0); acc_st(a + (A_RR), 0.0); acc_st(a + (A_SV + 2 * ((it + 1) & 1)), 0.0); }
    const mb_real mv = q.project ? (mb_real)(acc_ld(a + (A_SV + 2 * (it & 1))) / (double)N) : 0.f;
    mb_real pss = 0.f, pts = 0.f, ptt = 0.f, pst = 0.f;
    if (valid) {
        const mb_real* __restrict__ r = q.r + vb;
        const mb_real* __restrict__ v = q.v + vb;
        const float4 r4 = ld4(r + i), v4 = ld4(v + i);
        const mb_real sv[4] = {r4.x - alpha * (v4.x - mv), r4.y - alpha * (v4.y - mv), r4.z - alpha * (v4.z - mv), r4.w - alpha * (v4.w - mv)};
        mb_real t[4];
        mb_spmv4_core<DIMS>(D, q, b, i, sv, [r, v, alpha, mv](int n) { return r[n] - alpha * (v[n] - mv); }, t);
        st4(q.sbuf + vb + i, sv[0], sv[1], sv[2], sv[3]);
        st4(q.t + vb + i, t[0], t[1], t[2], t[3]);
        pss = sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2] + sv[3] * sv[3];
        pts = t[0] * sv[0] + t[1] * sv[1] + t[2] * sv[2] + t[3] * sv[3];
        ptt = t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
        pst = t[0] + t[1] + t[2] + t[3];
    }
    mb_real sums[4] = {pss, pts, ptt, pst};
    mb_block_sums<4>(sums, lds);
    { const int sl[4] = {A_SS, A_TS, A_TT, A_ST}; const mb_real vv[4] = {sums[0], sums[1], sums[2], sums[3]}; const bool on[4] = {true, true, true, (bool)q.project}; mb_acc_tail<4>(a, sl, vv, on); }
}

// ---- the additive multilevel preconditioner as kernels (meshes too large for the on-chip CG; right preconditioner of the
// pressure BiCGStab):  z = D^-1 r + 1/2 s^-1 Z4 D4^-1 Z4^T r + s^-1 Z8 A8^+ Z8^T r  with the tables of fg_mb_set_multilevel
// (aggregates = rectangles of cells inside the blocks, geometry-only Galerkin operators) and s = trace(P_env) / trace(S_geom)
// the per-env scale of the pressure matrix against the geometry-only one.  Three launches per application:
//   k_ml_restrict  one thread per 4 x 4 aggregate sums its rectangle of r                                   -> r4 [sys][n4]
//   k_ml_coarse    r8 = sums over the (at most four) children; z8 = A8^+ r8 / s, A8^+ symmetric so the matrix is read by
//                  columns; 16 rows x 64 column groups x 4 systems per workgroup                             -> z8 [sys][n8]
//   k_ml_prolong   z = r / diag + (1/2s) r4 / d4 + z8 at the cell's aggregates                              -> z  [sys][N]
struct MlDev {
    const uint16_t* a4; const uint16_t* parent4; const uint2* rect4; const uint2* child8; const mb_real* rd4; const mb_real* aci8;
    int n4, n8, ld8;
    mb_real* r4; mb_real* z8; const mb_real* scale_inv;   // work arrays [nsys][n4], [nsys][n8]; 1 / s per env
    // r4 a second time in the order the coarse kernel wants it -- [nsys][n8][4], slot pos4[a] = 4 parent + child index -- so that
    // its r8 is one coalesced 16-byte load instead of a child table followed by four gathers
    const uint32_t* pos4; mb_real* r4c;
    const uint16_t* p8c;   // parent4[a4[i]] per cell
};
// 1 / s per env: trace(S_geom) / trace(P_env); one workgroup per env
__global__ __launch_bounds__(1024) void k_ml_scale(const mb_real* __restrict__ diag, int N, mb_real geom_diag_sum, mb_real* __restrict__ scale_inv) {
    const int b = blockIdx.x;
    __shared__ double part[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < N; i += 1024) acc += (double)diag[(size_t)b * N + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += part[w];
        scale_inv[b] = (mb_real)((double)geom_diag_sum / t);
    }
}
// (four threads per aggregate, one per row of its rectangle: a thread's cells are one contiguous run, and the four row sums meet in a
//  fixed order)
__device__ __forceinline__ mb_real ml_quad_sum(mb_real v) {
    const int base = (threadIdx.x & 63) & ~3;
    const mb_real s0 = __shfl(v, base, 64), s1 = __shfl(v, base + 1, 64), s2 = __shfl(v, base + 2, 64), s3 = __shfl(v, base + 3, 64);
    return ((s0 + s1) + s2) + s3;
}
__global__ __launch_bounds__(FG_BLOCK) void k_ml_restrict(MlDev M, const mb_real* __restrict__ in, int N, const int32_t* __restrict__ flags,
                                                          const int32_t* __restrict__ sys_map) {
    const int tq = blockIdx.x * FG_BLOCK + threadIdx.x, a = tq >> 2, row = tq & 3, sys = sys_map ? sys_map[blockIdx.y] : (int)blockIdx.y;
    if (flag_ld(flags + sys) != 0) return;
    mb_real sum = 0.f;
    if (a < M.n4) {
        const uint2 rc = M.rect4[a];
        const int w = rc.y & 255, h = (rc.y >> 8) & 255, stride = rc.y >> 16;
        const mb_real* src = in + (size_t)sys * N + rc.x;
        for (int dy = row; dy < h; dy += 4)
            for (int dx = 0; dx < w; ++dx) sum += src[dy * stride + dx];
    }
    sum = ml_quad_sum(sum);
    if (a < M.n4 && row == 0) { M.r4[(size_t)sys * M.n4 + a] = sum; M.r4c[(size_t)sys * 4 * M.n8 + M.pos4[a]] = sum; }
}
// The restriction fused with the vector update that feeds it (the preconditioned BiCGStab applies M to p and to s right after
// forming them): the thread of an aggregate forms p (k_mbb_p4's update, convergence test and leader bookkeeping) or s (k_mbb_s4's)
// at its own cells, stores it and sums it -- one launch instead of two, twice per iteration.  The cells of an aggregate are a
// partition of the mesh (checked in fg_mb_set_multilevel), so every cell is written exactly once.
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_ml_restrict_p(MbDev D, MbSolve q, MlDev M, int it) {
    const int tq = blockIdx.x * FG_BLOCK + threadIdx.x, ag = tq >> 2, row = tq & 3, sys = q.sys_map ? q.sys_map[blockIdx.y] : (int)blockIdx.y, N = D.N;
    const bool leader = (blockIdx.x == 0 && threadIdx.x == 0);
    const size_t vb = (size_t)sys * N;
    FgDacc* a = q.acc + (size_t)sys * MB_ACC;
    const int f = flag_ld(q.flags + (sys));
    if (f == 4) { if (leader) flag_st(q.flags + (sys), 1); return; }
    if (f != 0) return;
    const mb_real crit = mb_rms(acc_ld(a + (A_RR)), N);
    if (!(crit >= q.tol)) { if (leader) mb_mark(q, sys, crit, (it == 0 && q.it_base == 0) ? -1 : it + q.it_base); return; }
    if (leader) {
        acc_st(a + (A_SS), 0.0); acc_st(a + (A_TS), 0.0); acc_st(a + (A_TT), 0.0); acc_st(a + (A_ST), 0.0);
        q.info[sys].final_residual = crit;
        q.info[sys].used_iterations = it + q.it_base - 1;
    }
    MB_BETA
    const mb_real mv = (q.project && it > 0) ? (mb_real)(acc_ld(a + (A_SV + 2 * ((it + 1) & 1))) / (double)N) : 0.f;
    mb_real sum = 0.f;
    if (ag < M.n4) {
        const uint2 rc = M.rect4[ag];
        const int w = rc.y & 255, h = (rc.y >> 8) & 255, stride = rc.y >> 16;
        for (int dy = row; dy < h; dy += 4)
            for (int dx = 0; dx < w; ++dx) {
                const size_t c = vb + rc.x + dy * stride + dx;
                mb_real pv;
                if (it == 0) pv = q.p[c];                       // p = r was laid down by the initialisation
                else {
                    const mb_real r = q.r[c];
                    if (restart) { q.rw[c] = r; pv = r; }
                    else pv = r + beta * (q.p[c] - omega * (q.v[c] - mv));
                    q.p[c] = pv;
                }
                sum += pv;
            }
    }
    sum = ml_quad_sum(sum);
    if (ag < M.n4 && row == 0) { M.r4[(size_t)sys * M.n4 + ag] = sum; M.r4c[(size_t)sys * 4 * M.n8 + M.pos4[ag]] = sum; }
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_ml_restrict_s(MbDev D, MbSolve q, MlDev M, int it) {
    const int tq = blockIdx.x * FG_BLOCK + threadIdx.x, ag = tq >> 2, row = tq & 3, sys = q.sys_map ? q.sys_map[blockIdx.y] : (int)blockIdx.y, N = D.N;
    const bool leader = (blockIdx.x == 0 && threadIdx.x == 0);
    const size_t vb = (size_t)sys * N;
    FgDacc* a = q.acc + (size_t)sys * MB_ACC;
    __shared__ mb_real lds[16];
    if (flag_ld(q.flags + (sys)) != 0) return;
    const mb_real alpha_raw = (mb_real)(acc_ld(a + (A_RHOE + (it & 1))) / acc_ld(a + (A_RV)));
    const mb_real alpha = isfinite(alpha_raw) ? alpha_raw : 0.f;   // rw.v == 0: see MB_BETA
    if (leader) { sc_st(q.sc + (sys * 2), alpha); acc_st(a + (A_RHO + ((it + 1) & 1)), 0.0); acc_st(a + (A_RR), 0.0); acc_st(a + (A_SV + 2 * ((it + 1) & 1)), 0.0); }
    const mb_real mv = q.project ? (mb_real)(acc_ld(a + (A_SV + 2 * (it & 1))) / (double)N) : 0.f;
    mb_real part = 0.f, sum = 0.f;
    if (ag < M.n4) {
        const uint2 rc = M.rect4[ag];
        const int w = rc.y & 255, h = (rc.y >> 8) & 255, stride = rc.y >> 16;
        for (int dy = row; dy < h; dy += 4)
            for (int dx = 0; dx < w; ++dx) {
                const size_t c = vb + rc.x + dy * stride + dx;
                const mb_real sv = q.r[c] - alpha * (q.v[c] - mv);
                q.r[c] = sv;
                part += sv * sv;
                sum += sv;
            }
    }
    sum = ml_quad_sum(sum);
    if (ag < M.n4 && row == 0) { M.r4[(size_t)sys * M.n4 + ag] = sum; M.r4c[(size_t)sys * 4 * M.n8 + M.pos4[ag]] = sum; }
    part = mb_block_sum(part, lds);
    if (threadIdx.x == 0) acc_add(a + A_SS, (double)part);
}
// One workgroup = 16 rows of A8^+ x SB systems; its 1024 threads are 16 rows x 64 column groups (a thread streams 1 / 64 of its
// row -- by columns, the matrix is symmetric --: 12 loads at Airfoil2D's 771 aggregates, all in flight at once), partial sums meet
// in LDS.  49 x 4 workgroups at 771 aggregates x 16 envs; with 64 rows per workgroup (13 x 4 workgroups, 49 dependent-latency
// loads per thread) the kernel took 12.3 us and was the largest single item of the preconditioned airfoil step.
// SB systems per workgroup (template): the matrix rows a workgroup streams serve SB right-hand sides, so 8 halve the L2 traffic of
// 4 once there are enough systems to fill the chip either way (mb_ml_apply picks).  LDS is sized to the mesh (n8p = n8 rounded up):
// r8 [SB][n8p] (re-used by the second folding stage, which needs 4 SB ROWS floats) and the partial sums [CG][SB][ROWS].
template <int SB>
__global__ __launch_bounds__(ML_ROWS * ML_CG) void k_ml_coarse(MlDev M, int nc, int nsys, const int32_t* __restrict__ flags, int n8p,
                                                               const int32_t* __restrict__ sys_map) {
    // (nsys: systems of the launch -- all of them, or the sys_map entries of a compacted launch)
#define ML_SYSK(k) (sys_map ? sys_map[min(sys0 + (k), nsys - 1)] : sys0 + (k))
    extern __shared__ mb_real l_dyn[];
    mb_real* l_r8 = l_dyn;
    const int r8_words = SB * n8p > 4 * SB * ML_ROWS ? SB * n8p : 4 * SB * ML_ROWS;
    mb_real* l_part = l_dyn + r8_words;                  // [ML_CG][SB][ML_ROWS]
    const int sys0 = blockIdx.y * SB, r = threadIdx.x & (ML_ROWS - 1), cg = threadIdx.x / ML_ROWS;
    bool on[SB], any = false;
#pragma unroll
    for (int k = 0; k < SB; ++k) { on[k] = sys0 + k < nsys && flag_ld(flags + ML_SYSK(k)) == 0; any = any || on[k]; }
    if (!any) return;
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        const float4* r4c = reinterpret_cast<const float4*>(M.r4c + (size_t)(on[k] ? ML_SYSK(k) : 0) * 4 * M.n8);
        for (int g = threadIdx.x; g < M.n8; g += ML_ROWS * ML_CG) {
            mb_real sum = 0.f;
            if (on[k]) {
                const float4 c = r4c[g];               // the (at most four) children in child order, absent ones 0
                sum = ((c.x + c.y) + c.z) + c.w;
            }
            l_r8[k * n8p + g] = sum;
        }
    }
    __syncthreads();
    const int row = blockIdx.x * ML_ROWS + r;
    mb_real acc[SB];
#pragma unroll
    for (int k = 0; k < SB; ++k) acc[k] = 0.f;
    if (row < M.n8) {
        const mb_real* col = M.aci8 + row;
#pragma unroll 4
        for (int j = cg; j < M.n8; j += ML_CG) {      // column groups interleave: a wave's four groups read four adjacent matrix rows
            const mb_real m = col[(size_t)j * M.ld8];
#pragma unroll
            for (int k = 0; k < SB; ++k) acc[k] += m * l_r8[k * n8p + j];
        }
    }
#pragma unroll
    for (int k = 0; k < SB; ++k) l_part[(cg * SB + k) * ML_ROWS + r] = acc[k];
    __syncthreads();
    // 64 partial sums per (system, row), folded in two stages: thread t < 4 SB ROWS = (quarter t / (SB ROWS), system t / ROWS % SB, row t % ROWS)
    if (threadIdx.x < SB * ML_ROWS * 4) {
        const int rr = threadIdx.x & (ML_ROWS - 1), k = (threadIdx.x / ML_ROWS) & (SB - 1), quarter = threadIdx.x / (SB * ML_ROWS);
        mb_real t = 0.f;
#pragma unroll
        for (int w = 0; w < ML_CG / 4; ++w) t += l_part[((quarter * (ML_CG / 4) + w) * SB + k) * ML_ROWS + rr];
        l_r8[threadIdx.x] = t;     // second stage in the (now free) r8 buffer: [quarter][system][row] = thread index
    }
    __syncthreads();
    if (threadIdx.x < SB * ML_ROWS) {
        const int rr = threadIdx.x & (ML_ROWS - 1), k = threadIdx.x / ML_ROWS;
        const int orow = blockIdx.x * ML_ROWS + rr;
        constexpr int Q = SB * ML_ROWS;
        if (orow < M.n8 && sys0 + k < nsys && flag_ld(flags + ML_SYSK(k)) == 0)
            M.z8[(size_t)ML_SYSK(k) * M.n8 + orow] =
                (l_r8[threadIdx.x] + l_r8[Q + threadIdx.x] + l_r8[2 * Q + threadIdx.x] + l_r8[3 * Q + threadIdx.x]) * M.scale_inv[ML_SYSK(k) / nc];
    }
#undef ML_SYSK
}
__global__ __launch_bounds__(FG_BLOCK) void k_ml_prolong(MlDev M, const mb_real* __restrict__ in, const mb_real* __restrict__ diag, int N, int nc,
                                                         const int32_t* __restrict__ flags, mb_real* __restrict__ out, const int32_t* __restrict__ sys_map) {
    const int i = blockIdx.x * FG_BLOCK + threadIdx.x, sys = sys_map ? sys_map[blockIdx.y] : (int)blockIdx.y;
    if (i >= N || flag_ld(flags + sys) != 0) return;
    const int b = sys / nc;
    const unsigned a = M.a4[i];
    const mb_real half_s = 0.5f * M.scale_inv[b];
    out[(size_t)sys * N + i] = in[(size_t)sys * N + i] * __builtin_amdgcn_rcpf(diag[(size_t)b * N + i]) +
                               half_s * M.rd4[a] * M.r4[(size_t)sys * M.n4 + a] + M.z8[(size_t)sys * M.n8 + M.p8c[i]];
}

// ---- CG (cgSolveGPU recurrence, cg_solver_kernel.cu:129-471) in two kernels per iteration.  The search direction is
// never read back through a third pass: k_mbc_ap forms p_it = r_it + beta p_{it-1} for the cell AND for its neighbours
// on the fly (p ping-pongs between two buffers so that the neighbours' old values are still there), which removes one
// launch per iteration from a solve that is launch-bound at these mesh sizes (14 k cells x 64 envs).
// accumulators: rho ring 0..2 (r_k.r_k in slot k % 3) | pAp ping-pong 3,4
// PM: how the residual is projected -- 0 not at all, 1 onto the complement of the constant (yp = 1/sqrt(N): no loads of
// yp at all), 2 onto the complement of a general unit vector yp (gathered with every neighbour)
template <int PM>
__device__ __forceinline__ mb_real mb_yp(const mb_real* __restrict__ yp, int i, mb_real yc) { return PM == 2 ? yp[i] : yc; }
template <int DIMS, int PM>
__global__ __launch_bounds__(FG_BLOCK) void k_mbc_ap(MbDev D, MbSolve q, mb_real* __restrict__ pA, mb_real* __restrict__ pB, int it_arg,
                                                      int project_mean) {
    MB_SYS
    const int it = it_arg >= 0 ? it_arg : q.it_ctr[0];
    if (leader && sys == 0) q.it_ctr[1] = it + 1;
    const mb_real* p_old = (it & 1) ? pA : pB;
    mb_real* p_new = (it & 1) ? pB : pA;
    if (flag_ld(q.flags + (sys)) != 0) return;
    // residual with its mean removed (project_mean): rho = |r|^2 - (sum r)^2 / N
    // residual with its component along the projection vector yp removed (|yp| = 1): rho = |r|^2 - (yp.r)^2
    const double sum_r = project_mean ? acc_ld(a + (C_SUM + it % 3)) : 0.0;
    const mb_real cy = (mb_real)sum_r;
    const double rho = acc_ld(a + (C_RHO + it % 3)) - sum_r * sum_r;
    const mb_real crit = mb_rms(rho, N);
    if (!(crit >= q.tol)) { if (leader) mb_mark(q, sys, crit, it); return; }
    double rho_prev = 1.0;
    if (it > 0) {
        const double sp = project_mean ? acc_ld(a + (C_SUM + (it + 2) % 3)) : 0.0;
        rho_prev = acc_ld(a + (C_RHO + (it + 2) % 3)) - sp * sp;
    }
    const bool fresh = (it == q.it_ctr[2]);  // first iteration after the start or a restart: p = r
    const mb_real beta = fresh ? 0.f : (mb_real)(rho / rho_prev);
    if (leader) {
        q.info[sys].final_residual = crit; q.info[sys].used_iterations = it;
        acc_st(a + (C_RHO + (it + 1) % 3), 0.0);  // accumulated by k_mbc_update of this iteration; nobody reads it here
        acc_st(a + (C_SUM + (it + 1) % 3), 0.0);
        // keep x_it when it beats the kept iterate by 2x: k_mbc_update of this iteration stores it before updating x
        if (q.best_x && (it == 0 || crit < 0.5f * sc_ld(q.sc + (sys * 2)) || (crit < q.accept_factor * q.tol && crit < sc_ld(q.sc + (sys * 2))))) {
            sc_st(q.sc + (sys * 2), crit); q.best_it[sys] = it;
        }
    }
    mb_real part = 0.f;
    if (valid) {
        constexpr int F = 2 * DIMS;
        const mb_real* r = q.r + vb;
        const mb_real* po = p_old + vb;
        const mb_real* yp = D.yproj;
        const mb_real yc = PM == 1 ? cy * mb_rsqrt((mb_real)N) : 0.f;   // cy * yp for the constant vector
        auto proj = [&](int c) { return PM == 0 ? r[c] : (PM == 1 ? r[c] - yc : r[c] - cy * yp[c]); };
        const mb_real pi = fresh ? proj(i) : proj(i) + beta * po[i];
        mb_real y = q.diag[(size_t)b * N + i] * pi;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int n = D.nbr[(size_t)f * N + i];
            if (n >= 0) y += q.off[((size_t)b * F + f) * N + i] * (fresh ? proj(n) : proj(n) + beta * po[n]);
        }
        p_new[vb + i] = pi;
        q.v[vb + i] = y;
        part = pi * y;
    }
    part = mb_block_sum(part, lds);
    if (threadIdx.x == 0) acc_add(a + C_PAP + (it & 1), (double)part);
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbc_update(MbDev D, MbSolve q, const mb_real* __restrict__ pA, const mb_real* __restrict__ pB,
                                                          int it_arg, int project_mean) {
    MB_SYS
    const int it = it_arg >= 0 ? it_arg : q.it_ctr[1] - 1;
    if (leader && sys == 0) q.it_ctr[0] = it + 1;
    const mb_real* p = (it & 1) ? pB : pA;
    if (flag_ld(q.flags + (sys)) != 0) return;
    const double sum_r = project_mean ? acc_ld(a + (C_SUM + it % 3)) : 0.0;
    const mb_real alpha = (mb_real)((acc_ld(a + (C_RHO + it % 3)) - sum_r * sum_r) / acc_ld(a + (C_PAP + (it & 1))));
    if (leader) acc_st(a + (C_PAP + ((it + 1) & 1)), 0.0);  // k_mbc_ap of the next iteration accumulates it; not read here
    mb_real part = 0.f, psum = 0.f;
    if (valid) {
        if (q.best_x && q.best_it[sys] == it) q.best_x[vb + i] = q.x[vb + i];
        q.x[vb + i] += alpha * p[vb + i];
        const mb_real r = q.r[vb + i] - alpha * q.v[vb + i];
        q.r[vb + i] = r;
        part = r * r;
        psum = r * D.yproj[i];
    }
    part = mb_block_sum(part, lds);
    if (project_mean) psum = mb_block_sum(psum, lds);
    { const int sl[2] = {C_RHO + (it + 1) % 3, C_SUM + (it + 1) % 3}; const mb_real vv[2] = {part, psum}; const bool on[2] = {true, (bool)project_mean}; mb_acc_tail<2>(a, sl, vv, on); }
}

// ---- the same two kernels with four consecutive cells per thread (N % 4 == 0): own-cell data moves as 128-bit loads, the
// 2 x 2d x 4 neighbour gathers of a thread are independent and overlap, and a quarter of the workgroups is launched --
// at 14 k cells x 64 envs the scalar kernels were bound by gather latency and workgroup turnover, not by bytes.
template <int DIMS, int PM>
__global__ __launch_bounds__(FG_BLOCK) void k_mbc_ap4(MbDev D, MbSolve q, mb_real* __restrict__ pA, mb_real* __restrict__ pB,
                                                       int it_arg, int project_mean) {
    const int i = (blockIdx.x * FG_BLOCK + threadIdx.x) * 4;
    const int sys = blockIdx.y, b = sys, N = D.N;
    const bool valid = i < N, leader = (blockIdx.x == 0 && threadIdx.x == 0);
    const size_t vb = (size_t)sys * N;
    FgDacc* a = q.acc + (size_t)sys * MB_ACC;
    __shared__ mb_real lds[4];
    const int it = it_arg >= 0 ? it_arg : q.it_ctr[0];
    if (leader && sys == 0) q.it_ctr[1] = it + 1;
    const mb_real* p_old = (it & 1) ? pA : pB;
    mb_real* p_new = (it & 1) ? pB : pA;
    if (flag_ld(q.flags + (sys)) != 0) return;
    // residual with its component along the projection vector yp removed (|yp| = 1): rho = |r|^2 - (yp.r)^2
    const double sum_r = project_mean ? acc_ld(a + (C_SUM + it % 3)) : 0.0;
    const mb_real cy = (mb_real)sum_r;
    const double rho = acc_ld(a + (C_RHO + it % 3)) - sum_r * sum_r;
    const mb_real crit = mb_rms(rho, N);
    if (!(crit >= q.tol)) { if (leader) mb_mark(q, sys, crit, it); return; }
    double rho_prev = 1.0;
    if (it > 0) {
        const double sp = project_mean ? acc_ld(a + (C_SUM + (it + 2) % 3)) : 0.0;
        rho_prev = acc_ld(a + (C_RHO + (it + 2) % 3)) - sp * sp;
    }
    const bool fresh = (it == q.it_ctr[2]);  // first iteration after the start or a restart: p = r
    const mb_real beta = fresh ? 0.f : (mb_real)(rho / rho_prev);
    if (leader) {
        q.info[sys].final_residual = crit; q.info[sys].used_iterations = it;
        acc_st(a + (C_RHO + (it + 1) % 3), 0.0);
        acc_st(a + (C_SUM + (it + 1) % 3), 0.0);
        if (q.best_x && (it == 0 || crit < 0.5f * sc_ld(q.sc + (sys * 2)) || (crit < q.accept_factor * q.tol && crit < sc_ld(q.sc + (sys * 2))))) {
            sc_st(q.sc + (sys * 2), crit); q.best_it[sys] = it;
        }
    }
    mb_real part = 0.f;
    if (valid) {
        constexpr int F = 2 * DIMS;
        const mb_real* r = q.r + vb;
        const mb_real* po = p_old + vb;
        const float4 r4 = *reinterpret_cast<const float4*>(r + i);
        const mb_real* yp = D.yproj;
        const mb_real yc = PM == 1 ? cy * mb_rsqrt((mb_real)N) : 0.f;   // cy * yp for the constant vector
        mb_real pi[4] = {r4.x - yc, r4.y - yc, r4.z - yc, r4.w - yc};
        if (PM == 2) {
            const float4 y4 = *reinterpret_cast<const float4*>(yp + i);
            pi[0] = r4.x - cy * y4.x; pi[1] = r4.y - cy * y4.y; pi[2] = r4.z - cy * y4.z; pi[3] = r4.w - cy * y4.w;
        }
        if (!fresh) {
            const float4 p4 = *reinterpret_cast<const float4*>(po + i);
            pi[0] += beta * p4.x; pi[1] += beta * p4.y; pi[2] += beta * p4.z; pi[3] += beta * p4.w;
        }
        const float4 d4 = *reinterpret_cast<const float4*>(q.diag + (size_t)b * N + i);
        mb_real y[4] = {d4.x * pi[0], d4.y * pi[1], d4.z * pi[2], d4.w * pi[3]};
        // the direction value of a neighbour is the same expression as the cell's own (pi): inside a block row the -x / +x
        // neighbours are the adjacent cells, i.e. this thread's other three cells or the first / last cell of the adjacent
        // lane -- taken from registers / a lane shuffle instead of two gathers each (a third to a half of all gathers)
        const int lane = threadIdx.x & 63;
        const mb_real from_prev = __shfl_up(pi[3], 1), from_next = __shfl_down(pi[0], 1);
        auto gather = [&](int n) {
            mb_real pn = PM == 2 ? r[n] - cy * yp[n] : r[n] - yc;
            if (!fresh) pn += beta * po[n];
            return pn;
        };
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int4 n4 = *reinterpret_cast<const int4*>(D.nbr + (size_t)f * N + i);
            const float4 o4 = *reinterpret_cast<const float4*>(q.off + ((size_t)b * F + f) * N + i);
            const int nn[4] = {n4.x, n4.y, n4.z, n4.w};
            const mb_real oo[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = nn[e] >= 0 ? nn[e] : i;  // prescribed face: coefficient is 0, read something valid
                mb_real pn;
                if (f == 0 && n == i + e - 1 && (e > 0 || lane > 0)) pn = e > 0 ? pi[e > 0 ? e - 1 : 0] : from_prev;
                else if (f == 1 && n == i + e + 1 && (e < 3 || lane < 63)) pn = e < 3 ? pi[e < 3 ? e + 1 : 3] : from_next;
                else pn = gather(n);
                y[e] += oo[e] * pn;
            }
        }
        *reinterpret_cast<float4*>(p_new + vb + i) = make_float4(pi[0], pi[1], pi[2], pi[3]);
        *reinterpret_cast<float4*>(q.v + vb + i) = make_float4(y[0], y[1], y[2], y[3]);
        part = pi[0] * y[0] + pi[1] * y[1] + pi[2] * y[2] + pi[3] * y[3];
    }
    part = mb_block_sum(part, lds);
    if (threadIdx.x == 0) acc_add(a + C_PAP + (it & 1), (double)part);
}
__global__ __launch_bounds__(FG_BLOCK) void k_mbc_update4(int N, MbSolve q, const mb_real* __restrict__ pA, const mb_real* __restrict__ pB,
                                                           int it_arg, int project_mean, const mb_real* __restrict__ yp) {
    const int i = (blockIdx.x * FG_BLOCK + threadIdx.x) * 4;
    const int sys = blockIdx.y;
    const bool valid = i < N, leader = (blockIdx.x == 0 && threadIdx.x == 0);
    const size_t vb = (size_t)sys * N;
    FgDacc* a = q.acc + (size_t)sys * MB_ACC;
    __shared__ mb_real lds[4];
    const int it = it_arg >= 0 ? it_arg : q.it_ctr[1] - 1;
    if (leader && sys == 0) q.it_ctr[0] = it + 1;
    const mb_real* p = (it & 1) ? pB : pA;
    if (flag_ld(q.flags + (sys)) != 0) return;
    const double sum_r = project_mean ? acc_ld(a + (C_SUM + it % 3)) : 0.0;
    const mb_real alpha = (mb_real)((acc_ld(a + (C_RHO + it % 3)) - sum_r * sum_r) / acc_ld(a + (C_PAP + (it & 1))));
    if (leader) acc_st(a + (C_PAP + ((it + 1) & 1)), 0.0);
    mb_real part = 0.f, psum = 0.f;
    if (valid) {
        float4 x4 = *reinterpret_cast<const float4*>(q.x + vb + i);
        if (q.best_x && q.best_it[sys] == it) *reinterpret_cast<float4*>(q.best_x + vb + i) = x4;
        const float4 p4 = *reinterpret_cast<const float4*>(p + vb + i);
        const float4 v4 = *reinterpret_cast<const float4*>(q.v + vb + i);
        float4 r4 = *reinterpret_cast<const float4*>(q.r + vb + i);
        x4.x += alpha * p4.x; x4.y += alpha * p4.y; x4.z += alpha * p4.z; x4.w += alpha * p4.w;
        r4.x -= alpha * v4.x; r4.y -= alpha * v4.y; r4.z -= alpha * v4.z; r4.w -= alpha * v4.w;
        *reinterpret_cast<float4*>(q.x + vb + i) = x4;
        *reinterpret_cast<float4*>(q.r + vb + i) = r4;
        part = r4.x * r4.x + r4.y * r4.y + r4.z * r4.z + r4.w * r4.w;
        const float4 y4 = *reinterpret_cast<const float4*>(yp + i);
        psum = r4.x * y4.x + r4.y * y4.y + r4.z * y4.z + r4.w * y4.w;
    }
    part = mb_block_sum(part, lds);
    if (project_mean) psum = mb_block_sum(psum, lds);
    { const int sl[2] = {C_RHO + (it + 1) % 3, C_SUM + (it + 1) % 3}; const mb_real vv[2] = {part, psum}; const bool on[2] = {true, (bool)project_mean}; mb_acc_tail<2>(a, sl, vv, on); }
}

// restart of the CG recurrence (the reference recomputes r = b - A x and resets p = r every residualResetSteps = 100
// iterations, cg_solver_kernel.cu:281-300): slots of iteration `it` are cleared by k_mbc_clear, then refilled here
__global__ void k_mbc_clear(MbSolve q, int nsys, int it) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0) q.it_ctr[2] = it;
    if (s >= nsys) return;
    acc_st(q.acc + ((size_t)s * MB_ACC + C_RHO + it % 3), 0.0);
    acc_st(q.acc + ((size_t)s * MB_ACC + C_SUM + it % 3), 0.0);
    acc_st(q.acc + ((size_t)s * MB_ACC + C_PAP), 0.0);      // both idle between iterations; a recovered system left NaN here
    acc_st(q.acc + ((size_t)s * MB_ACC + C_PAP + 1), 0.0);
}
// a system whose recurrence broke down (p.Pp <= 0 or overflow on the non-symmetric matrix: flag 2) goes back to its kept
// iterate and rejoins the iteration at the restart that follows
__global__ void k_mbs_recover(int N, MbSolve q) {
    const int sys = blockIdx.y, i = blockIdx.x * FG_BLOCK + threadIdx.x;
    if (flag_ld(q.flags + (sys)) != 2) return;
    if (i < N) {
        const mb_real v = q.best_x[(size_t)sys * N + i];
        q.x[(size_t)sys * N + i] = isfinite(v) ? v : 0.f;
    }
}
__global__ void k_mbs_recover_flags(MbSolve q, int nsys) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys || flag_ld(q.flags + (s)) != 2) return;
    flag_st(q.flags + (s), 0);
    q.info[s].is_finite = 1;
    q.info[s].converged = 0;
}
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbc_restart(MbDev D, MbSolve q, int it, int project_mean) {
    MB_SYS
    if (flag_ld(q.flags + (sys)) != 0) return;
    mb_real r = 0.f;
    if (valid) {
        r = q.rhs[vb + i] - mb_spmv<DIMS>(D, q, b, q.x + vb, i);
        q.r[vb + i] = r;
    }
    const mb_real s2 = mb_block_sum(r * r, lds);
    const mb_real s1 = project_mean ? mb_block_sum(valid ? r * D.yproj[i] : 0.f, lds) : 0.f;
    if (threadIdx.x == 0) {
        acc_add(a + C_RHO + it % 3, (double)s2);
        if (project_mean) acc_add(a + C_SUM + it % 3, (double)s1);
    }
}

__global__ void k_mbs_check(MbSolve q, fg_solve_info* __restrict__ mirror, int32_t* __restrict__ flag_mirror, int rr_slot,
                            int it, int n, int nsys, int final_pass, int sum_slot = -1, FgPollOut poll = FgPollOut{nullptr, 0}) {
    __shared__ uint32_t stage[64 * 3];
    const int first = blockIdx.x * blockDim.x, s = first + threadIdx.x;
    const bool valid = s < nsys;
    if (it < 0) {  // graph-replayed CG: iteration index and accumulator slots from the device counter
        it = q.it_ctr[0] - 1;
        rr_slot = C_RHO + (it + 1) % 3;
        if (sum_slot != -1) sum_slot = C_SUM + (it + 1) % 3;
        final_pass = (it + 1 >= q.max_iterations);
    }
    if (valid && flag_ld(q.flags + (s)) == 4) flag_st(q.flags + (s), 1);
    if (valid && flag_ld(q.flags + (s)) == 0) {
        double rr = acc_ld(q.acc + ((size_t)s * MB_ACC + rr_slot));
        if (sum_slot >= 0) { const double sr = acc_ld(q.acc + ((size_t)s * MB_ACC + sum_slot)); rr -= sr * sr; }
        const mb_real crit = (mb_real)sqrt(rr / (double)n);
        q.info[s].final_residual = crit;
        q.info[s].used_iterations = it + 1;
        if (!(crit >= q.tol)) {
            const bool finite = isfinite(crit);
            flag_st(q.flags + (s), finite ? 1 : 2);
            q.info[s].converged = finite ? 1 : 0;
            q.info[s].is_finite = finite ? 1 : 0;
        } else if (q.best_x && q.accept_factor > 0.f && sc_ld(q.sc + (s * 2)) <= q.accept_factor * q.tol && it - q.best_it[s] >= q.accept_window) {
            // hovering just above the tolerance (the residual of CG is not monotone, least of all on the non-symmetric
            // matrix): take the kept iterate instead of waiting for a lucky dip
            q.info[s].converged = 1;
            flag_st(q.flags + (s), 5);
        } else if (final_pass || (q.best_x && q.stall_limit > 0 && it - q.best_it[s] > q.stall_limit)) {
            // out of iterations, or no iterate has halved the best residual for stall_limit iterations (the reference
            // would run on to max_iterations and then hand back its best iterate, too)
            q.info[s].converged = 0;
            flag_st(q.flags + (s), 1);
        }
    }
#if !FG_MB_F64
    if (poll.gran) {
        // result words (FgPollOut, fg_internal.h), three per system: residual, info word, flag -- the verdicts travel in the words the host
        // spins on, no release and no write-back of this XCD's L2 (mb_poll unpacks them to where the mirrors leave them)
        uint32_t w[3] = {0u, 0u, 0u};
        if (valid) { const fg_solve_info v = q.info[s]; w[0] = __float_as_uint((float)v.final_residual); w[1] = fg_info_word(v); w[2] = (uint32_t)flag_ld(q.flags + (s)); }
        fg_poll_publish_records<3>(poll, first, min((int)blockDim.x, nsys - first), threadIdx.x, w, valid, stage);
        return;
    }
#endif
    if (!valid) return;
    mirror[s] = q.info[s];
    flag_mirror[s] = flag_ld(q.flags + (s));
    fg_poll_publish(poll, s);      // (after both mirrors: the host spins on this word instead of synchronising the stream)
}

// hand back the kept iterate of the systems that ended unconverged
__global__ void k_mbs_restore_best(int N, MbSolve q) {
    const int sys = blockIdx.y, i = blockIdx.x * FG_BLOCK + threadIdx.x;
    if (i >= N || (q.info[sys].converged && flag_ld(q.flags + (sys)) != 5) || flag_ld(q.flags + (sys)) == 3) return;
    q.x[(size_t)sys * N + i] = q.best_x[(size_t)sys * N + i];
    if (i == 0) { q.info[sys].final_residual = sc_ld(q.sc + (sys * 2)); q.info[sys].used_iterations = q.best_it[sys]; }
}

// ---------------------------------------------------------------------------------------------------------------
// ILU(0) of an ELL matrix on the mesh's neighbour table: the reference's preconditioner of the BiCG_precondition_fallback rung
// (cusparseScsrilu02 + two cusparseSpSV, bicgstab_solver_kernel.cu:191-226, 288-293).  General pattern (at an interior vertex shared by
// three cells two lower neighbours of a cell are neighbours of each other and an elimination step does update an off-diagonal),
// IKJ elimination row by row; rows are processed level by level (a row depends on its lower neighbours: the schedule comes from
// the neighbour table, host, once per mesh -- mb_ilu_prepare), one workgroup per env (factor) / per system (solves), a barrier
// between levels.  Sequential by nature -- it runs on the rung that repeats a FAILED solve, not on the step's fast path.
// W[f][i] = l_ik (neighbour across face f below i) or u_ij (above); ud[i] = u_ii.
// ---------------------------------------------------------------------------------------------------------------
template <int DIMS>
__global__ __launch_bounds__(1024) void k_mb_ilu_factor(MbDev D, const mb_real* __restrict__ dt, const mb_real* __restrict__ diag,
                                                         const mb_real* __restrict__ off, const int32_t* __restrict__ order,
                                                         const int32_t* __restrict__ start, int levels, mb_real* __restrict__ W,
                                                         mb_real* __restrict__ ud) {
    constexpr int F = 2 * DIMS;
    const int b = blockIdx.x, N = D.N;
    if (!mb_active(dt, b)) return;
    const mb_real* dg = diag + (size_t)b * N;
    const mb_real* of = off + (size_t)b * F * N;
    mb_real* w_ = W + (size_t)b * F * N;
    mb_real* u_ = ud + (size_t)b * N;
    for (int lv = 0; lv < levels; ++lv) {
        for (int pos = start[lv] + (int)threadIdx.x; pos < start[lv + 1]; pos += (int)blockDim.x) {
            const int i = order[pos];
            int nb[F];
            mb_real w[F];
            mb_real d = dg[i];
#pragma unroll
            for (int f = 0; f < F; ++f) { nb[f] = D.nbr[(size_t)f * N + i]; w[f] = nb[f] >= 0 ? of[(size_t)f * N + i] : 0.f; }
            int last = -1;
            for (int round = 0; round < F; ++round) {          // lower neighbours in increasing order of their index
                int k = 0x7fffffff, fk = -1;
#pragma unroll
                for (int f = 0; f < F; ++f) if (nb[f] >= 0 && nb[f] < i && nb[f] > last && nb[f] < k) { k = nb[f]; fk = f; }
                if (fk < 0) break;
                last = k;
                const mb_real l = w[fk] / u_[k];
                w[fk] = l;
#pragma unroll
                for (int g = 0; g < F; ++g) {                  // row k above its diagonal
                    const int j = D.nbr[(size_t)g * N + k];
                    if (j <= k) continue;
                    const mb_real ukj = w_[(size_t)g * N + k];
                    if (j == i) d -= l * ukj;
                    else {
#pragma unroll
                        for (int f2 = 0; f2 < F; ++f2) if (nb[f2] == j) w[f2] -= l * ukj;
                    }
                }
            }
            u_[i] = d;
#pragma unroll
            for (int f = 0; f < F; ++f) w_[(size_t)f * N + i] = w[f];
        }
        __syncthreads();
    }
}

// out = U^-1 L^-1 in for every system still iterating; grid = (nc, B)
template <int DIMS>
__global__ __launch_bounds__(1024) void k_mb_ilu_solve(MbDev D, int nc, const int32_t* __restrict__ flags, const int32_t* __restrict__ order_f,
                                                        const int32_t* __restrict__ start_f, int levels_f, const int32_t* __restrict__ order_b,
                                                        const int32_t* __restrict__ start_b, int levels_b, const mb_real* __restrict__ W,
                                                        const mb_real* __restrict__ ud, const mb_real* __restrict__ in, mb_real* __restrict__ out) {
    constexpr int F = 2 * DIMS;
    const int b = blockIdx.y, sys = b * nc + (int)blockIdx.x, N = D.N;
    if (flag_ld(flags + sys) != 0) return;
    const mb_real* w_ = W + (size_t)b * F * N;
    const mb_real* u_ = ud + (size_t)b * N;
    const mb_real* r = in + (size_t)sys * N;
    mb_real* y = out + (size_t)sys * N;
    for (int lv = 0; lv < levels_f; ++lv) {
        for (int pos = start_f[lv] + (int)threadIdx.x; pos < start_f[lv + 1]; pos += (int)blockDim.x) {
            const int i = order_f[pos];
            mb_real v = r[i];
#pragma unroll
            for (int f = 0; f < F; ++f) { const int k = D.nbr[(size_t)f * N + i]; if (k >= 0 && k < i) v -= w_[(size_t)f * N + i] * y[k]; }
            y[i] = v;
        }
        __syncthreads();
    }
    for (int lv = 0; lv < levels_b; ++lv) {
        for (int pos = start_b[lv] + (int)threadIdx.x; pos < start_b[lv + 1]; pos += (int)blockDim.x) {
            const int i = order_b[pos];
            mb_real v = y[i];
#pragma unroll
            for (int f = 0; f < F; ++f) { const int j = D.nbr[(size_t)f * N + i]; if (j > i) v -= w_[(size_t)f * N + i] * y[j]; }
            y[i] = v / u_[i];
        }
        __syncthreads();
    }
}

// ---- convergence verification of the preconditioned / refined BiCGStab.  The kernels above declare convergence on the
// RECURRENCE residual; after the residual spikes BiCGStab is known for (worst with a right preconditioner on a matrix it fits
// badly) the true residual b - A x of the fp32 iterate can sit far above it -- measured 1.1e-4 against a recurrence residual below
// 2e-6.  So a system that reports convergence is reopened once, its residual is recomputed from the iterate (in fp64 from the fp64
// iterate when refining) and it only stays converged if THAT meets the tolerance; otherwise it iterates on from the recomputed
// residual.  At most three rounds per solve (an fp32 residual cannot always be pushed below a tolerance at its rounding level).
__global__ void k_mbb_reopen(MbSolve q, int32_t* __restrict__ verified, int nsys) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    if (flag_ld(q.flags + s) == 1 && q.info[s].converged && q.info[s].is_finite && verified[s] == 0) {
        verified[s] = 2;
        for (int k = 0; k < MB_ACC; ++k) acc_st(q.acc + ((size_t)s * MB_ACC + k), 0.0);
        sc_st(q.sc + (s * 2), 1.f); sc_st(q.sc + (s * 2 + 1), 1.f);
        flag_st(q.flags + s, 0);
    }
}
__global__ void k_mbb_verify(MbSolve q, int32_t* __restrict__ verified, int n, int nsys, int last_round) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys || verified[s] != 2) return;
    const mb_real crit = (mb_real)sqrt(acc_ld(q.acc + ((size_t)s * MB_ACC + A_RR)) / (double)n);   // the recomputed residual
    if (crit < q.tol || last_round) {   // last round: ends here either way, reported as what it is
        verified[s] = 1;
        q.info[s].final_residual = crit;
        q.info[s].converged = crit < q.tol ? 1 : 0;
        q.info[s].is_finite = isfinite(crit) ? 1 : 0;
        flag_st(q.flags + s, isfinite(crit) ? 1 : 2);
    } else {
        verified[s] = 0;          // iterates on (flag 0) from the recomputed residual; checked again when it reports convergence
        q.info[s].converged = 0;
        q.info[s].final_residual = crit;
    }
}

int mb_poll(fg_mb_state* s, int nsys, hipStream_t st, bool& done, const FgPollOut& po, bool words) {
    // words: the polled kernel was k_mbs_check, whose verdicts arrive in the result words when the poll has them
    if (words && po.gran) {
        if (int rc = fg_poll_wait_words(&s->poll, po, 0, 3 * nsys, st)) return rc;
        for (int i = 0; i < nsys; ++i) {
            fg_solve_info& I = s->info_pinned[i];
            const uint32_t wd = fg_poll_word(&s->poll, 3 * i + 1);
            I.final_residual = fg_poll_word_float(&s->poll, 3 * i);
            I.used_iterations = (int32_t)(wd >> 2) - 1; I.converged = (wd >> 1) & 1; I.is_finite = wd & 1;
            s->flags_pinned[i] = (int32_t)fg_poll_word(&s->poll, 3 * i + 2);
        }
    } else if (int rc = fg_poll_wait(&s->poll, po, 0, nsys, st)) return rc;
    done = true;
    for (int i = 0; i < nsys; ++i) done = done && s->flags_pinned[i] != 0;
    return FG_OK;
}

}  // namespace

int mb_finish(fg_mb_state* s, int nsys, fg_solve_info* info_host, int* max_it) {
    int rc = FG_OK, m = 0;
    for (int i = 0; i < nsys; ++i) {
        if (info_host) info_host[i] = s->info_pinned[i];
        m = std::max(m, (int)s->info_pinned[i].used_iterations);
        if (!s->info_pinned[i].is_finite) rc = FG_ERR_NOT_FINITE;
        else if (!s->info_pinned[i].converged && rc == FG_OK) rc = FG_ERR_NOT_CONVERGED;
    }
    if (max_it) *max_it = m;
    FG_HIP_CHECK(hipGetLastError());
    return rc;
}

MbSolve mb_solve_ptrs(fg_mb_state* s, const mb_real* diag, const mb_real* off, const mb_real* rhs, mb_real* x, int nc, mb_real tol) {
    MbSolve q;
    memset(&q, 0, sizeof(q));  // the struct doubles as (part of) the key of the cached CG graph: no stray padding bytes
    q.diag = diag; q.off = off; q.rhs = rhs; q.x = x;
    q.r = s->w[0]; q.rw = s->w[1]; q.p = s->w[2]; q.v = s->w[3]; q.t = s->w[4];
    q.acc = s->acc; q.sc = s->sc; q.flags = s->flags; q.info = s->info_dev; q.nc = nc; q.tol = tol;
    q.best_x = nullptr; q.best_it = nullptr; q.stall_limit = 0;
    q.it_ctr = s->it_ctr; q.max_iterations = 0;
    q.accept_factor = 0.f; q.accept_window = 0;
    q.project = 0;
    return q;
}



// z = M in for every system still iterating (kernel form of the multilevel preconditioner; pressure systems, nc == 1)
MlDev mb_ml_dev(const fg_mb_state* s) {
    MlDev M;
    M.a4 = s->ml_a4; M.parent4 = s->ml_parent4; M.rect4 = s->ml_rect4; M.child8 = s->ml_child8; M.rd4 = s->ml_d4g; M.aci8 = s->ml_aci8;
    M.n4 = s->ml_n4; M.n8 = s->ml_n8; M.ld8 = (s->ml_n8 + 3) & ~3;
    M.r4 = s->ml_r4; M.z8 = s->ml_z8; M.scale_inv = s->ml_scale; M.pos4 = s->ml_pos4; M.r4c = s->ml_r4c; M.p8c = s->ml_p8c;
    return M;
}
// fused = 0: z = M in.  1 / 2: `in` is q.p / q.r and its update (k_mbb_p4 / k_mbb_s4) happens inside the restriction
// (k_ml_restrict_p / _s), iteration index `it`.
// the per-env scale of the multilevel preconditioner (sum_i P_ii / geom_diag_sum) for the matrix diagonal `diag`
void mb_ml_scale(fg_mb_state* s, const mb_real* diag, hipStream_t st) {
    hipLaunchKernelGGL(k_ml_scale, dim3(s->B), dim3(1024), 0, st, diag, s->N, s->ml_geom_diag_sum, s->ml_scale);
}

void mb_ml_apply(fg_mb_state* s, const MbSolve& q, const mb_real* in, mb_real* out, hipStream_t st, int fused, int it) {
    const MlDev M = mb_ml_dev(s);
    const int nsys = q.sys_map ? q.n_map : s->B * q.nc, n = s->N;    // (systems of the launch: compacted when only a few still iterate)
    const dim3 rgrid4((4 * M.n4 + FG_BLOCK - 1) / FG_BLOCK, nsys);   // four threads per aggregate
    if (fused == 1) { MB_DISPATCH(s, hipLaunchKernelGGL(k_ml_restrict_p<DIMS>, rgrid4, dim3(FG_BLOCK), 0, st, s->dev, q, M, it);); }
    else if (fused == 2) { MB_DISPATCH(s, hipLaunchKernelGGL(k_ml_restrict_s<DIMS>, rgrid4, dim3(FG_BLOCK), 0, st, s->dev, q, M, it);); }
    else
    hipLaunchKernelGGL(k_ml_restrict, rgrid4, dim3(FG_BLOCK), 0, st, M, in, n, (const int32_t*)q.flags, q.sys_map);
    {
        // systems per workgroup: 8 when that still leaves >= 2 workgroups per CU-pair of work (>= 32 systems) and the LDS fits 64 KB
        const int n8p = (M.n8 + 3) & ~3;
        const auto words = [&](int sb) { return (sb * n8p > 4 * sb * ML_ROWS ? sb * n8p : 4 * sb * ML_ROWS) + ML_CG * sb * ML_ROWS; };
        const int want = s->dbg_ml_sb ? s->dbg_ml_sb : (nsys >= 32 ? 8 : 4);
        if (want == 8 && words(8) * 4 <= 64 * 1024)
            hipLaunchKernelGGL(k_ml_coarse<8>, dim3((M.n8 + ML_ROWS - 1) / ML_ROWS, (nsys + 7) / 8), dim3(ML_ROWS * ML_CG), (size_t)words(8) * 4, st, M, q.nc, nsys, (const int32_t*)q.flags, n8p, q.sys_map);
        else
            hipLaunchKernelGGL(k_ml_coarse<4>, dim3((M.n8 + ML_ROWS - 1) / ML_ROWS, (nsys + 3) / 4), dim3(ML_ROWS * ML_CG), (size_t)words(4) * 4, st, M, q.nc, nsys, (const int32_t*)q.flags, n8p, q.sys_map);
    }
    hipLaunchKernelGGL(k_ml_prolong, dim3((n + FG_BLOCK - 1) / FG_BLOCK, nsys), dim3(FG_BLOCK), 0, st, M, in, q.diag, n, q.nc, (const int32_t*)q.flags, out, q.sys_map);
}

// level schedules of the ILU(0) sweeps from the neighbour table (once per mesh); false: the mesh does not qualify
bool mb_ilu_prepare(fg_mb_state* s) {
    if (s->ilu_state != 0) return s->ilu_state > 0;
    s->ilu_state = -1;
    const int N = s->N, F = s->F;
    std::vector<int> lf(N, 0), lb(N, 0);
    for (int i = 0; i < N; ++i) {
        int lv = 0;
        for (int f = 0; f < F; ++f) {
            const int k = s->h_nbr[(size_t)f * N + i];
            for (int g = 0; g < f; ++g) if (k >= 0 && s->h_nbr[(size_t)g * N + i] == k) return false;   // the same neighbour across two faces
            if (k == i) return false;
            if (k >= 0 && k < i) lv = std::max(lv, lf[k] + 1);
        }
        lf[i] = lv;
    }
    for (int i = N - 1; i >= 0; --i) {
        int lv = 0;
        for (int f = 0; f < F; ++f) { const int j = s->h_nbr[(size_t)f * N + i]; if (j > i) lv = std::max(lv, lb[j] + 1); }
        lb[i] = lv;
    }
    auto schedule = [&](const std::vector<int>& lev, std::vector<int32_t>& start, int32_t** order_dev, int32_t** start_dev) -> int {
        const int nl = *std::max_element(lev.begin(), lev.end()) + 1;
        start.assign(nl + 1, 0);
        for (int i = 0; i < N; ++i) start[lev[i] + 1]++;
        for (int l = 0; l < nl; ++l) start[l + 1] += start[l];
        std::vector<int32_t> order(N), fill(start.begin(), start.end() - 1);
        for (int i = 0; i < N; ++i) order[fill[lev[i]]++] = i;
        if (int rc = mb_alloc(s, order_dev, (size_t)N)) return rc;
        if (int rc = mb_alloc(s, start_dev, start.size())) return rc;
        FG_HIP_CHECK(hipMemcpy(*order_dev, order.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice));
        FG_HIP_CHECK(hipMemcpy(*start_dev, start.data(), sizeof(int32_t) * start.size(), hipMemcpyHostToDevice));
        return FG_OK;
    };
    if (schedule(lf, s->ilu_start_f, &s->ilu_order_f, &s->ilu_start_f_dev) != FG_OK) return false;
    if (schedule(lb, s->ilu_start_b, &s->ilu_order_b, &s->ilu_start_b_dev) != FG_OK) return false;
    const size_t BN = (size_t)s->B * N;
    if (mb_alloc(s, &s->ilu_w, BN * F) != FG_OK || mb_alloc(s, &s->ilu_ud, BN) != FG_OK || mb_alloc(s, &s->ilu_mp, BN * s->d) != FG_OK ||
        mb_alloc(s, &s->ilu_ms, BN * s->d) != FG_OK)
        return false;
    s->ilu_state = 1;
    return true;
}

void mb_ilu_factor(fg_mb_state* s, const mb_real* dt, const mb_real* diag, const mb_real* off, hipStream_t st) {
    MB_DISPATCH(s, hipLaunchKernelGGL(k_mb_ilu_factor<DIMS>, dim3(s->B), dim3(1024), 0, st, s->dev, dt, diag, off, (const int32_t*)s->ilu_order_f,
                                      (const int32_t*)s->ilu_start_f_dev, (int)s->ilu_start_f.size() - 1, s->ilu_w, s->ilu_ud););
}
void mb_ilu_apply(fg_mb_state* s, const MbSolve& q, const mb_real* in, mb_real* out, hipStream_t st) {
    MB_DISPATCH(s, hipLaunchKernelGGL(k_mb_ilu_solve<DIMS>, dim3(q.nc, s->B), dim3(1024), 0, st, s->dev, q.nc, (const int32_t*)q.flags,
                                      (const int32_t*)s->ilu_order_f, (const int32_t*)s->ilu_start_f_dev, (int)s->ilu_start_f.size() - 1,
                                      (const int32_t*)s->ilu_order_b, (const int32_t*)s->ilu_start_b_dev, (int)s->ilu_start_b.size() - 1,
                                      (const mb_real*)s->ilu_w, (const mb_real*)s->ilu_ud, in, out););
}

int mb_bicgstab(fg_mb_state* s, const mb_real* dt, const mb_real* diag, const mb_real* off, const mb_real* rhs, mb_real* x, int nc,
                mb_real tol, int max_iterations, int use_x0, int* max_it, hipStream_t st, int project, int refine, int multilevel,
                int pred_slot) {
    const int nsys = s->B * nc, n = s->N;
    MbSolve q = mb_solve_ptrs(s, diag, off, rhs, x, nc, tol);
    q.project = project ? 1 : 0;
    // multilevel right preconditioning of the pressure solve: the recurrence runs on P M, the iterate advances along M p, M s
    // multilevel == 2: right preconditioning by ILU(0) of the matrix itself (the reference's preconditioned rung; mb_ilu_*)
    const bool ilu = multilevel == 2 && s->ilu_state > 0;
    const bool ml = ilu || (multilevel == 1 && nc == 1 && s->ml_on && s->ml_a4 != nullptr && s->ml_mp != nullptr);
    if (ilu) {
        q.mp = s->ilu_mp; q.ms = s->ilu_ms;
        mb_ilu_factor(s, dt, diag, off, st);
    } else if (ml) {
        q.mp = s->ml_mp; q.ms = s->ml_ms;
        hipLaunchKernelGGL(k_ml_scale, dim3(s->B), dim3(1024), 0, st, diag, n, s->ml_geom_diag_sum, s->ml_scale);
    }
    const dim3 sg((nsys + 63) / 64), sb(64), grid((n + FG_BLOCK - 1) / FG_BLOCK, nsys), blk(FG_BLOCK);
    // a refined solve that has not converged after 1500 iterations is not going to: hand over to the caller's CG fallback
    // instead of spending the reference's 5000 (one hard env would stall the whole batch)
    if (refine && max_iterations > 1500) max_iterations = 1500;
    // Four-cells-per-thread kernels (k_mbb_*4) whenever the cell count allows; FG_MB_BICG_VEC4 (read at create) is the test /
    // harness switch between the two kernel forms, not a workaround: the failures it once bisected were exact breakdowns of the
    // recurrence (MB_BETA), deterministic per kernel form because the two forms sum in different orders.
    const int vec_mask = (n % 4 != 0) ? 0 : (s->dbg_vec_mask & 31);
    const dim3 grid4((n / 4 + FG_BLOCK - 1) / FG_BLOCK, nsys);
    auto keep_best = [&](int first) {
        hipLaunchKernelGGL(k_mbr_best_decide, sg, sb, 0, st, q, s->best_res, s->best_keep, n, nsys, first);
        hipLaunchKernelGGL(k_mbr_best_copy, grid, blk, 0, st, n, (const int32_t*)s->best_keep, (const double*)s->x64, s->x64_best);
    };
    // a preconditioned or refined solve verifies convergence on the true residual (k_mbb_reopen); the plain fp32 recurrence stays
    // the reference's (bicgstab_solver_kernel.cu declares convergence on the recurrence residual)
    // s and t in one launch unless the recurrence is right-preconditioned (t = A M s needs all of s first); FG_MB_BICG_FUSE=0
    // (read at create) keeps the two kernels
    const bool fused_st = !ml && s->dbg_fuse_st >= 1, fused_pv = !ml && s->dbg_fuse_st >= 2;
    // multilevel: p and s are formed inside the restriction that follows them -- two launches fewer per iteration, which pays
    // while the launches are latency-sized (Airfoil2D x 16: 32.3-33.9 -> 34.7-34.8 env-steps/s; rocprofv3: p + restriction 10.1 -> 7.3 us, s + restriction
    // 10.4 -> 7.1 us with four threads per aggregate) and not once they carry bytes (x 64: 16.0 -> 17.4 us, 15.2 -> 14.3 us).  So: up to 32 systems.
    // FG_MB_ML_FUSE=0 never, 2 always.
    const bool ml_fused = ml && !ilu && (s->dbg_ml_fuse == 2 || (s->dbg_ml_fuse == 1 && nsys <= 32));
    if (fused_st) q.sbuf = s->w[5];
    const bool verify = ml || refine;
    int verify_rounds = 0;
    if (verify) FG_HIP_CHECK(hipMemsetAsync(s->verified, 0, sizeof(int32_t) * nsys, st));
    hipLaunchKernelGGL(k_mbs_begin, sg, sb, 0, st, dt, q, nsys);
    if (refine) {
        hipLaunchKernelGGL(k_mbr_fold, grid, blk, 0, st, n, q, s->x64, use_x0 ? 1 : 0);
        MB_DISPATCH(s, hipLaunchKernelGGL(k_mbr_residual<DIMS>, grid, blk, 0, st, s->dev, q, (const double*)s->x64, project ? A_ST : -1, project ? 1 : 0););
    } else {
        MB_DISPATCH(s, hipLaunchKernelGGL(k_mbs_init<DIMS>, grid, blk, 0, st, s->dev, q, use_x0, project ? A_ST : -1, project ? 1 : 0););
    }
    if (project) hipLaunchKernelGGL(k_mbb_project_init, grid, blk, 0, st, n, q);
    if (refine) keep_best(1);
    bool done = false;
    // first convergence poll where the previous solve of this kind finished (kernels of converged systems exit at once, so running
    // a few launches past convergence costs ~2 us each, while every poll is a stream synchronisation: 10-20 us of idle GPU), then
    // every 2 (every 10 beyond 20) iterations
    int& pred = s->pred_bicg[pred_slot & 31];
    int next_poll = (pred > 2 && s->dbg_pred) ? pred : 2;
    const int BICG_RESTART = refine ? 100 : 200;
    int n_map = 0;      // systems of the compacted per-iteration launches (0: all systems, identity)
    auto update_map = [&]() -> int {     // after a poll: the systems that still iterate (flags_pinned == 0), when they are few
        if (!s->dbg_compact || ilu) return FG_OK;
        int act = 0;
        for (int i = 0; i < nsys; ++i) act += s->flags_pinned[i] == 0;
        if (act == 0 || act > 16 || 4 * act > nsys) { n_map = 0; return FG_OK; }
        bool same = (act == n_map);
        int k = 0;
        for (int i = 0; i < nsys; ++i)
            if (s->flags_pinned[i] == 0) { same = same && s->sys_map_pinned[k] == i; ++k; }
        if (same) return FG_OK;
        k = 0;
        for (int i = 0; i < nsys; ++i) if (s->flags_pinned[i] == 0) s->sys_map_pinned[k++] = i;
        // (the previous upload has executed: the poll that just returned was enqueued behind it)
        FG_HIP_CHECK(hipMemcpyAsync(s->sys_map_dev, s->sys_map_pinned, sizeof(int32_t) * act, hipMemcpyHostToDevice, st));
        n_map = act;
        return FG_OK;
    };
    for (int it = 0; it < max_iterations && !done; ++it) {
        if (it > 0 && it % BICG_RESTART == 0) {
            q.it_base = it;
            q.p = s->w[2]; q.v = s->w[3];   // the re-initialisation lays p = r down in buffer 0 of the pair (iteration index 0)
            hipLaunchKernelGGL(k_mbb_restart, sg, sb, 0, st, q, nsys);
            if (refine) {
                hipLaunchKernelGGL(k_mbr_fold, grid, blk, 0, st, n, q, s->x64, 2);
                MB_DISPATCH(s, hipLaunchKernelGGL(k_mbr_residual<DIMS>, grid, blk, 0, st, s->dev, q, (const double*)s->x64, project ? A_ST : -1, project ? 1 : 0););
            } else {
                MB_DISPATCH(s, hipLaunchKernelGGL(k_mbs_init<DIMS>, grid, blk, 0, st, s->dev, q, 1, project ? A_ST : -1, project ? 1 : 0););
            }
            if (project) hipLaunchKernelGGL(k_mbb_project_init, grid, blk, 0, st, n, q);
            if (refine) keep_best(0);
        }
        const int li = it - q.it_base;
        if (fused_pv) {   // p and v of iteration li live in buffer li & 1 of their pair
            q.p = (li & 1) ? s->w[6] : s->w[2]; q.p_prev = (li & 1) ? s->w[2] : s->w[6];
            q.v = (li & 1) ? s->w[7] : s->w[3]; q.v_prev = (li & 1) ? s->w[3] : s->w[7];
        }
        // Compacted launches: while only a few systems of the batch still iterate (envs differ: the slowest of 64 pressure systems
        // needs 2-3x the mean), the kernels of an iteration are launched over those systems only -- grid.y = n_map, system =
        // sys_map[blockIdx.y] -- instead of over all of them with most workgroups leaving at once (Airfoil2D x 64 with random actions:
        // 47 % of all launched iterations ran for 1-3 systems, ~5.5 us per launch against ~2.5)
        MbSolve qi = q;
        dim3 gi = grid, gi4 = grid4;
        if (n_map > 0) { qi.sys_map = s->sys_map_dev; qi.n_map = n_map; gi.y = gi4.y = (unsigned)n_map; }
        MB_DISPATCH(s, {   // vec_mask: which of the five kernels run in their four-cell form
            if (fused_pv) {
                if ((vec_mask & 3) == 3) hipLaunchKernelGGL(k_mbb_pv4<DIMS>, gi4, blk, 0, st, s->dev, qi, li); else hipLaunchKernelGGL(k_mbb_pv<DIMS>, gi, blk, 0, st, s->dev, qi, li);
            } else {
            if (ml_fused) {}   // p is formed inside the restriction (mb_ml_apply below)
            else if (vec_mask & 1) hipLaunchKernelGGL(k_mbb_p4<DIMS>, gi4, blk, 0, st, s->dev, qi, li); else hipLaunchKernelGGL(k_mbb_p<DIMS>, gi, blk, 0, st, s->dev, qi, li);
            if (ilu) mb_ilu_apply(s, qi, qi.p, s->ilu_mp, st);
            else if (ml) mb_ml_apply(s, qi, qi.p, s->ml_mp, st, ml_fused ? 1 : 0, li);
            if (vec_mask & 2) hipLaunchKernelGGL(k_mbb_v4<DIMS>, gi4, blk, 0, st, s->dev, qi, li); else hipLaunchKernelGGL(k_mbb_v<DIMS>, gi, blk, 0, st, s->dev, qi, li);
            }
            if (fused_st) {
                if ((vec_mask & 12) == 12) hipLaunchKernelGGL(k_mbb_st4<DIMS>, gi4, blk, 0, st, s->dev, qi, li); else hipLaunchKernelGGL(k_mbb_st<DIMS>, gi, blk, 0, st, s->dev, qi, li);
            } else {
                if (ml_fused) {}   // s is formed inside the restriction
                else if (vec_mask & 4) hipLaunchKernelGGL(k_mbb_s4<DIMS>, gi4, blk, 0, st, s->dev, qi, li); else hipLaunchKernelGGL(k_mbb_s<DIMS>, gi, blk, 0, st, s->dev, qi, li);
                if (ilu) mb_ilu_apply(s, qi, qi.r, s->ilu_ms, st);
                else if (ml) mb_ml_apply(s, qi, qi.r, s->ml_ms, st, ml_fused ? 2 : 0, li);
                if (vec_mask & 8) hipLaunchKernelGGL(k_mbb_t4<DIMS>, gi4, blk, 0, st, s->dev, qi, li); else hipLaunchKernelGGL(k_mbb_t<DIMS>, gi, blk, 0, st, s->dev, qi, li);
            }
            if (vec_mask & 16) hipLaunchKernelGGL(k_mbb_x4<DIMS>, gi4, blk, 0, st, s->dev, qi, li); else hipLaunchKernelGGL(k_mbb_x<DIMS>, gi, blk, 0, st, s->dev, qi, li);
        });
        if (it + 1 >= next_poll || it + 1 == max_iterations) {
            next_poll = it + 1 + (it < 20 ? 2 : 10);   // long (pressure) solves: fewer host round trips
            FgPollOut po = fg_poll_next(&s->poll);
            hipLaunchKernelGGL(k_mbs_check, sg, sb, 0, st, q, s->info_pinned, s->flags_pinned, A_RR, it, n, nsys, (int)(it + 1 == max_iterations), -1, po);
            if (int rc = mb_poll(s, nsys, st, done, po, true)) return rc;
            if (int rc = update_map()) return rc;
            if (nc == 1 && s->dbg_trace) {
                mb_real lo = 1e30f, hi = 0.f; int active = 0;
                for (int i = 0; i < nsys; ++i) { const mb_real c = s->info_pinned[i].final_residual; lo = c < lo ? c : lo; hi = c > hi ? c : hi; active += s->flags_pinned[i] == 0; }
                fprintf(stderr, "[mb_bicg] it %4d residual min %.3e max %.3e active %d\n", it + 1, lo, hi, active);
            }
            if (done && verify && verify_rounds < 3 && it + 1 < max_iterations) {   // see k_mbb_reopen
                bool any = false;
                for (int i = 0; i < nsys; ++i) any = any || (s->flags_pinned[i] == 1 && s->info_pinned[i].converged && s->info_pinned[i].is_finite);
                if (any) {
                    ++verify_rounds;
                    hipLaunchKernelGGL(k_mbb_reopen, sg, sb, 0, st, q, s->verified, nsys);
                    q.it_base = it + 1;
                    q.p = s->w[2]; q.v = s->w[3];
                    if (refine) {
                        hipLaunchKernelGGL(k_mbr_fold, grid, blk, 0, st, n, q, s->x64, 2);
                        MB_DISPATCH(s, hipLaunchKernelGGL(k_mbr_residual<DIMS>, grid, blk, 0, st, s->dev, q, (const double*)s->x64, project ? A_ST : -1, project ? 1 : 0););
                    } else {
                        MB_DISPATCH(s, hipLaunchKernelGGL(k_mbs_init<DIMS>, grid, blk, 0, st, s->dev, q, 1, project ? A_ST : -1, project ? 1 : 0););
                    }
                    if (project) hipLaunchKernelGGL(k_mbb_project_init, grid, blk, 0, st, n, q);
                    hipLaunchKernelGGL(k_mbb_verify, sg, sb, 0, st, q, s->verified, n, nsys, (int)(verify_rounds == 3));
                    if (refine) keep_best(0);
                    po = fg_poll_next(&s->poll);
                    hipLaunchKernelGGL(k_mbs_check, sg, sb, 0, st, q, s->info_pinned, s->flags_pinned, A_RR, it, n, nsys, 0, -1, po);
                    if (int rc = mb_poll(s, nsys, st, done, po, true)) return rc;
                    if (int rc = update_map()) return rc;
                    next_poll = it + 1 + 2;
                    if (nc == 1 && s->dbg_trace) {
                        int open = 0;
                        for (int i = 0; i < nsys; ++i) open += s->flags_pinned[i] == 0;
                        fprintf(stderr, "[mb_bicg] it %4d verification round %d: %d system(s) iterate on\n", it + 1, verify_rounds, open);
                    }
                }
            }
        }
    }
    if (refine) {
        hipLaunchKernelGGL(k_mbr_best_restore, grid, blk, 0, st, n, q, s->x64, (const double*)s->x64_best, (const mb_real*)s->best_res);
        hipLaunchKernelGGL(k_mbr_fold, grid, blk, 0, st, n, q, s->x64, 3);
        FG_HIP_CHECK(hipMemcpyAsync(s->info_pinned, s->info_dev, sizeof(fg_solve_info) * nsys, hipMemcpyDeviceToHost, st));
        FG_HIP_CHECK(hipStreamSynchronize(st));
    }
    const int frc = mb_finish(s, nsys, nullptr, max_it);
    if (max_it && frc == FG_OK) pred = *max_it < 200 ? *max_it : 200;
    if (frc == FG_ERR_NOT_CONVERGED && s->dbg_fail)      // FG_MB_TRACE_FAIL: how far from the tolerance a capped solve ended
        for (int i = 0; i < nsys; ++i)
            if (!s->info_pinned[i].converged)
                fprintf(stderr, "[mb_bicg] capped system %d (nc %d, ilu %d, project %d, refine %d): it %d residual %g tol %g\n", i, nc, ilu, project, refine,
                        (int)s->info_pinned[i].used_iterations, (double)s->info_pinned[i].final_residual, (double)tol);
    if (frc == FG_ERR_NOT_FINITE && s->dbg_fail) {   // rare path, FG_MB_TRACE_FAIL only: the recurrence scalars of the systems that broke down
        std::vector<FgDacc> acc_raw((size_t)nsys * MB_ACC);
        std::vector<double> acc((size_t)nsys * MB_ACC);
        std::vector<mb_real> sc((size_t)nsys * 2);
        (void)hipMemcpy(acc_raw.data(), s->acc, acc_raw.size() * sizeof(FgDacc), hipMemcpyDeviceToHost);
        for (size_t k = 0; k < acc.size(); ++k) acc[k] = fg_dacc_host_value(acc_raw[k]);
        (void)hipMemcpy(sc.data(), s->sc, sc.size() * sizeof(mb_real), hipMemcpyDeviceToHost);
        for (int i = 0; i < nsys; ++i) {
            if (s->info_pinned[i].is_finite) continue;
            fprintf(stderr, "[mb_bicg] non-finite system %d (nc %d, vec_mask %d, project %d, refine %d): it %d residual %g alpha %g omega %g acc",
                    i, nc, vec_mask, project, refine, (int)s->info_pinned[i].used_iterations, s->info_pinned[i].final_residual, sc[2 * i], sc[2 * i + 1]);
            for (int k = 0; k < MB_ACC; ++k) fprintf(stderr, " %g", acc[(size_t)i * MB_ACC + k]);
            fprintf(stderr, "\n");
        }
    }
    return frc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Point-Jacobi sweeps for the velocity systems (policy advection_jacobi, as csrc/fg_jacobi.hip on the single-block path).  The rows of
// the advection-diffusion matrix of the cylinder meshes are dominated by 1/dt at the envs' time steps: x <- D^-1 (b - O x) contracts
// the residual by ~0.2 per sweep there (profiles/jacobi_exp_multiblock.py: 12 sweeps to the criterion, where BiCGStab takes 5-6
// iterations of three launches with two matrix applications).  A sweep is one launch over the neighbour table -- one matrix pass, no
// dot product; the sweeps that are followed by a check also sum the residual of the iterate they started from, D (x_new - x) = b - C x.
// The airfoil's systems contract by 0.8 per sweep: the first check sees that and hands the solve to BiCGStab.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mbj_sweep(MbDev D, MbSolve q, const mb_real* __restrict__ xin, mb_real* __restrict__ xout, int from_zero,
                                                        int measure_slot) {
    MB_SYS
    if (flag_ld(q.flags + (sys)) != 0) return;
    constexpr int F = 2 * DIMS;
    mb_real part = 0.f;
    if (valid) {
        const mb_real dg = q.diag[(size_t)b * N + i];
        mb_real acc = q.rhs[vb + i];
        mb_real xold = 0.f;
        if (!from_zero) {
            xold = xin[vb + i];
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int nb = D.nbr[(size_t)f * N + i];
                if (nb >= 0) acc -= q.off[((size_t)b * F + f) * N + i] * xin[vb + nb];
            }
        }
        const mb_real xn = acc / dg;
        xout[vb + i] = xn;
        const mb_real r = dg * (xn - xold);
        part = r * r;
    }
    if (measure_slot >= 0) {
        part = mb_block_sum(part, lds);
        if (threadIdx.x == 0) acc_add(a + measure_slot, (double)part);
    }
}

// the same sweep for ALL components of an env in one thread: they share the matrix row and the neighbour indices (grid.y = env; a
// component whose system has stopped keeps its iterate)
template <int DIMS, int NC>
__global__ __launch_bounds__(FG_BLOCK) void k_mbj_sweep_env(MbDev D, MbSolve q, const mb_real* __restrict__ xin, mb_real* __restrict__ xout, int from_zero,
                                                            int measure_slot) {
    constexpr int F = 2 * DIMS;
    const int i = blockIdx.x * FG_BLOCK + threadIdx.x, b = blockIdx.y, N = D.N;
    __shared__ mb_real lds[16];
    bool live[NC], any = false;
#pragma unroll
    for (int c = 0; c < NC; ++c) { live[c] = flag_ld(q.flags + (b * NC + c)) == 0; any = any || live[c]; }
    if (!any) return;
    mb_real part[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) part[c] = 0.f;
    if (i < N) {
        const mb_real dg = q.diag[(size_t)b * N + i];
        const mb_real rd = (mb_real)1 / dg;
        mb_real acc[NC], xold[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { acc[c] = q.rhs[((size_t)b * NC + c) * N + i]; xold[c] = from_zero ? (mb_real)0 : xin[((size_t)b * NC + c) * N + i]; }
        if (!from_zero) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int nb = D.nbr[(size_t)f * N + i];
                if (nb >= 0) {
                    const mb_real o = q.off[((size_t)b * F + f) * N + i];
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[c] -= o * xin[((size_t)b * NC + c) * N + nb];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (!live[c]) continue;
            const mb_real xn = acc[c] * rd;
            xout[((size_t)b * NC + c) * N + i] = xn;
            const mb_real r = dg * (xn - xold[c]);
            part[c] = r * r;
        }
    }
    if (measure_slot >= 0) {
        mb_block_sums<NC>(part, lds);
        if (threadIdx.x < NC && live[threadIdx.x < NC ? threadIdx.x : 0]) {
            mb_real v = part[0];
#pragma unroll
            for (int c = 1; c < NC; ++c) if ((int)threadIdx.x == c) v = part[c];
            acc_add(q.acc + (size_t)(b * NC + threadIdx.x) * MB_ACC + measure_slot, (double)v);
        }
    }
}

// verdict behind a measuring sweep (k_mbs_check's rule on the sweep's own sum), mirrors, and both measured residuals for the host
__global__ void k_mbj_check(MbSolve q, fg_solve_info* __restrict__ mirror, int32_t* __restrict__ flag_mirror, mb_real* __restrict__ res2,
                            int slot_now, int slot_prev, int sweeps, int n, int nsys, FgPollOut poll) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    mb_real now = -1.f, prev = -1.f;
    if (flag_ld(q.flags + (s)) == 0) {
        now = mb_rms(acc_ld(q.acc + ((size_t)s * MB_ACC + slot_now)), n);
        if (slot_prev >= 0) prev = mb_rms(acc_ld(q.acc + ((size_t)s * MB_ACC + slot_prev)), n);
        q.info[s].final_residual = now;
        q.info[s].used_iterations = sweeps;
        if (!(now >= q.tol)) {
            const bool finite = isfinite(now);
            flag_st(q.flags + (s), finite ? 1 : 2);
            q.info[s].converged = finite ? 1 : 0;
            q.info[s].is_finite = finite ? 1 : 0;
        }
    }
    res2[2 * s] = now; res2[2 * s + 1] = prev;
    mirror[s] = q.info[s];
    flag_mirror[s] = flag_ld(q.flags + (s));
    fg_poll_publish(poll, s);
}
}  // namespace

int mb_jacobi(fg_mb_state* s, const mb_real* dt, const mb_real* diag, const mb_real* off, const mb_real* rhs, mb_real* x, int nc, mb_real tol,
              int use_x0, int* max_it, hipStream_t st, int pred_slot, int* outcome) {
    *outcome = 0;
    const int ps = pred_slot & 3;
    if (s->jac_skip[ps] > 0) { --s->jac_skip[ps]; return FG_OK; }
    *outcome = 2;
    // Check points at FIXED sweep counts -- 12, 14, ... 32 (beyond that BiCGStab is the cheaper iteration: the airfoil meshes need
    // 40-55 sweeps) -- each with a verdict on the device: a system stops at the first check point where its residual is below the
    // tolerance, whatever the host enqueued ahead.  The history of the handle only decides how far ahead that is (where the first
    // poll sits), so an env's iterate does not depend on it and replays repeat exactly.  Every odd sweep from the tenth on sums the
    // residual of the iterate it started from; a check reads the last two sums (the contraction per sweep, for the host).
    constexpr int FIRST = 12, STEP = 2, CHECKS = 11;
    const int nsys = s->B * nc, n = s->N;
    MbSolve q = mb_solve_ptrs(s, diag, off, rhs, x, nc, tol);
    const dim3 sg((nsys + 63) / 64), sb(64), grid((n + FG_BLOCK - 1) / FG_BLOCK, nsys), blk(FG_BLOCK);
    hipLaunchKernelGGL(k_mbs_begin, sg, sb, 0, st, dt, q, nsys);
    mb_real* buf[2] = {x, s->w[0]};      // sweep k writes buf[(k + 1) & 1]: the check points are even counts, so each ends in x
    int sweeps = 0, checks = 0;
    auto run_to_check = [&](int upto, const FgPollOut& po_last) {      // sweeps and checks up to check point number `upto` (1-based)
        while (checks < upto) {
            const int target = FIRST + STEP * checks;
            for (; sweeps < target; ++sweeps) {
                const int slot = (sweeps >= FIRST - 3 && (sweeps & 1)) ? (sweeps - (FIRST - 3)) / 2 : -1;      // sweeps 9, 11, ... -> slots 0, 1, ...
                const int w = (sweeps + 1) & 1;
                const int fz = (sweeps == 0 && !use_x0) ? 1 : 0;
                if (nc == s->d) {      // the velocity systems: all components of an env in one thread
                    const dim3 genv(grid.x, s->B);
                    if (s->d == 2) hipLaunchKernelGGL((k_mbj_sweep_env<2, 2>), genv, blk, 0, st, s->dev, q, (const mb_real*)buf[w ^ 1], buf[w], fz, slot);
                    else hipLaunchKernelGGL((k_mbj_sweep_env<3, 3>), genv, blk, 0, st, s->dev, q, (const mb_real*)buf[w ^ 1], buf[w], fz, slot);
                } else {
                    MB_DISPATCH(s, hipLaunchKernelGGL(k_mbj_sweep<DIMS>, grid, blk, 0, st, s->dev, q, (const mb_real*)buf[w ^ 1], buf[w], fz, slot););
                }
            }
            ++checks;
            const int now = (target - 1 - (FIRST - 3)) / 2;      // the sum of sweep target - 1; the one before it is two sweeps older
            hipLaunchKernelGGL(k_mbj_check, sg, sb, 0, st, q, s->info_pinned, s->flags_pinned, s->jac_res_pinned, now, now - 1,
                               sweeps, n, nsys, checks == upto ? po_last : FgPollOut{nullptr, 0});
        }
    };
    static_assert((FIRST + STEP * (CHECKS - 1) - 1 - (FIRST - 3)) / 2 < MB_ACC, "one accumulator per measuring sweep");
    // first poll at the check point the previous solve of this pass ended at
    int upto = 1;
    if (s->jac_sweeps[ps] > FIRST) upto = 1 + (s->jac_sweeps[ps] - FIRST + STEP - 1) / STEP;
    if (upto > CHECKS) upto = CHECKS;
    bool ok = false, by_cluster = false;
    // the sweeps of an env by a cluster of workgroups, one launch per solve (k_mbj_cluster, fg_mb_cluster.hip): the same sweeps, check
    // points and verdicts; its give-up rule is this loop's, applied per env
    if (mb_jacobi_cluster_ok(s, nc)) {
        bool fell_back = false, done = false;
        if (int rc = mb_jacobi_cluster(s, dt, diag, off, rhs, x, tol, use_x0, st, &fell_back, &done)) return rc;
        if (!fell_back) {
            by_cluster = true;
            bool bad = false;
            for (int i = 0; i < nsys; ++i) bad = bad || !s->info_pinned[i].is_finite;
            ok = done && !bad;
        }
    }
    while (!by_cluster) {
        const FgPollOut po = fg_poll_next(&s->poll);
        run_to_check(upto, po);
        bool done = false;
        if (int rc = mb_poll(s, nsys, st, done, po, false)) return rc;
        bool bad = false;
        double need = 0.0;
        for (int i = 0; i < nsys; ++i) {
            if (!s->info_pinned[i].is_finite) bad = true;
            if (s->flags_pinned[i] != 0) continue;
            const double r1 = s->jac_res_pinned[2 * i], r0 = s->jac_res_pinned[2 * i + 1];
            if (r0 > 0.0 && r1 > 0.0) {
                const double c = sqrt(r1 / r0);      // per sweep (the two measuring sweeps are two apart)
                if (!(c < 0.85)) bad = true;
                else { const double m = log((double)tol / r1) / log(c); need = m > need ? m : need; }
            } else {
                need = need > 4.0 ? need : 4.0;
            }
        }
        if (done && !bad) { ok = true; break; }
        if (bad) break;
        const int more = 1 + (int)(ceil(need) - 1) / STEP;      // check points still to go at that contraction
        if (checks + more > CHECKS) break;
        upto = checks + (more < 1 ? 1 : more);
    }
    if (!ok) {
        // handed over: 3 = BiCGStab may start from the sweeps' iterate, 2 = it starts from zero.  From the iterate only when every
        // system still iterating is finite and well above the tolerance: sweeps that stall just above it sit at fp32's attainable
        // accuracy, the residual of such an iterate is rounding noise, and a BiCGStab recurrence started on noise has been seen to
        // report convergence on a wrong iterate (Airfoil2D, resolution_div 2: one env's drag -5.0 for 0.34).
        bool from_iterate = true;
        for (int i = 0; i < nsys; ++i) {
            const double r1 = (double)s->jac_res_pinned[2 * i];
            from_iterate = from_iterate && s->info_pinned[i].is_finite && std::isfinite(r1);
            if (s->flags_pinned[i] == 0 && !(r1 > 8.0 * (double)tol)) from_iterate = false;
        }
        *outcome = from_iterate ? 3 : 2;
        s->jac_fails[ps] += 1;
        s->jac_skip[ps] = s->jac_fails[ps] > 6 ? 512 : (4 << s->jac_fails[ps]);
        s->jac_sweeps[ps] = 0;
        return FG_OK;
    }
    s->jac_fails[ps] = 0;
    const int frc = mb_finish(s, nsys, nullptr, max_it);
    int used = 0;
    for (int i = 0; i < nsys; ++i) used = std::max(used, (int)s->info_pinned[i].used_iterations);
    s->jac_sweeps[ps] = used > 0 ? used : FIRST;
    *outcome = 1;
    return frc;
}

// Pressure BiCGStab with the multilevel right preconditioner as a TRIAL.  On the Airfoil2D mesh the attempt converges in a third
// of the plain iterations (17 against 55-63), verified on the true residual -- but a geometry-only symmetric coarse operator is no
// safe preconditioner for that non-symmetric matrix in every state (the stiff solves right after an impulsive start exceed any
// sensible cap), so the attempt is capped (200 iterations), a failed attempt is repeated with the plain recurrence (from the kept
// iterate; from zero after a non-finite one), and the handle backs off: the next `backoff` solves run plain, the back-off doubles
// with every failure (4 ... 256) and halves with every success.
int mb_pressure_bicgstab(fg_mb_state* s, const mb_real* dt, mb_real tol, int max_iterations, int use_x0, int* max_it, hipStream_t st, int project,
                         int refine, int pred_slot) {
    const bool have_ml = s->ml_on && s->ml_a4 != nullptr && s->ml_mp != nullptr && s->d == 2;
    if (have_ml && s->ml_bicg_skip > 0) --s->ml_bicg_skip;
    else if (have_ml) {
        ++s->ml_bicg_attempts;
        const int cap = max_iterations < s->dbg_ml_cap ? max_iterations : s->dbg_ml_cap;
        const int rc = mb_bicgstab(s, dt, s->Pdiag, s->Poff, s->div, s->pres, 1, tol, cap, use_x0, max_it, st, project, refine, 1, (pred_slot + 16) & 31);
        if (rc == FG_OK) { s->ml_bicg_backoff = s->ml_bicg_backoff > 4 ? s->ml_bicg_backoff / 2 : 4; return rc; }
        if (rc != FG_ERR_NOT_CONVERGED && rc != FG_ERR_NOT_FINITE) return rc;
        ++s->ml_bicg_failures;
        s->ml_bicg_skip = s->ml_bicg_backoff;
        s->ml_bicg_backoff = s->ml_bicg_backoff < 256 ? s->ml_bicg_backoff * 2 : 256;
        use_x0 = (rc == FG_ERR_NOT_CONVERGED && refine) ? 1 : 0;   // the refined solver handed back its best refinement point
    }
    return mb_bicgstab(s, dt, s->Pdiag, s->Poff, s->div, s->pres, 1, tol, max_iterations, use_x0, max_it, st, project, refine, 0, pred_slot);
}

// project_mean: every residual is used with its mean removed.  For a symmetric matrix with the constant null space (an
// orthogonal mesh) that changes nothing; with cross-metric terms 1^T P != 0, the plain recurrence accumulates a constant
// residual component that no search direction can reduce (the solve stalls just above the envs' tolerance and cannot be
// warm-started), and removing it is what makes the singular system consistent.
int mb_cg(fg_mb_state* s, const mb_real* dt, const mb_real* diag, const mb_real* off, const mb_real* rhs, mb_real* x, mb_real tol,
          int max_iterations, int use_x0, int project_mean, mb_real stall_accept, int* max_it, hipStream_t st) {
    const int nsys = s->B, n = s->N;
    {
        const int pm = project_mean ? (s->yproj_const ? 1 : 2) : 0;
        if (mb_cluster_ok(s, pm, diag, off)) {
            bool fell_back = false;
            const int rc = mb_cg_cluster(s, dt, rhs, x, tol, max_iterations, use_x0, pm, stall_accept, max_it, st, &fell_back);
            if (!fell_back) return rc;
        }
        if (mb_onchip_ok(s, pm)) return mb_cg_onchip(s, dt, diag, off, rhs, x, tol, max_iterations, use_x0, pm, stall_accept, max_it, st);
    }
    MbSolve q = mb_solve_ptrs(s, diag, off, rhs, x, 1, tol);
    q.rw = nullptr;
    q.best_x = s->w[4]; q.best_it = s->best_it; q.stall_limit = s->cg_stall_limit;
    q.accept_factor = stall_accept > 1.f ? stall_accept : 0.f; q.accept_window = 20;
    const dim3 sg((nsys + 63) / 64), sb(64), grid((n + FG_BLOCK - 1) / FG_BLOCK, nsys), blk(FG_BLOCK);
    const bool vec4 = (n % 4 == 0) && !s->dbg_scalar_cg;   // FG_MB_SCALAR_CG=1 forces the one-cell-per-thread kernels
    const dim3 grid4((n / 4 + FG_BLOCK - 1) / FG_BLOCK, nsys);
    hipLaunchKernelGGL(k_mbs_begin, sg, sb, 0, st, dt, q, nsys);
    MB_DISPATCH(s, hipLaunchKernelGGL(k_mbs_init<DIMS>, grid, blk, 0, st, s->dev, q, use_x0, project_mean ? C_SUM : -1, 0););
    bool done = false;
    // CG_CHUNK iterations + the convergence check are one hipGraph: at 14 k cells x 64 envs a kernel runs 5-10 us, about
    // what the host needs to enqueue it, so the loop was launch-bound.  The kernels take their iteration index from a
    // device counter (q.it_ctr) so that one captured chunk serves every replay.
    constexpr int CG_CHUNK = 20, CG_RESTART = 100;
    q.max_iterations = ((max_iterations + CG_CHUNK - 1) / CG_CHUNK) * CG_CHUNK;
    FG_HIP_CHECK(hipMemsetAsync(s->it_ctr, 0, 3 * sizeof(int32_t), st));
    int active_now = 0;  // systems still iterating, from the poll before this chunk (all active ones at chunk 0)
    auto prof_collect = [&]() -> int {
        for (int k = 0; k < s->prof_used; ++k) {
            fg_f32 ms = 0.f;
            FG_HIP_CHECK(hipEventElapsedTime(&ms, s->prof_ev[2 * k], s->prof_ev[2 * k + 1]));
            const int kind = s->prof_kind[k];
            if (s->prof_active[k] > 0) {
                // algorithmic bytes per cell: stencil kernel r, p_old, diag, 2d off, p_new, v (+ the neighbour table, shared
                // by the env batch); update kernel x (r/w), p, v, r (r/w)
                const double per_cell = kind == 0 ? 4.0 * (5 + 2 * s->d) + 4.0 * 2 * s->d / (double)s->B : 24.0;
                s->prof_ms[kind] += ms;
                s->prof_bytes[kind] += per_cell * (double)n * s->prof_active[k];
                s->prof_n[kind] += 1;
            }
        }
        s->prof_used = 0;
        return FG_OK;
    };
    const int pm_mode = project_mean ? (s->yproj_const ? 1 : 2) : 0;
    auto enqueue_chunk = [&](bool sample, FgPollOut po) {   // po: sequence words of the poll that follows ({nullptr, 0} inside a captured graph)
        MB_DISPATCH_PM(s, pm_mode, {
            for (int k = 0; k < CG_CHUNK; ++k) {
                const bool ev = sample && k == 0 && s->prof_used + 2 <= 32;
                const int e0 = s->prof_used;
                if (ev) {
                    s->prof_kind[e0] = 0; s->prof_kind[e0 + 1] = 1;
                    s->prof_active[e0] = s->prof_active[e0 + 1] = active_now;
                    s->prof_used += 2;
                }
                s->prof_launches[0] += 1; s->prof_launches[1] += 1;
                if (vec4) {
                    if (ev) {
                        hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_mbc_ap4<DIMS, PM>), grid4, blk, 0, st, s->prof_ev[2 * e0], s->prof_ev[2 * e0 + 1], 0, s->dev, q, s->w[1], s->w[2], -1, project_mean);
                        hipExtLaunchKernelGGL(k_mbc_update4, grid4, blk, 0, st, s->prof_ev[2 * e0 + 2], s->prof_ev[2 * e0 + 3], 0, n, q, (const mb_real*)s->w[1], (const mb_real*)s->w[2], -1, project_mean, s->dev.yproj);
                    } else {
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mbc_ap4<DIMS, PM>), grid4, blk, 0, st, s->dev, q, s->w[1], s->w[2], -1, project_mean);
                        hipLaunchKernelGGL(k_mbc_update4, grid4, blk, 0, st, n, q, (const mb_real*)s->w[1], (const mb_real*)s->w[2], -1, project_mean, s->dev.yproj);
                    }
                } else {
                    if (ev) {
                        hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_mbc_ap<DIMS, PM>), grid, blk, 0, st, s->prof_ev[2 * e0], s->prof_ev[2 * e0 + 1], 0, s->dev, q, s->w[1], s->w[2], -1, project_mean);
                        hipExtLaunchKernelGGL(k_mbc_update<DIMS>, grid, blk, 0, st, s->prof_ev[2 * e0 + 2], s->prof_ev[2 * e0 + 3], 0, s->dev, q, (const mb_real*)s->w[1], (const mb_real*)s->w[2], -1, project_mean);
                    } else {
                        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mbc_ap<DIMS, PM>), grid, blk, 0, st, s->dev, q, s->w[1], s->w[2], -1, project_mean);
                        hipLaunchKernelGGL(k_mbc_update<DIMS>, grid, blk, 0, st, s->dev, q, (const mb_real*)s->w[1], (const mb_real*)s->w[2], -1, project_mean);
                    }
                }
            }
        });
        hipLaunchKernelGGL(k_mbs_check, sg, sb, 0, st, q, s->info_pinned, s->flags_pinned, 0, -1, n, nsys, 0, project_mean ? 0 : -1, po);
    };
    // the chunk can be replayed as a hipGraph (FG_MB_GRAPH=1); since the four-cells-per-thread kernels the loop is no
    // longer enqueue-bound and plain launches are as fast, so that is the default (and what the live profiler samples)
    const bool use_graph = s->dbg_graph && !s->prof_on;
    const bool trace = s->dbg_trace != 0;
    if (use_graph) {
        MbGraphKey key;
        memset(&key, 0, sizeof(key));
        key.q = q; key.vec4 = vec4; key.project_mean = pm_mode; key.stream = st;
        static_assert(sizeof(MbGraphKey) <= sizeof(s->cg_graph_key_storage), "graph key storage too small");
        MbGraphKey& stored = *reinterpret_cast<MbGraphKey*>(s->cg_graph_key_storage);
        if (!s->cg_graph_exec || memcmp(&key, &stored, sizeof(key)) != 0) {
            if (s->cg_graph_exec) { (void)hipGraphExecDestroy(s->cg_graph_exec); s->cg_graph_exec = nullptr; }
            hipGraph_t graph = nullptr;
            // captured on a private stream (the caller's may be the legacy default stream, which cannot capture); the
            // instantiated graph is then launched on the caller's stream
            if (!s->capture_stream) FG_HIP_CHECK(hipStreamCreateWithFlags(&s->capture_stream, hipStreamNonBlocking));
            const hipStream_t run_stream = st;
            st = s->capture_stream;
            FG_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            enqueue_chunk(false, FgPollOut{nullptr, 0});
            FG_HIP_CHECK(hipStreamEndCapture(st, &graph));
            st = run_stream;
            FG_HIP_CHECK(hipGraphInstantiate(&s->cg_graph_exec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
            memcpy(&stored, &key, sizeof(key));
        }
    }
    int recoveries = 0;
    bool need_restart = false;
    for (int it = 0; it < q.max_iterations && !done; it += CG_CHUNK) {
        if (it > 0 && (it % CG_RESTART == 0 || need_restart)) {
            need_restart = false;
            hipLaunchKernelGGL(k_mbc_clear, sg, sb, 0, st, q, nsys, it);
            MB_DISPATCH(s, hipLaunchKernelGGL(k_mbc_restart<DIMS>, grid, blk, 0, st, s->dev, q, it, project_mean););
        }
        if (it == 0) { active_now = 0; for (int i = 0; i < nsys; ++i) active_now += 1; }  // inactive envs exit in k_mbs_begin's flags; counted below after the first poll
        const FgPollOut po = use_graph ? FgPollOut{nullptr, 0} : fg_poll_next(&s->poll);
        if (use_graph) FG_HIP_CHECK(hipGraphLaunch(s->cg_graph_exec, st));
        else enqueue_chunk(s->prof_on && (s->prof_chunk++ % 4 == 0), po);
        if (int rc = mb_poll(s, nsys, st, done, po, true)) return rc;
        active_now = 0;
        for (int i = 0; i < nsys; ++i) active_now += s->flags_pinned[i] == 0;
        if (s->prof_used) if (int rc = prof_collect()) return rc;
        bool broke = false;
        for (int i = 0; i < nsys; ++i) broke = broke || s->flags_pinned[i] == 2;
        if (broke && recoveries < 3 && it + CG_CHUNK < q.max_iterations) {
            hipLaunchKernelGGL(k_mbs_recover, grid, blk, 0, st, n, q);
            hipLaunchKernelGGL(k_mbs_recover_flags, sg, sb, 0, st, q, nsys);
            ++recoveries;
            need_restart = true;
            done = false;
        }
        if (trace) {
            mb_real lo = 1e30f, hi = 0.f; int active = 0;
            for (int i = 0; i < nsys; ++i) { const mb_real c = s->info_pinned[i].final_residual; lo = c < lo ? c : lo; hi = c > hi ? c : hi; active += s->flags_pinned[i] == 0; }
            fprintf(stderr, "[mb_cg] it %4d residual min %.3e max %.3e active %d\n", it + CG_CHUNK, lo, hi, active);
        }
    }
    bool failed = false;
    for (int i = 0; i < nsys; ++i) failed = failed || !s->info_pinned[i].converged || s->flags_pinned[i] == 5;
    if (failed) {
        hipLaunchKernelGGL(k_mbs_restore_best, grid, blk, 0, st, n, q);
        FG_HIP_CHECK(hipMemcpyAsync(s->info_pinned, s->info_dev, sizeof(fg_solve_info) * nsys, hipMemcpyDeviceToHost, st));
        FG_HIP_CHECK(hipStreamSynchronize(st));
    }
    return mb_finish(s, nsys, nullptr, max_it);
}
