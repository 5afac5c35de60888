// Slotted CG accumulators and the residual verdict shared by the CG kernels (fg_poisson.hip) and the first kernel of a
// preconditioner application (fg_fdprecond.hip / fg_fdfft.hip).
#pragma once
#include "fg_internal.h"

#define FG_CG_SLOTS 64
#define FG_CG_NAMES 8  // rr ring 0..2 | pAp ring 3..4 | r.z ring 5..7 (preconditioned CG)

#ifdef __HIPCC__
__device__ __forceinline__ FgDacc* fg_acc_ptr(FgDacc* acc, int b, int name) {
    return acc + ((size_t)b * FG_CG_NAMES + name) * FG_CG_SLOTS;
}
// total of an accumulator; every lane of the calling wave gets the result.  Each slot is an order-independent FgDacc and
// the slots are summed by a fixed shuffle tree: the total does not depend on the order in which the workgroups arrived.
__device__ __forceinline__ double fg_acc_total(const FgDacc* a, int ns) {
    if (ns == 1) return acc_ld(a + (0));
    const int lane = threadIdx.x & 63;
    double v = (lane < ns) ? acc_ld(a + (lane)) : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void fg_acc_zero(FgDacc* a, int ns) {  // called by the first wave of the leader block
    const int lane = threadIdx.x & 63;
    if (lane < ns) acc_st(a + (lane), 0.0);
}
__device__ __forceinline__ void fg_acc_add(FgDacc* a, int ns, unsigned tile, double v) {
    acc_add(a + (tile & (unsigned)(ns - 1)), v);
}
#endif

// State of the pressure CG that follows a right-hand side, prepared by the kernel that builds it (k_div) -- what a k_cg_begin launch
// does: the ns accumulator slots in use, the mean accumulator of the p -= mean(p) pass, best-iterate state, flags from the activity
// mask, info cleared.  acc == nullptr: not folded.  All threads of ONE workgroup per env call fg_cg_begin_env.
struct FgCgBegin { FgDacc* acc; int32_t* flags; fg_solve_info* info; FgDacc* mean_sums; FgBest best; int track_best, ns;
                   FgDacc* xsum; };   // xsum (optional): [B][2] sum(x) accumulators of the fused CG (fg_fftcg.hip)
int fg_cg_slots(const fg_state* s);   // accumulator slots of this grid's CG launches (fg_poisson.hip)
#ifdef __HIPCC__
__device__ __forceinline__ void fg_cg_begin_env(const FgCgBegin& q, const fg_real* __restrict__ dt, int b) {
    for (int k = threadIdx.x; k < FG_CG_NAMES * q.ns; k += blockDim.x)   // only the ns slots in use (64 B each)
        acc_st(fg_acc_ptr(q.acc, b, k / q.ns) + k % q.ns, 0.0);
    if (threadIdx.x != 0) return;
    acc_st(q.mean_sums + b, 0.0);  // accumulator of the p -= mean(p) pass that follows the solve (fg_launch_mean_sub)
    if (q.xsum) { acc_st(q.xsum + 2 * b, 0.0); acc_st(q.xsum + 2 * b + 1, 0.0); }
    q.best.best_crit[b] = q.track_best ? INFINITY : 0.f;  // 0: no residual ever beats it, nothing is kept
    q.best.saved_crit[b] = INFINITY;
    q.best.save_at[b] = -1;
    const bool active = (dt == nullptr) || (dt[b] > 0.f);
    flag_st(q.flags + (b), active ? 0 : 3);
    q.info[b].final_residual = 0.f;
    q.info[b].used_iterations = -1;
    q.info[b].converged = active ? 0 : 1;
    q.info[b].is_finite = 1;
}
#endif

// Start of a pressure CG from the zero vector, folded into the kernel that writes its right-hand side (k_div; the state was
// prepared by the kernel in front of it, k_h): r = b, x = 0 and r.r into ring entry 0 -- what k_cg_residual<use_x0 = 0> does in a
// launch of its own.  Same per-workgroup partial sums, same accumulator: the same bits.  acc == nullptr: not folded.
struct FgCgStart { FgDacc* acc; fg_real* r; fg_real* x; int ns; };

// The verdict on rr_{it+1} that a k_cg_check launch between k_cg_update(it) and the preconditioner used to give, taken instead by
// EVERY workgroup of the preconditioner's first kernel from the same accumulator words (whole waves must call: the slot sum is a
// wave shuffle tree); the env's leader thread stores the flag and the info words.  An env found converged (or non-finite) is
// skipped by that kernel and, through flags, by everything launched after it; one that is not is judged again, with the same
// numbers, by k_cg_ap(it + 1).
struct FgCgJudge {
    FgDacc* acc;              // s->cg_acc (nullptr: no verdict)
    int32_t* flags; fg_solve_info* info;
    fg_real tol; int it, n, ns;
};
#ifdef __HIPCC__
__device__ __forceinline__ bool fg_cg_judge(const FgCgJudge& j, int b, bool leader) {   // true: env b needs no further work
    const double rr = fg_acc_total(fg_acc_ptr(j.acc, b, (j.it + 1) % 3), j.ns);
    const fg_real crit = (fg_real)sqrt(rr / (double)j.n);
    if (crit >= j.tol) return false;
    if (leader) {
        const bool finite = isfinite(crit);
        j.info[b].final_residual = crit;
        j.info[b].used_iterations = j.it;
        j.info[b].converged = finite ? 1 : 0;
        j.info[b].is_finite = finite ? 1 : 0;
        flag_st(j.flags + (b), finite ? 1 : 2);
    }
    return true;
}
#endif

// Fused CG (fg_fftcg.hip): the tridiagonal kernel of the preconditioner application that follows iteration `judge.it` is the first
// kernel to see rr_{it+1} complete.  EVERY workgroup takes the verdict (fg_cg_judge's rule); the env's leader workgroup stores it,
// or -- for an env that iterates on -- does what the leader of k_cg_ap did: progress words, the best-iterate decision on
// x_{it+1} (FgBest), and the reset of the r.r ring entry the next update kernel accumulates into.
struct FgCgLead { FgCgJudge judge; FgBest best; };      // judge.acc == nullptr: none
#ifdef __HIPCC__
__device__ __forceinline__ bool fg_cg_lead(const FgCgLead& q, int b, bool lead_block) {   // true: env b needs no further work
    const FgCgJudge& j = q.judge;
    const double rr = fg_acc_total(fg_acc_ptr(j.acc, b, (j.it + 1) % 3), j.ns);
    const fg_real crit = (fg_real)sqrt(rr / (double)j.n);
    const bool done = !(crit >= j.tol);
    if (lead_block && threadIdx.x < 64) {
        if (!done) fg_acc_zero(fg_acc_ptr(j.acc, b, (j.it + 2) % 3), j.ns);
        if (threadIdx.x == 0) {
            j.info[b].final_residual = crit;
            j.info[b].used_iterations = j.it;
            if (done) {
                const bool finite = isfinite(crit);
                j.info[b].converged = finite ? 1 : 0;
                j.info[b].is_finite = finite ? 1 : 0;
                flag_st(j.flags + (b), finite ? 1 : 2);
            } else {
                fg_best_decide(q.best, b, crit, j.it + 1);
            }
        }
    }
    return done;
}
#endif
