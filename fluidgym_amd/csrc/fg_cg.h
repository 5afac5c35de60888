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
