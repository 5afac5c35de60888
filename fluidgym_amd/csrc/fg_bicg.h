// Shared by the BiCGStab translation units (fg_bicgstab.hip: brick kernels, 2-D and 3-D; fg_bicgstab3d.hip: the z-marching
// two-kernel form in 3-D): pointer bundles, accumulator names of the two-kernel recurrence, the verdict bookkeeping.
#pragma once
#include "fg_internal.h"

#ifdef __HIPCC__
__device__ __forceinline__ fg_real fg_rms(double rr, int n) { return (fg_real)sqrt(rr / (double)n); }

// ok_flag: the value a finite verdict stores (1 done; 4 = converged on s, the x kernel still owes x += alpha p).  The flag is
// stored ONCE, after the info words: other workgroups of the env read it in the same launch, and a transient 1 before the 4
// would let one of them skip its half update.
__device__ __forceinline__ void fg_mark(int32_t* flags, fg_solve_info* info, int sys, fg_real crit, int it, int ok_flag = 1) {
    const bool finite = isfinite(crit);
    info[sys].final_residual = crit;
    info[sys].used_iterations = it;
    info[sys].converged = finite ? 1 : 0;
    info[sys].is_finite = finite ? 1 : 0;
    flag_st(flags + (sys), finite ? ok_flag : 2);
}
#endif

struct BicgPtrs {
    const fg_real* diag; const fg_real* off; const fg_real* rhs;
    fg_real* x; fg_real* r; fg_real* rw; fg_real* p; fg_real* v; fg_real* t;
    FgDacc* acc; fg_real* sc; int32_t* flags; fg_solve_info* info;
    int nc; fg_real tol;
    // right preconditioning (fg_linepre.hip): when set, v = C mp with mp = M^-1 p, t = C ms with ms = M^-1 s, and the iterate
    // advances along mp / ms; r, s and every dot product are those of C M^-1, so r stays the true residual of C x = rhs
    const fg_real* mp; const fg_real* ms;
};

// Accumulators of the two-kernel form (FgDacc, indexed with the parity e of the iteration that fills them): see fg_bicgstab.hip
constexpr int F_RV = 0, F_RR = 2, F_SS = 4, F_TS = 6, F_TT = 8, F_RS = 10, F_RT = 12, F_RHOE = 14;
static_assert(F_RHOE + 2 <= FG_ACC_DOUBLES, "fused BiCGStab accumulators");

struct BicgFused {
    fg_real* s; fg_real* p[2]; fg_real* v[2];   // s buffer; p / v of iteration i in p[i & 1] / v[i & 1]
    // fold0 (z-marching kernels, start vector zero): no init kernel -- r_0 = p_0 = rw = rhs, so kernel a(0) reads rhs, writes rw,
    // x = 0 and v_0 and sums r.r itself; kernel b(0) reads rhs for r_0 and kernel a(1) for p_0 (neither r nor p_0 is ever stored)
    int fold0;
};


#ifdef __HIPCC__
// Per-system decisions of the two-kernel iteration, taken by EVERY workgroup of an env from the same accumulator words (the
// leader workgroup also stores the derived scalars and resets the accumulator set nobody reads in its launch).  The logic of
// k_bicgf_a / k_bicgf_b (fg_bicgstab.hip), shared with the z-marching kernels; the results are made wave-uniform (SGPRs) so that
// the branches on them are scalar branches.
__device__ __forceinline__ float fg_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double fg_uniform(double v) {
    const long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ int fg_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct BicgDecA {
    int mode[3];   // 0 skip | 1 full update | 2 converged on s: x += alpha p only | 3 first iteration: v = C p
    fg_real alpha[3], omega[3], beta[3];
    bool restart[3];
    bool any;
};
__device__ __forceinline__ BicgDecA fg_bicgf_decide_a(const FgGrid& g, const BicgPtrs& q, int b, int it, bool leader, bool fold0 = false) {
    BicgDecA D;
    const int e = it & 1, pe = e ^ 1;
    D.any = false;
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
        D.mode[comp] = 0; D.alpha[comp] = D.omega[comp] = D.beta[comp] = 0; D.restart[comp] = false;
        if (comp >= q.nc) continue;
        const int sys = b * q.nc + comp;
        const int f = flag_ld(q.flags + (sys));
        if (f != 0 && f != 4) continue;      // (4: stored by this env's leader in THIS launch; s.s below gives the same verdict)
        FgDacc* a = q.acc + (size_t)sys * FG_ACC_DOUBLES;
        int mode = 0; fg_real alpha = 0, omega = 0, beta = 0; bool restart = false;
        if (it == 0 && fold0) {
            // r_0.r_0 is summed by this very launch: rho_0 is parked as NaN (kernel b(0) substitutes r.r, as after a restart) and
            // kernel b(0) judges the start vector
            if (leader) {
                acc_st(a + (F_RHOE + 0), (double)NAN);
                acc_st(a + (F_SS + 0), 0.0); acc_st(a + (F_TS + 0), 0.0); acc_st(a + (F_TT + 0), 0.0);
                acc_st(a + (F_RS + 0), 0.0); acc_st(a + (F_RT + 0), 0.0);
            }
            mode = 3;
        } else if (it == 0) {
            const double rr0 = acc_ld(a + (F_RR + 0));
            if (fg_rms(rr0, g.n) >= q.tol) {      // (else: the start vector already meets the tolerance, kernel b(0) marks it)
                if (leader) {
                    acc_st(a + (F_RHOE + 0), rr0);             // rho_0 = rw.r_0 = r_0.r_0
                    acc_st(a + (F_SS + 0), 0.0); acc_st(a + (F_TS + 0), 0.0); acc_st(a + (F_TT + 0), 0.0);
                    acc_st(a + (F_RS + 0), 0.0); acc_st(a + (F_RT + 0), 0.0);
                }
                mode = 3;
            }
        } else {
            const fg_real crit_s = fg_rms(acc_ld(a + (F_SS + pe)), g.n);
            alpha = sc_ld(q.sc + (sys * 2 + 0));
            if (!(crit_s >= q.tol)) {   // converged on s (bicgstab_solver_kernel.cu:305-329), or s.s not finite
                if (leader) fg_mark(q.flags, q.info, sys, crit_s, it - 1, 4);
                if (isfinite(crit_s)) mode = 2;
            } else {
                const fg_real omega_raw = (fg_real)(acc_ld(a + (F_TS + pe)) / acc_ld(a + (F_TT + pe)));
                omega = isfinite(omega_raw) ? omega_raw : (fg_real)0;
                const double rho_new = acc_ld(a + (F_RS + pe)) - (double)omega * acc_ld(a + (F_RT + pe));
                beta = (fg_real)(rho_new / acc_ld(a + (F_RHOE + pe))) * (alpha / omega);
                restart = !isfinite(beta);   // rho of the previous iteration exactly 0, or omega 0: rw = p = r, rho = r.r
                if (leader) {
                    sc_st(q.sc + (sys * 2 + 1), omega);
                    acc_st(a + (F_RHOE + e), restart ? (double)NAN : rho_new);   // NaN: kernel b takes r.r of this launch
                    acc_st(a + (F_SS + e), 0.0); acc_st(a + (F_TS + e), 0.0); acc_st(a + (F_TT + e), 0.0);
                    acc_st(a + (F_RS + e), 0.0); acc_st(a + (F_RT + e), 0.0);
                }
                mode = 1;
            }
        }
        D.mode[comp] = fg_uniform(mode);
        D.alpha[comp] = fg_uniform(alpha); D.omega[comp] = fg_uniform(omega); D.beta[comp] = fg_uniform(beta);
        D.restart[comp] = fg_uniform((int)restart) != 0;
        D.any = D.any || D.mode[comp] != 0;
    }
    return D;
}

struct BicgDecB {
    bool work[3];
    fg_real alpha[3];
    bool any;
};
__device__ __forceinline__ BicgDecB fg_bicgf_decide_b(const FgGrid& g, const BicgPtrs& q, int b, int it, bool leader) {
    BicgDecB D;
    const int e = it & 1;
    D.any = false;
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
        D.work[comp] = false; D.alpha[comp] = 0;
        if (comp >= q.nc) continue;
        const int sys = b * q.nc + comp;
        const int f = flag_ld(q.flags + (sys));
        if (f == 4) { if (leader) flag_st(q.flags + (sys), 1); continue; }   // kernel a applied x += alpha p: done
        if (f != 0) continue;
        FgDacc* a = q.acc + (size_t)sys * FG_ACC_DOUBLES;
        const double rr = acc_ld(a + (F_RR + e));
        const fg_real crit = fg_rms(rr, g.n);
        bool work = false; fg_real alpha = 0;
        if (!(crit >= q.tol)) {
            if (leader) fg_mark(q.flags, q.info, sys, crit, it == 0 ? -1 : it);
        } else {
            double rho = acc_ld(a + (F_RHOE + e));
            if (isnan(rho)) rho = rr;                       // breakdown restart decided by kernel a: rw = r, rho = r.r
            const fg_real alpha_raw = (fg_real)(rho / acc_ld(a + (F_RV + e)));
            alpha = isfinite(alpha_raw) ? alpha_raw : (fg_real)0;   // rw.v == 0 exactly: the iteration keeps its minimal-residual half
            if (leader) {
                q.info[sys].final_residual = crit;
                q.info[sys].used_iterations = it - 1;
                sc_st(q.sc + (sys * 2 + 0), alpha);
                acc_st(a + (F_RHOE + e), rho);              // (a workgroup that reads it after this store finds the same value)
                acc_st(a + (F_RV + (e ^ 1)), 0.0); acc_st(a + (F_RR + (e ^ 1)), 0.0);   // filled by kernel a(it + 1)
            }
            work = true;
        }
        D.work[comp] = fg_uniform((int)work) != 0;
        D.alpha[comp] = fg_uniform(alpha);
        D.any = D.any || D.work[comp];
    }
    return D;
}
#endif

// z-marching two-kernel iteration in 3-D (fg_bicgstab3d.hip).  fg_bicg3_ok: the grid fits the tiles (every thread valid) and fills
// the chip; zc_out = planes per z-chunk
bool fg_bicg3_ok(const fg_state* s, int nc, int* zc_out);
int fg_bicg3_launch_a(const fg_state* s, const BicgPtrs& q, const BicgFused& w, int it, int zc, int slot, hipStream_t st);
int fg_bicg3_launch_b(const fg_state* s, const BicgPtrs& q, const BicgFused& w, int it, int zc, int slot, hipStream_t st);
