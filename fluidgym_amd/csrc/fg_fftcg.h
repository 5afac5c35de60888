// Fused row kernels of the preconditioned pressure CG (fg_fftcg.hip) and of the Helmholtz-preconditioned BiCGStab (fg_fftbicg.hip):
// argument bundles and the host entry points the solver drivers (fg_poisson.hip, fg_bicgstab.hip) call.
#pragma once
#include "fg_internal.h"
#include "fg_cg.h"

// accumulator names of the fused CG inside the FG_CG_NAMES slots of fg_cg.h: the r.r ring keeps names 0..2 (k_cg_check and the
// verdicts read it as before), delta_k = z_k . P z_k takes the p.Ap pair, gamma_k = r_k . z_k the r.z ring
constexpr int FCG_DELTA = 3, FCG_GAMMA = 5;

struct FcgVectors {
    fg_real* z; fg_real* w;        // z_k = M^-1 r_k, w_k = P z_k
    fg_real* p; fg_real* s;        // p_k, s_k = P p_k
    fg_real* x; fg_real* r; fg_real* t1;   // iterate, residual, transformed residual (in / out of the tridiagonal kernel)
};

struct FcgUpdArgs {
    const float* z; const float* w; float* p; float* s; float* x; float* r; float* t1;
    const float2* tw; const float2* rot; float fs0, fs;
    const int32_t* flags; FgDacc* acc; double* alpha; FgDacc* xsum; FgBest best;
    int ns, rows, it, first; long env_stride;
    // first update of a solve started by k_fcg_div_fwd: x_0 = 0 and r_0 = b were never stored -- r_0 is read from the right-hand side,
    // x_0 is not read, and an env whose start vector already met the tolerance gets its zeros here
    const float* r0; int x_zero;
    const int32_t* lazy;      // envs whose first iterate met the tolerance (k_fcg_check0): only x = x_0 + alpha z is written
};
struct FcgInvArgs {
    const float* u; const float* r; const float* rA; float* z; float* w;
    const float2* tw; const float2* rot; float is0, is;
    const int32_t* flags; FgDacc* acc; int ns, rows, it; long env_stride;
    const float* hy; const float* rhy; const float* hx; const float* rhx; int fixed_x;
    // I'(0) of a solve: also d.w, w.w, d.d (d = r - w) and sum(z) (ring names FCG_GAMMA + 2, FCG_DELTA + 1, r.r ring entry 2,
    // xsum[2 b + 1] -- all idle until the second iteration), from which k_fcg_check0 knows |r_1| before any vector is updated
    FgDacc* xsum; int extras;
};

struct FcgDivArgs {
    const float* h; float* div; float* r; float* x; float* t1;
    const float* bvel[4];
    const float2* tw; const float2* rot; float fs0, fs;
    const float* dt; FgDacc* acc; int ns, rows; long n;
    const float* hx; const float* hy; int fixed_x;
};

#if !FG_F64
int fg_fcg_div_fwd(fg_state* s, const FgBounds& bnd, const fg_real* dt, const fg_real* hvec, fg_real* div, int ns, hipStream_t st);
bool fg_fcg_ok(const fg_state* s);     // the grid / preconditioner setup the fused kernels cover (and FG_CG_FUSED != 0)
int fg_fcg_update_fwd(fg_state* s, const FcgVectors& v, int it, int first, int ns, hipStream_t st, const fg_real* r0 = nullptr);
int fg_fcg_inv_apply(fg_state* s, const FcgVectors& v, const fg_real* rA, int it, int ns, hipStream_t st, int extras = 0);
// verdict on the FIRST iterate from the dot products of I'(0) (k_fcg_check0): flags / info / alpha / lazy marks
int fg_fcg_check0(fg_state* s, fg_real tol, int ns, hipStream_t st, FgPollOut poll);
#endif
