// fp64 build (libfluidgym_hip_f64.so, -DFG_REAL_DOUBLE) only: the kernels that exist in fp32 form alone -- the fast-diagonalisation
// preconditioner's MFMA basis changes and LDS FFT (its operator itself runs in doubles since round 6: fg_f64_fd.hip), the z-marching 3-D Poisson kernels, the y-line preconditioner -- are not
// instantiated for double.  The core translation units call them through these definitions: "not available", and the callers fall
// back to the generic kernels (plain CG, generic stencil kernels, unpreconditioned BiCGStab).
#include "fg_internal.h"

#if !FG_F64
#error "fg_f64_stubs.hip belongs to the fp64 build only"
#endif

bool fg_zmarch_ok(const fg_state*, int*) { return false; }
int fg_zmarch_apply(const fg_state*, const fg_real*, const fg_real*, fg_real*, int, hipStream_t) { return FG_ERR_UNSUPPORTED; }
int fg_zmarch_relax(const fg_state*, const fg_real*, const fg_real*, const fg_real*, fg_real*, fg_real, int, int, hipStream_t) { return FG_ERR_UNSUPPORTED; }
int fg_zmarch_cg_ap(const fg_state*, const fg_real*, const fg_real*, const fg_real*, fg_real*, fg_real*, FgDacc*, int32_t*, fg_solve_info*, int,
                    fg_real, int, int, int, int, int, hipStream_t) { return FG_ERR_UNSUPPORTED; }
bool fg_fd_dct_supported(int) { return false; }
int fg_fd_dct_forward(fg_state*, const fg_real*, fg_real*, hipStream_t, int, const FgCgJudge*) { return FG_ERR_UNSUPPORTED; }
int fg_fd_dct_inverse(fg_state*, const fg_real*, fg_real*, const fg_real*, FgDacc*, int, int, hipStream_t, int) { return FG_ERR_UNSUPPORTED; }
// (fg_fd_apply: fg_f64_fd.hip -- the preconditioner's operator in doubles, plain kernels; round 6)
int fg_line_alloc(fg_state*) {
    fg_set_error("the y-line preconditioner is an fp32 kernel family: not part of the fp64 build");
    return FG_ERR_UNSUPPORTED;
}
int fg_helm_alloc(fg_state*) { return FG_ERR_UNSUPPORTED; }
int fg_ilu_alloc(fg_state*) { fg_set_error("the ILU(0) preconditioner is not part of the fp64 build"); return FG_ERR_UNSUPPORTED; }
int fg_ilu_factor(fg_state*, const fg_real*, const fg_real*, hipStream_t) { return FG_ERR_UNSUPPORTED; }
int fg_ilu_apply(fg_state*, const fg_real*, const fg_real*, int, const fg_real*, fg_real*, hipStream_t) { return FG_ERR_UNSUPPORTED; }
int fg_helm_factor(fg_state*, const fg_real*, fg_real, int, int, int, hipStream_t, int) { return FG_ERR_UNSUPPORTED; }
int fg_fd_helmholtz_apply(fg_state*, int, const fg_real*, fg_real*, hipStream_t) { return FG_ERR_UNSUPPORTED; }
int fg_line_factor(fg_state*, const fg_real*, const fg_real*, int, hipStream_t) { return FG_ERR_UNSUPPORTED; }
int fg_line_apply(fg_state*, const fg_real*, const fg_real*, int, const fg_real*, fg_real*, hipStream_t) { return FG_ERR_UNSUPPORTED; }

extern "C" int fg_set_fd_fast_transform(fg_handle, int, float) {
    fg_set_error("the fast cosine transform belongs to the fp32 fast-diagonalisation preconditioner: not part of the fp64 build");
    return FG_ERR_UNSUPPORTED;
}
extern "C" int fg_set_fd_helmholtz(fg_handle, const float*) {
    fg_set_error("the Helmholtz preconditioner belongs to the fp32 fast-diagonalisation kernels: not part of the fp64 build");
    return FG_ERR_UNSUPPORTED;
}
