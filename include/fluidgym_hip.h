/*
 * fluidgym_hip.h -- C ABI of libfluidgym_hip.so, the MI355X (gfx950) native replacement for the
 * simulation hot path of safe-autonomous-systems/fluidgym (module `PISOtorch`,
 * reference: src/fluidgym/simulation/extensions/PISOtorch.cpp:40-670).
 *
 * Every entry point is `extern "C"`, takes plain pointers / sizes / scalars (no torch types),
 * returns an int status (FG_OK = 0, negative = error, never aborts -- the reference `exit(10)`s on a
 * CUDA error, PISO_multiblock_cuda_kernel.cu:40-46) and is asynchronous on the `stream` it is
 * given (a hipStream_t passed as void*; the reference synchronises the device around every kernel,
 * PISO_multiblock_cuda_kernel.cu:4512,4520).  All `float*` arguments are DEVICE pointers unless
 * the name ends in `_host`.
 *
 * Differences from the reference object model:
 *   - an ENV BATCH axis B is the outermost axis of every field (the reference asserts N == 1,
 *     domain_structs.cpp:2020); per-env time steps are a device array dt[B]; dt[b] <= 0 marks env b
 *     inactive for that call (its state is left untouched);
 *   - one block, rectilinear (tensor-product) orthogonal grid: metrics are per-axis cell widths;
 *     faces are PERIODIC or FIXED (Dirichlet velocity; Dirichlet/Neumann passive scalar);
 *   - matrices are never materialised as CSR: C is kept in stencil form (diag + 2d off-diagonals),
 *     P is applied matrix-free from 1/A;
 *   - the caller owns all field memory and binds it with fg_bind(); the handle owns solver
 *     workspace allocated once in fg_create() -- nothing is allocated on the step path.
 *
 * Layouts (fp32, C-contiguous): velocity [B,d,(Z,)Y,X]; pressure [B,1,(Z,)Y,X];
 * passive scalar [B,C,(Z,)Y,X]; velocity source [B,d,(Z,)Y,X]; boundary velocity of face f
 * [B,d,slab(f)] and boundary scalar [B,C,slab(f)] where slab(f) is the cell shape with extent 1
 * on the face axis.  Faces: 0..5 = -x,+x,-y,+y,-z,+z (PISO_multiblock_cuda_kernel.cu:211-236).
 */
#ifndef FLUIDGYM_HIP_H
#define FLUIDGYM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FG_ABI_VERSION 1

/* Scalar type of the fields, metrics and real-valued arguments of the SINGLE-BLOCK entry points (fg_create .. fg_poisson_fdcg).
 * libfluidgym_hip.so is built with fg_real = float (the reference's default dtype, envs/fluid_env.py:146);
 * libfluidgym_hip_f64.so is the same sources built with -DFG_REAL_DOUBLE: fg_real = double, for FluidEnv(dtype=torch.float64)
 * (assembly, BiCGStab, CG and the step drivers; the fp32-only fast-diagonalisation / z-marching / line-preconditioner kernels and
 * the multi-block path are not part of that build and their entry points return FG_ERR_UNSUPPORTED or are absent). */
#ifdef FG_REAL_DOUBLE
typedef double fg_real;
#else
typedef float fg_real;
#endif
#define FG_MAX_SCALARS 4

/* status codes */
#define FG_OK 0
#define FG_ERR_INVALID_ARG (-1)
#define FG_ERR_NOT_BOUND (-2)
#define FG_ERR_HIP (-3)
#define FG_ERR_UNSUPPORTED (-4)
#define FG_ERR_NOT_CONVERGED (-5) /* informational: a solve hit max_iterations */
#define FG_ERR_NOT_FINITE (-6)   /* a solver residual became NaN/Inf */
#define FG_ERR_FLUX_BALANCE (-7) /* |sum of boundary fluxes| > flux_balance_tol (simulation.py:223-231) */

/* boundary type of a face (reference BoundaryType::PERIODIC / FIXED, domain_structs_gpu.h:120-135) */
#define FG_PERIODIC 0
#define FG_FIXED 1
/* passive-scalar boundary condition (reference BoundaryConditionType) */
#define FG_DIRICHLET 0
#define FG_NEUMANN 1

/* bindable fields (fg_bind) */
enum fg_field {
    FG_VELOCITY = 0,       /* block.velocity           [B,d,N]                      */
    FG_PRESSURE = 1,       /* block.pressure           [B,N]                        */
    FG_SCALAR = 2,         /* block.passiveScalar      [B,C,N]                      */
    FG_VELOCITY_SOURCE = 3,/* block.velocitySource     [B,d,N] (NULL = none)        */
    FG_VISCOSITY_FIELD = 4,/* block.viscosity          [B,N] per-cell viscosity of the velocity system (NULL = the global one): Block.setViscosity
                              of the reference's SGS hook (tcf_env.py:441-474; getViscosityBlock, PISO_multiblock_cuda_kernel.cu:1816-1837) */
    FG_BOUND_VELOCITY = 8, /* + face: FixedBoundary.velocity      [B,d,slab]        */
    FG_BOUND_SCALAR = 16   /* + face: FixedBoundary.passiveScalar [B,C,slab]        */
};

/* pressure solver variants (the reference only has CG, cg_solver_kernel.cu:129-471) */
#define FG_SOLVER_CG 0
#define FG_SOLVER_JACOBI 1
#define FG_SOLVER_RBGS 2
#define FG_SOLVER_MGCG 3   /* reserved */
#define FG_SOLVER_FDCG 4   /* CG preconditioned by the separable constant-coefficient operator (fast diagonalisation) */

typedef struct fg_state* fg_handle;

typedef struct fg_config {
    int32_t dims;                 /* 2 or 3 */
    int32_t nx, ny, nz;           /* cells; nz = 1 in 2-D; >= 3 per used axis (domain_structs.cpp:1193) */
    int32_t batch;                /* B */
    int32_t n_scalars;            /* passive scalar channels C (0..FG_MAX_SCALARS) */
    int32_t face_type[6];         /* FG_PERIODIC / FG_FIXED; a FIXED face needs a FIXED partner */
    int32_t scalar_bc[6][FG_MAX_SCALARS]; /* FG_DIRICHLET / FG_NEUMANN per FIXED face & channel */
    int32_t device;               /* HIP device ordinal */
} fg_config;

/* result of a batched linear solve, one entry per system (reference LinearSolverResultInfo,
 * bicgstab_solver.h) */
typedef struct fg_solve_info {
    float final_residual;   /* RMS residual ||r||_2/sqrt(n) (cg_solver_kernel.cu:100-106) */
    int32_t used_iterations;
    int32_t converged;
    int32_t is_finite;
} fg_solve_info;

/* ---- lifetime ------------------------------------------------------------------------------ */
int fg_abi_version(void);
const char* fg_last_error(void);                 /* thread-local message of the last failure */

/* Domain()+CreateBlock()+PrepareSolve() (PISOtorch.cpp:420-500; domain_structs.cpp:2570-2693).
 * hx/hy/hz_host: per-axis cell widths (host arrays of nx/ny/nz floats; hz may be NULL in 2-D). */
int fg_create(const fg_config* cfg, const fg_real* hx_host, const fg_real* hy_host, const fg_real* hz_host,
              fg_handle* out);
int fg_destroy(fg_handle h);

/* Block.setVelocity()/setPressure()/... + Domain.UpdateDomainData() (domain_structs.cpp:3047-3283):
 * bind (borrow) a caller-owned device buffer.  field = enum fg_field (+ face for boundaries). */
int fg_bind(fg_handle h, int field, fg_real* ptr);
/* Domain.viscosity / Domain.setScalarViscosity (domain_structs.cpp:3070) */
int fg_set_viscosity(fg_handle h, fg_real viscosity);
int fg_set_scalar_viscosity(fg_handle h, int channel, fg_real viscosity);

/* Fast-diagonalisation preconditioner factors for FG_SOLVER_FDCG (host arrays, copied to the device):
 * Qx [nx,nx] / QxT: H-orthonormal eigenbasis of the 1-D x operator and its transpose, Qz / QzT the same
 * for z (NULL in 2-D), lower [ny]: sub-diagonal of the y operator, inv / cp [(nz,) ny, nx]: per-mode
 * reciprocal pivots and modified super-diagonal of the tridiagonal LU along y.  Built by
 * fluidgym_amd/simulation/fd_precond.py; requires FIXED y faces.  No reference counterpart (the
 * reference runs un-preconditioned CG, cg_solver_kernel.cu:129-471). */
int fg_set_fd_preconditioner(fg_handle h, const float* Qx_host, const float* QxT_host, const float* Qz_host,
                             const float* QzT_host, const float* lower_host, const float* inv_host,
                             const float* cp_host);
/* Marks the x axis as a cosine-transform axis: uniform cell width `cell_width`, FIXED faces, nx in {64, 128, 256, 512}.
 * The caller must have passed the orthonormal DCT-II basis / sqrt(cell_width) (modes in DCT order) as Qx to
 * fg_set_fd_preconditioner; the library then applies it as one FFT per row (csrc/fg_fdfft.hip) instead of a dense
 * GEMM.  Returns FG_ERR_UNSUPPORTED for other lengths / axes (the GEMM path stays in place). */
int fg_set_fd_fast_transform(fg_handle h, int axis, float cell_width);
/* returnBestResult of the pressure CG (SolveLinear(..., returnBestResult), cg_solver_kernel.cu:345-361): on (default), a
 * solve that ends unconverged hands back the best iterate it kept (within 2x of the lowest residual reached) instead
 * of the last one; off saves the occasional extra store pass over x. */
int fg_set_return_best(fg_handle h, int on);
/* residualResetSteps of the pressure CG (SolveLinear(..., residualResetSteps), cg_solver_kernel.cu:281-302): every `steps`
 * iterations the residual is recomputed from the iterate, r = b - P x, and the recurrence restarts from it.  Default 100, the value
 * the reference's non-orthogonal branch passes (PISOtorch_simulation.py:1913); 0 = never. */
int fg_set_cg_reset_steps(fg_handle h, int steps);
/* Start vector of the velocity (advection) solve.  The reference's split step has two rules (recorded from its own Python in
 * tests/golden/reference_split_step.json): its orthogonal branch starts from velocityResult (advect_use_prev_result,
 * PISOtorch_simulation.py:1689-1693), its non-orthogonal branch -- which its TCF env also runs on a rectilinear grid
 * (tcf_env.py:497) -- from zero on the first non-orthogonal pass (x=None, :1735-1742).  from_result 1 (default of a new handle): the
 * orthogonal-branch rule; 0: zero.  Same converged answer within the tolerance, different iteration counts. */
int fg_set_advection_start(fg_handle h, int from_result);
/* Native form of the turbulent-channel env's PRE hook (tcf_env.py "dynamic forcing", grid.py:147-176): before every PISO step a
 * uniform body force along `axis`, per env  G = 1/2 (coef_lo <u_axis>_lo + coef_hi <u_axis>_hi), the means taken over the cell
 * layers next to the -y / +y walls (coef = nu / wall distance of the layer: the two wall shear stresses), is added to the velocity
 * right-hand side as a velocity source of that value would be.  axis < 0 switches it off.  Needs FIXED y faces. */
int fg_set_wall_stress_forcing(fg_handle h, int axis, fg_real coef_lo, fg_real coef_hi);

/* ---- reductions used by the drivers --------------------------------------------------------- */
/* Domain.getMaxVelocity(withBounds=True, computational=True) (domain_structs.cpp:1580-1611) */
int fg_max_velocity(fg_handle h, fg_real* out_B, void* stream);
/* Domain.GetBoundaryFluxBalance (domain_structs.cpp:2476-2509) */
int fg_boundary_flux_balance(fg_handle h, fg_real* out_B, void* stream);

/* Both reductions with ONE device->host transfer and one stream sync: out_host[0..B) = flux balance,
 * out_host[B..2B) = max velocity (what Simulation.single_step + _PISO_adaptive_step read per substep,
 * simulation.py:223-231 and PISOtorch_simulation.py:2013-2014). */
int fg_step_diagnostics(fg_handle h, fg_real* out_host_2B, void* stream);
/* update_advective_boundaries for one FIXED face with characteristic velocity velm (host, d floats)
 * (PISOtorch_simulation.py:282-389); envs with dt_B[b] <= 0 are skipped. */
int fg_update_advective_boundary(fg_handle h, int face, const fg_real* velm_host, const fg_real* dt_B, void* stream);
/* balance_boundary_fluxes (PISOtorch_simulation.py:188-224): free_face_mask bit f = face f is free. */
int fg_balance_boundary_fluxes(fg_handle h, int free_face_mask, fg_real atol, const fg_real* dt_B, void* stream);

/* ---- PISO building blocks (one call = the reference free function of the same role) --------- */
/* SetupAdvectionMatrix (PISO_multiblock_cuda_kernel.cu:4525-4546, kernel :3616-3880) fused with
 * SetupAdvectionVelocity (:4692-4708, kernel :4296-4400) or, when for_scalar != 0, with
 * SetupAdvectionScalar (:4620-4637, kernel :4094-4198) for `channel`. */
int fg_setup_advection(fg_handle h, const fg_real* dt_B, int for_scalar, int channel, void* stream);
/* SolveLinear(C, RHS, x, useBiCG=True) (:7085-7118; bicgstab_solver_kernel.cu:63-411) for the
 * velocity components (for_scalar = 0; x0 = previous velocityResult) or a scalar channel.
 * info_host: d (or 1) * B entries, written after an internal stream sync. */
int fg_solve_advection(fg_handle h, int for_scalar, int channel, fg_real tol, int max_iterations,
                       fg_solve_info* info_host, void* stream);
/* Preconditioner policy of the advection-diffusion solves (scalar and velocity), the reference's preconditionBiCG /
 * BiCG_precondition_fallback (PISOtorch_simulation.py:503, 565; PISOtorch_diff.py:449-476; cuSPARSE ILU(0),
 * bicgstab_solver_kernel.cu:191-226, 288-293): mode 0 = plain BiCGStab (the reference's first rung, default), 1 = every solve
 * right-preconditioned, 2 = a solve that ends unconverged or non-finite is repeated from zero with the preconditioner.  The
 * preconditioner of modes 1 / 2 is the tridiagonal part of the matrix along y (csrc/fg_linepre.hip), factorised per solve and env;
 * 3 = every solve preconditioned by the separable Helmholtz operator (fg_set_fd_helmholtz); 4 / 5 = like 1 / 2 with the reference's
 * own preconditioner, ILU(0) of the matrix (csrc/fg_ilu0.hip: closed form on the stencil, hyperplane sweeps; every axis >= 4 cells).
 * fg_advection_retries: number of repeated solves since the last reset. */
int fg_set_advection_preconditioner(fg_handle h, int mode);
/* SGSviscosityIncompressibleSmagorinsky (PISO_multiblock_cuda_kernel.cu:6913-6966): out[B,N] = coefficient * Delta^2 * |S| with
 * |S| = sqrt(2 S:S) from the gradients of the bound velocity (getBlockDataGradient, :2997-3040: central differences, a Dirichlet face
 * counts as half a cell) and Delta^2 = the largest squared cell extent.  Asynchronous on `stream`. */
int fg_sgs_smagorinsky(fg_handle h, fg_real coefficient, fg_real* out_BN, void* stream);
/* Test / diagnosis entry, never on a step path: z = M^-1 r [B, nc, N] with the preconditioner of `mode` (1: y-line, 4: ILU(0)) built from
 * the advection-diffusion matrix currently assembled (fg_setup_advection).  Synchronises. */
int fg_debug_apply_preconditioner(fg_handle h, int mode, int nc, const fg_real* r, fg_real* z, void* stream);
/* mode 3 of fg_set_advection_preconditioner: every advection-diffusion solve right-preconditioned by the separable Helmholtz
 * operator I/dt - nu Laplacian (the matrix without its advective part), inverted by fast diagonalisation: basis change along the
 * PERIODIC, uniform transform axes (x, z; the eigenvectors of fg_set_fd_preconditioner), one tridiagonal solve along y per mode
 * and env.  lam_host [nz][nx]: sum of the transform axes' eigenvalues per mode (fluidgym_amd/simulation/fd_precond.py).  No
 * reference counterpart (its preconditioner for these solves is ILU(0), off by default); fp32 library only. */
int fg_set_fd_helmholtz(fg_handle h, const float* lam_host);
int fg_advection_retries(fg_handle h, int64_t* out, int32_t reset);
/* The reference's retry ladder of a linear solve (_linear_solve_wrapper, pict/PISOtorch_diff.py:410-476) on the single-block path.
 * fg_set_double_fallback(on): `solver_double_fallback` -- a solve that failed in fp32 (advection-diffusion BiCGStab: not converged;
 * pressure CG, which runs with returnBestResult: non-finite) is repeated in fp64 on the same fp32 matrix and right-hand side from a
 * cleared result (csrMat.toType(dp), rhs.to(dp)), BEFORE the preconditioned rung of fg_set_advection_preconditioner (modes 2 / 5).
 * fg_ladder: out4 = how often each rung ran since fg_create {advection fp64, advection preconditioned, pressure fp64, 0};
 * force_mask >= 0 (tests) makes first attempts count as failed: 1 advection, 2 pressure, 4 also the advection fp64 rung; -1 = only read. */
int fg_set_double_fallback(fg_handle h, int on);
/* fg_set_advection_jacobi(on): the velocity systems of uniform 2-D grids with walls in y (the channel family) are solved by point-Jacobi
 * sweeps, several per pass over the field with the tile kept on chip (csrc/fg_jacobi.hip), instead of the reference's BiCGStab
 * (bicgstab_solver_kernel.cu, called from PISOtorch_simulation.py:1735-1742) -- same system, same criterion (RMS residual < tol), another
 * iteration; a solve the sweeps do not settle goes to BiCGStab from a cleared start vector.  Off until called (the Python Simulation
 * calls it, policy `advection_jacobi`); FG_ADV_JACOBI=0/1 in the environment overrides.  fg_advection_jacobi_counts: solves settled
 * by the sweeps, solves handed on to BiCGStab. */
int fg_set_advection_jacobi(fg_handle h, int on);
int fg_advection_jacobi_counts(fg_handle h, int64_t* out2);
int fg_ladder(fg_handle h, int64_t* out4, int32_t force_mask);
/* Which kernels the NEXT un-preconditioned advection-diffusion solve of `nc` right-hand sides will run (tests, bench reports):
 * 0 = five kernels per BiCGStab iteration, 1 = two brick kernels (csrc/fg_bicgstab.hip k_bicgf_a / _b), 2 = two z-marching
 * LDS-ring kernels (csrc/fg_bicgstab3d.hip: 3-D grids that fit the tiles and fill the chip).  Replaces nothing in the reference
 * (bicgstab_solver_kernel.cu:63-411 has one form). */
int fg_advection_solver_form(fg_handle h, int nc, int32_t* out);
/* CopyScalarResultToBlocks (:6558-6746) */
int fg_copy_scalar_result_to_blocks(fg_handle h, int channel, void* stream);
/* SetupPressureMatrix (:5599-5615, kernel :4812-4978): rA = 1/A */
int fg_setup_pressure_matrix(fg_handle h, void* stream);
/* SetupPressureRHS (:5655-5672, kernels :5136-5255 + :5389-5434): h = H(u~), b = div h */
int fg_setup_pressure_rhs(fg_handle h, const fg_real* dt_B, void* stream);
/* SolveLinear(P, div, x, useBiCG=False) + `p -= mean(p)` (PISOtorch_simulation.py:1804-1821) +
 * CopyPressureResultToBlocks.  method = FG_SOLVER_*.  info_host: B entries. */
int fg_solve_pressure(fg_handle h, int method, fg_real tol, int max_iterations, int use_previous,
                      fg_solve_info* info_host, void* stream);
/* Opt-in accuracy mode of the pressure solves of this handle (round 6): mixed-precision iterative refinement -- the iterate is kept
 * in fp64, r = b - P x is formed in fp64 with the fp32 matrix entries promoted (what the reference's fp64 fallback does with its CSR
 * values, PISOtorch_diff.py:418-445), each correction P d = r / |r| is solved by the fp32 solver to `inner_relative_tol`, at most
 * `max_corrections` times or until RMS(r) < target_tol.  0 corrections = off (default).  fp32 library only. */
int fg_set_pressure_refinement(fg_handle h, int32_t max_corrections, fg_real target_tol, fg_real inner_relative_tol);
/* CorrectVelocity(version=1) (:6220-6236, kernel :5962-5995 + :816-849) */
int fg_correct_velocity(fg_handle h, void* stream);
/* CopyVelocityResultToBlocks / FromBlocks (:6558-6746) */
int fg_copy_velocity_result_to_blocks(fg_handle h, void* stream);
int fg_copy_velocity_result_from_blocks(fg_handle h, void* stream);

/* ---- fused driver: one _PISO_split_step without Python hooks --------------------------------
 * (PISOtorch_simulation.py:1431-2002, orthogonal branch).  buoyancy_axis >= 0 fuses the RBC
 * PRE_VELOCITY_SETUP hook: velocitySource[axis] = buoyancy_factor * T (rbc_env_base.py:285-297).
 * stats_host (optional, 4 ints): max iterations of {scalar, velocity, pressure0, pressure1}. */
typedef struct fg_step_options {
    int32_t corrector_steps;      /* 2 */
    int32_t advect_scalar;        /* solve passive scalars first */
    int32_t pressure_method;      /* FG_SOLVER_* */
    int32_t max_iterations;       /* 5000 (PISOtorch_simulation.py:564) */
    fg_real advection_tol;          /* RMS residual, 1e-5 default (PISOtorch_diff.py:247-253) */
    fg_real pressure_tol;
    int32_t buoyancy_axis;        /* -1 = none */
    fg_real buoyancy_factor;
    int32_t pressure_warm_start;  /* 1: start each pressure solve from the previous pressureResult (the
                                     reference passes x=None in its orthogonal branch and pressureResult
                                     in its non-orthogonal branch, PISOtorch_simulation.py:1804-1812 vs
                                     :1878-1882; the converged answer is the same) */
} fg_step_options;
int fg_piso_step(fg_handle h, const fg_real* dt_B, const fg_step_options* opt, int32_t* stats_host,
                 void* stream);
/* Iteration statistics of the linear solves since the last reset (host bookkeeping of the LinearSolverResultInfo every
 * SolveLinear call returns, bicgstab_solver.h): out13 = sum of iterations [4] | systems solved [4] | max iterations [4] |
 * PISO steps, kinds = {passive scalar, velocity, pressure corrector 0, pressure corrector 1}; a system = env x component.
 * reset != 0 clears them after the read. */
int fg_solver_counters(fg_handle h, int64_t* out13_host, int32_t reset);
/* per kind, the systems whose solve ended WITHOUT meeting its tolerance since the last reset of the counters (iteration cap
 * reached, best iterate returned: LinearSolverResultInfo.converged == false); cleared together with fg_solver_counters */
int fg_solver_unconverged(fg_handle h, int64_t* out4_host);
/* Simulation.single_step entirely on the native side (simulation.py:206-280 + _PISO_adaptive_step,
 * PISOtorch_simulation.py:2004-2064): flux-balance guard, per-env adaptive substeps
 * ts = t_rem / ceil(t_rem / (CFL / max_vel)) recomputed before every substep, the advective-outflow PRE
 * hook (update_advective_boundaries + balance_boundary_fluxes, :228-393) and fg_piso_step per substep.
 * One device->host read per substep (flux balance + max velocity), as in the reference, but no
 * interpreter in the loop.  out_host: [0..3] max solver iterations {scalar, velocity, pressure0,
 * pressure1} of the last substep, [4] substeps taken, [5] 1 if every solve converged;
 * flux_balance_host (optional, B floats) receives the guard values. */
typedef struct fg_sim_options {
    fg_step_options step;
    fg_real time_step;          /* physical time advanced per call */
    fg_real cfl;                /* adaptive_CFL */
    int32_t adaptive;         /* 1: substeps == -1 ("ADAPTIVE"); 0: `substeps` fixed steps of time_step */
    int32_t substeps;
    fg_real flux_balance_tol;   /* 1e-5 */
    int32_t outflow_mask;     /* bit f: FIXED face f is an advective outflow (0 = none) */
    fg_real outflow_velm[3];    /* characteristic velocity u_m */
    fg_real outflow_tol;        /* flux re-balancing triggers above 0.01 * outflow_tol */
    int32_t max_substeps;     /* safety cap (reference warns above 1000) */
} fg_sim_options;
int fg_single_step(fg_handle h, const fg_sim_options* opt, int32_t* out_host_6, fg_real* flux_balance_host,
                   void* stream);
/* fg_multi_step: `n` calls of fg_single_step in one C call -- the sim steps of one env step (FluidEnv.step runs
 * int(step_length / dt) of them, fluid_env.py:840-842) without a return to the interpreter in between.  bvel_schedule (optional,
 * n x 6 pointers, step-major): boundary velocity tensor bound to face f before sim step k, NULL = keep what is bound (the
 * smoothed control of the jets changes every sim step, cylinder_env_base.py:748-753).  out_host_6n receives the six words of
 * every step; steps_done the number of steps completed (n unless a step returned an error, which is passed on). */
int fg_multi_step(fg_handle h, const fg_sim_options* opt, int32_t n, const fg_real* const* bvel_schedule, int32_t* out_host_6n,
                  fg_real* flux_balance_host, int32_t* steps_done, void* stream);
/* make_divergence_free (PISOtorch_simulation.py:1320-1429) */
int fg_make_divergence_free(fg_handle h, fg_real tol, int max_iterations, fg_solve_info* info_host,
                            void* stream);

/* ---- access to solver vectors (tests / fixtures) -------------------------------------------- */
enum fg_buffer {
    FG_BUF_A = 0,          /* Adiag [B,N]                               */
    FG_BUF_C_OFF = 1,      /* C off-diagonals [B,2d,N] (face order)     */
    FG_BUF_ADV_RHS = 2,    /* velocityRHS [B,d,N] / scalarRHS [B,N]     */
    FG_BUF_VEL_RESULT = 3, /* velocityResult [B,d,N]                    */
    FG_BUF_H = 4,          /* pressureRHS (h) [B,d,N]                   */
    FG_BUF_DIV = 5,        /* pressureRHSdiv [B,N]                      */
    FG_BUF_P_RESULT = 6,   /* pressureResult [B,N]                      */
    FG_BUF_SCALAR_RESULT = 7 /* scalarResult [B,N] (one channel)        */
};
/* velocityResult := block velocity, pressureResult := 0 (call after re-initialising the fields) */
int fg_reset_solver_state(fg_handle h, void* stream);
/* The single-block handle's counterpart of fg_mb_solver_hints: per solve kind (4) the sweeps the last Jacobi solve needed, the
 * solves still to skip after a failure, the failures in a row (csrc/fg_jacobi.hip) -- 12 words that decide which iteration runs.
 * Domain.Clone / Restore carry them (reference: envs/fluid_env.py:1320-1363); fg_reset_solver_state clears them.  set = 0 reads. */
int fg_solver_hints(fg_handle h, int32_t* hints12, int32_t set);
int fg_get_buffer(fg_handle h, int which, fg_real** out_ptr, int64_t* out_count);
/* device-to-device copy of a solver vector into a caller buffer of fg_get_buffer's count */
int fg_read_buffer(fg_handle h, int which, fg_real* dst, void* stream);

/* ---- standalone pressure-Poisson kernels on caller arrays (micro-benchmark / tests) ----------
 * Operator: (P x)_c = sum_f off_f (x_N - x_c), off_f = (alpha_P rA_P + alpha_N rA_N)/2, no entry at
 * FIXED faces (PISO_multiblock_cuda_kernel.cu:4842-4889).  All arrays [B,N]. */
int fg_poisson_apply(fg_handle h, const fg_real* rA, const fg_real* x, fg_real* y, void* stream);
/* n_sweeps damped-Jacobi / red-black Gauss-Seidel sweeps on P x = b (x updated in place; Jacobi
 * uses the handle's scratch vector).  omega = relaxation factor. */
int fg_poisson_jacobi(fg_handle h, const fg_real* rA, const fg_real* b, fg_real* x, int n_sweeps, fg_real omega,
                      void* stream);
int fg_poisson_rbgs(fg_handle h, const fg_real* rA, const fg_real* b, fg_real* x, int n_sweeps, fg_real omega,
                    void* stream);
/* n_iterations of CG without convergence polling (timing) -- or a full solve when tol > 0. */
int fg_poisson_cg(fg_handle h, const fg_real* rA, const fg_real* b, fg_real* x, fg_real tol, int max_iterations,
                  int use_x0, fg_solve_info* info_host, void* stream);
/* same with the fast-diagonalisation preconditioner (FG_SOLVER_FDCG) */
int fg_poisson_fdcg(fg_handle h, const fg_real* rA, const fg_real* b, fg_real* x, fg_real tol, int max_iterations,
                    int use_x0, fg_solve_info* info_host, void* stream);

/* ---- live kernel timing for bench.py's roofline -----------------------------------------------
 * When enabled, every FG_PROF_PERIOD-th launch (default 32) of each solver kernel kind is issued with a
 * start/stop event pair on the solve's stream (kernel-accurate timestamps), and the systems still
 * iterating in that launch are counted on the device.  Kinds are 0 .. fg_profile_kinds()-1, named by
 * fg_profile_kind_name (k_cg_ap, k_cg_update, k_bicg_p/v/s/t/x, k_gemm_f32, k_gemm_sk, k_tridiag_y).
 * fg_profile_read returns for one kind: summed milliseconds and count of the sampled launches that did
 * work, their summed ALGORITHMIC bytes and flops (active systems x per-system figure),
 * the same restricted to launches in which every system was active, and the total launches of that kind
 * (sampled or not) since fg_profile_enable, and milliseconds / count over ALL sampled launches including
 * those that found every system converged (the figure rocprofv3 --stats averages).  Any output pointer
 * may be NULL.  Synchronises the device. */
/* STREAM triad a = b + scalar * c over n floats (n % 4 == 0), `reps` launches timed with events on `stream`: the measured
 * practical HBM roof beside the spec figure (bytes per launch = 12 n).  Synchronises. */
int fg_stream_triad(float* a, const float* b, const float* c, float scalar, int64_t n, int32_t reps, float* ms_per_launch, void* stream);
/* Litmus for the access pattern of the multi-kernel Krylov recurrences: `iterations` x five launches over
 * `nsys` systems of `cells` cells; sums accumulated with device-scope atomics are read by the following kernel and zeroed by a
 * leader workgroup; atomic_access = 10 x store + load with load 0 plain / 1 agent-scope atomic load and store 0 plain / 1
 * agent-scope atomic store / 2 atomic exchange (0 = the round-1 pattern).  bad_reads[12] counts, per slot of the record, reads that did not return the full sum
 * ([11]: reads of a flag word stored by the leader one to four kernels earlier that did not return it), bad_value[12] keeps the
 * first wrong value.  Synchronises. */
int fg_coherence_litmus(int32_t atomic_access, int32_t nsys, int32_t cells, int32_t iterations, int64_t* bad_reads, double* bad_value,
                        void* stream);
/* Order-independent reduction accumulator of the Krylov solvers (csrc/fg_internal.h FgDacc: every contribution split exactly into
 * four 42-bit fixed-point words added with integer atomics, so a dot product does not depend on the arrival order of the
 * workgroups -- what the reference gets from cuBLAS dots in a fixed order, cg_solver_kernel.cu:277,317).  fg_dacc_host_sum
 * evaluates plain + sum(values) with the very split / read-back code the kernels run (no GPU needed); fg_dacc_device_sum does the
 * same sum `reps` times on the device, one atomic contribution per thread with a different launch shape each time.  A value
 * that is NaN, Inf or >= 2^75 in magnitude makes the sum NaN. */
int fg_dacc_host_sum(const double* values, int64_t n, double plain, double* out_sum);
int fg_dacc_device_sum(const double* values_host, int64_t n, double plain, int32_t reps, double* out_sums_host, void* stream);
int fg_profile_enable(fg_handle h, int on);
int fg_profile_kinds(void);
const char* fg_profile_kind_name(int kind);
int fg_profile_read(fg_handle h, int kind, double* ms_sum, int64_t* samples, double* bytes_sum, double* flops_sum,
                    double* full_ms_sum, double* full_bytes_sum, int64_t* full_samples, int64_t* launches,
                    double* all_ms_sum, int64_t* all_samples);

/* ---- observation resampling -------------------------------------------------------
 * SampleTransformedGridLocalToGlobalMulti + _FillEmptyCells (extensions/resampling.cu:191-609; Python entry
 * sample_multi_coords_to_uniform_grid, pict/data/resample.py:254-358) for ONE rectilinear block, as a gather.
 * The caller supplies, per axis x, y(, z), the continuous output index of every source cell centre split into
 * floor (base) and fraction (frac), concatenated over the axes (lengths n_src[0] + n_src[1] (+ n_src[2])); the
 * index map must be non-decreasing along each axis (it is for the reference's AABB transforms on a rectilinear
 * block).  quirk3d != 0 reproduces the compiled kernel's 6-of-8 corner loop in 3-D (resampling.cu:320), 0 is the
 * full multilinear splat of the reference's pure-torch implementation (resample.py:361-548).
 * fg_resample: src [batch, channels, (nz,) ny, nx] -> dst [batch, channels, (oz,) oy, ox], both fp32 device
 * pointers, channels <= 8; fill_max_steps as fillMaxSteps (resampling.cu:242-290). */
/* Multi-block resampling = a fixed sparse operator per mesh (the folded splat + normalise + hole-fill chain of
 * SampleTransformedGridLocalToGlobalMulti + _FillEmptyCells, resampling.cu:191-609, built on the host once).  y[m][r] = sum_k w * x[m][idx]:
 * ELL rows of equal length K (sensor pixels: the observation of every env step) or CSR (whole render grid).  x: [m][n] device,
 * y: [m][rows] device; index / weight arrays on the device. */
int fg_sparse_apply_ell(const int32_t* idx, const float* w, int32_t rows, int32_t K, const float* x, int64_t n, int32_t m, float* y,
                        void* stream);
int fg_sparse_apply_csr(const int32_t* indptr, const int32_t* col, const float* val, int32_t rows, const float* x, int64_t n, int32_t m,
                        float* y, void* stream);
typedef struct fg_resampler_state* fg_resampler;
int fg_resampler_create(int dims, const int32_t* n_src, const int32_t* n_out, const int32_t* base_cat,
                        const float* frac_cat, int quirk3d, int device, fg_resampler* out);
int fg_resampler_destroy(fg_resampler r);
int fg_resample(fg_resampler r, const float* src, int batch, int channels, float* dst, int fill_max_steps,
                void* stream);

/* ---- env glue either side of the n sim steps of an env step (fp32 library only) ------------------------------
 * fg_envglue_jet_schedule: the action smoothing of the reference's jet envs (cylinder_env_base.py:748-753, a_k = a_{k-1} +
 * alpha (target - a_{k-1}) before every sim step) for the n sim steps of one env step at once: control[k][b] = target[b] +
 * (current[b] - target[b]) * decay[k] (decay[k] = (1 - alpha)^(k+1), device array of n), jets[k][wall][b][c][0][x] =
 * shape[c][x] * control[k][b] (two walls, two components, nx % 4 == 0), last[b] = control[n-1][b].  The slabs are what
 * fg_multi_step binds as the wall velocities of sim step k.
 * fg_envglue_channel_observe: what step() returns after them (fluid_env.py:749-800) for the channel env: cross[b] = mean(v^2),
 * shear[b] = shear_scale * (mean_x u[y=0] + mean_x u[y=ny-1]), reward[b] = -(shear + penalty * cross), the sensor gather
 * obs_velocity[b][s][c] = velocity[b][c][sensor[s]], obs_pressure[b][s] = pressure[b][sensor[s]] (sensor: flat cell indices,
 * int64, device).  One workgroup per env and a fixed summation tree: an env's result does not depend on the batch. */
int fg_envglue_jet_schedule(const float* target, const float* current, const float* decay, const float* shape, int32_t n,
                            int32_t batch, int32_t nx, float* jets, float* last, void* stream);
int fg_envglue_channel_observe(const float* velocity, const float* pressure, const int64_t* sensor, int32_t n_sensors,
                               int32_t batch, int32_t ny, int32_t nx, float shear_scale, float penalty, float* obs_velocity,
                               float* obs_pressure, float* cross, float* shear, float* reward, void* stream);

/* ---- multi-block, non-orthogonal domains -------------------------------------------
 * The reference's general Domain: several structured blocks of curvilinear cells (vertex coordinates ->
 * CoordsToTransforms, grid_gen.cu:298-390), joined by ConnectedBoundary with shuffled / inverted axes
 * (Block::ConnectBlock, domain_structs.cpp:1940-1950; ConnectBlocks :1080-1113), closed by FIXED Dirichlet
 * boundaries (Block::CloseBoundary :1995-2002) or periodic (Block::MakePeriodic :1952-1971), stepped by
 * _PISO_split_step's non-orthogonal branch (PISOtorch_simulation.py:1707-1972) with
 * nonOrthoFlags = CENTER_MATRIX | DIRECT_MATRIX | DIAGONAL_RHS (:479-487).  This is what the cylinder and airfoil
 * environments run on (envs/cylinder/grid.py:232-418).
 *
 * Build: fg_mb_create -> fg_mb_add_block per block (HOST vertex coordinates [d,(nz+1,)ny+1,nx+1], float32;
 * every face starts FIXED with zero velocity) -> fg_mb_connect / fg_mb_make_periodic -> fg_mb_finalize (builds the
 * mesh tables on the host once, allocates work space).  Faces are numbered -x,+x,-y,+y,-z,+z = 0..2d-1; the
 * connected-axis arguments use the same numbering with the low bit meaning "inverted", exactly as ConnectBlock.
 * Cells of all blocks are concatenated (block order, x fastest): N = fg_mb_sizes; the FIXED faces of all blocks are
 * concatenated into NB boundary slots (block order, face order, lowest remaining axis fastest); fg_mb_block_info
 * gives the offsets.  Fields are caller-owned fp32 device arrays: velocity [B,d,N] (updated in place), pressure
 * result [B,N] (read for the lagged corner terms, overwritten with the new mean-free pressure), boundary velocity
 * [B,d,NB] (Dirichlet values, may change between steps), optional velocity source [B,d,N].
 * fg_mb_set_reference_quirks (before finalize; default 1,1) keeps two behaviours of the reference that are not
 * geometrically motivated: diagonal walks over a connection land one layer inside the connected block
 * (PISO_multiblock_cuda_kernel.cu:2152, 2658, 2825), and the matrices drop the cross-metric terms on the inner face
 * of a first-layer cell when the far side of the block is a wall (:1952). */
typedef struct fg_mb_state* fg_mb_handle;
int fg_mb_create(int32_t dims, int32_t batch, int32_t device, fg_mb_handle* out);
int fg_mb_destroy(fg_mb_handle h);
int fg_mb_add_block(fg_mb_handle h, const fg_real* vertex_coords_host, int32_t nx, int32_t ny, int32_t nz, int32_t* block_id);
int fg_mb_connect(fg_mb_handle h, int32_t block1, int32_t face1, int32_t block2, int32_t face2, int32_t axis1, int32_t axis2);
int fg_mb_make_periodic(fg_mb_handle h, int32_t block, int32_t axis);
int fg_mb_set_reference_quirks(fg_mb_handle h, int32_t connected_diagonal_offset, int32_t first_layer_rule);
/* nonOrthoFlags of the reference (PISOtorch_simulation.py:479-487), before finalize: 25 = CENTER_MATRIX | DIRECT_MATRIX |
 * DIAGONAL_RHS, what Simulation(non_orthogonal=True) runs (default); 10 = DIRECT_RHS | DIAGONAL_RHS: every cross-metric term
 * lagged on the right-hand side, which leaves the pressure matrix symmetric with the exact constant null space */
int fg_mb_set_nonortho_flags(fg_mb_handle h, int32_t flags);
int fg_mb_finalize(fg_mb_handle h);
int fg_mb_sizes(fg_mb_handle h, int32_t* n_cells, int32_t* n_boundary_faces);
int fg_mb_block_info(fg_mb_handle h, int32_t block, int32_t* cell_offset, int32_t* boundary_slot0 /* [2d], -1 = not FIXED */);
int fg_mb_get_neighbors(fg_mb_handle h, int32_t* out_host /* [2d*N]: neighbour cell, or -1 - boundary slot */);
/* Every mesh table the kernels read (csrc/fg_mb.h), as built on the host: 0 nbr, 1 fcode, 2 T, 3 Tb, 4 bcell, 5 bface, 6 Vdiag,
 * 7 Voff, 8 KPp, 9 KPn, 10 SVc_idx, 11 SVc_w, 12 SVb_idx, 13 SVb_w, 14 SP_idx, 15 SP_face, 16 SP_wp, 17 SP_wn (4-byte elements;
 * out may be NULL to query the count).  A handle created with device < 0 is host-only: it builds and serves these tables
 * without touching a GPU (CPU parity tests of the topology code), every compute entry point refuses it. */
int fg_mb_get_host_table(fg_mb_handle h, int32_t which, void* out, int64_t* count);
int fg_mb_bind(fg_mb_handle h, fg_real* velocity, fg_real* pressure_result, fg_real* boundary_velocity, const fg_real* source);
int fg_mb_set_viscosity(fg_mb_handle h, fg_real nu);
typedef struct fg_mb_step_options {
    int32_t corrector_steps;           /* 2 */
    int32_t advect_non_ortho_steps;    /* 1 (airfoil 2) */
    int32_t pressure_non_ortho_steps;  /* 1 (cylinder 3-D and airfoil 4) */
    int32_t max_iterations;            /* 5000 */
    fg_real advection_tol;               /* RMS residual */
    fg_real pressure_tol;
    int32_t pressure_use_bicgstab;     /* 0: CG as the reference (pressure_use_BiCG=False, simulation.py:136) -- with the
                                          cross-metric terms the pressure matrix is not symmetric, CG only works while the
                                          mesh is close to orthogonal; 1: BiCGStab (restarted every 200 iterations);
                                          2: BiCGStab with the iterate kept in fp64 and the residual recomputed in fp64 at every
                                          restart (iterative refinement), best refinement point returned when unconverged -- the role of
                                          the reference's solver_double_fallback */
    int32_t pressure_warm_start;       /* 1: the first pressure solve of a corrector starts from the current pressure field
                                          instead of zero (the reference passes x=None there, PISOtorch_simulation.py:1878;
                                          same converged answer, fewer iterations) */
    int32_t pressure_project_mean;     /* 1: CG works on residuals with their mean removed -- identical on orthogonal meshes,
                                          and what keeps the solve from stalling on the constant residual component that the
                                          cross-metric terms feed (1^T P != 0); 0: the reference's plain recurrence */
    fg_real pressure_stall_accept;       /* > 1: a CG solve whose best iterate is within this factor of pressure_tol and has
                                          not improved for 20 iterations ends with that iterate: the reference's
                                          pressure matrix has a near-null LEFT vector y that is not constant, so a
                                          flux-balanced right-hand side keeps a component (y.b) y no iteration can remove
                                          -- a residual floor |y.b| / sqrt(N) that sits at 1.1e-5 on the reference's own
                                          cylinder mesh, right at its 1e-5 tolerance; 0: off */
    int32_t solver_double_fallback;    /* the reference's retry ladder (_linear_solve, PISOtorch_diff.py:410-476).  A solve counts as failed
                                          like there: the velocity solve when it did not converge, a pressure solve (returnBestResult)
                                          when its residual is non-finite.  1: a failed solve is repeated from zero with the iterate
                                          kept in fp64 and the residual b - M x recomputed in fp64 at every restart (the reference
                                          repeats it with matrix and vectors cast to fp64) */
    int32_t bicg_precondition_fallback; /* 1: a BiCGStab solve that (still) failed is repeated from zero with a preconditioner: here
                                          right diagonal scaling (M D^-1) y = b, x = D^-1 y -- same residual, same criterion; the
                                          reference's rung is cuSPARSE ILU(0), whose triangular solves are sequential */
} fg_mb_step_options;
/* dt_B: device array [B]; dt <= 0 leaves that env untouched.  stats_host (optional, 4 ints): max iterations of
 * {-, velocity, pressure corrector 0, pressure corrector 1}.  Returns FG_ERR_NOT_CONVERGED / FG_ERR_NOT_FINITE when a
 * solve failed (unconverged: fields updated from the best iterate, as with returnBestResult; non-finite: the envs concerned
 * are left as they were, see fg_mb_env_status), other negative codes on errors. */
int fg_mb_piso_step(fg_mb_handle h, const fg_real* dt_B, const fg_mb_step_options* opt, int32_t* stats_host, void* stream);
/* Per-env outcome of the last fg_mb_piso_step / fg_mb_single_step, host array [B]: 0 ok; 1 a solve of the batch ended
 * unconverged (best iterate used, returnBestResult); 2 a solve of THIS env was non-finite -- its step was not committed
 * (velocity as before the step, like solve_ok=False before CopyVelocityResultToBlocks, PISOtorch_simulation.py:1752-1757,
 * and Simulation.single_step -> False, simulation.py:259-280) while the other envs of the batch completed. */
int fg_mb_env_status(fg_mb_handle h, int32_t* out_B_host);
/* How often each rung of the retry ladder ran since the handle was created: out4 = {velocity fp64 rung, velocity preconditioned
 * rung, pressure fp64 rung, pressure last-resort CG}.  force_mask (tests): bit 0 / bit 1 make the FIRST attempt of every
 * velocity / pressure solve count as failed, so that the rungs can be exercised on systems that do not fail. */
int fg_mb_ladder(fg_mb_handle h, int64_t* out4_host, int32_t force_mask);
/* Tuning aid: with FG_MB_OC_VARIANT=256 in the environment at fg_mb_create, workgroup 0 of the on-chip CG counts shader-clock
 * cycles per phase of its loop; out12 = 11 phases + iterations of the last launch. */
int fg_mb_debug_cycles(fg_mb_handle h, uint64_t* out12_host);
/* What a handle remembers between solves besides the bound fields: where the previous solve of each place in the step finished
 * (its next solve polls there first; a verified / refined BiCGStab re-opens and restarts at its polls, so the schedule is part of
 * the arithmetic), the back-off state of the multilevel trial and (round 6) the back-off of the velocity sweeps -- skip, failures,
 * sweeps per non-orthogonal pass -- which decides whether a solve runs the sweeps or BiCGStab.  get_state / set_state of the envs
 * carry these 48 words so that a restored state replays bit for bit (reference: envs/fluid_env.py:1320-1363).  set = 0 reads, 1 writes. */
int fg_mb_solver_hints(fg_mb_handle h, int32_t* hints48, int32_t set);
/* as fg_solver_counters, for the multi-block path */
int fg_mb_solver_counters(fg_mb_handle h, int64_t* out13_host, int32_t reset);
/* The tuning / diagnosis switches a handle runs under (the FG_* environment variables read ONCE at create time, docs/SWITCHES.md,
 * and the solver policies set through the API), as one JSON object written to buf (NUL-terminated, at most n bytes; returns the
 * length needed when the buffer is too small, a negative status on error).  bench.py stores it with every result, so a number can
 * be traced to the code paths that produced it.  (No counterpart in the reference: its solver has no such switches.) */
int fg_config_dump(fg_handle h, char* buf, int n);
int fg_mb_config_dump(fg_mb_handle h, char* buf, int n);
int fg_mb_solver_unconverged(fg_mb_handle h, int64_t* out4_host);   /* as fg_solver_unconverged */
/* Simulation.single_step for such a domain (simulation.py:206-280): boundary-flux guard, per-env adaptive substeps
 * (_PISO_adaptive_step, PISOtorch_simulation.py:2004-2064), the advective-outflow PRE hook on ONE FIXED face given as
 * a range of boundary slots (update_advective_boundaries + balance_boundary_fluxes, :188-393; count 0 = none) and
 * fg_mb_piso_step per substep.  out_host as for fg_single_step: [0..3] max solver iterations of the last substep,
 * [4] substeps taken, [5] 1 if every solve converged; flux_host (optional, [B]) the boundary flux balance found. */
typedef struct fg_mb_sim_options {
    fg_mb_step_options step;
    fg_real time_step;
    fg_real cfl;
    int32_t adaptive;          /* 1: substeps from the CFL condition; 0: `substeps` equal steps */
    int32_t substeps;
    fg_real flux_balance_tol;
    int32_t outflow_slot0;
    int32_t outflow_count;
    fg_real outflow_velm[3];     /* characteristic velocity of the convective condition */
    fg_real outflow_tol;         /* tol of update_advective_boundaries (balance threshold = 0.01 tol) */
    int32_t max_substeps;      /* safety bound, 0 = none */
    int32_t outflow_slot0_b;   /* a second outflow face (the airfoil mesh has two: airfoil/grid.py:708-714); count 0 = none */
    int32_t outflow_count_b;
} fg_mb_sim_options;
int fg_mb_single_step(fg_mb_handle h, const fg_mb_sim_options* opt, int32_t* out_host, fg_real* flux_host, void* stream);
/* the PRE hook alone, same dt for every env (make_divergence_free runs it with dt = 1, PISOtorch_simulation.py:1334-1345) */
int fg_mb_update_advective_boundary(fg_mb_handle h, fg_real dt, int32_t slot0, int32_t count, int32_t slot0_b, int32_t count_b,
                                    const fg_real* velm, fg_real tol, void* stream);
/* Simulation.make_divergence_free (PISOtorch_simulation.py:1318-1429), without its PRE hook */
int fg_mb_make_divergence_free(fg_mb_handle h, const fg_mb_step_options* opt, void* stream);
/* Domain.GetBoundaryFluxBalance per env; synchronises */
int fg_mb_boundary_flux_balance(fg_mb_handle h, fg_real* out_B_host, void* stream);
/* host copies of the mesh tables: per boundary slot the owner cell, its face and Minv | det of the face
 * (k_CoordsToFaceTransforms, grid_gen.cu:398-470); per cell Minv | det (k_CoordsToTransforms, :298-354) */
int fg_mb_get_boundary_tables(fg_mb_handle h, int32_t* cell, int32_t* face, fg_real* transform);
int fg_mb_get_cell_transforms(fg_mb_handle h, fg_real* transform);
/* max |Minv u| over cells and boundary faces per env (Domain.getMaxVelocity(True, True)); synchronises */
int fg_mb_max_velocity(fg_mb_handle h, fg_real* out_B_host, void* stream);
/* pressure_project_mean keeps the CG residuals orthogonal to a unit vector: the constant by default, or y_host [N] (any scale).
 * fg_mb_unit_pressure_matrix leaves the pressure matrix for A = 1 in the P buffers so that a host routine can compute its left
 * near-null vector, the choice that removes the residual floor of non-orthogonal meshes */
int fg_mb_set_residual_projection(fg_mb_handle h, const fg_real* y_host);
/* Iterations a CG solve may go without improving its kept iterate before it ends with that iterate (default 400;
 * the reference has no such limit: its solves run to maxIterations and return the best result,
 * cg_solver_kernel.cu:345-361, PISOtorch_diff.py:266-371). */
int fg_mb_set_stall_limit(fg_mb_handle h, int32_t iterations);
/* Start vector of the first velocity solve of a step: 0 (default of a new handle) zero, the reference's non-orthogonal branch
 * (x=None at no_step 0, PISOtorch_simulation.py:1735-1742); 1 the current velocity (opt-in: fewer iterations, same answer within
 * the tolerance).  Later non-orthogonal passes always start from the previous pass's result as there. */
int fg_mb_set_advection_start(fg_mb_handle h, int from_result);
/* fg_mb_set_advection_jacobi(on): the velocity systems by point-Jacobi sweeps over the neighbour table (mb_jacobi, csrc/fg_mb_krylov.hip)
 * before the reference's BiCGStab (bicgstab_solver_kernel.cu:63-411 at PISOtorch_simulation.py:1735-1742) -- the single-block policy
 * `advection_jacobi` on the multi-block path: same systems, same criterion; a system the sweeps do not contract on (the airfoil meshes)
 * is handed to BiCGStab from a cleared start vector after the first check.  fg_mb_advection_jacobi_counts: solves settled by the
 * sweeps, solves handed on. */
int fg_mb_set_advection_jacobi(fg_mb_handle h, int on);
int fg_mb_advection_jacobi_counts(fg_mb_handle h, int64_t* out2);
/* Additive multilevel preconditioner of the pressure solves on 2-D meshes: Jacobi + 1/2 x Jacobi on 4 x 4 aggregates + the
 * dense pseudo-inverse on 8 x 8 aggregates, all from the geometry-only (A = 1) matrix and scaled per env.  Host arrays: a4 [N]
 * (aggregate of every cell), parent4 [n4] (8 x 8 aggregate of every 4 x 4 one), rect4 (every 4 x 4 aggregate as a rectangle of
 * cells of one block; checked against a4), d4g [n4], aci8 [n8 x n8]; n4 < 65535, n8 <= 2048.  Consumers: the on-chip CG
 * (k_mbc_onchip; up to 16 k cells, n4 <= 2048, n8 <= 512) applies it inside the persistent kernel, the pressure BiCGStab of any
 * mesh takes it as right preconditioner in kernel form (three launches per application).  a4 == NULL only switches it on / off
 * (enable).  The reference's CG / BiCGStab run without one (cg_solver_kernel.cu; its ILU0 is the fallback rung only); converged
 * answers agree to the solver tolerance, iteration counts drop 3-9x. */
int fg_mb_set_multilevel(fg_mb_handle h, int32_t n4, int32_t n8, const int32_t* a4_host, const int32_t* parent4_host,
                         const int32_t* rect4_host /* [n4][4]: first cell, width, height, row stride */, const fg_real* d4g_host,
                         const fg_real* aci8_host, fg_real geom_diag_sum, int32_t enable);
/* Stress harness of the multi-block velocity BiCGStab: solves the d systems per env held in the assembly buffers (FG_MB_BUF_A,
 * _C_OFF, _RHS, e.g. loaded from a dumped failing step) `reps` times from zero; out4 = solves, solves with a non-finite system,
 * unconverged solves, max iterations; acc_out / sc_out receive the recurrence words (accumulators, alpha / omega) as the last solve
 * left them.  Debugging aid (profiles/bicg_stress.py), not on any step path. */
int fg_mb_debug_bicgstab(fg_mb_handle h, fg_real tol, int32_t max_iterations, int32_t reps, int64_t* out4,
                         double* acc_out /* [B d][12] or NULL */, fg_real* sc_out /* [B d][2] or NULL */, void* stream);
/* z = M r [B,N] with the kernel form of the multilevel preconditioner on the pressure matrix currently assembled (unit test of
 * the three kernels behind the preconditioned pressure BiCGStab; synchronises). */
int fg_mb_multilevel_apply(fg_mb_handle h, const fg_real* r_BN, fg_real* z_BN, void* stream);
/* z = U^-1 L^-1 r [B,d,N] with ILU(0) of the velocity matrix assembled by the last step: the preconditioner of the multi-block
 * BiCG_precondition_fallback rung (cuSPARSE ILU(0) in the reference, bicgstab_solver_kernel.cu:191-226; level-scheduled on the
 * mesh's neighbour table here).  Test entry (synchronises); FG_ERR_UNSUPPORTED in the fp64 build and on meshes in which a cell
 * meets the same neighbour across two faces (the rung keeps right diagonal scaling there). */
int fg_mb_debug_ilu_apply(fg_mb_handle h, const fg_real* r_BdN, fg_real* z_BdN, void* stream);
/* The multilevel right preconditioner of the pressure BiCGStab is a trial with exponential back-off per handle:
 * out3 = current back-off in solves (0: no tables; 4: every attempt converges; up to 256), attempts, failed attempts (each repeated
 * with the plain recurrence). */
int fg_mb_multilevel_status(fg_mb_handle h, int32_t* out3);
/* Force on a closed wall from the bound fields: replaces the tensor arithmetic of the reference's envs/util/forces.py
 * (compute_forces_2d :193-276, compute_forces_3d :278-377) behind CylinderEnvBase._get_drag_and_lift (cylinder_env_base.py:676-700).
 * cell_index / slot_index [layers][n]: the wall-adjacent cell and the boundary slot of every wall face in ring order (device);
 * geom [5][n]: outward normal x, y, tangential spacing, wall distance, face length (device); out [B][2][layers] (device) =
 * sum over the ring of ((2 nu S - p I) n) * face length * area_scale.  Asynchronous on `stream`. */
int fg_mb_wall_forces(fg_mb_handle h, const int32_t* cell_index, const int32_t* slot_index, const fg_real* geom, int32_t n,
                      int32_t layers, fg_real area_scale, fg_real viscosity, fg_real* out, void* stream);
int fg_mb_unit_pressure_matrix(fg_mb_handle h, void* stream);
/* live timing of the CG kernel pair (kind 0: stencil kernel k_mbc_ap, 1: update kernel k_mbc_update): every fourth chunk of
 * iterations has its first pair issued with start/stop events; sums over sampled launches with live systems, their
 * ALGORITHMIC bytes (active systems x cells x per-cell figure) and the total launches since enable */
int fg_mb_profile_enable(fg_mb_handle h, int32_t on);
const char* fg_mb_profile_kind_name(int32_t kind);
int fg_mb_profile_read(fg_mb_handle h, int32_t kind, double* ms_sum, int64_t* samples, double* bytes_sum, int64_t* launches);
/* kind 2 = k_mbc_onchip, the CG that runs a whole solve of one env inside one workgroup (meshes up to 28 672 cells, 2-D):
 * every launch is timed; bytes = what it streams from L2 / Infinity Cache (off-diagonals + packed neighbour table per
 * iteration and cell); fg_mb_profile_iterations = CG iterations summed over envs and solves since enable */
int fg_mb_profile_iterations(fg_mb_handle h, int64_t* iterations);
#define FG_MB_BUF_A 0               /* [B,N]   diagonal of C */
#define FG_MB_BUF_C_OFF 1           /* [B,2d,N] */
#define FG_MB_BUF_RHS 2             /* [B,d,N] velocity right-hand side of the last solve */
#define FG_MB_BUF_H 3               /* [B,d,N] */
#define FG_MB_BUF_DIV 4             /* [B,N]   pressure right-hand side of the last solve */
#define FG_MB_BUF_P_DIAG 5
#define FG_MB_BUF_P_OFF 6
#define FG_MB_BUF_VELOCITY_RESULT 7
#define FG_MB_BUF_KRYLOV0 8          /* 8..12: the five Krylov work vectors [B,d,N] as the last solve left them (debugging) */
int fg_mb_get_buffer(fg_mb_handle h, int32_t which, const fg_real** ptr, int64_t* count);
int fg_mb_read_buffer(fg_mb_handle h, int32_t which, fg_real* dst_device, void* stream); /* device copy, synchronises */

/* ---- grid metrics --------------------------------------------------------------------------- */
/* CoordsToTransforms (grid_gen.cu:298-390): vertex coords [d,(nz+1,)ny+1,nx+1] ->
 * transforms [(nz,)ny,nx, 2 d^2 + 1] = M | Minv | det per cell. */
int fg_coords_to_transforms(const float* coords, float* transforms, int dims, int nx, int ny, int nz,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLUIDGYM_HIP_H */
