// C-ABI entry points of libfluidgym_hip.so (declared in include/fluidgym_hip.h): handle lifetime,
// field binding and the PISO step drivers.  Host code only; kernels live in fg_piso.hip,
// fg_poisson.hip, fg_bicgstab.hip, fg_metrics.hip.
#include <stdio.h>
#include <string.h>

#include <cmath>
#include <string>
#include <vector>

#include <stdlib.h>

#include "fg_internal.h"

static thread_local std::string g_last_error;
void fg_set_error(const std::string& msg) { g_last_error = msg; }

extern "C" int fg_abi_version(void) { return FG_ABI_VERSION; }
extern "C" const char* fg_last_error(void) { return g_last_error.c_str(); }

static FgBounds make_bounds(const fg_state* s, int channel) {
    FgBounds b;
    for (int f = 0; f < 6; ++f) {
        b.vel[f] = s->bvel[f];
        b.scal[f] = s->bscal[f];
        b.scalar_bc[f] = s->cfg.scalar_bc[f][channel < 0 ? 0 : channel];
    }
    return b;
}

static int check_bound(const fg_state* s, bool need_scalar) {
    FG_REQUIRE(s->velocity && s->pressure, FG_ERR_NOT_BOUND, "velocity/pressure not bound (fg_bind)");
    for (int f = 0; f < 2 * s->grid.dims; ++f) {
        if (!s->grid.fixed[f]) continue;
        FG_REQUIRE(s->bvel[f], FG_ERR_NOT_BOUND, "boundary velocity of a FIXED face not bound");
        if (need_scalar) FG_REQUIRE(s->bscal[f], FG_ERR_NOT_BOUND, "boundary scalar of a FIXED face not bound");
    }
    if (need_scalar) FG_REQUIRE(s->scalar, FG_ERR_NOT_BOUND, "passive scalar not bound");
    return FG_OK;
}

extern "C" int fg_create(const fg_config* cfg, const fg_real* hx, const fg_real* hy, const fg_real* hz, fg_handle* out) {
    FG_REQUIRE(cfg && out && hx && hy, FG_ERR_INVALID_ARG, "fg_create: null argument");
    FG_REQUIRE(cfg->dims == 2 || cfg->dims == 3, FG_ERR_INVALID_ARG, "dims must be 2 or 3");
    FG_REQUIRE(cfg->dims == 2 || hz, FG_ERR_INVALID_ARG, "hz required in 3-D");
    // minimum 3 cells per axis (domain_structs.cpp:1193-1195)
    FG_REQUIRE(cfg->nx >= 3 && cfg->ny >= 3 && (cfg->dims == 2 ? cfg->nz == 1 : cfg->nz >= 3), FG_ERR_INVALID_ARG,
               "block needs >= 3 cells per axis (nz = 1 in 2-D)");
    FG_REQUIRE(cfg->batch >= 1, FG_ERR_INVALID_ARG, "batch must be >= 1");
    FG_REQUIRE(cfg->n_scalars >= 0 && cfg->n_scalars <= FG_MAX_SCALARS, FG_ERR_INVALID_ARG, "bad n_scalars");
    for (int a = 0; a < cfg->dims; ++a)
        FG_REQUIRE(cfg->face_type[2 * a] == cfg->face_type[2 * a + 1], FG_ERR_INVALID_ARG,
                   "a FIXED face needs a FIXED partner (CloseBoundary closes both, domain_structs.cpp:1981-2002)");
    FG_REQUIRE((long long)cfg->nx * cfg->ny * cfg->nz * cfg->batch * 6 < (1LL << 31), FG_ERR_UNSUPPORTED,
               "index space exceeds int32 (reference index_t is int32 as well)");
    FG_HIP_CHECK(hipSetDevice(cfg->device));

    fg_state* s = new fg_state();
    memset(s, 0, sizeof(*s));
    s->cfg = *cfg;
    FgGrid& g = s->grid;
    g.dims = cfg->dims; g.nx = cfg->nx; g.ny = cfg->ny; g.nz = cfg->nz;
    g.n = cfg->nx * cfg->ny * cfg->nz; g.B = cfg->batch;
    for (int f = 0; f < 6; ++f) g.fixed[f] = (f < 2 * cfg->dims) ? (cfg->face_type[f] == FG_FIXED) : 0;
    s->vec = (!FG_F64 && cfg->nx % 4 == 0) ? 4 : 1;   // (the fp64 build runs scalar lanes: the float4 paths are fp32 idioms)
    s->viscosity = 0.f;

    const fg_real* hsrc[3] = {hx, hy, hz};
    const int hn[3] = {cfg->nx, cfg->ny, cfg->nz};
    for (int a = 0; a < 3; ++a) {
        std::vector<fg_real> h(hn[a], 1.f), rh(hn[a], 1.f);
        if (a < cfg->dims)
            for (int i = 0; i < hn[a]; ++i) {
                if (!(hsrc[a][i] > 0.f)) { delete s; fg_set_error("cell widths must be positive"); return FG_ERR_INVALID_ARG; }
                h[i] = hsrc[a][i]; rh[i] = 1.f / hsrc[a][i];
            }
        FG_HIP_CHECK(hipMalloc(&s->d_h[a], sizeof(fg_real) * hn[a]));
        FG_HIP_CHECK(hipMalloc(&s->d_rh[a], sizeof(fg_real) * hn[a]));
        FG_HIP_CHECK(hipMemcpy(s->d_h[a], h.data(), sizeof(fg_real) * hn[a], hipMemcpyHostToDevice));
        FG_HIP_CHECK(hipMemcpy(s->d_rh[a], rh.data(), sizeof(fg_real) * hn[a], hipMemcpyHostToDevice));
        g.h[a] = s->d_h[a]; g.rh[a] = s->d_rh[a];
    }
    const size_t BN = (size_t)g.B * g.n, d = g.dims;
    auto alloc = [&](fg_real** p, size_t count) -> hipError_t {
        hipError_t e = hipMalloc(p, sizeof(fg_real) * count);
        if (e == hipSuccess) e = hipMemset(*p, 0, sizeof(fg_real) * count);
        return e;
    };
    FG_HIP_CHECK(alloc(&s->A, BN));
    FG_HIP_CHECK(alloc(&s->rA, BN));
    FG_HIP_CHECK(alloc(&s->Coff, BN * 2 * d));
    FG_HIP_CHECK(alloc(&s->adv_rhs, BN * d));
    FG_HIP_CHECK(alloc(&s->vel_result, BN * d));
    FG_HIP_CHECK(alloc(&s->hvec, BN * d));
    FG_HIP_CHECK(alloc(&s->div, BN));
    FG_HIP_CHECK(alloc(&s->p_result, BN));
    FG_HIP_CHECK(alloc(&s->scal_result, BN));
    for (int i = 0; i < 8; ++i) FG_HIP_CHECK(alloc(&s->w[i], BN * d));
    const size_t nsys = (size_t)g.B * d;
    FG_HIP_CHECK(hipMalloc(&s->acc, sizeof(FgDacc) * nsys * FG_ACC_DOUBLES));
    FG_HIP_CHECK(hipMemset(s->acc, 0, sizeof(FgDacc) * nsys * FG_ACC_DOUBLES));
    FG_HIP_CHECK(hipMalloc(&s->flags, sizeof(int32_t) * nsys));
    FG_HIP_CHECK(hipMalloc(&s->info_dev, sizeof(fg_solve_info) * nsys));
    FG_HIP_CHECK(hipHostMalloc(&s->info_pinned, sizeof(fg_solve_info) * nsys));
    FG_HIP_CHECK(hipHostMalloc(&s->flags_pinned, sizeof(int32_t) * nsys));
    FG_HIP_CHECK(alloc(&s->scratch_B, (size_t)g.B * (4 + 2 * d)));
    FG_HIP_CHECK(hipMalloc(&s->d_bvel_ptrs, sizeof(fg_real*) * 6));
    FG_HIP_CHECK(hipMemset(s->d_bvel_ptrs, 0, sizeof(fg_real*) * 6));
    FG_HIP_CHECK(hipHostMalloc(&s->diag_pinned, sizeof(fg_real) * 2 * g.B));
    FG_HIP_CHECK(hipHostMalloc(&s->dt_pinned, sizeof(fg_real) * g.B));
    FG_HIP_CHECK(alloc(&s->dt_dev, g.B));
    s->t_rem_dev = nullptr;
#if !FG_F64
    FG_HIP_CHECK(hipMalloc(&s->t_rem_dev, sizeof(double) * g.B));
    FG_HIP_CHECK(hipMemset(s->t_rem_dev, 0, sizeof(double) * g.B));
#endif
    if (int rc = fg_poll_create(&s->poll, (int)(nsys > 2 * (size_t)g.B ? nsys : 2 * (size_t)g.B))) return rc;
    for (int k = 0; k < 4; ++k) { s->pred_bicg[k] = 2; s->pred_cg[k] = 1; }
    s->wall_forcing_axis = -1;
    s->adv_precond = 0; s->line_retries = 0; s->line_inv = nullptr; s->line_cp = nullptr; s->ilu_d = nullptr;
    s->double_fallback = 0; s->ladder_force = 0; s->r64_buf = nullptr; s->r64_acc = nullptr;
    s->ref64_outer = 0; s->ref64_tol = 0.f; s->ref64_inner = 1e-4f; s->ref64_x = nullptr; s->ref64_corrections = 0;
    for (int k = 0; k < 4; ++k) s->rung_count[k] = 0;
    s->fd_lam = nullptr; s->helm_diag = s->helm_lower = s->helm_upper = s->helm_tmp = nullptr;
    s->cg_wgs_per_slot = 256;      // (FG_CG_WGS_PER_SLOT until round 5: never changed by a test or a bench leg -- a constant since round 6)
    // FG_BICG3: z-marching two-kernel BiCGStab in 3-D (fg_bicgstab3d.hip): 0 never | > 0 always, with that z-chunk length (tests on
    // small grids) | unset: when the grid fits the tiles and fills the chip.  FG_BICG3_BXL: 16 / 32 float4 lanes along x (tile shape)
    { const char* ev = getenv("FG_BICG3"); s->bicg3_force = ev ? atoi(ev) : -1; }
    { const char* ev = getenv("FG_BICG3_BXL"); s->bicg3_bxl = ev ? atoi(ev) : 0; }
    { const char* ev = getenv("FG_BICG_SUB"); s->bicg_sub = ev ? atoi(ev) : -1; }
    { const char* ev = getenv("FG_REDUCE_WGS"); s->reduce_wgs = ev ? atoi(ev) : 0; }
    { const char* ev = getenv("FG_BICG3_MIX"); s->bicg3_mix = ev ? atoi(ev) : 3; }   // bit 0: kernel a, bit 1: kernel b as z-march (debugging)
    { const char* ev = getenv("FG_BICG_FUSED"); s->bicg_fused = ev ? atoi(ev) : 1; }   // 0 five kernels | 1 two kernels in 2-D (default) | 2 two kernels in 3-D as well   // read once, never on the step path
    s->tridiag_cb = 64;            // (FG_TRIDIAG_CB until round 5; 32 = half-width column blocks in k_tridiag_y_lds: measured, no gain)
    s->helm_cb_pref = 32;          // (FG_HELM_CB until round 5; 64-column workgroups of k_helm_apply_y: 8.1 against 7.0 us)
    { const char* ev = getenv("FG_HELM_ROWFORM"); s->helm_rowform_off = (ev && atoi(ev) == 0) ? 1 : 0; }   // 0: k_helm_coeffs + the array-form line kernels
    { const char* ev = getenv("FG_FD_FACFUSE"); s->fd_facfuse = ev ? atoi(ev) : 1; }
    { const char* ev = getenv("FG_DEV_DT"); s->dev_dt = ev ? atoi(ev) : 1; }
    // (default OFF: built for VERDICT r5 item 6 and measured on the RBC leg -- at the bench state's sub-step (0.02, not the 0.0125 the
    //  round-5 experiment costed) the velocity systems take 16-17 sweeps and the scalar systems do not contract by 0.7: 775 against 793 env-steps/s)
    { const char* ev = getenv("FG_ADV_LINESWEEP"); s->adv_linesweep = ev ? atoi(ev) : 0; }
    { const char* ev = getenv("FG_FD_ROWMEAN"); s->fd_rowmean = ev ? atoi(ev) : 1; }     // 0: the fused CG keeps the grid's A = 1 factors
    s->fd_row_epoch = -1; s->rA_epoch = 0; s->fd_row_part_epoch = -1;
    { const char* ev = getenv("FG_CG_FUSED"); s->cg_fused = ev ? atoi(ev) : 1; }       // 0: five-kernel preconditioned CG iteration (fg_poisson.hip)
    { const char* ev = getenv("FG_BICG_PFUSED"); s->bicg_pfused = ev ? atoi(ev) : 1; } // 0: eleven-launch Helmholtz-preconditioned BiCGStab iteration
    // FG_ADV_JACOBI: 1 = the velocity systems of uniform 2-D grids go to the Jacobi sweeps first (fg_jacobi.hip) | 0 = never | unset: off
    // until fg_set_advection_jacobi turns it on (the Python Simulation does, policy advection_jacobi)
    { const char* ev = getenv("FG_ADV_JACOBI"); s->adv_jacobi = ev ? atoi(ev) : 0; s->adv_jacobi_env = ev ? 1 : 0; }
    for (int k = 0; k < 4; ++k) s->jac_hist[k] = FgJacHist{0, 0, 0};
    s->jac_solves = s->jac_fallbacks = 0;
    FG_HIP_CHECK(hipHostMalloc(&s->jac_prev, sizeof(float) * 4 * nsys));   // (on-chip form: [nsys] previous residual | [nsys][2] residuals of passes 0 and 1; streaming form: [nsys][2])
    s->jac_rA_epoch = -1;
    FG_HIP_CHECK(hipMalloc(&s->fcg_alpha, sizeof(double) * 2 * (size_t)g.B));
    FG_HIP_CHECK(hipMemset(s->fcg_alpha, 0, sizeof(double) * 2 * (size_t)g.B));
    FG_HIP_CHECK(hipMalloc(&s->fcg_xsum, sizeof(FgDacc) * 2 * (size_t)g.B));
    FG_HIP_CHECK(hipMemset(s->fcg_xsum, 0, sizeof(FgDacc) * 2 * (size_t)g.B));
    FG_HIP_CHECK(hipMalloc(&s->fcg_lazy, sizeof(int32_t) * (size_t)g.B));
    FG_HIP_CHECK(hipMemset(s->fcg_lazy, 0, sizeof(int32_t) * (size_t)g.B));
    s->fcg_check0_ran = 0; s->fcg_lazy_on = 0; s->fcg_lazy_z = nullptr; s->fcg_unstored = 0; s->fcg_first_polls = 0;
    { const char* e = getenv("FG_FCG_FIRST"); s->fcg_first = (e && atoi(e) == 0) ? 0 : 1; }
    { const char* e = getenv("FG_JAC_SPEC"); s->jac_spec = (e && atoi(e) == 0) ? 0 : 1; }
    s->jac_spec_fn = nullptr; s->jac_spec_ctx = nullptr; s->jac_spec_done = 0; s->jac_spec_missed = 0;
    { const char* e = getenv("FG_FCG_SPEC"); s->fcg_spec = (e && atoi(e) == 0) ? 0 : 1; }
    s->fcg_spec_fn = nullptr; s->fcg_spec_ctx = nullptr; s->fcg_spec_done = 0;
    { const char* e = getenv("FG_JAC_PREFACTOR"); s->jac_prefactor = (e && atoi(e) == 0) ? 0 : 1; }
    { const char* e = getenv("FG_JAC_WARM"); s->jac_warm = e ? (atoi(e) != 0 ? 1 : 0) : -1; }      // (-1: where it pays, jac_warm_start)
    s->fcg_mean_ready = 0;
    s->cg_return_best = 1;
    s->cg_reset_steps = 100;
    s->adv_from_result = 1;
    FG_HIP_CHECK(hipMalloc(&s->cg_acc, sizeof(FgDacc) * (size_t)g.B * 8 * 64));
    FG_HIP_CHECK(hipMemset(s->cg_acc, 0, sizeof(FgDacc) * (size_t)g.B * 8 * 64));
    FG_HIP_CHECK(hipMalloc(&s->cg_best.best_crit, sizeof(fg_real) * g.B));
    FG_HIP_CHECK(hipMalloc(&s->cg_best.saved_crit, sizeof(fg_real) * g.B));
    FG_HIP_CHECK(hipMalloc(&s->cg_best.save_at, sizeof(int32_t) * g.B));
    FG_HIP_CHECK(alloc(&s->cg_best.best_x, BN));
    *out = s;
    return FG_OK;
}

extern "C" int fg_destroy(fg_handle s) {
    if (!s) return FG_OK;
    for (int a = 0; a < 3; ++a) { (void)hipFree(s->d_h[a]); (void)hipFree(s->d_rh[a]); }
    fg_real* owned[] = {s->A, s->rA, s->Coff, s->adv_rhs, s->vel_result, s->hvec, s->div, s->p_result, s->scal_result,
                      s->scratch_B};
    for (fg_real* p : owned) (void)hipFree(p);
    for (int i = 0; i < 8; ++i) (void)hipFree(s->w[i]);
    (void)hipFree(s->acc); (void)hipFree(s->flags); (void)hipFree(s->info_dev);
    (void)hipHostFree(s->info_pinned); (void)hipHostFree(s->flags_pinned);
    fg_prof_destroy(s);
    fg_poll_destroy(&s->poll);
    (void)hipFree(s->d_bvel_ptrs); (void)hipHostFree(s->diag_pinned); (void)hipHostFree(s->dt_pinned); (void)hipFree(s->dt_dev); (void)hipFree(s->t_rem_dev);
    float* fd[] = {s->fd_Qx, s->fd_QxT, s->fd_Qz, s->fd_QzT, s->fd_lower, s->fd_inv, s->fd_cp};
    for (float* p : fd) if (p) (void)hipFree(p);
    if (s->fd_dct_tw) { (void)hipFree(s->fd_dct_tw); (void)hipFree(s->fd_dct_rot); }
    (void)hipFree(s->cg_acc); (void)hipFree(s->fcg_alpha); (void)hipFree(s->fcg_xsum); (void)hipFree(s->fcg_lazy);
    if (s->jac_prev) (void)hipHostFree(s->jac_prev);
    (void)hipFree(s->fd_row_part); (void)hipFree(s->fd_lam_x); (void)hipFree(s->fd_row_inv); (void)hipFree(s->fd_row_cp); (void)hipFree(s->fd_row_lower);
    (void)hipFree(s->line_inv); (void)hipFree(s->line_cp); (void)hipFree(s->ilu_d);
    (void)hipFree(s->r64_buf); (void)hipFree(s->r64_acc); (void)hipFree(s->ref64_x); (void)hipFree(s->force_uniform);
    (void)hipFree(s->fd_lam); (void)hipFree(s->helm_diag); (void)hipFree(s->helm_lower); (void)hipFree(s->helm_upper); (void)hipFree(s->helm_tmp); (void)hipFree(s->helm_lower_row); (void)hipFree(s->helm_lower_row2); (void)hipFree(s->line_inv2); (void)hipFree(s->line_cp2);
    (void)hipFree(s->cg_best.best_crit); (void)hipFree(s->cg_best.saved_crit); (void)hipFree(s->cg_best.save_at); (void)hipFree(s->cg_best.best_x);
    delete s;
    return FG_OK;
}

static int sync_bvel_ptrs(fg_state* s);
extern "C" int fg_bind(fg_handle s, int field, fg_real* ptr) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    if (field == FG_VELOCITY) s->velocity = ptr;
    else if (field == FG_PRESSURE) s->pressure = ptr;
    else if (field == FG_SCALAR) s->scalar = ptr;
    else if (field == FG_VELOCITY_SOURCE) s->velocity_source = ptr;
    else if (field == FG_VISCOSITY_FIELD) s->visc_field = ptr;
    else if (field >= FG_BOUND_VELOCITY && field < FG_BOUND_VELOCITY + 6) {
        s->bvel[field - FG_BOUND_VELOCITY] = ptr;
        return sync_bvel_ptrs(s);
    }
    else if (field >= FG_BOUND_SCALAR && field < FG_BOUND_SCALAR + 6) s->bscal[field - FG_BOUND_SCALAR] = ptr;
    else FG_REQUIRE(false, FG_ERR_INVALID_ARG, "fg_bind: unknown field id");
    return FG_OK;
}

extern "C" int fg_set_viscosity(fg_handle s, fg_real v) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    s->viscosity = v;
    return FG_OK;
}
extern "C" int fg_set_scalar_viscosity(fg_handle s, int ch, fg_real v) {
    FG_REQUIRE(s && ch >= 0 && ch < FG_MAX_SCALARS, FG_ERR_INVALID_ARG, "bad channel");
    s->scalar_viscosity[ch] = v;
    s->scalar_viscosity_set = true;
    return FG_OK;
}

extern "C" int fg_set_fd_preconditioner(fg_handle s, const float* Qx, const float* QxT, const float* Qz, const float* QzT,
                                        const float* lower, const float* inv, const float* cp) {
    FG_REQUIRE(s && Qx && QxT && lower && inv && cp, FG_ERR_INVALID_ARG, "null argument");
    // (the arrays are floats in both builds: the fp64 library's plain kernels promote them on load, fg_f64_fd.hip)
    FG_REQUIRE(s->grid.fixed[2] && s->grid.fixed[3], FG_ERR_UNSUPPORTED, "FD preconditioner needs FIXED y faces");
    FG_REQUIRE(s->grid.dims == 2 || (Qz && QzT), FG_ERR_INVALID_ARG, "Qz required in 3-D");
    const size_t nx = s->grid.nx, ny = s->grid.ny, nz = s->grid.nz;
    auto up = [&](float** dst, const float* src, size_t count) -> hipError_t {
        if (*dst) (void)hipFree(*dst);
        hipError_t e = hipMalloc(dst, sizeof(float) * count);
        if (e == hipSuccess) e = hipMemcpy(*dst, src, sizeof(float) * count, hipMemcpyHostToDevice);
        return e;
    };
    FG_HIP_CHECK(up(&s->fd_Qx, Qx, nx * nx));
    FG_HIP_CHECK(up(&s->fd_QxT, QxT, nx * nx));
    if (s->grid.dims == 3) {
        FG_HIP_CHECK(up(&s->fd_Qz, Qz, nz * nz));
        FG_HIP_CHECK(up(&s->fd_QzT, QzT, nz * nz));
    }
    FG_HIP_CHECK(up(&s->fd_lower, lower, ny));
    FG_HIP_CHECK(up(&s->fd_inv, inv, nx * ny * nz));
    FG_HIP_CHECK(up(&s->fd_cp, cp, nx * ny * nz));
    s->fd_dct_x = 0;  // fg_set_fd_fast_transform marks the axis again for the new basis
    return FG_OK;
}

extern "C" int fg_set_return_best(fg_handle s, int on) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    s->cg_return_best = on ? 1 : 0;
    return FG_OK;
}

extern "C" int fg_set_cg_reset_steps(fg_handle s, int steps) {
    FG_REQUIRE(s && steps >= 0, FG_ERR_INVALID_ARG, "fg_set_cg_reset_steps: null handle or negative step count");
    s->cg_reset_steps = steps;
    return FG_OK;
}

extern "C" int fg_set_advection_start(fg_handle s, int from_result) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    s->adv_from_result = from_result ? 1 : 0;
    return FG_OK;
}

extern "C" int fg_set_wall_stress_forcing(fg_handle s, int axis, fg_real coef_lo, fg_real coef_hi) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    FG_REQUIRE(axis < s->grid.dims, FG_ERR_INVALID_ARG, "fg_set_wall_stress_forcing: bad axis");
    if (axis >= 0) {
        FG_REQUIRE(s->grid.fixed[2] && s->grid.fixed[3], FG_ERR_UNSUPPORTED, "fg_set_wall_stress_forcing: needs FIXED y faces (walls)");
        if (!s->force_uniform) {
            FG_HIP_CHECK(hipMalloc(&s->force_uniform, sizeof(fg_real) * (size_t)s->grid.B * s->grid.dims));
            FG_HIP_CHECK(hipMemset(s->force_uniform, 0, sizeof(fg_real) * (size_t)s->grid.B * s->grid.dims));
        }
    }
    s->wall_forcing_axis = axis; s->wall_forcing_coef[0] = coef_lo; s->wall_forcing_coef[1] = coef_hi;
    return FG_OK;
}

extern "C" int fg_set_advection_preconditioner(fg_handle s, int mode) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    FG_REQUIRE(mode >= 0 && mode <= 5, FG_ERR_INVALID_ARG, "fg_set_advection_preconditioner: mode must be 0 (off), 1 (line, always), 2 (line, fallback), 3 (Helmholtz, always), 4 (ILU(0), always) or 5 (ILU(0), fallback)");
    FG_REQUIRE(mode != 3 || s->fd_lam != nullptr, FG_ERR_INVALID_ARG, "fg_set_advection_preconditioner: mode 3 needs fg_set_fd_helmholtz");
    if (mode >= 4) { if (int rc = fg_ilu_alloc(s)) return rc; }
    else if (mode != 0)
        if (int rc = fg_line_alloc(s)) return rc;
    s->adv_precond = mode;
    return FG_OK;
}
extern "C" int fg_set_advection_jacobi(fg_handle s, int on) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    // (FG_ADV_JACOBI in the environment wins: A/B runs of whole envs; the fp64 build accepts the call and keeps the Krylov solver)
    if (!s->adv_jacobi_env) s->adv_jacobi = on ? 1 : 0;
    for (int k = 0; k < 4; ++k) s->jac_hist[k] = FgJacHist{0, 0, 0};
    return FG_OK;
}
extern "C" int fg_advection_jacobi_counts(fg_handle s, int64_t* out2) {
    FG_REQUIRE(s && out2, FG_ERR_INVALID_ARG, "null argument");
    out2[0] = s->jac_solves; out2[1] = s->jac_fallbacks;
    return FG_OK;
}
extern "C" int fg_set_double_fallback(fg_handle s, int on) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    s->double_fallback = on ? 1 : 0;     // (the fp64 build accepts it: nothing to fall back to there, as in the reference for fp64 fields)
    return FG_OK;
}
extern "C" int fg_ladder(fg_handle s, int64_t* out4, int32_t force_mask) {
    FG_REQUIRE(s && out4, FG_ERR_INVALID_ARG, "null argument");
    for (int k = 0; k < 4; ++k) out4[k] = s->rung_count[k];
    if (force_mask >= 0) s->ladder_force = force_mask;
    return FG_OK;
}
extern "C" int fg_debug_apply_preconditioner(fg_handle s, int mode, int nc, const fg_real* r, fg_real* z, void* stream) {
    FG_REQUIRE(s && r && z && nc >= 1 && nc <= s->grid.dims && (mode == 1 || mode == 4), FG_ERR_INVALID_ARG,
               "fg_debug_apply_preconditioner: mode 1 (y-line) or 4 (ILU(0)), 1 <= nc <= dims");
    hipStream_t st = (hipStream_t)stream;
    s->bicg_ready_nc = 0; s->cg_ready_ns = 0;
    FG_HIP_CHECK(hipMemsetAsync(s->flags, 0, sizeof(int32_t) * (size_t)s->grid.B * s->grid.dims, st));
    if (mode == 4) {
        if (int rc = fg_ilu_alloc(s)) return rc;
        if (int rc = fg_ilu_factor(s, s->A, s->Coff, st)) return rc;
        if (int rc = fg_ilu_apply(s, s->A, s->Coff, nc, r, z, st)) return rc;
    } else {
        if (int rc = fg_line_alloc(s)) return rc;
        if (int rc = fg_line_factor(s, s->A, s->Coff, nc, st)) return rc;
        if (int rc = fg_line_apply(s, s->A, s->Coff, nc, r, z, st)) return rc;
    }
    FG_HIP_CHECK(hipStreamSynchronize(st));
    return FG_OK;
}
extern "C" int fg_sgs_smagorinsky(fg_handle s, fg_real coefficient, fg_real* out_BN, void* stream) {
    FG_REQUIRE(s && out_BN, FG_ERR_INVALID_ARG, "fg_sgs_smagorinsky: null argument");
    if (int rc = check_bound(s, false)) return rc;
    return fg_launch_sgs(s, make_bounds(s, 0), coefficient, out_BN, (hipStream_t)stream);
}
extern "C" int fg_advection_retries(fg_handle s, int64_t* out, int32_t reset) {
    FG_REQUIRE(s && out, FG_ERR_INVALID_ARG, "null argument");
    *out = s->line_retries;
    if (reset) s->line_retries = 0;
    return FG_OK;
}

#if !FG_F64
bool fg_bicg3_ok(const fg_state* s, int nc, int* zc_out);
#endif
extern "C" int fg_advection_solver_form(fg_handle s, int nc, int32_t* out) {
    FG_REQUIRE(s && out, FG_ERR_INVALID_ARG, "null argument");
    int form = 0;
    if (s->bicg_fused) {
        int zc = 0;
#if !FG_F64
        if (s->grid.dims == 3 && (s->bicg3_mix & 3) == 3 && fg_bicg3_ok(s, nc, &zc)) form = 2;
        else
#endif
        if (s->grid.dims == 2 || s->bicg_fused >= 2) form = 1;
        (void)zc;
    }
    *out = form;
    return FG_OK;
}

extern "C" int fg_max_velocity(fg_handle s, fg_real* out_B, void* stream) {
    FG_REQUIRE(s && out_B, FG_ERR_INVALID_ARG, "null argument");
    if (int rc = check_bound(s, false)) return rc;
    return fg_launch_max_velocity(s, make_bounds(s, 0), out_B, (hipStream_t)stream);
}
extern "C" int fg_boundary_flux_balance(fg_handle s, fg_real* out_B, void* stream) {
    FG_REQUIRE(s && out_B, FG_ERR_INVALID_ARG, "null argument");
    if (int rc = check_bound(s, false)) return rc;
    return fg_launch_flux_balance(s, make_bounds(s, 0), out_B, (hipStream_t)stream);
}

extern "C" int fg_step_diagnostics(fg_handle s, fg_real* out_host, void* stream) {
    FG_REQUIRE(s && out_host, FG_ERR_INVALID_ARG, "null argument");
    if (int rc = check_bound(s, false)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int B = s->grid.B;
    fg_real* d = s->scratch_B;  // [0..B) flux balance, [B..2B) max velocity
    const FgBounds bnd = make_bounds(s, 0);
    if (int rc = fg_launch_flux_balance(s, bnd, d, st)) return rc;
    if (int rc = fg_launch_max_velocity(s, bnd, d + B, st)) return rc;
    FG_HIP_CHECK(hipMemcpyAsync(s->diag_pinned, d, sizeof(fg_real) * 2 * B, hipMemcpyDeviceToHost, st));
    FG_HIP_CHECK(hipStreamSynchronize(st));
    memcpy(out_host, s->diag_pinned, sizeof(fg_real) * 2 * B);
    return FG_OK;
}

static int sync_bvel_ptrs(fg_state*) { return FG_OK; }   // (the boundary pointers travel as kernel arguments: re-binding one costs nothing on the device)

extern "C" int fg_update_advective_boundary(fg_handle s, int face, const fg_real* velm, const fg_real* dt_B, void* stream) {
    FG_REQUIRE(s && velm && dt_B && face >= 0 && face < 2 * s->grid.dims, FG_ERR_INVALID_ARG, "bad argument");
    FG_REQUIRE(s->grid.fixed[face] && s->bvel[face], FG_ERR_INVALID_ARG, "face is not a bound FIXED face");
    if (int rc = check_bound(s, false)) return rc;
    return fg_launch_outflow(s, face, velm[face >> 1], dt_B, (hipStream_t)stream);
}

extern "C" int fg_balance_boundary_fluxes(fg_handle s, int free_face_mask, fg_real atol, const fg_real* dt_B, void* stream) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    if (int rc = check_bound(s, false)) return rc;
    return fg_launch_balance(s, make_bounds(s, 0), free_face_mask, atol, dt_B, (hipStream_t)stream);
}

static int setup_advection(fg_handle s, const fg_real* dt_B, int for_scalar, int channel, void* stream, int buoy_axis, fg_real buoy_factor);
extern "C" int fg_setup_advection(fg_handle s, const fg_real* dt_B, int for_scalar, int channel, void* stream) {
    return setup_advection(s, dt_B, for_scalar, channel, stream, -1, 0);
}
// buoy_axis >= 0 (velocity system): the buoyancy hook rides in the assembly (FgAdvArgs::buoy_T)
static int setup_advection(fg_handle s, const fg_real* dt_B, int for_scalar, int channel, void* stream, int buoy_axis, fg_real buoy_factor) {
    FG_REQUIRE(s && dt_B, FG_ERR_INVALID_ARG, "null argument");
    if (int rc = check_bound(s, for_scalar != 0)) return rc;
    FG_REQUIRE(!for_scalar || (channel >= 0 && channel < s->cfg.n_scalars), FG_ERR_INVALID_ARG, "bad scalar channel");
    s->cur_dt = dt_B;
    FgAdvArgs a;
    memset(&a, 0, sizeof(a));
    a.vel = s->velocity;
    a.dt = dt_B;
    a.for_scalar = for_scalar;
    a.channel = channel;
    a.n_scalars = s->cfg.n_scalars;
    a.A = s->A; a.Coff = s->Coff; a.rhs = s->adv_rhs;
    if (for_scalar) {
        a.scal = s->scalar + (size_t)channel * s->grid.n;
        a.scal_env_stride = (long)s->cfg.n_scalars * s->grid.n;
        // getViscosity(domain, forPassiveScalar, ch) (PISO_multiblock_cuda_kernel.cu:1803-1815)
        a.nu = s->scalar_viscosity_set ? s->scalar_viscosity[channel] : s->viscosity;
    } else {
        a.source = s->velocity_source;
        // wall-stress forcing (fg_set_wall_stress_forcing; the env's PRE hook, tcf_env.py / grid.py:147-176): recomputed from u^n for
        // EVERY velocity assembly -- the fused step and the hook-by-hook path (Simulation._split_step_hooked) alike; u^n does not change
        // between the PRE hook and this point of the step
        if (s->wall_forcing_axis >= 0)
            if (int rc = fg_launch_wall_forcing(s, (hipStream_t)stream)) return rc;
        a.force = s->wall_forcing_axis >= 0 ? s->force_uniform : nullptr;
        a.visc = s->visc_field;
        a.nu = s->viscosity;
        a.rA = s->rA;
        s->rA_epoch++;
        s->jac_rA_epoch = s->rA_epoch;      // (rA = 1 / A of THIS velocity matrix: what the streaming Jacobi sweeps read, fg_jacobi.hip)
#if !FG_F64
        if (fg_fd_rowmean_ok(s) && s->vec == 4 && (s->grid.nx & 63) == 0) {      // row sums of 1/A for the row-mean preconditioner ride in the assembly
            if (!s->fd_row_part) FG_HIP_CHECK(hipMalloc(&s->fd_row_part, sizeof(float) * (size_t)s->grid.B * s->grid.ny * (s->grid.nx / 64)));
            a.row_part = s->fd_row_part;
            s->fd_row_part_epoch = s->rA_epoch;
        }
#endif
        if (buoy_axis >= 0) {
            a.buoy_T = s->scalar; a.buoy_stride = (long)s->cfg.n_scalars * s->grid.n; a.buoy_axis = buoy_axis; a.buoy_factor = buoy_factor;
            a.source_w = s->velocity_source;
        }
    }
    return fg_launch_adv_build(s, make_bounds(s, channel), a, (hipStream_t)stream);
}

static int advection_solve(fg_state* s, FgBicgArgs a, fg_solve_info* info, hipStream_t st, int for_scalar = 0, int channel = 0);

extern "C" int fg_solve_advection(fg_handle s, int for_scalar, int channel, fg_real tol, int max_iterations,
                                  fg_solve_info* info_host, void* stream) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    FgBicgArgs a;
    a.diag = s->A; a.off = s->Coff; a.rhs = s->adv_rhs;
    a.dt = s->cur_dt;
    a.tol = tol; a.max_iterations = max_iterations;
    if (for_scalar) { a.x = s->scal_result; a.nc = 1; a.use_x0 = 0; a.kind = 0; }
    else { a.x = s->vel_result; a.nc = s->grid.dims; a.use_x0 = s->adv_from_result; }
    return advection_solve(s, a, info_host, (hipStream_t)stream, for_scalar, channel);
}

extern "C" int fg_copy_scalar_result_to_blocks(fg_handle s, int channel, void* stream) {
    FG_REQUIRE(s && s->scalar, FG_ERR_NOT_BOUND, "scalar not bound");
    const int n = s->grid.n;
    if (s->cfg.n_scalars == 1) return fg_launch_copy_active(s, s->cur_dt, s->scal_result, s->scalar, 1, (hipStream_t)stream);
    // strided destination: channel `channel` of [B,C,N] (all envs)
    FG_HIP_CHECK(hipMemcpy2DAsync(s->scalar + (size_t)channel * n, sizeof(fg_real) * n * s->cfg.n_scalars, s->scal_result,
                                  sizeof(fg_real) * n, sizeof(fg_real) * n, s->grid.B, hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream));
    return FG_OK;
}

extern "C" int fg_setup_pressure_matrix(fg_handle s, void* stream) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    s->rA_epoch++;
    return fg_launch_pressure_setup(s, s->cur_dt, (hipStream_t)stream);
}

extern "C" int fg_setup_pressure_rhs(fg_handle s, const fg_real* dt_B, void* stream) {
    FG_REQUIRE(s && dt_B, FG_ERR_INVALID_ARG, "null argument");
    if (int rc = check_bound(s, false)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = fg_launch_h(s, dt_B, s->vel_result, st)) return rc;
    return fg_launch_div(s, make_bounds(s, 0), dt_B, s->hvec, s->div, st);
}

// k_h and the divergence kernel of corrector 0, launched by fg_jacobi_solve behind its check kernel (fg_state::jac_spec_fn)
struct SpecH { fg_state* s; const fg_real* dt; const fg_step_options* opt; hipStream_t st; };
static int spec_h(void* p) {
    const SpecH* c = static_cast<const SpecH*>(p);
    fg_state* s = c->s;
    if (int rc = fg_launch_h(s, c->dt, s->vel_result, c->st)) return rc;
    return fg_launch_div(s, make_bounds(s, 0), c->dt, s->hvec, s->div, c->st, !c->opt->pressure_warm_start,
                         c->opt->pressure_method == FG_SOLVER_FDCG && s->fd_Qx != nullptr);
}

// the corrector of fg_piso_step in its unstored-pressure form (FgLazyRef), launched by fg_cg_solve behind k_fcg_check0 (fg_state::fcg_spec_fn)
struct SpecCorrect { fg_state* s; const fg_real* dt; bool last; hipStream_t st; };
static int spec_correct(void* p) {
    const SpecCorrect* c = static_cast<const SpecCorrect*>(p);
    fg_state* s = c->s;
    const FgMeanRef mean = {s->fcg_xsum, s->info_dev, s->pressure, s->fcg_lazy, s->fcg_alpha};
    const FgLazyRef lazy = {s->fcg_lazy, s->fcg_alpha, c->last ? s->p_result : nullptr};
    return fg_launch_correct(s, c->dt, s->rA, s->hvec, s->fcg_lazy_z, s->vel_result, c->st, c->last ? s->velocity : nullptr, c->last ? &mean : nullptr, &lazy);
}

static int solve_pressure(fg_state* s, const fg_real* dt, int method, fg_real tol, int max_iterations, int use_previous,
                          fg_solve_info* info_host, hipStream_t st, bool finalize = true, int kind = 2) {
    int rc = FG_OK;
    if (method == FG_SOLVER_CG || method == FG_SOLVER_FDCG) {
        FgCgArgs a;
        a.rA = s->rA; a.b = s->div; a.x = s->p_result;
        a.r = s->w[0]; a.p = s->w[1]; a.Ap = s->w[2];
        a.dt = dt; a.tol = tol; a.max_iterations = max_iterations; a.use_x0 = use_previous;
        a.reset_steps = s->cg_reset_steps;  // residual_reset_step=100 (PISOtorch_simulation.py:1913; fg_set_cg_reset_steps)
        a.precond = (method == FG_SOLVER_FDCG);
        a.kind = kind;
        a.check_every = a.precond ? 2 : 16;
        a.lazy_ok = finalize ? 0 : 1;      // (the fused step's correctors read the result through FgLazyRef)
#if !FG_F64
        if (s->ref64_outer > 0) a.lazy_ok = 0;      // (the refinement below starts from the STORED fp32 result)
#endif
        rc = fg_cg_solve(s, a, info_host, st);
#if !FG_F64
        // opt-in: fp64 residual, fp32 corrections (fg_refine64_pressure) -- the solve then ends on the fp64 residual's own verdict
        if (s->ref64_outer > 0 && (rc == FG_OK || rc == FG_ERR_NOT_CONVERGED)) {
            s->fcg_lazy_on = 0; s->fcg_check0_ran = 0; s->fcg_mean_ready = 0; s->fcg_spec_done = 0;      // (the result is rewritten in p_result)
            rc = fg_refine64_pressure(s, a, info_host, st);
        }
        // pressure solves run with returnBestResult: only a NON-FINITE solve counts as failed and is repeated in fp64
        // (solver_double_fallback, PISOtorch_diff.py:410-445)
        if (s->double_fallback && (rc == FG_ERR_NOT_FINITE || (s->ladder_force & 2))) {
            std::vector<fg_solve_info> tmp;
            fg_solve_info* info = info_host;
            if (!info) { tmp.assign(s->info_pinned, s->info_pinned + s->grid.B); info = tmp.data(); }
            s->rung_count[2] += 1;
            s->fcg_lazy_on = 0; s->fcg_check0_ran = 0; s->fcg_mean_ready = 0; s->fcg_spec_done = 0;      // (the repeat stores its result in p_result)
            rc = fg_rung64_cg(s, a, info, (s->ladder_force & 2) != 0, st);
        }
#endif
    } else {
        fg_set_error("fg_solve_pressure: only FG_SOLVER_CG drives the PISO step (Jacobi/RBGS are smoothers)");
        return FG_ERR_UNSUPPORTED;
    }
    if (rc != FG_OK && rc != FG_ERR_NOT_CONVERGED) return rc;
    // p -= mean(p); setPressureResult; CopyPressureResultToBlocks (PISOtorch_simulation.py:1817-1821, 1953).
    // finalize == false (inner correctors of the fused step): the velocity corrector only needs grad p and the next
    // solve starts from p_result, both blind to the constant, so the two passes over p are left to the last corrector.
    if (finalize)
        if (int rc2 = fg_launch_mean_sub(s, dt, s->p_result, s->pressure, st)) return rc2;
    return rc;
}

extern "C" int fg_solve_pressure(fg_handle s, int method, fg_real tol, int max_iterations, int use_previous,
                                 fg_solve_info* info_host, void* stream) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    if (int rc = check_bound(s, false)) return rc;
    return solve_pressure(s, s->cur_dt, method, tol, max_iterations, use_previous, info_host, (hipStream_t)stream);
}

extern "C" int fg_set_pressure_refinement(fg_handle s, int32_t max_corrections, fg_real target_tol, fg_real inner_relative_tol) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    FG_REQUIRE(!FG_F64 || max_corrections <= 0, FG_ERR_UNSUPPORTED, "fg_set_pressure_refinement: the fp64 build solves in fp64 already");
    FG_REQUIRE(max_corrections <= 0 || (target_tol > 0 && inner_relative_tol > 0 && inner_relative_tol < 1), FG_ERR_INVALID_ARG,
               "fg_set_pressure_refinement: tolerances must be positive, the relative one below 1");
    s->ref64_outer = max_corrections > 0 ? max_corrections : 0;
    s->ref64_tol = (float)target_tol; s->ref64_inner = (float)inner_relative_tol;
    return FG_OK;
}

extern "C" int fg_correct_velocity(fg_handle s, void* stream) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    if (int rc = check_bound(s, false)) return rc;
    return fg_launch_correct(s, s->cur_dt, s->rA, s->hvec, s->pressure, s->vel_result, (hipStream_t)stream);
}

extern "C" int fg_copy_velocity_result_to_blocks(fg_handle s, void* stream) {
    FG_REQUIRE(s && s->velocity, FG_ERR_NOT_BOUND, "velocity not bound");
    return fg_launch_copy_active(s, s->cur_dt, s->vel_result, s->velocity, s->grid.dims, (hipStream_t)stream);
}
extern "C" int fg_copy_velocity_result_from_blocks(fg_handle s, void* stream) {
    FG_REQUIRE(s && s->velocity, FG_ERR_NOT_BOUND, "velocity not bound");
    return fg_launch_copy_active(s, nullptr, s->velocity, s->vel_result, s->grid.dims, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// fused driver: _PISO_split_step, orthogonal branch (PISOtorch_simulation.py:1431-2002)
// ---------------------------------------------------------------------------------------------
static int max_iters(const fg_solve_info* info, int n) {
    int m = -1;
    for (int i = 0; i < n; ++i) m = info[i].used_iterations > m ? info[i].used_iterations : m;
    return m;
}

// One advection-diffusion solve under the handle's preconditioner policy (fg_set_advection_preconditioner), the single-block
// form of the reference's retry chain (_linear_solve, PISOtorch_diff.py:449-476): mode 1 preconditions every solve
// (preconditionBiCG), mode 2 repeats a solve that ended unconverged or non-finite from zero WITH the preconditioner
// (BiCG_precondition_fallback) -- the reference's preconditioner is cuSPARSE ILU(0), here the y-line solve of fg_linepre.hip.
static int advection_solve(fg_state* s, FgBicgArgs a, fg_solve_info* info, hipStream_t st, int for_scalar, int channel) {
    a.precond = (s->adv_precond == 1) ? 1 : (s->adv_precond == 3 ? 2 : (s->adv_precond == 4 ? 3 : 0));
    if (a.precond == 2 && !for_scalar && s->visc_field) a.precond = 0;   // the Helmholtz operator is built for ONE viscosity
    if (a.precond == 2) {   // Helmholtz (fast-diagonalisation) preconditioner: the diffusivity and wall treatment of THIS solve
        a.nu = for_scalar ? (s->scalar_viscosity_set ? s->scalar_viscosity[channel] : s->viscosity) : s->viscosity;
        a.wall_lo = for_scalar ? (s->cfg.scalar_bc[2][channel] == FG_DIRICHLET) : 1;
        a.wall_hi = for_scalar ? (s->cfg.scalar_bc[3][channel] == FG_DIRICHLET) : 1;
    }
    // The reference's ladder (_linear_solve_wrapper, PISOtorch_diff.py:410-476), in its order: the plain solve; if it failed (these
    // solves run without returnBestResult: "not converged" fails them) and solver_double_fallback is on, the same system in fp64
    // from a cleared result (fg_rung64.h); if that failed too and BiCG_precondition_fallback is on, the preconditioned solve from zero.
    const int nsys = s->grid.B * a.nc;
    std::vector<fg_solve_info> info_own;
    if (!info) { info_own.resize(nsys); info = info_own.data(); }      // (the rungs below read and update the solve infos in place)
    int rc = fg_bicgstab_solve(s, a, info, st);
    bool failed = (rc == FG_ERR_NOT_CONVERGED || rc == FG_ERR_NOT_FINITE) || (s->ladder_force & 1);
#if !FG_F64
    if (failed && s->double_fallback && rc != FG_ERR_HIP) {
        s->rung_count[0] += 1;
        rc = fg_rung64_bicgstab(s, a, info, (s->ladder_force & 1) != 0, st);
        if (rc != FG_OK && rc != FG_ERR_NOT_CONVERGED && rc != FG_ERR_NOT_FINITE) return rc;
        failed = (rc != FG_OK) || (s->ladder_force & 4);
    }
#endif
    if (failed && (s->adv_precond == 2 || s->adv_precond == 5)) {
        s->line_retries += 1;
        s->rung_count[1] += 1;
        a.precond = s->adv_precond == 5 ? 3 : 1; a.use_x0 = 0;
        rc = fg_bicgstab_solve(s, a, info, st);
    }
    (void)nsys;
    return rc;
}

extern "C" int fg_piso_step(fg_handle s, const fg_real* dt_B, const fg_step_options* opt, int32_t* stats_host,
                            void* stream) {
    FG_REQUIRE(s && dt_B && opt, FG_ERR_INVALID_ARG, "null argument");
    const bool scalar = opt->advect_scalar && s->cfg.n_scalars > 0;
    if (int rc = check_bound(s, scalar)) return rc;
    FG_REQUIRE(opt->buoyancy_axis < s->grid.dims, FG_ERR_INVALID_ARG, "bad buoyancy axis");
    FG_REQUIRE(opt->buoyancy_axis < 0 || (s->velocity_source && s->scalar), FG_ERR_NOT_BOUND,
               "buoyancy needs velocity source and scalar bound");
    hipStream_t st = (hipStream_t)stream;
    const int B = s->grid.B, d = s->grid.dims;
    std::vector<fg_solve_info> info((size_t)B * d);
    int32_t stats[4] = {-1, -1, -1, -1};
    int status = FG_OK;
    auto soft = [&](int rc) {  // non-convergence is reported, not fatal (pressure_return_best_result=True)
        if (rc == FG_ERR_NOT_CONVERGED) { status = rc; return FG_OK; }
        return rc;
    };
    // (PRE hook fused: the wall-stress forcing of the turbulent-channel env is computed by the velocity fg_setup_advection below)
#if !FG_F64
    // Helmholtz-preconditioned solves (RBC): the factors of the scalar AND the velocity system depend on dt, the diffusivities and
    // the wall conditions only -- one launch for both, ahead of the assemblies (fg_helm_factor_pair; the solves find the record)
    // (not when the line sweeps are expected to settle both systems: the BiCGStab that needs the factors then never runs -- a solve the
    //  sweeps hand over makes its own, fg_helm_factor)
    const bool sweeps_first = s->adv_linesweep && s->grid.dims == 2 && s->jac_hist[0].skip == 0 && s->jac_hist[1].skip == 0;
    if (scalar && s->cfg.n_scalars == 1 && s->adv_precond == 3 && s->fd_lam && !s->visc_field && !sweeps_first) {
        const float nu2[2] = {s->scalar_viscosity_set ? s->scalar_viscosity[0] : s->viscosity, s->viscosity};
        const int wlo[2] = {s->cfg.scalar_bc[2][0] == FG_DIRICHLET, 1}, whi[2] = {s->cfg.scalar_bc[3][0] == FG_DIRICHLET, 1};
        if (int rc = fg_helm_factor_pair(s, dt_B, nu2, wlo, whi, st)) return rc;
    }
#endif
    // ---- passive scalars (:1471-1644)
    if (scalar) {
        FgRange range_scalar("scalar_advection");
        for (int ch = 0; ch < s->cfg.n_scalars; ++ch) {
            if (int rc = fg_setup_advection(s, dt_B, 1, ch, stream)) return rc;
            FgBicgArgs a;
            a.diag = s->A; a.off = s->Coff; a.rhs = s->adv_rhs; a.x = s->scal_result; a.nc = 1;
            a.dt = dt_B; a.tol = opt->advection_tol; a.max_iterations = opt->max_iterations; a.use_x0 = 0; a.kind = 0;
            if (int rc = soft(advection_solve(s, a, info.data(), st, 1, ch))) return rc;
            const int m = max_iters(info.data(), B);
            stats[0] = m > stats[0] ? m : stats[0];
            s->ctr.add(0, info.data(), B);
            // CopyScalarResultToBlocks for active envs
            const int n = s->grid.n;
            if (s->cfg.n_scalars == 1) {
                if (int rc = fg_launch_copy_active(s, dt_B, s->scal_result, s->scalar, 1, st)) return rc;
            } else {
                (void)n;
                fg_set_error("multi-channel scalar copy with inactive envs not implemented");
                return FG_ERR_UNSUPPORTED;
            }
        }
    }
    // ---- PRE_VELOCITY_SETUP hook fused: RBC buoyancy (rbc_env_base.py:285-297)
    //      (folded into the assembly that reads the source: FgAdvArgs::buoy_T -- no k_buoyancy launch, the source field is written, not
    //      written and read back)
    // ---- velocity predictor (:1646-1762)
    fg_range_push("velocity_assembly_and_solve");
    struct PopOnce { bool armed = true; void fire() { if (armed) { fg_range_pop(); armed = false; } } ~PopOnce() { fire(); } } range_velocity;
    if (int rc = setup_advection(s, dt_B, 0, 0, stream, opt->buoyancy_axis, opt->buoyancy_factor)) return rc;
    {
        FgBicgArgs a;
        a.diag = s->A; a.off = s->Coff; a.rhs = s->adv_rhs; a.x = s->vel_result; a.nc = d;
        a.dt = dt_B; a.tol = opt->advection_tol; a.max_iterations = opt->max_iterations; a.use_x0 = s->adv_from_result;
        SpecH spec_h_ctx = {s, dt_B, opt, st};
        s->jac_spec_done = 0;
        if (opt->corrector_steps > 0) { s->jac_spec_fn = spec_h; s->jac_spec_ctx = &spec_h_ctx; }
        const int arc = soft(advection_solve(s, a, info.data(), st));
        s->jac_spec_fn = nullptr; s->jac_spec_ctx = nullptr;
        if (arc) return arc;
        stats[1] = max_iters(info.data(), B * d);
        s->ctr.add(1, info.data(), B * d);
    }
    // ---- correctors (:1777-1972); rA = 1/A was written by the velocity fg_setup_advection above
    range_velocity.fire();
    for (int c = 0; c < opt->corrector_steps; ++c) {
        FgRange range_corr(c == 0 ? "pressure_corrector_0" : "pressure_corrector_1+");
        const bool last = (c + 1 == opt->corrector_steps);
        if (!(c == 0 && s->jac_spec_done)) {      // (corrector 0: these two may already run behind the sweeps' check kernel, spec_h)
            if (int rc = fg_launch_h(s, dt_B, s->vel_result, st)) return rc;
            if (int rc = fg_launch_div(s, make_bounds(s, 0), dt_B, s->hvec, s->div, st, !opt->pressure_warm_start,
                                       opt->pressure_method == FG_SOLVER_FDCG && s->fd_Qx != nullptr))
                return rc;
        }
        s->jac_spec_done = 0;
        // The mean removal + block copy of the last corrector's pressure (setPressureResult, CopyPressureResultToBlocks: :1922-1925, 1953):
        // when the solver left sum(p) behind (fused CG, fg_fftcg.hip) the corrector does both where it reads p for the gradient --
        // pressureResult then keeps its constant, which nothing downstream sees (grad p; the next solve starts from zero or from it)
        SpecCorrect spec_ctx = {s, dt_B, last, st};
        s->fcg_spec_fn = spec_correct; s->fcg_spec_ctx = &spec_ctx;
        const int prc = soft(solve_pressure(s, dt_B, opt->pressure_method, opt->pressure_tol, opt->max_iterations,
                                            opt->pressure_warm_start ? 1 : 0,
                                            info.data(), st, false, c == 0 ? 0 : 1));
        s->fcg_spec_fn = nullptr; s->fcg_spec_ctx = nullptr;
        if (prc) return prc;
        if (c < 2) { stats[2 + c] = max_iters(info.data(), B); s->ctr.add(2 + c, info.data(), B); }
        const bool mean_folded = last && s->fcg_mean_ready;
        if (last && !mean_folded)
            if (int rc = fg_launch_mean_sub(s, dt_B, s->p_result, s->pressure, st)) return rc;
        const FgMeanRef mean = {s->fcg_xsum, s->info_dev, s->pressure, s->fcg_check0_ran ? s->fcg_lazy : nullptr, s->fcg_alpha};
        // the solve ended on the first iterate of every env and stored no pressure: the corrector reads alpha z (FgLazyRef; the last
        // one also stores pressureResult)
        const bool lazy_p = s->fcg_lazy_on && (!last || mean_folded);
        FG_REQUIRE(!s->fcg_lazy_on || lazy_p, FG_ERR_UNSUPPORTED, "fg_piso_step: an unmaterialised pressure without its folded mean");
        const FgLazyRef lazy = {s->fcg_lazy, s->fcg_alpha, last ? s->p_result : nullptr};
        // the last corrector also writes the block velocity of active envs: CopyVelocityResultToBlocks (:1974)
        if (s->fcg_spec_done && lazy_p) continue;      // (this very launch already ran behind the verdict kernel: spec_correct)
        if (int rc = fg_launch_correct(s, dt_B, s->rA, s->hvec, lazy_p ? s->fcg_lazy_z : ((last && !mean_folded) ? s->pressure : s->p_result),
                                       s->vel_result, st, last ? s->velocity : nullptr, mean_folded ? &mean : nullptr, lazy_p ? &lazy : nullptr))
            return rc;
    }
    if (opt->corrector_steps <= 0)
        if (int rc = fg_launch_copy_active(s, dt_B, s->vel_result, s->velocity, d, st)) return rc;
    if (stats_host) memcpy(stats_host, stats, sizeof(stats));
    s->ctr.piso_steps += 1;
    return status;
}

extern "C" int fg_config_dump(fg_handle s, char* buf, int n) {
    FG_REQUIRE(s && buf && n > 0, FG_ERR_INVALID_ARG, "fg_config_dump: bad argument");
    char tmp[3072];
    const int len = snprintf(tmp, sizeof(tmp),
        "{\"build\": \"%s\", \"FG_CG_FUSED\": %d, \"FG_BICG_PFUSED\": %d, \"FG_BICG_FUSED\": %d, \"FG_BICG_SUB\": %d, \"FG_BICG3\": %d, "
        "\"FG_BICG3_BXL\": %d, \"FG_BICG3_MIX\": %d, \"FG_REDUCE_WGS\": %d, \"FG_CG_WGS_PER_SLOT\": %d, \"FG_TRIDIAG_CB\": %d, \"FG_HELM_CB\": %d, "
        "\"FG_HELM_ROWFORM\": %d, \"FG_FD_ROWMEAN\": %d, \"FG_POLL_SPIN\": %d, \"FG_PROF_PERIOD\": %d, \"fast_transform_x\": %d, \"fd_preconditioner\": %d, "
        "\"helmholtz\": %d, \"advection_preconditioner\": %d, \"advection_from_result\": %d, \"return_best\": %d, \"cg_reset_steps\": %d, "
        "\"double_fallback\": %d, \"wall_forcing_axis\": %d, \"FG_ADV_JACOBI\": %d, \"FG_FCG_FIRST\": %d, \"FG_JAC_WARM\": %d, \"first_iterate_polls\": %ld, \"unstored_pressure_solves\": %ld, \"FG_JAC_SPEC\": %d, \"FG_FCG_SPEC\": %d, \"jacobi_speculation_misses\": %ld, \"jacobi_floor_released\": %ld, \"pressure_refinement_corrections\": %ld}",
        FG_F64 ? "f64" : "f32", s->cg_fused, s->bicg_pfused, s->bicg_fused, s->bicg_sub, s->bicg3_force, s->bicg3_bxl, s->bicg3_mix, s->reduce_wgs,
        s->cg_wgs_per_slot, s->tridiag_cb, s->helm_cb_pref, s->helm_rowform_off ? 0 : 1, s->fd_rowmean, s->poll.spin, s->prof.period, s->fd_dct_x, s->fd_Qx ? 1 : 0,
        s->fd_lam ? 1 : 0, s->adv_precond, s->adv_from_result, s->cg_return_best, s->cg_reset_steps, s->double_fallback, s->wall_forcing_axis, s->adv_jacobi, s->fcg_first, s->jac_warm,
        s->fcg_first_polls, s->fcg_unstored, s->jac_spec, s->fcg_spec, s->jac_spec_missed, s->jac_floor_released, s->ref64_corrections);
    if (len >= n) return len + 1;
    memcpy(buf, tmp, (size_t)len + 1);
    return FG_OK;
}

extern "C" int fg_solver_unconverged(fg_handle s, int64_t* out4) {
    FG_REQUIRE(s != nullptr && out4 != nullptr, FG_ERR_INVALID_ARG, "fg_solver_unconverged: bad argument");
    for (int k = 0; k < 4; ++k) out4[k] = s->ctr.unconv[k];
    return FG_OK;
}

extern "C" int fg_solver_counters(fg_handle s, int64_t* out13, int32_t reset) {
    FG_REQUIRE(s != nullptr, FG_ERR_INVALID_ARG, "fg_solver_counters: null handle");
    if (out13) s->ctr.write(out13);
    if (reset) s->ctr.reset();
    return FG_OK;
}

extern "C" int fg_single_step(fg_handle s, const fg_sim_options* o, int32_t* out, fg_real* flux_host, void* stream);
extern "C" int fg_multi_step(fg_handle s, const fg_sim_options* o, int32_t n, const fg_real* const* bvel_schedule, int32_t* out_6n,
                             fg_real* flux_host, int32_t* steps_done, void* stream) {
    FG_REQUIRE(s && o && out_6n && n >= 0, FG_ERR_INVALID_ARG, "fg_multi_step: bad argument");
    if (steps_done) *steps_done = 0;
    for (int k = 0; k < n; ++k) {
        if (bvel_schedule)
            for (int f = 0; f < 2 * s->grid.dims; ++f)
                if (const fg_real* p = bvel_schedule[(size_t)k * 6 + f])
                    if (int rc = fg_bind(s, FG_BOUND_VELOCITY + f, const_cast<fg_real*>(p))) return rc;
        if (int rc = fg_single_step(s, o, out_6n + 6 * (size_t)k, flux_host, stream)) return rc;
        if (steps_done) *steps_done = k + 1;
        // the loop this call stands in for (Simulation.multi_step) stops at the first step whose solves did not converge unless the
        // best iterate is accepted (pressure_return_best_result): so does this one -- *steps_done says how far it got
        if (out_6n[6 * (size_t)k + 5] == 0 && !s->cg_return_best) break;
    }
    return FG_OK;
}

// np.isclose(a, 0) with the default rtol=1e-5, atol=1e-8
static inline bool is_close_zero(double a) { return std::fabs(a) <= 1e-8; }

extern "C" int fg_single_step(fg_handle s, const fg_sim_options* o, int32_t* out, fg_real* flux_host, void* stream) {
    fg_htrace("single_step_in");
    FgRange range_sim("fg_single_step");
    FG_REQUIRE(s && o && out, FG_ERR_INVALID_ARG, "null argument");
    if (int rc = check_bound(s, o->step.advect_scalar && s->cfg.n_scalars > 0)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int B = s->grid.B;
    const FgBounds bnd = make_bounds(s, 0);
    std::vector<double> t_rem(B, (double)o->time_step);
    int32_t stats[4] = {-1, -1, -1, -1};
    int substeps = 0, all_ok = 1;
    bool first = true;
    int fixed_left = o->adaptive ? 0 : (o->substeps > 0 ? o->substeps : 1);
    for (;;) {
        bool any = false;
        if (o->adaptive) {
            for (int b = 0; b < B; ++b) any = any || (t_rem[b] > 0 && !is_close_zero(t_rem[b]));
        } else {
            any = fixed_left > 0;
        }
        if (!any) break;
        // the reference's sub-step rule for env b from its CFL maximum (PISOtorch_simulation.py:2013-2031)
        auto advance = [&](int b, fg_real max_vel) -> fg_real {
            fg_real ts = 0.f;
            if (!o->adaptive) {
                ts = o->time_step;
            } else if (t_rem[b] > 0 && !is_close_zero(t_rem[b])) {
                const double mv = (double)max_vel;
                const double max_ts = is_close_zero(mv) ? t_rem[b] : (double)o->cfl / mv;
                double tsd;
                if (max_ts >= t_rem[b]) tsd = t_rem[b];
                else tsd = t_rem[b] / (double)(long long)std::ceil(t_rem[b] / max_ts);
                t_rem[b] -= tsd;
                ts = (fg_real)tsd;  // the reference rounds ts through the domain dtype (PISOtorch_simulation.py:2029-2031)
            }
            return ts;
        };
        auto flux_guard = [&]() -> int {
            fg_real worst = 0.f;
            for (int b = 0; b < B; ++b) {
                if (flux_host) flux_host[b] = s->diag_pinned[b];
                const fg_real a = std::fabs(s->diag_pinned[b]);
                worst = (a > worst || a != a) ? a : worst;
            }
            if (!(worst <= o->flux_balance_tol)) {
                fg_set_error("Domain boundary fluxes not balanced, cannot proceed with simulation step.");
                return FG_ERR_FLUX_BALANCE;
            }
            return FG_OK;
        };
        bool deferred = false;      // the CFL maxima are read AFTER the PISO step: its sub-steps were taken on the device (FgDtRule)
        FgPollOut po_cfl = FgPollOut{nullptr, 0};
        if (o->adaptive || first) {
            // one transfer: [0..B) boundary flux balance, [B..2B) max |Minv u| (CFL velocity)
            // both kernels publish straight into the host-pinned diag_pinned (one workgroup per env writes the flux
            // balance; the last workgroup of each env mirrors the max velocity): the read-back is a stream synchronise
            // The host waits on sequence words published behind the results (FgPoll, fg_internal.h): ONE word behind the CFL maxima
            // (fg_publish_max), which the in-order stream also puts behind the flux balances of the kernel in front of it; without the
            // CFL kernel, one word per env behind the flux balance.
            const FgPollOut po = fg_poll_next(&s->poll);
            po_cfl = po;
            // (with the CFL kernel in the step, the guard is computed by workgroup 0 of every env of THAT launch: one kernel, one word)
            if (first && !o->adaptive) { if (int rc = fg_launch_flux_balance(s, bnd, s->diag_pinned, st, FgPollOut{po.seq, po.value})) return rc; }
            if (o->adaptive) {
                // (fp32 build: the maxima and the flux balances travel in the polled result words themselves -- FgPollOut::gran,
                //  words [12 B, 13 B) and [13 B, 14 B), out of reach of the polls of the step -- and are unpacked to where the mirror form
                //  leaves them.  With them the sub-steps are taken on the device by the kernel itself and the host reads the maxima after
                //  the PISO step, for its loop control: no round trip between the CFL kernel and the step -- FgDtRule, FG_DEV_DT=0 = before)
                deferred = po.gran != nullptr && s->dev_dt && s->t_rem_dev != nullptr;
                const FgDtRule rule = deferred ? FgDtRule{s->t_rem_dev, s->dt_dev, (float)o->cfl, (double)o->time_step, first ? 1 : 0}
                                               : FgDtRule{nullptr, nullptr, 0.f, 0.0, 0};
                if (int rc = fg_launch_max_velocity(s, bnd, s->scratch_B + B, st, s->diag_pinned + B, po.seq ? FgPollOut{po.seq + B, po.value, po.gran} : po,
                                                    first ? s->scratch_B : nullptr, first ? s->diag_pinned : nullptr, rule))
                    return rc;
            }
            fg_htrace("maxvel_launched");
            if (!deferred) {
                if (o->adaptive && po.gran) {
                    if (int rc = fg_poll_wait_words(&s->poll, po, 12 * B, first ? 2 * B : B, st)) return rc;
                    for (int b = 0; b < B; ++b) {
                        s->diag_pinned[B + b] = fg_poll_word_float(&s->poll, 12 * B + b);
                        if (first) s->diag_pinned[b] = fg_poll_word_float(&s->poll, 13 * B + b);
                    }
                } else if (int rc = fg_poll_wait(&s->poll, po, o->adaptive ? B : 0, o->adaptive ? 1 : B, st)) return rc;
                fg_htrace("maxvel_poll_done");
                if (first) if (int rc = flux_guard()) return rc;
            }
        }
        if (!deferred)
            for (int b = 0; b < B; ++b) s->dt_pinned[b] = advance(b, o->adaptive ? s->diag_pinned[B + b] : (fg_real)0);
        // PRE hook: advective outflow + flux re-balancing (cylinder_env_base.py:280-300)
        const bool folded = o->outflow_mask && fg_outflow_folds(s, o->outflow_mask);     // small slabs: the update rides in the balance launch
        // the time steps reach the device as kernel arguments of that launch (up to 64 envs) instead of through a copy in front of it
        // (deferred: they are in dt_dev already, written by the CFL kernel)
        const bool dt_by_value = folded && B <= 64 && !deferred;
        if (!dt_by_value && !deferred) FG_HIP_CHECK(hipMemcpyAsync(s->dt_dev, s->dt_pinned, sizeof(fg_real) * B, hipMemcpyHostToDevice, st));
        if (o->outflow_mask) {
            for (int f = 0; f < 2 * s->grid.dims; ++f)
                if ((o->outflow_mask >> f) & 1) FG_REQUIRE(s->grid.fixed[f], FG_ERR_INVALID_ARG, "outflow face is not FIXED");
            if (!folded)
                for (int f = 0; f < 2 * s->grid.dims; ++f)
                    if ((o->outflow_mask >> f) & 1)
                        if (int rc = fg_launch_outflow(s, f, o->outflow_velm[f >> 1], s->dt_dev, st)) return rc;
            fg_real velm3[3] = {o->outflow_velm[0], o->outflow_velm[1], o->outflow_velm[2]};
            if (int rc = fg_launch_balance(s, bnd, o->outflow_mask, (fg_real)0.01 * o->outflow_tol, s->dt_dev, st, folded ? o->outflow_mask : 0, velm3,
                                           dt_by_value ? s->dt_pinned : nullptr))
                return rc;
        }
        int rc = fg_piso_step(s, s->dt_dev, &o->step, stats, stream);
        if (rc == FG_ERR_NOT_CONVERGED) all_ok = 0;
        else if (rc != FG_OK) return rc;
        // dt_pinned must not be rewritten before the H2D copy above has executed: fg_piso_step's solver polls
        // synchronise the stream, so the copy is complete here.
        if (deferred) {
            // the maxima the device took its sub-steps from (published long ago): the host repeats the rule for its loop control, and the
            // flux-balance guard of the sim step's first sub-step is judged here -- the error is the same, only raised after a PISO step
            // on the unbalanced domain instead of before it
            if (int rc2 = fg_poll_wait_words(&s->poll, po_cfl, 12 * B, first ? 2 * B : B, st)) return rc2;
            for (int b = 0; b < B; ++b) {
                s->diag_pinned[B + b] = fg_poll_word_float(&s->poll, 12 * B + b);
                if (first) s->diag_pinned[b] = fg_poll_word_float(&s->poll, 13 * B + b);
            }
            if (first) if (int rc2 = flux_guard()) return rc2;
            for (int b = 0; b < B; ++b) s->dt_pinned[b] = advance(b, s->diag_pinned[B + b]);
        }
        ++substeps;
        first = false;
        if (!o->adaptive) --fixed_left;
        if (substeps >= (o->max_substeps > 0 ? o->max_substeps : 100000)) break;
    }
    fg_htrace("single_step_out");
    for (int i = 0; i < 4; ++i) out[i] = stats[i];
    out[4] = substeps;
    out[5] = all_ok;
    return FG_OK;
}

extern "C" int fg_make_divergence_free(fg_handle s, fg_real tol, int max_iterations, fg_solve_info* info_host,
                                       void* stream) {
    // make_divergence_free (PISOtorch_simulation.py:1320-1429): A := 1, dt := 1, h := u
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    if (int rc = check_bound(s, false)) return rc;
    s->cur_dt = nullptr;
    hipStream_t st = (hipStream_t)stream;
    const size_t BN = (size_t)s->grid.B * s->grid.n;
    std::vector<fg_real> ones(BN, 1.f);
    FG_HIP_CHECK(hipMemcpyAsync(s->rA, ones.data(), sizeof(fg_real) * BN, hipMemcpyHostToDevice, st));
    s->rA_epoch++;
    FG_HIP_CHECK(hipStreamSynchronize(st));
    if (int rc = fg_launch_copy_active(s, nullptr, s->velocity, s->hvec, s->grid.dims, st)) return rc;
    if (int rc = fg_launch_div(s, make_bounds(s, 0), nullptr, s->hvec, s->div, st)) return rc;
    int rc = solve_pressure(s, nullptr, FG_SOLVER_CG, tol, max_iterations, 0, info_host, st);
    if (rc != FG_OK && rc != FG_ERR_NOT_CONVERGED) return rc;
    if (int rc2 = fg_launch_correct(s, nullptr, s->rA, s->hvec, s->pressure, s->vel_result, st)) return rc2;
    if (int rc2 = fg_launch_copy_active(s, nullptr, s->vel_result, s->velocity, s->grid.dims, st)) return rc2;
    return rc;
}

extern "C" int fg_reset_solver_state(fg_handle s, void* stream) {
    FG_REQUIRE(s && s->velocity, FG_ERR_NOT_BOUND, "velocity not bound");
    for (int k = 0; k < 4; ++k) s->jac_hist[k] = FgJacHist{0, 0, 0};   // the sweeps' back-off decides WHICH solver runs: not carried across a reset
    FG_HIP_CHECK(hipMemsetAsync(s->p_result, 0, sizeof(fg_real) * (size_t)s->grid.B * s->grid.n, (hipStream_t)stream));
    return fg_launch_copy_active(s, nullptr, s->velocity, s->vel_result, s->grid.dims, (hipStream_t)stream);
}

// What a handle remembers between solves besides the bound fields and decides which iteration runs: per solve kind the sweeps the
// last Jacobi solve needed (where the next one polls first), the solves still to skip after a failure and the failures in a row
// (fg_jacobi.hip).  get_state / set_state carry these 12 words so that a restored state replays bit for bit.
extern "C" int fg_solver_hints(fg_handle s, int32_t* hints12, int32_t set) {
    FG_REQUIRE(s != nullptr && hints12 != nullptr, FG_ERR_INVALID_ARG, "fg_solver_hints: bad argument");
    for (int k = 0; k < 4; ++k) {
        FgJacHist& H = s->jac_hist[k];
        if (set) { H.sweeps = hints12[3 * k]; H.skip = hints12[3 * k + 1]; H.fails = hints12[3 * k + 2]; }
        else { hints12[3 * k] = H.sweeps; hints12[3 * k + 1] = H.skip; hints12[3 * k + 2] = H.fails; }
    }
    return FG_OK;
}

extern "C" int fg_get_buffer(fg_handle s, int which, fg_real** out_ptr, int64_t* out_count) {
    FG_REQUIRE(s && out_ptr && out_count, FG_ERR_INVALID_ARG, "null argument");
    const int64_t BN = (int64_t)s->grid.B * s->grid.n, d = s->grid.dims;
    switch (which) {
        case FG_BUF_A: *out_ptr = s->A; *out_count = BN; break;
        case FG_BUF_C_OFF: *out_ptr = s->Coff; *out_count = BN * 2 * d; break;
        case FG_BUF_ADV_RHS: *out_ptr = s->adv_rhs; *out_count = BN * d; break;
        case FG_BUF_VEL_RESULT: *out_ptr = s->vel_result; *out_count = BN * d; break;
        case FG_BUF_H: *out_ptr = s->hvec; *out_count = BN * d; break;
        case FG_BUF_DIV: *out_ptr = s->div; *out_count = BN; break;
        case FG_BUF_P_RESULT: *out_ptr = s->p_result; *out_count = BN; break;
        case FG_BUF_SCALAR_RESULT: *out_ptr = s->scal_result; *out_count = BN; break;
        default:
            if (which >= 100 && which < 108) { *out_ptr = s->w[which - 100]; *out_count = BN * d; break; }   // Krylov work vectors (tests)
            FG_REQUIRE(false, FG_ERR_INVALID_ARG, "unknown buffer id");
    }
    return FG_OK;
}

extern "C" int fg_read_buffer(fg_handle s, int which, fg_real* dst, void* stream) {
    fg_real* src = nullptr;
    int64_t count = 0;
    if (int rc = fg_get_buffer(s, which, &src, &count)) return rc;
    FG_REQUIRE(dst, FG_ERR_INVALID_ARG, "null destination");
    FG_HIP_CHECK(hipMemcpyAsync(dst, src, sizeof(fg_real) * count, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FG_OK;
}

// ---- standalone Poisson entry points ------------------------------------------------------------
extern "C" int fg_poisson_apply(fg_handle s, const fg_real* rA, const fg_real* x, fg_real* y, void* stream) {
    FG_REQUIRE(s && rA && x && y, FG_ERR_INVALID_ARG, "null argument");
    return fg_poisson_apply_launch(s, rA, x, y, (hipStream_t)stream);
}

extern "C" int fg_poisson_jacobi(fg_handle s, const fg_real* rA, const fg_real* b, fg_real* x, int n_sweeps, fg_real omega,
                                 void* stream) {
    FG_REQUIRE(s && rA && b && x && n_sweeps >= 0, FG_ERR_INVALID_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    fg_real* buf[2] = {x, s->w[5]};
    for (int i = 0; i < n_sweeps; ++i)
        if (int rc = fg_poisson_jacobi_launch(s, rA, b, buf[i & 1], buf[(i + 1) & 1], omega, st)) return rc;
    if (n_sweeps & 1)
        FG_HIP_CHECK(hipMemcpyAsync(x, s->w[5], sizeof(fg_real) * (size_t)s->grid.B * s->grid.n, hipMemcpyDeviceToDevice, st));
    return FG_OK;
}

extern "C" int fg_poisson_rbgs(fg_handle s, const fg_real* rA, const fg_real* b, fg_real* x, int n_sweeps, fg_real omega,
                               void* stream) {
    FG_REQUIRE(s && rA && b && x && n_sweeps >= 0, FG_ERR_INVALID_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < n_sweeps; ++i) {
        if (int rc = fg_poisson_rbgs_launch(s, rA, b, x, omega, 0, st)) return rc;
        if (int rc = fg_poisson_rbgs_launch(s, rA, b, x, omega, 1, st)) return rc;
    }
    return FG_OK;
}

extern "C" int fg_poisson_cg(fg_handle s, const fg_real* rA, const fg_real* b, fg_real* x, fg_real tol, int max_iterations,
                             int use_x0, fg_solve_info* info_host, void* stream) {
    FG_REQUIRE(s && rA && b && x, FG_ERR_INVALID_ARG, "null argument");
    FgCgArgs a;
    a.rA = rA; a.b = b; a.x = x;
    a.r = s->w[0]; a.p = s->w[1]; a.Ap = s->w[2];
    a.dt = nullptr; a.tol = tol; a.max_iterations = max_iterations; a.use_x0 = use_x0;
    a.reset_steps = 100;
    a.precond = 0;
    a.check_every = (tol > 0.f) ? 16 : (1 << 30);  // tol <= 0: run exactly max_iterations without polling
    return fg_cg_solve(s, a, info_host, (hipStream_t)stream);
}

extern "C" int fg_poisson_fdcg(fg_handle s, const fg_real* rA, const fg_real* b, fg_real* x, fg_real tol, int max_iterations,
                               int use_x0, fg_solve_info* info_host, void* stream) {
    FG_REQUIRE(s && rA && b && x, FG_ERR_INVALID_ARG, "null argument");
    FgCgArgs a;
    a.rA = rA; a.b = b; a.x = x;
    a.r = s->w[0]; a.p = s->w[1]; a.Ap = s->w[2];
    a.dt = nullptr; a.tol = tol; a.max_iterations = max_iterations; a.use_x0 = use_x0;
    a.reset_steps = 0;
    a.precond = 1;
    a.check_every = (tol > 0.f) ? 2 : (1 << 30);
    return fg_cg_solve(s, a, info_host, (hipStream_t)stream);
}

extern "C" int fg_coords_to_transforms(const float* coords, float* transforms, int dims, int nx, int ny, int nz,
                                       void* stream) {
    FG_REQUIRE(coords && transforms && (dims == 2 || dims == 3), FG_ERR_INVALID_ARG, "bad argument");
#if FG_F64
    fg_set_error("fg_coords_to_transforms is an fp32 entry point: use libfluidgym_hip.so");
    return FG_ERR_UNSUPPORTED;
#else
    return fg_metrics_launch(coords, transforms, dims, nx, ny, nz, (hipStream_t)stream);
#endif
}
