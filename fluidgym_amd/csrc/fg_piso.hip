// PISO assembly / correction stencils for gfx950 (wave64, 256-thread workgroups, float4 rows).
//
// One workgroup = one 64-wide x tile of one env; fields are [B, C, (Z,) Y, X] so every global
// access is a 256-byte coalesced row segment.  Matrices are never written as CSR (the reference
// writes value+index arrays, PISO_multiblock_cuda_kernel.cu:3861-3877): the advection matrix C is
// kept as diag (= A) + 2d off-diagonal fields in face order, and the pressure matrix is applied
// matrix-free from rA = 1/A (fg_poisson.hip).
#include "fg_internal.h"
#include "fg_cg.h"
#include "fg_fftcg.h"

namespace {

template <int VEC>
__device__ __forceinline__ fg_real fg_elem_mask(int e, fg_real lo_mask_or_hi, bool is_edge_elem) {
    return is_edge_elem ? lo_mask_or_hi : 1.f;
}

// boundary-slab index of the thread's vector for face axis `a`
template <int DIMS, int VEC>
__device__ __forceinline__ int fg_slab_index(const FgGrid& g, const FgCtx<DIMS, VEC>& c, int a) {
    if (a == 0) return c.k * g.ny + c.j;           // slab [nz, ny, 1]
    if (a == 1) return c.k * g.nx + c.i0;          // slab [nz, 1, nx]
    return c.j * g.nx + c.i0;                      // slab [1, ny, nx]
}
__device__ __forceinline__ int fg_slab_size(const FgGrid& g, int a) {
    return (a == 0) ? g.ny * g.nz : (a == 1) ? g.nx * g.nz : g.nx * g.ny;
}

// ---------------------------------------------------------------------------------------------
// Advection-diffusion matrix (stencil form) + RHS.
//   reference: PISO_build_matrix (PISO_multiblock_cuda_kernel.cu:3616-3880),
//              kPISO_build_advection_RHS (:4296-4400), kPISO_build_scalar_advection_RHS (:4094-4198)
//   diag  = J/dt + sum_f [ s_f F_f/2 + (alpha_P + alpha_N) nu/2 ]     non-prescribed faces
//                + sum_f (1-slip) 2 nu alpha_P                         prescribed faces (:3846)
//   off_f = s_f F_f/2 - (alpha_P + alpha_N) nu/2 ;  row /= J ; A = diag/J
//   rhs_c = [ J q_c/dt + sum_FIXED q_b,c ( -s_f U_b + (1-slip) nu 2 alpha_b ) ]/J + S_c
// Rectilinear grid: U_a = u_a * (J/h_a), alpha_a = (J/h_a)/h_a, boundary transform == adjacent cell.
// ---------------------------------------------------------------------------------------------
// VISC: a per-cell viscosity field (getViscosityBlock, K.cu:1816-1837; Block.setViscosity of the SGS hook, tcf_env.py:441-474): the
// face coefficient becomes (alpha_P nu_P + alpha_N nu_N) / 2 (:3745) and a prescribed face takes the adjacent cell's value
// (getViscosityFixedBoundary, :1840-1843).  A separate instance: the default path is the scalar-nu kernel, bit for bit.
template <int DIMS, int VEC, bool SCALAR, bool VISC = false>
__global__ __launch_bounds__(FG_BLOCK) void k_adv_build(FgGrid g, FgBounds bnd, FgAdvArgs a, int tiles_x,
                                                         int tiles_y, int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    if (a.begin.acc) {   // the BiCGStab solve on this system comes next: the first workgroup of the env prepares its state
        if (fg_xcd_remap(blockIdx.x, gridDim.x) % (unsigned)tiles == 0 && (int)threadIdx.x < a.begin.nc)
            fg_bicg_begin_sys(a.begin, a.dt, c.b * a.begin.nc + threadIdx.x);
    }
    const fg_real dt = a.dt[c.b];
    if (!(dt > 0.f) || !c.valid) return;
    const size_t N = g.n;
    const fg_real* __restrict__ vel = a.vel + (size_t)c.b * DIMS * N;
    const FgMetric<DIMS, VEC> m = fg_metrics<DIMS, VEC>(g, c);
    const fg_real nu = a.nu;
    const fg_real rdt = 1.f / dt;

    fg_real J[VEC], diag[VEC], off[2 * DIMS][VEC];
    fg_real bsum[SCALAR ? 1 : DIMS][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        J[e] = m.hx[e] * m.hy * m.hz;
        diag[e] = J[e] * rdt;
#pragma unroll
        for (int q = 0; q < (SCALAR ? 1 : DIMS); ++q) bsum[q][e] = 0.f;
    }
    FgVec<VEC> u[DIMS];

#pragma unroll
    for (int ax = 0; ax < DIMS; ++ax) {
        FgVec<VEC> lo, hi;
        fg_gather_axis<DIMS, VEC>(vel + ax * N, c, ax, u[ax], lo, hi);
        FgVec<VEC> nu_c, nu_lo, nu_hi;
        if constexpr (VISC) fg_gather_axis<DIMS, VEC>(a.visc + (size_t)c.b * N, c, ax, nu_c, nu_lo, nu_hi);
        const int f_lo = 2 * ax, f_hi = 2 * ax + 1;
        const int slab = fg_slab_index<DIMS, VEC>(g, c, ax);
        const int slab_n = fg_slab_size(g, ax);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            fg_real area, rh_p, rh_lo, rh_hi, mask_lo, mask_hi;
            if (ax == 0) {
                area = m.hy * m.hz;
                rh_p = m.rhx[e];
                rh_lo = (e == 0) ? m.rhx_m : m.rhx[e > 0 ? e - 1 : 0];
                rh_hi = (e == VEC - 1) ? m.rhx_p : m.rhx[e < VEC - 1 ? e + 1 : VEC - 1];
                mask_lo = (e == 0) ? c.mxm : 1.f;
                mask_hi = (e == VEC - 1) ? c.mxp : 1.f;
            } else if (ax == 1) {
                area = m.hx[e] * m.hz;
                rh_p = m.rhy; rh_lo = m.rhy_m; rh_hi = m.rhy_p;
                mask_lo = c.mym; mask_hi = c.myp;
            } else {
                area = m.hx[e] * m.hy;
                rh_p = m.rhz; rh_lo = m.rhz_m; rh_hi = m.rhz_p;
                mask_lo = c.mzm; mask_hi = c.mzp;
            }
            const fg_real Uc = u[ax].v[e] * area;
            const fg_real al_p = area * rh_p;
            const fg_real nu_p = VISC ? nu_c.v[e] : nu;   // (the code below reads nu_p where the scalar kernel read nu)
            // ---- lower face (s = -1)
            if (mask_lo != 0.f) {
                const fg_real ff = -0.25f * (Uc + lo.v[e] * area);
                fg_real visc;
                if constexpr (VISC) visc = 0.5f * (nu_p * al_p + nu_lo.v[e] * (area * rh_lo));
                else visc = 0.5f * nu * (al_p + area * rh_lo);
                diag[e] += ff + visc;
                off[f_lo][e] = ff - visc;
            } else {
                off[f_lo][e] = 0.f;
                const int bi = slab + ((ax == 0) ? 0 : e);
                const fg_real* bv = bnd.vel[f_lo] + (size_t)c.b * DIMS * slab_n;
                const fg_real Ub = bv[ax * slab_n + bi] * area;  // boundary contravariant flux (:1593-1599)
                if constexpr (SCALAR) {
                    const fg_real tb = bnd.scal[f_lo][((size_t)c.b * a.n_scalars + a.channel) * slab_n + bi];
                    const bool dir = bnd.scalar_bc[f_lo] == FG_DIRICHLET;
                    if (dir) diag[e] += 2.f * nu * al_p;
                    bsum[0][e] += tb * Ub + (dir ? tb * nu * 2.f * al_p : tb * nu);  // -T_b*(s U_b), s=-1
                } else {
                    diag[e] += 2.f * nu_p * al_p;
#pragma unroll
                    for (int q = 0; q < DIMS; ++q) {
                        const fg_real ub = bv[q * slab_n + bi];
                        bsum[q][e] += ub * Ub + ub * nu_p * 2.f * al_p;
                    }
                }
            }
            // ---- upper face (s = +1)
            if (mask_hi != 0.f) {
                const fg_real ff = 0.25f * (Uc + hi.v[e] * area);
                fg_real visc;
                if constexpr (VISC) visc = 0.5f * (nu_p * al_p + nu_hi.v[e] * (area * rh_hi));
                else visc = 0.5f * nu * (al_p + area * rh_hi);
                diag[e] += ff + visc;
                off[f_hi][e] = ff - visc;
            } else {
                off[f_hi][e] = 0.f;
                const int bi = slab + ((ax == 0) ? 0 : e);
                const fg_real* bv = bnd.vel[f_hi] + (size_t)c.b * DIMS * slab_n;
                const fg_real Ub = bv[ax * slab_n + bi] * area;
                if constexpr (SCALAR) {
                    const fg_real tb = bnd.scal[f_hi][((size_t)c.b * a.n_scalars + a.channel) * slab_n + bi];
                    const bool dir = bnd.scalar_bc[f_hi] == FG_DIRICHLET;
                    if (dir) diag[e] += 2.f * nu * al_p;
                    bsum[0][e] += -tb * Ub + (dir ? tb * nu * 2.f * al_p : tb * nu);
                } else {
                    diag[e] += 2.f * nu_p * al_p;
#pragma unroll
                    for (int q = 0; q < DIMS; ++q) {
                        const fg_real ub = bv[q * slab_n + bi];
                        bsum[q][e] += -ub * Ub + ub * nu_p * 2.f * al_p;
                    }
                }
            }
        }
    }
    // ---- write A, off-diagonals, RHS
    FgVec<VEC> out;
    fg_real rJ[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { rJ[e] = 1.f / J[e]; out.v[e] = diag[e] * rJ[e]; }
    fg_store<VEC>(a.A + (size_t)c.b * N + c.idx, out);
    if (a.rA) {  // rA = 1/A for the pressure system, same expression as k_reciprocal (saves its pass over A)
        FgVec<VEC> r;
#pragma unroll
        for (int e = 0; e < VEC; ++e) r.v[e] = 1.f / out.v[e];
        fg_store<VEC>(a.rA + (size_t)c.b * N + c.idx, r);
#if !FG_F64
        if constexpr (DIMS == 2 && VEC == 4) {
            // row sums of 1/A for the row-mean preconditioner of the pressure CG (k_fd_rowmean_factor, fg_fdprecond.hip): the 16 lanes
            // of a tile row add up their 64 cells on the VALU (DPP) and the tile's partial goes to its own slot [b][j][x-tile] -- no
            // atomics, so the factorisation sums the partials in a fixed order (bit-reproducible steps)
            if (a.row_part) {
                float v = (r.v[0] + r.v[1]) + (r.v[2] + r.v[3]);
#define FG_DPP16(x, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), ctrl, 0xf, 0xf, false))
                v += FG_DPP16(v, 0xb1);      // quad_perm [1,0,3,2]
                v += FG_DPP16(v, 0x4e);      // quad_perm [2,3,0,1]
                v += FG_DPP16(v, 0x124);     // row_ror:4
                v += FG_DPP16(v, 0x128);     // row_ror:8  -> every lane of the 16-lane row holds the row's sum
#undef FG_DPP16
                if ((threadIdx.x & 15) == 0) a.row_part[((size_t)c.b * g.ny + c.j) * tiles_x + c.i0 / 64] = v;
            }
        }
#endif
    }
#pragma unroll
    for (int f = 0; f < 2 * DIMS; ++f) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) out.v[e] = off[f][e] * rJ[e];
        fg_store<VEC>(a.Coff + ((size_t)c.b * 2 * DIMS + f) * N + c.idx, out);
    }
    if constexpr (SCALAR) {
        const FgVec<VEC> T = fg_load<VEC>(a.scal + (size_t)c.b * a.scal_env_stride + c.idx);
#pragma unroll
        for (int e = 0; e < VEC; ++e) out.v[e] = (J[e] * T.v[e] * rdt + bsum[0][e]) * rJ[e];
        fg_store<VEC>(a.rhs + (size_t)c.b * N + c.idx, out);
    } else {
#pragma unroll
        for (int q = 0; q < DIMS; ++q) {
            FgVec<VEC> S;
            const bool have_s = a.source != nullptr || a.buoy_T != nullptr;
            if (a.buoy_T) {      // folded buoyancy hook (FgAdvArgs): S = factor * T along the buoyancy axis, 0 elsewhere
                if (q == a.buoy_axis) {
                    S = fg_load<VEC>(a.buoy_T + (size_t)c.b * a.buoy_stride + c.idx);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) S.v[e] = a.buoy_factor * S.v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) S.v[e] = 0.f;
                }
                fg_store<VEC>(a.source_w + ((size_t)c.b * DIMS + q) * N + c.idx, S);
            } else if (a.source) S = fg_load<VEC>(a.source + ((size_t)c.b * DIMS + q) * N + c.idx);
            const fg_real F = a.force ? a.force[c.b * DIMS + q] : 0.f;     // uniform body force of the env (the native wall-stress forcing)
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                out.v[e] = (J[e] * u[q].v[e] * rdt + bsum[q][e]) * rJ[e] + ((have_s ? S.v[e] : 0.f) + F);
            fg_store<VEC>(a.rhs + ((size_t)c.b * DIMS + q) * N + c.idx, out);
        }
    }
}

// rA = 1/A  (PISO_build_pressure_matrix reads Adiag, :4839,4871; we keep the reciprocal so the
// Poisson operator needs no divisions)
__global__ __launch_bounds__(FG_BLOCK) void k_reciprocal(const fg_real* __restrict__ A, fg_real* __restrict__ rA,
                                                          const fg_real* __restrict__ dt, int n, int n4) {
    const int b = blockIdx.y;
    if (dt && !(dt[b] > 0.f)) return;
    const fg_real4* src = reinterpret_cast<const fg_real4*>(A + (size_t)b * n);
    fg_real4* dst = reinterpret_cast<fg_real4*>(rA + (size_t)b * n);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const fg_real4 v = src[i];
        dst[i] = make_fg_real4(1.f / v.x, 1.f / v.y, 1.f / v.z, 1.f / v.w);
    }
    if (blockIdx.x == 0) {
        for (int i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) rA[(size_t)b * n + i] = 1.f / A[(size_t)b * n + i];
    }
}

// ---------------------------------------------------------------------------------------------
// H operator: h_c = rA [ rhs_c - sum_f off_f u~_{N_f,c} ]
//   reference PISO_build_pressure_rhs (:5136-5255): rDiag*(u^n_c/dt - H + S_bnd/J + S_c); our
//   velocity RHS already holds u^n_c/dt + S_bnd/J + S_c (same terms, :4314-4387 vs :5165-5249).
// ---------------------------------------------------------------------------------------------
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_h(FgGrid g, const fg_real* __restrict__ dt,
                                                 const fg_real* __restrict__ rA_, const fg_real* __restrict__ Coff,
                                                 const fg_real* __restrict__ rhs, const fg_real* __restrict__ velr,
                                                 fg_real* __restrict__ hvec, FgCgBegin begin, int tiles_x, int tiles_y, int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    // the pressure solve on the right-hand side built from this h comes next: the first workgroup of the env prepares its state
    // (fg_cg.h; here and not in k_div, which ALSO starts the solve -- r = b, x = 0, r.r -- and must find the accumulators zeroed)
    if (begin.acc && fg_xcd_remap(blockIdx.x, gridDim.x) % (unsigned)tiles == 0) fg_cg_begin_env(begin, dt, c.b);
    if (!(dt[c.b] > 0.f) || !c.valid) return;
    const size_t N = g.n;
    const FgVec<VEC> rA = fg_load<VEC>(rA_ + (size_t)c.b * N + c.idx);
    FgVec<VEC> off[2 * DIMS];
#pragma unroll
    for (int f = 0; f < 2 * DIMS; ++f) off[f] = fg_load<VEC>(Coff + ((size_t)c.b * 2 * DIMS + f) * N + c.idx);
#pragma unroll
    for (int q = 0; q < DIMS; ++q) {
        const size_t base = ((size_t)c.b * DIMS + q) * N;
        const FgNbr<DIMS, VEC> u = fg_gather<DIMS, VEC>(velr + base, c);
        const FgVec<VEC> r = fg_load<VEC>(rhs + base + c.idx);
        FgVec<VEC> out;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            fg_real H = off[0].v[e] * u.xm.v[e] + off[1].v[e] * u.xp.v[e] + off[2].v[e] * u.ym.v[e] +
                      off[3].v[e] * u.yp.v[e];
            if constexpr (DIMS == 3) H += off[4].v[e] * u.zm.v[e] + off[5].v[e] * u.zp.v[e];
            out.v[e] = rA.v[e] * (r.v[e] - H);
        }
        fg_store<VEC>(hvec + base + c.idx, out);
    }
}

// ---------------------------------------------------------------------------------------------
// div of the fluxes of h: b = sum_a F_{a+} - F_{a-},  F = (U_P + U_N)/2, prescribed faces use the
// boundary-velocity flux (computeFluxesNDLoop :1567-1645; k_computePressureRHSdivergenceFromFlux
// :5389-5434; timeStepNorm = false)
// ---------------------------------------------------------------------------------------------
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_div(FgGrid g, FgBounds bnd, const fg_real* __restrict__ dt,
                                                   const fg_real* __restrict__ hvec, fg_real* __restrict__ div, FgCgStart start,
                                                   int tiles_x, int tiles_y, int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    if (dt && !(dt[c.b] > 0.f)) return;
    if (!c.valid && !start.acc) return;
    const size_t N = g.n;
    const FgMetric<DIMS, VEC> m = fg_metrics<DIMS, VEC>(g, c);
    fg_real acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ax = 0; ax < DIMS; ++ax) {
        FgVec<VEC> ctr, lo, hi;
        fg_gather_axis<DIMS, VEC>(hvec + ((size_t)c.b * DIMS + ax) * N, c, ax, ctr, lo, hi);
        const int slab = fg_slab_index<DIMS, VEC>(g, c, ax);
        const int slab_n = fg_slab_size(g, ax);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            fg_real area, mask_lo, mask_hi;
            if (ax == 0) { area = m.hy * m.hz; mask_lo = (e == 0) ? c.mxm : 1.f; mask_hi = (e == VEC - 1) ? c.mxp : 1.f; }
            else if (ax == 1) { area = m.hx[e] * m.hz; mask_lo = c.mym; mask_hi = c.myp; }
            else { area = m.hx[e] * m.hy; mask_lo = c.mzm; mask_hi = c.mzp; }
            const int bi = slab + ((ax == 0) ? 0 : e);
            fg_real F_hi, F_lo;
            if (mask_hi != 0.f) F_hi = 0.5f * (ctr.v[e] + hi.v[e]) * area;
            else F_hi = bnd.vel[2 * ax + 1][((size_t)c.b * DIMS + ax) * slab_n + bi] * area;
            if (mask_lo != 0.f) F_lo = 0.5f * (ctr.v[e] + lo.v[e]) * area;
            else F_lo = bnd.vel[2 * ax][((size_t)c.b * DIMS + ax) * slab_n + bi] * area;
            acc[e] += F_hi - F_lo;
        }
    }
    FgVec<VEC> out;
#pragma unroll
    for (int e = 0; e < VEC; ++e) out.v[e] = acc[e];
    if (c.valid) fg_store<VEC>(div + (size_t)c.b * N + c.idx, out);
    if (start.acc) {
        // the start of the CG that follows from the zero vector (FgCgStart, fg_cg.h): r = b, x = 0, r.r -- k_cg_residual's arithmetic
        __shared__ fg_real lds[4];
        fg_real part[1] = {0.f};
        if (c.valid) {
            FgVec<VEC> z;
#pragma unroll
            for (int e = 0; e < VEC; ++e) { z.v[e] = 0.f; part[0] += out.v[e] * out.v[e]; }
            fg_store<VEC>(start.x + (size_t)c.b * N + c.idx, z);
            fg_store<VEC>(start.r + (size_t)c.b * N + c.idx, out);
        }
        fg_block_sum<1>(part, lds);
        if (threadIdx.x == 0) {
            const unsigned tile = fg_xcd_remap(blockIdx.x, gridDim.x) % tiles;
            fg_acc_add(fg_acc_ptr(start.acc, c.b, 0), start.ns, tile, (double)part[0]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// u_c = h_c - rA (grad p)_c ; grad: central difference * 1/2, one-sided * 1 at a prescribed face,
// times Minv (PISO_update_velocity :5962-5995; getPressureGradient :816-849)
// ---------------------------------------------------------------------------------------------
template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_correct(FgGrid g, const fg_real* __restrict__ dt,
                                                       const fg_real* __restrict__ rA_, const fg_real* __restrict__ hvec,
                                                       const fg_real* __restrict__ p, fg_real* __restrict__ vel_out,
                                                       fg_real* __restrict__ vel_copy, FgMeanRef mean, FgLazyRef lazy, int tiles_x, int tiles_y, int tiles) {
    // vel_copy (optional): the block velocity of active envs, written alongside the result by the last corrector of a step
    // (CopyVelocityResultToBlocks, PISOtorch_simulation.py:1974, without a pass of its own)
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    if ((dt && !(dt[c.b] > 0.f)) || !c.valid) return;
    const size_t N = g.n;
    const FgMetric<DIMS, VEC> m = fg_metrics<DIMS, VEC>(g, c);
    const FgVec<VEC> rA = fg_load<VEC>(rA_ + (size_t)c.b * N + c.idx);
    FgNbr<DIMS, VEC> P = fg_gather<DIMS, VEC>(p + (size_t)c.b * N, c);
    if (lazy.lazy) {
        // `p` is z_0 of the fused CG: this env's pressure is alpha z (or zero: it stopped at its start vector) -- FgLazyRef
        const bool on = lazy.lazy[c.b] == 1;
        const fg_real sc = on ? (fg_real)lazy.alpha[2 * c.b] : (fg_real)0;
        auto scale = [&](FgVec<VEC>& v) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = on ? sc * v.v[e] : (fg_real)0;
        };
        scale(P.c); scale(P.xm); scale(P.xp); scale(P.ym); scale(P.yp);
        if constexpr (DIMS == 3) { scale(P.zm); scale(P.zp); }
        if (lazy.p_res) fg_store<VEC>(lazy.p_res + (size_t)c.b * N + c.idx, P.c);
    }
    if (mean.sums) {
        // p - mean(p) to the block (PISOtorch_simulation.py:1922-1925, 1953): the solver's update kernels summed the iterate they
        // left (FgMeanRef, fg_internal.h); an env that ended on its first iterate from zero has sum(x) = alpha sum(z)
        const int used = mean.info[c.b].used_iterations;
        const bool first_only = mean.lazy && mean.lazy[c.b] == 1;
        const fg_real mu = first_only ? (fg_real)(mean.alpha[2 * c.b] * acc_ld(mean.sums + (2 * c.b + 1)) / (double)g.n)
                           : used < 0 ? (fg_real)0 : (fg_real)(acc_ld(mean.sums + (2 * c.b + (used & 1))) / (double)g.n);
        FgVec<VEC> pc;
#pragma unroll
        for (int e = 0; e < VEC; ++e) pc.v[e] = P.c.v[e] - mu;
        fg_store<VEC>(mean.p_copy + (size_t)c.b * N + c.idx, pc);
    }
#pragma unroll
    for (int q = 0; q < DIMS; ++q) {
        const size_t base = ((size_t)c.b * DIMS + q) * N;
        const FgVec<VEC> h = fg_load<VEC>(hvec + base + c.idx);
        FgVec<VEC> out;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            fg_real lo, hi, fac, rh;
            if (q == 0) {
                lo = P.xm.v[e]; hi = P.xp.v[e]; rh = m.rhx[e];
                const fg_real ml = (e == 0) ? c.mxm : 1.f, mh = (e == VEC - 1) ? c.mxp : 1.f;
                fac = (ml == 0.f || mh == 0.f) ? 1.f : 0.5f;
            } else if (q == 1) {
                lo = P.ym.v[e]; hi = P.yp.v[e]; rh = m.rhy;
                fac = (c.mym == 0.f || c.myp == 0.f) ? 1.f : 0.5f;
            } else {
                lo = P.zm.v[e]; hi = P.zp.v[e]; rh = m.rhz;
                fac = (c.mzm == 0.f || c.mzp == 0.f) ? 1.f : 0.5f;
            }
            out.v[e] = h.v[e] - rA.v[e] * ((hi - lo) * fac * rh);
        }
        fg_store<VEC>(vel_out + base + c.idx, out);
        if (vel_copy) fg_store<VEC>(vel_copy + base + c.idx, out);
    }
}

// ---------------------------------------------------------------------------------------------
// CFL velocity: max over components of |u_c / h_c| over cells and FIXED boundaries
// (Domain::getMaxVelocity(withBounds, computational), domain_structs.cpp:1360-1366, 1580-1611).
// Non-negative floats order like their bit patterns, so the reduction finishes with one integer
// atomicMax per workgroup.
// ---------------------------------------------------------------------------------------------
// Per-env maximum across workgroups: atomicMax on the (non-negative) fg_real bit pattern; the workgroup that arrives last
// (per-env arrival counter, zeroed with out_B) copies the final value into the host-pinned mirror, so the host reads the
// CFL velocity after a stream synchronise without a device-to-host copy.  mirror_B == nullptr: plain atomicMax.
// (non-negative reals order like their bit patterns: int for fp32, long long for the fp64 build)
#if FG_F64
typedef long long fg_bits;
__device__ __forceinline__ fg_bits fg_real_bits(fg_real v) { return __double_as_longlong(v); }
__device__ __forceinline__ fg_real fg_bits_real(fg_bits b) { return __longlong_as_double(b); }
#else
typedef int fg_bits;
__device__ __forceinline__ fg_bits fg_real_bits(fg_real v) { return __float_as_int(v); }
__device__ __forceinline__ fg_real fg_bits_real(fg_bits b) { return __int_as_float(b); }
#endif
// sum of FIXED-boundary contravariant fluxes, lower faces negated
// (Domain::GetGlobalFluxBalance, domain_structs.cpp:2476-2509).  One workgroup per env, fp64 sum; the result is valid in thread 0.
template <int DIMS>
__device__ __forceinline__ double fg_flux_balance_block(const FgGrid& g, const FgBounds& bnd, int b) {
    double acc = 0.0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int f = 0; f < 2 * DIMS; ++f) {
        if (!g.fixed[f]) continue;
        const int ax = f >> 1;
        const int slab_n = fg_slab_size(g, ax);
        const double sgn = (f & 1) ? 1.0 : -1.0;
        // slab index s = i0 + n0 * i1 over the two tangential axes; waves take rows i1, lanes run along i0: no integer
        // division per cell and the loads of a row are independent (the s % n, s / n form cost 26 us at 128 x 64 x 64)
        const int n0 = (ax == 0) ? g.ny : g.nx;
        const int n1 = (DIMS == 3) ? ((ax == 2) ? g.ny : g.nz) : 1;
        const fg_real* __restrict__ h0 = (ax == 0) ? g.h[1] : g.h[0];
        const fg_real* __restrict__ h1 = (DIMS == 3) ? ((ax == 2) ? g.h[1] : g.h[2]) : nullptr;
        const fg_real* __restrict__ v = bnd.vel[f] + ((size_t)b * DIMS + ax) * slab_n;
        // rows in groups of four per wave with all their loads requested together: one workgroup per env walks the whole slab, so
        // the loop is a latency chain (TCF 128 x 64 x 64: 31 us with one row in flight per wave; the sum order per thread -- row by
        // row, products in fp32, sums in fp64 -- is unchanged)
        constexpr int RU = 4, WV = FG_BLOCK / 64;
        for (int i1b = wave; i1b < n1; i1b += WV * RU) {
            for (int i0 = lane; i0 < n0; i0 += 64) {
                fg_real val[RU], a1[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int i1 = i1b + u * WV;
                    const bool ok = i1 < n1;
                    a1[u] = ok ? (h1 ? h1[i1] : (fg_real)1) : (fg_real)0;
                    val[u] = ok ? v[(size_t)i1 * n0 + i0] : (fg_real)0;
                }
                const fg_real w0 = h0[i0];
#pragma unroll
                for (int u = 0; u < RU; ++u) acc += sgn * (double)(val[u] * (w0 * a1[u]));
            }
        }
    }
    __shared__ double lds_fb[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) lds_fb[threadIdx.x >> 6] = acc;
    __syncthreads();
    return lds_fb[0] + lds_fb[1] + lds_fb[2] + lds_fb[3];
}
// Called by the whole FIRST WAVE of the workgroup (lane 0 carries the workgroup's maximum).
// flux_B / flux_mirror (optional): the flux balances workgroup 0 of every env left in flux_B (fg_flux_balance_block, in the same launch)
// are mirrored with the maxima.
__device__ __forceinline__ void fg_publish_max(fg_real* out_B, int32_t* done_B, fg_real* mirror_B, int b, fg_real mx, const FgPollOut& poll, int B,
                                               fg_real* flux_B = nullptr, fg_real* flux_mirror = nullptr, const FgDtRule& rule = FgDtRule{nullptr, nullptr, 0.f, 0.0, 0}) {
    const int lane = threadIdx.x & 63;
    int last_of_all = 0;
    if (lane == 0) {
        if (!mirror_B) {
            atomicMax(reinterpret_cast<fg_bits*>(out_B) + b, fg_real_bits(mx));
        } else {
            // Order "my maximum is in" before "I have arrived" without a fence: the RETURNING atomicMax is consumed (the wave
            // waits for its result, i.e. until the device-scope atomic has been performed) before the arrival counter is bumped.
            // An agent-scope __threadfence() here writes back L2 in each of the 2048 workgroups: 14 us -> 45 us for this kernel.
            fg_bits prev = atomicMax(reinterpret_cast<fg_bits*>(out_B) + b, fg_real_bits(mx));
#if FG_F64
            asm volatile("" ::"v"((int)(prev >> 32)), "v"((int)prev));
#else
            asm volatile("" ::"v"(prev));
#endif
            // the last workgroup of the env bumps the counter of finished envs (behind the per-env counters: done_B[B])
            if (atomicAdd(done_B + b, 1) == (int)gridDim.x - 1) last_of_all = (atomicAdd(done_B + B, 1) == B - 1);
        }
    }
    if (!mirror_B) return;
    // The workgroup that finishes the LAST env hands everything to the host: atomic reads of the final values (which also leave the
    // maxima and every counter zeroed for the next launch: no memset in front of it), the host-pinned mirror, and ONE sequence
    // word behind it -- one L2 write-back at the very end of the kernel (64 of them, one per env, while the other workgroups
    // were still streaming, cost 10 us: fg_poll_publish).
    if (__shfl(last_of_all, 0, 64)) {
#if !FG_F64
        if (poll.gran) {
            // result words (fg_internal.h FgPollOut): the maxima as words [12 B, 13 B), the flux balances as [13 B, 14 B) -- no mirror, no release,
            // no write-back of this XCD's L2 at the end of the kernel
            for (int e = lane; e < B; e += 64) {
                const fg_bits mbits = __hip_atomic_exchange(reinterpret_cast<fg_bits*>(out_B) + e, (fg_bits)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (rule.t_rem) {
                    // the sub-step of env e (FgDtRule): fg_single_step's host loop statement by statement, in doubles
                    double tr = rule.reset ? rule.time_step : rule.t_rem[e];
                    float ts = 0.f;
                    if (tr > 0.0 && !(fabs(tr) <= 1e-8)) {
                        const double mv = (double)__uint_as_float(mbits);
                        const double max_ts = (fabs(mv) <= 1e-8) ? tr : (double)rule.cfl / mv;
                        const double tsd = (max_ts >= tr) ? tr : tr / (double)(long long)ceil(tr / max_ts);
                        tr -= tsd;
                        ts = (float)tsd;
                    }
                    rule.t_rem[e] = tr;
                    rule.dt_out[e] = ts;
                }
                // (words [12 B, 14 B): nothing else the step polls reaches them, so the host may read them after the PISO step)
                fg_poll_publish_word(poll, 12 * B + e, mbits);
                if (flux_B) fg_poll_publish_word(poll, 13 * B + e, __hip_atomic_exchange(reinterpret_cast<fg_bits*>(flux_B) + e, (fg_bits)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                atomicExch(done_B + e, 0);
            }
            if (lane == 0) atomicExch(done_B + B, 0);
            return;
        }
#endif
        for (int e = lane; e < B; e += 64) {
            mirror_B[e] = fg_bits_real(__hip_atomic_exchange(reinterpret_cast<fg_bits*>(out_B) + e, (fg_bits)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (flux_B) flux_mirror[e] = fg_bits_real(__hip_atomic_exchange(reinterpret_cast<fg_bits*>(flux_B) + e, (fg_bits)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            atomicExch(done_B + e, 0);
        }
        if (lane == 0) {
            atomicExch(done_B + B, 0);
            fg_poll_publish(poll, 0);
        }
    }
}

template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_max_velocity(FgGrid g, FgBounds bnd, const fg_real* __restrict__ vel,
                                                            fg_real* __restrict__ out_B, int32_t* __restrict__ done_B,
                                                            fg_real* __restrict__ mirror_B, FgPollOut poll, fg_real* flux_B, fg_real* flux_mirror, FgDtRule rule) {
    const int b = blockIdx.y;
    const size_t N = g.n;
    fg_real mx = 0.f;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < g.n; idx += gridDim.x * blockDim.x) {
        const int i = idx % g.nx;
        const int j = (idx / g.nx) % g.ny;
        const int k = idx / (g.nx * g.ny);
        const fg_real rh[3] = {g.rh[0][i], g.rh[1][j], DIMS == 3 ? g.rh[2][k] : 1.f};
#pragma unroll
        for (int q = 0; q < DIMS; ++q) mx = FG_FMAX(mx, FG_FABS(vel[((size_t)b * DIMS + q) * N + idx] * rh[q]));
    }
    {
        // boundary slabs: spread over ALL workgroups of the env.  With workgroup 0 scanning them alone the kernel's
        // duration was that one workgroup's chain of dependent loads: 14 us in 2-D (256-cell slabs) and 99 us at
        // 128 x 64 x 64 (8192-cell slabs), for a reduction that otherwise takes a few microseconds.
#pragma unroll
        for (int f = 0; f < 2 * DIMS; ++f) {
            if (!g.fixed[f]) continue;
            const int ax = f >> 1;
            const int slab_n = fg_slab_size(g, ax);
            const int edge = (f & 1) ? ((ax == 0) ? g.nx - 1 : (ax == 1) ? g.ny - 1 : g.nz - 1) : 0;
            for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < slab_n; s += gridDim.x * blockDim.x) {
                int i, j, k;
                if (ax == 0) { i = edge; j = s % g.ny; k = s / g.ny; }
                else if (ax == 1) { j = edge; i = s % g.nx; k = s / g.nx; }
                else { k = edge; i = s % g.nx; j = s / g.nx; }
                const fg_real rh[3] = {g.rh[0][i], g.rh[1][j], DIMS == 3 ? g.rh[2][k] : 1.f};
#pragma unroll
                for (int q = 0; q < DIMS; ++q)
                    mx = FG_FMAX(mx, FG_FABS(bnd.vel[f][((size_t)b * DIMS + q) * slab_n + s] * rh[q]));
            }
        }
    }
    __shared__ fg_real lds[4];
    mx = fg_wave_max(mx);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (flux_B && blockIdx.x == 0) {
        // the flux-balance guard of the step rides in this launch: workgroup 0 of the env computes it exactly as k_flux_balance does and
        // leaves it in flux_B with a RETURNING atomic, consumed before this workgroup's arrival below (fg_publish_max: the same ordering
        // as for the maxima); the workgroup that finishes the last env mirrors it to the host with the maxima
        const double total = fg_flux_balance_block<DIMS>(g, bnd, b);
        if (threadIdx.x == 0) {
            fg_bits prevf = __hip_atomic_exchange(reinterpret_cast<fg_bits*>(flux_B) + b, fg_real_bits((fg_real)total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if FG_F64
            asm volatile("" ::"v"((int)(prevf >> 32)), "v"((int)prevf));
#else
            asm volatile("" ::"v"(prevf));
#endif
        }
    }
    if (threadIdx.x < 64) {
        mx = FG_FMAX(FG_FMAX(lds[0], lds[1]), FG_FMAX(lds[2], lds[3]));
        fg_publish_max(out_B, done_B, mirror_B, b, mx, poll, (int)gridDim.y, flux_B, flux_mirror, rule);
    }
}

// Row-wise fg_real4 variant (nx % 4 == 0): a wave walks whole rows, so there is no per-element integer division, the
// x metrics come as one fg_real4 per lane and the y/z metrics are wave-uniform (the element-wise form above spent
// 14 us on 16.8 MB at B = 64, 256 x 128).  Workgroup 0 of each env also scans the boundary slabs, as above.
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_max_velocity_rows(FgGrid g, FgBounds bnd, const fg_real* __restrict__ vel,
                                                                 fg_real* __restrict__ out_B, int32_t* __restrict__ done_B,
                                                                 fg_real* __restrict__ mirror_B, int rows_per_block, FgPollOut poll,
                                                                 fg_real* flux_B, fg_real* flux_mirror, FgDtRule rule) {
    const int b = blockIdx.y;
    const size_t N = g.n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows = g.ny * g.nz, nx4 = g.nx >> 2;
    const int r_end = min(rows, (int)(blockIdx.x + 1) * rows_per_block);
    fg_real mx = 0.f;
    for (int r = blockIdx.x * rows_per_block + wave; r < r_end; r += FG_BLOCK / 64) {
        const int k = r / g.ny, j = r - k * g.ny;
        const fg_real rhy = g.rh[1][j], rhz = (DIMS == 3) ? g.rh[2][k] : 1.f;
        for (int i4 = lane; i4 < nx4; i4 += 64) {
            const fg_real4 rhx = *reinterpret_cast<const fg_real4*>(g.rh[0] + 4 * i4);
            const size_t o = (size_t)r * g.nx + 4 * i4;
            const fg_real4 u = *reinterpret_cast<const fg_real4*>(vel + ((size_t)b * DIMS + 0) * N + o);
            const fg_real4 v = *reinterpret_cast<const fg_real4*>(vel + ((size_t)b * DIMS + 1) * N + o);
            mx = FG_FMAX(mx, FG_FMAX(FG_FMAX(FG_FABS(u.x * rhx.x), FG_FABS(u.y * rhx.y)), FG_FMAX(FG_FABS(u.z * rhx.z), FG_FABS(u.w * rhx.w))));
            mx = FG_FMAX(mx, rhy * FG_FMAX(FG_FMAX(FG_FABS(v.x), FG_FABS(v.y)), FG_FMAX(FG_FABS(v.z), FG_FABS(v.w))));
            if constexpr (DIMS == 3) {
                const fg_real4 w = *reinterpret_cast<const fg_real4*>(vel + ((size_t)b * DIMS + 2) * N + o);
                mx = FG_FMAX(mx, rhz * FG_FMAX(FG_FMAX(FG_FABS(w.x), FG_FABS(w.y)), FG_FMAX(FG_FABS(w.z), FG_FABS(w.w))));
            }
        }
    }
    {
        // boundary slabs: spread over ALL workgroups of the env.  With workgroup 0 scanning them alone the kernel's
        // duration was that one workgroup's chain of dependent loads: 14 us in 2-D (256-cell slabs) and 99 us at
        // 128 x 64 x 64 (8192-cell slabs), for a reduction that otherwise takes a few microseconds.
#pragma unroll
        for (int f = 0; f < 2 * DIMS; ++f) {
            if (!g.fixed[f]) continue;
            const int ax = f >> 1;
            const int slab_n = fg_slab_size(g, ax);
            const int edge = (f & 1) ? ((ax == 0) ? g.nx - 1 : (ax == 1) ? g.ny - 1 : g.nz - 1) : 0;
            for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < slab_n; s += gridDim.x * blockDim.x) {
                int i, j, k;
                if (ax == 0) { i = edge; j = s % g.ny; k = s / g.ny; }
                else if (ax == 1) { j = edge; i = s % g.nx; k = s / g.nx; }
                else { k = edge; i = s % g.nx; j = s / g.nx; }
                const fg_real rh[3] = {g.rh[0][i], g.rh[1][j], DIMS == 3 ? g.rh[2][k] : 1.f};
#pragma unroll
                for (int q = 0; q < DIMS; ++q)
                    mx = FG_FMAX(mx, FG_FABS(bnd.vel[f][((size_t)b * DIMS + q) * slab_n + s] * rh[q]));
            }
        }
    }
    __shared__ fg_real lds[4];
    mx = fg_wave_max(mx);
    if (lane == 0) lds[wave] = mx;
    __syncthreads();
    if (flux_B && blockIdx.x == 0) {
        // the flux-balance guard of the step rides in this launch: workgroup 0 of the env computes it exactly as k_flux_balance does and
        // leaves it in flux_B with a RETURNING atomic, consumed before this workgroup's arrival below (fg_publish_max: the same ordering
        // as for the maxima); the workgroup that finishes the last env mirrors it to the host with the maxima
        const double total = fg_flux_balance_block<DIMS>(g, bnd, b);
        if (threadIdx.x == 0) {
            fg_bits prevf = __hip_atomic_exchange(reinterpret_cast<fg_bits*>(flux_B) + b, fg_real_bits((fg_real)total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if FG_F64
            asm volatile("" ::"v"((int)(prevf >> 32)), "v"((int)prevf));
#else
            asm volatile("" ::"v"(prevf));
#endif
        }
    }
    if (threadIdx.x < 64) {
        mx = FG_FMAX(FG_FMAX(lds[0], lds[1]), FG_FMAX(lds[2], lds[3]));
        fg_publish_max(out_B, done_B, mirror_B, b, mx, poll, (int)gridDim.y, flux_B, flux_mirror, rule);
    }
}

template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_flux_balance(FgGrid g, FgBounds bnd, fg_real* __restrict__ out_B, FgPollOut poll) {
    const int b = blockIdx.x;
    const double total = fg_flux_balance_block<DIMS>(g, bnd, b);
    if (threadIdx.x == 0) {
        out_B[b] = (fg_real)total;
        fg_poll_publish(poll, b);      // (out_B may be host-pinned: the host then spins on this word, fg_single_step)
    }
}

// dst = src for active envs (Copy*ResultTo/FromBlocks, :6558-6746); fg_real4 grid-stride rows
__global__ __launch_bounds__(FG_BLOCK) void k_copy_active(const fg_real* __restrict__ dt, const fg_real* __restrict__ src,
                                                           fg_real* __restrict__ dst, long per_env) {
    const int b = blockIdx.y;
    if (dt && !(dt[b] > 0.f)) return;
    const fg_real* s = src + (size_t)b * per_env;
    fg_real* d = dst + (size_t)b * per_env;
    const long n4 = (per_env % 4 == 0) ? per_env / 4 : 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        reinterpret_cast<fg_real4*>(d)[i] = reinterpret_cast<const fg_real4*>(s)[i];
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < per_env; i += blockDim.x) d[i] = s[i];
}

// RBC buoyancy hook fused natively: source[axis] = factor * T, other components 0
// (rbc_env_base.py:285-297: velocitySource = cat([0, T*buoyancy_factor(, 0)]))
__global__ __launch_bounds__(FG_BLOCK) void k_buoyancy(const fg_real* __restrict__ dt, const fg_real* __restrict__ T,
                                                        long t_env_stride, fg_real* __restrict__ source, int dims,
                                                        int n, int axis, fg_real factor) {
    const int b = blockIdx.y;
    if (dt && !(dt[b] > 0.f)) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const fg_real t = T[(size_t)b * t_env_stride + i];
        for (int q = 0; q < dims; ++q) source[((size_t)b * dims + q) * n + i] = (q == axis) ? factor * t : 0.f;
    }
}

// p -= mean(p) per env (PISOtorch_simulation.py:1817-1820), written to pressureResult and the block
__global__ __launch_bounds__(FG_BLOCK) void k_sum_env(const fg_real* __restrict__ dt, const fg_real* __restrict__ p,
                                                       FgDacc* __restrict__ sums, int n) {
    const int b = blockIdx.y;
    if (dt && !(dt[b] > 0.f)) return;
    const fg_real* __restrict__ pb = p + (size_t)b * n;
    fg_real acc = 0.f;
    const int n4 = ((n & 3) == 0 && (reinterpret_cast<size_t>(pb) & 15) == 0) ? n >> 2 : 0;  // fg_real4 body, scalar tail
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const fg_real4 v = reinterpret_cast<const fg_real4*>(pb)[i];
        acc += (v.x + v.y) + (v.z + v.w);
    }
    for (int i = n4 * 4 + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += pb[i];
    __shared__ fg_real lds[4];
    fg_real v[1] = {acc};
    fg_block_sum<1>(v, lds);
    if (threadIdx.x == 0) acc_add(sums + b, (double)v[0]);
}
__global__ __launch_bounds__(FG_BLOCK) void k_sub_mean(const fg_real* __restrict__ dt, fg_real* __restrict__ p,
                                                        fg_real* __restrict__ p_copy, const FgDacc* __restrict__ sums,
                                                        int n) {
    const int b = blockIdx.y;
    if (dt && !(dt[b] > 0.f)) return;
    const fg_real mean = (fg_real)(acc_ld(sums + b) / (double)n);
    fg_real* __restrict__ pb = p + (size_t)b * n;
    fg_real* __restrict__ cb = p_copy ? p_copy + (size_t)b * n : nullptr;
    const bool al = (n & 3) == 0 && (reinterpret_cast<size_t>(pb) & 15) == 0 && (!cb || (reinterpret_cast<size_t>(cb) & 15) == 0);
    const int n4 = al ? n >> 2 : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        fg_real4 v = reinterpret_cast<fg_real4*>(pb)[i];
        v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
        reinterpret_cast<fg_real4*>(pb)[i] = v;
        if (cb) reinterpret_cast<fg_real4*>(cb)[i] = v;
    }
    for (int i = n4 * 4 + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const fg_real v = pb[i] - mean;
        pb[i] = v;
        if (cb) cb[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Convective outflow boundary (update_advective_boundaries, PISOtorch_simulation.py:282-389):
//   t = 1 - 1/(1 + 2 dt (Minv_row_n . u_m)),  phi_b <- phi_b - t (phi_b - phi_cell)
// for the velocity (all components) and every passive scalar channel of one FIXED face.
// Rectilinear grid: Minv_row_n . u_m = u_m[axis] / h_axis(boundary cell).
// ---------------------------------------------------------------------------------------------
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_outflow(FgGrid g, int face, fg_real velm_axis, const fg_real* __restrict__ dt,
                                                       const fg_real* __restrict__ vel, fg_real* __restrict__ bvel,
                                                       const fg_real* __restrict__ scal, fg_real* __restrict__ bscal,
                                                       int n_scalars) {
    const int b = blockIdx.y;
    const fg_real dtb = dt[b];
    if (!(dtb > 0.f)) return;
    const int ax = face >> 1;
    const int slab_n = fg_slab_size(g, ax);
    const int edge = (face & 1) ? ((ax == 0) ? g.nx - 1 : (ax == 1) ? g.ny - 1 : g.nz - 1) : 0;
    const fg_real tcoef = 1.f - 1.f / (1.f + 2.f * dtb * velm_axis * g.rh[ax][edge]);
    const size_t N = g.n;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < slab_n; s += gridDim.x * blockDim.x) {
        int i, j, k;
        if (ax == 0) { i = edge; j = s % g.ny; k = s / g.ny; }
        else if (ax == 1) { j = edge; i = s % g.nx; k = s / g.nx; }
        else { k = edge; i = s % g.nx; j = s / g.nx; }
        const size_t cell = ((size_t)k * g.ny + j) * g.nx + i;
#pragma unroll
        for (int q = 0; q < DIMS; ++q) {
            fg_real* pb = bvel + ((size_t)b * DIMS + q) * slab_n + s;
            const fg_real vb = *pb;
            *pb = vb - tcoef * (vb - vel[((size_t)b * DIMS + q) * N + cell]);
        }
        for (int ch = 0; ch < n_scalars; ++ch) {
            fg_real* pb = bscal + ((size_t)b * n_scalars + ch) * slab_n + s;
            const fg_real sb = *pb;
            *pb = sb - tcoef * (sb - scal[((size_t)b * n_scalars + ch) * N + cell]);
        }
    }
}

// balance_boundary_fluxes (PISOtorch_simulation.py:188-224): per env, if |flux_fixed + flux_free| exceeds
// atol scale the whole velocity of the free faces by -flux_fixed / flux_free.  One workgroup per env.
// fold (optional, fold.mask != 0): the convective update of the masked faces (k_outflow's body, same expressions) by this env's
// workgroup in front of the balance -- fg_single_step's PRE hook as ONE launch when the slabs are small.
struct FgOutflowFold {
    // dt_n > 0: the time steps of the substep arrive BY VALUE (dt_v, batches of up to 64 envs) and this kernel -- the first of the
    // substep -- stores them into the device array every later kernel reads: no host-to-device copy in front of it
    int dt_n; fg_real dt_v[64]; fg_real* dt_out;
    int mask;                       // faces updated first (0: none)
    fg_real velm[3];                // mean outflow velocity per axis
    const fg_real* vel; const fg_real* scal;
    fg_real* bscal[6]; int nsc[6];
};
struct FgBvelRw { fg_real* p[6]; };      // the writable boundary velocities, by value (no device-side pointer table: re-binding a boundary is host-only)
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_balance_fluxes(FgGrid g, FgBounds bnd, FgBvelRw bvel_rw_, int free_mask,
                                                              fg_real atol, const fg_real* __restrict__ dt, FgOutflowFold fold) {
    const int b = blockIdx.x;
    const fg_real dtb = fold.dt_n ? fold.dt_v[b] : (dt ? dt[b] : (fg_real)1);
    if (fold.dt_n && threadIdx.x == 0) fold.dt_out[b] = dtb;
    if ((dt || fold.dt_n) && !(dtb > 0.f)) return;
    if (fold.mask) {
        for (int face = 0; face < 2 * DIMS; ++face) {
            if (!((fold.mask >> face) & 1)) continue;
            const int ax = face >> 1;
            const int slab_n = fg_slab_size(g, ax);
            const int edge = (face & 1) ? ((ax == 0) ? g.nx - 1 : (ax == 1) ? g.ny - 1 : g.nz - 1) : 0;
            const fg_real tcoef = 1.f - 1.f / (1.f + 2.f * dtb * fold.velm[ax] * g.rh[ax][edge]);
            const size_t N = g.n;
            fg_real* bvel = bvel_rw_.p[face];
            for (int s = threadIdx.x; s < slab_n; s += blockDim.x) {
                int i, j, k;
                if (ax == 0) { i = edge; j = s % g.ny; k = s / g.ny; }
                else if (ax == 1) { j = edge; i = s % g.nx; k = s / g.nx; }
                else { k = edge; i = s % g.nx; j = s / g.nx; }
                const size_t cell = ((size_t)k * g.ny + j) * g.nx + i;
#pragma unroll
                for (int q = 0; q < DIMS; ++q) {
                    fg_real* pb = bvel + ((size_t)b * DIMS + q) * slab_n + s;
                    const fg_real vb = *pb;
                    *pb = vb - tcoef * (vb - fold.vel[((size_t)b * DIMS + q) * N + cell]);
                }
                for (int ch = 0; ch < fold.nsc[face]; ++ch) {
                    fg_real* pb = fold.bscal[face] + ((size_t)b * fold.nsc[face] + ch) * slab_n + s;
                    const fg_real sb = *pb;
                    *pb = sb - tcoef * (sb - fold.scal[((size_t)b * fold.nsc[face] + ch) * N + cell]);
                }
            }
        }
        __syncthreads();     // the balance below reads what this workgroup just wrote (one CU, one L1)
    }
    double fixed = 0.0, freef = 0.0;
    for (int f = 0; f < 2 * DIMS; ++f) {
        if (!g.fixed[f]) continue;
        const int ax = f >> 1;
        const int slab_n = fg_slab_size(g, ax);
        const double sgn = (f & 1) ? 1.0 : -1.0;
        double acc = 0.0;
        for (int s = threadIdx.x; s < slab_n; s += blockDim.x) {
            int i = 0, j = 0, k = 0;
            if (ax == 0) { j = s % g.ny; k = s / g.ny; }
            else if (ax == 1) { i = s % g.nx; k = s / g.nx; }
            else { i = s % g.nx; j = s / g.nx; }
            fg_real area;
            if (ax == 0) area = g.h[1][j] * (DIMS == 3 ? g.h[2][k] : 1.f);
            else if (ax == 1) area = g.h[0][i] * (DIMS == 3 ? g.h[2][k] : 1.f);
            else area = g.h[0][i] * g.h[1][j];
            acc += sgn * (double)(bnd.vel[f][((size_t)b * DIMS + ax) * slab_n + s] * area);
        }
        if ((free_mask >> f) & 1) freef += acc; else fixed += acc;
    }
    __shared__ double lds[8];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { fixed += __shfl_down(fixed, o, 64); freef += __shfl_down(freef, o, 64); }
    if ((threadIdx.x & 63) == 0) { lds[threadIdx.x >> 6] = fixed; lds[4 + (threadIdx.x >> 6)] = freef; }
    __syncthreads();
    fixed = lds[0] + lds[1] + lds[2] + lds[3];
    freef = lds[4] + lds[5] + lds[6] + lds[7];
    if (!(fabs(fixed + freef) > (double)atol)) return;
    const fg_real scale = (fg_real)(-fixed / freef);
    for (int f = 0; f < 2 * DIMS; ++f) {
        if (!((free_mask >> f) & 1) || !g.fixed[f]) continue;
        const int slab_n = fg_slab_size(g, f >> 1);
        fg_real* v = bvel_rw_.p[f] + (size_t)b * DIMS * slab_n;
        for (int s = threadIdx.x; s < DIMS * slab_n; s += blockDim.x) v[s] *= scale;
    }
}

// Workgroups per env in the reduction kernels (sum, max).  Each ends in same-address atomics, which serialise at ~0.1 us
// apiece, so fewer, fatter workgroups win as long as the chip still has a few hundred of them in total
// (measured: B = 64, 256 x 128: 32 -> 8 per env takes the max-velocity pass 18.5 -> 7.3 us; B = 8, 128 x 64 x 64:
// 256 -> 32 per env 31 -> 21 us, 8 per env 66 us).
inline unsigned fg_reduce_wgs(const fg_state* s) {
    if (s->reduce_wgs > 0) return (unsigned)s->reduce_wgs;      // FG_REDUCE_WGS (tuning runs)
    const unsigned per_env = (512 + s->grid.B - 1) / s->grid.B;
    return per_env < 8 ? 8 : (per_env > 64 ? 64 : per_env);
}

inline dim3 stride_grid(const fg_state* s, long per_env_elems) {
    long blocks = (per_env_elems / 4 + FG_BLOCK - 1) / FG_BLOCK;
    const long cap = (2048 + s->grid.B - 1) / s->grid.B;  // ~8 workgroups per CU over the batch
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return dim3((unsigned)blocks, (unsigned)s->grid.B, 1);
}

}  // namespace


int fg_launch_adv_build(const fg_state* s, const FgBounds& bnd, const FgAdvArgs& a_in, hipStream_t st) {
    // the state of the BiCGStab solve that follows is prepared by this launch (FgBicgBegin, fg_internal.h; fg_bicgstab_solve skips its
    // k_bicg_begin when the record matches its own arguments)
    FgAdvArgs a = a_in;
    a.begin.acc = s->acc; a.begin.sc = s->scratch_B + 4 * s->grid.B; a.begin.flags = s->flags; a.begin.info = s->info_dev;
    a.begin.nc = a.for_scalar ? 1 : s->grid.dims;
    s->bicg_ready_nc = a.begin.nc; s->bicg_ready_dt = a.dt; s->cg_ready_ns = 0;
    // live profile: u (d) + the matrix (1 + 2 d) + the right-hand sides (d) written, u and the source read: 44 B per cell in 2-D, 56 in 3-D
    const double nd = (double)s->grid.n;
    const int slot = fg_prof_slot(s, FG_PK_ADV_BUILD, nullptr, s->grid.B, (s->grid.dims == 3 ? 56.0 : 44.0) * nd, 30.0 * nd, st);
    FG_DISPATCH(s, {
        const FgLaunch L = fg_launch_geometry<DIMS, VEC>(s->grid);
        if (a.for_scalar)
            FG_LAUNCH_P(s, slot, (k_adv_build<DIMS, VEC, true>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, bnd, a, L.tiles_x,
                        L.tiles_y, L.tiles);
        else if (a.visc)
            FG_LAUNCH_P(s, slot, (k_adv_build<DIMS, VEC, false, true>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, bnd, a, L.tiles_x,
                        L.tiles_y, L.tiles);
        else
            FG_LAUNCH_P(s, slot, (k_adv_build<DIMS, VEC, false>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, bnd, a, L.tiles_x,
                        L.tiles_y, L.tiles);
    });
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

// SGSviscosityIncompressibleSmagorinsky (K.cu:6913-6966) on the rectilinear block: nu_t = C Delta^2 |S|,  |S| = sqrt(2 S:S),
// S = sym(grad u) from getBlockDataGradient (K.cu:2997-3040): (value above - value below) / distance along every axis, where a
// neighbour across a periodic or interior face is the cell (distance contribution 1) and a FIXED face contributes its (Dirichlet)
// boundary value at half a cell (0.5); then times Minv = 1 / h.  Delta^2 = max_a h_a^2 (squared column lengths of M; the kernel
// never takes the root, :6959).  One cell per thread: a PRE-hook kernel, once per step.
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_sgs_smagorinsky(FgGrid g, FgBounds bnd, const fg_real* __restrict__ vel, fg_real coefficient,
                                                               fg_real* __restrict__ out) {
    const int b = blockIdx.y, c = blockIdx.x * FG_BLOCK + threadIdx.x;
    if (c >= g.n) return;
    const int ext[3] = {g.nx, g.ny, DIMS == 3 ? g.nz : 1};
    const int pos[3] = {c % g.nx, (c / g.nx) % g.ny, c / (g.nx * g.ny)};
    const int stride[3] = {1, g.nx, g.nx * g.ny};
    const fg_real* u = vel + (size_t)b * DIMS * g.n;
    fg_real grad[DIMS][DIMS];   // [component][axis]
    fg_real delta = 0.f;
#pragma unroll
    for (int a = 0; a < DIMS; ++a) {
        const fg_real h = g.h[a][pos[a]];
        delta = FG_FMAX(delta, h * h);
        fg_real dist = 2.f;
        fg_real diff[DIMS];
#pragma unroll
        for (int q = 0; q < DIMS; ++q) diff[q] = 0.f;
#pragma unroll
        for (int up = 0; up < 2; ++up) {
            const int f = 2 * a + up;
            const bool at = up ? (pos[a] + 1 == ext[a]) : (pos[a] == 0);
            if (at && g.fixed[f]) {
                // slab index of this cell on face f: the remaining axes, lowest fastest
                int si = 0, mul = 1;
#pragma unroll
                for (int t = 0; t < DIMS; ++t) if (t != a) { si += pos[t] * mul; mul *= ext[t]; }
                const fg_real* bv = bnd.vel[f] + (size_t)b * DIMS * mul;
#pragma unroll
                for (int q = 0; q < DIMS; ++q) diff[q] += (up ? 1.f : -1.f) * bv[q * mul + si];
                dist -= 0.5f;
            } else {
                const int n = at ? (up ? c - (ext[a] - 1) * stride[a] : c + (ext[a] - 1) * stride[a]) : (up ? c + stride[a] : c - stride[a]);
#pragma unroll
                for (int q = 0; q < DIMS; ++q) diff[q] += (up ? 1.f : -1.f) * u[(size_t)q * g.n + n];
            }
        }
#pragma unroll
        for (int q = 0; q < DIMS; ++q) grad[q][a] = diff[q] / dist * g.rh[a][pos[a]];
    }
    fg_real d = 0.f;
#pragma unroll
    for (int i = 0; i < DIMS; ++i)
#pragma unroll
        for (int j = i; j < DIMS; ++j) {
            fg_real sij = 0.5f * (grad[i][j] + grad[j][i]);
            sij *= sij;
            d += (i != j) ? 2.f * sij : sij;
        }
    out[(size_t)b * g.n + c] = coefficient * delta * FG_SQRT(2.f * d);
}

// Wall-stress forcing of the turbulent-channel env (tcf_env.py PRE hook, grid.py:147-176): per env the mean of u[axis] over the cell
// layer next to the -y wall and over the one next to the +y wall, G = 1/2 (coef_lo mean_lo + coef_hi mean_hi) -> force[b][axis],
// the other components 0.  One workgroup of 1024 threads per env; rows in a fixed order, sums in fp64.
__global__ __launch_bounds__(1024) void k_wall_forcing(FgGrid g, const fg_real* __restrict__ vel, int axis, fg_real coef_lo, fg_real coef_hi,
                                                        fg_real* __restrict__ force, int dims) {
    const int b = blockIdx.x;
    const size_t N = g.n;
    const fg_real* __restrict__ u = vel + ((size_t)b * dims + axis) * N;
    double lo = 0.0, hi = 0.0;
    const int rows = g.nz;       // one (z) row of nx cells per layer and z
    for (int idx = threadIdx.x; idx < rows * g.nx; idx += blockDim.x) {
        const int k = idx / g.nx, i = idx - k * g.nx;
        lo += (double)u[((size_t)k * g.ny + 0) * g.nx + i];
        hi += (double)u[((size_t)k * g.ny + (g.ny - 1)) * g.nx + i];
    }
    __shared__ double lds[2][16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo += __shfl_down(lo, o, 64); hi += __shfl_down(hi, o, 64); }
    if ((threadIdx.x & 63) == 0) { lds[0][threadIdx.x >> 6] = lo; lds[1][threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sl = 0.0, sh = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { sl += lds[0][w]; sh += lds[1][w]; }
        const double n = (double)rows * g.nx;
        const fg_real tau_lo = coef_lo * (fg_real)(sl / n), tau_hi = coef_hi * (fg_real)(sh / n);
        for (int q = 0; q < dims; ++q) force[b * dims + q] = (q == axis) ? (fg_real)0.5 * (tau_lo + tau_hi) : (fg_real)0;
    }
}

int fg_launch_wall_forcing(const fg_state* s, hipStream_t st) {
    hipLaunchKernelGGL(k_wall_forcing, dim3(s->grid.B), dim3(1024), 0, st, s->grid, (const fg_real*)s->velocity, s->wall_forcing_axis,
                       s->wall_forcing_coef[0], s->wall_forcing_coef[1], s->force_uniform, s->grid.dims);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_launch_sgs(const fg_state* s, const FgBounds& bnd, fg_real coefficient, fg_real* out, hipStream_t st) {
    const dim3 grid((s->grid.n + FG_BLOCK - 1) / FG_BLOCK, s->grid.B);
    if (s->grid.dims == 2) hipLaunchKernelGGL(k_sgs_smagorinsky<2>, grid, dim3(FG_BLOCK), 0, st, s->grid, bnd, (const fg_real*)s->velocity, coefficient, out);
    else hipLaunchKernelGGL(k_sgs_smagorinsky<3>, grid, dim3(FG_BLOCK), 0, st, s->grid, bnd, (const fg_real*)s->velocity, coefficient, out);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_launch_pressure_setup(const fg_state* s, const fg_real* dt, hipStream_t st) {
    const int n = s->grid.n;
    hipLaunchKernelGGL(k_reciprocal, stride_grid(s, n), dim3(FG_BLOCK), 0, st, s->A, s->rA, dt, n, (n % 4 == 0) ? n / 4 : 0);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_launch_h(const fg_state* s, const fg_real* dt, const fg_real* vel_result, hipStream_t st) {
    // the state of the pressure CG that follows is prepared by this launch (FgCgBegin, fg_cg.h; fg_cg_solve skips its k_cg_begin when
    // the record matches its own arguments)
    fg_htrace("h_launch_in");
    FgCgBegin begin;
    begin.acc = s->cg_acc; begin.flags = s->flags; begin.info = s->info_dev; begin.mean_sums = s->acc; begin.best = s->cg_best;
    begin.track_best = s->cg_return_best; begin.ns = fg_cg_slots(s); begin.xsum = s->fcg_xsum;
    s->cg_ready_ns = begin.ns; s->cg_ready_best = begin.track_best; s->cg_ready_dt = dt; s->bicg_ready_nc = 0; s->cg_start_ready = 0;
    const int slot_h = fg_prof_slot(s, FG_PK_H, nullptr, s->grid.B, (s->grid.dims == 3 ? 68.0 : 52.0) * (double)s->grid.n, 6.0 * s->grid.dims * s->grid.dims * (double)s->grid.n, st);
    FG_DISPATCH(s, {
        const FgLaunch L = fg_launch_geometry<DIMS, VEC>(s->grid);
        FG_LAUNCH_P(s, slot_h, (k_h<DIMS, VEC>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, dt, s->rA, s->Coff, s->adv_rhs,
                    vel_result, s->hvec, begin, L.tiles_x, L.tiles_y, L.tiles);
    });
    FG_HIP_CHECK(hipGetLastError());
    fg_htrace("h_launch_out");
    return FG_OK;
}

int fg_launch_div(const fg_state* s, const FgBounds& bnd, const fg_real* dt, const fg_real* hvec, fg_real* div,
                  hipStream_t st, bool cg_from_zero, bool fused_fwd) {
    // cg_from_zero: the pressure CG on this right-hand side starts from the zero vector and its state was prepared by the k_h in
    // front of this launch -- then this kernel also starts it (FgCgStart: r = b into w[0], x = 0 into p_result, r.r) and fg_cg_solve
    // finds the record and skips k_cg_residual
    FgCgStart start = {nullptr, nullptr, nullptr, 1};
    s->cg_start_ready = 0;
    if (cg_from_zero && dt != nullptr && s->cg_ready_ns > 0 && s->cg_ready_dt == dt && div == s->div) {
#if !FG_F64
        if (fused_fwd && fg_fcg_ok(s)) {
            // fused CG on a fast-transform grid: one row kernel writes the right-hand side, starts the solve AND transforms r_0
            // (k_fcg_div_fwd, fg_fftcg.hip); fg_cg_solve finds the record (2) and goes straight to the tridiagonal solve
            s->cg_start_ready = 2;
            return fg_fcg_div_fwd(const_cast<fg_state*>(s), bnd, dt, hvec, div, s->cg_ready_ns, st);
        }
#endif
        start.acc = s->cg_acc; start.r = s->w[0]; start.x = s->p_result; start.ns = s->cg_ready_ns;
        s->cg_start_ready = 1;
    }
    const int slot_d = fg_prof_slot(s, FG_PK_DIV, nullptr, s->grid.B, 28.0 * (double)s->grid.n, 4.0 * s->grid.dims * (double)s->grid.n, st);
    FG_DISPATCH(s, {
        const FgLaunch L = fg_launch_geometry<DIMS, VEC>(s->grid);
        FG_LAUNCH_P(s, slot_d, (k_div<DIMS, VEC>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, bnd, dt, hvec, div, start, L.tiles_x,
                    L.tiles_y, L.tiles);
    });
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_launch_correct(const fg_state* s, const fg_real* dt, const fg_real* rA, const fg_real* hvec, const fg_real* p,
                      fg_real* vel_out, hipStream_t st, fg_real* vel_copy, const FgMeanRef* mean, const FgLazyRef* lazy) {
    fg_htrace("correct_in");
    const FgMeanRef mr = mean ? *mean : FgMeanRef{nullptr, nullptr, nullptr, nullptr, nullptr};
    const FgLazyRef lz = lazy ? *lazy : FgLazyRef{nullptr, nullptr, nullptr};
    // live profile: rA, h (d), p read, u (d) written -- plus the block copy / pressure the last corrector also writes
    const double cb = (s->grid.dims == 3 ? 36.0 : 28.0) + (vel_copy ? 4.0 * s->grid.dims : 0.0) + ((mean && mean->sums) || (lazy && lazy->p_res) ? 4.0 : 0.0);
    const int slot_c = fg_prof_slot(s, FG_PK_CORRECT, nullptr, s->grid.B, cb * (double)s->grid.n, 5.0 * s->grid.dims * (double)s->grid.n, st);
    FG_DISPATCH(s, {
        const FgLaunch L = fg_launch_geometry<DIMS, VEC>(s->grid);
        FG_LAUNCH_P(s, slot_c, (k_correct<DIMS, VEC>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, dt, rA, hvec, p, vel_out, vel_copy, mr, lz,
                    L.tiles_x, L.tiles_y, L.tiles);
    });
    FG_HIP_CHECK(hipGetLastError());
    fg_htrace("correct_out");
    return FG_OK;
}

int fg_launch_max_velocity(const fg_state* s, const FgBounds& bnd, fg_real* out_B, hipStream_t st, fg_real* mirror_B, FgPollOut poll,
                           fg_real* flux_B, fg_real* flux_mirror, FgDtRule rule) {
    FG_REQUIRE(!rule.t_rem || (poll.gran && mirror_B && !FG_F64), FG_ERR_INVALID_ARG, "the device-side sub-step rule needs the result-word form");
    FG_REQUIRE(!flux_B || (mirror_B && flux_mirror), FG_ERR_INVALID_ARG, "the flux balance rides in the mirrored form only");
    // out_B [B] and the arrival counters right behind it (scratch_B rows 1 and 2) are zeroed together
    int32_t* done_B = reinterpret_cast<int32_t*>(out_B + s->grid.B);
    FG_REQUIRE(!mirror_B || out_B == s->scratch_B + s->grid.B, FG_ERR_INVALID_ARG, "mirror needs the scratch row as out_B");
    // (the mirrored form cleans up after itself: the memset -- maxima, per-env counters and the counter of finished envs behind
    // them -- is only needed after something else used the rows)
    if (!mirror_B || !s->maxvel_clean)
        FG_HIP_CHECK(hipMemsetAsync(out_B, 0, mirror_B ? sizeof(fg_real) * s->grid.B + sizeof(int32_t) * (s->grid.B + 1) : sizeof(fg_real) * s->grid.B, st));
    s->maxvel_clean = mirror_B ? 1 : 0;
    if ((s->grid.nx & 3) == 0) {
        const int rows = s->grid.ny * s->grid.nz;
        int rpb = 4;  // rows per workgroup: one per wave, more when that still leaves >= 8 workgroups per CU ...
        while ((long)((rows + 2 * rpb - 1) / (2 * rpb)) * s->grid.B >= 2048) rpb *= 2;
        while ((rows + rpb - 1) / rpb > (int)fg_reduce_wgs(s)) rpb *= 2;  // ... and few per env: they meet in same-address atomics
        dim3 grid((rows + rpb - 1) / rpb, s->grid.B);
        const int slot_m = fg_prof_slot(s, FG_PK_MAXVEL, nullptr, s->grid.B, 4.0 * s->grid.dims * (double)s->grid.n, 2.0 * s->grid.dims * (double)s->grid.n, st);
        if (s->grid.dims == 2)
            FG_LAUNCH_P(s, slot_m, k_max_velocity_rows<2>, grid, dim3(FG_BLOCK), 0, st, s->grid, bnd, s->velocity, out_B, done_B,
                        mirror_B, rpb, poll, flux_B, flux_mirror, rule);
        else
            FG_LAUNCH_P(s, slot_m, k_max_velocity_rows<3>, grid, dim3(FG_BLOCK), 0, st, s->grid, bnd, s->velocity, out_B, done_B,
                        mirror_B, rpb, poll, flux_B, flux_mirror, rule);
        FG_HIP_CHECK(hipGetLastError());
        return FG_OK;
    }
    dim3 grid = stride_grid(s, (long)s->grid.n * 4);
    if (s->grid.dims == 2)
        hipLaunchKernelGGL(k_max_velocity<2>, grid, dim3(FG_BLOCK), 0, st, s->grid, bnd, s->velocity, out_B, done_B, mirror_B, poll, flux_B, flux_mirror, rule);
    else
        hipLaunchKernelGGL(k_max_velocity<3>, grid, dim3(FG_BLOCK), 0, st, s->grid, bnd, s->velocity, out_B, done_B, mirror_B, poll, flux_B, flux_mirror, rule);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_launch_flux_balance(const fg_state* s, const FgBounds& bnd, fg_real* out_B, hipStream_t st, FgPollOut poll) {
    if (s->grid.dims == 2)
        hipLaunchKernelGGL(k_flux_balance<2>, dim3(s->grid.B), dim3(FG_BLOCK), 0, st, s->grid, bnd, out_B, poll);
    else
        hipLaunchKernelGGL(k_flux_balance<3>, dim3(s->grid.B), dim3(FG_BLOCK), 0, st, s->grid, bnd, out_B, poll);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_launch_copy_active(const fg_state* s, const fg_real* dt, const fg_real* src, fg_real* dst, int comps,
                          hipStream_t st) {
    const long per_env = (long)comps * s->grid.n;
    hipLaunchKernelGGL(k_copy_active, stride_grid(s, per_env), dim3(FG_BLOCK), 0, st, dt, src, dst, per_env);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_launch_buoyancy(const fg_state* s, const fg_real* dt, const fg_real* T, long t_env_stride, fg_real* source, int axis,
                       fg_real factor, hipStream_t st) {
    hipLaunchKernelGGL(k_buoyancy, stride_grid(s, (long)s->grid.n * 4), dim3(FG_BLOCK), 0, st, dt, T, t_env_stride,
                       source, s->grid.dims, s->grid.n, axis, factor);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_launch_mean_sub(const fg_state* s, const fg_real* dt, fg_real* p, fg_real* p_copy, hipStream_t st) {
    FgDacc* sums = s->acc;  // first B accumulators of the pool: free between solves, zeroed by k_cg_begin
    dim3 grid = stride_grid(s, (long)s->grid.n * 4);
    // every workgroup ends in atomics on its env's accumulator and same-address atomics serialise (~0.1 us each):
    // few workgroups per env for the reduction pass (fg_reduce_wgs)
    dim3 rgrid(grid.x > fg_reduce_wgs(s) ? fg_reduce_wgs(s) : grid.x, grid.y);
    hipLaunchKernelGGL(k_sum_env, rgrid, dim3(FG_BLOCK), 0, st, dt, p, sums, s->grid.n);
    hipLaunchKernelGGL(k_sub_mean, grid, dim3(FG_BLOCK), 0, st, dt, p, p_copy, sums, s->grid.n);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_launch_outflow(const fg_state* s, int face, fg_real velm_axis, const fg_real* dt, hipStream_t st) {
    const int slab_n = (face >> 1) == 0 ? s->grid.ny * s->grid.nz : (face >> 1) == 1 ? s->grid.nx * s->grid.nz : s->grid.nx * s->grid.ny;
    dim3 grid((slab_n + FG_BLOCK - 1) / FG_BLOCK, s->grid.B);
    const int nsc = (s->scalar && s->bscal[face]) ? s->cfg.n_scalars : 0;
    if (s->grid.dims == 2)
        hipLaunchKernelGGL(k_outflow<2>, grid, dim3(FG_BLOCK), 0, st, s->grid, face, velm_axis, dt, s->velocity, s->bvel[face],
                           s->scalar, s->bscal[face], nsc);
    else
        hipLaunchKernelGGL(k_outflow<3>, grid, dim3(FG_BLOCK), 0, st, s->grid, face, velm_axis, dt, s->velocity, s->bvel[face],
                           s->scalar, s->bscal[face], nsc);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_launch_balance(const fg_state* s, const FgBounds& bnd, int free_mask, fg_real atol, const fg_real* dt, hipStream_t st,
                      int outflow_mask, const fg_real* outflow_velm, const fg_real* dt_host_values) {
    FgOutflowFold fold;
    fold = FgOutflowFold{};
    if (dt_host_values) {       // (the caller checked B <= 64)
        fold.dt_n = s->grid.B; fold.dt_out = s->dt_dev;
        for (int b = 0; b < s->grid.B; ++b) fold.dt_v[b] = dt_host_values[b];
    }
    if (outflow_mask) {     // (the caller checked the slab sizes: fg_outflow_folds)
        fold.mask = outflow_mask; fold.vel = s->velocity; fold.scal = s->scalar;
        for (int a = 0; a < 3; ++a) fold.velm[a] = outflow_velm[a];
        for (int f = 0; f < 6; ++f) { fold.bscal[f] = s->bscal[f]; fold.nsc[f] = (s->scalar && s->bscal[f]) ? s->cfg.n_scalars : 0; }
    }
    FgBvelRw rw;
    for (int f = 0; f < 6; ++f) rw.p[f] = s->bvel[f];
    if (s->grid.dims == 2)
        hipLaunchKernelGGL(k_balance_fluxes<2>, dim3(s->grid.B), dim3(FG_BLOCK), 0, st, s->grid, bnd, rw, free_mask, atol, dt, fold);
    else
        hipLaunchKernelGGL(k_balance_fluxes<3>, dim3(s->grid.B), dim3(FG_BLOCK), 0, st, s->grid, bnd, rw, free_mask, atol, dt, fold);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

// the convective update of these faces can ride in the balance launch: slabs of at most four rounds of the env's one workgroup
bool fg_outflow_folds(const fg_state* s, int outflow_mask) {
    for (int f = 0; f < 2 * s->grid.dims; ++f) {
        if (!((outflow_mask >> f) & 1)) continue;
        const int ax = f >> 1;
        const long slab_n = ax == 0 ? (long)s->grid.ny * s->grid.nz : ax == 1 ? (long)s->grid.nx * s->grid.nz : (long)s->grid.nx * s->grid.ny;
        if (slab_n > 4 * FG_BLOCK) return false;
    }
    return outflow_mask != 0;
}
