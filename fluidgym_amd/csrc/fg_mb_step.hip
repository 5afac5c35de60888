// Multi-block, non-orthogonal PISO step on gfx950: assembly kernels, ELL Krylov solvers and the C ABI (fg_mb_*).
//
// One thread per (env, cell); the env batch is the slow grid axis, so consecutive lanes read consecutive cells of one
// env's fields (coalesced) while the mesh tables (fg_mb.h) are shared by all envs and stay cached.  Replaces, for
// meshes with connected blocks and cross metrics, the same reference calls as the single-block path:
//   PISO_build_matrix (K.cu:3616-3880), kPISO_build_advection_RHS (:4296-4400), PISO_build_pressure_matrix (:4812-4978),
//   PISO_build_pressure_rhs (:5136-5255), k_computePressureRHSdivergenceFromFlux (:5389-5434),
//   k_pressureRHSaddNonOrthoComponents (:5471-5493), PISO_update_velocity (:5962-5995),
//   bicgstabSolveGPU / cgSolveGPU (bicgstab_solver_kernel.cu:63-411, cg_solver_kernel.cu:129-471),
// driven in the order of _PISO_split_step's non-orthogonal branch (PISOtorch_simulation.py:1707-1972).
// "K.cu" = extensions/PISO_multiblock_cuda_kernel.cu.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "fg_mb.h"
#include "fg_mb_solve.h"

namespace {




template <int DIMS>
__device__ __forceinline__ void mb_load_T(const mb_real* __restrict__ T, int i, mb_real (&mi)[DIMS * DIMS], mb_real& det) {
    const mb_real* t = T + (size_t)i * (DIMS * DIMS + 1);
#pragma unroll
    for (int q = 0; q < DIMS * DIMS; ++q) mi[q] = t[q];
    det = t[DIMS * DIMS];
}

// contravariant components det * Minv u of the cells, and of the boundary faces along their own axis
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mb_contra(MbDev D, const mb_real* __restrict__ dt, const mb_real* __restrict__ u,
                                                         const mb_real* __restrict__ ub, mb_real* __restrict__ cc,
                                                         mb_real* __restrict__ fb) {
    MB_CELL
    if (!mb_active(dt, b)) return;
    if (valid) {
        mb_real mi[DIMS * DIMS], det;
        mb_load_T<DIMS>(D.T, i, mi, det);
        mb_real v[DIMS];
#pragma unroll
        for (int c = 0; c < DIMS; ++c) v[c] = u[((size_t)b * DIMS + c) * N + i];
#pragma unroll
        for (int a = 0; a < DIMS; ++a) {
            mb_real s = 0.f;
#pragma unroll
            for (int c = 0; c < DIMS; ++c) s += mi[a * DIMS + c] * v[c];
            cc[((size_t)b * DIMS + a) * N + i] = det * s;
        }
    }
    if (fb != nullptr && i < D.NB) {
        mb_real mi[DIMS * DIMS], det;
        mb_load_T<DIMS>(D.Tb, i, mi, det);
        const int axis = D.bface[i] >> 1;
        mb_real s = 0.f;
#pragma unroll
        for (int c = 0; c < DIMS; ++c) s += mi[axis * DIMS + c] * ub[((size_t)b * DIMS + c) * D.NB + i];
        fb[(size_t)b * D.NB + i] = det * s;
    }
}

// face flux, not multiplied by the face sign (computeFluxesNDLoop, K.cu:1568-1645)
template <int DIMS>
__device__ __forceinline__ mb_real mb_flux(const MbDev& D, const mb_real* __restrict__ cc_b, const mb_real* __restrict__ fb_b,
                                         int f, int i) {
    const int n = D.nbr[(size_t)f * D.N + i];
    if (n < 0) return fb_b[-1 - n];
    const int code = D.fcode[(size_t)f * D.N + i];
    const mb_real vn = cc_b[(size_t)(code & 3) * D.N + n];
    return 0.5f * (((code & 4) ? -vn : vn) + cc_b[(size_t)(f >> 1) * D.N + i]);
}

template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mb_matrix(MbDev D, const mb_real* __restrict__ dt, mb_real nu,
                                                         const mb_real* __restrict__ cc, const mb_real* __restrict__ fb,
                                                         mb_real* __restrict__ Cdiag, mb_real* __restrict__ Coff,
                                                         mb_real* __restrict__ rA) {
    MB_CELL
    if (!valid || !mb_active(dt, b)) return;
    constexpr int F = 2 * DIMS;
    const mb_real det = D.T[(size_t)i * (DIMS * DIMS + 1) + DIMS * DIMS];
    const mb_real* cc_b = cc + (size_t)b * DIMS * N;
    const mb_real* fb_b = fb + (size_t)b * D.NB;
    mb_real diag = det / dt[b] + nu * D.Vdiag[i];
    const mb_real rdet = 1.f / det;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        mb_real o = 0.f;
        if (D.nbr[(size_t)f * N + i] >= 0) {
            const mb_real ff = ((f & 1) ? 0.5f : -0.5f) * mb_flux<DIMS>(D, cc_b, fb_b, f, i);
            diag += ff;
            o = (ff + nu * D.Voff[(size_t)f * N + i]) * rdet;
        }
        Coff[((size_t)b * F + f) * N + i] = o;
    }
    diag *= rdet;
    Cdiag[(size_t)b * N + i] = diag;
    rA[(size_t)b * N + i] = 1.f / diag;
}

// boundary sources of one cell and component: -u_b flux_b n + 2 nu alpha_b u_b over its prescribed faces
template <int DIMS>
__device__ __forceinline__ mb_real mb_boundary_source(const MbDev& D, const mb_real* __restrict__ ub_c,
                                                    const mb_real* __restrict__ fb_b, mb_real nu, int i) {
    mb_real s = 0.f;
#pragma unroll
    for (int f = 0; f < 2 * DIMS; ++f) {
        const int n = D.nbr[(size_t)f * D.N + i];
        if (n >= 0) continue;
        const int k = -1 - n;
        const mb_real* t = D.Tb + (size_t)k * (DIMS * DIMS + 1);
        mb_real a = 0.f;
#pragma unroll
        for (int q = 0; q < DIMS; ++q) a += t[(f >> 1) * DIMS + q] * t[(f >> 1) * DIMS + q];
        a *= t[DIMS * DIMS];
        const mb_real vel = ub_c[k];
        s += vel * (2.f * nu * a - ((f & 1) ? fb_b[k] : -fb_b[k]));
    }
    return s;
}

// velocity right-hand side (kPISO_build_advection_RHS): grid.z = component
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mb_vrhs(MbDev D, const mb_real* __restrict__ dt, mb_real nu,
                                                       const mb_real* __restrict__ u_old, const mb_real* __restrict__ u_res,
                                                       const mb_real* __restrict__ ub, const mb_real* __restrict__ fb,
                                                       const mb_real* __restrict__ src, mb_real* __restrict__ rhs) {
    MB_CELL
    if (!valid || !mb_active(dt, b)) return;
    const int c = blockIdx.z;
    const size_t vb = ((size_t)b * DIMS + c) * N;
    const mb_real* ub_c = ub + ((size_t)b * DIMS + c) * D.NB;
    const mb_real det = D.T[(size_t)i * (DIMS * DIMS + 1) + DIMS * DIMS];
    mb_real r = det * u_old[vb + i] / dt[b];
    r += mb_boundary_source<DIMS>(D, ub_c, fb + (size_t)b * D.NB, nu, i);
    mb_real S = 0.f;
    for (int k = 0; k < D.KC; ++k) S += D.SVc_w[(size_t)k * N + i] * u_res[vb + D.SVc_idx[(size_t)k * N + i]];
    for (int k = 0; k < D.KB; ++k) S += D.SVb_w[(size_t)k * N + i] * ub_c[D.SVb_idx[(size_t)k * N + i]];
    r -= nu * S;
    r /= det;
    if (src) r += src[vb + i];
    rhs[vb + i] = r;
}

// pressure matrix from the coefficient pairs (PISO_build_pressure_matrix)
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mb_pmatrix(MbDev D, const mb_real* __restrict__ dt, const mb_real* __restrict__ rA,
                                                          mb_real* __restrict__ Pdiag, mb_real* __restrict__ Poff,
                                                          mb_real* __restrict__ Poff4, const uint16_t* __restrict__ cell_slot = nullptr,
                                                          mb_real* __restrict__ Poff4s = nullptr, mb_real* __restrict__ Pdiag_s = nullptr,
                                                          int slot_stride = fg_mb_state::OC_SLOTS) {
    MB_CELL
    if (!valid || !mb_active(dt, b)) return;
    constexpr int F = 2 * DIMS;
    const mb_real* ra = rA + (size_t)b * N;
    mb_real o4[4] = {0.f, 0.f, 0.f, 0.f}, dv = 0.f;
    const mb_real rp = ra[i];
    mb_real rn[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int n = D.nbr[(size_t)f * N + i];
        rn[f] = n >= 0 ? ra[n] : 0.f;
    }
#pragma unroll
    for (int g = 0; g <= F; ++g) {
        mb_real v = 0.f;
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const size_t q = ((size_t)g * F + f) * N + i;
            v += D.KPp[q] * rp + D.KPn[q] * rn[f];
        }
        if (g == 0) { Pdiag[(size_t)b * N + i] = v; dv = v; }
        else {
            const mb_real o = (D.nbr[(size_t)(g - 1) * N + i] >= 0) ? v : 0.f;
            Poff[((size_t)b * F + (g - 1)) * N + i] = o;
            if (DIMS == 2) o4[(g - 1) & 3] = o;
        }
    }
    // the same off-diagonals once more, interleaved per cell: the on-chip CG fetches a cell's four with one 16-byte load
    if (DIMS == 2 && Poff4) *reinterpret_cast<float4*>(Poff4 + ((size_t)b * N + i) * 4) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    // ... and in the slot order of the aggregate-owned on-chip CG (fg_mb.h: OC_SLOTS) or of the cluster CG (fg_mb_cluster.hip),
    // diagonal included
    if (DIMS == 2 && cell_slot) {
        const size_t sl = (size_t)b * slot_stride + cell_slot[i];
        *reinterpret_cast<float4*>(Poff4s + sl * 4) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        Pdiag_s[sl] = dv;
    }
}

// which slot-ordered copy of the pressure matrix k_mb_pmatrix writes beside the cell-ordered one: the cluster CG's (when its
// tables are installed and the mesh is one it takes), else the aggregate-owned on-chip CG's, else none
struct MbSlots { const uint16_t* cell_slot; mb_real* off4; mb_real* diag; int stride; };
static bool mb_cluster_wanted(const fg_mb_state* s) { return s->cl_on && (s->cl_mode == 2 || s->N > 8 * 1024); }
static MbSlots mb_slots(const fg_mb_state* s) {
    if (mb_cluster_wanted(s)) return MbSlots{s->cl_cell_slot, s->cl_off4, s->cl_diag, FG_CL_G * s->cl_S};
    if (s->oc_agg) return MbSlots{s->oc_cell_slot, s->Poff4s, s->Pdiag_s, fg_mb_state::OC_SLOTS};
    return MbSlots{nullptr, nullptr, nullptr, 0};
}
static void mb_slots_written(fg_mb_state* s) {
    if (mb_cluster_wanted(s)) s->cl_matrix_stale = false;
    else s->oc_matrix_stale = false;
}
#define SLOTS mb_slots(s)

// h = (u_old/dt - H u* + S) / A   (PISO_build_pressure_rhs): grid.z = component
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mb_h(MbDev D, const mb_real* __restrict__ dt, mb_real nu,
                                                    const mb_real* __restrict__ rA, const mb_real* __restrict__ Coff,
                                                    const mb_real* __restrict__ u_old, const mb_real* __restrict__ u_star,
                                                    const mb_real* __restrict__ ub, const mb_real* __restrict__ fb,
                                                    const mb_real* __restrict__ src, mb_real* __restrict__ h) {
    MB_CELL
    if (!valid || !mb_active(dt, b)) return;
    constexpr int F = 2 * DIMS;
    const int c = blockIdx.z;
    const size_t vb = ((size_t)b * DIMS + c) * N;
    const mb_real det = D.T[(size_t)i * (DIMS * DIMS + 1) + DIMS * DIMS];
    mb_real H = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int n = D.nbr[(size_t)f * N + i];
        if (n >= 0) H += Coff[((size_t)b * F + f) * N + i] * u_star[vb + n];
    }
    mb_real S = mb_boundary_source<DIMS>(D, ub + ((size_t)b * DIMS + c) * D.NB, fb + (size_t)b * D.NB, nu, i) / det;
    if (src) S += src[vb + i];
    h[vb + i] = rA[(size_t)b * N + i] * (u_old[vb + i] / dt[b] - H + S);
}

// div of the flux of h, plus the lagged corner terms of the pressure (add_corner != 0)
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mb_div(MbDev D, const mb_real* __restrict__ dt, const mb_real* __restrict__ cc,
                                                      const mb_real* __restrict__ fb, const mb_real* __restrict__ rA,
                                                      const mb_real* __restrict__ p, int add_corner,
                                                      mb_real* __restrict__ div) {
    MB_CELL
    if (!valid || !mb_active(dt, b)) return;
    const mb_real* cc_b = cc + (size_t)b * DIMS * N;
    const mb_real* fb_b = fb + (size_t)b * D.NB;
    mb_real s = 0.f;
#pragma unroll
    for (int a = 0; a < DIMS; ++a) s += mb_flux<DIMS>(D, cc_b, fb_b, 2 * a + 1, i) - mb_flux<DIMS>(D, cc_b, fb_b, 2 * a, i);
    if (add_corner) {
        const mb_real* ra = rA + (size_t)b * N;
        const mb_real* pb = p + (size_t)b * N;
        const mb_real rp = ra[i];
        for (int k = 0; k < D.KPN; ++k) {
            const size_t q = (size_t)k * N + i;
            const mb_real wp = D.SP_wp[q], wn = D.SP_wn[q];
            if (wp == 0.f && wn == 0.f) continue;
            const int n = D.nbr[(size_t)D.SP_face[q] * N + i];
            s += (wp * rp + wn * (n >= 0 ? ra[n] : 0.f)) * pb[D.SP_idx[q]];
        }
    }
    div[(size_t)b * N + i] = s;
}

// u = h - (1/A) Minv^T grad_xi p   (PISO_update_velocity + getPressureGradient, K.cu:816-849)
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mb_correct(MbDev D, const mb_real* __restrict__ dt, const mb_real* __restrict__ rA,
                                                          const mb_real* __restrict__ h, const mb_real* __restrict__ p,
                                                          mb_real* __restrict__ u, mb_real* __restrict__ u_copy = nullptr) {
    MB_CELL
    if (!valid || !mb_active(dt, b)) return;
    const mb_real* pb = p + (size_t)b * N;
    mb_real mi[DIMS * DIMS], det;
    mb_load_T<DIMS>(D.T, i, mi, det);
    mb_real g[DIMS];
#pragma unroll
    for (int a = 0; a < DIMS; ++a) {
        const int nl = D.nbr[(size_t)(2 * a) * N + i], nh = D.nbr[(size_t)(2 * a + 1) * N + i];
        const mb_real fac = (nl < 0 || nh < 0) ? 1.f : 0.5f;
        g[a] = (pb[nh < 0 ? i : nh] - pb[nl < 0 ? i : nl]) * fac;
    }
    const mb_real ra = rA[(size_t)b * N + i];
#pragma unroll
    for (int c = 0; c < DIMS; ++c) {
        mb_real gp = 0.f;
#pragma unroll
        for (int a = 0; a < DIMS; ++a) gp += g[a] * mi[a * DIMS + c];
        const size_t q = ((size_t)b * DIMS + c) * N + i;
        const mb_real uc = h[q] - ra * gp;
        u[q] = uc;
        if (u_copy) u_copy[q] = uc;
    }
}


// max |Minv u| over cells and boundary faces (Block::getMaxVelocity, domain_structs.cpp:1580-1611)
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mb_maxvel(MbDev D, const mb_real* __restrict__ u, const mb_real* __restrict__ ub,
                                                         mb_real* __restrict__ out) {
    MB_CELL
    mb_real m = 0.f;
    if (valid) {
        const mb_real* t = D.T + (size_t)i * (DIMS * DIMS + 1);
#pragma unroll
        for (int a = 0; a < DIMS; ++a) {
            mb_real s = 0.f;
#pragma unroll
            for (int c = 0; c < DIMS; ++c) s += t[a * DIMS + c] * u[((size_t)b * DIMS + c) * N + i];
            m = mb_fmax(m, mb_fabs(s));
        }
    }
    if (i < D.NB) {
        const mb_real* t = D.Tb + (size_t)i * (DIMS * DIMS + 1);
#pragma unroll
        for (int a = 0; a < DIMS; ++a) {
            mb_real s = 0.f;
#pragma unroll
            for (int c = 0; c < DIMS; ++c) s += t[a * DIMS + c] * ub[((size_t)b * DIMS + c) * D.NB + i];
            m = mb_fmax(m, mb_fabs(s));
        }
    }
    m = fg_wave_max(m);
    __shared__ mb_real lds[4];
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = mb_fmax(mb_fmax(lds[0], lds[1]), mb_fmax(lds[2], lds[3]));
#if FG_MB_F64
        atomicMax(reinterpret_cast<unsigned long long*>(out) + b, (unsigned long long)__double_as_longlong(m));  // non-negative doubles order like their bits
#else
        atomicMax(reinterpret_cast<int*>(out) + b, __float_as_int(m));  // non-negative floats order like ints
#endif
    }
}

constexpr int MB_SUM_WGS = 8;
__global__ void k_mb_sum(int N, const mb_real* __restrict__ dt, const mb_real* __restrict__ x, mb_real* __restrict__ out) {
    const int b = blockIdx.y;
    if (!mb_active(dt, b)) return;
    mb_real s = 0.f;
    for (int i = blockIdx.x * FG_BLOCK + threadIdx.x; i < N; i += gridDim.x * FG_BLOCK) s += x[(size_t)b * N + i];
    __shared__ mb_real lds[4];
    s = mb_block_sum(s, lds);
    if (threadIdx.x == 0) out[b * MB_SUM_WGS + blockIdx.x] = s;   // one partial per workgroup, summed in index order by k_mb_sub_mean
}
__global__ void k_mb_sub_mean(int N, const mb_real* __restrict__ dt, const mb_real* __restrict__ sum, mb_real* __restrict__ x,
                              mb_real* __restrict__ copy, mb_real* __restrict__ copy_backup = nullptr) {
    const int b = blockIdx.y, i = blockIdx.x * FG_BLOCK + threadIdx.x;
    if (i >= N) return;
    // copy_backup (the first mean removal of a PISO step): what `copy` held before this kernel overwrites it, for EVERY env -- the
    // pressure a dropped env gets back at the end of the step (an env masked out earlier in the step still has it in `copy`)
    if (copy_backup) copy_backup[(size_t)b * N + i] = copy[(size_t)b * N + i];
    if (!mb_active(dt, b)) return;
    const mb_real* ps = sum + b * MB_SUM_WGS;
    const mb_real v = x[(size_t)b * N + i] - (((ps[0] + ps[1]) + (ps[2] + ps[3])) + ((ps[4] + ps[5]) + (ps[6] + ps[7]))) / (mb_real)N;
    x[(size_t)b * N + i] = v;
    if (copy) copy[(size_t)b * N + i] = v;
}
__global__ void k_mb_copy(size_t per_env, const mb_real* __restrict__ dt, const mb_real* __restrict__ src, mb_real* __restrict__ dst) {
    const int b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * FG_BLOCK + threadIdx.x;
    if (i >= per_env || !mb_active(dt, b)) return;
    dst[(size_t)b * per_env + i] = src[(size_t)b * per_env + i];
}


// ---- boundary bookkeeping of Simulation.single_step (simulation.py:206-280) on the flat boundary slots --------------
// update_advective_boundaries (PISOtorch_simulation.py:228-393): u_b <- u_b - t (u_b - u_cell), t = 1 - 1/(1 + 2 dt Minv_b[axis].velm)
template <int DIMS>
__global__ void k_mb_outflow(MbDev D, const mb_real* __restrict__ dt, const mb_real* __restrict__ u, mb_real* __restrict__ ub,
                             int slot0, int count, mb_real v0, mb_real v1, mb_real v2) {
    const int b = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count || !mb_active(dt, b)) return;
    const int sl = slot0 + k;
    const mb_real* t = D.Tb + (size_t)sl * (DIMS * DIMS + 1);
    const int axis = D.bface[sl] >> 1;
    const mb_real velm[3] = {v0, v1, v2};
    mb_real adv = 0.f;
#pragma unroll
    for (int c = 0; c < DIMS; ++c) adv += t[axis * DIMS + c] * velm[c];
    const mb_real w = 1.f - 1.f / (1.f + 2.f * dt[b] * adv);
    const int cell = D.bcell[sl];
#pragma unroll
    for (int c = 0; c < DIMS; ++c) {
        const size_t q = ((size_t)b * DIMS + c) * D.NB + sl;
        ub[q] -= w * (ub[q] - u[((size_t)b * DIMS + c) * D.N + cell]);
    }
}
// signed boundary fluxes (get_fixed_boundary_fluxes, :88-105): out[b][0] = slots outside [slot0, slot0+count), out[b][1] = inside
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mb_bflux(MbDev D, const mb_real* __restrict__ ub, int slot0, int count, int slot0b, int countb,
                                                        mb_real* __restrict__ out) {
    const int b = blockIdx.x;
    mb_real fx = 0.f, fr = 0.f;
    for (int sl = threadIdx.x; sl < D.NB; sl += FG_BLOCK) {
        const mb_real* t = D.Tb + (size_t)sl * (DIMS * DIMS + 1);
        const int f = D.bface[sl], axis = f >> 1;
        mb_real s = 0.f;
#pragma unroll
        for (int c = 0; c < DIMS; ++c) s += t[axis * DIMS + c] * ub[((size_t)b * DIMS + c) * D.NB + sl];
        s *= t[DIMS * DIMS] * ((f & 1) ? 1.f : -1.f);
        if ((sl >= slot0 && sl < slot0 + count) || (sl >= slot0b && sl < slot0b + countb)) fr += s; else fx += s;
    }
    __shared__ mb_real lds[4];
    fx = mb_block_sum(fx, lds);
    fr = mb_block_sum(fr, lds);
    if (threadIdx.x == 0) { out[2 * b] = fx; out[2 * b + 1] = fr; }
}
// balance_boundary_fluxes (:188-224): scale the free boundary so that the total flux vanishes
template <int DIMS>
__global__ void k_mb_balance(MbDev D, const mb_real* __restrict__ dt, const mb_real* __restrict__ sums, mb_real atol,
                             mb_real* __restrict__ ub, int slot0, int count) {
    const int b = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count || !mb_active(dt, b)) return;
    const mb_real fx = sums[2 * b], fr = sums[2 * b + 1];
    if (mb_fabs(fx + fr) <= atol) return;
    const mb_real scale = -fx / fr;
#pragma unroll
    for (int c = 0; c < DIMS; ++c) ub[((size_t)b * DIMS + c) * D.NB + slot0 + k] *= scale;
}
__global__ void k_mb_fill(size_t n, mb_real v, mb_real* __restrict__ x) {
    const size_t i = (size_t)blockIdx.x * FG_BLOCK + threadIdx.x;
    if (i < n) x[i] = v;
}

// envs with a non-finite system leave the step: dt = 0 masks them out of every later kernel of the call (incl. the final
// copy of the velocity result), so their state stays what it was before the step; status 2 is recorded
__global__ __launch_bounds__(FG_BLOCK) void k_mb_restore_failed(int N, const int32_t* __restrict__ fail, const mb_real* __restrict__ src, mb_real* __restrict__ dst) {
    const int b = blockIdx.y, i = blockIdx.x * FG_BLOCK + threadIdx.x;
    if (i < N && fail[b] == 2) dst[(size_t)b * N + i] = src[(size_t)b * N + i];
}
__global__ void k_mb_mask_failed(int B, int nc, const fg_solve_info* __restrict__ info, mb_real* __restrict__ dt, int32_t* __restrict__ fail) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B || !(dt[b] > 0.f)) return;
    bool bad = false;
    for (int c = 0; c < nc; ++c) bad = bad || !info[b * nc + c].is_finite;
    if (bad) { dt[b] = 0.f; fail[b] = 2; }
}

// right diagonal scaling for the preconditioned BiCGStab rung: M' = M D^-1 (unit diagonal), solved for y = D x
template <int DIMS>
__global__ __launch_bounds__(FG_BLOCK) void k_mb_scale_cols(MbDev D, const mb_real* __restrict__ dt, const mb_real* __restrict__ diag,
                                                             const mb_real* __restrict__ off, mb_real* __restrict__ diag_s, mb_real* __restrict__ off_s) {
    MB_CELL
    if (!valid || !mb_active(dt, b)) return;
    constexpr int F = 2 * DIMS;
    diag_s[(size_t)b * N + i] = 1.f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int n = D.nbr[(size_t)f * N + i];
        off_s[((size_t)b * F + f) * N + i] = n >= 0 ? off[((size_t)b * F + f) * N + i] / diag[(size_t)b * N + n] : 0.f;
    }
}
__global__ __launch_bounds__(FG_BLOCK) void k_mb_unscale(int N, int nc, const mb_real* __restrict__ dt, const mb_real* __restrict__ diag, mb_real* __restrict__ x) {
    const int i = blockIdx.x * FG_BLOCK + threadIdx.x, sys = blockIdx.y, b = sys / nc;
    if (i >= N || !mb_active(dt, b)) return;
    x[(size_t)sys * N + i] /= diag[(size_t)b * N + i];
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
extern "C" int fg_mb_create(int32_t dims, int32_t batch, int32_t device, fg_mb_handle* out) {
    FG_REQUIRE(out != nullptr, FG_ERR_INVALID_ARG, "fg_mb_create: out is null");
    FG_REQUIRE(dims == 2 || dims == 3, FG_ERR_INVALID_ARG, "fg_mb_create: dims must be 2 or 3");
    FG_REQUIRE(batch >= 1, FG_ERR_INVALID_ARG, "fg_mb_create: batch must be >= 1");
    if (device >= 0) FG_HIP_CHECK(hipSetDevice(device));
    fg_mb_state* s = new fg_mb_state();
    s->d = dims; s->F = 2 * dims; s->B = batch;
    s->host_only = device < 0;
    {   // debug switches: the only getenv calls of this path, never on the step path
        const char* e = getenv("FG_MB_BICG_VEC4");
        s->dbg_vec_mask = !e ? 31 : ((e[0] == '1' && e[1] == 0) ? 31 : atoi(e));   // bit per kernel: 1 p, 2 v, 4 s, 8 t, 16 x
        s->dbg_scalar_cg = 0;      // (FG_MB_SCALAR_CG until round 5: the one-cell CG kernels are what the fp64 build runs, set below)
        e = getenv("FG_MB_BICG_FUSE"); s->dbg_fuse_st = e ? atoi(e) : 2;   // 0 five kernels, 1 s / t fused, 2 also p / v (default)
        s->dbg_pred = 1;           // (FG_MB_PRED until round 5)
        e = getenv("FG_ADV_JACOBI"); s->adv_jacobi = e ? atoi(e) : 0; s->adv_jacobi_env = e ? 1 : 0;      // (as on the single-block path: fg_api.hip)
        e = getenv("FG_MB_ML_FUSE"); s->dbg_ml_fuse = e ? atoi(e) : 1;
        s->dbg_ml_sb = 0;          // (FG_MB_ML_SB until round 5) systems per workgroup of k_ml_coarse: by batch size
        e = getenv("FG_MB_ML_TRY_CAP"); if (e && atoi(e) > 0) s->dbg_ml_cap = atoi(e);
        s->dbg_graph = 0;          // (FG_MB_GRAPH until round 5: the chunked CG replayed as a hipGraph -- no mesh of the envs takes the chunked CG any more)
        s->dbg_trace = getenv("FG_MB_TRACE") != nullptr;
        if (const char* e = getenv("FG_MB_COMPACT")) s->dbg_compact = atoi(e);
        if (const char* e = getenv("FG_MB_OC_RTG_NT")) s->oc_rtg_nt = atoi(e);   // 1: the register-resident form on 16-24 k cells instead of k_mbc_l2
        s->dbg_fail = getenv("FG_MB_TRACE_FAIL") != nullptr;
        e = getenv("FG_MB_ONCHIP"); s->onchip_mode = (e && e[0] == '0') ? 0 : 1;
        e = getenv("FG_MB_OC_AGG"); s->dbg_oc_agg = (e && e[0] == '0') ? 0 : 1;
        e = getenv("FG_MB_CL_JACOBI"); s->cl_jacobi = (e && e[0] == '0') ? 0 : 1;
        e = getenv("FG_MB_CL_HALF"); s->cl_half = (e && e[0] == '0') ? 0 : 1;
        e = getenv("FG_MB_CL_NEAR"); s->cl_near = (e && e[0] == '0') ? 0 : 1;
        e = getenv("FG_MB_CL_MAXCL"); s->cl_max_clusters = e ? atoi(e) : 0;
        e = getenv("FG_MB_CLUSTER"); s->cl_mode = e ? atoi(e) : 1;   // 0 never, 1 meshes beyond 8 k cells (default), 2 every mesh its tables fit
        e = getenv("FG_MB_RUNG_ILU"); s->dbg_rung_ilu = (e && e[0] == '0') ? 0 : 1;
        e = getenv("FG_MB_OC_VARIANT"); s->oc_variant = e ? atoi(e) : 0;   // bit 0: no compiler fences in the stencil pass; bit 1: split [F][N] coefficient layout
    }
#if FG_MB_F64
    // the fp64 build runs the one-cell-per-thread kernels: the four-cell forms (float4), the on-chip CG and the multilevel
    // preconditioner are written for 32-bit words (fg_mb.h)
    s->dbg_vec_mask = 0; s->dbg_scalar_cg = 1; s->onchip_mode = 0; s->dbg_oc_agg = 0; s->cl_mode = 0;
#endif
    *out = s;
    return FG_OK;
}

extern "C" int fg_mb_config_dump(fg_mb_handle s, char* buf, int n) {
    FG_REQUIRE(s && buf && n > 0, FG_ERR_INVALID_ARG, "fg_mb_config_dump: bad argument");
    char tmp[2048];
    const int len = snprintf(tmp, sizeof(tmp),
        "{\"FG_MB_BICG_VEC4\": %d, \"FG_MB_SCALAR_CG\": %d, \"FG_MB_BICG_FUSE\": %d, \"FG_MB_PRED\": %d, \"FG_MB_ML_FUSE\": %d, \"FG_MB_ML_SB\": %d, "
        "\"FG_MB_ML_TRY_CAP\": %d, \"FG_MB_ML_WARMUP\": %d, \"FG_MB_GRAPH\": %d, \"FG_MB_TRACE\": %d, \"FG_MB_COMPACT\": %d, \"FG_MB_OC_RTG_NT\": %d, "
        "\"FG_MB_ONCHIP\": %d, \"FG_MB_OC_AGG\": %d, \"FG_MB_RUNG_ILU\": %d, \"FG_MB_OC_VARIANT\": %d, \"FG_MB_CLUSTER\": %d, "
        "\"cluster_on\": %d, \"cluster_members_per_thread\": %d, \"cluster_threads\": %d, \"cluster_halo_max\": %d, \"cluster_solves\": %lld, \"cluster_fallbacks\": %lld, \"cluster_jacobi_solves\": %lld}",
        (int)s->dbg_vec_mask, (int)s->dbg_scalar_cg, (int)s->dbg_fuse_st, (int)s->dbg_pred, (int)s->dbg_ml_fuse, (int)s->dbg_ml_sb, (int)s->dbg_ml_cap,
        (int)s->dbg_ml_warmup, (int)s->dbg_graph, (int)s->dbg_trace, (int)s->dbg_compact, (int)s->oc_rtg_nt, (int)s->onchip_mode, (int)s->dbg_oc_agg,
        (int)s->dbg_rung_ilu, (int)s->oc_variant, (int)s->cl_mode, (int)(mb_cluster_wanted(s) ? 1 : 0), (int)s->cl_cpt, (int)s->cl_nt, (int)s->cl_n_halo_max,
        s->cl_solves, s->cl_fallbacks, s->cl_jacobi_solves);
    if (len >= n) return len + 1;
    memcpy(buf, tmp, (size_t)len + 1);
    return FG_OK;
}

extern "C" int fg_mb_destroy(fg_mb_handle s) {
    if (!s) return FG_OK;
    if (s->cg_graph_exec) (void)hipGraphExecDestroy(s->cg_graph_exec);
    if (s->prof_ev[0]) for (int k = 0; k < 64; ++k) (void)hipEventDestroy(s->prof_ev[k]);
    if (s->prof_ev_oc[0]) for (int k = 0; k < 2; ++k) (void)hipEventDestroy(s->prof_ev_oc[k]);
    if (s->capture_stream) (void)hipStreamDestroy(s->capture_stream);
    for (void* p : s->owned) (void)hipFree(p);
    if (s->info_pinned) (void)hipHostFree(s->info_pinned);
    if (s->red_pinned) (void)hipHostFree(s->red_pinned);
    if (s->flags_pinned) (void)hipHostFree(s->flags_pinned);
    if (s->jac_res_pinned) (void)hipHostFree(s->jac_res_pinned);
    fg_poll_destroy(&s->poll);
    if (s->sys_map_pinned) (void)hipHostFree(s->sys_map_pinned);
    if (s->red2_pinned) (void)hipHostFree(s->red2_pinned);
    if (s->dt_pinned) (void)hipHostFree(s->dt_pinned);
    if (s->env_fail_pinned) (void)hipHostFree(s->env_fail_pinned);
    delete s;
    return FG_OK;
}

extern "C" int fg_mb_add_block(fg_mb_handle s, const mb_real* coords, int32_t nx, int32_t ny, int32_t nz, int32_t* block_id) {
    FG_REQUIRE(s && coords, FG_ERR_INVALID_ARG, "fg_mb_add_block: null argument");
    FG_REQUIRE(!s->finalized, FG_ERR_INVALID_ARG, "fg_mb_add_block: domain already finalized");
    if (s->d == 2) nz = 1;
    FG_REQUIRE(nx >= 3 && ny >= 3 && (s->d == 2 || nz >= 3), FG_ERR_INVALID_ARG,
               "fg_mb_add_block: every spatial dimension must be at least 3 cells (domain_structs.cpp Block::Block)");
    MbBlock b;
    b.size[0] = nx; b.size[1] = ny; b.size[2] = nz;
    b.ncells = nx * ny * nz;
    const size_t nv = (size_t)(nx + 1) * (ny + 1) * (s->d == 3 ? nz + 1 : 1) * s->d;
    b.coords.resize(nv);
    for (size_t k = 0; k < nv; ++k) b.coords[k] = (double)coords[k];
    s->blocks.push_back(std::move(b));
    if (block_id) *block_id = (int32_t)s->blocks.size() - 1;
    return FG_OK;
}

extern "C" int fg_mb_connect(fg_mb_handle s, int32_t b1, int32_t face1, int32_t b2, int32_t face2, int32_t axis1, int32_t axis2) {
    FG_REQUIRE(s && !s->finalized, FG_ERR_INVALID_ARG, "fg_mb_connect: null or finalized handle");
    const int nb = (int)s->blocks.size(), d = s->d;
    FG_REQUIRE(b1 >= 0 && b1 < nb && b2 >= 0 && b2 < nb, FG_ERR_INVALID_ARG, "fg_mb_connect: block index out of range");
    FG_REQUIRE(face1 >= 0 && face1 < 2 * d && face2 >= 0 && face2 < 2 * d && axis1 >= 0 && axis1 < 2 * d, FG_ERR_INVALID_ARG,
               "fg_mb_connect: face / axis index out of range");
    // ConnectBlocks (domain_structs.cpp:1080-1113)
    int axes1[3] = {face2, 0, 0}, axes2[3] = {face1, 0, 0};
    if (d > 1) {
        axes1[1] = axis1;
        const int f1d = face1 >> 1, f2d = face2 >> 1;
        bool swapped = false;
        if (d == 2 || (axis1 >> 1) == (f2d + 1) % d) {
            axes2[1] = (((f1d + 1) % d) << 1) | (axis1 & 1);
        } else {
            FG_REQUIRE((axis2 >> 1) == (f2d + 1) % d, FG_ERR_INVALID_ARG, "fg_mb_connect: invalid connection");
            axes2[1] = (((f1d + 2) % d) << 1) | (axis2 & 1);
            swapped = true;
        }
        if (d > 2) {
            axes1[2] = axis2;
            axes2[2] = !swapped ? ((((f1d + 2) % d) << 1) | (axis2 & 1)) : ((((f1d + 1) % d) << 1) | (axis1 & 1));
        }
    }
    // the connected faces must have matching extents
    for (int k = 1; k < d; ++k) {
        const int a1 = ((face1 >> 1) + k) % d, a2 = axes1[k] >> 1;
        FG_REQUIRE(s->blocks[b1].size[a1] == s->blocks[b2].size[a2], FG_ERR_INVALID_ARG,
                   "fg_mb_connect: the connected faces differ in resolution");
    }
    MbBound& x = s->blocks[b1].bounds[face1];
    x.type = FG_MB_CONNECTED; x.other = b2;
    MbBound& y = s->blocks[b2].bounds[face2];
    y.type = FG_MB_CONNECTED; y.other = b1;
    for (int k = 0; k < 3; ++k) { x.axes[k] = axes1[k]; y.axes[k] = axes2[k]; }
    return FG_OK;
}

extern "C" int fg_mb_make_periodic(fg_mb_handle s, int32_t block, int32_t axis) {
    FG_REQUIRE(s && !s->finalized, FG_ERR_INVALID_ARG, "fg_mb_make_periodic: null or finalized handle");
    FG_REQUIRE(block >= 0 && block < (int)s->blocks.size() && axis >= 0 && axis < s->d, FG_ERR_INVALID_ARG,
               "fg_mb_make_periodic: index out of range");
    s->blocks[block].bounds[2 * axis].type = FG_MB_PERIODIC;
    s->blocks[block].bounds[2 * axis + 1].type = FG_MB_PERIODIC;
    return FG_OK;
}

extern "C" int fg_mb_set_reference_quirks(fg_mb_handle s, int32_t connected_diagonal_offset, int32_t first_layer_rule) {
    FG_REQUIRE(s && !s->finalized, FG_ERR_INVALID_ARG, "fg_mb_set_reference_quirks: null or finalized handle");
    s->quirk_diag_offset = connected_diagonal_offset ? 1 : 0;
    s->quirk_first_layer = first_layer_rule ? 1 : 0;
    return FG_OK;
}

extern "C" int fg_mb_set_nonortho_flags(fg_mb_handle s, int32_t flags) {
    FG_REQUIRE(s && !s->finalized, FG_ERR_INVALID_ARG, "fg_mb_set_nonortho_flags: null or finalized handle");
    FG_REQUIRE(flags == 25 || flags == 10, FG_ERR_UNSUPPORTED,
               "fg_mb_set_nonortho_flags: 25 (CENTER_MATRIX | DIRECT_MATRIX | DIAGONAL_RHS) or 10 (DIRECT_RHS | DIAGONAL_RHS)");
    s->nonortho_flags = flags;
    return FG_OK;
}

extern "C" int fg_mb_finalize(fg_mb_handle s) {
    FG_REQUIRE(s && !s->finalized && !s->blocks.empty(), FG_ERR_INVALID_ARG, "fg_mb_finalize: nothing to finalize");
    if (int rc = fg_mb_build_tables(s)) return rc;
    if (s->host_only) { s->finalized = true; return FG_OK; }
    const size_t B = s->B, N = s->N, NB = s->NB, d = s->d, F = s->F;
    if (int rc = mb_alloc(s, &s->cc, B * d * N)) return rc;
    if (int rc = mb_alloc(s, &s->fb, B * NB)) return rc;
    if (int rc = mb_alloc(s, &s->Cdiag, B * N)) return rc;
    if (int rc = mb_alloc(s, &s->Coff, B * F * N)) return rc;
    if (int rc = mb_alloc(s, &s->rA, B * N)) return rc;
    if (int rc = mb_alloc(s, &s->rhs, B * d * N)) return rc;
    if (int rc = mb_alloc(s, &s->ures, B * d * N)) return rc;
    if (int rc = mb_alloc(s, &s->hvec, B * d * N)) return rc;
    if (int rc = mb_alloc(s, &s->div, B * N)) return rc;
    if (int rc = mb_alloc(s, &s->Pdiag, B * N)) return rc;
    if (int rc = mb_alloc(s, &s->Poff, B * F * N)) return rc;
    if (int rc = mb_alloc(s, &s->pres, B * N)) return rc;
    if (int rc = mb_alloc(s, &s->Sdiag, B * N)) return rc;      // column-scaled matrix of the preconditioned BiCGStab rung
    if (int rc = mb_alloc(s, &s->Soff, B * F * N)) return rc;
    for (int k = 0; k < 8; ++k)
        if (int rc = mb_alloc(s, &s->w[k], B * d * N)) return rc;
    if (int rc = mb_alloc(s, &s->acc, B * d * MB_ACC)) return rc;
    if (int rc = mb_alloc(s, &s->sc, B * d * 2)) return rc;
    if (int rc = mb_alloc(s, &s->flags, B * d)) return rc;
    if (int rc = mb_alloc(s, &s->info_dev, B * d)) return rc;
    if (int rc = mb_alloc(s, &s->red, B)) return rc;
    if (int rc = mb_alloc(s, &s->red8, B * MB_SUM_WGS)) return rc;
    if (int rc = mb_alloc(s, &s->pres_bak, B * N)) return rc;
    if (int rc = mb_alloc(s, &s->best_it, B * d)) return rc;
    if (int rc = mb_alloc(s, &s->yproj, N)) return rc;
    {
        std::vector<mb_real> ones(N, 1.f / std::sqrt((mb_real)N));
        FG_HIP_CHECK(hipMemcpy(s->yproj, ones.data(), sizeof(mb_real) * N, hipMemcpyHostToDevice));
        s->dev.yproj = s->yproj;
    }
    if (int rc = mb_alloc(s, &s->red2, 2 * B)) return rc;
    if (int rc = mb_alloc(s, &s->it_ctr, 4)) return rc;
    if (int rc = mb_alloc(s, &s->dt_dev, B)) return rc;
    if (int rc = mb_alloc(s, &s->dt_step, B)) return rc;
    if (int rc = mb_alloc(s, &s->oc_dbg, (size_t)16)) return rc;
    if (int rc = mb_alloc(s, &s->env_fail, B)) return rc;
    // fp64 iterate + best refinement point of the refined BiCGStab: allocated here, nothing is allocated on the step path
    if (int rc = mb_alloc(s, &s->x64, B * d * N)) return rc;        // B * d systems: the fp64 rung also serves the velocity solves
    if (int rc = mb_alloc(s, &s->x64_best, B * d * N)) return rc;
    if (int rc = mb_alloc(s, &s->best_res, B * d)) return rc;
    if (int rc = mb_alloc(s, &s->best_keep, B * d)) return rc;
    if (N < 65535) {   // packed neighbour table of the on-chip CG
        std::vector<uint32_t> packed((size_t)(F / 2) * N);
        for (size_t w = 0; w < F / 2; ++w)
            for (size_t i = 0; i < N; ++i) {
                const int32_t n0 = s->h_nbr[(2 * w) * N + i], n1 = s->h_nbr[(2 * w + 1) * N + i];
                packed[i * (F / 2) + w] = (uint32_t)(n0 >= 0 ? n0 : 0xffff) | ((uint32_t)(n1 >= 0 ? n1 : 0xffff) << 16);
            }
        if (int rc = mb_alloc(s, &s->nbr16, packed.size())) return rc;
        FG_HIP_CHECK(hipMemcpy(s->nbr16, packed.data(), sizeof(uint32_t) * packed.size(), hipMemcpyHostToDevice));
        if (d == 2 && N <= (size_t)28 * 1024 && !FG_MB_F64)
            if (int rc = mb_alloc(s, &s->Poff4, B * N * 4)) return rc;
    }
    s->env_status.assign(B, 0);
    FG_HIP_CHECK(hipHostMalloc((void**)&s->env_fail_pinned, sizeof(int32_t) * B, hipHostMallocDefault));
    FG_HIP_CHECK(hipHostMalloc((void**)&s->info_pinned, sizeof(fg_solve_info) * B * d, hipHostMallocDefault));
    FG_HIP_CHECK(hipHostMalloc((void**)&s->red_pinned, sizeof(mb_real) * B, hipHostMallocDefault));
    FG_HIP_CHECK(hipHostMalloc((void**)&s->red2_pinned, sizeof(mb_real) * 2 * B, hipHostMallocDefault));
    FG_HIP_CHECK(hipHostMalloc((void**)&s->dt_pinned, sizeof(mb_real) * 2 * B, hipHostMallocDefault));   // two halves (fg_mb_single_step)
    FG_HIP_CHECK(hipHostMalloc((void**)&s->flags_pinned, sizeof(int32_t) * B * d, hipHostMallocDefault));
    FG_HIP_CHECK(hipHostMalloc((void**)&s->jac_res_pinned, sizeof(mb_real) * 2 * B * d, hipHostMallocDefault));
    if (int rc = fg_poll_create(&s->poll, B * d > 2 * B ? B * d : 2 * B)) return rc;
    if (int rc = mb_alloc(s, &s->sys_map_dev, (size_t)B * d)) return rc;
    FG_HIP_CHECK(hipHostMalloc((void**)&s->sys_map_pinned, sizeof(int32_t) * B * d, hipHostMallocDefault));
    if (int rc = mb_alloc(s, &s->verified, (size_t)B * d)) return rc;
    s->finalized = true;
    return FG_OK;
}

extern "C" int fg_mb_sizes(fg_mb_handle s, int32_t* n_cells, int32_t* n_boundary_faces) {
    FG_REQUIRE(s && s->finalized, FG_ERR_INVALID_ARG, "fg_mb_sizes: domain not finalized");
    if (n_cells) *n_cells = s->N;
    if (n_boundary_faces) *n_boundary_faces = s->NB;
    return FG_OK;
}

extern "C" int fg_mb_block_info(fg_mb_handle s, int32_t block, int32_t* cell_offset, int32_t* boundary_slot0 /*[2d]*/) {
    FG_REQUIRE(s && s->finalized && block >= 0 && block < (int)s->blocks.size(), FG_ERR_INVALID_ARG, "fg_mb_block_info: bad argument");
    if (cell_offset) *cell_offset = s->blocks[block].offset;
    if (boundary_slot0)
        for (int f = 0; f < s->F; ++f)
            boundary_slot0[f] = s->blocks[block].bounds[f].type == FG_MB_FIXED ? s->blocks[block].bounds[f].slot0 : -1;
    return FG_OK;
}

extern "C" int fg_mb_bind(fg_mb_handle s, mb_real* velocity, mb_real* pressure_result, mb_real* boundary_velocity, const mb_real* source) {
    FG_REQUIRE(s && s->finalized, FG_ERR_INVALID_ARG, "fg_mb_bind: domain not finalized");
    FG_REQUIRE(!s->host_only, FG_ERR_UNSUPPORTED, "fg_mb_bind: this handle was created host-only (device < 0): tables only, no compute");
    FG_REQUIRE(velocity && pressure_result && (boundary_velocity || s->NB == 0), FG_ERR_INVALID_ARG, "fg_mb_bind: null field");
    s->velocity = velocity; s->pressure = pressure_result; s->bvel = boundary_velocity; s->source = source;
    return FG_OK;
}

// ---- drag / lift on a closed wall (envs/util/forces.py:193-377 of the reference, compute_forces_2d / _3d): traction
// (2 nu S - p I) n on every wall face, S from the one-sided normal derivative (cell - wall) / distance and a central tangential
// derivative over the ring of wall-adjacent cells, times the face length (area), summed over the ring.  One workgroup per (layer,
// env); the host code did this with ~40 tensor ops per sim step (290 us of host time against a 1.4 ms PISO step).
__global__ __launch_bounds__(FG_BLOCK) void k_mb_wall_forces(int d, int N, int NB, int n, int layers, const mb_real* __restrict__ u,
                                                              const mb_real* __restrict__ ub, const mb_real* __restrict__ p,
                                                              const int32_t* __restrict__ cell_index, const int32_t* __restrict__ slot_index,
                                                              const mb_real* __restrict__ geom, mb_real area_scale, mb_real nu,
                                                              mb_real* __restrict__ out) {
    __shared__ mb_real lds[8];
    const int layer = blockIdx.x, b = blockIdx.y;
    const mb_real* ue = u + (size_t)b * d * N;
    const mb_real* ube = ub + (size_t)b * d * NB;
    const mb_real* pe = p + (size_t)b * N;
    const int32_t* ci = cell_index + (size_t)layer * n;
    const int32_t* si = slot_index + (size_t)layer * n;
    mb_real f[2] = {0.f, 0.f};
    for (int j = threadIdx.x; j < n; j += FG_BLOCK) {
        const int c = ci[j], cl = ci[j + 1 == n ? 0 : j + 1], cr = ci[j == 0 ? n - 1 : j - 1], sl = si[j];   // roll(-1) = "left", roll(+1) = "right"
        const mb_real nx = geom[j], ny = geom[n + j], tl = geom[2 * n + j], wd = geom[3 * n + j], fl = geom[4 * n + j] * area_scale;
        const mb_real tx = ny, ty = -nx;
        mb_real dn[2], dt[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            dn[q] = (ue[(size_t)q * N + c] - ube[(size_t)q * NB + sl]) / wd;
            dt[q] = (ue[(size_t)q * N + cr] - ue[(size_t)q * N + cl]) / (2.f * tl);
        }
        const mb_real du_dx = dn[0] * nx + dt[0] * tx, du_dy = dn[0] * ny + dt[0] * ty;
        const mb_real dv_dx = dn[1] * nx + dt[1] * tx, dv_dy = dn[1] * ny + dt[1] * ty;
        const mb_real sxy = 0.5f * (du_dy + dv_dx), two_nu = 2.f * nu, pc = pe[c];
        f[0] += ((two_nu * du_dx - pc) * nx + two_nu * sxy * ny) * fl;
        f[1] += (two_nu * sxy * nx + (two_nu * dv_dy - pc) * ny) * fl;
    }
    mb_block_sums<2>(f, lds);
    if (threadIdx.x == 0) {
        out[((size_t)b * 2 + 0) * layers + layer] = f[0];
        out[((size_t)b * 2 + 1) * layers + layer] = f[1];
    }
}

extern "C" int fg_mb_wall_forces(fg_mb_handle s, const int32_t* cell_index, const int32_t* slot_index, const mb_real* geom, int32_t n,
                                 int32_t layers, mb_real area_scale, mb_real viscosity, mb_real* out, void* stream) {
    FG_REQUIRE(s && s->finalized && s->velocity && s->pressure && s->bvel, FG_ERR_NOT_BOUND, "fg_mb_wall_forces: fields not bound");
    FG_REQUIRE(cell_index && slot_index && geom && out && n > 0 && layers > 0, FG_ERR_INVALID_ARG, "fg_mb_wall_forces: bad argument");
    hipLaunchKernelGGL(k_mb_wall_forces, dim3(layers, s->B), dim3(FG_BLOCK), 0, (hipStream_t)stream, s->d, s->N, s->NB, n, layers,
                       (const mb_real*)s->velocity, (const mb_real*)s->bvel, (const mb_real*)s->pressure, cell_index, slot_index, geom, area_scale,
                       viscosity, out);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

extern "C" int fg_mb_set_viscosity(fg_mb_handle s, mb_real nu) {
    FG_REQUIRE(s && nu > 0.f, FG_ERR_INVALID_ARG, "fg_mb_set_viscosity: viscosity must be positive");
    s->nu = nu;
    return FG_OK;
}

extern "C" int fg_mb_piso_step(fg_mb_handle s, const mb_real* dt_B, const fg_mb_step_options* opt, int32_t* stats_host, void* stream) {
    FG_REQUIRE(s && s->finalized && s->velocity, FG_ERR_NOT_BOUND, "fg_mb_piso_step: fields not bound");
    FG_REQUIRE(dt_B && opt, FG_ERR_INVALID_ARG, "fg_mb_piso_step: null argument");
    FG_REQUIRE(s->nu > 0.f, FG_ERR_INVALID_ARG, "fg_mb_piso_step: viscosity not set");
    hipStream_t st = (hipStream_t)stream;
    const int B = s->B, N = s->N, d = s->d, NB = s->NB;
    const int cells = std::max(N, NB);
    const dim3 blk(FG_BLOCK), gc((cells + FG_BLOCK - 1) / FG_BLOCK, B), gn((N + FG_BLOCK - 1) / FG_BLOCK, B),
        gv((N + FG_BLOCK - 1) / FG_BLOCK, B, d);
    const MbDev& D = s->dev;
    int soft_rc = FG_OK;
    auto soft = [&](int rc) {  // non-convergence is reported, everything else aborts the step
        if (rc == FG_ERR_NOT_CONVERGED || rc == FG_ERR_NOT_FINITE) { if (soft_rc == FG_OK || rc == FG_ERR_NOT_FINITE) soft_rc = rc; return FG_OK; }
        return rc;
    };
    int its[4] = {0, 0, 0, 0};
    // working copy of dt: envs whose solve turns out non-finite are masked out of the rest of the step (k_mb_mask_failed).
    // (fg_mb_single_step uploads its time steps straight into the working copy; env_fail is all zeros between steps: the rare path
    //  below that reads it clears it again -- two launches per PISO step less)
    if (dt_B != s->dt_step) FG_HIP_CHECK(hipMemcpyAsync(s->dt_step, dt_B, sizeof(mb_real) * B, hipMemcpyDeviceToDevice, st));
    dt_B = s->dt_step;
    auto mask_failed = [&](int nc) {
        hipLaunchKernelGGL(k_mb_mask_failed, dim3((B + 63) / 64), dim3(64), 0, st, B, nc, (const fg_solve_info*)s->info_dev, s->dt_step, s->env_fail);
    };
    // pressure of the envs at the start of the step: a dropped env gets it back at the end (its velocity is never committed; its
    // pressure is written by every corrector's mean removal, so a failure in corrector 1 would leave corrector 0's behind).  The
    // first mean removal of the step saves it while it overwrites it (k_mb_sub_mean: no copy launch of its own)
    bool pressure_saved = false;
    const size_t vel_env = (size_t)d * N;
    const dim3 gcopy((unsigned)((vel_env + FG_BLOCK - 1) / FG_BLOCK), B);
    MB_DISPATCH(s, {
        // ---- predictor (SIM.py:1646-1762, non-orthogonal branch)
        hipLaunchKernelGGL(k_mb_contra<DIMS>, gc, blk, 0, st, D, dt_B, s->velocity, s->bvel, s->cc, s->fb);
        hipLaunchKernelGGL(k_mb_matrix<DIMS>, gn, blk, 0, st, D, dt_B, s->nu, s->cc, s->fb, s->Cdiag, s->Coff, s->rA);
        hipLaunchKernelGGL(k_mb_copy, gcopy, blk, 0, st, vel_env, dt_B, s->velocity, s->ures);  // CopyVelocityResultFromBlocks
        for (int no = 0; no < opt->advect_non_ortho_steps; ++no) {
            FgRange range_vel("mb_velocity_solve");
            hipLaunchKernelGGL(k_mb_vrhs<DIMS>, gv, blk, 0, st, D, dt_B, s->nu, s->velocity, s->ures, s->bvel, s->fb, s->source, s->rhs);
            int m = 0;
            // initial guess: zero on the first non-orthogonal pass, the previous pass's result after that (x = None if no_step == 0
            // or not advect_non_ortho_reuse_result, PISOtorch_simulation.py:1735-1742; tests/golden/reference_split_step.json);
            // fg_mb_set_advection_start(1): the current velocity on the first pass (opt-in)
            int vrc = FG_OK, jac = 0;
            const int start_x0 = (no > 0 || s->adv_from_result) ? 1 : 0;
            // policy advection_jacobi: the sweeps first (mb_jacobi); what they do not settle goes to BiCGStab from a cleared start vector
            if (s->adv_jacobi && s->jac_res_pinned) {
                vrc = mb_jacobi(s, dt_B, s->Cdiag, s->Coff, s->rhs, s->ures, d, opt->advection_tol, start_x0, &m, st, no & 3, &jac);
                if (jac == 1) s->jac_solves += 1;
                else if (jac >= 2) s->jac_fallbacks += 1;
            }
            if (jac != 1)      // (after sweeps that gave up: from their last iterate when it is finite -- closer than either start vector)
                vrc = mb_bicgstab(s, dt_B, s->Cdiag, s->Coff, s->rhs, s->ures, d, opt->advection_tol,
                                  opt->max_iterations, jac == 2 ? 0 : (jac == 3 ? 1 : start_x0), &m, st, 0, 0, 0, no & 3);
            // ---- the reference's retry ladder (_linear_solve, PISOtorch_diff.py:410-476).  The advection solve runs without
            // returnBestResult, so "not solved" = any system unconverged (or non-finite); every rung starts from zero
            // ("do not start with a possibly corrupted result tensor", :429-431)
            auto v_failed = [](int rc) { return rc == FG_ERR_NOT_FINITE || rc == FG_ERR_NOT_CONVERGED; };
            if ((s->ladder_force & 1) && vrc == FG_OK) vrc = FG_ERR_NOT_CONVERGED;
            if (v_failed(vrc) && opt->solver_double_fallback) {   // fp64 rung: fp64 iterate, residual b - C x recomputed in fp64
                ++s->ladder[0];
                vrc = mb_bicgstab(s, dt_B, s->Cdiag, s->Coff, s->rhs, s->ures, d, opt->advection_tol, opt->max_iterations, 0, &m, st, 0, 1);
                if ((s->ladder_force & 4) && vrc == FG_OK) vrc = FG_ERR_NOT_CONVERGED;
            }
            if (v_failed(vrc) && opt->bicg_precondition_fallback) {
                ++s->ladder[1];
                if (s->dbg_rung_ilu && !FG_MB_F64 && mb_ilu_prepare(s)) {
                    // preconditioned rung as in the reference: BiCGStab right-preconditioned by ILU(0) of the matrix
                    vrc = mb_bicgstab(s, dt_B, s->Cdiag, s->Coff, s->rhs, s->ures, d, opt->advection_tol, opt->max_iterations, 0, &m, st, 0, 0, 2);
                } else {   // meshes the schedule does not cover, the fp64 build: right diagonal scaling, (C D^-1) y = b, x = D^-1 y
                    hipLaunchKernelGGL(k_mb_scale_cols<DIMS>, gn, blk, 0, st, D, dt_B, s->Cdiag, s->Coff, s->Sdiag, s->Soff);
                    vrc = mb_bicgstab(s, dt_B, s->Sdiag, s->Soff, s->rhs, s->ures, d, opt->advection_tol, opt->max_iterations, 0, &m, st);
                    hipLaunchKernelGGL(k_mb_unscale, dim3((N + FG_BLOCK - 1) / FG_BLOCK, B * d), blk, 0, st, N, d, dt_B, (const mb_real*)s->Cdiag, s->ures);
                }
            }
            if (vrc == FG_ERR_NOT_FINITE) {
                fg_set_error("fg_mb_piso_step: the velocity (BiCGStab) solve produced a non-finite residual");
                mask_failed(d);   // solve_ok = False before CopyVelocityResultToBlocks (PISOtorch_simulation.py:1752-1757): state intact
            }
            if (int rc = soft(vrc)) return rc;
            its[1] = std::max(its[1], m);
            s->ctr.add(1, s->info_pinned, B * d);
        }
        // ---- correctors (SIM.py:1777-1972)
        for (int c = 0; c < opt->corrector_steps; ++c) {
            // P depends on A = diag(C) and the mesh only: the same matrix in every corrector of a step (the reference rebuilds it,
            // SetupPressureMatrix inside the loop, PISOtorch_simulation.py:1790-1800, to the same values)
            if (c == 0) { hipLaunchKernelGGL(k_mb_pmatrix<DIMS>, gn, blk, 0, st, D, dt_B, s->rA, s->Pdiag, s->Poff, s->Poff4, SLOTS.cell_slot, SLOTS.off4, SLOTS.diag, SLOTS.stride); mb_slots_written(s); }
            for (int ps = 0; ps < opt->pressure_non_ortho_steps; ++ps) {
                FgRange range_p("mb_pressure_solve");
                if (ps == 0) {
                    hipLaunchKernelGGL(k_mb_h<DIMS>, gv, blk, 0, st, D, dt_B, s->nu, s->rA, s->Coff, s->velocity, s->ures,
                                       s->bvel, s->fb, s->source, s->hvec);
                    hipLaunchKernelGGL(k_mb_contra<DIMS>, gc, blk, 0, st, D, dt_B, s->hvec, s->bvel, s->cc, (mb_real*)nullptr);
                }
                hipLaunchKernelGGL(k_mb_div<DIMS>, gn, blk, 0, st, D, dt_B, s->cc, s->fb, s->rA, s->pressure, 1, s->div);
                int m = 0;
                const int warm = (ps > 0 || opt->pressure_warm_start) ? 1 : 0;
                if (ps == 0 && warm)  // start from the pressure field (the previous solve's result, or a restored state)
                    hipLaunchKernelGGL(k_mb_copy, dim3((N + FG_BLOCK - 1) / FG_BLOCK, B), blk, 0, st, (size_t)N, dt_B, s->pressure, s->pres);
                auto solve = [&](int use_x0, int force_cg = 0) {
                    return (opt->pressure_use_bicgstab && !force_cg)
                               ? mb_pressure_bicgstab(s, dt_B, opt->pressure_tol, opt->max_iterations, use_x0, &m, st, opt->pressure_project_mean,
                                                      opt->pressure_use_bicgstab == 2, 4 + 4 * (c & 1) + (ps & 3))
                               : mb_cg(s, dt_B, s->Pdiag, s->Poff, s->div, s->pres, opt->pressure_tol, opt->max_iterations, use_x0,
                                       opt->pressure_project_mean, opt->pressure_stall_accept, &m, st);
                };
                int prc = solve(warm);
                if ((s->ladder_force & 2) && prc == FG_OK) prc = FG_ERR_NOT_FINITE;
                // a warm-started solve that ends unconverged falls back to the reference's start from zero
                if (prc == FG_ERR_NOT_CONVERGED && ps == 0 && warm) prc = solve(0);
                // ---- retry ladder (PISOtorch_diff.py:410-476): pressure solves run with returnBestResult, so only a NON-FINITE
                // residual counts as "not solved".  fp64 rung: the refined BiCGStab (fp64 iterate and residual) from zero -- the
                // reference repeats CG in fp64, which on the non-symmetric matrix stalls exactly like fp32 CG
                if (prc == FG_ERR_NOT_FINITE && opt->solver_double_fallback && opt->pressure_use_bicgstab != 2) {
                    ++s->ladder[2];
                    prc = mb_bicgstab(s, dt_B, s->Pdiag, s->Poff, s->div, s->pres, 1, opt->pressure_tol, opt->max_iterations, 0, &m, st,
                                      opt->pressure_project_mean, 1);
                }
                // BiCGStab keeps no best iterate: a solve of it that broke down or ran out of iterations is repeated with CG,
                // which hands back its best iterate (last resort; the refined solver hands back its best refinement point itself)
                if (opt->pressure_use_bicgstab && (prc == FG_ERR_NOT_FINITE || (prc == FG_ERR_NOT_CONVERGED && opt->pressure_use_bicgstab != 2))) {
                    ++s->ladder[3];
                    prc = solve(0, 1);
                }
                if (prc == FG_ERR_NOT_FINITE) {
                    fg_set_error("fg_mb_piso_step: the pressure solve produced a non-finite residual");
                    mask_failed(1);
                }
                if (int rc = soft(prc)) return rc;
                if (c < 2) { its[2 + c] = std::max(its[2 + c], m); s->ctr.add(2 + c, s->info_pinned, B); }
                hipLaunchKernelGGL(k_mb_sum, dim3(MB_SUM_WGS, B), blk, 0, st, N, dt_B, s->pres, s->red8);
                hipLaunchKernelGGL(k_mb_sub_mean, gn, blk, 0, st, N, dt_B, s->red8, s->pres, s->pressure, pressure_saved ? (mb_real*)nullptr : s->pres_bak);
                pressure_saved = true;
            }
            // the last corrector also writes the block velocity of active envs: CopyVelocityResultToBlocks without a pass of its own
            hipLaunchKernelGGL(k_mb_correct<DIMS>, gn, blk, 0, st, D, dt_B, s->rA, s->hvec, s->pressure, s->ures,
                               (c + 1 == opt->corrector_steps) ? s->velocity : (mb_real*)nullptr);
        }
        if (opt->corrector_steps <= 0) hipLaunchKernelGGL(k_mb_copy, gcopy, blk, 0, st, vel_env, dt_B, s->ures, s->velocity);  // CopyVelocityResultToBlocks
        if (!pressure_saved) hipLaunchKernelGGL(k_mb_copy, dim3((unsigned)((N + FG_BLOCK - 1) / FG_BLOCK), B), blk, 0, st, (size_t)N, (const mb_real*)nullptr, (const mb_real*)s->pressure, s->pres_bak);
    });
    FG_HIP_CHECK(hipGetLastError());
    if (stats_host) for (int k = 0; k < 4; ++k) stats_host[k] = its[k];
    s->ctr.piso_steps += 1;
    for (int b = 0; b < B; ++b) s->env_status[b] = 0;
    if (soft_rc == FG_ERR_NOT_FINITE) {   // rare path: which envs were dropped; their pressure goes back to what it was
        hipLaunchKernelGGL(k_mb_restore_failed, dim3((unsigned)((N + FG_BLOCK - 1) / FG_BLOCK), B), blk, 0, st, N, (const int32_t*)s->env_fail, (const mb_real*)s->pres_bak, s->pressure);
        FG_HIP_CHECK(hipMemcpyAsync(s->env_fail_pinned, s->env_fail, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
        FG_HIP_CHECK(hipMemsetAsync(s->env_fail, 0, sizeof(int32_t) * B, st));
        FG_HIP_CHECK(hipStreamSynchronize(st));
        for (int b = 0; b < B; ++b) s->env_status[b] = s->env_fail_pinned[b];
    } else if (soft_rc == FG_ERR_NOT_CONVERGED) {
        for (int b = 0; b < B; ++b) s->env_status[b] = 1;   // some solve of the batch ended on its best iterate
    }
    return soft_rc;
}

extern "C" int fg_mb_solver_unconverged(fg_mb_handle s, int64_t* out4) {
    FG_REQUIRE(s != nullptr && out4 != nullptr, FG_ERR_INVALID_ARG, "fg_mb_solver_unconverged: bad argument");
    for (int k = 0; k < 4; ++k) out4[k] = s->ctr.unconv[k];
    return FG_OK;
}

extern "C" int fg_mb_solver_counters(fg_mb_handle s, int64_t* out13, int32_t reset) {
    FG_REQUIRE(s != nullptr, FG_ERR_INVALID_ARG, "fg_mb_solver_counters: null handle");
    if (out13) s->ctr.write(out13);
    if (reset) s->ctr.reset();
    return FG_OK;
}

// Tables of the multilevel preconditioner of the on-chip pressure CG (built on the host from the geometry-only pressure matrix,
// simulation/multiblock.py::set_pressure_multilevel); a4 / parent4 as int32 on the host, stored as 16-bit on the device.
extern "C" int fg_mb_set_multilevel(fg_mb_handle s, int32_t n4, int32_t n8, const int32_t* a4_host, const int32_t* parent4_host,
                                    const int32_t* rect4_host, const mb_real* d4g_host, const mb_real* aci8_host, mb_real geom_diag_sum,
                                    int32_t enable) {
    FG_REQUIRE(s && s->finalized && !s->host_only, FG_ERR_INVALID_ARG, "fg_mb_set_multilevel: domain not finalized (or host-only)");
    FG_REQUIRE(!FG_MB_F64, FG_ERR_UNSUPPORTED, "fg_mb_set_multilevel: the multilevel preconditioner is not part of the fp64 build (plain recurrences there)");
    if (!a4_host) { s->ml_on = enable && s->ml_a4 != nullptr; return FG_OK; }   // switch only
    FG_REQUIRE(s->d == 2 && n4 > 0 && n4 < 65535 && n8 > 0 && n8 <= ML_N8_MAX && parent4_host && rect4_host && d4g_host && aci8_host &&
                   geom_diag_sum != 0.f,
               FG_ERR_INVALID_ARG, "fg_mb_set_multilevel: 2-D meshes with at most 65534 / 2048 aggregates");
    std::vector<uint16_t> a4(s->N), p4(n4);
    for (int i = 0; i < s->N; ++i) { FG_REQUIRE(a4_host[i] >= 0 && a4_host[i] < n4, FG_ERR_INVALID_ARG, "fg_mb_set_multilevel: aggregate id out of range"); a4[i] = (uint16_t)a4_host[i]; }
    for (int a = 0; a < n4; ++a) { FG_REQUIRE(parent4_host[a] >= 0 && parent4_host[a] < n8, FG_ERR_INVALID_ARG, "fg_mb_set_multilevel: parent id out of range"); p4[a] = (uint16_t)parent4_host[a]; }
    // every aggregate must be the rectangle it is declared as (the kernel sums it by its shape), and an 8 x 8 aggregate has at
    // most four children
    std::vector<uint2> rect(n4), child(n8, make_uint2(0xffffffffu, 0xffffffffu));
    std::vector<uint32_t> pos4(n4);
    {
        std::vector<int> count(n4, 0);
        for (int i = 0; i < s->N; ++i) count[a4[i]]++;
        for (int a = 0; a < n4; ++a) {
            const int first = rect4_host[4 * a], w = rect4_host[4 * a + 1], h = rect4_host[4 * a + 2], stride = rect4_host[4 * a + 3];
            FG_REQUIRE(first >= 0 && w >= 1 && w <= 255 && h >= 1 && h <= 255 && stride >= w && stride <= 65535 && w * h == count[a] &&
                           (long)first + (long)(h - 1) * stride + w <= (long)s->N,
                       FG_ERR_INVALID_ARG, "fg_mb_set_multilevel: aggregate is not the declared rectangle");
            for (int dy = 0; dy < h; ++dy)
                for (int dx = 0; dx < w; ++dx)
                    FG_REQUIRE(a4[first + dy * stride + dx] == a, FG_ERR_INVALID_ARG, "fg_mb_set_multilevel: aggregate is not the declared rectangle");
            rect[a] = make_uint2((unsigned)first, (unsigned)w | ((unsigned)h << 8) | ((unsigned)stride << 16));
        }
        std::vector<int> nch(n8, 0);
        for (int a = 0; a < n4; ++a) {
            const int p = p4[a], k = nch[p]++;
            FG_REQUIRE(k < 4, FG_ERR_INVALID_ARG, "fg_mb_set_multilevel: an 8 x 8 aggregate has more than four children");
            pos4[a] = 4u * (unsigned)p + (unsigned)k;
            unsigned* words = &child[p].x;
            unsigned& wref = words[k >> 1];
            wref = (k & 1) ? ((wref & 0x0000ffffu) | ((unsigned)a << 16)) : ((wref & 0xffff0000u) | (unsigned)a);
        }
    }
    if (!s->ml_a4 || n4 > s->ml_cap4 || n8 > s->ml_cap8) {   // (a second, larger table set leaves the first to fg_mb_destroy)
        const int c4 = n4 > OC_N4 ? n4 : OC_N4, c8 = n8 > OC_N8 ? n8 : OC_N8;
        if (int rc = mb_alloc(s, &s->ml_rect4, (size_t)c4)) return rc;
        if (int rc = mb_alloc(s, &s->ml_child8, (size_t)c8)) return rc;
        if (!s->ml_p8c) if (int rc = mb_alloc(s, &s->ml_p8c, (size_t)s->N)) return rc;
        if (!s->ml_a4) if (int rc = mb_alloc(s, &s->ml_a4, (size_t)s->N)) return rc;
        if (int rc = mb_alloc(s, &s->ml_parent4, (size_t)c4)) return rc;
        if (int rc = mb_alloc(s, &s->ml_d4g, (size_t)c4)) return rc;
        if (int rc = mb_alloc(s, &s->ml_aci8, (size_t)c8 * ((c8 + 3) & ~3))) return rc;   // rows padded to a multiple of four
        if (int rc = mb_alloc(s, &s->ml_r4, (size_t)s->B * c4)) return rc;
        if (int rc = mb_alloc(s, &s->ml_z8, (size_t)s->B * c8)) return rc;
        if (int rc = mb_alloc(s, &s->ml_r4c, (size_t)s->B * 4 * c8)) return rc;
        if (int rc = mb_alloc(s, &s->ml_pos4, (size_t)c4)) return rc;
        if (!s->ml_scale) {
            if (int rc = mb_alloc(s, &s->ml_scale, (size_t)s->B)) return rc;
            if (int rc = mb_alloc(s, &s->ml_mp, (size_t)s->B * s->N)) return rc;
            if (int rc = mb_alloc(s, &s->ml_ms, (size_t)s->B * s->N)) return rc;
        }
        s->ml_cap4 = c4; s->ml_cap8 = c8;
    }
    FG_HIP_CHECK(hipMemcpy(s->ml_a4, a4.data(), sizeof(uint16_t) * s->N, hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->ml_parent4, p4.data(), sizeof(uint16_t) * n4, hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->ml_rect4, rect.data(), sizeof(uint2) * n4, hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->ml_child8, child.data(), sizeof(uint2) * n8, hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->ml_pos4, pos4.data(), sizeof(uint32_t) * n4, hipMemcpyHostToDevice));
    {
        std::vector<uint16_t> p8c(s->N);
        for (int i = 0; i < s->N; ++i) p8c[i] = p4[a4[i]];
        FG_HIP_CHECK(hipMemcpy(s->ml_p8c, p8c.data(), sizeof(uint16_t) * s->N, hipMemcpyHostToDevice));
    }
    FG_HIP_CHECK(hipMemset(s->ml_r4c, 0, sizeof(mb_real) * (size_t)s->B * 4 * s->ml_cap8));   // the slots of absent children are never written
    std::vector<mb_real> rd4(n4);
    for (int a = 0; a < n4; ++a) { FG_REQUIRE(d4g_host[a] != 0.f, FG_ERR_INVALID_ARG, "fg_mb_set_multilevel: zero Galerkin diagonal"); rd4[a] = 1.f / d4g_host[a]; }
    const int ld = (n8 + 3) & ~3;
    std::vector<mb_real> padded((size_t)n8 * ld, 0.f);
    for (int r = 0; r < n8; ++r)
        for (int c = 0; c < n8; ++c) padded[(size_t)r * ld + c] = aci8_host[(size_t)r * n8 + c];
    FG_HIP_CHECK(hipMemcpy(s->ml_d4g, rd4.data(), sizeof(mb_real) * n4, hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->ml_aci8, padded.data(), sizeof(mb_real) * padded.size(), hipMemcpyHostToDevice));
    s->ml_n4 = n4; s->ml_n8 = n8; s->ml_geom_diag_sum = geom_diag_sum; s->ml_on = enable != 0;
    // ---- aggregate-owned layout of the on-chip CG (fg_mb.h): thread 4 A + c owns child c of 8 x 8 aggregate A
    s->oc_agg = false;
    bool fits = s->d == 2 && s->nbr16 != nullptr && 4 * n8 <= 1024 && s->N <= fg_mb_state::OC_SLOTS;
    for (int a = 0; a < n4 && fits; ++a) fits = rect4_host[4 * a + 1] * rect4_host[4 * a + 2] <= 16;
    if (fits) {
        constexpr int S = fg_mb_state::OC_SLOTS;
        std::vector<int32_t> slot_cell(S, -1), cnt(1024, 0);
        std::vector<uint16_t> cell_slot(s->N, 0xffff);
        std::vector<mb_real> d4t(1024, 0.f);
        for (int A = 0; A < n8; ++A) {
            const unsigned words[2] = {child[A].x, child[A].y};
            for (int c = 0; c < 4; ++c) {
                const unsigned a = (words[c >> 1] >> (16 * (c & 1))) & 0xffffu;
                if (a == 0xffffu) continue;
                const int t = 4 * A + c;
                const int first = rect4_host[4 * a], w = rect4_host[4 * a + 1], h = rect4_host[4 * a + 2], stride = rect4_host[4 * a + 3];
                cnt[t] = w * h;
                d4t[t] = rd4[a];
                for (int dy = 0; dy < h; ++dy)
                    for (int dx = 0; dx < w; ++dx) {
                        const int cell = first + dy * stride + dx, slot = t + 1024 * (dy * w + dx);
                        slot_cell[slot] = cell;
                        cell_slot[cell] = (uint16_t)slot;
                    }
            }
        }
        for (int i = 0; i < s->N && fits; ++i) fits = cell_slot[i] != 0xffff;   // every cell owned (0xFFFF is no slot: 16383 is the last)
        if (fits) {
            // neighbours as BYTE offsets into the LDS vector (slot * 4); a prescribed face -- and every face of a hole -- points at
            // the slot itself (its coefficient is zero: k_mb_pmatrix)
            std::vector<uint2> nbr(S);
            for (int sl = 0; sl < S; ++sl) {
                const int cell = slot_cell[sl];
                unsigned v[4];
                for (int f = 0; f < 4; ++f) {
                    const int32_t n = cell >= 0 ? s->h_nbr[(size_t)f * s->N + cell] : -1;
                    v[f] = 4u * (unsigned)(n >= 0 ? cell_slot[n] : sl);
                }
                nbr[sl] = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
            }
            if (!s->oc_slot_cell) {
                if (int rc = mb_alloc(s, &s->oc_slot_cell, (size_t)S)) return rc;
                if (int rc = mb_alloc(s, &s->oc_cell_slot, (size_t)s->N)) return rc;
                if (int rc = mb_alloc(s, &s->oc_nbr, (size_t)S)) return rc;
                if (int rc = mb_alloc(s, &s->oc_d4g, (size_t)1024)) return rc;
                if (int rc = mb_alloc(s, &s->oc_cnt, (size_t)1024)) return rc;
                if (int rc = mb_alloc(s, &s->Poff4s, (size_t)s->B * S * 4)) return rc;
                if (int rc = mb_alloc(s, &s->Pdiag_s, (size_t)s->B * S)) return rc;
                if (int rc = mb_alloc(s, &s->oc_bestx, (size_t)s->B * S)) return rc;
            }
            // holes stay zero for good: k_mb_pmatrix and the solver write the slots of cells only
            FG_HIP_CHECK(hipMemset(s->Poff4s, 0, sizeof(mb_real) * (size_t)s->B * S * 4));
            FG_HIP_CHECK(hipMemset(s->Pdiag_s, 0, sizeof(mb_real) * (size_t)s->B * S));
            FG_HIP_CHECK(hipMemset(s->oc_bestx, 0, sizeof(mb_real) * (size_t)s->B * S));
            FG_HIP_CHECK(hipMemcpy(s->oc_slot_cell, slot_cell.data(), sizeof(int32_t) * S, hipMemcpyHostToDevice));
            FG_HIP_CHECK(hipMemcpy(s->oc_cell_slot, cell_slot.data(), sizeof(uint16_t) * s->N, hipMemcpyHostToDevice));
            FG_HIP_CHECK(hipMemcpy(s->oc_nbr, nbr.data(), sizeof(uint2) * S, hipMemcpyHostToDevice));
            FG_HIP_CHECK(hipMemcpy(s->oc_d4g, d4t.data(), sizeof(mb_real) * 1024, hipMemcpyHostToDevice));
            FG_HIP_CHECK(hipMemcpy(s->oc_cnt, cnt.data(), sizeof(int32_t) * 1024, hipMemcpyHostToDevice));
            s->oc_agg = true;
            s->oc_matrix_stale = true;   // the slot-ordered copy of the matrix does not exist yet (fg_mb_step.hip: mb_cg_onchip)
        }
    }
    // ---- cluster layout of the cluster CG (fg_mb_cluster.hip)
    return mb_cluster_build(s, n4, n8, rect4_host, p4.data(), child.data(), rd4.data(), padded.data());
}

// cycle counts per phase of the on-chip CG (workgroup 0 of the last launch; FG_MB_OC_VARIANT=256): [0..10] phases, [11] iterations
extern "C" int fg_mb_debug_cycles(fg_mb_handle s, uint64_t* out12) {
    FG_REQUIRE(s && s->oc_dbg && out12, FG_ERR_INVALID_ARG, "fg_mb_debug_cycles: not available");
    FG_HIP_CHECK(hipMemcpy(out12, s->oc_dbg, sizeof(uint64_t) * 12, hipMemcpyDeviceToHost));
    return FG_OK;
}

// [0] current back-off of the pressure BiCGStab's multilevel trial (0 = no tables installed; 4 = every attempt converges), [1] attempts,
// [2] attempts that did not converge and were repeated with the plain recurrence
extern "C" int fg_mb_multilevel_status(fg_mb_handle s, int32_t* out3) {
    FG_REQUIRE(s && out3, FG_ERR_INVALID_ARG, "fg_mb_multilevel_status: bad argument");
    out3[0] = (s->ml_on && s->ml_a4 != nullptr) ? s->ml_bicg_backoff : 0;
    out3[1] = s->ml_bicg_attempts;
    out3[2] = s->ml_bicg_failures;
    return FG_OK;
}

// z = U^-1 L^-1 r with ILU(0) of the velocity matrix currently assembled (the last step's), d systems per env: unit entry of the
// preconditioned rung's preconditioner (tests/test_gpu_mb.py), not on any step path.
extern "C" int fg_mb_debug_ilu_apply(fg_mb_handle s, const mb_real* r_BdN, mb_real* z_BdN, void* stream) {
    FG_REQUIRE(s && s->finalized && !s->host_only && r_BdN && z_BdN, FG_ERR_INVALID_ARG, "fg_mb_debug_ilu_apply: bad argument");
    FG_REQUIRE(!FG_MB_F64, FG_ERR_UNSUPPORTED, "fg_mb_debug_ilu_apply: not part of the fp64 build");
    FG_REQUIRE(mb_ilu_prepare(s), FG_ERR_UNSUPPORTED, "fg_mb_debug_ilu_apply: the mesh does not qualify (a cell with the same neighbour across two faces)");
    hipStream_t st = (hipStream_t)stream;
    MbSolve q = mb_solve_ptrs(s, s->Cdiag, s->Coff, nullptr, nullptr, s->d, 0.f);
    FG_HIP_CHECK(hipMemsetAsync(s->flags, 0, sizeof(int32_t) * s->B * s->d, st));
    mb_ilu_factor(s, nullptr, s->Cdiag, s->Coff, st);
    mb_ilu_apply(s, q, r_BdN, z_BdN, st);
    FG_HIP_CHECK(hipStreamSynchronize(st));
    return FG_OK;
}

// z = M r with the kernel form of the multilevel preconditioner, for every env, on the pressure matrix currently assembled
// (fg_mb_unit_pressure_matrix or the last step): the unit test of mb_ml_apply (tests/test_gpu_mb.py), not on any step path.
extern "C" int fg_mb_multilevel_apply(fg_mb_handle s, const mb_real* r_BN, mb_real* z_BN, void* stream) {
    FG_REQUIRE(s && s->finalized && !s->host_only && r_BN && z_BN, FG_ERR_INVALID_ARG, "fg_mb_multilevel_apply: bad argument");
    FG_REQUIRE(s->ml_a4 != nullptr && s->ml_mp != nullptr, FG_ERR_UNSUPPORTED, "fg_mb_multilevel_apply: no tables installed (fg_mb_set_multilevel)");
    hipStream_t st = (hipStream_t)stream;
    MbSolve q = mb_solve_ptrs(s, s->Pdiag, s->Poff, nullptr, nullptr, 1, 0.f);
    FG_HIP_CHECK(hipMemsetAsync(s->flags, 0, sizeof(int32_t) * s->B, st));
    mb_ml_scale(s, s->Pdiag, st);
    mb_ml_apply(s, q, r_BN, z_BN, st);
    FG_HIP_CHECK(hipStreamSynchronize(st));
    return FG_OK;
}

// Stress harness of the velocity BiCGStab (profiles/bicg_stress.py): solves the systems currently held in the assembly buffers
// (diagonal, off-diagonals, right-hand side; FG_MB_BUF_A / _C_OFF / _RHS) `reps` times from zero, exactly as fg_mb_piso_step's first
// attempt does, and counts the outcomes: [0] solves, [1] with a non-finite system, [2] unconverged, [3] max iterations seen.
extern "C" int fg_mb_debug_bicgstab(fg_mb_handle s, mb_real tol, int32_t max_iterations, int32_t reps, int64_t* out4, double* acc_out,
                                    mb_real* sc_out, void* stream) {
    FG_REQUIRE(s && s->finalized && !s->host_only && out4 && reps > 0, FG_ERR_INVALID_ARG, "fg_mb_debug_bicgstab: bad argument");
    hipStream_t st = (hipStream_t)stream;
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    for (int r = 0; r < reps; ++r) {
        int m = 0;
        const int rc = mb_bicgstab(s, nullptr, s->Cdiag, s->Coff, s->rhs, s->ures, s->d, tol, max_iterations, 0, &m, st);
        out4[0] += 1;
        if (rc == FG_ERR_NOT_FINITE) { out4[1] += 1; if (out4[1] > 20) break; }
        else if (rc == FG_ERR_NOT_CONVERGED) out4[2] += 1;
        else if (rc != FG_OK) return rc;
        if (m > out4[3]) out4[3] = m;
    }
    // the recurrence words as the last solve left them ([B d][12] accumulators, [B d][2] alpha / omega): with max_iterations = k
    // for k = 1, 2, ... this is the history of a (deterministic) solve
    if (acc_out) {
        std::vector<FgDacc> raw((size_t)MB_ACC * s->B * s->d);
        FG_HIP_CHECK(hipMemcpy(raw.data(), s->acc, sizeof(FgDacc) * raw.size(), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < raw.size(); ++k) acc_out[k] = fg_dacc_host_value(raw[k]);
    }
    if (sc_out) FG_HIP_CHECK(hipMemcpy(sc_out, s->sc, sizeof(mb_real) * 2 * s->B * s->d, hipMemcpyDeviceToHost));
    return FG_OK;
}

extern "C" int fg_mb_solver_hints(fg_mb_handle s, int32_t* hints48, int32_t set) {
    FG_REQUIRE(s != nullptr && hints48 != nullptr, FG_ERR_INVALID_ARG, "fg_mb_solver_hints: bad argument");
    int* trial[4] = {&s->ml_bicg_attempts, &s->ml_bicg_failures, &s->ml_bicg_skip, &s->ml_bicg_backoff};
    for (int k = 0; k < 32; ++k) { if (set) s->pred_bicg[k] = hints48[k]; else hints48[k] = s->pred_bicg[k]; }
    for (int k = 0; k < 4; ++k) { if (set) *trial[k] = hints48[32 + k]; else hints48[32 + k] = *trial[k]; }
    // the velocity sweeps' back-off (mb_jacobi): whether a solve goes through the sweeps or straight to BiCGStab
    for (int k = 0; k < 4; ++k) {
        if (set) { s->jac_skip[k] = hints48[36 + k]; s->jac_fails[k] = hints48[40 + k]; s->jac_sweeps[k] = hints48[44 + k]; }
        else { hints48[36 + k] = s->jac_skip[k]; hints48[40 + k] = s->jac_fails[k]; hints48[44 + k] = s->jac_sweeps[k]; }
    }
    return FG_OK;
}

extern "C" int fg_mb_ladder(fg_mb_handle s, int64_t* out4, int32_t force_mask) {
    FG_REQUIRE(s != nullptr, FG_ERR_INVALID_ARG, "fg_mb_ladder: null handle");
    if (out4) for (int k = 0; k < 4; ++k) out4[k] = s->ladder[k];
    s->ladder_force = force_mask;
    return FG_OK;
}

extern "C" int fg_mb_env_status(fg_mb_handle s, int32_t* out_B_host) {
    FG_REQUIRE(s && s->finalized && out_B_host, FG_ERR_INVALID_ARG, "fg_mb_env_status: bad argument");
    for (int b = 0; b < s->B; ++b) out_B_host[b] = s->env_status[b];
    return FG_OK;
}



// the CFL maxima into red_pinned, enqueued without a wait (the caller synchronises: fg_mb_single_step reads the flux-balance guard
// of the step and the first maxima behind ONE wait)
static int mb_enqueue_max_velocity(fg_mb_state* s, hipStream_t st) {
    const int cells = std::max(s->N, s->NB);
    FG_HIP_CHECK(hipMemsetAsync(s->red, 0, sizeof(mb_real) * s->B, st));
    MB_DISPATCH(s, hipLaunchKernelGGL(k_mb_maxvel<DIMS>, dim3((cells + FG_BLOCK - 1) / FG_BLOCK, s->B), dim3(FG_BLOCK), 0, st,
                                      s->dev, s->velocity, s->bvel, s->red););
    FG_HIP_CHECK(hipMemcpyAsync(s->red_pinned, s->red, sizeof(mb_real) * s->B, hipMemcpyDeviceToHost, st));
    return FG_OK;
}

extern "C" int fg_mb_max_velocity(fg_mb_handle s, mb_real* out_B_host, void* stream) {
    FG_REQUIRE(s && s->finalized && s->velocity && out_B_host, FG_ERR_NOT_BOUND, "fg_mb_max_velocity: fields not bound");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = mb_enqueue_max_velocity(s, st)) return rc;
    FG_HIP_CHECK(hipStreamSynchronize(st));
    for (int b = 0; b < s->B; ++b) out_B_host[b] = s->red_pinned[b];
    return FG_OK;
}

static int mb_outflow_pre(fg_mb_state* s, const mb_real* dt_dev, int slot0, int count, int slot0b, int countb, const mb_real* velm,
                          mb_real tol, hipStream_t st) {
    const int r0[2] = {slot0, slot0b}, rn[2] = {count, countb};
    MB_DISPATCH(s, {
        for (int k = 0; k < 2; ++k)
            if (rn[k] > 0)
                hipLaunchKernelGGL(k_mb_outflow<DIMS>, dim3((rn[k] + 63) / 64, s->B), dim3(64), 0, st, s->dev, dt_dev, s->velocity, s->bvel,
                                   r0[k], rn[k], velm[0], velm[1], velm[2]);
        hipLaunchKernelGGL(k_mb_bflux<DIMS>, dim3(s->B), dim3(FG_BLOCK), 0, st, s->dev, s->bvel, slot0, count, slot0b, countb, s->red2);
        for (int k = 0; k < 2; ++k)
            if (rn[k] > 0)
                hipLaunchKernelGGL(k_mb_balance<DIMS>, dim3((rn[k] + 63) / 64, s->B), dim3(64), 0, st, s->dev, dt_dev, s->red2, 0.01f * tol,
                                   s->bvel, r0[k], rn[k]);
    });
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

// update_advective_boundaries + balance_boundary_fluxes for one FIXED face with the same dt for every env (the PRE hook as
// make_divergence_free runs it, with time_step = 1: PISOtorch_simulation.py:1334-1345)
extern "C" int fg_mb_update_advective_boundary(fg_mb_handle s, mb_real dt, int32_t slot0, int32_t count, int32_t slot0_b, int32_t count_b,
                                               const mb_real* velm, mb_real tol, void* stream) {
    FG_REQUIRE(s && s->finalized && s->velocity && velm, FG_ERR_NOT_BOUND, "fg_mb_update_advective_boundary: fields not bound");
    FG_REQUIRE(slot0 >= 0 && count > 0 && slot0 + count <= s->NB && count_b >= 0 && (count_b == 0 || (slot0_b >= 0 && slot0_b + count_b <= s->NB)),
               FG_ERR_INVALID_ARG, "fg_mb_update_advective_boundary: slots out of range");
    hipStream_t st = (hipStream_t)stream;
    for (int b = 0; b < s->B; ++b) s->dt_pinned[b] = dt;
    FG_HIP_CHECK(hipMemcpyAsync(s->dt_dev, s->dt_pinned, sizeof(mb_real) * s->B, hipMemcpyHostToDevice, st));
    if (int rc = mb_outflow_pre(s, s->dt_dev, slot0, count, slot0_b, count_b, velm, tol, st)) return rc;
    FG_HIP_CHECK(hipStreamSynchronize(st));
    return FG_OK;
}

static bool mb_close_zero(double v) { return std::fabs(v) <= 1e-8; }  // np.isclose(v, 0) defaults

extern "C" int fg_mb_boundary_flux_balance(fg_mb_handle s, mb_real* out_B_host, void* stream) {
    FG_REQUIRE(s && s->finalized && s->velocity && out_B_host, FG_ERR_NOT_BOUND, "fg_mb_boundary_flux_balance: fields not bound");
    hipStream_t st = (hipStream_t)stream;
    MB_DISPATCH(s, hipLaunchKernelGGL(k_mb_bflux<DIMS>, dim3(s->B), dim3(FG_BLOCK), 0, st, s->dev, s->bvel, 0, 0, 0, 0, s->red2););
    FG_HIP_CHECK(hipMemcpyAsync(s->red2_pinned, s->red2, sizeof(mb_real) * 2 * s->B, hipMemcpyDeviceToHost, st));
    FG_HIP_CHECK(hipStreamSynchronize(st));
    for (int b = 0; b < s->B; ++b) out_B_host[b] = s->red2_pinned[2 * b] + s->red2_pinned[2 * b + 1];
    return FG_OK;
}

extern "C" int fg_mb_single_step(fg_mb_handle s, const fg_mb_sim_options* o, int32_t* out, mb_real* flux_host, void* stream) {
    FG_REQUIRE(s && s->finalized && s->velocity, FG_ERR_NOT_BOUND, "fg_mb_single_step: fields not bound");
    FG_REQUIRE(o && out, FG_ERR_INVALID_ARG, "fg_mb_single_step: null argument");
    FG_REQUIRE((o->outflow_count == 0 || (o->outflow_slot0 >= 0 && o->outflow_slot0 + o->outflow_count <= s->NB)) &&
                   (o->outflow_count_b == 0 || (o->outflow_count > 0 && o->outflow_slot0_b >= 0 && o->outflow_slot0_b + o->outflow_count_b <= s->NB)),
               FG_ERR_INVALID_ARG,
               "fg_mb_single_step: outflow slots out of range");
    hipStream_t st = (hipStream_t)stream;
    const int B = s->B;
    // flux-balance guard (simulation.py:221-229); with adaptive sub-steps the CFL maxima of the first sub-step are enqueued behind it
    // and both come back behind ONE wait
    bool maxvel_ready = false;
    {
        std::vector<mb_real> fl(B);
        // (one workgroup per env writes its two sums: straight into the host-pinned words, no device-to-host copy)
        MB_DISPATCH(s, hipLaunchKernelGGL(k_mb_bflux<DIMS>, dim3(s->B), dim3(FG_BLOCK), 0, st, s->dev, s->bvel, 0, 0, 0, 0, s->red2_pinned););
        if (o->adaptive) { if (int rc = mb_enqueue_max_velocity(s, st)) return rc; maxvel_ready = true; }
        FG_HIP_CHECK(hipStreamSynchronize(st));
        for (int b = 0; b < B; ++b) fl[b] = s->red2_pinned[2 * b] + s->red2_pinned[2 * b + 1];
        mb_real worst = 0.f;
        for (int b = 0; b < B; ++b) {
            if (flux_host) flux_host[b] = fl[b];
            const mb_real a = std::fabs(fl[b]);
            worst = (a > worst || a != a) ? a : worst;
        }
        if (!(worst <= o->flux_balance_tol)) {
            fg_set_error("Domain boundary fluxes not balanced, cannot proceed with simulation step.");
            return FG_ERR_FLUX_BALANCE;
        }
    }
    std::vector<double> t_rem(B, (double)o->time_step);
    std::vector<mb_real> mv(B, 0.f);
    std::vector<int32_t> status(B, 0);
    int32_t stats[4] = {-1, -1, -1, -1};
    int substeps = 0, all_ok = 1;
    int fixed_left = o->adaptive ? 0 : (o->substeps > 0 ? o->substeps : 1);
    for (;;) {
        bool any = false;
        if (o->adaptive) for (int b = 0; b < B; ++b) any = any || (t_rem[b] > 0 && !mb_close_zero(t_rem[b]));
        else any = fixed_left > 0;
        if (!any) break;
        if (o->adaptive) {
            if (maxvel_ready) { for (int b = 0; b < B; ++b) mv[b] = s->red_pinned[b]; maxvel_ready = false; }   // (read with the guard above)
            else if (int rc = fg_mb_max_velocity(s, mv.data(), stream)) return rc;
        }
        // _PISO_adaptive_step (PISOtorch_simulation.py:2004-2064): ts = t_rem / ceil(t_rem / (CFL / max_vel)), per env
        s->dt_slot ^= 1;
        mb_real* dt_slot = s->dt_pinned + (size_t)s->dt_slot * B;     // two halves used in turn (see the end of the loop)
        for (int b = 0; b < B; ++b) {
            mb_real ts = 0.f;
            if (status[b] == 2) {
                ts = 0.f;   // dropped out of this step (non-finite solve or state)
            } else if (!o->adaptive) {
                ts = o->time_step / (mb_real)(o->substeps > 0 ? o->substeps : 1);
            } else if (t_rem[b] > 0 && !mb_close_zero(t_rem[b]) && !std::isfinite(mv[b])) {
                status[b] = 2; t_rem[b] = 0.0;   // a state that is already non-finite: nothing to step
            } else if (t_rem[b] > 0 && !mb_close_zero(t_rem[b])) {
                const double max_ts = mb_close_zero(mv[b]) ? t_rem[b] : (double)o->cfl / (double)mv[b];
                const double tsd = (max_ts >= t_rem[b]) ? t_rem[b] : t_rem[b] / (double)(long long)std::ceil(t_rem[b] / max_ts);
                t_rem[b] -= tsd;
                ts = (mb_real)tsd;
            }
            dt_slot[b] = ts;
        }
        FG_HIP_CHECK(hipMemcpyAsync(s->dt_step, dt_slot, sizeof(mb_real) * B, hipMemcpyHostToDevice, st));   // (the step's working copy)
        if (o->outflow_count > 0)  // PRE hook of the cylinder / airfoil envs (cylinder_env_base.py:280-300)
            if (int rc = mb_outflow_pre(s, s->dt_step, o->outflow_slot0, o->outflow_count, o->outflow_slot0_b, o->outflow_count_b,
                                        o->outflow_velm, o->outflow_tol, st))
                return rc;
        const int rc = fg_mb_piso_step(s, s->dt_step, &o->step, stats, stream);
        if (rc == FG_ERR_NOT_CONVERGED || rc == FG_ERR_NOT_FINITE) {
            all_ok = 0;
            for (int b = 0; b < B; ++b) {
                status[b] = std::max(status[b], s->env_status[b]);
                // a non-finite solve: that env's substep was not committed; it sits out the rest of this step with its
                // state intact (Simulation.single_step returns False there, simulation.py:259-280), the others finish
                if (s->env_status[b] == 2) t_rem[b] = 0.0;
            }
        } else if (rc != FG_OK) return rc;
        // (no wait here: the time steps of the next sub-step go into the OTHER half of dt_pinned, and the wait for its CFL maxima
        //  -- or for the next step's guard -- comes before this half is written again)
        ++substeps;
        if (!o->adaptive) --fixed_left;
        if (substeps >= (o->max_substeps > 0 ? o->max_substeps : 100000)) break;
    }
    for (int i = 0; i < 4; ++i) out[i] = stats[i];
    out[4] = substeps;
    out[5] = all_ok;
    s->env_status = status;
    return FG_OK;
}

// Simulation.make_divergence_free (PISOtorch_simulation.py:1318-1429): one projection with A = 1, dt = 1, h = u
extern "C" int fg_mb_make_divergence_free(fg_mb_handle s, const fg_mb_step_options* opt, void* stream) {
    FG_REQUIRE(s && s->finalized && s->velocity && opt, FG_ERR_NOT_BOUND, "fg_mb_make_divergence_free: fields not bound");
    hipStream_t st = (hipStream_t)stream;
    const int B = s->B, N = s->N, d = s->d, NB = s->NB;
    const int cells = std::max(N, NB);
    const dim3 blk(FG_BLOCK), gc((cells + FG_BLOCK - 1) / FG_BLOCK, B), gn((N + FG_BLOCK - 1) / FG_BLOCK, B);
    const size_t vel_env = (size_t)d * N, BN = (size_t)B * N;
    const dim3 gcopy((unsigned)((vel_env + FG_BLOCK - 1) / FG_BLOCK), B);
    const MbDev& D = s->dev;
    int soft_rc = FG_OK;
    hipLaunchKernelGGL(k_mb_fill, dim3((unsigned)((BN + FG_BLOCK - 1) / FG_BLOCK)), blk, 0, st, BN, 1.f, s->rA);
    hipLaunchKernelGGL(k_mb_copy, gcopy, blk, 0, st, vel_env, (const mb_real*)nullptr, s->velocity, s->hvec);
    MB_DISPATCH(s, {
        hipLaunchKernelGGL(k_mb_contra<DIMS>, gc, blk, 0, st, D, (const mb_real*)nullptr, s->hvec, s->bvel, s->cc, s->fb);
        hipLaunchKernelGGL(k_mb_pmatrix<DIMS>, gn, blk, 0, st, D, (const mb_real*)nullptr, s->rA, s->Pdiag, s->Poff, s->Poff4, SLOTS.cell_slot, SLOTS.off4, SLOTS.diag, SLOTS.stride); mb_slots_written(s);
        for (int ps = 0; ps < opt->pressure_non_ortho_steps; ++ps) {
            hipLaunchKernelGGL(k_mb_div<DIMS>, gn, blk, 0, st, D, (const mb_real*)nullptr, s->cc, s->fb, s->rA, s->pressure, 1, s->div);
            int m = 0;
            const int prc = opt->pressure_use_bicgstab
                                ? mb_pressure_bicgstab(s, nullptr, opt->pressure_tol, opt->max_iterations, ps > 0, &m, st, opt->pressure_project_mean,
                                                       opt->pressure_use_bicgstab == 2, 12 + (ps & 3))
                                : mb_cg(s, nullptr, s->Pdiag, s->Poff, s->div, s->pres, opt->pressure_tol, opt->max_iterations, ps > 0,
                                        opt->pressure_project_mean, opt->pressure_stall_accept, &m, st);
            if (prc == FG_ERR_NOT_CONVERGED || prc == FG_ERR_NOT_FINITE) soft_rc = prc;
            else if (prc != FG_OK) return prc;
            hipLaunchKernelGGL(k_mb_sum, dim3(MB_SUM_WGS, B), blk, 0, st, N, (const mb_real*)nullptr, s->pres, s->red8);
            hipLaunchKernelGGL(k_mb_sub_mean, gn, blk, 0, st, N, (const mb_real*)nullptr, s->red8, s->pres, s->pressure);
        }
        hipLaunchKernelGGL(k_mb_correct<DIMS>, gn, blk, 0, st, D, (const mb_real*)nullptr, s->rA, s->hvec, s->pressure, s->velocity);
    });
    FG_HIP_CHECK(hipGetLastError());
    return soft_rc;
}

// host copies of the boundary-slot tables: owner cell, face, Minv | det
extern "C" int fg_mb_get_boundary_tables(fg_mb_handle s, int32_t* cell, int32_t* face, mb_real* transform) {
    FG_REQUIRE(s && s->finalized, FG_ERR_INVALID_ARG, "fg_mb_get_boundary_tables: domain not finalized");
    const int tw = s->d * s->d + 1;
    for (int k = 0; k < s->NB; ++k) {
        if (cell) cell[k] = s->h_bcell[k];
        if (face) face[k] = s->h_bface[k];
        if (transform) for (int q = 0; q < tw; ++q) transform[(size_t)k * tw + q] = s->h_Tb[(size_t)k * tw + q];
    }
    return FG_OK;
}
extern "C" int fg_mb_get_cell_transforms(fg_mb_handle s, mb_real* transform /* [N][d*d+1] Minv | det */) {
    FG_REQUIRE(s && s->finalized && transform, FG_ERR_INVALID_ARG, "fg_mb_get_cell_transforms: bad argument");
    std::copy(s->h_T.begin(), s->h_T.end(), transform);
    return FG_OK;
}

// Vector the CG residuals are kept orthogonal to (pressure_project_mean): default the constant; the left near-null vector of
// the pressure matrix removes the residual floor the constant leaves on non-orthogonal meshes.  Host array [N].
extern "C" int fg_mb_set_residual_projection(fg_mb_handle s, const mb_real* y_host) {
    FG_REQUIRE(s && s->finalized, FG_ERR_INVALID_ARG, "fg_mb_set_residual_projection: domain not finalized");
    FG_REQUIRE(!s->host_only, FG_ERR_UNSUPPORTED, "fg_mb_set_residual_projection: host-only handle");
    std::vector<mb_real> y(s->N);
    double nrm = 0.0;
    for (int i = 0; i < s->N; ++i) { y[i] = y_host ? y_host[i] : 1.f; nrm += (double)y[i] * y[i]; }
    FG_REQUIRE(nrm > 0.0 && std::isfinite(nrm), FG_ERR_INVALID_ARG, "fg_mb_set_residual_projection: zero or non-finite vector");
    const mb_real sc = (mb_real)(1.0 / std::sqrt(nrm));
    for (int i = 0; i < s->N; ++i) y[i] *= sc;
    FG_HIP_CHECK(hipMemcpy(s->yproj, y.data(), sizeof(mb_real) * s->N, hipMemcpyHostToDevice));
    s->yproj_const = (y_host == nullptr);
    return FG_OK;
}
// CG solves end with their best iterate once no iterate has improved on it (by 2x, or at all below the acceptance band) for
// this many iterations; default 400.  Meshes whose pressure system has a residual floor above the tolerance
// spend that many iterations per solve for nothing, so their envs lower it.
extern "C" int fg_mb_set_stall_limit(fg_mb_handle s, int32_t iterations) {
    FG_REQUIRE(s != nullptr, FG_ERR_INVALID_ARG, "fg_mb_set_stall_limit: null handle");
    FG_REQUIRE(iterations >= 20, FG_ERR_INVALID_ARG, "fg_mb_set_stall_limit: at least one chunk of 20 iterations");
    s->cg_stall_limit = iterations;
    return FG_OK;
}
extern "C" int fg_mb_set_advection_jacobi(fg_mb_handle s, int on) {
    FG_REQUIRE(s != nullptr, FG_ERR_INVALID_ARG, "fg_mb_set_advection_jacobi: null handle");
    if (!s->adv_jacobi_env) s->adv_jacobi = on ? 1 : 0;      // (FG_ADV_JACOBI in the environment wins: A/B runs of whole envs)
    for (int k = 0; k < 4; ++k) { s->jac_sweeps[k] = 0; s->jac_skip[k] = 0; s->jac_fails[k] = 0; }
    return FG_OK;
}
extern "C" int fg_mb_advection_jacobi_counts(fg_mb_handle s, int64_t* out2) {
    FG_REQUIRE(s != nullptr && out2 != nullptr, FG_ERR_INVALID_ARG, "fg_mb_advection_jacobi_counts: bad argument");
    out2[0] = s->jac_solves; out2[1] = s->jac_fallbacks;
    return FG_OK;
}
extern "C" int fg_mb_set_advection_start(fg_mb_handle s, int from_result) {
    FG_REQUIRE(s != nullptr, FG_ERR_INVALID_ARG, "fg_mb_set_advection_start: null handle");
    s->adv_from_result = from_result ? 1 : 0;
    return FG_OK;
}
// builds the pressure matrix for A = 1 into the P buffers (FG_MB_BUF_P_DIAG / P_OFF): the geometry-only matrix whose left
// near-null vector fg_mb_set_residual_projection wants
extern "C" int fg_mb_unit_pressure_matrix(fg_mb_handle s, void* stream) {
    FG_REQUIRE(s && s->finalized, FG_ERR_INVALID_ARG, "fg_mb_unit_pressure_matrix: domain not finalized");
    FG_REQUIRE(!s->host_only, FG_ERR_UNSUPPORTED, "fg_mb_unit_pressure_matrix: host-only handle");
    hipStream_t st = (hipStream_t)stream;
    const size_t BN = (size_t)s->B * s->N;
    hipLaunchKernelGGL(k_mb_fill, dim3((unsigned)((BN + FG_BLOCK - 1) / FG_BLOCK)), dim3(FG_BLOCK), 0, st, BN, 1.f, s->rA);
    MB_DISPATCH(s, hipLaunchKernelGGL(k_mb_pmatrix<DIMS>, dim3((s->N + FG_BLOCK - 1) / FG_BLOCK, s->B), dim3(FG_BLOCK), 0, st, s->dev,
                                      (const mb_real*)nullptr, s->rA, s->Pdiag, s->Poff, s->Poff4, SLOTS.cell_slot, SLOTS.off4, SLOTS.diag, SLOTS.stride); mb_slots_written(s););
    FG_HIP_CHECK(hipStreamSynchronize(st));
    return FG_OK;
}

extern "C" int fg_mb_profile_enable(fg_mb_handle s, int32_t on) {
    FG_REQUIRE(s && s->finalized, FG_ERR_INVALID_ARG, "fg_mb_profile_enable: domain not finalized");
    FG_REQUIRE(!s->host_only, FG_ERR_UNSUPPORTED, "fg_mb_profile_enable: host-only handle");
    if (on && !s->prof_ev[0]) {
        for (int k = 0; k < 64; ++k) FG_HIP_CHECK(hipEventCreate(&s->prof_ev[k]));
        for (int k = 0; k < 2; ++k) FG_HIP_CHECK(hipEventCreate(&s->prof_ev_oc[k]));
    }
    s->prof_on = on ? 1 : 0;
    s->prof_used = 0; s->prof_chunk = 0; s->prof_its = 0;
    for (int k = 0; k < 3; ++k) { s->prof_ms[k] = 0; s->prof_bytes[k] = 0; s->prof_n[k] = 0; s->prof_launches[k] = 0; }
    return FG_OK;
}
extern "C" const char* fg_mb_profile_kind_name(int32_t kind) {
    return kind == 0 ? "k_mbc_ap" : (kind == 1 ? "k_mbc_update" : (kind == 2 ? "k_mbc_onchip" : nullptr));
}
extern "C" int fg_mb_profile_iterations(fg_mb_handle s, int64_t* iterations) {
    FG_REQUIRE(s && iterations, FG_ERR_INVALID_ARG, "fg_mb_profile_iterations: bad argument");
    *iterations = s->prof_its;
    return FG_OK;
}
extern "C" int fg_mb_profile_read(fg_mb_handle s, int32_t kind, double* ms_sum, int64_t* samples, double* bytes_sum, int64_t* launches) {
    FG_REQUIRE(s && kind >= 0 && kind < 3, FG_ERR_INVALID_ARG, "fg_mb_profile_read: bad argument");
    if (ms_sum) *ms_sum = s->prof_ms[kind];
    if (samples) *samples = s->prof_n[kind];
    if (bytes_sum) *bytes_sum = s->prof_bytes[kind];
    if (launches) *launches = s->prof_launches[kind];
    return FG_OK;
}

// intermediate buffers for the parity tests
extern "C" int fg_mb_get_buffer(fg_mb_handle s, int32_t which, const mb_real** ptr, int64_t* count) {
    FG_REQUIRE(s && s->finalized && ptr && count, FG_ERR_INVALID_ARG, "fg_mb_get_buffer: bad argument");
    FG_REQUIRE(!s->host_only, FG_ERR_UNSUPPORTED, "fg_mb_get_buffer: host-only handle");
    const int64_t B = s->B, N = s->N, d = s->d, F = s->F;
    switch (which) {
        case FG_MB_BUF_A: *ptr = s->Cdiag; *count = B * N; break;
        case FG_MB_BUF_C_OFF: *ptr = s->Coff; *count = B * F * N; break;
        case FG_MB_BUF_RHS: *ptr = s->rhs; *count = B * d * N; break;
        case FG_MB_BUF_H: *ptr = s->hvec; *count = B * d * N; break;
        case FG_MB_BUF_DIV: *ptr = s->div; *count = B * N; break;
        case FG_MB_BUF_P_DIAG: *ptr = s->Pdiag; *count = B * N; break;
        case FG_MB_BUF_P_OFF: *ptr = s->Poff; *count = B * F * N; break;
        case FG_MB_BUF_VELOCITY_RESULT: *ptr = s->ures; *count = B * d * N; break;
        case FG_MB_BUF_KRYLOV0: case FG_MB_BUF_KRYLOV0 + 1: case FG_MB_BUF_KRYLOV0 + 2: case FG_MB_BUF_KRYLOV0 + 3: case FG_MB_BUF_KRYLOV0 + 4:
            *ptr = s->w[which - FG_MB_BUF_KRYLOV0]; *count = B * d * N; break;
        default: fg_set_error("fg_mb_get_buffer: unknown buffer id"); return FG_ERR_INVALID_ARG;
    }
    return FG_OK;
}

extern "C" int fg_mb_read_buffer(fg_mb_handle s, int32_t which, mb_real* dst_device, void* stream) {
    const mb_real* p = nullptr;
    int64_t n = 0;
    if (int rc = fg_mb_get_buffer(s, which, &p, &n)) return rc;
    FG_REQUIRE(dst_device != nullptr, FG_ERR_INVALID_ARG, "fg_mb_read_buffer: null destination");
    FG_HIP_CHECK(hipMemcpyAsync(dst_device, p, sizeof(mb_real) * n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    FG_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return FG_OK;
}

// host copies of the mesh tables (fg_mb.h) for tests: which = FG_MB_TABLE_*; *count receives the number of 4-byte elements,
// out may be NULL to query it
extern "C" int fg_mb_get_host_table(fg_mb_handle s, int32_t which, void* out, int64_t* count) {
    FG_REQUIRE(s && s->finalized && count, FG_ERR_INVALID_ARG, "fg_mb_get_host_table: bad argument");
    const void* src = nullptr;
    size_t n = 0;
#define T_(id, v) case id: src = s->v.data(); n = s->v.size(); break;
    switch (which) {
        T_(0, h_nbr) T_(1, h_fcode) T_(2, h_T) T_(3, h_Tb) T_(4, h_bcell) T_(5, h_bface) T_(6, h_Vdiag) T_(7, h_Voff) T_(8, h_KPp)
        T_(9, h_KPn) T_(10, h_SVc_idx) T_(11, h_SVc_w) T_(12, h_SVb_idx) T_(13, h_SVb_w) T_(14, h_SP_idx) T_(15, h_SP_face)
        T_(16, h_SP_wp) T_(17, h_SP_wn)
        default: fg_set_error("fg_mb_get_host_table: unknown table id"); return FG_ERR_INVALID_ARG;
    }
#undef T_
    *count = (int64_t)n;
    if (out && n) memcpy(out, src, n * 4);
    return FG_OK;
}

// host copy of the neighbour table [2d][N] (global index, or -1 - boundary slot)
extern "C" int fg_mb_get_neighbors(fg_mb_handle s, int32_t* out /*[2d*N]*/) {
    FG_REQUIRE(s && s->finalized && out, FG_ERR_INVALID_ARG, "fg_mb_get_neighbors: bad argument");
    std::copy(s->h_nbr.begin(), s->h_nbr.end(), out);
    return FG_OK;
}
