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

namespace {

constexpr int MB_ACC = 12;  // doubles per system
constexpr int A_RHO = 0, A_RV = 2, A_SS = 3, A_TS = 4, A_TT = 5, A_RR = 6, A_SV = 7, A_ST = 8;  // A_SV, A_ST: sum v, sum t (projection)
// A_RHOE + (it & 1): the rho the recurrence of iteration `it` actually uses -- rw.r, or r.r after a breakdown restart (k_mbb_p)
constexpr int A_RHOE = 10;

#define MB_CELL                                         \
    const int i = blockIdx.x * FG_BLOCK + threadIdx.x;  \
    const int b = blockIdx.y;                           \
    const int N = D.N;                                  \
    const bool valid = i < N;

__device__ __forceinline__ bool mb_active(const mb_real* dt, int b) { return dt == nullptr || dt[b] > 0.f; }

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
                                                          mb_real* __restrict__ Poff4s = nullptr, mb_real* __restrict__ Pdiag_s = nullptr) {
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
    // ... and in the slot order of the aggregate-owned on-chip CG (fg_mb.h: OC_SLOTS), diagonal included
    if (DIMS == 2 && cell_slot) {
        const size_t sl = (size_t)b * fg_mb_state::OC_SLOTS + cell_slot[i];
        *reinterpret_cast<float4*>(Poff4s + sl * 4) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        Pdiag_s[sl] = dv;
    }
}

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
                                                          mb_real* __restrict__ u) {
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
        u[q] = h[q] - ra * gp;
    }
}

__device__ __forceinline__ mb_real mb_block_sum(mb_real v, mb_real* lds) {
    v = fg_wave_sum(v);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    const mb_real r = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return r;
}
// NV sums with ONE barrier pair (lds: NV * 4 floats): in the launch-bound Krylov kernels the reduction tail is a visible
// share of the run time, and four sums one after the other are eight barriers
template <int NV>
__device__ __forceinline__ void mb_block_sums(mb_real (&v)[NV], mb_real* lds) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = fg_wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) lds[k * 4 + (threadIdx.x >> 6)] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = lds[k * 4] + lds[k * 4 + 1] + lds[k * 4 + 2] + lds[k * 4 + 3];
    __syncthreads();
}

// The accumulator adds of a reduction tail with one LANE per value (threads 0 .. NV-1 side by side) instead of thread 0 doing them
// one after the other: the exact split of a value into the FgDacc words and its atomics are a dependent chain of ~100
// instructions, and in launches whose workgroups are all resident at once that per-workgroup tail is exposed (round 4: 11.7 of
// 36.9 us in the single-block k_bicgf_b, profiles/micro_bicg2d.py).
template <int NV>
__device__ __forceinline__ void mb_acc_tail(FgDacc* a, const int (&slot)[NV], const mb_real (&val)[NV], const bool (&on)[NV]) {
    if (threadIdx.x < NV) {
        int sl = slot[0]; mb_real v = val[0]; bool o = on[0];
#pragma unroll
        for (int k = 1; k < NV; ++k)
            if ((int)threadIdx.x == k) { sl = slot[k]; v = val[k]; o = on[k]; }
        if (o) acc_add(a + sl, (double)v);
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
                              mb_real* __restrict__ copy) {
    const int b = blockIdx.y, i = blockIdx.x * FG_BLOCK + threadIdx.x;
    if (i >= N || !mb_active(dt, b)) return;
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

// ---------------------------------------------------------------------------------------------------------------
// Krylov solvers on the ELL matrix (diag [B][N], off [B][F][N], shared neighbour table).  System sys = b * nc + comp
// is blockIdx.y; the scalars of the recurrences live in device accumulators (fp64) exactly as in the single-block
// solvers (fg_bicgstab.hip, fg_poisson.hip), so the host only polls convergence.
// ---------------------------------------------------------------------------------------------------------------
struct MbSolve {
    const mb_real* diag; const mb_real* off; const mb_real* rhs;
    mb_real* x; mb_real* r; mb_real* rw; mb_real* p; mb_real* v; mb_real* t;
    FgDacc* acc; mb_real* sc; int32_t* flags; fg_solve_info* info;
    int nc; mb_real tol;
    // best-iterate tracking of the CG pressure solve (returnBestResult, cg_solver_kernel.cu:345-361): sc[2 sys] holds the
    // residual of the kept iterate, best_it the iteration it belongs to, best_x the iterate itself
    mb_real* best_x; int32_t* best_it; int stall_limit;
    // device-side iteration index of the graph-replayed CG: ctr[0] read by k_mbc_ap*, ctr[1] - 1 by k_mbc_update*
    int32_t* it_ctr; int max_iterations;
    int it_base;  // BiCGStab: iteration index of the last restart (kernels run on the index since then, reports add this)
    // stall acceptance (off when 0): a system whose kept iterate is within accept_factor * tol and has not improved for
    // accept_window iterations ends with that iterate and counts as converged
    mb_real accept_factor; int accept_window;
    // BiCGStab on the singular pressure system: 1 = iterate on Q P with Q = I - 1 1^T / N (all vectors mean-free), which removes
    // the null space the plain recurrence breaks down on
    int project;
    // right preconditioning (multilevel, mb_ml_apply): when set, v = A mp with mp = M p, t = A ms with ms = M s, and the iterate
    // advances along mp / ms; the recurrence itself (p, s, r and all dot products) is the one of A M
    const mb_real* mp; const mb_real* ms;
    // fused s / t kernel (k_mbb_st*): s lives in its own buffer (the neighbours' s is recomputed from r and v, which must still be
    // there), k_mbb_x reads it from here and takes over the convergence-on-s decision; null = the separate s and t kernels
    mb_real* sbuf;
    // fused p / v kernel (k_mbb_pv*): p and v of the previous iteration (the neighbours' new p is recomputed from them, so the new p
    // and v go to the other buffer of a pair); q.p / q.v are the current ones.  Null = the separate p and v kernels
    const mb_real* p_prev; const mb_real* v_prev;
};

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
    const int sys = blockIdx.y;                         \
    const int b = sys / q.nc;                           \
    const int N = D.N;                                  \
    const bool valid = i < N;                           \
    const bool leader = (blockIdx.x == 0 && threadIdx.x == 0); \
    const size_t vb = (size_t)sys * N;                  \
    FgDacc* a = q.acc + (size_t)sys * MB_ACC;           \
    __shared__ mb_real lds[16];                           \
    (void)b; (void)leader; (void)a; (void)lds; (void)valid;

// accumulator / scalar / flag words: only through acc_ld / acc_st, sc_ld / sc_st, flag_ld / flag_st (fg_internal.h)
__device__ __forceinline__ mb_real mb_rms(double rr, int n) { return (mb_real)sqrt(rr / (double)n); }
// ok_flag: what a finite verdict stores (1 done; 4 = converged on s, the x kernel still owes x += alpha p) -- ONE store of the
// flag, after the info words (other workgroups of the env read it in the same launch)
__device__ __forceinline__ void mb_mark(const MbSolve& q, int sys, mb_real crit, int it, int ok_flag = 1) {
    const bool finite = isfinite(crit);
    q.info[sys].final_residual = crit;
    q.info[sys].used_iterations = it;
    q.info[sys].converged = finite ? 1 : 0;
    q.info[sys].is_finite = finite ? 1 : 0;
    flag_st(q.flags + (sys), finite ? ok_flag : 2);
}

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
    const int sys = blockIdx.y;                                   \
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
__global__ __launch_bounds__(FG_BLOCK) void k_ml_restrict(MlDev M, const mb_real* __restrict__ in, int N, const int32_t* __restrict__ flags) {
    const int tq = blockIdx.x * FG_BLOCK + threadIdx.x, a = tq >> 2, row = tq & 3, sys = blockIdx.y;
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
    const int tq = blockIdx.x * FG_BLOCK + threadIdx.x, ag = tq >> 2, row = tq & 3, sys = blockIdx.y, N = D.N;
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
    const int tq = blockIdx.x * FG_BLOCK + threadIdx.x, ag = tq >> 2, row = tq & 3, sys = blockIdx.y, N = D.N;
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
constexpr int ML_N8_MAX = 2048, ML_ROWS = 16, ML_CG = 64;   // coarse solve: rows per workgroup, column groups
// One workgroup = 16 rows of A8^+ x SB systems; its 1024 threads are 16 rows x 64 column groups (a thread streams 1 / 64 of its
// row -- by columns, the matrix is symmetric --: 12 loads at Airfoil2D's 771 aggregates, all in flight at once), partial sums meet
// in LDS.  49 x 4 workgroups at 771 aggregates x 16 envs; with 64 rows per workgroup (13 x 4 workgroups, 49 dependent-latency
// loads per thread) the kernel took 12.3 us and was the largest single item of the preconditioned airfoil step.
// SB systems per workgroup (template): the matrix rows a workgroup streams serve SB right-hand sides, so 8 halve the L2 traffic of
// 4 once there are enough systems to fill the chip either way (mb_ml_apply picks).  LDS is sized to the mesh (n8p = n8 rounded up):
// r8 [SB][n8p] (re-used by the second folding stage, which needs 4 SB ROWS floats) and the partial sums [CG][SB][ROWS].
template <int SB>
__global__ __launch_bounds__(ML_ROWS * ML_CG) void k_ml_coarse(MlDev M, int nc, int nsys, const int32_t* __restrict__ flags, int n8p) {
    extern __shared__ mb_real l_dyn[];
    mb_real* l_r8 = l_dyn;
    const int r8_words = SB * n8p > 4 * SB * ML_ROWS ? SB * n8p : 4 * SB * ML_ROWS;
    mb_real* l_part = l_dyn + r8_words;                  // [ML_CG][SB][ML_ROWS]
    const int sys0 = blockIdx.y * SB, r = threadIdx.x & (ML_ROWS - 1), cg = threadIdx.x / ML_ROWS;
    bool on[SB], any = false;
#pragma unroll
    for (int k = 0; k < SB; ++k) { on[k] = sys0 + k < nsys && flag_ld(flags + sys0 + k) == 0; any = any || on[k]; }
    if (!any) return;
#pragma unroll
    for (int k = 0; k < SB; ++k) {
        const float4* r4c = reinterpret_cast<const float4*>(M.r4c + (size_t)(sys0 + k) * 4 * M.n8);
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
        if (orow < M.n8 && sys0 + k < nsys && flag_ld(flags + sys0 + k) == 0)
            M.z8[(size_t)(sys0 + k) * M.n8 + orow] =
                (l_r8[threadIdx.x] + l_r8[Q + threadIdx.x] + l_r8[2 * Q + threadIdx.x] + l_r8[3 * Q + threadIdx.x]) * M.scale_inv[(sys0 + k) / nc];
    }
}
__global__ __launch_bounds__(FG_BLOCK) void k_ml_prolong(MlDev M, const mb_real* __restrict__ in, const mb_real* __restrict__ diag, int N, int nc,
                                                         const int32_t* __restrict__ flags, mb_real* __restrict__ out) {
    const int i = blockIdx.x * FG_BLOCK + threadIdx.x, sys = blockIdx.y;
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
constexpr int C_RHO = 0, C_PAP = 3, C_SUM = 8;  // C_SUM ring 8..10: yp . r_k (residual projection, see mb_cg)
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
                            int it, int n, int nsys, int final_pass, int sum_slot = -1) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    if (it < 0) {  // graph-replayed CG: iteration index and accumulator slots from the device counter
        it = q.it_ctr[0] - 1;
        rr_slot = C_RHO + (it + 1) % 3;
        if (sum_slot != -1) sum_slot = C_SUM + (it + 1) % 3;
        final_pass = (it + 1 >= q.max_iterations);
    }
    if (flag_ld(q.flags + (s)) == 4) flag_st(q.flags + (s), 1);
    if (flag_ld(q.flags + (s)) == 0) {
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
    mirror[s] = q.info[s];
    flag_mirror[s] = flag_ld(q.flags + (s));
}

// hand back the kept iterate of the systems that ended unconverged
__global__ void k_mbs_restore_best(int N, MbSolve q) {
    const int sys = blockIdx.y, i = blockIdx.x * FG_BLOCK + threadIdx.x;
    if (i >= N || (q.info[sys].converged && flag_ld(q.flags + (sys)) != 5) || flag_ld(q.flags + (sys)) == 3) return;
    q.x[(size_t)sys * N + i] = q.best_x[(size_t)sys * N + i];
    if (i == 0) { q.info[sys].final_residual = sc_ld(q.sc + (sys * 2)); q.info[sys].used_iterations = q.best_it[sys]; }
}

// ---------------------------------------------------------------------------------------------------------------
// On-chip CG: ONE workgroup of 1024 threads runs the WHOLE pressure solve of one env.  The cylinder meshes have 14-25 k
// cells; at that size a CG iteration of the two-kernel form above lasts as long as its launches and streams the
// five matrix fields plus five vectors of every env through the memory system every iteration.  Here the solver state of
// an env lives in the CU: r, x, P p (and the matrix diagonal) in registers -- CPT cells per thread, cell i = thread + k 1024,
// so global accesses coalesce and LDS accesses are conflict-free -- and the search direction p in LDS, where the
// neighbour gathers of the stencil hit it.  What still streams per iteration is the off-diagonal part of the matrix (4 B
// per face and cell, from L2 / Infinity Cache) and the packed neighbour table (2 B per face and cell, shared by all envs).
// No kernel launches, no device-scope atomics, no host polls inside a solve: envs are independent, so are the workgroups.
// Same recurrence, same projection of the residual, same restart / best-iterate / stall rules as mb_cg's kernels; only
// the summation order of the dot products differs (per-thread partials, wave shuffle, 16 wave sums added in fp64).
// ---------------------------------------------------------------------------------------------------------------
#ifndef OC_AGG_GROUP
#define OC_AGG_GROUP 2   // members of the aggregate-owned stencil pass loaded per batch
#endif
#ifndef OC_AGG_PIPE
#define OC_AGG_PIPE 0    // ... double-buffered (measured: 16.2-16.9 us per iteration either way; G = 4 spills: 20.9)
#endif
constexpr int OC_MAX_WAVES = 16;   // workgroups of 1024 or 512 threads (NT): 512 threads get 256 registers each

// Two-level-plus additive preconditioner of the on-chip CG (fg_mb_set_multilevel): M r = D^-1 r + 1/2 Z4 D4^-1 Z4^T r + Z8 A8^+ Z8^T r with
// piecewise-constant aggregates of 4 x 4 and 8 x 8 cells inside the blocks, D4 the diagonal of the Galerkin operator Z4^T S Z4 and
// A8^+ the (dense) pseudo-inverse of Z8^T S Z8, S = symmetric part of the pressure matrix for A = 1 (geometry only), scaled per env.
constexpr int OC_N4 = 2048, OC_N8 = 512;
struct OcPre {
    const uint16_t* a4;        // [N]   4 x 4 aggregate of every cell
    const uint16_t* parent4;   // [n4]  8 x 8 aggregate of every 4 x 4 aggregate
    const uint2* rect4;        // [n4]  the aggregate as a rectangle of cells: .x = first cell, .y = width | height << 8 | row stride << 16
    const uint2* child8;       // [n8]  the (up to four) 4 x 4 aggregates of an 8 x 8 aggregate, 16 bits each, 0xFFFF = none
    const mb_real* d4g;          // [n4]  1 / diag(Z4^T S Z4)
    const mb_real* aci8;         // [n8][ld] pseudo-inverse of Z8^T S Z8, row pitch ld = n8 rounded up to a multiple of 4
    int n4, n8;
    mb_real geom_diag_sum;       // sum_i S_ii: the env's scale is sum_i P_ii / geom_diag_sum (P = S / A with A nearly constant)
};

// Aggregate-owned layout (AGG): thread t = 4 * (8 x 8 aggregate) + child owns the (up to 16) cells of one 4 x 4 aggregate; its
// member m sits in slot t + 1024 m.  Restriction to the 4 x 4 level is then a sum over the thread's own registers, the 8 x 8 level
// a sum over the four lanes of a quad, prolongation a register broadcast: the residual copy to LDS, the two gather passes, the
// correction table and three of the six barriers of the cell-ordered preconditioner pass go away, and so do its table loads
// (rectangles, children, parents, aggregate ids).  Matrix, neighbour table and kept iterate live in slot order (k_mb_pmatrix
// writes the first; fg_mb_set_multilevel builds the second), so every per-iteration access is coalesced as before.
struct OcAgg {
    const int32_t* slot_cell;   // [16384] cell of a slot, -1 = hole
    const uint2* nbr;           // [16384] neighbour slots
    const mb_real* d4g;           // [1024]
    const int32_t* cnt;         // [1024]
    const mb_real* off4;          // [B][16384][4]
    const mb_real* diag;          // [B][16384]
    mb_real* bestx;               // [B][16384]
};

struct OcParams {
    OcPre pre;
    OcAgg agg;
    const uint32_t* nbr16;   // [N][F/2] words: one 8-byte load per cell in 2-D
    const mb_real* off4;       // [B][N][4] off-diagonals interleaved per cell (2-D), or null: q.off [B][F][N] is read instead
    int fence;               // compiler fence every four cells of the stencil pass (bounds the loads in flight)
    int dbg;                 // FG_MB_OC_VARIANT >> 8: bit 0 = per-phase cycle counts of workgroup 0 into dbg_out (fg_mb_debug_cycles)
    unsigned long long* dbg_out;   // [16] cycles per phase, summed over the iterations of the launch
    const mb_real* dt;         // [B] or null
    const mb_real* yp;         // [N] projection vector (PM == 2)
    int use_x0, project_mean, restart_every, check_every, max_iterations, stall_limit, accept_window;
    mb_real accept_factor, tol;
    fg_solve_info* info_host;   // pinned host mirrors of info[] and of the iterations run: written by the kernel itself, so the
    int32_t* its_host;          // host needs one stream synchronisation after the launch and no device-to-host copies
};

// block sum of two values in fp64.  `red` is a ring of three slot pairs used in turn (`phase` advances per call): a wave that
// already runs ahead into the next reduction writes another slot, so ONE barrier per reduction is enough (a slot is rewritten
// three reductions later, with two barriers in between).
template <int NT, bool RING = true>
__device__ __forceinline__ void oc_reduce2(mb_real a, mb_real b, double (*red)[2][OC_MAX_WAVES], int& phase, double& A, double& B) {
    a = fg_wave_sum(a);
    b = fg_wave_sum(b);
    double(*slot)[OC_MAX_WAVES] = red[RING ? phase : 0];
    if (RING) phase = phase == 2 ? 0 : phase + 1;
    if ((threadIdx.x & 63) == 0) { slot[0][threadIdx.x >> 6] = (double)a; slot[1][threadIdx.x >> 6] = (double)b; }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { sa += slot[0][w]; sb += slot[1][w]; }
    if (!RING) __syncthreads();   // single slot: nobody may rewrite it before everybody has read it
    A = sa; B = sb;   // (marking the sums wave-uniform with v_readfirstlane was measured: 16.4 -> 18.2 us per preconditioned iteration, 11.5 -> 20.6 plain)
}

// y_k = (M v)(cell k of this thread) for the vector v held in LDS; off-diagonals and neighbours stream from memory: per cell
// one 8-byte load (four packed neighbour indices) and one 16-byte load (four coefficients) in 2-D.
template <int DIMS, int CPT, bool DG_REGS, bool NB_REGS, int NT>
__device__ __forceinline__ void oc_spmv(const MbSolve& q, const OcParams& o, int sys, int N, unsigned tl, const mb_real* __restrict__ v_lds,
                                        const mb_real (&dg)[CPT], const uint2 (&nbk)[NB_REGS ? CPT : 1], mb_real (&y)[CPT]) {
    constexpr int F = 2 * DIMS;
    const mb_real* __restrict__ off = q.off + (size_t)sys * F * N;
    const mb_real* __restrict__ diag = q.diag + (size_t)sys * N;
    const unsigned un = (unsigned)N;
    if (DIMS == 2 && o.off4 != nullptr) {
        const float4* __restrict__ off4 = reinterpret_cast<const float4*>(o.off4) + (size_t)sys * N;
        const uint2* __restrict__ nb2 = reinterpret_cast<const uint2*>(o.nbr16);
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const unsigned i = tl + (unsigned)k * NT;
            mb_real acc = 0.f;
            if (i < un) {
                const uint2 u = NB_REGS ? nbk[NB_REGS ? k : 0] : nb2[i];
                const float4 c = off4[i];
                const mb_real d = DG_REGS ? dg[k] : diag[i];
                const uint32_t n0 = u.x & 0xffffu, n1 = u.x >> 16, n2 = u.y & 0xffffu, n3 = u.y >> 16;
                // prescribed face (0xFFFF): no matrix entry; the gather reads the cell itself so that it stays in bounds
                const mb_real v0 = v_lds[n0 != 0xffffu ? n0 : i], v1 = v_lds[n1 != 0xffffu ? n1 : i];
                const mb_real v2 = v_lds[n2 != 0xffffu ? n2 : i], v3 = v_lds[n3 != 0xffffu ? n3 : i];
                acc = d * v_lds[i];
                acc += n0 != 0xffffu ? c.x * v0 : 0.f;
                acc += n1 != 0xffffu ? c.y * v1 : 0.f;
                acc += n2 != 0xffffu ? c.z * v2 : 0.f;
                acc += n3 != 0xffffu ? c.w * v3 : 0.f;
            }
            y[k] = acc;
            if ((k & 3) == 3 && o.fence) asm volatile("" ::: "memory");
        }
        return;
    }
    const uint32_t* __restrict__ nb = o.nbr16;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const unsigned i = tl + (unsigned)k * NT;
        mb_real acc = 0.f;
        if (i < un) {
            acc = (DG_REGS ? dg[k] : diag[i]) * v_lds[i];
#pragma unroll
            for (int w = 0; w < DIMS; ++w) {
                const uint32_t u = nb[i * (unsigned)DIMS + (unsigned)w];
                const uint32_t n0 = u & 0xffffu, n1 = u >> 16;
                const mb_real c0 = off[(unsigned)(2 * w) * un + i], c1 = off[(unsigned)(2 * w + 1) * un + i];
                const mb_real v0 = v_lds[n0 != 0xffffu ? n0 : i], v1 = v_lds[n1 != 0xffffu ? n1 : i];
                acc += n0 != 0xffffu ? c0 * v0 : 0.f;
                acc += n1 != 0xffffu ? c1 * v1 : 0.f;
            }
        }
        y[k] = acc;
        if ((k & 3) == 3 && o.fence) asm volatile("" ::: "memory");
    }
}

// Buffer-resource addressing for the slot-ordered arrays (cdna_hip_programming.md T8): one descriptor in SGPRs per array, ONE
// per-thread byte offset, the member offset k NT as the scalar offset -- a flat pointer costs a 64-bit VGPR pair per member.
typedef unsigned int oc_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int oc_u32x2 __attribute__((ext_vector_type(2)));
using oc_rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ oc_rsrc oc_make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// the same for the aggregate-owned layout: every index is a slot, a thread's first `cnt` members are cells.  Unconditional: a
// hole's matrix row is all zeros and its neighbour word all ones.  The loads of G members are issued as ONE batch (explicit
// arrays + a scheduling barrier): left to itself the compiler, short of registers, loads and waits member by member -- sixteen
// serial L2 round trips per stencil pass, 13 of the 28 us of a preconditioned iteration (knock-out builds, -DFG_MB_OC_KNOCK).
template <int CPT, int NT, int G>
__device__ __forceinline__ mb_real oc_spmv_agg(const OcParams& o, int sys, unsigned tl, const mb_real* __restrict__ v_lds, mb_real* __restrict__ y_lds) {
    static_assert(CPT % G == 0, "member groups");
    constexpr int NB = CPT / G;
    mb_real part = 0.f;   // this thread's share of v . (M v)
    constexpr unsigned S = CPT * NT;
    const oc_rsrc R_off = oc_make_rsrc(o.agg.off4 + (size_t)sys * S * 4, S * 16u);
    const oc_rsrc R_dg = oc_make_rsrc(o.agg.diag + (size_t)sys * S, S * 4u);
    const oc_rsrc R_nb = oc_make_rsrc(o.agg.nbr, S * 8u);
    // two batches of G members: batch b + 1 is requested before batch b is consumed (OC_AGG_PIPE), so one L2 round trip is exposed
    // per stencil pass instead of one per batch
    oc_u32x2 ub[2][G];
    oc_u32x4 cb[2][G];
    mb_real dd[2][G];
    auto request = [&](int b, int buf) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const unsigned so = (unsigned)(b * G + g) * NT;
#if defined(FG_MB_OC_KNOCK) && (FG_MB_OC_KNOCK & 2)
            const unsigned i = tl + so;
            ub[buf][g].x = ((i ^ 1u) * 4u) | (((i ^ 2u) * 4u) << 16); ub[buf][g].y = ((i ^ 4u) * 4u) | (((i ^ 8u) * 4u) << 16);
            cb[buf][g].x = cb[buf][g].y = cb[buf][g].z = cb[buf][g].w = __float_as_uint(-0.2f);
            dd[buf][g] = 1.f;
#else
            ub[buf][g] = __builtin_amdgcn_raw_buffer_load_b64(R_nb, tl * 8u, so * 8u, 0);
            cb[buf][g] = __builtin_amdgcn_raw_buffer_load_b128(R_off, tl * 16u, so * 16u, 0);
            dd[buf][g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(R_dg, tl * 4u, so * 4u, 0));
#endif
        }
    };
    if (OC_AGG_PIPE) request(0, 0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int buf = OC_AGG_PIPE ? (b & 1) : 0;
        if (OC_AGG_PIPE) { if (b + 1 < NB) request(b + 1, (b + 1) & 1); }
        else request(b, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const unsigned i = tl + (unsigned)(b * G + g) * NT;
            // the table holds BYTE offsets into the LDS vector (slot * 4 <= 65 532), and a prescribed face points at the cell itself
            // with the zero coefficient k_mb_pmatrix wrote for it: no shifts, compares or selects -- the kernel is VALU-bound
            // (~130 vector instructions per member and iteration at four waves per SIMD), so every one of them counts
            const uint32_t a0 = ub[buf][g].x & 0xffffu, a1 = ub[buf][g].x >> 16, a2 = ub[buf][g].y & 0xffffu, a3 = ub[buf][g].y >> 16;
            const char* vb_ = reinterpret_cast<const char*>(v_lds);
            const mb_real vc = v_lds[i];
#if defined(FG_MB_OC_KNOCK) && (FG_MB_OC_KNOCK & 4)
            const mb_real v0 = vc, v1 = vc, v2 = vc, v3 = vc;
#else
            const mb_real v0 = *reinterpret_cast<const mb_real*>(vb_ + a0), v1 = *reinterpret_cast<const mb_real*>(vb_ + a1);
            const mb_real v2 = *reinterpret_cast<const mb_real*>(vb_ + a2), v3 = *reinterpret_cast<const mb_real*>(vb_ + a3);
#endif
            mb_real acc = dd[buf][g] * vc;
            acc += __uint_as_float(cb[buf][g].x) * v0;
            acc += __uint_as_float(cb[buf][g].y) * v1;
            acc += __uint_as_float(cb[buf][g].z) * v2;
            acc += __uint_as_float(cb[buf][g].w) * v3;
            y_lds[i] = acc;
            part += vc * acc;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return part;
}

template <int DIMS, int CPT, int PM, bool DG_REGS, int NT, bool NBR = true, bool RING = true, bool PRE = false, bool AGG = false>
__global__ __launch_bounds__(NT) void k_mbc_onchip(MbDev D, MbSolve q, OcParams o) {
    static_assert(!AGG || (PRE && DIMS == 2 && CPT == 16 && NT == 1024 && PM != 2 && !DG_REGS && !NBR), "aggregate-owned layout: 16 slots x 1024 threads, preconditioned, 2-D");
    __shared__ mb_real v_lds[CPT * NT];
    __shared__ double red[3][2][OC_MAX_WAVES];
    __shared__ mb_real l_r4[(PRE && !AGG) ? OC_N4 : 1], l_r8[PRE ? OC_N8 : 1];
    // r - mean r of the preconditioner pass, where the aggregate sums gather it; once they have, the same memory holds the
    // per-wave partial sums of the coarse solve
    constexpr int LP8 = AGG ? 256 : OC_N8;   // AGG: 4 n8 <= 1024 threads
    constexpr int RT = PRE ? ((!AGG && CPT * NT > OC_MAX_WAVES * LP8) ? CPT * NT : OC_MAX_WAVES * LP8) : 1;
    __shared__ __attribute__((aligned(16))) mb_real l_rt[RT];
    mb_real (*l_part)[LP8] = reinterpret_cast<mb_real (*)[LP8]>(l_rt);
    // AGG: M p (and, before the stencil pass, z) lives in LDS instead of 16 registers per thread -- the cell-ordered preconditioned
    // instance spills ~100 registers, and what that costs is the stencil pass: its 48 loads per thread no longer overlap
    // (knock-out builds, -DFG_MB_OC_KNOCK: 13 of 28 us per iteration are those loads, against 11.5 us for the whole plain iteration)
    __shared__ mb_real ap_lds[AGG ? CPT * NT : 1];
#define OC_AP(k, i) (*(AGG ? &ap_lds[i] : &ap[k]))
    static_assert(!PRE || NT == 1024, "the coarse solve of the preconditioner gives every one of the 16 waves its own set of columns");
    int phase = 0;
    const int sys = blockIdx.x, N = D.N, t = threadIdx.x;
    const size_t vb = (size_t)sys * N;
    // AGG: indices are slots; the thread's members k < cnt are cells.  Otherwise cell i = t + k NT < N.
    const int cnt = AGG ? o.agg.cnt[t] : 0;
    const mb_real d4g_t = AGG ? o.agg.d4g[t] : 0.f;
    // AGG: every slot is valid memory and holes hold zeros (matrix, kept iterate, LDS vectors), so nothing that touches memory is
    // conditional -- a per-member branch `k < cnt` around the loads of the stencil pass serialises their latencies (16 round
    // trips instead of one batch); only the values that would not be zero by themselves (r - mean, z) are masked with a select
#define OC_OK(k, i) (AGG || ((i) < (unsigned)N))
#define OC_M(k) (!AGG || ((k) < cnt))
    const size_t sb = AGG ? (size_t)sys * (CPT * NT) : vb;                      // base of the per-iteration arrays of this env
    const mb_real* __restrict__ diag_it = AGG ? o.agg.diag + sb : q.diag + vb;    // diagonal in the index space of the iteration
    if (!mb_active(o.dt, sys)) {
        if (t == 0) {
            flag_st(q.flags + (sys), 3);
            q.info[sys].final_residual = 0.f; q.info[sys].used_iterations = -1; q.info[sys].converged = 1; q.info[sys].is_finite = 1;
            o.info_host[sys] = q.info[sys];
            o.its_host[sys] = 0;
        }
        return;
    }
    mb_real r[CPT], x[CPT], ap[CPT], dg[CPT];
    // the packed neighbour indices of the thread's cells never change: kept in registers when they fit (2-D, interleaved
    // coefficient layout), which also takes the index load out of the stencil's dependency chain (index -> LDS address)
    constexpr bool NB_REGS = NBR && (DIMS == 2 && CPT * (NT / 512) <= 32);   // 2 registers per cell: up to 16 (32) cells at 1024 (512) threads
    uint2 nbk[NB_REGS ? CPT : 1];
    if (NB_REGS) {
#pragma unroll
        for (int k = 0; k < (NB_REGS ? CPT : 1); ++k) {
            const unsigned i = t + (unsigned)k * NT;
            nbk[k] = i < (unsigned)N ? reinterpret_cast<const uint2*>(o.nbr16)[i] : make_uint2(0xffffffffu, 0xffffffffu);
        }
    }
    const mb_real rsqn = mb_rsqrt((mb_real)N);
    const mb_real* __restrict__ rhs = q.rhs + vb;
    mb_real* __restrict__ bestx = AGG ? o.agg.bestx + sb : q.best_x + vb;
    // ---- start: x = x0 or 0, r = rhs (- M x0 through a residual pass)
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const unsigned i = t + (unsigned)k * NT;
        const bool ok = AGG ? (k < cnt) : (i < (unsigned)N);
        const unsigned cell = (AGG && ok) ? (unsigned)o.agg.slot_cell[i] : i;   // rhs and x are in cell order: a gather, once per solve
        x[k] = (ok && o.use_x0) ? q.x[vb + cell] : 0.f;
        r[k] = ok ? rhs[cell] : 0.f;
        if (AGG) { v_lds[i] = 0.f; ap_lds[i] = 0.f; }   // holes stay zero: only their owner ever reads them
        dg[k] = (DG_REGS && ok) ? q.diag[vb + i] : 0.f;
        ap[k] = 0.f;
    }
    double rr = 0.0, sr = 0.0;
    mb_real inv_s = 1.f;   // 1 / (scale of this env's matrix against the geometry-only one)
    if (PRE) {
        mb_real sd = 0.f;
#pragma unroll
        for (int k = 0; k < CPT; ++k) { const unsigned i = t + (unsigned)k * NT; if (OC_OK(k, i)) sd += diag_it[i]; }
        double dsum, unused0;
        oc_reduce2<NT, RING>(sd, 0.f, red, phase, dsum, unused0);
        inv_s = (mb_real)((double)o.pre.geom_diag_sum / dsum);
    }
    if (!o.use_x0) {
        mb_real s2 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const unsigned i = t + (unsigned)k * NT;
            if (OC_OK(k, i)) { s2 += r[k] * r[k]; s1 += r[k] * (PM == 2 ? o.yp[i] : rsqn); }
        }
        oc_reduce2<NT, RING>(s2, s1, red, phase, rr, sr);
        if (PM == 0) sr = 0.0;
    }
    double rz = 0.0, rz_prev = 1.0;
    // One loop, ONE stencil pass per trip: a trip is either a CG iteration (vector in LDS = the new search direction) or a
    // residual pass r = rhs - M x (vector in LDS = x: start from x0, the restart every 100 iterations, a recovery).
    int it = 0, best_it = 0, recoveries = 0, outcome = 0;   // outcome: 1 converged, 2 non-finite, 3 accepted on the kept iterate, 4 out of iterations / stalled
    mb_real best = 3.0e38f, crit = 0.f;
    bool fresh = true, restarted = true, residual_pass = o.use_x0 != 0, recovering = false;
    double rho = 0.0, rho_prev = 1.0;
    // per-phase cycle counters: a BUILD switch (-DFG_MB_OC_CYCLES, profiles/onchip_micro.py variant 256) -- as a run-time switch
    // the 24 extra registers pushed every instance of this kernel into scratch (30 -> 60 us per preconditioned iteration)
#ifdef FG_MB_OC_CYCLES
    unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tph = 0;
#define OC_PHASE(k) do { if (o.dbg & 1) { const unsigned long long now_ = clock64(); ph[k] += now_ - tph; tph = now_; } } while (0)
    if (o.dbg & 1) tph = clock64();
#else
#define OC_PHASE(k) do { } while (0)
#endif
    for (;;) {
        // the thread index is laundered once per trip: per-cell 64-bit addresses are invariants of this loop, and the
        // compiler otherwise hoists all of them out of it (CPT x 8 register pairs) and spills them
        unsigned tl = t;
        asm volatile("" : "+v"(tl));
        mb_real beta = 0.f, cy = 0.f;
        if (!residual_pass) {
            rho = rr - sr * sr;   // |r - (yp.r) yp|^2
            crit = mb_rms(rho, N);
            if (!(crit >= o.tol)) {
                if (isfinite(crit)) { outcome = 1; break; }
                // the recurrence broke down (p.Pp <= 0 or overflow on the non-symmetric matrix): back to the kept iterate
                if (recovering || recoveries >= 3 || it + o.check_every >= o.max_iterations) { outcome = 2; break; }
                ++recoveries;
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const unsigned i = tl + (unsigned)k * NT;
                    if (OC_OK(k, i)) { const mb_real v = bestx[i]; x[k] = isfinite(v) ? v : 0.f; }
                }
                residual_pass = true; recovering = true;
            } else {
                recovering = false;
                // keep x_it when it beats the kept iterate by 2x (or at all inside the acceptance band): returnBestResult
                if (it == 0 || crit < 0.5f * best || (crit < o.accept_factor * o.tol && crit < best)) {
                    best = crit; best_it = it;
#pragma unroll
                    for (int k = 0; k < CPT; ++k) { const unsigned i = tl + (unsigned)k * NT; if (OC_OK(k, i)) bestx[i] = x[k]; }
                }
                if (it > 0 && it % o.check_every == 0) {   // the cadence of k_mbs_check in the chunked solver
                    if (o.accept_factor > 0.f && best <= o.accept_factor * o.tol && (it - 1) - best_it >= o.accept_window) { outcome = 3; break; }
                    if (o.stall_limit > 0 && (it - 1) - best_it > o.stall_limit) { outcome = 4; break; }
                }
                if (it >= o.max_iterations) { outcome = 4; break; }
                if (it > 0 && it % o.restart_every == 0 && !restarted) residual_pass = true;   // residualResetSteps (cg_solver_kernel.cu:281-300)
            }
        }
        mb_real zbar = 0.f;
        OC_PHASE(0);   // checks, best-iterate store
        if (!residual_pass) {
            restarted = false;
            beta = fresh ? 0.f : (mb_real)(rho / rho_prev);
            cy = (mb_real)sr;
            if (PRE && AGG) {
                // ---- z = M (r - mean r), aggregate-owned: the 4 x 4 sum is a sum over the thread's registers (members in the
                // row-major order the gather of the cell-ordered form walks), the 8 x 8 sum the sum of a quad's lanes in child order
                const mb_real rm = PM == 1 ? cy * rsqn : 0.f;
                const int n8 = o.pre.n8;
                mb_real r4 = 0.f;
#pragma unroll
                for (int k = 0; k < CPT; ++k) r4 += (k < cnt) ? r[k] - rm : 0.f;
                {
                    const int q0 = (t & 63) & ~3;
                    const mb_real c0 = __shfl(r4, q0, 64), c1 = __shfl(r4, q0 + 1, 64), c2 = __shfl(r4, q0 + 2, 64), c3 = __shfl(r4, q0 + 3, 64);
                    if ((t & 3) == 0 && (t >> 2) < n8) l_r8[t >> 2] = ((c0 + c1) + c2) + c3;
                }
                __syncthreads();
                {
                    const int ld = (n8 + 3) & ~3, nq = ld >> 2, grp = t >> 6;
                    for (int qd = t & 63; qd < nq; qd += 64) {
                        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#if !(defined(FG_MB_OC_KNOCK) && (FG_MB_OC_KNOCK & 1))
#pragma unroll 8
                        for (int c = grp; c < n8; c += OC_MAX_WAVES) {
                            const float4 a = *reinterpret_cast<const float4*>(o.pre.aci8 + (unsigned)c * (unsigned)ld + 4u * (unsigned)qd);
                            const mb_real rc = l_r8[c];
                            acc.x += a.x * rc; acc.y += a.y * rc; acc.z += a.z * rc; acc.w += a.w * rc;
                        }
#endif
                        *reinterpret_cast<float4*>(&l_part[grp][4 * qd]) = acc;
                    }
                }
                __syncthreads();
                const oc_rsrc R_dg = oc_make_rsrc(diag_it, (unsigned)(CPT * NT) * 4u);
                mb_real corr = 0.f;
                if (cnt > 0) {
                    mb_real e = 0.f;
#pragma unroll
                    for (int g = 0; g < OC_MAX_WAVES; ++g) e += l_part[g][t >> 2];
                    corr = inv_s * (0.5f * r4 * d4g_t + e);
                }
                mb_real s_rz = 0.f, s_z = 0.f;
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const unsigned i = tl + (unsigned)k * NT;
                    const mb_real rt = (k < cnt) ? r[k] - rm : 0.f;
#if defined(FG_MB_OC_KNOCK) && (FG_MB_OC_KNOCK & 8)
                    const mb_real z = (k < cnt) ? rt + corr : 0.f;
#else
                    const mb_real dk = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(R_dg, tl * 4u, (unsigned)k * NT * 4u, 0));
                    const mb_real z = (k < cnt) ? rt * __builtin_amdgcn_rcpf(dk) + corr : 0.f;
#endif
                    OC_AP(k, i) = z;
                    s_rz += rt * z; s_z += z;
                }
                double zsum;
                oc_reduce2<NT, RING>(s_rz, s_z, red, phase, rz, zsum);   // its barrier also orders the reads of l_part before the next pass writes it
                zbar = PM == 1 ? (mb_real)(zsum / (double)N) : 0.f;
                beta = fresh ? 0.f : (mb_real)(rz / rz_prev);
                OC_PHASE(6);
            } else if (PRE) {
                // ---- z = M (r - mean r): restrict to the 4 x 4 and 8 x 8 aggregates (LDS atomics), dense coarse solve by the
                // waves (one row per wave and pass, lanes over the columns), corrections summed top-down into l_r4
                const mb_real rm = PM == 1 ? cy * rsqn : 0.f;
                const int n4 = o.pre.n4, n8 = o.pre.n8;
                // restriction as GATHERS: every aggregate is a rectangle of cells of one block, so one thread sums it out of the
                // LDS copy of the residual; 8 x 8 aggregates sum their (up to four) children.  LDS mb_real atomics did this first
                // and cost 21 us per application (14 k atomics on 912 addresses)
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const unsigned i = tl + (unsigned)k * NT;
                    if (i < (unsigned)N) l_rt[i] = r[k] - rm;
                }
                __syncthreads();
                OC_PHASE(1);   // residual copy to LDS
                for (int a = t; a < n4; a += NT) {
                    const uint2 rc = o.pre.rect4[a];
                    const unsigned w = rc.y & 0xffu, h = (rc.y >> 8) & 0xffu, stride = rc.y >> 16;
                    mb_real sum = 0.f;
                    for (unsigned dy = 0; dy < h; ++dy)
                        for (unsigned dx = 0; dx < w; ++dx) sum += l_rt[rc.x + dy * stride + dx];
                    l_r4[a] = sum;
                }
                __syncthreads();
                OC_PHASE(2);   // 4 x 4 sums
                for (int a = t; a < n8; a += NT) {
                    const uint2 ch = o.pre.child8[a];
                    const unsigned c0 = ch.x & 0xffffu, c1 = ch.x >> 16, c2 = ch.y & 0xffffu, c3 = ch.y >> 16;
                    l_r8[a] = (c0 != 0xffffu ? l_r4[c0] : 0.f) + (c1 != 0xffffu ? l_r4[c1] : 0.f) + (c2 != 0xffffu ? l_r4[c2] : 0.f) +
                              (c3 != 0xffffu ? l_r4[c3] : 0.f);
                }
                __syncthreads();
                OC_PHASE(3);   // 8 x 8 sums
                // e8 = A8^+ r8.  A8^+ is symmetric, so row r is read as column r of consecutive rows: wave g takes the columns
                // c = g, g + 16, ..., lane q the four rows 4q .. 4q+3 -- every load is a 16-byte access, a wave reads 1 KiB
                // contiguous, and the ~15 loads of a lane are independent (128 KiB in flight per workgroup: the matrix, 208 KB
                // at 228 aggregates, streams from L2 once per iteration).  A wave-per-row version spent 14 dependent
                // load -> reduce round trips here (18 us), one thread per (row, quarter of the columns) 10 us.
                {
                    const int ld = (n8 + 3) & ~3, nq = ld >> 2, grp = t >> 6;
                    for (int qd = t & 63; qd < nq; qd += 64) {
                        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
                        for (int c = grp; c < n8; c += OC_MAX_WAVES) {
                            const float4 a = *reinterpret_cast<const float4*>(o.pre.aci8 + (unsigned)c * (unsigned)ld + 4u * (unsigned)qd);
                            const mb_real rc = l_r8[c];
                            acc.x += a.x * rc; acc.y += a.y * rc; acc.z += a.z * rc; acc.w += a.w * rc;
                        }
                        *reinterpret_cast<float4*>(&l_part[grp][4 * qd]) = acc;
                    }
                }
                __syncthreads();
                OC_PHASE(4);   // coarse solve
                // corrections summed top-down into the 4 x 4 table: its own half-weighted Jacobi term + the coarse solution of its
                // parent (the 16 per-wave partial sums are added here, by every child: one barrier less than a separate pass)
                for (int a = t; a < n4; a += NT) {
                    const int row = o.pre.parent4[a];
                    mb_real e = 0.f;
#pragma unroll
                    for (int g = 0; g < OC_MAX_WAVES; ++g) e += l_part[g][row];
                    l_r4[a] = inv_s * (0.5f * l_r4[a] * o.pre.d4g[a] + e);   // d4g holds reciprocals
                }
                __syncthreads();
                OC_PHASE(5);   // correction table
                mb_real s_rz = 0.f, s_z = 0.f;
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const unsigned i = tl + (unsigned)k * NT;
                    if (i < (unsigned)N) {
                        const mb_real rt = r[k] - rm;
                        const mb_real z = rt * __builtin_amdgcn_rcpf(DG_REGS ? dg[k] : q.diag[vb + i]) + l_r4[o.pre.a4[i]];   // v_rcp_f32: a preconditioner needs no IEEE division
                        ap[k] = z;   // ap is free until the stencil pass rewrites it
                        s_rz += rt * z; s_z += z;
                    }
                }
                double zsum;
                oc_reduce2<NT, RING>(s_rz, s_z, red, phase, rz, zsum);
                zbar = PM == 1 ? (mb_real)(zsum / (double)N) : 0.f;
                beta = fresh ? 0.f : (mb_real)(rz / rz_prev);
                OC_PHASE(6);   // z pass + r.z reduction
            }
        }
        // ---- the vector the stencil is applied to: x, or p = (r - (yp.r) yp) + beta p (every thread rewrites its own cells)
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const unsigned i = tl + (unsigned)k * NT;
            if (OC_OK(k, i)) {
                mb_real v;
                if (residual_pass) v = x[k];
                else {
                    v = PM == 0 ? r[k] : (PM == 1 ? r[k] - cy * rsqn : r[k] - cy * o.yp[i]);
                    if (PRE) v = OC_M(k) ? OC_AP(k, i) - zbar : 0.f;   // z of this cell, parked in ap by the preconditioner pass
                    if (!fresh) v += beta * v_lds[i];
                }
                v_lds[i] = v;
            }
        }
        __syncthreads();
        OC_PHASE(7);   // direction update
        mb_real part = 0.f;
        if constexpr (AGG) part = oc_spmv_agg<CPT, NT, OC_AGG_GROUP>(o, sys, tl, v_lds, ap_lds);
        else oc_spmv<DIMS, CPT, DG_REGS, NB_REGS, NT>(q, o, sys, N, tl, v_lds, dg, nbk, ap);
        OC_PHASE(8);   // stencil pass
        mb_real s2 = 0.f, s1 = 0.f;
        if (residual_pass) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                const unsigned i = tl + (unsigned)k * NT;
                if (AGG ? (k < cnt) : (i < (unsigned)N)) {
                    r[k] = rhs[AGG ? (unsigned)o.agg.slot_cell[i] : i] - OC_AP(k, i);
                    s2 += r[k] * r[k];
                    s1 += r[k] * (PM == 2 ? o.yp[i] : rsqn);
                }
            }
            oc_reduce2<NT, RING>(s2, s1, red, phase, rr, sr);   // its barrier also separates the stencil's LDS reads from the next writes
            if (PM == 0) sr = 0.0;
            residual_pass = false; fresh = true; restarted = true;
            continue;
        }
        if (!AGG) {
#pragma unroll
            for (int k = 0; k < CPT; ++k) { const unsigned i = tl + (unsigned)k * NT; if (OC_OK(k, i)) part += v_lds[i] * ap[k]; }
        }
        double pap, unused;
        oc_reduce2<NT, RING>(part, 0.f, red, phase, pap, unused);
        OC_PHASE(9);   // p.Pp reduction
        const mb_real alpha = (mb_real)((PRE ? rz : rho) / pap);
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const unsigned i = tl + (unsigned)k * NT;
            if (OC_OK(k, i)) {
                x[k] += alpha * v_lds[i];
                r[k] -= alpha * OC_AP(k, i);
                s2 += r[k] * r[k];
                s1 += r[k] * (PM == 2 ? o.yp[i] : rsqn);
            }
        }
        oc_reduce2<NT, RING>(s2, s1, red, phase, rr, sr);
        if (PM == 0) sr = 0.0;
        rho_prev = rho;
        rz_prev = rz;
        fresh = false;
        ++it;
        OC_PHASE(10);  // x, r update + r.r reduction
    }
#ifdef FG_MB_OC_CYCLES
    if ((o.dbg & 1) && sys == 0 && t == 0 && o.dbg_out) {
        for (int k = 0; k < 11; ++k) o.dbg_out[k] = ph[k];
        o.dbg_out[11] = (unsigned long long)it;
    }
#endif
#undef OC_PHASE
    // ---- hand back: the last iterate when converged, the kept one otherwise (k_mbs_restore_best)
    const bool use_best = (outcome == 3 || outcome == 4 || (outcome == 2 && best < 3.0e38f));
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const unsigned i = t + (unsigned)k * NT;
        if (AGG ? (k < cnt) : (i < (unsigned)N)) q.x[vb + (AGG ? (unsigned)o.agg.slot_cell[i] : i)] = use_best ? bestx[i] : x[k];
    }
#undef OC_OK
#undef OC_M
#undef OC_AP
    if (t == 0) {
        flag_st(q.flags + (sys), outcome == 2 ? 2 : (outcome == 3 ? 5 : 1));
        q.info[sys].final_residual = use_best ? best : crit;
        q.info[sys].used_iterations = use_best ? best_it : it;
        q.info[sys].converged = (outcome == 1 || outcome == 3) ? 1 : 0;
        q.info[sys].is_finite = outcome != 2 ? 1 : 0;
        q.best_it[sys] = it;   // total iterations run (profiling: the host sums them)
        o.info_host[sys] = q.info[sys];
        o.its_host[sys] = it;
    }
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

#define MB_DISPATCH(s, ...)                    \
    do {                                       \
        if ((s)->d == 2) { constexpr int DIMS = 2; __VA_ARGS__ } \
        else { constexpr int DIMS = 3; __VA_ARGS__ }            \
    } while (0)

#define MB_DISPATCH_PM(s, pm, ...)                                             \
    do {                                                                        \
        if ((pm) == 0) { constexpr int PM = 0; MB_DISPATCH(s, __VA_ARGS__); }   \
        else if ((pm) == 1) { constexpr int PM = 1; MB_DISPATCH(s, __VA_ARGS__); } \
        else { constexpr int PM = 2; MB_DISPATCH(s, __VA_ARGS__); }              \
    } while (0)

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

int mb_poll(fg_mb_state* s, int nsys, hipStream_t st, bool& done) {
    FG_HIP_CHECK(hipStreamSynchronize(st));
    done = true;
    for (int i = 0; i < nsys; ++i) done = done && s->flags_pinned[i] != 0;
    return FG_OK;
}

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

template <typename T>
int mb_alloc(fg_mb_state* s, T** p, size_t count);

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
void mb_ml_apply(fg_mb_state* s, const MbSolve& q, const mb_real* in, mb_real* out, hipStream_t st, int fused = 0, int it = 0) {
    const MlDev M = mb_ml_dev(s);
    const int nsys = s->B * q.nc, n = s->N;
    const dim3 rgrid4((4 * M.n4 + FG_BLOCK - 1) / FG_BLOCK, nsys);   // four threads per aggregate
    if (fused == 1) { MB_DISPATCH(s, hipLaunchKernelGGL(k_ml_restrict_p<DIMS>, rgrid4, dim3(FG_BLOCK), 0, st, s->dev, q, M, it);); }
    else if (fused == 2) { MB_DISPATCH(s, hipLaunchKernelGGL(k_ml_restrict_s<DIMS>, rgrid4, dim3(FG_BLOCK), 0, st, s->dev, q, M, it);); }
    else
    hipLaunchKernelGGL(k_ml_restrict, rgrid4, dim3(FG_BLOCK), 0, st, M, in, n, (const int32_t*)q.flags);
    {
        // systems per workgroup: 8 when that still leaves >= 2 workgroups per CU-pair of work (>= 32 systems) and the LDS fits 64 KB
        const int n8p = (M.n8 + 3) & ~3;
        const auto words = [&](int sb) { return (sb * n8p > 4 * sb * ML_ROWS ? sb * n8p : 4 * sb * ML_ROWS) + ML_CG * sb * ML_ROWS; };
        const int want = s->dbg_ml_sb ? s->dbg_ml_sb : (nsys >= 32 ? 8 : 4);
        if (want == 8 && words(8) * 4 <= 64 * 1024)
            hipLaunchKernelGGL(k_ml_coarse<8>, dim3((M.n8 + ML_ROWS - 1) / ML_ROWS, (nsys + 7) / 8), dim3(ML_ROWS * ML_CG), (size_t)words(8) * 4, st, M, q.nc, nsys, (const int32_t*)q.flags, n8p);
        else
            hipLaunchKernelGGL(k_ml_coarse<4>, dim3((M.n8 + ML_ROWS - 1) / ML_ROWS, (nsys + 3) / 4), dim3(ML_ROWS * ML_CG), (size_t)words(4) * 4, st, M, q.nc, nsys, (const int32_t*)q.flags, n8p);
    }
    hipLaunchKernelGGL(k_ml_prolong, dim3((n + FG_BLOCK - 1) / FG_BLOCK, nsys), dim3(FG_BLOCK), 0, st, M, in, q.diag, n, q.nc, (const int32_t*)q.flags, out);
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
                mb_real tol, int max_iterations, int use_x0, int* max_it, hipStream_t st, int project = 0, int refine = 0, int multilevel = 0,
                int pred_slot = 31) {
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
        MB_DISPATCH(s, {   // vec_mask: which of the five kernels run in their four-cell form
            if (fused_pv) {
                if ((vec_mask & 3) == 3) hipLaunchKernelGGL(k_mbb_pv4<DIMS>, grid4, blk, 0, st, s->dev, q, li); else hipLaunchKernelGGL(k_mbb_pv<DIMS>, grid, blk, 0, st, s->dev, q, li);
            } else {
            if (ml_fused) {}   // p is formed inside the restriction (mb_ml_apply below)
            else if (vec_mask & 1) hipLaunchKernelGGL(k_mbb_p4<DIMS>, grid4, blk, 0, st, s->dev, q, li); else hipLaunchKernelGGL(k_mbb_p<DIMS>, grid, blk, 0, st, s->dev, q, li);
            if (ilu) mb_ilu_apply(s, q, q.p, s->ilu_mp, st);
            else if (ml) mb_ml_apply(s, q, q.p, s->ml_mp, st, ml_fused ? 1 : 0, li);
            if (vec_mask & 2) hipLaunchKernelGGL(k_mbb_v4<DIMS>, grid4, blk, 0, st, s->dev, q, li); else hipLaunchKernelGGL(k_mbb_v<DIMS>, grid, blk, 0, st, s->dev, q, li);
            }
            if (fused_st) {
                if ((vec_mask & 12) == 12) hipLaunchKernelGGL(k_mbb_st4<DIMS>, grid4, blk, 0, st, s->dev, q, li); else hipLaunchKernelGGL(k_mbb_st<DIMS>, grid, blk, 0, st, s->dev, q, li);
            } else {
                if (ml_fused) {}   // s is formed inside the restriction
                else if (vec_mask & 4) hipLaunchKernelGGL(k_mbb_s4<DIMS>, grid4, blk, 0, st, s->dev, q, li); else hipLaunchKernelGGL(k_mbb_s<DIMS>, grid, blk, 0, st, s->dev, q, li);
                if (ilu) mb_ilu_apply(s, q, q.r, s->ilu_ms, st);
                else if (ml) mb_ml_apply(s, q, q.r, s->ml_ms, st, ml_fused ? 2 : 0, li);
                if (vec_mask & 8) hipLaunchKernelGGL(k_mbb_t4<DIMS>, grid4, blk, 0, st, s->dev, q, li); else hipLaunchKernelGGL(k_mbb_t<DIMS>, grid, blk, 0, st, s->dev, q, li);
            }
            if (vec_mask & 16) hipLaunchKernelGGL(k_mbb_x4<DIMS>, grid4, blk, 0, st, s->dev, q, li); else hipLaunchKernelGGL(k_mbb_x<DIMS>, grid, blk, 0, st, s->dev, q, li);
        });
        if (it + 1 >= next_poll || it + 1 == max_iterations) {
            next_poll = it + 1 + (it < 20 ? 2 : 10);   // long (pressure) solves: fewer host round trips
            hipLaunchKernelGGL(k_mbs_check, sg, sb, 0, st, q, s->info_pinned, s->flags_pinned, A_RR, it, n, nsys, (int)(it + 1 == max_iterations));
            if (int rc = mb_poll(s, nsys, st, done)) return rc;
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
                    hipLaunchKernelGGL(k_mbs_check, sg, sb, 0, st, q, s->info_pinned, s->flags_pinned, A_RR, it, n, nsys, 0);
                    if (int rc = mb_poll(s, nsys, st, done)) return rc;
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

#define OC_FIRST(a, ...) a
#define OC_LAUNCH(CPT_, PM_, DGR_, ...)                                                                                        \
    do {                                                                                                                       \
        if (ev) hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_mbc_onchip<2, CPT_, PM_, DGR_, __VA_ARGS__>), dim3(nsys), dim3(OC_FIRST(__VA_ARGS__)), 0, st, \
                                      s->prof_ev_oc[0], s->prof_ev_oc[1], 0, s->dev, q, o);                                    \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mbc_onchip<2, CPT_, PM_, DGR_, __VA_ARGS__>), dim3(nsys), dim3(OC_FIRST(__VA_ARGS__)), 0, st, s->dev, q, o); \
    } while (0)
#define OC_LAUNCH_PM(CPT_, DGR_, ...) do { if (pm_mode == 0) OC_LAUNCH(CPT_, 0, DGR_, __VA_ARGS__); else OC_LAUNCH(CPT_, 1, DGR_, __VA_ARGS__); } while (0)

constexpr int OC_MAX_CELLS = 28 * 1024;

bool mb_onchip_ok(const fg_mb_state* s, int pm_mode) {
    return s->onchip_mode && s->d == 2 && s->nbr16 != nullptr && s->N <= OC_MAX_CELLS && pm_mode != 2;
}

// the whole CG solve of every env in one launch (k_mbc_onchip); same arguments and results as mb_cg
int mb_cg_onchip(fg_mb_state* s, const mb_real* dt, const mb_real* diag, const mb_real* off, const mb_real* rhs, mb_real* x, mb_real tol,
                 int max_iterations, int use_x0, int pm_mode, mb_real stall_accept, int* max_it, hipStream_t st) {
#if FG_MB_F64
    (void)dt; (void)diag; (void)off; (void)rhs; (void)x; (void)tol; (void)max_iterations; (void)use_x0; (void)pm_mode; (void)stall_accept; (void)max_it; (void)st;
    fg_set_error("the on-chip CG is not part of the fp64 build");   // (never reached: fg_mb_create switches it off there)
    return FG_ERR_UNSUPPORTED;
#else
    const int nsys = s->B, n = s->N;
    MbSolve q = mb_solve_ptrs(s, diag, off, rhs, x, 1, tol);
    q.best_x = s->w[4]; q.best_it = s->best_it;
    constexpr int CG_CHUNK = 20, CG_RESTART = 100;
    OcParams o;
    o.nbr16 = s->nbr16; o.dt = dt; o.yp = s->dev.yproj;
    o.off4 = (off == s->Poff && !(s->oc_variant & 2)) ? s->Poff4 : nullptr;
    o.fence = (s->oc_variant & 1) ? 0 : 1;
    o.dbg = s->oc_variant >> 8;
    o.dbg_out = s->oc_dbg;
    o.use_x0 = use_x0; o.project_mean = pm_mode; o.restart_every = CG_RESTART; o.check_every = CG_CHUNK;
    o.max_iterations = ((max_iterations + CG_CHUNK - 1) / CG_CHUNK) * CG_CHUNK;
    o.stall_limit = s->cg_stall_limit; o.accept_window = 20;
    o.accept_factor = stall_accept > 1.f ? stall_accept : 0.f; o.tol = tol;
    o.info_host = s->info_pinned; o.its_host = s->flags_pinned;
    const bool ev = s->prof_on != 0;
    // Instances, chosen by measurement on the cylinder mesh (profiles/r02_onchip_variants.txt; 64 envs x 14 232 cells, us per
    // iteration): 16 cells per thread with the two-barrier reduction 11.7-11.8; the same with the one-barrier ring 14.5; 14
    // cells per thread 15.8-16.3; neighbour indices in registers 17.4; 512 threads x 256 registers 18.3 -- what the
    // compiler's schedule makes of each form decides, not the instruction count.  Small meshes keep the indices in registers.
    const bool pre = s->ml_on && s->ml_a4 != nullptr && n <= 16 * 1024 && s->ml_n4 <= OC_N4 && s->ml_n8 <= OC_N8;   // LDS: p, r - mean r and the aggregate tables
    o.pre.a4 = s->ml_a4; o.pre.parent4 = s->ml_parent4; o.pre.d4g = s->ml_d4g; o.pre.aci8 = s->ml_aci8;
    o.pre.rect4 = s->ml_rect4; o.pre.child8 = s->ml_child8;
    o.pre.n4 = s->ml_n4; o.pre.n8 = s->ml_n8; o.pre.geom_diag_sum = s->ml_geom_diag_sum;
    // the aggregate-owned layout (fg_mb.h) when its tables are installed and the matrix is the pressure matrix k_mb_pmatrix wrote
    // (sixteen members per thread whatever the mesh: below 8 k cells the cell-ordered instances with four / eight cells per thread
    //  are faster -- measured per iteration: 1 984 cells 8.2 against 12.8 us, 6 192 cells 11.7 against 13.6, 14 232 cells 26.9 against 14.6)
    const bool agg = pre && s->oc_agg && s->dbg_oc_agg && !s->oc_matrix_stale && diag == s->Pdiag && off == s->Poff && n > 8 * 1024;
    o.agg.slot_cell = s->oc_slot_cell; o.agg.nbr = s->oc_nbr; o.agg.d4g = s->oc_d4g; o.agg.cnt = s->oc_cnt;
    o.agg.off4 = s->Poff4s; o.agg.diag = s->Pdiag_s; o.agg.bestx = s->oc_bestx;
#define OC_LAUNCH_PRE(CPT_, DGR_, NBR_) do { if (pre) OC_LAUNCH_PM(CPT_, DGR_, 1024, NBR_, false, true); else OC_LAUNCH_PM(CPT_, DGR_, 1024, NBR_, false, false); } while (0)
#ifndef OC_AGG_RING
#define OC_AGG_RING false
#endif
    if (agg) OC_LAUNCH_PM(16, false, 1024, false, OC_AGG_RING, true, true);
    else if (n <= 4 * 1024) OC_LAUNCH_PRE(4, true, true);
    else if (n <= 8 * 1024) OC_LAUNCH_PRE(8, true, true);
    else if (n <= 16 * 1024) OC_LAUNCH_PRE(16, false, false);
    else if (n <= 24 * 1024) OC_LAUNCH_PM(24, false, 1024, false, false, false);
    else OC_LAUNCH_PM(28, false, 1024, false, false, false);
#undef OC_LAUNCH_PRE
    FG_HIP_CHECK(hipStreamSynchronize(st));   // info_pinned / flags_pinned (iterations run) were written by the kernel
    if (ev) {
        fg_f32 ms = 0.f;
        FG_HIP_CHECK(hipEventElapsedTime(&ms, s->prof_ev_oc[0], s->prof_ev_oc[1]));
        long long its = 0;
        for (int i = 0; i < nsys; ++i) its += s->flags_pinned[i] > 0 ? s->flags_pinned[i] : 0;
        // bytes the kernel streams: per iteration and cell the off-diagonals (4 F) and the packed neighbour table (2 F);
        // per solve and cell rhs, x0 / x, diagonal, kept iterate (about 20 B)
        s->prof_ms[2] += ms;
        s->prof_bytes[2] += (double)its * n * (6.0 * s->F) + (double)nsys * n * 20.0;
        s->prof_n[2] += 1; s->prof_launches[2] += 1;
        s->prof_its += its;
    }
    return mb_finish(s, nsys, nullptr, max_it);
#endif
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
    auto enqueue_chunk = [&](bool sample) {
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
        hipLaunchKernelGGL(k_mbs_check, sg, sb, 0, st, q, s->info_pinned, s->flags_pinned, 0, -1, n, nsys, 0, project_mean ? 0 : -1);
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
            enqueue_chunk(false);
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
        if (use_graph) FG_HIP_CHECK(hipGraphLaunch(s->cg_graph_exec, st));
        else enqueue_chunk(s->prof_on && (s->prof_chunk++ % 4 == 0));
        if (int rc = mb_poll(s, nsys, st, done)) return rc;
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

template <typename T>
int mb_alloc(fg_mb_state* s, T** p, size_t count) {
    void* q = nullptr;
    FG_HIP_CHECK(hipMalloc(&q, (count ? count : 1) * sizeof(T)));
    FG_HIP_CHECK(hipMemset(q, 0, (count ? count : 1) * sizeof(T)));
    s->owned.push_back(q);
    *p = (T*)q;
    return FG_OK;
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
        e = getenv("FG_MB_SCALAR_CG"); s->dbg_scalar_cg = (e && e[0] == '1') ? 1 : 0;
        e = getenv("FG_MB_BICG_FUSE"); s->dbg_fuse_st = e ? atoi(e) : 2;   // 0 five kernels, 1 s / t fused, 2 also p / v (default)
        e = getenv("FG_MB_PRED"); s->dbg_pred = (e && e[0] == '0') ? 0 : 1;
        e = getenv("FG_MB_ML_FUSE"); s->dbg_ml_fuse = e ? atoi(e) : 1;
        e = getenv("FG_MB_ML_SB"); s->dbg_ml_sb = e ? atoi(e) : 0;   // systems per workgroup of k_ml_coarse: 4 / 8, 0 = by batch size
        e = getenv("FG_MB_ML_TRY_CAP"); if (e && atoi(e) > 0) s->dbg_ml_cap = atoi(e);
        e = getenv("FG_MB_ML_WARMUP"); if (e && atoi(e) > 0) { s->dbg_ml_warmup = atoi(e); s->ml_bicg_skip = s->dbg_ml_warmup; }
        s->dbg_graph = getenv("FG_MB_GRAPH") != nullptr;
        s->dbg_trace = getenv("FG_MB_TRACE") != nullptr;
        s->dbg_fail = getenv("FG_MB_TRACE_FAIL") != nullptr;
        e = getenv("FG_MB_ONCHIP"); s->onchip_mode = (e && e[0] == '0') ? 0 : 1;
        e = getenv("FG_MB_OC_AGG"); s->dbg_oc_agg = (e && e[0] == '0') ? 0 : 1;
        e = getenv("FG_MB_RUNG_ILU"); s->dbg_rung_ilu = (e && e[0] == '0') ? 0 : 1;
        e = getenv("FG_MB_OC_VARIANT"); s->oc_variant = e ? atoi(e) : 0;   // bit 0: no compiler fences in the stencil pass; bit 1: split [F][N] coefficient layout
    }
#if FG_MB_F64
    // the fp64 build runs the one-cell-per-thread kernels: the four-cell forms (float4), the on-chip CG and the multilevel
    // preconditioner are written for 32-bit words (fg_mb.h)
    s->dbg_vec_mask = 0; s->dbg_scalar_cg = 1; s->onchip_mode = 0; s->dbg_oc_agg = 0;
#endif
    *out = s;
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
    FG_HIP_CHECK(hipHostMalloc((void**)&s->dt_pinned, sizeof(mb_real) * B, hipHostMallocDefault));
    FG_HIP_CHECK(hipHostMalloc((void**)&s->flags_pinned, sizeof(int32_t) * B * d, hipHostMallocDefault));
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
    // working copy of dt: envs whose solve turns out non-finite are masked out of the rest of the step (k_mb_mask_failed)
    FG_HIP_CHECK(hipMemcpyAsync(s->dt_step, dt_B, sizeof(mb_real) * B, hipMemcpyDeviceToDevice, st));
    FG_HIP_CHECK(hipMemsetAsync(s->env_fail, 0, sizeof(int32_t) * B, st));
    dt_B = s->dt_step;
    auto mask_failed = [&](int nc) {
        hipLaunchKernelGGL(k_mb_mask_failed, dim3((B + 63) / 64), dim3(64), 0, st, B, nc, (const fg_solve_info*)s->info_dev, s->dt_step, s->env_fail);
    };
    // pressure of the envs at the start of the step: a dropped env gets it back at the end (its velocity is never committed; its
    // pressure is written by every corrector's mean removal, so a failure in corrector 1 would leave corrector 0's behind)
    hipLaunchKernelGGL(k_mb_copy, dim3((unsigned)((N + FG_BLOCK - 1) / FG_BLOCK), B), blk, 0, st, (size_t)N, (const mb_real*)nullptr, (const mb_real*)s->pressure, s->pres_bak);
    const size_t vel_env = (size_t)d * N;
    const dim3 gcopy((unsigned)((vel_env + FG_BLOCK - 1) / FG_BLOCK), B);
    MB_DISPATCH(s, {
        // ---- predictor (SIM.py:1646-1762, non-orthogonal branch)
        hipLaunchKernelGGL(k_mb_contra<DIMS>, gc, blk, 0, st, D, dt_B, s->velocity, s->bvel, s->cc, s->fb);
        hipLaunchKernelGGL(k_mb_matrix<DIMS>, gn, blk, 0, st, D, dt_B, s->nu, s->cc, s->fb, s->Cdiag, s->Coff, s->rA);
        hipLaunchKernelGGL(k_mb_copy, gcopy, blk, 0, st, vel_env, dt_B, s->velocity, s->ures);  // CopyVelocityResultFromBlocks
        for (int no = 0; no < opt->advect_non_ortho_steps; ++no) {
            hipLaunchKernelGGL(k_mb_vrhs<DIMS>, gv, blk, 0, st, D, dt_B, s->nu, s->velocity, s->ures, s->bvel, s->fb, s->source, s->rhs);
            int m = 0;
            // initial guess: zero on the first non-orthogonal pass, the previous pass's result after that (x = None if no_step == 0
            // or not advect_non_ortho_reuse_result, PISOtorch_simulation.py:1735-1742; tests/golden/reference_split_step.json);
            // fg_mb_set_advection_start(1): the current velocity on the first pass (opt-in)
            int vrc = mb_bicgstab(s, dt_B, s->Cdiag, s->Coff, s->rhs, s->ures, d, opt->advection_tol,
                                  opt->max_iterations, (no > 0 || s->adv_from_result) ? 1 : 0, &m, st, 0, 0, 0, no & 3);
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
            if (c == 0) { hipLaunchKernelGGL(k_mb_pmatrix<DIMS>, gn, blk, 0, st, D, dt_B, s->rA, s->Pdiag, s->Poff, s->Poff4, s->oc_agg ? s->oc_cell_slot : nullptr, s->Poff4s, s->Pdiag_s); s->oc_matrix_stale = false; }
            for (int ps = 0; ps < opt->pressure_non_ortho_steps; ++ps) {
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
                hipLaunchKernelGGL(k_mb_sub_mean, gn, blk, 0, st, N, dt_B, s->red8, s->pres, s->pressure);
            }
            hipLaunchKernelGGL(k_mb_correct<DIMS>, gn, blk, 0, st, D, dt_B, s->rA, s->hvec, s->pressure, s->ures);
        }
        hipLaunchKernelGGL(k_mb_copy, gcopy, blk, 0, st, vel_env, dt_B, s->ures, s->velocity);  // CopyVelocityResultToBlocks
    });
    FG_HIP_CHECK(hipGetLastError());
    if (stats_host) for (int k = 0; k < 4; ++k) stats_host[k] = its[k];
    s->ctr.piso_steps += 1;
    for (int b = 0; b < B; ++b) s->env_status[b] = 0;
    if (soft_rc == FG_ERR_NOT_FINITE) {   // rare path: which envs were dropped; their pressure goes back to what it was
        hipLaunchKernelGGL(k_mb_restore_failed, dim3((unsigned)((N + FG_BLOCK - 1) / FG_BLOCK), B), blk, 0, st, N, (const int32_t*)s->env_fail, (const mb_real*)s->pres_bak, s->pressure);
        FG_HIP_CHECK(hipMemcpyAsync(s->env_fail_pinned, s->env_fail, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
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
    return FG_OK;
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
    hipLaunchKernelGGL(k_ml_scale, dim3(s->B), dim3(1024), 0, st, (const mb_real*)s->Pdiag, s->N, s->ml_geom_diag_sum, s->ml_scale);
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

extern "C" int fg_mb_solver_hints(fg_mb_handle s, int32_t* hints36, int32_t set) {
    FG_REQUIRE(s != nullptr && hints36 != nullptr, FG_ERR_INVALID_ARG, "fg_mb_solver_hints: bad argument");
    int* trial[4] = {&s->ml_bicg_attempts, &s->ml_bicg_failures, &s->ml_bicg_skip, &s->ml_bicg_backoff};
    for (int k = 0; k < 32; ++k) { if (set) s->pred_bicg[k] = hints36[k]; else hints36[k] = s->pred_bicg[k]; }
    for (int k = 0; k < 4; ++k) { if (set) *trial[k] = hints36[32 + k]; else hints36[32 + k] = *trial[k]; }
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



extern "C" int fg_mb_max_velocity(fg_mb_handle s, mb_real* out_B_host, void* stream) {
    FG_REQUIRE(s && s->finalized && s->velocity && out_B_host, FG_ERR_NOT_BOUND, "fg_mb_max_velocity: fields not bound");
    hipStream_t st = (hipStream_t)stream;
    const int cells = std::max(s->N, s->NB);
    FG_HIP_CHECK(hipMemsetAsync(s->red, 0, sizeof(mb_real) * s->B, st));
    MB_DISPATCH(s, hipLaunchKernelGGL(k_mb_maxvel<DIMS>, dim3((cells + FG_BLOCK - 1) / FG_BLOCK, s->B), dim3(FG_BLOCK), 0, st,
                                      s->dev, s->velocity, s->bvel, s->red););
    FG_HIP_CHECK(hipMemcpyAsync(s->red_pinned, s->red, sizeof(mb_real) * s->B, hipMemcpyDeviceToHost, st));
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
    // flux-balance guard (simulation.py:221-229)
    {
        std::vector<mb_real> fl(B);
        if (int rc = fg_mb_boundary_flux_balance(s, fl.data(), stream)) return rc;
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
        if (o->adaptive)
            if (int rc = fg_mb_max_velocity(s, mv.data(), stream)) return rc;
        // _PISO_adaptive_step (PISOtorch_simulation.py:2004-2064): ts = t_rem / ceil(t_rem / (CFL / max_vel)), per env
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
            s->dt_pinned[b] = ts;
        }
        FG_HIP_CHECK(hipMemcpyAsync(s->dt_dev, s->dt_pinned, sizeof(mb_real) * B, hipMemcpyHostToDevice, st));
        if (o->outflow_count > 0)  // PRE hook of the cylinder / airfoil envs (cylinder_env_base.py:280-300)
            if (int rc = mb_outflow_pre(s, s->dt_dev, o->outflow_slot0, o->outflow_count, o->outflow_slot0_b, o->outflow_count_b,
                                        o->outflow_velm, o->outflow_tol, st))
                return rc;
        const int rc = fg_mb_piso_step(s, s->dt_dev, &o->step, stats, stream);
        if (rc == FG_ERR_NOT_CONVERGED || rc == FG_ERR_NOT_FINITE) {
            all_ok = 0;
            for (int b = 0; b < B; ++b) {
                status[b] = std::max(status[b], s->env_status[b]);
                // a non-finite solve: that env's substep was not committed; it sits out the rest of this step with its
                // state intact (Simulation.single_step returns False there, simulation.py:259-280), the others finish
                if (s->env_status[b] == 2) t_rem[b] = 0.0;
            }
        } else if (rc != FG_OK) return rc;
        FG_HIP_CHECK(hipStreamSynchronize(st));  // dt_pinned is rewritten next round
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
        hipLaunchKernelGGL(k_mb_pmatrix<DIMS>, gn, blk, 0, st, D, (const mb_real*)nullptr, s->rA, s->Pdiag, s->Poff, s->Poff4, s->oc_agg ? s->oc_cell_slot : nullptr, s->Poff4s, s->Pdiag_s); s->oc_matrix_stale = false;
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
                                      (const mb_real*)nullptr, s->rA, s->Pdiag, s->Poff, s->Poff4, s->oc_agg ? s->oc_cell_slot : nullptr, s->Poff4s, s->Pdiag_s); s->oc_matrix_stale = false;);
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
