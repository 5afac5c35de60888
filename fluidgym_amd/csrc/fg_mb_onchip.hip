// Multi-block path: the pressure CG of one env as ONE workgroup that runs the whole solve on chip (k_mbc_onchip) and its host
// launcher.  Split out of fg_mb_step.hip in round 4 (that file was one 3 956-line translation unit); replaces cgSolveGPU
// (cg_solver_kernel.cu:129-471) on the pressure matrix of PISO_multiblock_cuda_kernel.cu:4812-4978 for meshes of up to 28 k cells
// (the cylinder family, tolerance cylinder_env_base.py:315).  gfx950 / wave64 only.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "fg_mb.h"
#include "fg_mb_solve.h"

namespace {

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
    int32_t* its_host;          // host needs one wait after the launch and no device-to-host copies
    FgPollOut poll;             // sequence word per env behind the mirrors (fg_internal.h FgPoll)
    mb_real* rt_scratch;          // [B][N] r - mean r of the preconditioner pass (RTG instances: meshes beyond the LDS budget)
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

// RTG (cell-ordered preconditioned form, meshes of 16-24 k cells: the cylinder's `medium` / `hard` ids): the copy of r - mean r
// that the 4 x 4 aggregate sums gather lives in a per-env global scratch vector (L2) instead of LDS -- p alone is 96 KB there.
template <int DIMS, int CPT, int PM, bool DG_REGS, int NT, bool NBR = true, bool RING = true, bool PRE = false, bool AGG = false, bool RTG = false>
__global__ __launch_bounds__(NT) void k_mbc_onchip(MbDev D, MbSolve q, OcParams o) {
    static_assert(!RTG || (PRE && !AGG), "global residual copy: cell-ordered preconditioned form only");
    static_assert(!AGG || (PRE && DIMS == 2 && CPT == 16 && NT == 1024 && PM != 2 && !DG_REGS && !NBR), "aggregate-owned layout: 16 slots x 1024 threads, preconditioned, 2-D");
    __shared__ mb_real v_lds[CPT * NT];
    __shared__ double red[3][2][OC_MAX_WAVES];
    __shared__ mb_real l_r4[(PRE && !AGG) ? OC_N4 : 1], l_r8[PRE ? OC_N8 : 1];
    // r - mean r of the preconditioner pass, where the aggregate sums gather it; once they have, the same memory holds the
    // per-wave partial sums of the coarse solve
    constexpr int LP8 = AGG ? 256 : OC_N8;   // AGG: 4 n8 <= 1024 threads
    constexpr int RT = PRE ? ((!AGG && !RTG && CPT * NT > OC_MAX_WAVES * LP8) ? CPT * NT : OC_MAX_WAVES * LP8) : 1;
    __shared__ __attribute__((aligned(16))) mb_real l_rt[RT];
    mb_real (*l_part)[LP8] = reinterpret_cast<mb_real (*)[LP8]>(l_rt);
    // AGG: M p (and, before the stencil pass, z) lives in LDS instead of 16 registers per thread -- the cell-ordered preconditioned
    // instance spills ~100 registers, and what that costs is the stencil pass: its 48 loads per thread no longer overlap
    // (knock-out builds, -DFG_MB_OC_KNOCK: 13 of 28 us per iteration are those loads, against 11.5 us for the whole plain iteration)
    __shared__ mb_real ap_lds[AGG ? CPT * NT : 1];
#define OC_AP(k, i) (*(AGG ? &ap_lds[i] : &ap[k]))
    static_assert(!AGG || NT == 1024, "aggregate-owned coarse solve: sixteen waves");   // (the cell-ordered form gives every one of its NT / 64 waves its own set of columns)
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
            fg_poll_publish(o.poll, sys);
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
                mb_real* __restrict__ rt_g = RTG ? o.rt_scratch + vb : nullptr;
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const unsigned i = tl + (unsigned)k * NT;
                    if (i < (unsigned)N) { if constexpr (RTG) rt_g[i] = r[k] - rm; else l_rt[i] = r[k] - rm; }
                }
                __syncthreads();   // (RTG: the waves of a workgroup share the CU's L1, which the stores above went through)
                OC_PHASE(1);   // residual copy to LDS
                for (int a = t; a < n4; a += NT) {
                    const uint2 rc = o.pre.rect4[a];
                    const unsigned w = rc.y & 0xffu, h = (rc.y >> 8) & 0xffu, stride = rc.y >> 16;
                    mb_real sum = 0.f;
                    if constexpr (RTG) {
                        // all (up to 16) loads of the rectangle requested at once: clamped addresses, zero weight outside it --
                        // same order of the sum as the loop of the LDS form
                        mb_real val[4][4];
#pragma unroll
                        for (unsigned dy = 0; dy < 4; ++dy)
#pragma unroll
                            for (unsigned dx = 0; dx < 4; ++dx)
                                val[dy][dx] = rt_g[rc.x + (dy < h ? dy : h - 1) * stride + (dx < w ? dx : w - 1)];
#pragma unroll
                        for (unsigned dy = 0; dy < 4; ++dy)
#pragma unroll
                            for (unsigned dx = 0; dx < 4; ++dx) sum += (dy < h && dx < w) ? val[dy][dx] : 0.f;
                    } else {
                        for (unsigned dy = 0; dy < h; ++dy)
                            for (unsigned dx = 0; dx < w; ++dx) sum += l_rt[rc.x + dy * stride + dx];
                    }
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
                        for (int c = grp; c < n8; c += (NT / 64)) {
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
                    for (int g = 0; g < (NT / 64); ++g) e += l_part[g][row];
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
        fg_poll_publish(o.poll, sys);      // (after the mirrors: the host spins on this word instead of synchronising the stream)
    }
}


// ---------------------------------------------------------------------------------------------------------------
// The same preconditioned solve for meshes of 16-24 k cells (the cylinder's `medium` / `hard` ids, 23 k cells): at that size p alone
// takes 96 of the 160 KB of LDS and the register-resident form above spills a third of its state (68 us per iteration, measured).
// Here only the search direction p lives in the CU (LDS, where the stencil gathers hit it); x, r and M p / z are per-env vectors in
// global memory that never leave L2 (64 envs x 3 x 92 KB), always read and written by the thread that owns the cell (cell i = thread
// + k 1024, members in batches of four so that a batch's loads are in flight together), so the loops need no register arrays, and
// the 4 x 4 aggregate sums gather r straight from that vector (sum(r - mean) = sum(r) - mean * cells: no residual copy).  Per
// iteration and cell ~74 B cross the CU's L1 (stencil 32, update 20, z pass 14, direction 4, aggregate sums 4).  Recurrence, projection,
// restart / best-iterate / stall rules and preconditioner are those of k_mbc_onchip<PRE, !AGG>, statement by statement.
// ---------------------------------------------------------------------------------------------------------------
constexpr int OC_L2_CELLS = 24 * 1024;
#ifndef OC_L2_G
#define OC_L2_G 4      // members per batch of loads in flight together (-DOC_L2_G=8: A/B builds)
#endif
template <int PM>
__global__ __launch_bounds__(1024) void k_mbc_l2(MbDev D, MbSolve q, OcParams o) {
    static_assert(PM != 2, "mean projection or none");
    constexpr int NT = 1024, G = OC_L2_G, NW = NT / 64;
    __shared__ mb_real v_lds[OC_L2_CELLS];
    __shared__ double red[3][2][OC_MAX_WAVES];
    __shared__ mb_real l_r4[OC_N4], l_r8[OC_N8];
    __shared__ __attribute__((aligned(16))) mb_real l_rt[NW * OC_N8];
    mb_real (*l_part)[OC_N8] = reinterpret_cast<mb_real (*)[OC_N8]>(l_rt);
    int phase = 0;
    const int sys = blockIdx.x, N = D.N, t = threadIdx.x;
    const unsigned un = (unsigned)N, ut = (unsigned)t;
    const size_t vb = (size_t)sys * N;
    if (!mb_active(o.dt, sys)) {
        if (t == 0) {
            flag_st(q.flags + (sys), 3);
            q.info[sys].final_residual = 0.f; q.info[sys].used_iterations = -1; q.info[sys].converged = 1; q.info[sys].is_finite = 1;
            o.info_host[sys] = q.info[sys];
            o.its_host[sys] = 0;
            fg_poll_publish(o.poll, sys);
        }
        return;
    }
    const mb_real rsqn = mb_rsqrt((mb_real)N);
    const mb_real* __restrict__ rhs = q.rhs + vb;
    const mb_real* __restrict__ diag = q.diag + vb;
    const float4* __restrict__ off4 = reinterpret_cast<const float4*>(o.off4) + (size_t)sys * N;
    const uint2* __restrict__ nb2 = reinterpret_cast<const uint2*>(o.nbr16);
    mb_real* xg = q.x + vb;       // (no __restrict__: the vectors are re-read after they were written)
    mb_real* rg = q.r + vb;
    mb_real* ag = q.v + vb;       // M p of the stencil pass; before it, z of the preconditioner pass
    mb_real* bestx = q.best_x + vb;
    const oc_rsrc R_dg = oc_make_rsrc(diag, un * 4u), R_off = oc_make_rsrc(off4, un * 16u), R_nb = oc_make_rsrc(nb2, un * 8u);
    // batches of G members: the loads of a batch are requested together (out-of-range members read cell 0 and contribute nothing)
#define L2_BATCHES(i0) for (unsigned i0 = ut; i0 < un; i0 += G * NT)
#define L2_MEMBERS(g, i, ok, i0) _Pragma("unroll") for (int g = 0; g < G; ++g) if (const unsigned i = (i0) + (unsigned)g * NT; true) if (const bool ok = i < un; true)
    // ---- start: x = x0 or 0, r = rhs (- M x0 through a residual pass)
    double rr = 0.0, sr = 0.0;
    mb_real inv_s = 1.f;
    {
        mb_real sd = 0.f, s2 = 0.f, s1 = 0.f;
        L2_BATCHES(i0) {
            mb_real rv[G], dv[G];
            L2_MEMBERS(g, i, ok, i0) { rv[g] = rhs[ok ? i : 0]; dv[g] = diag[ok ? i : 0]; }
            L2_MEMBERS(g, i, ok, i0) {
                if (ok) {
                    rg[i] = rv[g];
                    if (!o.use_x0) xg[i] = 0.f;
                    sd += dv[g]; s2 += rv[g] * rv[g]; s1 += rv[g] * rsqn;
                }
            }
        }
        double dsum, unused0;
        oc_reduce2<NT, false>(sd, 0.f, red, phase, dsum, unused0);
        inv_s = (mb_real)((double)o.pre.geom_diag_sum / dsum);
        if (!o.use_x0) {
            oc_reduce2<NT, false>(s2, s1, red, phase, rr, sr);
            if (PM == 0) sr = 0.0;
        }
    }
    double rz = 0.0, rz_prev = 1.0;
    int it = 0, best_it = 0, recoveries = 0, outcome = 0;   // outcome: 1 converged, 2 non-finite, 3 accepted on the kept iterate, 4 out of iterations / stalled
    mb_real best = 3.0e38f, crit = 0.f;
    bool fresh = true, restarted = true, residual_pass = o.use_x0 != 0, recovering = false;
    double rho = 0.0;
    const int n4 = o.pre.n4, n8 = o.pre.n8;
    const int ngrp = min(NW, NT / (((n8 + 3) & ~3) >> 2));    // column groups of the coarse solve (l_part holds NW rows)
    for (;;) {
        mb_real beta = 0.f, cy = 0.f;
        if (!residual_pass) {
            rho = rr - sr * sr;   // |r - (yp.r) yp|^2
            crit = mb_rms(rho, N);
            if (!(crit >= o.tol)) {
                if (isfinite(crit)) { outcome = 1; break; }
                // the recurrence broke down: back to the kept iterate
                if (recovering || recoveries >= 3 || it + o.check_every >= o.max_iterations) { outcome = 2; break; }
                ++recoveries;
                L2_BATCHES(i0) {
                    mb_real bv[G];
                    L2_MEMBERS(g, i, ok, i0) bv[g] = bestx[ok ? i : 0];
                    L2_MEMBERS(g, i, ok, i0) if (ok) xg[i] = isfinite(bv[g]) ? bv[g] : 0.f;
                }
                residual_pass = true; recovering = true;
            } else {
                recovering = false;
                // keep x_it when it beats the kept iterate by 2x (or at all inside the acceptance band): returnBestResult
                if (it == 0 || crit < 0.5f * best || (crit < o.accept_factor * o.tol && crit < best)) {
                    best = crit; best_it = it;
                    L2_BATCHES(i0) {
                        mb_real xv[G];
                        L2_MEMBERS(g, i, ok, i0) xv[g] = xg[ok ? i : 0];
                        L2_MEMBERS(g, i, ok, i0) if (ok) bestx[i] = xv[g];
                    }
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
        if (!residual_pass) {
            restarted = false;
            cy = (mb_real)sr;
            // ---- z = M (r - mean r): 4 x 4 aggregate sums gathered from r (every aggregate is a rectangle of cells of one block),
            // 8 x 8 sums of their children, dense coarse solve by the waves, corrections summed top-down into l_r4
            const mb_real rm = PM == 1 ? cy * rsqn : 0.f;
            for (int a = t; a < n4; a += NT) {
                const uint2 rc = o.pre.rect4[a];
                const unsigned w = rc.y & 0xffu, h = (rc.y >> 8) & 0xffu, stride = rc.y >> 16;
                mb_real val[4][4];
#pragma unroll
                for (unsigned dy = 0; dy < 4; ++dy)
#pragma unroll
                    for (unsigned dx = 0; dx < 4; ++dx) val[dy][dx] = rg[rc.x + (dy < h ? dy : h - 1) * stride + (dx < w ? dx : w - 1)];
                mb_real sum = 0.f;
#pragma unroll
                for (unsigned dy = 0; dy < 4; ++dy)
#pragma unroll
                    for (unsigned dx = 0; dx < 4; ++dx) sum += (dy < h && dx < w) ? val[dy][dx] - rm : 0.f;
                l_r4[a] = sum;
            }
            __syncthreads();
            for (int a = t; a < n8; a += NT) {
                const uint2 ch = o.pre.child8[a];
                const unsigned c0 = ch.x & 0xffffu, c1 = ch.x >> 16, c2 = ch.y & 0xffffu, c3 = ch.y >> 16;
                l_r8[a] = (c0 != 0xffffu ? l_r4[c0] : 0.f) + (c1 != 0xffffu ? l_r4[c1] : 0.f) + (c2 != 0xffffu ? l_r4[c2] : 0.f) +
                          (c3 != 0xffffu ? l_r4[c3] : 0.f);
            }
            __syncthreads();
            {   // e8 = A8^+ r8: wave g takes the columns c = g, g + 16, ..., lane q the four rows 4q .. 4q+3 (k_mbc_onchip)
                // (here thread = (row quad qd, column group grp) with ngrp = 1024 / quads groups, so that one round keeps nearly every
                //  thread busy: 90 quads x 11 groups of 33 columns at 360 aggregates, against two rounds of 23 with 26 of 64 lanes in the
                //  second -- per-phase cycle counts of a -DFG_MB_OC_CYCLES build: 30.0 k -> 21.5 k cycles per iteration)
                const int ld = (n8 + 3) & ~3, nq = ld >> 2;
                const int qd = t % nq, grp = t / nq;
                if (grp < ngrp) {
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 12
                    for (int c = grp; c < n8; c += ngrp) {
                        const float4 a = *reinterpret_cast<const float4*>(o.pre.aci8 + (unsigned)c * (unsigned)ld + 4u * (unsigned)qd);
                        const mb_real rc = l_r8[c];
                        acc.x += a.x * rc; acc.y += a.y * rc; acc.z += a.z * rc; acc.w += a.w * rc;
                    }
                    *reinterpret_cast<float4*>(&l_part[grp][4 * qd]) = acc;
                }
            }
            __syncthreads();
            for (int a = t; a < n4; a += NT) {
                const int row = o.pre.parent4[a];
                mb_real e = 0.f;
                for (int g = 0; g < ngrp; ++g) e += l_part[g][row];
                l_r4[a] = inv_s * (0.5f * l_r4[a] * o.pre.d4g[a] + e);   // d4g holds reciprocals
            }
            __syncthreads();
            mb_real s_rz = 0.f, s_z = 0.f;
            L2_BATCHES(i0) {
                mb_real rv[G], dv[G]; unsigned av[G];
                L2_MEMBERS(g, i, ok, i0) { rv[g] = rg[ok ? i : 0]; dv[g] = diag[ok ? i : 0]; av[g] = o.pre.a4[ok ? i : 0]; }
                L2_MEMBERS(g, i, ok, i0) {
                    if (ok) {
                        const mb_real rt = rv[g] - rm;
                        const mb_real z = rt * __builtin_amdgcn_rcpf(dv[g]) + l_r4[av[g]];   // v_rcp_f32: a preconditioner needs no IEEE division
                        ag[i] = z;
                        s_rz += rt * z; s_z += z;
                    }
                }
            }
            double zsum;
            oc_reduce2<NT, false>(s_rz, s_z, red, phase, rz, zsum);
            zbar = PM == 1 ? (mb_real)(zsum / (double)N) : 0.f;
            beta = fresh ? 0.f : (mb_real)(rz / rz_prev);
        }
        // ---- the vector the stencil is applied to: x, or p = (z - mean z) + beta p (every thread rewrites its own cells)
        L2_BATCHES(i0) {
            mb_real sv[G];
            L2_MEMBERS(g, i, ok, i0) sv[g] = residual_pass ? xg[ok ? i : 0] : ag[ok ? i : 0];
            L2_MEMBERS(g, i, ok, i0) {
                if (ok) {
                    mb_real v = sv[g];
                    if (!residual_pass) { v -= zbar; if (!fresh) v += beta * v_lds[i]; }
                    v_lds[i] = v;
                }
            }
        }
        __syncthreads();
        // ---- stencil pass: M p (or M x) of the thread's cells, p . M p
        mb_real part = 0.f, s2 = 0.f, s1 = 0.f;
        L2_BATCHES(i0) {
            // (buffer-resource loads for the three matrix streams: one descriptor each in SGPRs, one per-thread offset, the member
            //  offset as the scalar offset; out-of-range members read 0 -- stencil pass 20.3 k -> 16.2 k cycles per iteration)
            uint2 ub[G]; float4 cb[G]; mb_real dd[G], rh[G];
            L2_MEMBERS(g, i, ok, i0) {
                const unsigned so = (i0 - ut) + (unsigned)g * NT;      // member offset in cells (wave-uniform)
                const oc_u32x2 u2 = __builtin_amdgcn_raw_buffer_load_b64(R_nb, ut * 8u, so * 8u, 0);
                const oc_u32x4 c4 = __builtin_amdgcn_raw_buffer_load_b128(R_off, ut * 16u, so * 16u, 0);
                ub[g] = make_uint2(u2.x, u2.y);
                cb[g] = make_float4(__uint_as_float(c4.x), __uint_as_float(c4.y), __uint_as_float(c4.z), __uint_as_float(c4.w));
                dd[g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(R_dg, ut * 4u, so * 4u, 0));
                if (residual_pass) rh[g] = rhs[ok ? i : 0];
            }
            L2_MEMBERS(g, i, ok, i0) {
                if (ok) {
                    const uint32_t n0 = ub[g].x & 0xffffu, n1 = ub[g].x >> 16, n2 = ub[g].y & 0xffffu, n3 = ub[g].y >> 16;
                    // prescribed face (0xFFFF): no matrix entry; the gather reads the cell itself so that it stays in bounds
                    const mb_real vc = v_lds[i];
                    const mb_real v0 = v_lds[n0 != 0xffffu ? n0 : i], v1 = v_lds[n1 != 0xffffu ? n1 : i];
                    const mb_real v2 = v_lds[n2 != 0xffffu ? n2 : i], v3 = v_lds[n3 != 0xffffu ? n3 : i];
                    mb_real acc = dd[g] * vc;
                    acc += n0 != 0xffffu ? cb[g].x * v0 : 0.f;
                    acc += n1 != 0xffffu ? cb[g].y * v1 : 0.f;
                    acc += n2 != 0xffffu ? cb[g].z * v2 : 0.f;
                    acc += n3 != 0xffffu ? cb[g].w * v3 : 0.f;
                    if (residual_pass) {
                        const mb_real rnew = rh[g] - acc;
                        rg[i] = rnew;
                        s2 += rnew * rnew; s1 += rnew * rsqn;
                    } else {
                        ag[i] = acc;
                        part += vc * acc;
                    }
                }
            }
        }
        if (residual_pass) {
            oc_reduce2<NT, false>(s2, s1, red, phase, rr, sr);
            if (PM == 0) sr = 0.0;
            residual_pass = false; fresh = true; restarted = true;
            continue;
        }
        double pap, unused;
        oc_reduce2<NT, false>(part, 0.f, red, phase, pap, unused);
        const mb_real alpha = (mb_real)(rz / pap);
        L2_BATCHES(i0) {
            mb_real xv[G], rv[G], av[G];
            L2_MEMBERS(g, i, ok, i0) { xv[g] = xg[ok ? i : 0]; rv[g] = rg[ok ? i : 0]; av[g] = ag[ok ? i : 0]; }
            L2_MEMBERS(g, i, ok, i0) {
                if (ok) {
                    xg[i] = xv[g] + alpha * v_lds[i];
                    const mb_real rnew = rv[g] - alpha * av[g];
                    rg[i] = rnew;
                    s2 += rnew * rnew; s1 += rnew * rsqn;
                }
            }
        }
        oc_reduce2<NT, false>(s2, s1, red, phase, rr, sr);   // (its barriers also order this pass's stores of r before the aggregate gathers of the next)
        if (PM == 0) sr = 0.0;
        rz_prev = rz;
        fresh = false;
        ++it;
    }
    // ---- hand back: the last iterate when converged (it is in place), the kept one otherwise (k_mbs_restore_best)
    const bool use_best = (outcome == 3 || outcome == 4 || (outcome == 2 && best < 3.0e38f));
    if (use_best) {
        L2_BATCHES(i0) {
            mb_real bv[G];
            L2_MEMBERS(g, i, ok, i0) bv[g] = bestx[ok ? i : 0];
            L2_MEMBERS(g, i, ok, i0) if (ok) xg[i] = bv[g];
        }
    }
#undef L2_BATCHES
#undef L2_MEMBERS
    if (t == 0) {
        flag_st(q.flags + (sys), outcome == 2 ? 2 : (outcome == 3 ? 5 : 1));
        q.info[sys].final_residual = use_best ? best : crit;
        q.info[sys].used_iterations = use_best ? best_it : it;
        q.info[sys].converged = (outcome == 1 || outcome == 3) ? 1 : 0;
        q.info[sys].is_finite = outcome != 2 ? 1 : 0;
        q.best_it[sys] = it;   // total iterations run (profiling: the host sums them)
        o.info_host[sys] = q.info[sys];
        o.its_host[sys] = it;
        fg_poll_publish(o.poll, sys);
    }
}

}  // namespace

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
    o.poll = fg_poll_next(&s->poll);
    const bool ev = s->prof_on != 0;
    // Instances, chosen by measurement on the cylinder mesh (profiles/r02_onchip_variants.txt; 64 envs x 14 232 cells, us per
    // iteration): 16 cells per thread with the two-barrier reduction 11.7-11.8; the same with the one-barrier ring 14.5; 14
    // cells per thread 15.8-16.3; neighbour indices in registers 17.4; 512 threads x 256 registers 18.3 -- what the
    // compiler's schedule makes of each form decides, not the instruction count.  Small meshes keep the indices in registers.
    // LDS: p, r - mean r and the aggregate tables; 16-24 k cells: r - mean r in a global scratch vector (RTG instance)
    const bool pre = s->ml_on && s->ml_a4 != nullptr && n <= 24 * 1024 && s->ml_n4 <= OC_N4 && s->ml_n8 <= OC_N8;
    o.rt_scratch = nullptr;
    if (pre && n > 16 * 1024) {
        if (!s->oc_rt_scratch) { if (int rc = mb_alloc(s, &s->oc_rt_scratch, (size_t)s->B * s->N)) return rc; }
        o.rt_scratch = s->oc_rt_scratch;
    }
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
    else if (n <= OC_L2_CELLS && pre && o.off4 != nullptr && s->oc_rtg_nt != 1) {
        // 16-24 k cells: vectors in L2, p in LDS (k_mbc_l2; FG_MB_OC_RTG_NT=1 picks the register-resident form with its residual copy in global memory)
        if (ev) {
            if (pm_mode == 0) hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_mbc_l2<0>), dim3(nsys), dim3(1024), 0, st, s->prof_ev_oc[0], s->prof_ev_oc[1], 0, s->dev, q, o);
            else hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_mbc_l2<1>), dim3(nsys), dim3(1024), 0, st, s->prof_ev_oc[0], s->prof_ev_oc[1], 0, s->dev, q, o);
        } else {
            if (pm_mode == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mbc_l2<0>), dim3(nsys), dim3(1024), 0, st, s->dev, q, o);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mbc_l2<1>), dim3(nsys), dim3(1024), 0, st, s->dev, q, o);
        }
    }
    else if (n <= 24 * 1024 && pre) OC_LAUNCH_PM(24, false, 1024, false, false, true, false, true);
    else if (n <= 24 * 1024) OC_LAUNCH_PM(24, false, 1024, false, false, false);
    else OC_LAUNCH_PM(28, false, 1024, false, false, false);
#undef OC_LAUNCH_PRE
    // info_pinned / flags_pinned (iterations run) were written by the kernel; with profiling events the stream is synchronised
    if (int rc = fg_poll_wait(&s->poll, ev ? FgPollOut{nullptr, 0} : o.poll, 0, nsys, st)) return rc;
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
