// Two-kernel BiCGStab iteration on the 3-D advection-diffusion matrix as z-marching LDS-ring kernels (gfx950).
//
// Replaces bicgstabSolveGPU (bicgstab_solver_kernel.cu:63-411) on the matrix of PISO_build_matrix
// (PISO_multiblock_cuda_kernel.cu:3616-3880) for BASELINE config 4 (TCF 128 x 64 x 64 x 8; tolerance envs/tcf/tcf_env.py:491).
// Same recurrence, accumulators, verdicts and breakdown guards as k_bicgf_a / k_bicgf_b of fg_bicgstab.hip (the two-kernel form:
// rho_{i+1} = rw.s - omega rw.t); what changes is how a workgroup walks the grid.  The brick kernels give a workgroup 64 x 4 x 4
// cells and gather every neighbour of every field from memory -- in 3-D the two-kernel form then re-reads four fields across six
// faces (measured slower than five kernels).  Here a workgroup owns a 64 x 16 (or 128 x 8) tile and MARCHES
// over a z-chunk like fg_poisson3d.hip:
//   * the field the stencil is applied to (s = r - alpha v in kernel b; p_new = s - omega t + beta (p - omega v) in kernel a) is
//     formed ONCE per cell -- centre float4 + the thread's halo duty -- and committed to an LDS ring holding planes k-1 .. k+2 of
//     all nc components; x / y / z neighbours are LDS reads, so the "recompute at the neighbours" of the brick form costs only the
//     halo loads (2 rows + 2 columns per tile and plane, 2 planes per chunk);
//   * the (1 + 6) matrix fields of the env are loaded once per cell and applied to all nc right-hand sides;
//   * every load of the next step (plane k+2 of the vectors, plane k+1 of matrix and rw: up to 400 B per thread) is issued before
//     plane k is computed: two workgroups per CU (62 KB of LDS each, <= 256 VGPRs) keep > 100 KB per CU in flight, which is what
//     the 2 us HBM round trip needs at 20 GB/s per CU;
//   * one barrier per plane (four-slot ring), one workgroup reduction per launch.
// Algorithmic bytes per cell and system: a 40 + 28/nc, b 20 + 28/nc (as the 2-D two-kernel form).
#include <math.h>
#include <stdlib.h>

#include "fg_internal.h"
#include "fg_zmarch.h"
#include "fg_bicg.h"

namespace {

constexpr int NSLOT = 4;

template <int BXL>
__device__ __forceinline__ void z_halo_axpy(FgVec<4>& a, Halo& ha, float cb, const FgVec<4>& b, const Halo& hb) {   // a += cb * b
#pragma unroll
    for (int e = 0; e < 4; ++e) { a.v[e] += cb * b.v[e]; ha.y.v[e] += cb * hb.y.v[e]; }
    ha.x += cb * hb.x;
}

struct Row7 { FgVec<4> d, o[6]; };

// y = C X for the thread's four cells of plane `sc`, every neighbour from the ring (same summation order as fg_apply_nbr)
template <int BXL>
__device__ __forceinline__ void z_apply7(const float* __restrict__ tm, const float* __restrict__ tc, const float* __restrict__ tp,
                                         int cen, const Row7& m, FgVec<4>& xc, FgVec<4>& y) {
    constexpr int LP = ZT<BXL>::LP;
    const float4 C = *reinterpret_cast<const float4*>(tc + cen);
    const float xl = tc[cen - 1], xr = tc[cen + 4];
    const float4 Ym = *reinterpret_cast<const float4*>(tc + cen - LP);
    const float4 Yp = *reinterpret_cast<const float4*>(tc + cen + LP);
    const float4 Zm = *reinterpret_cast<const float4*>(tm + cen);
    const float4 Zp = *reinterpret_cast<const float4*>(tp + cen);
    const float cv[4] = {C.x, C.y, C.z, C.w};
    const float ym[4] = {Ym.x, Ym.y, Ym.z, Ym.w}, yp[4] = {Yp.x, Yp.y, Yp.z, Yp.w};
    const float zm[4] = {Zm.x, Zm.y, Zm.z, Zm.w}, zp[4] = {Zp.x, Zp.y, Zp.z, Zp.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xm_ = (e == 0) ? xl : cv[e > 0 ? e - 1 : 0];
        const float xp_ = (e == 3) ? xr : cv[e < 3 ? e + 1 : 3];
        float v = m.d.v[e] * cv[e] + m.o[0].v[e] * xm_ + m.o[1].v[e] * xp_ + m.o[2].v[e] * ym[e] + m.o[3].v[e] * yp[e];
        v += m.o[4].v[e] * zm[e] + m.o[5].v[e] * zp[e];
        y.v[e] = v;
        xc.v[e] = cv[e];
    }
}

// PATH 1: every component of the env takes the full update (mode 1, no restart) -- the lock-step case; PATH 2: every component is
// in its first launch (mode 3).  These bodies are free of branches on the per-system decisions, so the loads of a step are
// issued as one batch; PATH 0 is the general one (systems of an env in different states)
template <int BXL, int NC, int PATH>
__device__ __forceinline__ void b3_a_body(const FgGrid& g, const BicgPtrs& q, const BicgFused& w, int it, const ZCtx& c, const BicgDecA& D,
                                          float (*ring)[NC][ZT<BXL>::LROWS * ZT<BXL>::LP], float* red, int ZC) {
    constexpr int LP = ZT<BXL>::LP;
    const int e = it & 1, pe = e ^ 1;
    const bool fold = w.fold0 != 0;          // (uniform kernel argument)
    constexpr bool FAST = PATH != 0;
    auto m1 = [&](int comp) { return PATH == 1 || (PATH == 0 && D.mode[comp] == 1); };
    auto m3 = [&](int comp) { return PATH == 2 || (PATH == 0 && D.mode[comp] == 3); };
    auto rst = [&](int comp) { return PATH == 0 && D.restart[comp]; };

    const unsigned N4 = (unsigned)g.n * 4u, plane_b = (unsigned)(g.nx * g.ny) * 4u, vec_b = NC * N4;
    const size_t sysb = (size_t)c.b * NC * g.n;
    const rsrc_t R_s = z_rsrc(w.s + sysb, vec_b), R_t = z_rsrc(q.t + sysb, vec_b);
    // p of the previous iteration; p_0 of a folded start is the right-hand side itself.  In the first launch (mode 3) the same
    // descriptor is the source of p_0: rhs (folded) or p[0] as the init kernel left it
    const rsrc_t R_po = z_rsrc((fold && it <= 1 ? q.rhs : (it == 0 ? w.p[0] : w.p[pe])) + sysb, vec_b);
    const rsrc_t R_vo = z_rsrc(w.v[pe] + sysb, vec_b);
    const rsrc_t R_pn = z_rsrc(w.p[e] + sysb, vec_b), R_vn = z_rsrc(w.v[e] + sysb, vec_b);
    const rsrc_t R_x = z_rsrc(q.x + sysb, vec_b), R_r = z_rsrc(q.r + sysb, vec_b), R_rw = z_rsrc(q.rw + sysb, vec_b);
    const rsrc_t R_d = z_rsrc(q.diag + (size_t)c.b * g.n, N4), R_o = z_rsrc(q.off + (size_t)c.b * 6 * g.n, 6 * N4);
    const unsigned vo_c = (unsigned)c.row_c * 4u;
    const unsigned vo_hy = (unsigned)((c.ly == 0) ? c.row_ym : c.row_yp) * 4u;
    const unsigned vo_hx = (unsigned)((c.lx == 0) ? c.col_xm : c.col_xp) * 4u;
    const int cen = (c.ly + 1) * LP + 4 + c.lx * 4;

    float part[2 * NC];   // [comp][rw.v | r.r]
#pragma unroll
    for (int k = 0; k < 2 * NC; ++k) part[k] = 0.f;

    if constexpr (!FAST) {
        // converged on s (bicgstab_solver_kernel.cu:305-329): x += alpha p, nothing else -- a plain stream over the chunk
#pragma unroll
        for (int comp = 0; comp < NC; ++comp) {
            if (D.mode[comp] != 2) continue;
            for (int k = c.k0; k < c.k1; ++k) {
                const unsigned so = (unsigned)k * plane_b + comp * N4;
                FgVec<4> x = z_bload4(R_x, vo_c, so);
                const FgVec<4> p = z_bload4(R_po, vo_c, so);
#pragma unroll
                for (int j = 0; j < 4; ++j) x.v[j] += D.alpha[comp] * p.v[j];
                z_bstore4(R_x, vo_c, so, x);
            }
        }
        bool march = false;
#pragma unroll
        for (int comp = 0; comp < NC; ++comp) march = march || D.mode[comp] == 1 || D.mode[comp] == 3;
        if (!march) return;
    }

    struct St { FgVec<4> s, t, p, v, x; Halo hs, ht, hp, hv; };
    auto stage = [&](int k, int comp, bool own) -> St {
        St r = {};
        const unsigned so = (unsigned)z_plane(g, k) * plane_b + comp * N4;
        if (m3(comp)) {            // first iteration: p_0
            r.p = z_bload4(R_po, vo_c, so);
            r.hp = z_load_halo<BXL>(c, R_po, vo_hy, vo_hx, so);
        } else if (m1(comp)) {
            r.s = z_bload4(R_s, vo_c, so); r.hs = z_load_halo<BXL>(c, R_s, vo_hy, vo_hx, so);
            r.t = z_bload4(R_t, vo_c, so); r.ht = z_load_halo<BXL>(c, R_t, vo_hy, vo_hx, so);
            if (!rst(comp)) {
                r.p = z_bload4(R_po, vo_c, so); r.hp = z_load_halo<BXL>(c, R_po, vo_hy, vo_hx, so);
                r.v = z_bload4(R_vo, vo_c, so); r.hv = z_load_halo<BXL>(c, R_vo, vo_hy, vo_hx, so);
            } else if (own) {
                r.p = z_bload4(R_po, vo_c, so);     // the x update still needs p_old at the centre
            }
            if (own) r.x = z_bload4(R_x, vo_c, so);
        }
        return r;
    };
    // forms p_it of the plane (centre + halo duty), commits it to the ring; the cell's own updates (x, r, p; r.r) ride along
    auto commit = [&](int slot, int k, int comp, bool own, St& r) {
        const unsigned so = (unsigned)k * plane_b + comp * N4;           // (own planes are never wrapped)
        if (m1(comp)) {
            const float al = D.alpha[comp], om = D.omega[comp], be = D.beta[comp];
            if (own) {
#pragma unroll
                for (int j = 0; j < 4; ++j) r.x.v[j] += al * r.p.v[j] + om * r.s.v[j];
            }
            z_halo_axpy<BXL>(r.s, r.hs, -om, r.t, r.ht);                     // r_it = s - omega t
            if (own) {
                z_bstore4(R_x, vo_c, so, r.x);
                z_bstore4(R_r, vo_c, so, r.s);
#pragma unroll
                for (int j = 0; j < 4; ++j) part[2 * comp + 1] += r.s.v[j] * r.s.v[j];
            }
            if (rst(comp)) {          // rw = p = r
                r.p = r.s; r.hp = r.hs;
                if (own) z_bstore4(R_rw, vo_c, so, r.s);
            } else {                        // p_it = r + beta (p - omega v)
                z_halo_axpy<BXL>(r.p, r.hp, -om, r.v, r.hv);
#pragma unroll
                for (int j = 0; j < 4; ++j) { r.p.v[j] = r.s.v[j] + be * r.p.v[j]; r.hp.y.v[j] = r.hs.y.v[j] + be * r.hp.y.v[j]; }
                r.hp.x = r.hs.x + be * r.hp.x;
            }
            if (own) z_bstore4(R_pn, vo_c, so, r.p);
        } else if (m3(comp) && fold && own) {   // folded start: rw = r_0 = rhs, x_0 = 0, r.r
            z_bstore4(R_rw, vo_c, so, r.p);
            FgVec<4> z4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { z4.v[j] = 0.f; part[2 * comp + 1] += r.p.v[j] * r.p.v[j]; }
            z_bstore4(R_x, vo_c, so, z4);
        }
        if (m1(comp) || m3(comp)) z_fill_tile<BXL>(ring[slot][comp], c, r.p, r.hp);
    };
    // matrix rows and rw of ONE plane: requested right after the previous plane has consumed them (they land while the commit
    // of the staged plane waits for its own, older loads) -- a second set of 40 registers for a full step of prefetch spilled
    struct Mat { Row7 m; FgVec<4> rw[NC]; };
    auto load_mat = [&](int k) -> Mat {
        Mat r = {};
        const unsigned so = (unsigned)k * plane_b;
        r.m.d = z_bload4(R_d, vo_c, so);
#pragma unroll
        for (int f = 0; f < 6; ++f) r.m.o[f] = z_bload4(R_o, vo_c, so + f * N4);
#pragma unroll
        for (int comp = 0; comp < NC; ++comp)
            if (m1(comp) || (m3(comp) && !fold)) r.rw[comp] = z_bload4(R_rw, vo_c, so + comp * N4);
        return r;
    };
    auto plane = [&](int k, int sm, int sc, int sp, const Mat& M) {
#pragma unroll
        for (int comp = 0; comp < NC; ++comp) {
            if (!m1(comp) && !m3(comp)) continue;
            FgVec<4> pc, y;
            z_apply7<BXL>(ring[sm][comp], ring[sc][comp], ring[sp][comp], cen, M.m, pc, y);
            z_bstore4(R_vn, vo_c, (unsigned)k * plane_b + comp * N4, y);
            // rw = p where this launch (re)defines rw: after a breakdown restart and in a folded first launch (the store of this
            // launch may not be visible to a load of the same launch)
            const bool rs = (m1(comp) && rst(comp)) || (m3(comp) && fold);
#pragma unroll
            for (int j = 0; j < 4; ++j) part[2 * comp] += (rs ? pc.v[j] : M.rw[comp].v[j]) * y.v[j];
        }
    };

    // Odd z-chunks march DOWNWARD: a chunk and its neighbour then touch the planes they share (each other's z halo) at the same
    // time -- both start or both end there -- so the second reader finds them in the XCD's L2 instead of HBM (a chunk's halo
    // planes are 2 / ZC of its vector loads).  Step i works on plane kk(i) = k0 + i (up) or k1 - 1 - i (down); "behind" / "ahead"
    // are the planes kk - dir / kk + dir, and the stencil's -z / +z neighbours swap slots with the direction.
    const int dir = ((c.k0 / ZC) & 1) ? -1 : 1;
    const int nplanes = c.k1 - c.k0;
    auto kk = [&](int i) { return dir > 0 ? c.k0 + i : c.k1 - 1 - i; };     // i = -1 and i = nplanes are the halo planes
    // prologue: steps -1, 0, 1 -> slots 0, 1, 2 as a pipeline over the 3 NC (plane, component) items with three items in
    // flight (a whole plane ahead of the one being committed would need 240 registers)
    {
        constexpr int NI = 3 * NC, DEPTH = NC >= 3 ? 3 : NC;
        St pipe[DEPTH];
        auto item_stage = [&](int i) -> St {
            const int qd = i / NC, comp = i % NC;
            return stage(kk(qd - 1), comp, (qd >= 1) && (qd - 1 < nplanes));
        };
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) pipe[i] = item_stage(i);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int qd = i / NC, comp = i % NC;
            __builtin_amdgcn_sched_barrier(0);
            commit(qd, kk(qd - 1), comp, (qd >= 1) && (qd - 1 < nplanes), pipe[i % DEPTH]);
            __builtin_amdgcn_sched_barrier(0);
            if (i + DEPTH < NI) pipe[i % DEPTH] = item_stage(i + DEPTH);
        }
    }
    Mat M = load_mat(kk(0));
    __syncthreads();
    int sm = 0, sc = 1, sp = 2, sf = 3;
#pragma unroll 1
    for (int i = 0; i < nplanes; ++i) {
        const bool more = (i + 1 < nplanes);
        const bool own2 = (i + 2 < nplanes);
        St nxt[NC];
        if (more) {
#pragma unroll
            for (int comp = 0; comp < NC; ++comp) nxt[comp] = stage(kk(i + 2), comp, own2);
        }
        __builtin_amdgcn_sched_barrier(0);
        plane(kk(i), dir > 0 ? sm : sp, sc, dir > 0 ? sp : sm, M);
        if (!more) break;
        __builtin_amdgcn_sched_barrier(0);
        M = load_mat(kk(i + 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int comp = 0; comp < NC; ++comp) commit(sf, kk(i + 2), comp, own2, nxt[comp]);   // the free slot: last read two steps ago
        __syncthreads();                                                                     // the plane ahead is visible; the plane behind retired
        const int t3 = sm; sm = sc; sc = sp; sp = sf; sf = t3;
    }
    const float tot = fg_block_sum_lanes<2 * NC>(part, red);    // thread t < 2 NC holds value t: [comp][rw.v | r.r]
    if (threadIdx.x < 2 * NC) {
        const int comp = threadIdx.x >> 1, kind = threadIdx.x & 1;
        bool on = false;
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (k == comp) on = kind == 0 ? (m1(k) || m3(k)) : (m1(k) || (m3(k) && fold));
        if (on) acc_add(q.acc + (size_t)(c.b * NC + comp) * FG_ACC_DOUBLES + ((kind ? F_RR : F_RV) + e), (double)tot);
    }
}

// --------------------------------------------------------------------------------------------------------------------------
// k_bicg3_a(it): finish iteration it - 1 (x, r, p) and start iteration it (v = C p, rw.v, r.r)
// --------------------------------------------------------------------------------------------------------------------------
template <int BXL, int NC>
__global__ __launch_bounds__(FG_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_bicg3_a(FgGrid g, BicgPtrs q, BicgFused w, int it,
                                                                                                  int tiles_x, int tiles_y, int zchunks, int ZC) {
    constexpr int LP = ZT<BXL>::LP, LROWS = ZT<BXL>::LROWS;
    const ZCtx c = z_make_ctx<BXL>(g, tiles_x, tiles_y, zchunks, ZC);
    const unsigned per_env = tiles_x * tiles_y * zchunks;
    const unsigned tile_id = fg_xcd_remap(blockIdx.x, gridDim.x) % per_env;
    const bool leader = (threadIdx.x == 0) && (tile_id == 0);
    const BicgDecA D = fg_bicgf_decide_a(g, q, c.b, it, leader, w.fold0 != 0);
    if (!D.any) return;
    __shared__ __attribute__((aligned(16))) float ring[NSLOT][NC][LROWS * LP];
    __shared__ float red[2 * NC * 4];
    bool all1 = true, all3 = true;
#pragma unroll
    for (int comp = 0; comp < NC; ++comp) {
        all1 = all1 && D.mode[comp] == 1 && !D.restart[comp];
        all3 = all3 && D.mode[comp] == 3;
    }
    if (all1) b3_a_body<BXL, NC, 1>(g, q, w, it, c, D, ring, red, ZC);
    else if (all3) b3_a_body<BXL, NC, 2>(g, q, w, it, c, D, ring, red, ZC);
    else b3_a_body<BXL, NC, 0>(g, q, w, it, c, D, ring, red, ZC);
}

// --------------------------------------------------------------------------------------------------------------------------
// k_bicg3_b(it): convergence test on r_it, alpha, s = r - alpha v, t = C s, the five dot products
// --------------------------------------------------------------------------------------------------------------------------
template <int BXL, int NC, bool FAST>
__device__ __forceinline__ void b3_b_body(const FgGrid& g, const BicgPtrs& q, const BicgFused& w, int it, const ZCtx& c, const BicgDecB& D,
                                          float (*ring)[NC][ZT<BXL>::LROWS * ZT<BXL>::LP], float* red, int ZC) {
    constexpr int LP = ZT<BXL>::LP;
    const int e = it & 1;
    auto wk = [&](int comp) { return FAST || D.work[comp]; };

    const unsigned N4 = (unsigned)g.n * 4u, plane_b = (unsigned)(g.nx * g.ny) * 4u, vec_b = NC * N4;
    const size_t sysb = (size_t)c.b * NC * g.n;
    const rsrc_t R_r = z_rsrc((w.fold0 && it == 0 ? q.rhs : q.r) + sysb, vec_b);      // folded start: r_0 is the right-hand side
    const rsrc_t R_v = z_rsrc(w.v[e] + sysb, vec_b), R_rw = z_rsrc(q.rw + sysb, vec_b);
    const rsrc_t R_s = z_rsrc(w.s + sysb, vec_b), R_t = z_rsrc(q.t + sysb, vec_b);
    const rsrc_t R_d = z_rsrc(q.diag + (size_t)c.b * g.n, N4), R_o = z_rsrc(q.off + (size_t)c.b * 6 * g.n, 6 * N4);
    const unsigned vo_c = (unsigned)c.row_c * 4u;
    const unsigned vo_hy = (unsigned)((c.ly == 0) ? c.row_ym : c.row_yp) * 4u;
    const unsigned vo_hx = (unsigned)((c.lx == 0) ? c.col_xm : c.col_xp) * 4u;
    const int cen = (c.ly + 1) * LP + 4 + c.lx * 4;

    float part[5 * NC];   // [comp][s.s | t.s | t.t | rw.s | rw.t]
#pragma unroll
    for (int k = 0; k < 5 * NC; ++k) part[k] = 0.f;

    struct St { FgVec<4> r, v; Halo hr, hv; };
    auto stage = [&](int k, int comp) -> St {
        St r = {};
        const unsigned so = (unsigned)z_plane(g, k) * plane_b + comp * N4;
        if (wk(comp)) {
            r.r = z_bload4(R_r, vo_c, so); r.hr = z_load_halo<BXL>(c, R_r, vo_hy, vo_hx, so);
            r.v = z_bload4(R_v, vo_c, so); r.hv = z_load_halo<BXL>(c, R_v, vo_hy, vo_hx, so);
        }
        return r;
    };
    auto commit = [&](int slot, int comp, St& r) {
        if (!wk(comp)) return;
        z_halo_axpy<BXL>(r.r, r.hr, -D.alpha[comp], r.v, r.hv);       // s = r - alpha v
        z_fill_tile<BXL>(ring[slot][comp], c, r.r, r.hr);
    };
    struct Mat { Row7 m; FgVec<4> rw[NC]; };
    auto load_mat = [&](int k) -> Mat {
        Mat r = {};
        const unsigned so = (unsigned)k * plane_b;
        r.m.d = z_bload4(R_d, vo_c, so);
#pragma unroll
        for (int f = 0; f < 6; ++f) r.m.o[f] = z_bload4(R_o, vo_c, so + f * N4);
#pragma unroll
        for (int comp = 0; comp < NC; ++comp)
            if (wk(comp)) r.rw[comp] = z_bload4(R_rw, vo_c, so + comp * N4);
        return r;
    };
    auto plane = [&](int k, int sm, int sc, int sp, const Mat& M) {
#pragma unroll
        for (int comp = 0; comp < NC; ++comp) {
            if (!wk(comp)) continue;
            FgVec<4> sv, t;
            z_apply7<BXL>(ring[sm][comp], ring[sc][comp], ring[sp][comp], cen, M.m, sv, t);
            const unsigned so = (unsigned)k * plane_b + comp * N4;
            z_bstore4(R_s, vo_c, so, sv);
            z_bstore4(R_t, vo_c, so, t);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part[5 * comp + 0] += sv.v[j] * sv.v[j];
                part[5 * comp + 1] += t.v[j] * sv.v[j];
                part[5 * comp + 2] += t.v[j] * t.v[j];
                part[5 * comp + 3] += M.rw[comp].v[j] * sv.v[j];
                part[5 * comp + 4] += M.rw[comp].v[j] * t.v[j];
            }
        }
    };

    // (odd chunks march downward: see kernel a)
    const int dir = ((c.k0 / ZC) & 1) ? -1 : 1;
    const int nplanes = c.k1 - c.k0;
    auto kk = [&](int i) { return dir > 0 ? c.k0 + i : c.k1 - 1 - i; };
    {   // prologue: steps -1, 0, 1 -> slots 0, 1, 2, all requested before the first commit
        St s0[3][NC];
#pragma unroll
        for (int qd = 0; qd < 3; ++qd)
#pragma unroll
            for (int comp = 0; comp < NC; ++comp) s0[qd][comp] = stage(kk(qd - 1), comp);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qd = 0; qd < 3; ++qd)
#pragma unroll
            for (int comp = 0; comp < NC; ++comp) commit(qd, comp, s0[qd][comp]);
    }
    Mat mcur = load_mat(kk(0));
    __syncthreads();
    int sm = 0, sc = 1, sp = 2, sf = 3;
#pragma unroll 1
    for (int i = 0; i < nplanes; ++i) {
        const bool more = (i + 1 < nplanes);
        St nxt[NC];
        Mat mnxt;      // (a full step of prefetch for matrix and rw: this kernel has the registers for it)
        if (more) {
#pragma unroll
            for (int comp = 0; comp < NC; ++comp) nxt[comp] = stage(kk(i + 2), comp);
            mnxt = load_mat(kk(i + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        plane(kk(i), dir > 0 ? sm : sp, sc, dir > 0 ? sp : sm, mcur);
        if (!more) break;
#pragma unroll
        for (int comp = 0; comp < NC; ++comp) commit(sf, comp, nxt[comp]);
        __syncthreads();
        const int t3 = sm; sm = sc; sc = sp; sp = sf; sf = t3;
        mcur = mnxt;
    }
    const float tot = fg_block_sum_lanes<5 * NC>(part, red);    // thread t < 5 NC holds value t: [comp][s.s | t.s | t.t | rw.s | rw.t]
    if (threadIdx.x < 5 * NC) {
        const int comp = threadIdx.x / 5, kind = threadIdx.x - 5 * comp;
        bool on = false;
#pragma unroll
        for (int k = 0; k < NC; ++k)
            if (k == comp) on = wk(k);
        if (on) acc_add(q.acc + (size_t)(c.b * NC + comp) * FG_ACC_DOUBLES + (F_SS + 2 * kind + e), (double)tot);
    }
}

template <int BXL, int NC>
__global__ __launch_bounds__(FG_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_bicg3_b(FgGrid g, BicgPtrs q, BicgFused w, int it,
                                                                                                  int tiles_x, int tiles_y, int zchunks, int ZC) {
    constexpr int LP = ZT<BXL>::LP, LROWS = ZT<BXL>::LROWS;
    const ZCtx c = z_make_ctx<BXL>(g, tiles_x, tiles_y, zchunks, ZC);
    const unsigned per_env = tiles_x * tiles_y * zchunks;
    const unsigned tile_id = fg_xcd_remap(blockIdx.x, gridDim.x) % per_env;
    const bool leader = (threadIdx.x == 0) && (tile_id == 0);
    const BicgDecB D = fg_bicgf_decide_b(g, q, c.b, it, leader);
    if (!D.any) return;
    __shared__ __attribute__((aligned(16))) float ring[NSLOT][NC][LROWS * LP];
    __shared__ float red[5 * NC * 4];
    bool fast = true;
#pragma unroll
    for (int comp = 0; comp < NC; ++comp) fast = fast && D.work[comp];
    if (fast) b3_b_body<BXL, NC, true>(g, q, w, it, c, D, ring, red, ZC);
    else b3_b_body<BXL, NC, false>(g, q, w, it, c, D, ring, red, ZC);
}

}  // namespace

// geometry -----------------------------------------------------------------------------------------------------------------
static int b3_pick_bxl(const fg_state* s) {
    const FgGrid& g = s->grid;
    const int forced = s->bicg3_bxl;
    if (forced == 16 && g.nx % 64 == 0 && g.ny % 16 == 0) return 16;
    if (forced == 32 && g.nx % 128 == 0 && g.ny % 8 == 0) return 32;
    // 128 x 8 tiles where they fit: whole 512-byte rows per wave access -- measured on TCF 128 x 64 x 64 x 8 (profiles/micro_bicg3d.py):
    // 0.935 ms per four-iteration solve against 0.979 with 64 x 16 tiles, although those have the smaller x/y halo
    if (g.nx % 128 == 0 && g.ny % 8 == 0) return 32;
    if (g.nx % 64 == 0 && g.ny % 16 == 0) return 16;
    return 0;
}

bool fg_bicg3_ok(const fg_state* s, int nc, int* zc_out) {
    const FgGrid& g = s->grid;
    // FG_BICG3 (read at fg_create): 0 = never (brick kernels), > 0 = always with that chunk length, unset = the rule below
    const int force = s->bicg3_force;
    if (force == 0 || g.dims != 3 || s->vec != 4 || (nc != 1 && nc != 3) || g.nz < 4) return false;
    const int bxl = b3_pick_bxl(s);
    if (bxl == 0) return false;     // tiles must coincide with the grid: every thread valid
    const long tiles = (long)(g.nx / (bxl * 4)) * (g.ny / (FG_BLOCK / bxl));
    if (force > 0) { *zc_out = force < g.nz ? force : g.nz; return true; }
    // two workgroups per CU is what LDS allows; chunks as long as that still fills the chip, at least 4 planes
    int cus = 256;
    {
        static const int dev_cus = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
            return n > 0 ? n : 256;
        }();
        cus = dev_cus;
    }
    int zc = 32;
    while (zc > 4 && tiles * ((g.nz + zc - 1) / zc) * g.B < 2L * cus) zc /= 2;
    if (tiles * ((g.nz + zc - 1) / zc) * g.B < cus) return false;      // a grid this small stays on the brick kernels
    *zc_out = zc;
    return true;
}

template <int WHICH>
static int b3_launch(const fg_state* s, const BicgPtrs& q, const BicgFused& w, int it, int zc, int slot, hipStream_t st) {
    const FgGrid& g = s->grid;
    const int bxl = b3_pick_bxl(s);
    const int tx = g.nx / (bxl * 4), ty = g.ny / (FG_BLOCK / bxl), zch = (g.nz + zc - 1) / zc;
    const dim3 grid((unsigned)(tx * ty * zch * g.B));
#define B3_GO(BXL, NC)                                                                                                            \
    do {                                                                                                                          \
        if (WHICH == 0) FG_LAUNCH_P(s, slot, (k_bicg3_a<BXL, NC>), grid, dim3(FG_BLOCK), 0, st, g, q, w, it, tx, ty, zch, zc);     \
        else FG_LAUNCH_P(s, slot, (k_bicg3_b<BXL, NC>), grid, dim3(FG_BLOCK), 0, st, g, q, w, it, tx, ty, zch, zc);                \
    } while (0)
    if (bxl == 16) { if (q.nc == 3) B3_GO(16, 3); else B3_GO(16, 1); }
    else { if (q.nc == 3) B3_GO(32, 3); else B3_GO(32, 1); }
#undef B3_GO
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_bicg3_launch_a(const fg_state* s, const BicgPtrs& q, const BicgFused& w, int it, int zc, int slot, hipStream_t st) {
    return b3_launch<0>(s, q, w, it, zc, slot, st);
}
int fg_bicg3_launch_b(const fg_state* s, const BicgPtrs& q, const BicgFused& w, int it, int zc, int slot, hipStream_t st) {
    return b3_launch<1>(s, q, w, it, zc, slot, st);
}
