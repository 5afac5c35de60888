// 3-D pressure-Poisson kernels with z-marching register planes + LDS x/y halo tiles (gfx950).
//
// The generic kernels of fg_poisson.hip give every workgroup a 64 x 4 x 4 brick; at 256^3 rocprofv3 PMC
// showed 27-45 % more bytes fetched than the algorithm needs (profiles/r01_b_poisson256_pmc.csv): with
// bricks 4 planes thick the z-halo planes are re-fetched after they have left the XCD's 4 MiB L2.
// Here a workgroup owns a 64 (x) x 16 (y) tile and MARCHES over a chunk of ZC planes:
//   * each field element is loaded from global memory once per workgroup (float4 per lane) and kept in
//     registers for the three planes k-1, k, k+1 it contributes to (z neighbours never touch memory);
//   * x / y neighbours of the current plane go through an LDS tile (18 x 68 floats per field, double
//     buffered -> one barrier per plane); only the tile's 2 halo rows + 2 halo columns are extra loads;
//   * the loads of plane k+2 are issued before plane k is consumed (software prefetch).
// z-halo over-fetch drops from 2/4 to 2/ZC planes per brick.
#include <math.h>
#include <stdlib.h>

#include "fg_internal.h"
#include "fg_zmarch.h"

namespace {

// metrics of the thread's cells for plane k (x/y parts are loop invariant)
__device__ __forceinline__ void z_metrics(const FgGrid& g, const ZCtx& c, FgMetric<3, 4>& m) {
    const int i0 = c.valid ? c.i0 : 0, j = c.valid ? c.j : 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { m.hx[e] = g.h[0][i0 + e]; m.rhx[e] = g.rh[0][i0 + e]; }
    m.rhx_m = g.rh[0][(i0 == 0) ? g.nx - 1 : i0 - 1];
    m.rhx_p = g.rh[0][(i0 + 4 == g.nx) ? 0 : i0 + 4];
    m.hy = g.h[1][j]; m.rhy = g.rh[1][j];
    m.rhy_m = g.rh[1][(j == 0) ? g.ny - 1 : j - 1];
    m.rhy_p = g.rh[1][(j == g.ny - 1) ? 0 : j + 1];
}
// p: PHYSICAL plane; flip: the chunk marches downward, i.e. its logical -z / +z neighbours are the physical +z / -z ones
__device__ __forceinline__ void z_metrics_plane(const FgGrid& g, int p, bool flip, FgMetric<3, 4>& m, FgCtx<3, 4>& fc) {
    m.hz = g.h[2][p]; m.rhz = g.rh[2][p];
    const float rh_lo = g.rh[2][(p == 0) ? g.nz - 1 : p - 1], rh_hi = g.rh[2][(p == g.nz - 1) ? 0 : p + 1];
    const float m_lo = (p == 0 && g.fixed[4]) ? 0.f : 1.f, m_hi = (p == g.nz - 1 && g.fixed[5]) ? 0.f : 1.f;
    m.rhz_m = flip ? rh_hi : rh_lo; m.rhz_p = flip ? rh_lo : rh_hi;
    fc.mzm = flip ? m_hi : m_lo; fc.mzp = flip ? m_lo : m_hi;
}

enum { MODE_APPLY = 0, MODE_RELAX = 1, MODE_CG_AP = 2 };

struct Z3Args {
    const float* rA;      // [B,N]
    const float* x;       // apply/relax: x ; cg_ap: z (preconditioned residual or r)
    const float* x2;      // relax: b ; cg_ap: p_prev
    float* y;             // apply: y ; relax: xnew ; cg_ap: p_out
    float* y2;            // cg_ap: Ap
    float omega; int color;           // relax (color < 0: Jacobi)
    // cg_ap
    FgDacc* acc; int32_t* flags; fg_solve_info* info; int32_t* prof_active; FgBest best;
    float tol; int it; int first; int ns; int num_base;
};

__device__ __forceinline__ FgDacc* z_acc_ptr(FgDacc* acc, int b, int name) { return acc + ((size_t)b * 8 + name) * 64; }
__device__ __forceinline__ double z_acc_total(const FgDacc* a, int ns) {
    if (ns == 1) return acc_ld(a + (0));
    const int lane = threadIdx.x & 63;
    double v = (lane < ns) ? acc_ld(a + (lane)) : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// SB (PPB == 1 only): one barrier per plane instead of two -- the ring has a fourth slot, plane k+2 is committed into the slot
// nobody reads during step k (it held plane k-2), and the barrier that ends the step both publishes it and retires plane k-1.
// 50 KB of LDS per workgroup: three workgroups per CU, so the launch geometry aims at 768 workgroups per round (launch_march).
template <int MODE, int BXL, int PPB, bool SB = false>
__global__ __launch_bounds__(FG_BLOCK) __attribute__((amdgpu_waves_per_eu(MODE == 2 && PPB == 1 && !SB ? 4 : 1))) void k_poisson3_march(FgGrid g, Z3Args a, int tiles_x, int tiles_y, int zchunks,
                                                              int ZC) {
    constexpr int TY = ZT<BXL>::TY, LP = ZT<BXL>::LP, LROWS = ZT<BXL>::LROWS;
    const ZCtx c = z_make_ctx<BXL>(g, tiles_x, tiles_y, zchunks, ZC);
    // Odd z-chunks march DOWNWARD (round 4, as fg_bicgstab3d.hip): a chunk and its neighbour then touch the planes they share --
    // each other's z halo, 2 / ZC of the loads -- at the same time, and the second reader finds them in the XCD's L2.  The march
    // itself runs in logical plane indices k0 .. k1-1 as before; phys() maps a logical plane (halo planes k0-1 and k1 included)
    // to the plane in memory, and z_metrics_plane swaps the roles of the z neighbours.
    const bool flip = ((c.k0 / ZC) & 1) != 0;
    auto phys = [&](int k) { return flip ? (c.k0 + c.k1 - 1 - k) : k; };
    float beta = 0.f;
    bool use_prev = false;
    unsigned tile_id = 0;
    if constexpr (MODE == MODE_CG_AP) {
        if (flag_ld(a.flags + (c.b)) != 0) return;
        const unsigned per_env = tiles_x * tiles_y * zchunks;
        tile_id = fg_xcd_remap(blockIdx.x, gridDim.x) % per_env;
        const double rr_new = z_acc_total(z_acc_ptr(a.acc, c.b, a.it % 3), a.ns);
        const float crit = (float)sqrt(rr_new / (double)g.n);
        if (!(crit >= a.tol)) {
            if (tile_id == 0 && threadIdx.x == 0) {
                const bool finite = isfinite(crit);
                flag_st(a.flags + (c.b), finite ? 1 : 2);
                a.info[c.b].final_residual = crit;
                a.info[c.b].used_iterations = a.it - 1;
                a.info[c.b].converged = finite ? 1 : 0;
                a.info[c.b].is_finite = finite ? 1 : 0;
            }
            return;
        }
        const double num_new = (a.num_base == 0) ? rr_new : z_acc_total(z_acc_ptr(a.acc, c.b, a.num_base + a.it % 3), a.ns);
        const double num_old = a.first ? 1.0 : z_acc_total(z_acc_ptr(a.acc, c.b, a.num_base + (a.it + 2) % 3), a.ns);
        if (tile_id == 0 && threadIdx.x < 64) {
            const int lane = threadIdx.x;
            if (lane < a.ns) {
                acc_st(z_acc_ptr(a.acc, c.b, (a.it + 1) % 3) + lane, 0.0);
                if (a.num_base) acc_st(z_acc_ptr(a.acc, c.b, a.num_base + (a.it + 1) % 3) + lane, 0.0);
            }
            if (lane == 0) {
                a.info[c.b].final_residual = crit;
                a.info[c.b].used_iterations = a.it - 1;
                if (a.prof_active) atomicAdd(a.prof_active, 1);
                fg_best_decide(a.best, c.b, crit, a.it);
            }
        }
        beta = a.first ? 0.f : (float)(num_new / num_old);
        use_prev = !a.first;
    }

    // LDS ring of PPB + 2 planes per stencil field: planes k-1 .. k+PPB are resident while planes k .. k+PPB-1 are computed,
    // so the z neighbours are LDS reads too and a thread only holds the in-flight loads of the next PPB planes in registers
    // (register pressure decides occupancy here: the register-plane variant needed 164 VGPRs = 3 waves/SIMD).  PPB = 2 halves
    // the barriers per plane and doubles the bytes a thread has in flight; plane q of a chunk lives in slot (q - k0 + 1) % NS.
    constexpr int NS = PPB + 2;
    constexpr int NSLOT = NS + (SB ? 1 : 0);
    __shared__ __attribute__((aligned(16))) float ring_p[NSLOT][LROWS * LP];
    __shared__ __attribute__((aligned(16))) float ring_a[NSLOT][LROWS * LP];
    __shared__ float red[4];

    const size_t env = (size_t)c.b * g.n;
    const unsigned plane_b = (unsigned)(g.nx * g.ny) * 4u;   // bytes per plane
    const unsigned env_b = (unsigned)g.n * 4u;
    const rsrc_t R_a = z_rsrc(a.rA + env, env_b);
    const rsrc_t R_x = z_rsrc(a.x + env, env_b);
    const rsrc_t R_x2 = z_rsrc(a.x2 ? a.x2 + env : a.x + env, env_b);
    const rsrc_t R_y = z_rsrc(a.y + env, env_b);
    const rsrc_t R_y2 = z_rsrc(a.y2 ? a.y2 + env : a.y + env, env_b);
    const unsigned vo_c = (unsigned)c.row_c * 4u;
    const unsigned vo_hy = (unsigned)((c.ly == 0) ? c.row_ym : c.row_yp) * 4u;
    const unsigned vo_hx = (unsigned)((c.lx == 0) ? c.col_xm : c.col_xp) * 4u;
    (void)TY;

    FgMetric<3, 4> m;
    z_metrics(g, c, m);
    FgCtx<3, 4> fc;  // only the mask members are used
    fc.mxm = c.mxm; fc.mxp = c.mxp; fc.mym = c.mym; fc.myp = c.myp;
    // thread-invariant factors of the face coefficients (see the plane loop)
    float hyx[4], hxy[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { hyx[e] = m.hy * m.rhx[e]; hxy[e] = m.hx[e] * m.hy; }
    const float hyx_m = m.hy * m.rhx_m, hyx_p = m.hy * m.rhx_p;
    const float s_ycl = c.mym * m.rhy, s_ynl = c.mym * m.rhy_m, s_ycr = c.myp * m.rhy, s_ynr = c.myp * m.rhy_p;

    struct Staged { FgVec<4> p, a; Halo hp, ha; };
    auto stage = [&](int k) -> Staged {  // global loads of one plane (centre + this thread's halo duty)
        Staged r;
        const unsigned so = (unsigned)z_plane(g, phys(k)) * plane_b;
        r.p = z_bload4(R_x, vo_c, so);
        r.hp = z_load_halo<BXL>(c, R_x, vo_hy, vo_hx, so);
        if constexpr (MODE == MODE_CG_AP) {
            if (use_prev) {
                const FgVec<4> w = z_bload4(R_x2, vo_c, so);
                const Halo hw = z_load_halo<BXL>(c, R_x2, vo_hy, vo_hx, so);
#pragma unroll
                for (int e = 0; e < 4; ++e) { r.p.v[e] += beta * w.v[e]; r.hp.y.v[e] += beta * hw.y.v[e]; }
                r.hp.x += beta * hw.x;
            }
        }
        r.a = z_bload4(R_a, vo_c, so);
        r.ha = z_load_halo<BXL>(c, R_a, vo_hy, vo_hx, so);
        return r;
    };
    auto commit = [&](int slot, const Staged& r) {
        z_fill_tile<BXL>(ring_p[slot], c, r.p, r.hp);
        z_fill_tile<BXL>(ring_a[slot], c, r.a, r.ha);
    };
    // prologue: planes k0-1 .. k0+PPB -> slots 0 .. NS-1.  All planes are requested before the first commit: done one after
    // the other the prologue cost one exposed memory round trip per plane and z-chunk.
    {
        Staged s0[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) s0[q] = stage(c.k0 - 1 + q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NS; ++q) commit(q, s0[q]);
    }
    FgVec<4> bvec[PPB];
    if constexpr (MODE == MODE_RELAX) {
#pragma unroll
        for (int u = 0; u < PPB; ++u) bvec[u] = z_bload4(R_x2, vo_c, (unsigned)phys(min(c.k0 + u, c.k1 - 1)) * plane_b);
    }
    __syncthreads();
    float dot = 0.f;
    const int cen = (c.ly + 1) * LP + 4 + c.lx * 4;  // LDS offset of this thread's centre vector
    int s_first = 0;                                  // ring slot of plane k-1
    // z faces: every cell's half coefficient gz = (0.5 rh_z) (hx hy a) is formed once, a face is the sum of the two cells it
    // separates, and a thread marching in z carries gz of the plane above and the face it shares with it: exactly one face
    // evaluation per cell and plane, and both sides of a face use the same bits.
    // (not in the CG kernel: the eight carried registers cost it a wave of occupancy, 156 -> 169 us per CG iteration at 256^3)
    constexpr bool CARRY = (MODE != MODE_CG_AP);
    float gz_c[4], face_zm[4];
    bool have_carry = false;
    auto plane = [&](int k, int sm, int sc, int sp, const FgVec<4>& bv) {
        z_metrics_plane(g, phys(k), flip, m, fc);
        const float4 P_c = *reinterpret_cast<const float4*>(&ring_p[sc][cen]);
        const float4 A_c = *reinterpret_cast<const float4*>(&ring_a[sc][cen]);
        float pcv[4] = {P_c.x, P_c.y, P_c.z, P_c.w};
        float acv[4] = {A_c.x, A_c.y, A_c.z, A_c.w};
        float y[4] = {0.f, 0.f, 0.f, 0.f}, dg[4] = {0.f, 0.f, 0.f, 0.f};
        // Face coefficients  c_f = mask_f * 0.5 * area_f * (rh_c a_c + rh_n a_n)  (fg_poisson_coef).  The kernels are
        // VALU-limited (PMC: Jacobi 24 % VALU-active per wave at 4 waves per SIMD), so the products are arranged to
        // share work: every cell's half coefficient g = 0.5 area rh a is formed once per direction and an x face is
        // the sum of the two cells it separates (5 faces for 4 cells instead of 8 one-sided evaluations); thread- and
        // plane-invariant factors are hoisted (s_y*, hyx*, hxy below; no -ffast-math, so the compiler may not).
        const float hh = 0.5f * m.hz;
        {   // x faces
            const float al = ring_a[sc][cen - 1], ar = ring_a[sc][cen + 4];
            const float pl = ring_p[sc][cen - 1], pr = ring_p[sc][cen + 4];
            float gx[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) gx[e] = hh * (hyx[e] * acv[e]);
            float face[5];
            face[0] = fc.mxm * (hh * (hyx_m * al) + gx[0]);
#pragma unroll
            for (int e = 1; e < 4; ++e) face[e] = gx[e - 1] + gx[e];
            face[4] = fc.mxp * (gx[3] + hh * (hyx_p * ar));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pm_ = (e == 0) ? pl : pcv[e > 0 ? e - 1 : 0];
                const float pp_ = (e == 3) ? pr : pcv[e < 3 ? e + 1 : 3];
                y[e] += face[e] * (pm_ - pcv[e]) + face[e + 1] * (pp_ - pcv[e]);
                dg[e] -= face[e] + face[e + 1];
            }
        }
        float hxa[4];  // hx a_c, shared by the y and z directions
#pragma unroll
        for (int e = 0; e < 4; ++e) hxa[e] = m.hx[e] * acv[e];
        {   // y faces
            const float4 Pm = *reinterpret_cast<const float4*>(&ring_p[sc][cen - LP]);
            const float4 Pp = *reinterpret_cast<const float4*>(&ring_p[sc][cen + LP]);
            const float4 Am = *reinterpret_cast<const float4*>(&ring_a[sc][cen - LP]);
            const float4 Ap = *reinterpret_cast<const float4*>(&ring_a[sc][cen + LP]);
            const float pm_[4] = {Pm.x, Pm.y, Pm.z, Pm.w}, pp_[4] = {Pp.x, Pp.y, Pp.z, Pp.w};
            const float am_[4] = {Am.x, Am.y, Am.z, Am.w}, ap_[4] = {Ap.x, Ap.y, Ap.z, Ap.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float tc = hh * hxa[e], hhx = hh * m.hx[e];
                const float cl = s_ycl * tc + s_ynl * (hhx * am_[e]);
                const float cr = s_ycr * tc + s_ynr * (hhx * ap_[e]);
                y[e] += cl * (pm_[e] - pcv[e]) + cr * (pp_[e] - pcv[e]);
                dg[e] -= cl + cr;
            }
        }
        if constexpr (!CARRY) {   // z faces, both evaluated from the ring (the CG kernel: no registers to spare for the carry)
            const float4 Pm = *reinterpret_cast<const float4*>(&ring_p[sm][cen]);
            const float4 Pp = *reinterpret_cast<const float4*>(&ring_p[sp][cen]);
            const float4 Am = *reinterpret_cast<const float4*>(&ring_a[sm][cen]);
            const float4 Ap = *reinterpret_cast<const float4*>(&ring_a[sp][cen]);
            const float pm_[4] = {Pm.x, Pm.y, Pm.z, Pm.w}, pp_[4] = {Pp.x, Pp.y, Pp.z, Pp.w};
            const float am_[4] = {Am.x, Am.y, Am.z, Am.w}, ap_[4] = {Ap.x, Ap.y, Ap.z, Ap.w};
            const float zcl = fc.mzm * 0.5f * m.rhz, znl = fc.mzm * 0.5f * m.rhz_m;
            const float zcr = fc.mzp * 0.5f * m.rhz, znr = fc.mzp * 0.5f * m.rhz_p;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float uc = m.hy * hxa[e];
                const float cl = zcl * uc + znl * (hxy[e] * am_[e]);
                const float cr = zcr * uc + znr * (hxy[e] * ap_[e]);
                y[e] += cl * (pm_[e] - pcv[e]) + cr * (pp_[e] - pcv[e]);
                dg[e] -= cl + cr;
            }
        } else {   // z faces with the carried half coefficients
            const float4 Pm = *reinterpret_cast<const float4*>(&ring_p[sm][cen]);
            const float4 Pp = *reinterpret_cast<const float4*>(&ring_p[sp][cen]);
            const float4 Ap = *reinterpret_cast<const float4*>(&ring_a[sp][cen]);
            const float pm_[4] = {Pm.x, Pm.y, Pm.z, Pm.w}, pp_[4] = {Pp.x, Pp.y, Pp.z, Pp.w};
            const float ap_[4] = {Ap.x, Ap.y, Ap.z, Ap.w};
            const float wz_p = 0.5f * m.rhz_p;
            if (!have_carry) {   // first plane of the chunk (uniform branch): the face below from plane k-1 in the ring
                const float4 Am = *reinterpret_cast<const float4*>(&ring_a[sm][cen]);
                const float am_[4] = {Am.x, Am.y, Am.z, Am.w};
                const float wz_c = 0.5f * m.rhz, wz_m = 0.5f * m.rhz_m;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gz_c[e] = wz_c * (hxy[e] * acv[e]);
                    face_zm[e] = fc.mzm * (wz_m * (hxy[e] * am_[e]) + gz_c[e]);
                }
                have_carry = true;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gz_p = wz_p * (hxy[e] * ap_[e]);
                const float face_zp = fc.mzp * (gz_c[e] + gz_p);
                y[e] += face_zm[e] * (pm_[e] - pcv[e]) + face_zp * (pp_[e] - pcv[e]);
                dg[e] -= face_zm[e] + face_zp;
                gz_c[e] = gz_p; face_zm[e] = face_zp;      // the plane above: its centre half coefficient and lower face
            }
        }
        FgVec<4> out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (MODE == MODE_RELAX) {
                float v = pcv[e] + a.omega * (bv.v[e] - y[e]) * __builtin_amdgcn_rcpf(dg[e]);  // v_rcp_f32 (1 ulp): the IEEE division was ~10 VALU ops per cell in a VALU-bound kernel
                if (a.color >= 0 && (((c.i0 + e + c.j + phys(k)) & 1) != a.color)) v = pcv[e];
                out.v[e] = v;
            } else {
                out.v[e] = y[e];
                if constexpr (MODE == MODE_CG_AP) dot += pcv[e] * y[e];
            }
        }
        if (c.valid) {
            const unsigned so = (unsigned)phys(k) * plane_b;
            if constexpr (MODE == MODE_CG_AP) {
                z_bstore4(R_y2, vo_c, so, out);  // Ap
                FgVec<4> pc_;
                pc_.v[0] = pcv[0]; pc_.v[1] = pcv[1]; pc_.v[2] = pcv[2]; pc_.v[3] = pcv[3];
                z_bstore4(R_y, vo_c, so, pc_);   // p
            } else {
                z_bstore4(R_y, vo_c, so, out);
            }
        }
    };
    if constexpr (PPB == 1) {   // one plane per barrier pair: three rotating slot indices, one staged plane
        int sm = 0, sc = 1, sp = 2, sf = 3;
        (void)sf;
#pragma unroll 1
        for (int k = c.k0; k < c.k1; ++k) {
            const bool more = (k + 1 < c.k1);
            Staged nxt;
            FgVec<4> bnext;
            if (more) {
                nxt = stage(k + 2);
                if constexpr (MODE == MODE_RELAX) bnext = z_bload4(R_x2, vo_c, (unsigned)phys(k + 1) * plane_b);
            }
            __builtin_amdgcn_sched_barrier(0);
            plane(k, sm, sc, sp, bvec[0]);
            if (!more) break;
            if constexpr (SB) {
                commit(sf, nxt);        // the free slot: last read as plane k-2, a barrier ago
                __syncthreads();        // plane k+2 is visible; every wave is done with plane k-1
                const int t3 = sm; sm = sc; sc = sp; sp = sf; sf = t3;
            } else {
                __syncthreads();            // every wave is done reading slot sm (plane k-1)
                commit(sm, nxt);            // plane k+2 takes its place
                __syncthreads();
                const int t3 = sm; sm = sc; sc = sp; sp = t3;
            }
            if constexpr (MODE == MODE_RELAX) bvec[0] = bnext;
        }
    } else
#pragma unroll 1
    for (int k = c.k0; k < c.k1; k += PPB) {
        // ---- prefetch the next PPB planes (consumed after this step's arithmetic) and their b
        const bool more = (k + PPB < c.k1);
        Staged nxt[PPB];
        FgVec<4> bnext[PPB];
        if (more) {
#pragma unroll
            for (int u = 0; u < PPB; ++u) {
                nxt[u] = stage(k + PPB + 1 + u);
                if constexpr (MODE == MODE_RELAX) bnext[u] = z_bload4(R_x2, vo_c, (unsigned)phys(min(k + PPB + u, c.k1 - 1)) * plane_b);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- planes k .. k+PPB-1 from the LDS ring
#pragma unroll
        for (int u = 0; u < PPB; ++u) {
            if (u == 0 || k + u < c.k1) {
                int sm = s_first + u, sc = s_first + u + 1, sp = s_first + u + 2;
                sm = sm >= NS ? sm - NS : sm; sc = sc >= NS ? sc - NS : sc; sp = sp >= NS ? sp - NS : sp;
                plane(k + u, sm, sc, sp, bvec[u]);
            }
        }
        if (!more) break;
        __syncthreads();            // every wave is done reading the slots of planes k-1 .. k+PPB-2
#pragma unroll
        for (int u = 0; u < PPB; ++u) {   // planes k+PPB+1+u take their place
            int sl = s_first + u;
            sl = sl >= NS ? sl - NS : sl;
            commit(sl, nxt[u]);
        }
        __syncthreads();
        s_first += PPB;
        s_first = s_first >= NS ? s_first - NS : s_first;
        if constexpr (MODE == MODE_RELAX) {
#pragma unroll
            for (int u = 0; u < PPB; ++u) bvec[u] = bnext[u];
        }
    }
    if constexpr (MODE == MODE_CG_AP) {
        float part[1] = {c.valid ? dot : 0.f};
        fg_block_sum<1>(part, red);
        if (threadIdx.x == 0) acc_add(z_acc_ptr(a.acc, c.b, 3 + (a.it & 1)) + (tile_id & (unsigned)(a.ns - 1)), (double)part[0]);
    }
}

}  // namespace

// geometry -----------------------------------------------------------------------------------------
static int z_shape() {  // lanes along x: 16 (64 x 16 tile), 32 (128 x 8), 64 (256 x 4); FG_ZMARCH_BXL overrides
    static const int v = 0;      // (FG_ZMARCH_BXL until round 5: the tile width follows the grid)
    return v;
}
static int z_pick_bxl(const FgGrid& g) {
    const int forced = z_shape();
    if (forced == 16 || forced == 32 || forced == 64) return forced;
    // widest tile the grid is a multiple of: 256 x 4 measured best at 256^3 (1 KiB contiguous per wave access)
    if (g.nx % 256 == 0 && g.ny % 4 == 0) return 64;
    if (g.nx % 128 == 0 && g.ny % 8 == 0) return 32;
    return 16;
}

bool fg_zmarch_ok(const fg_state* s, int* zc_out) {
    const FgGrid& g = s->grid;
    const int bxl = z_pick_bxl(g);
    const int TX = bxl * 4, TY = FG_BLOCK / bxl;
    // tiles must coincide with the grid (every thread valid; halo loaders handle wrap / clamp at tile edges)
    if (g.dims != 3 || s->vec != 4 || g.nz < 32 || (g.nx % TX) != 0 || (g.ny % TY) != 0) return false;
    const int tiles = (g.nx / TX) * (g.ny / TY);
    static const int force = [] { const char* e = getenv("FG_FORCE_ZMARCH"); return e ? atoi(e) : 0; }();
    if (force < 0) return false;            // FG_FORCE_ZMARCH=-1: always use the generic brick kernels
    if (force > 0) { *zc_out = force; return true; }  // FG_FORCE_ZMARCH=ZC: tests exercise small grids
    int zc = 32;
    while (zc > 8 && (long)tiles * ((g.nz + zc - 1) / zc) * g.B < 2048) zc /= 2;  // >= 8 workgroups per CU
    if ((long)tiles * ((g.nz + zc - 1) / zc) * g.B < 512) return false;
    *zc_out = zc;
    return true;
}

template <int MODE>
static int launch_march(const fg_state* s, const Z3Args& a, int zc, hipStream_t st, int slot = -1) {
    const FgGrid& g = s->grid;
    const int bxl = z_pick_bxl(g);
    const int tx = g.nx / (bxl * 4), ty = g.ny / (FG_BLOCK / bxl);
    // apply carries the least per-plane state and gains from longer chunks (two halo planes per chunk): twice the planes while
    // that leaves >= 4 workgroups per CU (256^3: 16 planes, 43.7 us against 46.3 with 8)
    if (MODE == MODE_APPLY && zc < 32 && (long)tx * ty * ((g.nz + 2 * zc - 1) / (2 * zc)) * g.B >= 1024) zc *= 2;
    const int zch = (g.nz + zc - 1) / zc;
    dim3 grid((unsigned)(tx * ty * zch * g.B));
    // measured at 256^3 (profiles/zmarch_sweep.sh, r02): the Jacobi / RB-GS sweep is fastest with two planes per barrier pair
    // (55.5 us against 58.8), apply and the CG kernel with one (registers: 152-170 with two)
    constexpr int PPB = (MODE == MODE_RELAX) ? 2 : 1;
    // Single-barrier ring (SB) for the sweep and the CG kernel: three workgroups fit a CU (50 KB of LDS each), so it pays when the
    // launch is ONE round of 3 x CUs workgroups with balanced chunks -- 256^3: 64 tiles x 12 chunks of 22 planes = 768 workgroups,
    // Jacobi 56.0 -> 52.2 us, CG iteration 153-155 -> 143-148 us; with any other chunk length (a second, partial round) it loses
    // (20 planes: 72 us, 24: 55 us), and the bare apply (one round of 1024 already) gains nothing from it (49 us against 46).
    // Most of the gain is the geometry (one full round, 9 % z-halo planes instead of 25 %); two planes in flight per thread on
    // top of it: nothing (Jacobi 53.7 us), and in the two-barrier apply it spills (124 VGPRs at four waves already).
    // FG_ZMARCH_SB: bit MODE forces it on with the caller's chunk length, 0 = never; unset = the rule above.
    static const int sb_mask = [] { const char* e = getenv("FG_ZMARCH_SB"); return e ? atoi(e) : -1; }();
    bool sb = sb_mask >= 0 && ((sb_mask >> MODE) & 1);
    if (sb_mask < 0 && MODE != MODE_APPLY) {
        static const int slots = [] {
            int dev = 0, cus = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
            return 3 * cus;
        }();
        const long layer = (long)tx * ty * g.B;                       // workgroups per layer of chunks
        const int chunks = layer > 0 ? (int)(slots / layer) : 0;
        if (chunks >= 2 && layer * chunks == slots) {                 // fills the round exactly
            const int zc_sb = (g.nz + chunks - 1) / chunks;
            if (zc_sb >= 16 && (g.nz + zc_sb - 1) / zc_sb == chunks) { sb = true; zc = zc_sb; }
        }
    }
    if (sb) {
        const int zch = (g.nz + zc - 1) / zc;
        const dim3 grid((unsigned)(tx * ty * zch * g.B));
        if (bxl == 16) FG_LAUNCH_P(s, slot, (k_poisson3_march<MODE, 16, 1, true>), grid, dim3(FG_BLOCK), 0, st, g, a, tx, ty, zch, zc);
        else if (bxl == 32) FG_LAUNCH_P(s, slot, (k_poisson3_march<MODE, 32, 1, true>), grid, dim3(FG_BLOCK), 0, st, g, a, tx, ty, zch, zc);
        else FG_LAUNCH_P(s, slot, (k_poisson3_march<MODE, 64, 1, true>), grid, dim3(FG_BLOCK), 0, st, g, a, tx, ty, zch, zc);
        FG_HIP_CHECK(hipGetLastError());
        return FG_OK;
    }
    if (bxl == 16) FG_LAUNCH_P(s, slot, (k_poisson3_march<MODE, 16, PPB>), grid, dim3(FG_BLOCK), 0, st, g, a, tx, ty, zch, zc);
    else if (bxl == 32) FG_LAUNCH_P(s, slot, (k_poisson3_march<MODE, 32, PPB>), grid, dim3(FG_BLOCK), 0, st, g, a, tx, ty, zch, zc);
    else FG_LAUNCH_P(s, slot, (k_poisson3_march<MODE, 64, PPB>), grid, dim3(FG_BLOCK), 0, st, g, a, tx, ty, zch, zc);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_zmarch_apply(const fg_state* s, const float* rA, const float* x, float* y, int zc, hipStream_t st) {
    Z3Args a = {};
    a.rA = rA; a.x = x; a.y = y;
    return launch_march<MODE_APPLY>(s, a, zc, st);
}
int fg_zmarch_relax(const fg_state* s, const float* rA, const float* b, const float* x, float* xnew, float omega,
                    int color, int zc, hipStream_t st) {
    Z3Args a = {};
    a.rA = rA; a.x = x; a.x2 = b; a.y = xnew; a.omega = omega; a.color = color;
    return launch_march<MODE_RELAX>(s, a, zc, st);
}
int fg_zmarch_cg_ap(const fg_state* s, const float* rA, const float* z, const float* p_in, float* p_out, float* Ap,
                    FgDacc* acc, int32_t* flags, fg_solve_info* info, int prof_slot, float tol, int it, int first,
                    int ns, int num_base, int zc, hipStream_t st) {
    Z3Args a = {};
    a.rA = rA; a.x = z; a.x2 = p_in; a.y = p_out; a.y2 = Ap;
    a.acc = acc; a.flags = flags; a.info = info; a.prof_active = prof_slot >= 0 ? s->prof.active_dev + prof_slot : nullptr;
    a.best = s->cg_best;
    a.tol = tol; a.it = it; a.first = first; a.ns = ns; a.num_base = num_base;
    return launch_march<MODE_CG_AP>(s, a, zc, st, prof_slot);
}
