// Multi-block path: the pressure CG of one env spread over a CLUSTER of workgroups (k_mbc_cluster), its table builder and host
// launcher.  Round 6.  Replaces cgSolveGPU (cg_solver_kernel.cu:129-471) on the pressure matrix of
// PISO_multiblock_cuda_kernel.cu:4812-4978 for the cylinder family (tolerance cylinder_env_base.py:315), as k_mbc_onchip /
// k_mbc_l2 (fg_mb_onchip.hip) do with ONE workgroup per env -- 64 envs on 64 of the chip's 256 CUs for 61-72 % of the GPU time of
// a cylinder step (profiles/r05_c_*, r05_d_*).  gfx950 / wave64 only.
//
// A cluster is CL_G = 4 workgroups; each owns a contiguous range of the mesh's 8 x 8 aggregates (so every 4 x 4 aggregate, every
// restriction and every prolongation of the multilevel preconditioner is local to one workgroup), a quarter of the cells.  At a
// quarter of the mesh per workgroup the WHOLE solver state of a cell fits in registers: its matrix row (diagonal, four
// off-diagonals, row sum), the four neighbour addresses, r, x, p and s = P p -- nothing of the matrix streams per iteration
// (k_mbc_onchip re-reads 6 F bytes per cell and iteration from L2, k_mbc_l2 ~74).  Only the vector the stencil is applied to lives
// in LDS, with a halo of the neighbouring workgroups' cells behind the owned slots.
//
// What crosses workgroups crosses as 8-byte {value, tag} granules written by ONE write-through (sc1) store and polled with sc1
// loads: the data is the flag, no fence, no atomic read-modify-write, no dependence on where a workgroup runs
// (cdna_hip_programming.md section 6 Guideline 16 form R2; MI355X_MICROARCH.md "handoff-1to1": ~1 us per hop for <= 4 KB).
// Workgroups of a cluster are numbered onto one XCD for speed only.  Tags are epochs counted per env ACROSS launches (the
// counter lives on the device), granule buffers alternate with the epoch's parity, so a granule is rewritten only after
// every reader has published the exchange in between.
//
// TWO exchanges per iteration.  The recurrence is the one of k_mbc_onchip with the operator applied to z instead of p
// (s = P p follows by linearity: s = P z~ + beta s), so that ONE halo exchange carries z together with every sum that became
// known with it:
//     [P]  z = M (r - mean r)                          local: Jacobi + 4 x 4 aggregates + own rows of the coarse solve
//     [XB] halo of z | r.z, sum z | r.r, 1.r           -> beta, mean z, and the verdict on r (same RMS criterion, same rules)
//          | this workgroup's rows of Z8^T r           -> the coarse residual every workgroup needs for the NEXT z
//     [S]  w = P (z - mean z), p = z~ + beta p, s = w + beta s
//     [XC] p.s, 1.s | this workgroup's rows of Z8^T s  -> alpha; coarse residual and mean of the next r by a ONE-step recurrence
//          x += alpha p, r -= alpha s                     from the exact values [XB] carried
// p.Pp is computed (p.s), never derived from a scalar recurrence: the matrix of these meshes is not symmetric and
// the Chronopoulos-Gear identity does not hold on it.  Restart (residualResetSteps, cg_solver_kernel.cu:281-300), best iterate
// (returnBestResult, :345-361), breakdown recovery, stall / acceptance rules: k_mbc_onchip's, statement by statement; every
// decision is taken by every workgroup from the same doubles summed in the same order, so the cluster never disagrees.
//
// A workgroup that does not get a granule in time (its peers are not resident: another process's cluster kernel shares the GPU)
// gives up, marks the env, and the host repeats the batch with the one-workgroup kernels -- the grid never exceeds what one
// launch keeps resident (clusters loop over envs), so a lone process cannot get there.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "fg_mb.h"
#include "fg_mb_solve.h"

namespace {

constexpr int CL_G = FG_CL_G;            // workgroups per cluster (fg_mb.h)
[[maybe_unused]] constexpr int CL_NSC = 12;               // scalar words per workgroup and exchange: four doubles, the XCD id
constexpr unsigned CL_SPIN_LIMIT = 400000u;   // polls of one granule before a workgroup gives up (>= 100 ms)

typedef unsigned long long cl_u64;

struct ClParams {
    // mesh tables (shared by all envs); 8 x 8 aggregates in CLUSTER order (sorted along x, so that a workgroup's range is a slab)
    const int32_t* slot_cell;    // [G][S] cell of a slot, -1 = hole
    const uint2* nbr;            // [G][S] the four neighbour addresses as byte offsets into the workgroup's LDS vector (halo behind the slots)
    const mb_real* d4g;          // [G][NT] 1 / diag(Z4^T S Z4) of the thread's 4 x 4 aggregate
    const uint32_t* tinfo;       // [G][NT] valid members (bits 0-7) | members whose neighbours are all the workgroup's own (8-15) | cells of the thread's 4 x 4 aggregate << 16
    const int32_t* out_slot;     // [G][n_out_max] slot published at position k of the workgroup's halo box
    const uint32_t* halo_src;    // [G][n_halo_max] producer workgroup << 16 | position in its box
    const mb_real* cnt8;         // [n8] cells of an 8 x 8 aggregate
    const mb_real* aci8;         // [n8][ld] (pseudo-)inverse of the coarse operator in cluster order, rows padded to a multiple of four
    const _Float16* aci16;       // [n8][ld16] the same as fp16 times 1 / aci16_unscale (a power of two), rows padded to a multiple of eight
    int ld16; mb_real aci16_unscale;
    int a8_first[CL_G + 1];      // 8 x 8 aggregates [a8_first[g], a8_first[g + 1]) belong to workgroup g
    int n_out[CL_G], n_halo[CL_G];
    int n_out_max, n_halo_max, n8g_max, n8, ld8, W;   // W: granules of one box = n_out_max + n8g_max + CL_NSC
    mb_real geom_diag_sum;
    // per env
    const mb_real* off4;         // [B][G][S][4] slot order, holes 0 (k_mb_pmatrix)
    const mb_real* diag;         // [B][G][S]
    mb_real* bestx;              // [B][G][S]
    cl_u64* box;                 // [B][2][G][W] granules
    uint32_t* epoch;             // [B] last epoch an env's cluster used (atomicMax at the end of a solve)
    uint32_t* abort_at;          // [B] epoch base of the launch in which a workgroup of the env gave up
    // solve
    const mb_real* dt;           // [B] or null
    int B, N, n_clusters, allow_near;
    unsigned long long* dbg_out;   // -DFG_CL_CYCLES: cycles per phase of workgroup 0 of env 0 (fg_mb_debug_cycles)
    int use_x0, pm, restart_every, check_every, max_iterations, stall_limit, accept_window;
    mb_real accept_factor, tol;
    fg_solve_info* info_host; int32_t* its_host; FgPollOut poll;
};

// `near`: every workgroup of the cluster reported the same XCD, i.e. they share ONE L2 -- the store may stay in it (sc0: the line is
// kept; an sc1 store drops it, and the reader then fetches at the cross-XCD rate -- MI355X_MICROARCH.md, "stores of each flavour");
// readers bypass their L1 either way (sc1 loads), so this is a speed path chosen from what the hardware reported, never assumed
__device__ __forceinline__ void cl_store(cl_u64* p, uint32_t tag, uint32_t value, bool near) {
    const cl_u64 v = ((cl_u64)tag << 32) | (cl_u64)value;
    if (near) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one granule: re-read until it carries `tag`; false after CL_SPIN_LIMIT polls or when a peer has given up in this launch
__device__ __forceinline__ bool cl_poll(const cl_u64* p, uint32_t tag, const uint32_t* abort_at, uint32_t base, uint32_t& value) {
    unsigned spins = 0;
    for (;;) {
        const cl_u64 v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(v >> 32) == tag) { value = (uint32_t)v; return true; }
        __builtin_amdgcn_s_sleep(1);
        if (++spins >= CL_SPIN_LIMIT) return false;
        if ((spins & 255u) == 0u && __hip_atomic_load(abort_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == base) return false;
    }
}

// a value every lane holds alike (read from LDS, or derived from such): moved to scalar registers, so that the solver's dozen of
// workgroup-uniform doubles do not occupy two vector registers each for the whole solve
__device__ __forceinline__ double cl_uni(double v) {
    const long long b = __double_as_longlong(v);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)((unsigned long long)b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
}
#if !FG_MB_F64
__device__ __forceinline__ mb_real cl_uni(mb_real v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
#endif

// sum over the 2^k lanes of an aligned group; every lane of the group ends with the same bits (a + b == b + a)
template <int LANES>
__device__ __forceinline__ mb_real cl_group_sum(mb_real v) {
#pragma unroll
    for (int o = 1; o < LANES; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// CPT members per thread: a thread owns CPT / 4 rows of up to four cells of ONE 4 x 4 aggregate, R = 16 / CPT threads side by
// side own the aggregate, 4 R an 8 x 8 aggregate (its up to four children); NT threads per workgroup.
// ACI: where the workgroup's rows of the coarse inverse live -- 1: in LDS as fp32 (at most 64 rows of 256 columns: the cylinder's
// `easy` mesh); 2: in LDS as fp16 with one power-of-two scale (at most 96 rows of 384 columns: `medium` / `hard`; the rounding is a
// symmetric 2^-11 perturbation of the coarse part of the preconditioner, two orders below its smallest eigenvalue -- the system and
// the verdicts on its residual are untouched); 0: streamed from L2 every iteration.
typedef _Float16 cl_half8 __attribute__((ext_vector_type(8)));
template <int CPT, int NT, int PM, int ACI>
__global__ __launch_bounds__(NT) void k_mbc_cluster(MbSolve q, ClParams o) {
    static_assert(CPT == 8, "members per thread");   // (four members in 1024 threads was built and measured: 12.8 against 8.6 us per iteration)
    static_assert(PM == 0 || PM == 1, "mean projection or none");
    constexpr int S = CPT * NT, R = 16 / CPT, GL = 4 * R, NW = NT / 64;
    constexpr int HCAP = NT;                   // halo slots behind the owned ones: one granule per thread
    constexpr int N8MAX = 1024;
    constexpr int RPW = 8, ROWS = NW * RPW;    // coarse solve: rows per wave, rows a workgroup can own
    constexpr int ACI_ROWS = 64, ACI_LD = 256;
    constexpr int A16_ROWS = 96, A16_LD = 384;
    // (the off-diagonals in LDS instead of registers -- one 16-byte read per member and use -- for the instances whose threads get
    //  fewer than 256 registers: built and measured on the 23 k-cell mesh, 27.1 against 19.7 us per iteration: the LDS pipe, not the
    //  spilled registers, then sets the pace)
    static_assert(S + HCAP <= 16383, "16-bit byte offsets into the LDS vector");
    __shared__ __attribute__((aligned(16))) mb_real zl[S + HCAP];
    __shared__ __attribute__((aligned(16))) mb_real l_r8[N8MAX];    // Z8^T r (raw sums) of ALL aggregates as the last exchange [XB] carried them, zeros behind n8
    __shared__ __attribute__((aligned(16))) mb_real l_s8[N8MAX];    // Z8^T s of the last exchange [XC]
    __shared__ mb_real l_e8[ROWS], l_acn[ROWS];                     // coarse solution of the own aggregates; (A8^+ cells)(own rows)
    __shared__ __attribute__((aligned(16))) mb_real l_aci[ACI == 1 ? ACI_ROWS * ACI_LD : 4];
    __shared__ __attribute__((aligned(16))) _Float16 l_a16[ACI == 2 ? A16_ROWS * A16_LD : 8];
    __shared__ double l_wave[4][NW];                                // per-wave partial sums
    __shared__ double l_tot[4];                                     // the cluster's sums of the last exchange
    __shared__ mb_real l_f[2];                                      // scalars one thread derives from them for everybody (beta | alpha)
    __shared__ uint32_t l_xcc[CL_G];                                // the XCD every workgroup of the cluster reported
    __shared__ int l_fail;
#ifdef FG_CL_CYCLES
    __shared__ unsigned long long l_ph[16], l_tph;
#define CL_PH(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = clock64(); l_ph[k] += now_ - l_tph; l_tph = now_; } } while (0)
#else
#define CL_PH(k) do { } while (0)
#endif
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // workgroups of a cluster on one XCD (block b runs on XCD b % 8: speed only)
    int cluster, g;
    {
        const int b = blockIdx.x, ncl = o.n_clusters;
        if ((ncl & 7) == 0) { const int xcd = b & 7, j = b >> 3; cluster = (j / CL_G) * 8 + xcd; g = j % CL_G; }
        else { cluster = b / CL_G; g = b % CL_G; }
    }
    const int N = o.N, n8 = o.n8, a0 = o.a8_first[g], n8g = o.a8_first[g + 1] - a0;
    const mb_real rsqn = mb_rsqrt((mb_real)N);
    const double inv_n = 1.0 / (double)N;
    // ---- thread constants: the mesh is the same for every env of the loop
    const uint32_t tinfo = o.tinfo[g * NT + t];
    const uint32_t vmask = tinfo & 0xffu, imask = (tinfo >> 8) & 0xffu;
    const mb_real cnt4 = (mb_real)(tinfo >> 16);
    const mb_real d4g_t = o.d4g[g * NT + t];
    const int la = t / GL;                      // local 8 x 8 aggregate of the thread (la < n8g when it owns cells)
    const bool a8_leader = (t % GL) == 0 && la < n8g;
    uint2 nb[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) nb[k] = o.nbr[(size_t)g * S + t + k * NT];
    // what this thread publishes and collects in an exchange (at most one granule of each kind per thread: mb_cluster_build), as
    // granule offsets inside a box -- kept in registers: a table read per exchange is an L2 round trip in front of every store / poll
    const int n_out = o.n_out[g], n_halo = o.n_halo[g], n_rem = n8 - n8g;
    const int out_sl = t < n_out ? o.out_slot[g * o.n_out_max + t] : 0;
    uint32_t halo_at = 0, a8_at = 0; int a8_A = 0;
    if (t < n_halo) { const uint32_t src = o.halo_src[g * o.n_halo_max + t]; halo_at = (src >> 16) * (uint32_t)o.W + (src & 0xffffu); }
    if (t < n_rem) {   // the t-th aggregate that is not this workgroup's: the peers' ranges in order
        a8_A = t < a0 ? t : t + n8g;
        int gp = 0;
#pragma unroll
        for (int c = 1; c < CL_G; ++c) gp += (a8_A >= o.a8_first[c]) ? 1 : 0;
        a8_at = (uint32_t)(gp * o.W + o.n_out_max + (a8_A - o.a8_first[gp]));
    }
    const int sc_at = o.n_out_max + o.n8g_max; // first scalar word of a box
    const uint32_t my_xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xfu;   // HW_REG_XCC_ID, bits 3:0
    const int nq = o.ld8 >> 2;
    if (ACI == 1) {
        for (int i = t; i < ACI_ROWS * ACI_LD; i += NT) l_aci[i] = 0.f;
        __syncthreads();
        for (int i = t; i < n8g * nq; i += NT) {
            const int row = i / nq, qd = i - row * nq;
            *reinterpret_cast<float4*>(&l_aci[row * ACI_LD + 4 * qd]) = *reinterpret_cast<const float4*>(o.aci8 + (size_t)(a0 + row) * o.ld8 + 4 * qd);
        }
    }
    if (ACI == 2) {
        for (int i = t; i < A16_ROWS * A16_LD / 8; i += NT) reinterpret_cast<float4*>(l_a16)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        const int no = o.ld16 >> 3;
        for (int i = t; i < n8g * no; i += NT) {
            const int row = i / no, oc = i - row * no;
            *reinterpret_cast<float4*>(&l_a16[row * A16_LD + 8 * oc]) = *reinterpret_cast<const float4*>(o.aci16 + (size_t)(a0 + row) * o.ld16 + 8 * oc);
        }
    }
    for (int c = t; c < N8MAX; c += NT) { l_r8[c] = 0.f; l_s8[c] = 0.f; }
    __syncthreads();
    // Rows [a0, a0 + n8g) of A8^+ times v = v1 - f v2 (all columns; v in LDS, zeros behind n8): wave w takes rows 8 w .. 8 w + 7, its
    // lanes the quads of columns -- v is read once per wave, a row's quads by consecutive lanes -- and the eight row sums are formed
    // by a transposing reduction (ten shuffles instead of eight full wave sums).  Returns the sum of row 8 w + cr_row(lane) in every
    // lane (k_mbc_onchip's dense coarse solve; one thread per (row, eighth of the columns) was LDS-bandwidth bound: 1 885 cycles).
    const int cr_j = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
    const int cr_row = wave * RPW + cr_j;
    auto coarse_rows = [&](const mb_real* v1, const mb_real* v2, mb_real f) {
        mb_real acc[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) acc[j] = 0.f;
        if (ACI == 2) {
            // one pass: a lane takes eight columns (two quads of v, one 16-byte read of eight halves per row)
            float4 c0 = *reinterpret_cast<const float4*>(&v1[8 * lane]), c1 = *reinterpret_cast<const float4*>(&v1[8 * lane + 4]);
            if (v2) {
                const float4 d0 = *reinterpret_cast<const float4*>(&v2[8 * lane]), d1 = *reinterpret_cast<const float4*>(&v2[8 * lane + 4]);
                c0.x -= f * d0.x; c0.y -= f * d0.y; c0.z -= f * d0.z; c0.w -= f * d0.w;
                c1.x -= f * d1.x; c1.y -= f * d1.y; c1.z -= f * d1.z; c1.w -= f * d1.w;
            }
            const int oc = lane < A16_LD / 8 ? lane : 0;
            const mb_real on = lane < A16_LD / 8 ? o.aci16_unscale : 0.f;
            cl_half8 a[RPW];
#pragma unroll
            for (int j = 0; j < RPW; ++j) a[j] = *reinterpret_cast<const cl_half8*>(&l_a16[(wave * RPW + j) * A16_LD + 8 * oc]);
#pragma unroll
            for (int j = 0; j < RPW; ++j)
                acc[j] = on * ((((float)a[j][0] * c0.x + (float)a[j][1] * c0.y) + ((float)a[j][2] * c0.z + (float)a[j][3] * c0.w)) +
                               (((float)a[j][4] * c1.x + (float)a[j][5] * c1.y) + ((float)a[j][6] * c1.z + (float)a[j][7] * c1.w)));
        } else {
        const int passes = ACI ? ACI_LD / 256 : (nq + 63) >> 6;
        for (int ps = 0; ps < passes; ++ps) {
            const int qd = lane + 64 * ps;
            if (ACI || qd < nq) {
                float4 c = *reinterpret_cast<const float4*>(&v1[4 * qd]);
                if (v2) { const float4 d = *reinterpret_cast<const float4*>(&v2[4 * qd]); c.x -= f * d.x; c.y -= f * d.y; c.z -= f * d.z; c.w -= f * d.w; }
                float4 a[RPW];
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    const int row = wave * RPW + j, rowc = row < n8g ? row : 0;
                    if (ACI) a[j] = *reinterpret_cast<const float4*>(&l_aci[rowc * ACI_LD + 4 * qd]);
                    else a[j] = *reinterpret_cast<const float4*>(o.aci8 + (size_t)(a0 + rowc) * o.ld8 + 4 * qd);
                }
#pragma unroll
                for (int j = 0; j < RPW; ++j) acc[j] += ((a[j].x * c.x + a[j].y * c.y) + a[j].z * c.z) + a[j].w * c.w;
            }
        }
        }
        const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
        mb_real b4[4], b2[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const mb_real send = h5 ? acc[i] : acc[i + 4], keep = h5 ? acc[i + 4] : acc[i]; b4[i] = keep + __shfl_xor(send, 32, 64); }
#pragma unroll
        for (int i = 0; i < 2; ++i) { const mb_real send = h4 ? b4[i] : b4[i + 2], keep = h4 ? b4[i + 2] : b4[i]; b2[i] = keep + __shfl_xor(send, 16, 64); }
        mb_real d;
        { const mb_real send = h3 ? b2[0] : b2[1], keep = h3 ? b2[1] : b2[0]; d = keep + __shfl_xor(send, 8, 64); }
        d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 1, 64);
        return d;
    };
    {   // (A8^+ cells)(own rows): the mean of r enters the coarse right-hand side as mean * cells -- one multiply per row instead of a pass
        for (int c = t; c < N8MAX; c += NT) zl[c] = c < n8 ? o.cnt8[c] : 0.f;
        __syncthreads();
        const mb_real acn = coarse_rows(zl, nullptr, 0.f);
        if ((lane & 7) == 0 && cr_row < n8g) l_acn[cr_row] = acn;
        __syncthreads();
    }
#ifdef FG_CL_CYCLES
    if (t < 16) l_ph[t] = 0;
#endif

    for (int sys = cluster; sys < o.B; sys += o.n_clusters) {
        if (!mb_active(o.dt, sys)) {
            if (g == 0 && t == 0) {
                flag_st(q.flags + sys, 3);
                q.info[sys].final_residual = 0.f; q.info[sys].used_iterations = -1; q.info[sys].converged = 1; q.info[sys].is_finite = 1;
                o.info_host[sys] = q.info[sys];
                o.its_host[sys] = 0;
                fg_poll_publish(o.poll, sys);
            }
            continue;
        }
        const size_t vb = (size_t)sys * N, sb = ((size_t)sys * CL_G + g) * S;
        const uint32_t base = __hip_atomic_load(o.epoch + sys, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t epoch = 0;
        bool near = false;   // until the cluster's workgroups have told each other where they run
        cl_u64* const box_env = o.box + (size_t)sys * 2 * CL_G * o.W;
        if (t == 0) l_fail = 0;
        // ---- matrix row and start vector of the thread's cells
        mb_real off[CPT][4], dg[CPT], r[CPT], x[CPT], p[CPT], s[CPT], w[CPT];
        mb_real sd = 0.f;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const float4 c = *reinterpret_cast<const float4*>(o.off4 + (sb + t + k * NT) * 4);
            off[k][0] = c.x; off[k][1] = c.y; off[k][2] = c.z; off[k][3] = c.w;
            dg[k] = o.diag[sb + t + k * NT];
            const bool ok = (vmask >> k) & 1u;
            const int cell = ok ? o.slot_cell[(size_t)g * S + t + k * NT] : 0;
            r[k] = ok ? q.rhs[vb + cell] : 0.f;
            x[k] = (ok && o.use_x0) ? q.x[vb + cell] : 0.f;
            p[k] = 0.f; s[k] = 0.f; w[k] = 0.f;
            sd += dg[k];
        }
        __syncthreads();   // l_fail cleared; the LDS arrays of the previous env are free

        // w_k = (P v)(member k) for the vector in zl (own slots + halo); members of `which` only
        auto stencil = [&](uint32_t which) {
            const char* vbase = reinterpret_cast<const char*>(zl);
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                if ((which >> k) & 1u) {
                    const uint32_t a0_ = nb[k].x & 0xffffu, a1_ = nb[k].x >> 16, a2_ = nb[k].y & 0xffffu, a3_ = nb[k].y >> 16;
                    const mb_real vc = zl[t + k * NT];
                    const mb_real v0 = *reinterpret_cast<const mb_real*>(vbase + a0_), v1 = *reinterpret_cast<const mb_real*>(vbase + a1_);
                    const mb_real v2 = *reinterpret_cast<const mb_real*>(vbase + a2_), v3 = *reinterpret_cast<const mb_real*>(vbase + a3_);
                    mb_real acc = dg[k] * vc;
                    acc += off[k][0] * v0; acc += off[k][1] * v1; acc += off[k][2] * v2; acc += off[k][3] * v3;
                    w[k] = acc;
                }
            }
        };

        // The solve as a sequence of STAGES with one exchange each (one copy of the exchange code):
        //   RESID   zl = x | halo                      -> r = rhs - P x                                      -> GATHER
        //   GATHER  Z8^T r, r.r, 1.r (, sum diag)       -> coarse residual of all aggregates, norms           -> P
        //   P       z = M (r - mean r) | halo, Z8^T r, r.z, sum z, r.r, 1.r -> verdict on x_it; beta, mean z  -> S (or RESID / out)
        //   S       w = P z~, p, s | Z8^T s, p.s, 1.s   -> alpha; x, r; coarse residual and mean by one step  -> P
        enum { ST_RESID = 0, ST_GATHER = 1, ST_P = 2, ST_S = 3 };
        int stage = o.use_x0 ? ST_RESID : ST_GATHER;
#ifdef FG_CL_CYCLES
        if (t == 0) l_tph = clock64();
#endif
        double rr = 0.0, sr = 0.0, sr_next = 0.0, rz = 0.0, rz_prev = 1.0, zmean = 0.0;
        // verdicts on rho = |r - mean r|^2 against tol^2 N: the RMS criterion of cg_solver_kernel.cu:100-106 without a square root
        // and a division per iteration in every thread
        const double thr2 = (double)o.tol * (double)o.tol * (double)N, acc2 = (double)o.accept_factor * (double)o.accept_factor * thr2;
        double rho = 0.0, best_rho = 1.0e300;
        mb_real inv_s = 1.f, beta = 0.f, alpha_c = 0.f;   // alpha_c: the step since l_r8 was exact (0 right after an exchange that carried it)
        int it = 0, best_it = 0, recoveries = 0, outcome = 0;   // outcome: 1 converged, 2 non-finite, 3 accepted on the kept iterate, 4 out of iterations / stalled, 5 a granule never came
        bool fresh = true, restarted = true, recovering = false, first_gather = true, have_best = false;
        mb_real* const bestx = o.bestx + sb;
        for (;;) {
            // the thread index is laundered once per trip: the 64-bit global addresses built from it (kept iterate, granule boxes, tables)
            // are invariants of this loop, and the compiler otherwise hoists all of them out of it and spills them (k_mbc_onchip's remedy)
            int tl = t;
            asm volatile("" : "+v"(tl));
            // likewise what it would derive from the matrix row once and keep (the decoded neighbour addresses: 32 registers for 16; the
            // reciprocal diagonal and the row sum: 16 more) -- recomputed where used, a handful of VALU operations per member
#pragma unroll
            for (int k = 0; k < CPT; ++k) asm volatile("" : "+v"(nb[k].x), "+v"(nb[k].y), "+v"(dg[k]));
            bool HALO = false, A8 = false, SC = false;
            mb_real a8val = 0.f;
            mb_real* a8dst = l_r8;
            mb_real ws[4] = {0.f, 0.f, 0.f, 0.f};
            if (stage == ST_RESID) {
#pragma unroll
                for (int k = 0; k < CPT; ++k) zl[t + k * NT] = x[k];
                HALO = true;
            } else if (stage == ST_GATHER) {
                mb_real s2 = 0.f, s1 = 0.f, r4 = 0.f;
#pragma unroll
                for (int k = 0; k < CPT; ++k) { s2 += r[k] * r[k]; s1 += r[k] * rsqn; r4 += r[k]; }
                a8val = cl_group_sum<GL>(r4);
                ws[0] = fg_wave_sum(s2); ws[1] = fg_wave_sum(s1); ws[2] = fg_wave_sum(sd);
                A8 = true; SC = true;
            } else if (stage == ST_P) {
                // ---- z = M (r - mean r).  Coarse level: rows [a0, a0 + n8g) of A8^+ (Z8^T r - mean * cells), with Z8^T r of the
                // current residual = the exact sums of the last exchange minus alpha * Z8^T s (one recurrence step) -- formed where
                // it is used, so nothing has to be written and waited for between the update and this
                CL_PH(0);
                const mb_real rm = PM ? (mb_real)sr_next * rsqn : 0.f;
                const mb_real e8 = coarse_rows(l_r8, alpha_c != 0.f ? l_s8 : nullptr, alpha_c);
                if ((lane & 7) == 0 && cr_row < n8g) l_e8[cr_row] = e8 - rm * l_acn[cr_row];
                CL_PH(1);
                // 4 x 4 level: the sum over the R lanes of the aggregate; the exact Z8^T r of the own aggregates goes out with z
                mb_real r4raw = 0.f, s2 = 0.f, s1 = 0.f;
#pragma unroll
                for (int k = 0; k < CPT; ++k) { r4raw += r[k]; s2 += r[k] * r[k]; s1 += r[k] * rsqn; }
                const mb_real r4 = cl_group_sum<R>(r4raw) - rm * cnt4;
                a8val = cl_group_sum<GL>(r4raw);
                __syncthreads();
                CL_PH(2);
                const mb_real corr = inv_s * (0.5f * r4 * d4g_t + (la < n8g ? l_e8[la] : 0.f));
                mb_real s_rz = 0.f, s_z = 0.f;
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const bool ok = (vmask >> k) & 1u;
                    const mb_real rt = ok ? r[k] - rm : 0.f;
                    const mb_real z = ok ? rt * __builtin_amdgcn_rcpf(dg[k]) + corr : 0.f;   // v_rcp_f32: a preconditioner needs no IEEE division
                    zl[t + k * NT] = z;
                    s_rz += rt * z; s_z += z;
                }
                ws[0] = fg_wave_sum(s_rz); ws[1] = fg_wave_sum(s_z); ws[2] = fg_wave_sum(s2); ws[3] = fg_wave_sum(s1);
                HALO = true; A8 = true; SC = true;
                CL_PH(3);
            } else {
                // ---- p = z~ + beta p; s = P z~ + beta s with P z~ = P z - mean z * (row sum): P z sits in w (the members without a
                // halo neighbour were done while stage P's granules travelled, the others behind its exchange)
                const mb_real zbar = PM ? (mb_real)zmean : 0.f;
                mb_real s_ps = 0.f, s_s = 0.f, s4 = 0.f;
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const mb_real zt = ((vmask >> k) & 1u) ? zl[t + k * NT] - zbar : 0.f;
                    const mb_real rs = (((dg[k] + off[k][0]) + off[k][1]) + off[k][2]) + off[k][3];
                    p[k] = zt + beta * p[k];
                    s[k] = (w[k] - zbar * rs) + beta * s[k];
                    s_ps += p[k] * s[k]; s_s += s[k] * rsqn; s4 += s[k];
                }
                a8val = cl_group_sum<GL>(s4);
                a8dst = l_s8;
                ws[0] = fg_wave_sum(s_ps); ws[1] = fg_wave_sum(s_s);
                A8 = true; SC = true;
                CL_PH(6);
            }
            // ---- the exchange: publish this workgroup's granules, collect the peers'.  HALO: the boundary values of zl; A8: `a8val` of
            // every aggregate leader -> a8dst[all aggregates]; SC: four sums (wave sums -> workgroup sums -> the cluster's, l_tot), added
            // in the order of the workgroups by the one thread that owns the sum, in every workgroup alike
            {
                ++epoch;
                const uint32_t tag = base + epoch;
                cl_u64* const bx = box_env + (size_t)(epoch & 1u) * CL_G * o.W;
                cl_u64* const mine = bx + (size_t)g * o.W;
                if (SC && lane == 0) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) l_wave[v][wave] = (double)ws[v];
                }
                __syncthreads();   // zl and the wave sums are written
                if (stage == ST_P) CL_PH(10);
                if (HALO && tl < n_out) cl_store(mine + tl, tag, __float_as_uint(zl[out_sl]), near);
                if (A8 && a8_leader) { cl_store(mine + o.n_out_max + la, tag, __float_as_uint(a8val), near); a8dst[a0 + la] = a8val; }
                // the four sums: by the LAST wave, sixteen lanes per sum -- lane 16 v + w adds wave w's part (a butterfly: every lane of
                // the sixteen ends with the workgroup's sum), lanes 16 v and 16 v + 1 publish its halves
                double own = 0.0;
                const bool sc_wave = SC && wave == NW - 1;
                const int sc_v = lane >> 4, sc_w = lane & 15;
                if (sc_wave) {
                    own = sc_w < NW ? l_wave[sc_v][sc_w] : 0.0;
#pragma unroll
                    for (int m = 1; m < 16; m <<= 1) own += __shfl_xor(own, m, 64);
                    const cl_u64 bits = (cl_u64)__double_as_longlong(own);
                    if (sc_w < 2) cl_store(mine + sc_at + 2 * sc_v + sc_w, tag, (uint32_t)(sc_w ? (bits >> 32) : bits), near);
                }
                if (SC && tl == NT - 65) { cl_store(mine + sc_at + 8, tag, my_xcc, near); l_xcc[g] = my_xcc; }
                if (stage == ST_P) CL_PH(11);
                // while the granules travel: the stencil of the members whose neighbours are all this workgroup's own
                if (stage == ST_P) { stencil(imask); CL_PH(12); }
                bool ok = true;
                if (HALO && tl < n_halo) {
                    uint32_t v = 0;
                    ok &= cl_poll(bx + halo_at, tag, o.abort_at + sys, base, v);
                    zl[S + tl] = __uint_as_float(v);
                }
                if (A8 && tl < n_rem) {
                    uint32_t v = 0;
                    ok &= cl_poll(bx + a8_at, tag, o.abort_at + sys, base, v);
                    a8dst[a8_A] = __uint_as_float(v);
                }
                if (sc_wave) {
                    // lane 16 v + c collects workgroup c's halves of sum v (one round trip for all of them), the sixteen lanes then add
                    // the four parts in the order of the workgroups
                    double part = own;
                    if (sc_w < CL_G && sc_w != g) {
                        const cl_u64* src = bx + (size_t)sc_w * o.W + sc_at + 2 * sc_v;
                        unsigned spins = 0;
                        for (;;) {
                            const cl_u64 lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const cl_u64 hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((uint32_t)(lo >> 32) == tag && (uint32_t)(hi >> 32) == tag) {
                                part = __longlong_as_double((long long)(((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull)));
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins >= CL_SPIN_LIMIT || ((spins & 255u) == 0u && __hip_atomic_load(o.abort_at + sys, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == base)) { ok = false; break; }
                        }
                    }
                    double sum = 0.0;
#pragma unroll
                    for (int c = 0; c < CL_G; ++c) sum += __shfl(part, (lane & 48) + c, 64);
                    if (sc_w == 0) {
                        l_tot[sc_v] = sum;
                        // the one division of the exchange, by the thread that holds its numerator or denominator -- v_rcp_f64 (2^-26) and a
                        // multiply: the quotient is rounded to fp32 anyway
                        if (stage == ST_P && sc_v == 0) l_f[0] = (mb_real)(sum * __builtin_amdgcn_rcp(rz_prev));   // beta = r.z / (r.z)_prev
                        if (stage == ST_S && sc_v == 0) l_f[1] = (mb_real)(rz * __builtin_amdgcn_rcp(sum));        // alpha = r.z / p.s
                    }
                }
                if (SC && tl == NT - 65) {
#pragma unroll
                    for (int c = 0; c < CL_G; ++c)
                        if (c != g) { uint32_t v = 0; ok &= cl_poll(bx + (size_t)c * o.W + sc_at + 8, tag, o.abort_at + sys, base, v); l_xcc[c] = v; }
                }
                if (!ok) l_fail = 1;
                if (stage == ST_P) CL_PH(13);
                __syncthreads();
                if (SC && !near && o.allow_near) {
                    bool same = true;
#pragma unroll
                    for (int c = 0; c < CL_G; ++c) same = same && l_xcc[c] == my_xcc;
                    near = same;
                }
            }
            CL_PH(stage == ST_P ? 4 : (stage == ST_S ? 7 : 9));
            if (l_fail) { outcome = 5; break; }
            if (stage == ST_RESID) {
                stencil(0xffu);
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const bool ok = (vmask >> k) & 1u;
                    const int cell = ok ? o.slot_cell[(size_t)g * S + tl + k * NT] : 0;
                    r[k] = ok ? q.rhs[vb + cell] - w[k] : 0.f;
                }
                __syncthreads();   // the stencil's reads of zl before the next stage rewrites it
                stage = ST_GATHER;
            } else if (stage == ST_GATHER) {
                rr = cl_uni(l_tot[0]); sr = PM ? cl_uni(l_tot[1]) : 0.0;
                if (first_gather) { inv_s = cl_uni((mb_real)((double)o.geom_diag_sum / l_tot[2])); first_gather = false; }
                sr_next = sr;
                alpha_c = 0.f;
                fresh = true; restarted = true;
                stage = ST_P;
            } else if (stage == ST_P) {
                rz = cl_uni(l_tot[0]);
                const double zsum = cl_uni(l_tot[1]);
                rr = cl_uni(l_tot[2]); sr = PM ? cl_uni(l_tot[3]) : 0.0;
                alpha_c = 0.f;   // l_r8 is exact again
                // ---- the verdict on x_it / r_it (k_mbc_onchip's rules)
                bool need_residual = false;
                rho = cl_uni(rr - sr * sr);   // |r - (1.r) 1 / N|^2
                const bool finite = isfinite(rho) && rho >= 0.0;
                if (!(rho >= thr2)) {
                    if (finite) { outcome = 1; break; }
                    // the recurrence broke down (p.Pp <= 0 or overflow on the non-symmetric matrix): back to the kept iterate
                    if (recovering || recoveries >= 3 || it + o.check_every >= o.max_iterations) { outcome = 2; break; }
                    ++recoveries;
#pragma unroll
                    for (int k = 0; k < CPT; ++k) { const mb_real v = bestx[tl + k * NT]; x[k] = isfinite(v) ? v : 0.f; }
                    need_residual = true; recovering = true;
                } else {
                    recovering = false;
                    // keep x_it when it beats the kept iterate by 2x (or at all inside the acceptance band): returnBestResult
                    if (it == 0 || rho < 0.25 * best_rho || (rho < acc2 && rho < best_rho)) {
                        best_rho = rho; best_it = it; have_best = true;
#pragma unroll
                        for (int k = 0; k < CPT; ++k) bestx[tl + k * NT] = x[k];
                    }
                    if (it > 0 && it % o.check_every == 0) {   // the cadence of k_mbs_check in the chunked solver
                        if (o.accept_factor > 0.f && best_rho <= acc2 && (it - 1) - best_it >= o.accept_window) { outcome = 3; break; }
                        if (o.stall_limit > 0 && (it - 1) - best_it > o.stall_limit) { outcome = 4; break; }
                    }
                    if (it >= o.max_iterations) { outcome = 4; break; }
                    if (it > 0 && it % o.restart_every == 0 && !restarted) need_residual = true;   // residualResetSteps (cg_solver_kernel.cu:281-300)
                }
                if (need_residual) { stage = ST_RESID; continue; }
                restarted = false;
                CL_PH(14);
                stencil(vmask & ~imask);   // the members with a neighbour in the halo
                CL_PH(5);
                beta = fresh ? 0.f : cl_uni(l_f[0]);
                zmean = cl_uni(zsum * inv_n);
                stage = ST_S;
            } else {
                const double ssum = cl_uni(l_tot[1]);
                const mb_real alpha = cl_uni(l_f[1]);
#pragma unroll
                for (int k = 0; k < CPT; ++k) { x[k] += alpha * p[k]; r[k] -= alpha * s[k]; }
                alpha_c = alpha;
                sr_next = cl_uni(sr - (double)alpha * ssum);   // the mean of the new residual: one recurrence step from the exact value stage P carried
                rz_prev = rz;
                fresh = false;
                ++it;
                stage = ST_P;
                CL_PH(8);
            }
        }
        // ---- hand back: the last iterate when converged, the kept one otherwise (k_mbs_restore_best)
        const bool use_best = (outcome == 3 || outcome == 4 || (outcome == 2 && have_best));
        if (outcome != 5) {
#pragma unroll
            for (int k = 0; k < CPT; ++k)
                if ((vmask >> k) & 1u) q.x[vb + o.slot_cell[(size_t)g * S + t + k * NT]] = use_best ? bestx[t + k * NT] : x[k];
        }
        if (t == 0) {
            __hip_atomic_fetch_max(o.epoch + sys, base + epoch + 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (outcome == 5) __hip_atomic_store(o.abort_at + sys, base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (g == 0 && t == 0) {
            flag_st(q.flags + sys, outcome == 2 ? 2 : (outcome == 3 ? 5 : 1));
            q.info[sys].final_residual = mb_rms(use_best ? best_rho : rho, N);
            q.info[sys].used_iterations = use_best ? best_it : it;
            q.info[sys].converged = (outcome == 1 || outcome == 3) ? 1 : 0;
            q.info[sys].is_finite = (outcome != 2) ? 1 : 0;
            q.best_it[sys] = it;   // total iterations run (profiling: the host sums them)
            o.info_host[sys] = q.info[sys];
            o.its_host[sys] = outcome == 5 ? -77 : it;   // -77: the host repeats the solve with the one-workgroup kernels
#ifdef FG_CL_CYCLES
            if (sys == 0 && o.dbg_out) { for (int k = 0; k < 10; ++k) o.dbg_out[k] = (l_ph[k] & 0xffffffffull) | ((k < 6 ? l_ph[10 + k] : 0ull) << 32); o.dbg_out[10] = near ? 1 : 0; o.dbg_out[11] = (unsigned long long)it; }
#endif
            fg_poll_publish(o.poll, sys);
        }
        __syncthreads();   // every thread is done with this env's LDS before the next env's start writes it
    }
}

// (k_mbs_begin of fg_mb_krylov.hip, for this translation unit: the flags and info words of the systems of a solve)
__global__ void k_mbs_begin_cl(const mb_real* __restrict__ dt, MbSolve q, int nsys) {
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

// ---------------------------------------------------------------------------------------------------------------
// The velocity systems' Jacobi sweeps (mb_jacobi, fg_mb_krylov.hip: k_mbj_sweep_env, one launch per sweep over the neighbour table,
// 12 us each at 64 envs x 14 k cells -- 18 % of a cylinder step) by the same clusters: a workgroup keeps the matrix row, the
// right-hand sides and the iterate of its cells (both components: they share the row) in registers, the iterate it gathers from in
// LDS with the halo behind it, and ONE exchange per sweep carries the halo -- and, behind a measuring sweep, the residual sums the
// verdict is taken from.  Same sweep, same arithmetic per cell (x = (b - sum_f o_f x_f) / d in face order, the residual of the
// iterate a sweep started from as d (x_new - x_old)), same check points (12, 14, ... 32 sweeps), same verdict per system, so an
// env's iterate is what the launched sweeps give it; the give-up rule the host applies to the batch (contraction per sweep above
// 0.85, or more than 32 sweeps needed) is applied per env here, and the host hands the batch to BiCGStab when any env gave up.
// ---------------------------------------------------------------------------------------------------------------
struct ClJParams {
    const int32_t* slot_cell; const uint2* nbr; const uint32_t* tinfo; const int32_t* out_slot; const uint32_t* halo_src;
    int n_out[CL_G], n_halo[CL_G];
    int n_out_max, n_halo_max, WJ;       // WJ: granules of one box = 2 n_out_max + CL_NSC
    cl_u64* box; uint32_t* epoch; uint32_t* abort_at;
    int B, N, n_clusters, allow_near, use_x0;
    mb_real tol;
    fg_solve_info* info_host; int32_t* flags_host; mb_real* res2_host; FgPollOut poll;
};

template <int NT>
__global__ __launch_bounds__(NT) void k_mbj_cluster(MbSolve q, ClJParams o) {
    constexpr int CPT = 8, S = CPT * NT, HCAP = NT, NW = NT / 64, NC = 2, F = 4;
    constexpr int FIRST = 12, STEP = 2, CHECKS = 11;     // mb_jacobi's check points
    static_assert(S + HCAP <= 16383, "16-bit byte offsets into the LDS vectors");
    __shared__ __attribute__((aligned(16))) mb_real xl[NC][S + HCAP];
    __shared__ double l_wave[NC][NW];
    __shared__ double l_tot[NC];
    __shared__ uint32_t l_xcc[CL_G];
    __shared__ int l_fail;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int cluster, g;
    {
        const int b = blockIdx.x, ncl = o.n_clusters;
        if ((ncl & 7) == 0) { const int xcd = b & 7, j = b >> 3; cluster = (j / CL_G) * 8 + xcd; g = j % CL_G; }
        else { cluster = b / CL_G; g = b % CL_G; }
    }
    const int N = o.N;
    const uint32_t vmask = o.tinfo[g * NT + t] & 0xffu;
    uint2 nb[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) nb[k] = o.nbr[(size_t)g * S + t + k * NT];
    const int n_out = o.n_out[g], n_halo = o.n_halo[g];
    const int out_sl = t < n_out ? o.out_slot[g * o.n_out_max + t] : 0;
    uint32_t halo_at = 0;
    if (t < n_halo) { const uint32_t src = o.halo_src[g * o.n_halo_max + t]; halo_at = (src >> 16) * (uint32_t)o.WJ + 2u * (src & 0xffffu); }
    const int sc_at = 2 * o.n_out_max;
    const uint32_t my_xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xfu;

    for (int b = cluster; b < o.B; b += o.n_clusters) {
        // systems b NC + c; k_mbs_begin has set their flags (0 live, 3 the env sits out) and cleared their info
        bool live[NC], any = false;
#pragma unroll
        for (int c = 0; c < NC; ++c) { live[c] = flag_ld(q.flags + (b * NC + c)) == 0; any = any || live[c]; }
        if (!any) {
            if (g == 0 && t < NC) {
                const int sy = b * NC + t;
                o.info_host[sy] = q.info[sy]; o.flags_host[sy] = flag_ld(q.flags + sy); o.res2_host[2 * sy] = -1.f; o.res2_host[2 * sy + 1] = -1.f;
                fg_poll_publish(o.poll, sy);
            }
            continue;
        }
        const uint32_t base = __hip_atomic_load(o.epoch + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t epoch = 0;
        bool near = false;
        cl_u64* const box_env = o.box + (size_t)b * 2 * CL_G * o.WJ;
        if (t == 0) l_fail = 0;
        // ---- matrix row (cell order: gathered once per solve), right-hand sides, start vector
        mb_real off[CPT][F], dg[CPT], rd[CPT], rhs[CPT][NC], x[CPT][NC];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const bool ok = (vmask >> k) & 1u;
            const int cell = ok ? o.slot_cell[(size_t)g * S + t + k * NT] : 0;
            const uint32_t self = 4u * (uint32_t)(t + k * NT);
            const uint32_t na[F] = {nb[k].x & 0xffffu, nb[k].x >> 16, nb[k].y & 0xffffu, nb[k].y >> 16};
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const mb_real v = ok ? q.off[((size_t)b * F + f) * N + cell] : 0.f;
                off[k][f] = (ok && na[f] != self) ? v : 0.f;      // a prescribed face points at the cell itself: no matrix entry (k_mbj_sweep_env skips it)
            }
            dg[k] = ok ? q.diag[(size_t)b * N + cell] : 1.f;
            rd[k] = (mb_real)1 / dg[k];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                rhs[k][c] = ok ? q.rhs[((size_t)b * NC + c) * N + cell] : 0.f;
                x[k][c] = (ok && o.use_x0) ? q.x[((size_t)b * NC + c) * N + cell] : 0.f;
            }
        }
        __syncthreads();
        int sweeps = 0, checks = 0, used[NC] = {0, 0}, outcome = 0;     // outcome: 1 every live system settled, 2 gave up, 5 a granule never came
        double res_now[NC] = {-1.0, -1.0}, res_prev[NC] = {-1.0, -1.0};
        bool fin[NC] = {true, true}, conv[NC] = {false, false};
        for (;;) {
            int tl = t;
            asm volatile("" : "+v"(tl));
            const bool from_zero = sweeps == 0 && !o.use_x0;
            // what the sweep before this one measured goes out with this sweep's halo
            const bool measured = sweeps >= FIRST - 2 && !(sweeps & 1);     // sweeps 9, 11, ... measure; their sums travel at sweep counts 10, 12, ...
            if (!from_zero) {
                // ---- the exchange: x of the boundary cells (both components), the sums of the last measuring sweep
#pragma unroll
                for (int k = 0; k < CPT; ++k) { xl[0][t + k * NT] = x[k][0]; xl[1][t + k * NT] = x[k][1]; }
                ++epoch;
                const uint32_t tag = base + epoch;
                cl_u64* const bx = box_env + (size_t)(epoch & 1u) * CL_G * o.WJ;
                cl_u64* const mine = bx + (size_t)g * o.WJ;
                __syncthreads();
                if (tl < n_out) { cl_store(mine + 2 * tl, tag, __float_as_uint(xl[0][out_sl]), near); cl_store(mine + 2 * tl + 1, tag, __float_as_uint(xl[1][out_sl]), near); }
                const bool sc_wave = wave == NW - 1;
                const int sc_v = (lane >> 4) & 1, sc_w = lane & 15;
                double own = 0.0;
                if (sc_wave && measured && lane < 32) {
                    own = sc_w < NW ? l_wave[sc_v][sc_w] : 0.0;
#pragma unroll
                    for (int m = 1; m < 16; m <<= 1) own += __shfl_xor(own, m, 64);
                    const cl_u64 bits = (cl_u64)__double_as_longlong(own);
                    if (sc_w < 2) cl_store(mine + sc_at + 2 * sc_v + sc_w, tag, (uint32_t)(sc_w ? (bits >> 32) : bits), near);
                }
                if (!near && tl == NT - 65) { cl_store(mine + sc_at + 8, tag, my_xcc, near); l_xcc[g] = my_xcc; }
                bool ok = true;
                if (tl < n_halo) {
                    uint32_t v0 = 0, v1 = 0;
                    ok &= cl_poll(bx + halo_at, tag, o.abort_at + b, base, v0);
                    ok &= cl_poll(bx + halo_at + 1, tag, o.abort_at + b, base, v1);
                    xl[0][S + tl] = __uint_as_float(v0); xl[1][S + tl] = __uint_as_float(v1);
                }
                if (sc_wave && measured && lane < 32) {
                    double part = own;
                    if (sc_w < CL_G && sc_w != g) {
                        const cl_u64* src = bx + (size_t)sc_w * o.WJ + sc_at + 2 * sc_v;
                        unsigned spins = 0;
                        for (;;) {
                            const cl_u64 lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const cl_u64 hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((uint32_t)(lo >> 32) == tag && (uint32_t)(hi >> 32) == tag) {
                                part = __longlong_as_double((long long)(((hi & 0xffffffffull) << 32) | (lo & 0xffffffffull)));
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins >= CL_SPIN_LIMIT || ((spins & 255u) == 0u && __hip_atomic_load(o.abort_at + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == base)) { ok = false; break; }
                        }
                    }
                    double sum = 0.0;
#pragma unroll
                    for (int c = 0; c < CL_G; ++c) sum += __shfl(part, (lane & 16) + c, 64);
                    if (sc_w == 0) l_tot[sc_v] = sum;
                }
                if (!near && tl == NT - 65) {
#pragma unroll
                    for (int c = 0; c < CL_G; ++c)
                        if (c != g) { uint32_t v = 0; ok &= cl_poll(bx + (size_t)c * o.WJ + sc_at + 8, tag, o.abort_at + b, base, v); l_xcc[c] = v; }
                }
                if (!ok) l_fail = 1;
                __syncthreads();
                if (!near && o.allow_near) {
                    bool same = true;
#pragma unroll
                    for (int c = 0; c < CL_G; ++c) same = same && l_xcc[c] == my_xcc;
                    near = same;
                }
                if (l_fail) { outcome = 5; break; }
                // ---- behind a measuring sweep: its residual; at a check point (12, 14, ... sweeps done) the verdict per system
                if (measured) {
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        if (live[c]) { res_prev[c] = res_now[c]; res_now[c] = (double)mb_rms(cl_uni(l_tot[c]), N); }
                    if (sweeps >= FIRST) {
                        ++checks;
                        bool all = true, bad = false;
                        double need = 0.0;
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            if (!live[c]) continue;
                            const mb_real now = (mb_real)res_now[c];
                            used[c] = sweeps;
                            if (!(now >= o.tol)) { live[c] = false; fin[c] = isfinite(now); conv[c] = fin[c]; continue; }
                            all = false;
                            // mb_jacobi's rule for going on, per env: the contraction per sweep from the two measuring sweeps (two apart)
                            const double r1 = res_now[c], r0 = res_prev[c];
                            if (r0 > 0.0 && r1 > 0.0) {
                                const double cs = sqrt(r1 / r0);
                                if (!(cs < 0.85)) bad = true;
                                else { const double m = log((double)o.tol / r1) / log(cs); need = m > need ? m : need; }
                            } else need = need > 4.0 ? need : 4.0;
                        }
                        if (all) { outcome = 1; break; }
                        if (bad || !(need < 1.0e6)) { outcome = 2; break; }
                        const int more = 1 + (int)(ceil(need) - 1) / STEP;
                        if (checks + more > CHECKS) { outcome = 2; break; }
                    }
                }
            }
            // ---- the sweep: x = (b - sum_f o_f x_f) / d, the residual of the iterate it started from as d (x_new - x_old)
            const bool measure = sweeps >= FIRST - 3 && (sweeps & 1);
            mb_real part[NC] = {0.f, 0.f};
            const char* b0 = reinterpret_cast<const char*>(xl[0]);
            const char* b1 = reinterpret_cast<const char*>(xl[1]);
#pragma unroll
            for (int k = 0; k < CPT; ++k) {
                mb_real acc0 = rhs[k][0], acc1 = rhs[k][1];
                if (!from_zero) {
                    const uint32_t na[F] = {nb[k].x & 0xffffu, nb[k].x >> 16, nb[k].y & 0xffffu, nb[k].y >> 16};
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        acc0 -= off[k][f] * *reinterpret_cast<const mb_real*>(b0 + na[f]);
                        acc1 -= off[k][f] * *reinterpret_cast<const mb_real*>(b1 + na[f]);
                    }
                }
                const bool ok = (vmask >> k) & 1u;
                if (live[0] && ok) { const mb_real xn = acc0 * rd[k]; const mb_real r = dg[k] * (xn - x[k][0]); part[0] += r * r; x[k][0] = xn; }
                if (live[1] && ok) { const mb_real xn = acc1 * rd[k]; const mb_real r = dg[k] * (xn - x[k][1]); part[1] += r * r; x[k][1] = xn; }
            }
            if (measure) {
                const mb_real w0 = fg_wave_sum(part[0]), w1 = fg_wave_sum(part[1]);
                if (lane == 0) { l_wave[0][wave] = (double)w0; l_wave[1][wave] = (double)w1; }
            }
            ++sweeps;
            __syncthreads();   // every gather of this sweep is done before the next sweep's iterate overwrites the LDS vectors
        }
        // ---- hand back: the iterate of every system (a system that stopped kept its own)
        if (outcome != 5) {
#pragma unroll
            for (int k = 0; k < CPT; ++k)
                if ((vmask >> k) & 1u) {
                    const int cell = o.slot_cell[(size_t)g * S + t + k * NT];
#pragma unroll
                    for (int c = 0; c < NC; ++c) q.x[((size_t)b * NC + c) * N + cell] = x[k][c];
                }
        }
        if (t == 0) {
            __hip_atomic_fetch_max(o.epoch + b, base + epoch + 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (outcome == 5) __hip_atomic_store(o.abort_at + b, base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (g == 0 && t < NC) {
            const int c = t, sy = b * NC + c;
            const bool was_live = flag_ld(q.flags + sy) == 0;
            if (was_live) {
                const bool stopped = c == 0 ? !live[0] : !live[1];
                const bool f_ = c == 0 ? fin[0] : fin[1], cv = c == 0 ? conv[0] : conv[1];
                q.info[sy].final_residual = (mb_real)(c == 0 ? res_now[0] : res_now[1]);
                q.info[sy].used_iterations = stopped ? (c == 0 ? used[0] : used[1]) : sweeps;
                if (stopped) { q.info[sy].converged = cv ? 1 : 0; q.info[sy].is_finite = f_ ? 1 : 0; flag_st(q.flags + sy, f_ ? 1 : 2); }
            }
            o.info_host[sy] = q.info[sy];
            o.flags_host[sy] = outcome == 5 ? -77 : flag_ld(q.flags + sy);
            o.res2_host[2 * sy] = (mb_real)(c == 0 ? res_now[0] : res_now[1]); o.res2_host[2 * sy + 1] = (mb_real)(c == 0 ? res_prev[0] : res_prev[1]);
            fg_poll_publish(o.poll, sy);
        }
        __syncthreads();
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Host: tables of the cluster layout, built once per mesh behind the multilevel tables (fg_mb_set_multilevel).
// ---------------------------------------------------------------------------------------------------------------
int mb_cluster_build(fg_mb_state* s, int n4, int n8, const int32_t* rect4_host, const uint16_t* parent4, const uint2* child8,
                     const mb_real* rd4, const mb_real* aci8_padded) {
    s->cl_on = false;
#if FG_MB_F64
    (void)n4; (void)n8; (void)rect4_host; (void)parent4; (void)child8; (void)rd4; (void)aci8_padded;
    return FG_OK;
#else
    if (s->d != 2 || s->cl_mode == 0 || n8 < CL_G || n8 > 1024 || s->N >= 65535) return FG_OK;
    for (int a = 0; a < n4; ++a)
        if (rect4_host[4 * a + 1] > 4 || rect4_host[4 * a + 2] > 4) return FG_OK;
    const int G = CL_G, N = s->N;
    // ---- cluster order of the 8 x 8 aggregates: sorted along the mesh's longer direction, so that a contiguous range is a slab and
    // what crosses between workgroups is a cut through the shorter one (index order -- block by block, row by row -- gave the
    // cylinder mesh 904 halo cells per workgroup, a quarter of what it owns)
    std::vector<int> cells8(n8, 0);
    std::vector<double> cx(n8, 0.0), cy(n8, 0.0);
    {
        std::vector<double> px(N, 0.0), py(N, 0.0);
        for (const MbBlock& b : s->blocks) {
            const int nx = b.size[0], ny = b.size[1], vx = nx + 1, vy = ny + 1;
            for (int iy = 0; iy < ny; ++iy)
                for (int ix = 0; ix < nx; ++ix) {
                    double sx = 0.0, sy = 0.0;
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx) {
                            sx += b.coords[((size_t)0 * vy + iy + dy) * vx + ix + dx];
                            sy += b.coords[((size_t)1 * vy + iy + dy) * vx + ix + dx];
                        }
                    px[b.offset + iy * nx + ix] = 0.25 * sx; py[b.offset + iy * nx + ix] = 0.25 * sy;
                }
        }
        for (int a = 0; a < n4; ++a) {
            const int cell0 = rect4_host[4 * a], w = rect4_host[4 * a + 1], h = rect4_host[4 * a + 2], stride = rect4_host[4 * a + 3], A = parent4[a];
            for (int dy = 0; dy < h; ++dy)
                for (int dx = 0; dx < w; ++dx) { cx[A] += px[cell0 + dy * stride + dx]; cy[A] += py[cell0 + dy * stride + dx]; }
            cells8[A] += w * h;
        }
        for (int A = 0; A < n8; ++A) { cx[A] /= std::max(cells8[A], 1); cy[A] /= std::max(cells8[A], 1); }
    }
    std::vector<int> order(n8);   // order[A'] = aggregate (in the numbering of the multilevel tables) at cluster position A'
    for (int A = 0; A < n8; ++A) order[A] = A;
    {
        const auto mmx = std::minmax_element(cx.begin(), cx.end()), mmy = std::minmax_element(cy.begin(), cy.end());
        const std::vector<double>& key = (*mmx.second - *mmx.first) >= (*mmy.second - *mmy.first) ? cx : cy;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    }
    // ---- contiguous ranges of the order with about the same number of cells
    int first[CL_G + 1];
    {
        first[0] = 0;
        long done = 0; int A = 0;
        for (int g = 0; g < G; ++g) {
            const long want = ((long)N * (g + 1)) / G;
            while (A < n8 && (done < want || A < first[g] + 1) && n8 - A > G - 1 - g) { done += cells8[order[A]]; ++A; }
            first[g + 1] = (g + 1 == G) ? n8 : A;
        }
    }
    int n8g_max = 0;
    for (int g = 0; g < G; ++g) n8g_max = std::max(n8g_max, first[g + 1] - first[g]);
    // eight members per thread; threads per workgroup: the smallest instance whose threads cover the largest range
    const int cpt = 8;
    int nt = 0;
    if (n8g_max * 8 <= 512) nt = 512;
    else if (n8g_max * 8 <= 768) nt = 768;
    else if (n8g_max * 8 <= 1024) nt = 1024;
    else return FG_OK;
    const int S = cpt * nt, R = 16 / cpt, GL = 4 * R, rows_pt = cpt / 4;
    std::vector<int32_t> slot_cell((size_t)G * S, -1);
    std::vector<uint16_t> cell_slot(N, 0xffff);
    std::vector<int> cell_wg(N, -1);
    std::vector<mb_real> d4t((size_t)G * nt, 0.f), cnt8(n8, 0.f);
    std::vector<uint32_t> tinfo((size_t)G * nt, 0u);
    for (int Ap = 0; Ap < n8; ++Ap) cnt8[Ap] = (mb_real)cells8[order[Ap]];
    for (int g = 0; g < G; ++g) {
        for (int Ap = first[g]; Ap < first[g + 1]; ++Ap) {
            const int la = Ap - first[g], A = order[Ap];
            const unsigned words[2] = {child8[A].x, child8[A].y};
            for (int c = 0; c < 4; ++c) {
                const unsigned a = (words[c >> 1] >> (16 * (c & 1))) & 0xffffu;
                if (a == 0xffffu) continue;
                const int cell0 = rect4_host[4 * a], w = rect4_host[4 * a + 1], h = rect4_host[4 * a + 2], stride = rect4_host[4 * a + 3];
                for (int part = 0; part < R; ++part) {
                    const int t = la * GL + c * R + part;
                    uint32_t mask = 0;
                    for (int k = 0; k < cpt; ++k) {
                        const int dy = part * rows_pt + k / 4, dx = k % 4;
                        if (dy >= h || dx >= w) continue;
                        const int cell = cell0 + dy * stride + dx, slot = t + nt * k;
                        slot_cell[(size_t)g * S + slot] = cell;
                        cell_slot[cell] = (uint16_t)(g * S + slot);
                        cell_wg[cell] = g;
                        mask |= 1u << k;
                    }
                    tinfo[(size_t)g * nt + t] = mask | ((uint32_t)(w * h) << 16);
                    d4t[(size_t)g * nt + t] = rd4[a];
                }
            }
        }
    }
    for (int i = 0; i < N; ++i)
        if (cell_wg[i] < 0) return FG_OK;   // a cell no aggregate owns: the mesh keeps the one-workgroup kernels
    // ---- halo: neighbours owned by another workgroup; a producer publishes every such cell once
    std::vector<std::vector<int>> out_cells(G), halo_cells(G);
    std::vector<int> out_index(N, -1);
    std::vector<std::vector<int>> halo_index(G, std::vector<int>(N, -1));
    for (int g = 0; g < G; ++g)
        for (int sl = 0; sl < S; ++sl) {
            const int cell = slot_cell[(size_t)g * S + sl];
            if (cell < 0) continue;
            bool interior = true;
            for (int f = 0; f < 4; ++f) {
                const int32_t n = s->h_nbr[(size_t)f * N + cell];
                if (n < 0 || cell_wg[n] == g) continue;
                interior = false;
                if (halo_index[g][n] < 0) { halo_index[g][n] = (int)halo_cells[g].size(); halo_cells[g].push_back(n); }
                if (out_index[n] < 0) { const int gp = cell_wg[n]; out_index[n] = (int)out_cells[gp].size(); out_cells[gp].push_back(n); }
            }
            if (interior) tinfo[(size_t)g * nt + sl % nt] |= 1u << (8 + sl / nt);   // its stencil needs no granule
        }
    int n_out_max = 1, n_halo_max = 1;
    for (int g = 0; g < G; ++g) { n_out_max = std::max(n_out_max, (int)out_cells[g].size()); n_halo_max = std::max(n_halo_max, (int)halo_cells[g].size()); }
    if (n_halo_max > nt || n_out_max > nt || n8 - (first[1] - first[0]) > nt || 16383 - S < nt) return FG_OK;   // one granule of each kind per thread
    for (int g = 0; g < G; ++g) if (n8 - (first[g + 1] - first[g]) > nt) return FG_OK;
    std::vector<int32_t> out_slot((size_t)G * n_out_max, 0);
    std::vector<uint32_t> halo_src((size_t)G * n_halo_max, 0u);
    for (int g = 0; g < G; ++g) {
        for (size_t k = 0; k < out_cells[g].size(); ++k) out_slot[(size_t)g * n_out_max + k] = cell_slot[out_cells[g][k]] - g * S;
        for (size_t h = 0; h < halo_cells[g].size(); ++h) {
            const int n = halo_cells[g][h];
            halo_src[(size_t)g * n_halo_max + h] = ((uint32_t)cell_wg[n] << 16) | (uint32_t)out_index[n];
        }
    }
    // neighbours as BYTE offsets into the workgroup's LDS vector: an owned cell's slot, a halo position behind the slots, or -- a
    // prescribed face, every face of a hole -- the slot itself (its coefficient is zero: k_mb_pmatrix)
    std::vector<uint2> nbr((size_t)G * S);
    for (int g = 0; g < G; ++g)
        for (int sl = 0; sl < S; ++sl) {
            const int cell = slot_cell[(size_t)g * S + sl];
            unsigned v[4];
            for (int f = 0; f < 4; ++f) {
                const int32_t n = cell >= 0 ? s->h_nbr[(size_t)f * N + cell] : -1;
                unsigned idx = (unsigned)sl;
                if (n >= 0) idx = cell_wg[n] == g ? (unsigned)(cell_slot[n] - g * S) : (unsigned)(S + halo_index[g][n]);
                v[f] = 4u * idx;
            }
            nbr[(size_t)g * S + sl] = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
        }
    // the coarse inverse in cluster order (rows and columns)
    const int ld = (n8 + 3) & ~3;
    std::vector<mb_real> acip((size_t)n8 * ld, 0.f);
    for (int rp = 0; rp < n8; ++rp)
        for (int cp = 0; cp < n8; ++cp) acip[(size_t)rp * ld + cp] = aci8_padded[(size_t)order[rp] * ld + order[cp]];
    // ... and as fp16 (round to nearest even) behind a power-of-two scale that puts the largest entry at 2^14 .. 2^15
    const int ld16 = (n8 + 7) & ~7;
    std::vector<uint16_t> aci16((size_t)n8 * ld16, 0);
    float unscale = 1.f;
    {
        float amax = 0.f;
        for (const mb_real v : acip) amax = std::max(amax, (float)fabs((double)v));
        int e = 0;
        if (amax > 0.f) { (void)frexpf(amax, &e); }            // amax = m 2^e, m in [0.5, 1)
        const float scale = ldexpf(1.f, 15 - e);               // amax * scale in [2^14, 2^15)
        unscale = ldexpf(1.f, e - 15);
        auto to_half = [](float f) -> uint16_t {
            uint32_t x; memcpy(&x, &f, 4);
            const uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
            if (x >= 0x47800000u) return (uint16_t)(sign | 0x7bffu);                 // beyond the largest half: clamp (never reached here)
            if (x < 0x38800000u) {                                                   // subnormal half (or zero)
                if (x < 0x33000000u) return (uint16_t)sign;
                const int shift = 126 - (int)(x >> 23);                             // 14 .. 24
                uint32_t m = (x & 0x7fffffu) | 0x800000u;
                const uint32_t half = m >> (shift), rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
                return (uint16_t)(sign | (half + ((rem > mid || (rem == mid && (half & 1u))) ? 1u : 0u)));
            }
            const uint32_t h = ((x - 0x38000000u) >> 13), rem = x & 0x1fffu;
            return (uint16_t)(sign | (h + ((rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ? 1u : 0u)));
        };
        for (int rp = 0; rp < n8; ++rp)
            for (int cp = 0; cp < n8; ++cp) aci16[(size_t)rp * ld16 + cp] = to_half((float)acip[(size_t)rp * ld + cp] * scale);
    }
    const int W = n_out_max + n8g_max + CL_NSC;
    if (int rc = mb_alloc(s, &s->cl_aci16, aci16.size())) return rc;
    FG_HIP_CHECK(hipMemcpy(s->cl_aci16, aci16.data(), sizeof(uint16_t) * aci16.size(), hipMemcpyHostToDevice));
    s->cl_aci16_unscale = unscale;
    if (int rc = mb_alloc(s, &s->cl_slot_cell, (size_t)G * S)) return rc;
    if (int rc = mb_alloc(s, &s->cl_cell_slot, (size_t)N)) return rc;
    if (int rc = mb_alloc(s, &s->cl_nbr, (size_t)G * S)) return rc;
    if (int rc = mb_alloc(s, &s->cl_d4g, (size_t)G * nt)) return rc;
    if (int rc = mb_alloc(s, &s->cl_tinfo, (size_t)G * nt)) return rc;
    if (int rc = mb_alloc(s, &s->cl_out_slot, (size_t)G * n_out_max)) return rc;
    if (int rc = mb_alloc(s, &s->cl_halo_src, (size_t)G * n_halo_max)) return rc;
    if (int rc = mb_alloc(s, &s->cl_cnt8, (size_t)n8)) return rc;
    if (int rc = mb_alloc(s, &s->cl_aci8, acip.size())) return rc;
    if (int rc = mb_alloc(s, &s->cl_off4, (size_t)s->B * G * S * 4)) return rc;   // (mb_alloc zeroes: holes stay 0 for good)
    if (int rc = mb_alloc(s, &s->cl_diag, (size_t)s->B * G * S)) return rc;
    if (int rc = mb_alloc(s, &s->cl_bestx, (size_t)s->B * G * S)) return rc;
    if (int rc = mb_alloc(s, &s->cl_box, (size_t)s->B * 2 * G * W)) return rc;
    if (int rc = mb_alloc(s, &s->cl_epoch, (size_t)s->B)) return rc;
    s->cl_WJ = 2 * n_out_max + CL_NSC;
    if (int rc = mb_alloc(s, &s->cl_jbox, (size_t)s->B * 2 * G * s->cl_WJ)) return rc;
    if (int rc = mb_alloc(s, &s->cl_jepoch, (size_t)s->B)) return rc;
    if (int rc = mb_alloc(s, &s->cl_jabort, (size_t)s->B)) return rc;
    FG_HIP_CHECK(hipMemset(s->cl_jabort, 0xff, sizeof(uint32_t) * s->B));
    if (int rc = mb_alloc(s, &s->cl_abort, (size_t)s->B)) return rc;
    FG_HIP_CHECK(hipMemset(s->cl_abort, 0xff, sizeof(uint32_t) * s->B));   // no launch has base 0xFFFFFFFF
    FG_HIP_CHECK(hipMemcpy(s->cl_slot_cell, slot_cell.data(), sizeof(int32_t) * slot_cell.size(), hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->cl_cell_slot, cell_slot.data(), sizeof(uint16_t) * cell_slot.size(), hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->cl_nbr, nbr.data(), sizeof(uint2) * nbr.size(), hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->cl_d4g, d4t.data(), sizeof(mb_real) * d4t.size(), hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->cl_tinfo, tinfo.data(), sizeof(uint32_t) * tinfo.size(), hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->cl_out_slot, out_slot.data(), sizeof(int32_t) * out_slot.size(), hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->cl_halo_src, halo_src.data(), sizeof(uint32_t) * halo_src.size(), hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->cl_cnt8, cnt8.data(), sizeof(mb_real) * cnt8.size(), hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemcpy(s->cl_aci8, acip.data(), sizeof(mb_real) * acip.size(), hipMemcpyHostToDevice));
    s->cl_cpt = cpt; s->cl_nt = nt; s->cl_S = S; s->cl_W = W; s->cl_n8g_max = n8g_max; s->cl_n_out_max = n_out_max; s->cl_n_halo_max = n_halo_max;
    for (int g = 0; g <= G; ++g) s->cl_first[g] = first[g];
    for (int g = 0; g < G; ++g) { s->cl_n_out[g] = (int)out_cells[g].size(); s->cl_n_halo[g] = (int)halo_cells[g].size(); }
    {
        hipDeviceProp_t prop;
        int dev = 0;
        FG_HIP_CHECK(hipGetDevice(&dev));
        FG_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        s->cl_cus = prop.multiProcessorCount;
    }
    s->cl_on = true;
    s->cl_matrix_stale = true;
    return FG_OK;
#endif
}

bool mb_cluster_ok(const fg_mb_state* s, int pm_mode, const mb_real* diag, const mb_real* off) {
    if (!s->cl_on || !s->ml_on || s->cl_matrix_stale || pm_mode == 2 || diag != s->Pdiag || off != s->Poff) return false;
    if (s->cl_mode == 1 && s->N <= 8 * 1024) return false;   // small meshes: one workgroup with four / eight cells per thread is faster
    return s->cl_cus >= CL_G;
}

// the whole CG solve of every env in one launch of clusters; same arguments and results as mb_cg.  *fell_back: a workgroup gave up
// on a granule (FG_OK is returned and nothing was solved: the caller runs the one-workgroup kernels)
int mb_cg_cluster(fg_mb_state* s, const mb_real* dt, const mb_real* rhs, mb_real* x, mb_real tol, int max_iterations, int use_x0, int pm_mode,
                  mb_real stall_accept, int* max_it, hipStream_t st, bool* fell_back) {
    *fell_back = false;
#if FG_MB_F64
    (void)dt; (void)rhs; (void)x; (void)tol; (void)max_iterations; (void)use_x0; (void)pm_mode; (void)stall_accept; (void)max_it; (void)st;
    fg_set_error("the cluster CG is not part of the fp64 build");
    return FG_ERR_UNSUPPORTED;
#else
    const int nsys = s->B, n = s->N;
    MbSolve q = mb_solve_ptrs(s, s->Pdiag, s->Poff, rhs, x, 1, tol);
    q.best_x = s->w[4]; q.best_it = s->best_it;
    constexpr int CG_CHUNK = 20, CG_RESTART = 100;
    ClParams o;
    memset(&o, 0, sizeof(o));
    o.slot_cell = s->cl_slot_cell; o.nbr = s->cl_nbr; o.d4g = s->cl_d4g; o.tinfo = s->cl_tinfo; o.out_slot = s->cl_out_slot;
    o.halo_src = s->cl_halo_src; o.cnt8 = s->cl_cnt8; o.aci8 = s->cl_aci8;
    for (int g = 0; g <= CL_G; ++g) o.a8_first[g] = s->cl_first[g];
    for (int g = 0; g < CL_G; ++g) { o.n_out[g] = s->cl_n_out[g]; o.n_halo[g] = s->cl_n_halo[g]; }
    o.n_out_max = s->cl_n_out_max; o.n_halo_max = s->cl_n_halo_max; o.n8g_max = s->cl_n8g_max; o.n8 = s->ml_n8; o.ld8 = (s->ml_n8 + 3) & ~3; o.W = s->cl_W;
    o.geom_diag_sum = s->ml_geom_diag_sum;
    o.aci16 = reinterpret_cast<const _Float16*>(s->cl_aci16); o.ld16 = (s->ml_n8 + 7) & ~7; o.aci16_unscale = s->cl_aci16_unscale;
    o.off4 = s->cl_off4; o.diag = s->cl_diag; o.bestx = s->cl_bestx; o.box = reinterpret_cast<cl_u64*>(s->cl_box);
    o.epoch = s->cl_epoch; o.abort_at = s->cl_abort;
    o.dt = dt; o.B = nsys; o.N = n; o.allow_near = s->cl_near; o.dbg_out = s->oc_dbg;
    o.n_clusters = std::max(1, std::min(nsys, s->cl_cus / CL_G));   // one workgroup per CU: every workgroup of the grid is resident
    if (s->cl_max_clusters > 0) o.n_clusters = std::min(o.n_clusters, s->cl_max_clusters);
    o.use_x0 = use_x0; o.pm = pm_mode; o.restart_every = CG_RESTART; o.check_every = CG_CHUNK;
    o.max_iterations = ((max_iterations + CG_CHUNK - 1) / CG_CHUNK) * CG_CHUNK;
    o.stall_limit = s->cg_stall_limit; o.accept_window = 20;
    o.accept_factor = stall_accept > 1.f ? stall_accept : 0.f; o.tol = tol;
    o.info_host = s->info_pinned; o.its_host = s->flags_pinned;
    o.poll = fg_poll_next(&s->poll);
    const bool ev = s->prof_on != 0;
    const dim3 grid(o.n_clusters * CL_G);
#define CL_LAUNCH(CPT_, NT_, PM_, ACI_)                                                                                                    \
    do {                                                                                                                              \
        if (ev) hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_mbc_cluster<CPT_, NT_, PM_, ACI_>), grid, dim3(NT_), 0, st, s->prof_ev_oc[0], s->prof_ev_oc[1], 0, q, o); \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mbc_cluster<CPT_, NT_, PM_, ACI_>), grid, dim3(NT_), 0, st, q, o);                  \
    } while (0)
#define CL_LAUNCH_PM(CPT_, NT_, ACI_) do { if (pm_mode == 0) CL_LAUNCH(CPT_, NT_, 0, ACI_); else CL_LAUNCH(CPT_, NT_, 1, ACI_); } while (0)
    const bool aci = s->cl_n8g_max <= 64 && o.ld8 <= 256;   // the workgroup's rows of the coarse inverse fit its LDS copy
    const bool a16 = s->cl_n8g_max <= 96 && o.ld16 <= 384 && s->cl_aci16 != nullptr && s->cl_half;   // ... or as fp16
    if (s->cl_nt == 512 && aci) CL_LAUNCH_PM(8, 512, 1);
    else if (s->cl_nt == 512) CL_LAUNCH_PM(8, 512, 0);
    else if (s->cl_nt == 768 && a16) CL_LAUNCH_PM(8, 768, 2);
    else if (s->cl_nt == 768) CL_LAUNCH_PM(8, 768, 0);
    else CL_LAUNCH_PM(8, 1024, 0);
#undef CL_LAUNCH_PM
#undef CL_LAUNCH
    if (int rc = fg_poll_wait(&s->poll, ev ? FgPollOut{nullptr, 0} : o.poll, 0, nsys, st)) return rc;
    for (int i = 0; i < nsys; ++i)
        if (s->flags_pinned[i] == -77) { *fell_back = true; ++s->cl_fallbacks; return FG_OK; }
    ++s->cl_solves;
    if (ev) {
        fg_f32 ms = 0.f;
        FG_HIP_CHECK(hipEventElapsedTime(&ms, s->prof_ev_oc[0], s->prof_ev_oc[1]));
        long long its = 0;
        for (int i = 0; i < nsys; ++i) its += s->flags_pinned[i] > 0 ? s->flags_pinned[i] : 0;
        // bytes the kernel streams: per solve and cell the matrix row (20 B), neighbour addresses (8), rhs, x, kept iterate (about
        // 12); per iteration the rows of the coarse operator (n8 x n8 floats per env) and the granules (16 B per halo cell)
        s->prof_ms[2] += ms;
        s->prof_bytes[2] += (double)nsys * n * 40.0 + (double)its * ((double)o.n8 * o.ld8 * 4.0 + 16.0 * CL_G * o.n_halo_max);
        s->prof_n[2] += 1; s->prof_launches[2] += 1;
        s->prof_its += its;
    }
    return mb_finish(s, nsys, nullptr, max_it);
#endif
}

// mb_jacobi's sweeps by the clusters (k_mbj_cluster): the 2-D velocity systems of the meshes the cluster CG takes.  *fell_back: a
// workgroup gave up on a granule (nothing was solved: the caller runs the launch-per-sweep path); otherwise *done says whether every
// system settled -- when not, the pinned mirrors hold the flags and the last two measured residuals of every system, as mb_jacobi's
// check kernel leaves them.
bool mb_jacobi_cluster_ok(const fg_mb_state* s, int nc) {
    return s->cl_on && s->cl_jacobi && (s->cl_mode == 2 || s->N > 8 * 1024) && nc == 2 && s->d == 2 && s->cl_cus >= CL_G && s->cl_jbox != nullptr;
}
int mb_jacobi_cluster(fg_mb_state* s, const mb_real* dt, const mb_real* diag, const mb_real* off, const mb_real* rhs, mb_real* x, mb_real tol,
                      int use_x0, hipStream_t st, bool* fell_back, bool* done) {
    *fell_back = false; *done = false;
#if FG_MB_F64
    (void)dt; (void)diag; (void)off; (void)rhs; (void)x; (void)tol; (void)use_x0; (void)st;
    fg_set_error("the cluster sweeps are not part of the fp64 build");
    return FG_ERR_UNSUPPORTED;
#else
    const int nsys = s->B * 2;
    MbSolve q = mb_solve_ptrs(s, diag, off, rhs, x, 2, tol);
    hipLaunchKernelGGL(k_mbs_begin_cl, dim3((nsys + 63) / 64), dim3(64), 0, st, dt, q, nsys);
    ClJParams o;
    memset(&o, 0, sizeof(o));
    o.slot_cell = s->cl_slot_cell; o.nbr = s->cl_nbr; o.tinfo = s->cl_tinfo; o.out_slot = s->cl_out_slot; o.halo_src = s->cl_halo_src;
    for (int g = 0; g < CL_G; ++g) { o.n_out[g] = s->cl_n_out[g]; o.n_halo[g] = s->cl_n_halo[g]; }
    o.n_out_max = s->cl_n_out_max; o.n_halo_max = s->cl_n_halo_max; o.WJ = s->cl_WJ;
    o.box = reinterpret_cast<cl_u64*>(s->cl_jbox); o.epoch = s->cl_jepoch; o.abort_at = s->cl_jabort;
    o.B = s->B; o.N = s->N; o.allow_near = s->cl_near; o.use_x0 = use_x0; o.tol = tol;
    o.n_clusters = std::max(1, std::min(s->B, s->cl_cus / CL_G));
    if (s->cl_max_clusters > 0) o.n_clusters = std::min(o.n_clusters, s->cl_max_clusters);
    o.info_host = s->info_pinned; o.flags_host = s->flags_pinned; o.res2_host = s->jac_res_pinned;
    o.poll = fg_poll_next(&s->poll);
    const dim3 grid(o.n_clusters * CL_G);
    if (s->cl_nt == 512) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mbj_cluster<512>), grid, dim3(512), 0, st, q, o);
    else if (s->cl_nt == 768) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mbj_cluster<768>), grid, dim3(768), 0, st, q, o);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mbj_cluster<1024>), grid, dim3(1024), 0, st, q, o);
    if (int rc = fg_poll_wait(&s->poll, o.poll, 0, nsys, st)) return rc;
    bool all = true;
    for (int i = 0; i < nsys; ++i) {
        if (s->flags_pinned[i] == -77) { *fell_back = true; ++s->cl_fallbacks; return FG_OK; }
        all = all && s->flags_pinned[i] != 0;
    }
    ++s->cl_jacobi_solves;
    *done = all;
    return FG_OK;
#endif
}
