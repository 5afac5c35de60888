// Point-Jacobi sweeps for the velocity systems: several per pass over the field with the region on chip (2-D uniform-grid envs: the
// channel family), one per launch elsewhere (3-D: the turbulent channel).
//
// The advection-diffusion matrix of a PISO step is A = D + O with D = 1/dt + (diffusive and advective face sums) and O the four
// neighbour coefficients (k_adv_build, fg_piso.hip; reference PISO_build_advection_matrix :4170-4312).  On the channel grids at
// the envs' time steps the rows are strongly diagonally dominant -- sum|O| / D = 0.34 (256 x 128) and 0.63 (512 x 256), measured
// on the bench states -- and the reference's BiCGStab (bicgstab_solver_kernel.cu) needs 4-5 / 7-8 iterations of two
// matrix applications each, every one a full pass over x, r, p, v, s, t and the matrix in HBM (~48 floats per cell and iteration
// for the two components).  The stationary iteration x <- D^-1 (b - O x) contracts the residual by 0.18 / 0.47 per sweep on
// the same systems (11 / 24 sweeps to the reference's criterion, RMS residual < tol; profiles/jacobi_exp_*.py) and needs NO
// dot product between sweeps, so S sweeps run on one tile that stays on chip:
//
//   * a workgroup owns a region of 8192 cells: ROWS full rows of the grid tiling y (no halo in x: the row is complete; periodic x
//     wraps inside it, FIXED x has zero coefficients there), or a band of ALL rows, 8192 / ny columns wide, tiling x (FIXED x only);
//     of a region the outer S cells towards a neighbouring region are halo: after S sweeps the inner ones are exact sweeps of the
//     global iteration.  Regions at a wall need no halo on that side.  Shape and S depend on the grid alone (jac_plan).
//   * each of the 512 threads keeps a strip of 4 rows x 4 columns in registers: the pre-scaled coefficients O/D (16 x 4), the
//     right-hand sides b/D and the iterates of BOTH components (the matrix is shared by them);
//     x neighbours come from the neighbouring lanes (DPP row shifts in narrow bands, ds_bpermute otherwise), y neighbours across
//     strips from a ping-pong LDS array that holds only the top and bottom row of every strip (64 KB).  One barrier per sweep; the
//     two inner rows of a strip are updated in front of it.
//   * workgroups are numbered so that the regions of an env run on one XCD (fg_xcd_remap): their halo loads meet in its L2.
//   * the last sweep of a pass also gives the residual of the iterate it started from: b - A x_k = D (x_{k+1} - x_k).  Its sum of
//     squares over the region's output rows goes to the system's accumulator ring (FgDacc, order-independent); the NEXT pass -- or
//     the check kernel behind the last enqueued pass -- takes the verdict from it with the rule of the Krylov kernels
//     (RMS residual < tol; not finite = failed), per env: both components stop together.
//
// Per pass and cell: 9 floats read (5 matrix, 2 b, 2 x; x (ROWS_loaded / ny) for the halo rows) and 2 written for S sweeps, against
// ~24 floats per matrix application of the two-kernel BiCGStab.  The host enqueues the number of passes the previous solve of the
// same kind needed, then k_jac_check and one poll; what it learns there (contraction per pass) sizes what follows.  A solve that
// does not contract (rows not dominant enough: refined grids, large time steps) is handed to BiCGStab from a cleared start
// vector, and the kind backs off from trying again for a while.  Same system, same tolerance, same criterion: another iteration,
// like the preconditioners of the other solves (fluidgym_amd/simulation/policy.py: advection_jacobi).
//
// Grids without a region shape (3-D) get the same sweeps one launch at a time (k_jac_stream, below).
// fp32 library only (the fp64 build keeps the plain recurrences).
#include <mutex>

#include "fg_internal.h"
#include "fg_bicg.h"

#if !FG_F64
namespace {

constexpr int JAC_THREADS = 512;
constexpr int JAC_CELLS = 8192;
constexpr int JAC_MAX_PASSES = 24;

struct JacArgs {
    const float* diag; const float* off; const float* rhs;   // [B,N], [B,4,N], [B,2,N]
    const float* xin; float* xout;                            // [B,2,N] (xin unused in a pass that starts from zero)
    FgDacc* acc; int32_t* flags; fg_solve_info* info;
    float tol;
    int pass, sweeps, zero_start, nx, ny, n, tiles;
    int xcd_remap;      // FG_JAC_XCD (default 1): regions of an env on one XCD
};

__device__ __forceinline__ void jac_mark(const JacArgs& a, int sys, float crit, int sweeps_done) {
    const bool finite = isfinite(crit);
    a.info[sys].final_residual = crit;
    a.info[sys].used_iterations = sweeps_done;
    a.info[sys].converged = (finite && crit < a.tol) ? 1 : 0;
    a.info[sys].is_finite = finite ? 1 : 0;
    flag_st(a.flags + sys, finite ? 1 : 2);
}

// verdict on the pass before `pass` for env b (both systems): true = the env needs no further work.  Every workgroup of the env
// comes to it from the same accumulator words; `leader` stores it (or, for an env that goes on, resets the ring entry pass + 1).
__device__ __forceinline__ bool jac_verdict(const JacArgs& a, int b, int pass, bool leader) {
    const int sys0 = 2 * b;
    if (flag_ld(a.flags + sys0) != 0 && flag_ld(a.flags + sys0 + 1) != 0) return true;
    if (pass == 0) return false;
    FgDacc* A0 = a.acc + (size_t)sys0 * FG_ACC_DOUBLES;
    FgDacc* A1 = A0 + FG_ACC_DOUBLES;
    const int e = (pass - 1) % 3;
    const float c0 = fg_rms(acc_ld(A0 + e), a.n), c1 = fg_rms(acc_ld(A1 + e), a.n);
    const bool bad = !isfinite(c0) || !isfinite(c1);
    const bool done = bad || (c0 < a.tol && c1 < a.tol);
    if (leader) {
        if (done) {
            jac_mark(a, sys0, c0, pass * a.sweeps - 1);      // (used_iterations: the 0-based index of the last sweep, as the Krylov solvers count)
            jac_mark(a, sys0 + 1, c1, pass * a.sweeps - 1);
        } else {
            // (the sums of passes 0 and 1 outlive the ring in slots 3 and 4: the give-up rule of the host reads THEM, whatever it has
            //  enqueued ahead -- ADVICE r5: which solver runs must not depend on the handle's history)
            if (pass == 2 || pass == 3) {      // (a check and the pass behind it both come here: the second finds the entry already cleared)
                const double v0 = acc_ld(A0 + (pass + 1) % 3), v1 = acc_ld(A1 + (pass + 1) % 3);
                if (v0 != 0.0) acc_st(A0 + 1 + pass, v0);
                if (v1 != 0.0) acc_st(A1 + 1 + pass, v1);
            }
            acc_st(A0 + (pass + 1) % 3, 0.0);
            acc_st(A1 + (pass + 1) % 3, 0.0);
        }
    }
    return done;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// value of the lane below / above within the 16-lane DPP row (row_shr:1 / row_shl:1); the first / last lane of a row keeps its own
__device__ __forceinline__ float dpp_from_below(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_above(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x101, 0xf, 0xf, false));
}

// XT = false: the region is ROWS full rows (4 Q = nx columns), regions tile the y axis.  XT = true: the region is ALL rows (ROWS = ny)
// of a band of 4 Q columns, regions tile the x axis (FIXED x faces only: the band's lane wrap-around is halo, never the periodic
// neighbour) -- the better shape for wide grids, where a full-row region is only 16 rows high and half of what it loads is halo.
template <int Q, bool XT>
__global__ __launch_bounds__(JAC_THREADS) void k_jac_pass(JacArgs a) {
    constexpr int NX = 4 * Q;                  // columns of the region
    constexpr int STRIPS = JAC_THREADS / Q;    // strips of 4 rows
    constexpr int ROWS = 4 * STRIPS;
    static_assert(Q * STRIPS == JAC_THREADS && ROWS * NX == JAC_CELLS, "region shape");
    constexpr bool SPLIT = Q > 64;             // a row spans several waves: the x neighbours at the wave seams go through LDS
    static_assert(!(SPLIT && XT), "bands are at most one wave wide");
    constexpr int WPR = SPLIT ? Q / 64 : 1;    // waves per row
    // dynamic LDS (64 KB + the seams: above the static limit, fg_jacobi_lds_ready):
    //   edge [buffer][comp][2 * strip + (0 top | 1 bottom row of the strip)][column]
    //   seam [buffer][comp][row][2 * wave-of-row + (0 first | 1 last cell of the wave's part of the row)]   (SPLIT only)
    extern __shared__ __attribute__((aligned(16))) float jac_lds[];
    float (*edge)[2][2 * STRIPS][NX] = reinterpret_cast<float (*)[2][2 * STRIPS][NX]>(jac_lds);
    float (*seam)[2][ROWS][2 * WPR] = reinterpret_cast<float (*)[2][ROWS][2 * WPR]>(jac_lds + 2 * 2 * 2 * STRIPS * NX);
    __shared__ float red[2][JAC_THREADS / 64];
    // XCD-aware order: the hardware places workgroup id on XCD id % 8; every XCD gets a contiguous run of (env, region) pairs, so that
    // neighbouring regions of an env -- which share their halo rows / columns and, in narrow bands, halves of 128-byte lines -- run on
    // one XCD at about the same time and find each other's loads in its L2
    const unsigned lid = a.xcd_remap ? fg_xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y) : blockIdx.x + gridDim.x * blockIdx.y;
    const int b = (int)(lid / gridDim.x), tile = (int)(lid % gridDim.x), t = threadIdx.x, lane = t & 63;
    if (jac_verdict(a, b, a.pass, tile == 0 && t == 0)) return;
    // tiles along the tiled axis (y for full-row regions, x for bands): `span` cells per region, of which the outer S towards a
    // neighbouring region are halo
    constexpr int SPAN = XT ? NX : ROWS;
    const int extent = XT ? a.nx : a.ny;
    const int S = a.sweeps, TY = SPAN - 2 * S;
    int start = 0, out0 = 0, out1 = extent;
    if (a.tiles > 1) {
        out0 = tile == 0 ? 0 : (SPAN - S) + (tile - 1) * TY;
        out1 = tile == a.tiles - 1 ? extent : (SPAN - S) + tile * TY;
        start = tile == 0 ? 0 : out0 - S;
        if (start > extent - SPAN) start = extent - SPAN;
    }
    const int col = t % Q, strip = t / Q;
    const int row0 = 4 * strip;               // first region row of the strip
    const size_t ld = (size_t)a.nx;           // row pitch of the fields
    const size_t cell0 = XT ? (size_t)row0 * ld + (size_t)(start + 4 * col) : (size_t)(start + row0) * ld + (size_t)(4 * col);
    const size_t n = (size_t)a.n;
    // which of the thread's cells the region answers for: rows of the strip (full-row regions) or columns of the quad (bands)
    const int lo = out0 - (XT ? start + 4 * col : start + row0), hi = out1 - (XT ? start + 4 * col : start + row0);   // cell k of 0..3 is inside iff lo <= k < hi
    auto inside = [&](int r, int e) { const int k = XT ? e : r; return k >= lo && k < hi; };
    // ---- the strip's part of the system: all loads first, then the scaling
    float4 of[4][4], bp[2][4], xo[2][4];
    {
    float4 dg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        dg[r] = ld4(a.diag + (size_t)b * n + cell0 + (size_t)r * ld);
#pragma unroll
        for (int f = 0; f < 4; ++f) of[f][r] = ld4(a.off + ((size_t)b * 4 + f) * n + cell0 + (size_t)r * ld);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bp[c][r] = ld4(a.rhs + ((size_t)b * 2 + c) * n + cell0 + (size_t)r * ld);
            xo[c][r] = a.zero_start ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4(a.xin + ((size_t)b * 2 + c) * n + cell0 + (size_t)r * ld);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float4 rd = make_float4(1.f / dg[r].x, 1.f / dg[r].y, 1.f / dg[r].z, 1.f / dg[r].w);
#pragma unroll
        for (int f = 0; f < 4; ++f) { of[f][r].x *= rd.x; of[f][r].y *= rd.y; of[f][r].z *= rd.z; of[f][r].w *= rd.w; }
#pragma unroll
        for (int c = 0; c < 2; ++c) { bp[c][r].x *= rd.x; bp[c][r].y *= rd.y; bp[c][r].z *= rd.z; bp[c][r].w *= rd.w; }
    }
    }
    // lanes that hold the x neighbours of this thread's quad (same row: the quads of a row are consecutive threads)
    const int lane_l = SPLIT ? ((lane + 63) & 63) : (lane - (col) + ((col + Q - 1) % Q));
    const int lane_r = SPLIT ? ((lane + 1) & 63) : (lane - (col) + ((col + 1) % Q));
    const int wrow = SPLIT ? (col >> 6) : 0;              // which wave of the row this thread sits in
    const int up_row = strip > 0 ? 2 * (strip - 1) + 1 : 0;                 // bottom row of the strip above (clamped: rows outside the
    const int dn_row = strip < STRIPS - 1 ? 2 * (strip + 1) : 2 * strip + 1;  // region are a wall -- zero coefficient -- or halo)
    float part[2] = {0.f, 0.f};
    for (int k = 0; k < S; ++k) {
        const int cur = k & 1;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            *reinterpret_cast<float4*>(&edge[cur][c][2 * strip][4 * col]) = xo[c][0];
            *reinterpret_cast<float4*>(&edge[cur][c][2 * strip + 1][4 * col]) = xo[c][3];
            if constexpr (SPLIT) {
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) seam[cur][c][row0 + r][2 * wrow] = xo[c][r].x;
                }
                if (lane == 63) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) seam[cur][c][row0 + r][2 * wrow + 1] = xo[c][r].w;
                }
            }
        }
        // one cell row of the strip: x neighbours from the neighbouring lanes, y neighbours `up` / `dn` (the thread's own rows, or the
        // strip edges of the neighbours from LDS for rows 0 and 3)
        auto row_update = [&](int c, int r, const float4& up, const float4& dn) -> float4 {
            float xl, xr;
            if constexpr (XT && Q <= 16) {
                // narrow bands (a band row is 8 or 16 lanes = at most one 16-lane DPP row): what comes in over a band's edge is halo
                // (or meets a zero coefficient at a wall), so the neighbour lanes of the DPP row will do -- a VALU move instead of
                // a trip through the LDS crossbar
                xl = dpp_from_below(xo[c][r].w); xr = dpp_from_above(xo[c][r].x);
            } else {
                xl = __shfl(xo[c][r].w, lane_l, 64); xr = __shfl(xo[c][r].x, lane_r, 64);
            }
            if constexpr (SPLIT) {
                if (lane == 0) xl = seam[cur][c][row0 + r][2 * ((wrow + WPR - 1) % WPR) + 1];
                if (lane == 63) xr = seam[cur][c][row0 + r][2 * ((wrow + 1) % WPR)];
            }
            const float4 x = xo[c][r];
            float4 v;
            v.x = bp[c][r].x - of[0][r].x * xl - of[1][r].x * x.y - of[2][r].x * up.x - of[3][r].x * dn.x;
            v.y = bp[c][r].y - of[0][r].y * x.x - of[1][r].y * x.z - of[2][r].y * up.y - of[3][r].y * dn.y;
            v.z = bp[c][r].z - of[0][r].z * x.y - of[1][r].z * x.w - of[2][r].z * up.z - of[3][r].z * dn.z;
            v.w = bp[c][r].w - of[0][r].w * x.z - of[1][r].w * xr - of[2][r].w * up.w - of[3][r].w * dn.w;
            return v;
        };
        float4 xn[2][4];
        // the two inner rows of the strip need nothing from LDS: they are updated while the strip edges are on their way (with two
        // waves per SIMD nothing else would fill the wait for the barrier); rows that span several waves read their seams after it
        if constexpr (!SPLIT) {
#pragma unroll
            for (int c = 0; c < 2; ++c) { xn[c][1] = row_update(c, 1, xo[c][0], xo[c][2]); xn[c][2] = row_update(c, 2, xo[c][1], xo[c][3]); }
        }
        __syncthreads();
        const bool last = (k == S - 1);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float4 upq = *reinterpret_cast<const float4*>(&edge[cur][c][up_row][4 * col]);
            const float4 dnq = *reinterpret_cast<const float4*>(&edge[cur][c][dn_row][4 * col]);
            if constexpr (SPLIT) { xn[c][1] = row_update(c, 1, xo[c][0], xo[c][2]); xn[c][2] = row_update(c, 2, xo[c][1], xo[c][3]); }
            xn[c][0] = row_update(c, 0, upq, xo[c][1]);
            xn[c][3] = row_update(c, 3, xo[c][2], dnq);
            if (last) {
                // residual of the iterate this sweep started from, on the cells this region answers for (the diagonal is read again
                // here -- from L2 -- instead of living in 16 registers through the sweeps)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (XT ? (hi > 0 && lo < 4) : inside(r, 0)) {
                        const float4 d4 = ld4(a.diag + (size_t)b * n + cell0 + (size_t)r * ld);
                        const float r0 = d4.x * (xn[c][r].x - xo[c][r].x), r1 = d4.y * (xn[c][r].y - xo[c][r].y);
                        const float r2 = d4.z * (xn[c][r].z - xo[c][r].z), r3 = d4.w * (xn[c][r].w - xo[c][r].w);
                        part[c] += (inside(r, 0) ? r0 * r0 : 0.f) + (inside(r, 1) ? r1 * r1 : 0.f) + (inside(r, 2) ? r2 * r2 : 0.f) + (inside(r, 3) ? r3 * r3 : 0.f);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) xo[c][r] = xn[c][r];
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* dst = a.xout + ((size_t)b * 2 + c) * n + cell0 + (size_t)r * ld;
            if (inside(r, 0) && inside(r, 3)) *reinterpret_cast<float4*>(dst) = xo[c][r];
            else if (XT) {      // a band's output range ends inside this quad
                if (inside(r, 0)) dst[0] = xo[c][r].x;
                if (inside(r, 1)) dst[1] = xo[c][r].y;
                if (inside(r, 2)) dst[2] = xo[c][r].z;
                if (inside(r, 3)) dst[3] = xo[c][r].w;
            }
        }
        const float s = fg_wave_sum(part[c]);
        if (lane == 0) red[c][t >> 6] = s;
    }
    __syncthreads();
    if (t < 2) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < JAC_THREADS / 64; ++w) s += red[t][w];
        acc_add(a.acc + (size_t)(2 * b + t) * FG_ACC_DOUBLES + a.pass % 3, (double)s);
    }
}

// behind the last enqueued pass: its verdict (what the next pass would have found), the info mirror and the poll words; for envs
// that go on, the residual of the pass before rides along so that the host can size what it enqueues next
__global__ void k_jac_check(JacArgs a, fg_solve_info* __restrict__ mirror, float* __restrict__ prev, int B, FgPollOut poll) {
    __shared__ uint32_t stage[64 * 10];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = b < B;
    uint32_t w[10] = {};
    if (valid) {
        const int sys0 = 2 * b;
        // a.pass = number of passes enqueued so far: the verdict pass number a.pass would take
        const FgDacc* A0 = a.acc + (size_t)sys0 * FG_ACC_DOUBLES;
        const FgDacc* A1 = A0 + FG_ACC_DOUBLES;
        float p0 = -1.f, p1 = -1.f;      // (read before the verdict: an env that goes on resets this ring entry for pass a.pass + 1)
        if (a.pass >= 2) { const int e2 = (a.pass - 2) % 3; p0 = fg_rms(acc_ld(A0 + e2), a.n); p1 = fg_rms(acc_ld(A1 + e2), a.n); }
        // the residuals passes 0 and 1 measured (ring entries 0 and 1 until passes 2 and 3 moved them to slots 3 and 4)
        float f00 = -1.f, f01 = -1.f, f10 = -1.f, f11 = -1.f;
        if (a.pass >= 2) {
            f00 = fg_rms(acc_ld(A0 + (a.pass >= 3 ? 3 : 0)), a.n); f10 = fg_rms(acc_ld(A1 + (a.pass >= 3 ? 3 : 0)), a.n);
            f01 = fg_rms(acc_ld(A0 + (a.pass >= 4 ? 4 : 1)), a.n); f11 = fg_rms(acc_ld(A1 + (a.pass >= 4 ? 4 : 1)), a.n);
        }
        const bool done = jac_verdict(a, b, a.pass, true);
        if (!done) {
            const int e = (a.pass - 1) % 3;
            a.info[sys0].final_residual = fg_rms(acc_ld(A0 + e), a.n); a.info[sys0 + 1].final_residual = fg_rms(acc_ld(A1 + e), a.n);
            a.info[sys0].used_iterations = a.info[sys0 + 1].used_iterations = a.pass * a.sweeps - 1;
            a.info[sys0].converged = a.info[sys0 + 1].converged = 0;
        }
        if (poll.gran) {
            // result words (FgPollOut, fg_internal.h), ten per env: the two infos, the residuals of the pass before, those of passes 0 / 1
            const fg_solve_info i0 = a.info[sys0], i1 = a.info[sys0 + 1];
            w[0] = __float_as_uint(i0.final_residual); w[1] = fg_info_word(i0); w[2] = __float_as_uint(i1.final_residual); w[3] = fg_info_word(i1);
            w[4] = __float_as_uint(p0); w[5] = __float_as_uint(p1);
            w[6] = __float_as_uint(f00); w[7] = __float_as_uint(f01); w[8] = __float_as_uint(f10); w[9] = __float_as_uint(f11);
        } else {
            prev[sys0] = p0; prev[sys0 + 1] = p1;
            float* first2 = prev + 2 * B;      // [2 B][2]
            first2[2 * sys0] = f00; first2[2 * sys0 + 1] = f01; first2[2 * (sys0 + 1)] = f10; first2[2 * (sys0 + 1) + 1] = f11;
            mirror[sys0] = a.info[sys0]; mirror[sys0 + 1] = a.info[sys0 + 1];
            fg_poll_publish(poll, sys0);
            fg_poll_publish(poll, sys0 + 1);
        }
    }
    if (poll.gran) {      // (every lane: the wave publishes together, the writers' records first -- no lane may run ahead of them)
        const int first = blockIdx.x * blockDim.x;
        fg_poll_publish_records<10>(poll, first, min((int)blockDim.x, B - first), threadIdx.x, w, valid, stage);
    }
}

// envs whose last iterate was written to the work buffer: into the result.  Pass p wrote the result vector iff (p & 1) == last_parity;
// an env that stopped with the verdict of pass q + 1 has the output of pass q = (used + 1) / S - 1 (the host launches this only when its
// copy of the sweep counts says some env needs it)
__global__ __launch_bounds__(256) void k_jac_settle(const float* __restrict__ work, float* __restrict__ x, const fg_solve_info* __restrict__ info,
                                                    int sweeps, int last_parity, int n2 /* floats per env: 2 n */) {
    const int b = blockIdx.y;
    const int used = info[2 * b].used_iterations + 1;
    if (used <= 0 || ((used / sweeps - 1) & 1) == last_parity) return;
    const float4* src = reinterpret_cast<const float4*>(work + (size_t)b * n2);
    float4* dst = reinterpret_cast<float4*>(x + (size_t)b * n2);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n2 / 4; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------------------------------------
// Streaming form for the grids the on-chip regions do not cover (3-D: the turbulent channel): one launch per sweep.  A sweep is
// k_h's arithmetic -- x_new = rA (rhs - sum_f off_f x[N_f]), all components of a cell in one thread -- plus, in the sweeps that are
// followed by a check, the sum of squares of the residual of the iterate it started from, (x_new - x) / rA = b - C x.  The TCF systems
// (CFL 0.1: sum|O| / D = 0.08) contract by 0.03 per sweep: 7 sweeps of one matrix pass each where BiCGStab takes 4 iterations of
// two passes over the matrix and six over the vectors (profiles/jacobi_exp_tcf.py).
// ---------------------------------------------------------------------------------------------------------------------------
struct JacStreamArgs {
    const float* rA; const float* off; const float* rhs;
    const float* xin; float* xout;
    FgDacc* acc; const int32_t* flags; const float* dt;
    int from_zero, measure_slot;      // measure_slot < 0: no residual sum in this sweep
    int ax_slot;                      // >= 0: this sweep also sums (A x_new)^2, the scale of the fp32 rounding floor of the residual
};

template <int DIMS, int VEC>
__global__ __launch_bounds__(FG_BLOCK) void k_jac_stream(FgGrid g, JacStreamArgs a, int tiles_x, int tiles_y, int tiles) {
    const FgCtx<DIMS, VEC> c = fg_make_ctx<DIMS, VEC>(g, tiles_x, tiles_y, tiles);
    __shared__ float lds[DIMS * 4];
    bool live = false;
#pragma unroll
    for (int q = 0; q < DIMS; ++q) live = live || flag_ld(a.flags + (c.b * DIMS + q)) == 0;
    if (!live) return;
    const size_t N = g.n;
    float part[DIMS], pax[DIMS];
#pragma unroll
    for (int q = 0; q < DIMS; ++q) { part[q] = 0.f; pax[q] = 0.f; }
    if (c.valid) {
        const FgVec<VEC> rA = fg_load<VEC>(a.rA + (size_t)c.b * N + c.idx);
        FgVec<VEC> off[2 * DIMS];
        if (!a.from_zero) {
#pragma unroll
            for (int f = 0; f < 2 * DIMS; ++f) off[f] = fg_load<VEC>(a.off + ((size_t)c.b * 2 * DIMS + f) * N + c.idx);
        }
#pragma unroll
        for (int q = 0; q < DIMS; ++q) {
            const size_t base = ((size_t)c.b * DIMS + q) * N;
            const FgVec<VEC> r = fg_load<VEC>(a.rhs + base + c.idx);
            FgVec<VEC> out;
            if (a.from_zero) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) { out.v[e] = rA.v[e] * r.v[e]; const float d = r.v[e]; part[q] += d * d; }
            } else {
                const FgNbr<DIMS, VEC> u = fg_gather<DIMS, VEC>(a.xin + base, c);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float H = off[0].v[e] * u.xm.v[e] + off[1].v[e] * u.xp.v[e] + off[2].v[e] * u.ym.v[e] + off[3].v[e] * u.yp.v[e];
                    if constexpr (DIMS == 3) H += off[4].v[e] * u.zm.v[e] + off[5].v[e] * u.zp.v[e];
                    out.v[e] = rA.v[e] * (r.v[e] - H);
                    const float d = (out.v[e] - u.c.v[e]) / rA.v[e];
                    part[q] += d * d;
                    const float ax = r.v[e] - H;      // = A x_new (the diagonal term of the row)
                    pax[q] += ax * ax;
                }
            }
            fg_store<VEC>(a.xout + base + c.idx, out);
        }
    }
    if (a.measure_slot >= 0) {
#pragma unroll
        for (int q = 0; q < DIMS; ++q) {
            const float w = fg_wave_sum(part[q]);
            if ((threadIdx.x & 63) == 0) lds[q * 4 + (threadIdx.x >> 6)] = w;
        }
        __syncthreads();
        if (threadIdx.x < DIMS)
            acc_add(a.acc + (size_t)(c.b * DIMS + threadIdx.x) * FG_ACC_DOUBLES + a.measure_slot,
                    (double)(lds[threadIdx.x * 4] + lds[threadIdx.x * 4 + 1] + lds[threadIdx.x * 4 + 2] + lds[threadIdx.x * 4 + 3]));
        if (a.ax_slot >= 0) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < DIMS; ++q) {
                const float w = fg_wave_sum(pax[q]);
                if ((threadIdx.x & 63) == 0) lds[q * 4 + (threadIdx.x >> 6)] = w;
            }
            __syncthreads();
            if (threadIdx.x < DIMS)
                acc_add(a.acc + (size_t)(c.b * DIMS + threadIdx.x) * FG_ACC_DOUBLES + a.ax_slot,
                        (double)(lds[threadIdx.x * 4] + lds[threadIdx.x * 4 + 1] + lds[threadIdx.x * 4 + 2] + lds[threadIdx.x * 4 + 3]));
        }
    }
}

// verdict behind a measuring sweep of the streaming form, per env (its components stop together), mirrors, both measured residuals
// A system has converged when the measured residual is below the tolerance -- or below what an fp32 iterate can show: the residual
// b - C x is a difference of terms of size |A x|, each carrying half an ulp, so its RMS cannot be resolved below ~2^-24 RMS(A x)
// (TCF: |u| ~ 1, 1 / dt = 256: the sweeps' measure settles at 3.7e-6 for the x component against a tolerance of 1e-6, the accepted level is
// 2^-23 RMS(A x) = 2e-5; the reference's fp32 BiCGStab "reaches" such a tolerance only in its
// recurrence residual, bicgstab_solver_kernel.cu:305-329 -- the true residual of its result sits at the same floor).  The channel
// family's tolerance is above its floor (7e-6 / 1e-5), so the rule is inactive there.
__global__ void k_jac_stream_check(FgDacc* __restrict__ acc, int32_t* __restrict__ flags, fg_solve_info* __restrict__ info, fg_solve_info* __restrict__ mirror,
                                   float* __restrict__ res2, float tol, int nc, int slot_now, int slot_prev, int ax_slot, int sweeps, int n, int B,
                                   FgPollOut poll) {
    __shared__ uint32_t stage[64 * 12];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = b < B;
    // result words (FgPollOut, fg_internal.h), twelve per env: per component the info (2), the residual measured now, the one before
    uint32_t w[12] = {};
    float rnow[3] = {-1.f, -1.f, -1.f}, rprev[3] = {-1.f, -1.f, -1.f};
    if (valid) {
        bool running = false;
        for (int q = 0; q < nc; ++q) running = running || flag_ld(flags + (b * nc + q)) == 0;
        if (running) {
            bool bad = false, all = true;
            for (int q = 0; q < nc; ++q) {
                const float now = fg_rms(acc_ld(acc + (size_t)(b * nc + q) * FG_ACC_DOUBLES + slot_now), n);
                const float floor32 = ax_slot >= 0 ? 1.1920929e-7f * fg_rms(acc_ld(acc + (size_t)(b * nc + q) * FG_ACC_DOUBLES + ax_slot), n) : 0.f;
                // (the floor rule is capped at 8 x the tolerance: beyond that the measure is not rounding, the system is not solved -- ADVICE r5)
                bad = bad || !isfinite(now); all = all && (now < tol || (now < floor32 && now < 8.f * tol));
            }
            for (int q = 0; q < nc; ++q) {
                const int sys = b * nc + q;
                const float now = fg_rms(acc_ld(acc + (size_t)sys * FG_ACC_DOUBLES + slot_now), n);
                const float prev = slot_prev >= 0 ? fg_rms(acc_ld(acc + (size_t)sys * FG_ACC_DOUBLES + slot_prev), n) : -1.f;
                if (q < 3) { rnow[q] = now; rprev[q] = prev; }
                info[sys].final_residual = now;
                info[sys].used_iterations = sweeps - 1;
                if (bad || all) {
                    const bool finite = isfinite(now);
                    info[sys].converged = (finite && all) ? 1 : 0;
                    info[sys].is_finite = finite ? 1 : 0;
                    flag_st(flags + sys, finite ? 1 : 2);
                }
            }
        }
        if (poll.gran) {
            for (int q = 0; q < nc && q < 3; ++q) {
                const fg_solve_info v = info[b * nc + q];
                w[4 * q] = __float_as_uint(v.final_residual); w[4 * q + 1] = fg_info_word(v);
                w[4 * q + 2] = __float_as_uint(rnow[q]); w[4 * q + 3] = __float_as_uint(rprev[q]);
            }
        } else {
            for (int q = 0; q < nc; ++q) {
                res2[2 * (b * nc + q)] = q < 3 ? rnow[q] : -1.f; res2[2 * (b * nc + q) + 1] = q < 3 ? rprev[q] : -1.f;
                mirror[b * nc + q] = info[b * nc + q]; fg_poll_publish(poll, b * nc + q);
            }
        }
    }
    if (poll.gran) {      // (every lane: the wave publishes together)
        const int first = blockIdx.x * blockDim.x;
        fg_poll_publish_records<12>(poll, first, min((int)blockDim.x, B - first), threadIdx.x, w, valid, stage);
    }
}

int jac_tiles(int ny, int rows, int sweeps) {
    if (ny == rows) return 1;
    const int ty = rows - 2 * sweeps, rest = ny - 2 * (rows - sweeps);
    return 2 + (rest > 0 ? (rest + ty - 1) / ty : 0);
}

constexpr size_t JAC_LDS = sizeof(float) * (2 * 2 * 2 * (JAC_CELLS / 4) + 2 * 2 * 128 * 2);   // edge rows + seams (<= 128 row-waves)

#define JAC_FOR_EACH_KERNEL(X) X(16, false) X(32, false) X(64, false) X(128, false) X(8, true) X(16, true) X(32, true) X(64, true)

// dynamic LDS above 64 KB needs an explicit opt-in per kernel -- and per DEVICE: the attribute belongs to the device that is current
// when it is set (a process with handles on two GPUs launched k_jac_pass on the second one without it)
bool jac_lds_ready() {
    static std::mutex mu;
    static int states[64] = {0};      // per device: 0 not tried, 1 granted, -1 refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return false; }
    std::lock_guard<std::mutex> lock(mu);
    int& state = states[dev];
    if (state == 0) {
        state = 1;
#define JAC_OPT_IN(Q, XT)                                                                                                              \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_jac_pass<Q, XT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)JAC_LDS) != hipSuccess) { \
            (void)hipGetLastError(); state = -1;                                                                                       \
        }
        JAC_FOR_EACH_KERNEL(JAC_OPT_IN)
#undef JAC_OPT_IN
    }
    return state == 1;
}

// Region shape of a grid: full rows tiling y, or bands of all rows tiling x (FIXED x only) -- whichever loads fewer rows / columns per
// sweep.  A function of the grid alone (never of the handle's history): the iterate of an env must not depend on it.
struct JacPlan { bool ok, xt; int q, span, sweeps, tiles; double cost; };
// sweeps per pass by region depth along the tiled axis.  Bands take more: their halo costs columns of ALL rows, and a pass of 12 sweeps
// over 64-column bands (1.5 x the columns loaded) settles the headline's systems in ONE pass over the matrix (measured, same box:
// 7 100-7 170 env-steps/s against 6 400-6 780 with two passes of 6 over full rows); 8 on 32-column bands (512 x 256: 1 834 against
// 1 466 with full rows of 16)
int jac_sweeps_for(int span, bool xt) { return xt ? (span <= 32 ? 8 : 12) : (span <= 16 ? 4 : (span <= 32 ? 6 : 8)); }
JacPlan jac_plan(const FgGrid& G) {
    JacPlan best = {false, false, 0, 0, 0, 0, 1e30};
    if (G.dims != 2 || !(G.fixed[2] && G.fixed[3])) return best;
    // FG_JAC_SHAPE (A/B runs): 1 = full rows only, 2 = bands only; FG_JAC_SWEEPS = sweeps per pass (clamped to what the region allows)
    static const int force_shape = [] { const char* e = getenv("FG_JAC_SHAPE"); return e ? atoi(e) : 0; }();
    static const int force_sweeps = [] { const char* e = getenv("FG_JAC_SWEEPS"); return e ? atoi(e) : 0; }();
    auto sweeps_of = [&](int span, bool xt) { int S = force_sweeps > 0 ? force_sweeps : jac_sweeps_for(span, xt); const int cap = (span - 4) / 2; return S > cap ? cap : (S < 1 ? 1 : S); };
    if (force_shape != 2 && (G.nx == 64 || G.nx == 128 || G.nx == 256 || G.nx == 512)) {
        const int rows = JAC_CELLS / G.nx;
        if (G.ny >= rows) {
            const int S = sweeps_of(rows, false), tiles = jac_tiles(G.ny, rows, S);
            const double cost = (9.0 * tiles * rows / G.ny + 2.0) / S;
            if (cost < best.cost) best = JacPlan{true, false, G.nx / 4, rows, S, tiles, cost};
        }
    }
    if (force_shape != 1 && G.fixed[0] && G.fixed[1] && (G.ny == 32 || G.ny == 64 || G.ny == 128 || G.ny == 256)) {
        const int w = JAC_CELLS / G.ny;
        if (G.nx >= w && (G.nx & 3) == 0) {
            const int S = sweeps_of(w, true), tiles = jac_tiles(G.nx, w, S);
            const double cost = (9.0 * tiles * w / G.nx + 2.0) / S;
            if (cost < best.cost) best = JacPlan{true, true, w / 4, w, S, tiles, cost};
        }
    }
    return best;
}

int launch_pass(const fg_state* s, int slot, const JacPlan& P, const JacArgs& a, hipStream_t st) {
    const dim3 grid(a.tiles, s->grid.B), block(JAC_THREADS);
#define JAC_CASE(Q, XT) if (P.q == Q && P.xt == XT) { FG_LAUNCH_P(s, slot, (k_jac_pass<Q, XT>), grid, block, JAC_LDS, st, a); return FG_OK; }
    JAC_FOR_EACH_KERNEL(JAC_CASE)
#undef JAC_CASE
    fg_set_error("Jacobi sweeps: unsupported region shape");
    return FG_ERR_UNSUPPORTED;
}

}  // namespace

static bool jac_onchip_ok(const fg_state* s, const FgBicgArgs& a) { return a.nc == 2 && jac_plan(s->grid).ok && jac_lds_ready(); }
// the streaming form takes the velocity systems of everything else (3-D; 2-D grids no region shape fits), where rA = 1 / diag is at hand
static bool jac_stream_ok(const fg_state* s, const FgBicgArgs& a) { return a.nc == s->grid.dims && a.diag == s->A && s->rA != nullptr && s->rA_epoch == s->jac_rA_epoch; }

// The sweeps start from the block velocity (the velocity systems of a step: the result goes to velocityResult).  A static rule, so
// that an env's bits do not depend on the handle's history: it pays where a cold start needs more than two passes -- the 512 x 256
// grid contracts by 0.47 per sweep, 24 sweeps from zero, 16.6 from u^n (`large_env` 2 065 / 2 100 -> 2 292 / 2 287 env-steps/s, same
// box) -- and costs where one pass suffices anyway (256 x 128: the same 12 sweeps plus 17 MB more to read, 7 553 / 7 672 -> 7 384 /
// 7 199) or sweeps are cheap to add (the streaming form: 7 sweeps either way).  Default: on-chip grids of 2^17 cells and more.
static bool jac_warm_start(const fg_state* s, const FgBicgArgs& a, bool onchip) {
    if (!(a.x == s->vel_result && a.nc == s->grid.dims && s->velocity != nullptr)) return false;
    if (s->jac_warm >= 0) return s->jac_warm != 0;
    return onchip && s->grid.n >= (1 << 17);
}

bool fg_jacobi_ok(const fg_state* s, const FgBicgArgs& a) {
    if (!s->adv_jacobi || a.precond || s->jac_prev == nullptr) return false;
    return jac_onchip_ok(s, a) || jac_stream_ok(s, a);
}

static int jacobi_stream_solve(fg_state* s, const FgBicgArgs& a, fg_solve_info* info_host, hipStream_t st, int* outcome);

// *outcome: 0 = not tried (the kind is backing off: the prepared solve state is untouched), 1 = solved here, 2 = tried and given up --
// the caller then runs BiCGStab from a cleared start vector behind a fresh k_bicg_begin
int fg_jacobi_solve(fg_state* s, const FgBicgArgs& a, fg_solve_info* info_host, hipStream_t st, int* outcome) {
    *outcome = 0;
    FgJacHist& H = s->jac_hist[a.kind & 3];
    if (H.skip > 0) { --H.skip; return FG_OK; }
    *outcome = 2;
    if (!jac_onchip_ok(s, a)) return jacobi_stream_solve(s, a, info_host, st, outcome);
    const FgGrid& G = s->grid;
    const int B = G.B, n = G.n, nsys = 2 * B;
    // Region shape and sweeps per pass: fixed by the grid, NOT by the history of the handle -- an env stops at the first pass boundary
    // where the verdict holds, so its iterate (its bits) does not depend on how many passes were enqueued ahead, and a replayed step
    // (get_state -> set_state -> step) repeats exactly (jac_sweeps_for).
    const JacPlan plan = jac_plan(G);
    const int S = plan.sweeps;
    // passes to enqueue before the first check: what the previous solve of this kind needed (the history only decides how much is
    // enqueued ahead of the poll)
    int P = H.sweeps > 0 ? (H.sweeps + S - 1) / S : 2;
    JacArgs q = {};
    q.diag = a.diag; q.off = a.off; q.rhs = a.rhs; q.acc = s->acc; q.flags = s->flags; q.info = s->info_dev; q.tol = a.tol;
    q.sweeps = S; q.nx = G.nx; q.ny = G.ny; q.n = n; q.tiles = plan.tiles;
    { static const int xr = [] { const char* e = getenv("FG_JAC_XCD"); return e ? atoi(e) : 1; }(); q.xcd_remap = xr; }
    float* work = s->w[0];
    const double bytes_sys = 0.5 * 4.0 * n * 11.0, flops_sys = 0.5 * n * 2.0 * 9.0 * S;
    int passes = 0;
    // pass p writes the result vector when p has the parity of the last planned pass (a start from a.x reads it in pass 0,
    // so that pass has to write the work buffer)
    // Start vector.  The sweeps are this library's own iteration, so their start is not the reference's BiCGStab start (zero on the
    // non-orthogonal branch): the first pass reads the BLOCK velocity, u^n -- an O(dt) distance from the solution -- straight from
    // where it lives (velocityResult is not read, so pass 0 may write it and no parity is lost).  FG_JAC_WARM=0 / other systems:
    // a.use_x0 as BiCGStab has it.
    const bool warm = jac_warm_start(s, a, true);
    const bool from_x = a.use_x0 && !warm;
    int last_parity = (P - 1) & 1;
    if (from_x && last_parity == 0) last_parity = 1;
    auto enqueue = [&](int count) -> int {
        for (int k = 0; k < count; ++k, ++passes) {
            q.pass = passes; q.zero_start = (passes == 0 && !from_x && !warm) ? 1 : 0;
            const bool to_x = ((passes & 1) == last_parity);
            q.xin = to_x ? work : a.x; q.xout = to_x ? a.x : work;
            if (warm && passes == 0) q.xin = s->velocity;
            const int slot = fg_prof_slot(s, FG_PK_JAC_PASS, s->flags, nsys, q.zero_start ? bytes_sys * 9.0 / 11.0 : bytes_sys, flops_sys, st);
            if (int rc = launch_pass(s, slot, plan, q, st)) return rc;
        }
        return FG_OK;
    };
    bool spec = false, spec_tried = false;
    auto check = [&]() -> int {
        fg_prof_prefetch(s, st);
        const FgPollOut po = fg_poll_next(&s->poll);
        q.pass = passes;
        hipLaunchKernelGGL(k_jac_check, dim3((B + 63) / 64), dim3(64), 0, st, q, s->info_pinned, s->jac_prev, B, po);
        // (something for the GPU to do while the host turns the poll around: the pressure preconditioner's factors for this 1/A)
        if (a.diag == s->A && s->jac_prefactor) if (int rc = fg_fd_rowmean_prefactor(s, a.dt, st)) return rc;
        // (and the corrector's first kernels, when this check is expected to end the solve with the iterate in the result vector)
        if (!spec_tried && s->jac_spec && s->jac_spec_fn && P == 1 && passes == 1 && last_parity == 0 && a.diag == s->A) {
            if (int rc = s->jac_spec_fn(s->jac_spec_ctx)) return rc;
            spec = true;
        }
        spec_tried = true;
        fg_htrace("jac_check_launched");
        int rc;
        if (po.gran) {      // (the verdicts arrive in the polled words themselves: unpacked to where the mirror form leaves them)
            rc = fg_poll_wait_words(&s->poll, po, 0, 10 * B, st);
            if (rc == FG_OK)
                for (int b = 0; b < B; ++b)
                    for (int c = 0; c < 2; ++c) {
                        const int i = 2 * b + c;
                        fg_solve_info& I = s->info_pinned[i];
                        const uint32_t w = fg_poll_word(&s->poll, 10 * b + 2 * c + 1);
                        I.final_residual = fg_poll_word_float(&s->poll, 10 * b + 2 * c);
                        I.used_iterations = (int32_t)(w >> 2) - 1; I.converged = (w >> 1) & 1; I.is_finite = w & 1;
                        s->jac_prev[i] = fg_poll_word_float(&s->poll, 10 * b + 4 + c);
                        s->jac_prev[nsys + 2 * i] = fg_poll_word_float(&s->poll, 10 * b + 6 + 2 * c);
                        s->jac_prev[nsys + 2 * i + 1] = fg_poll_word_float(&s->poll, 10 * b + 7 + 2 * c);
                    }
        } else rc = fg_poll_wait(&s->poll, po, 0, nsys, st);
        fg_htrace("jac_poll_done");
        return rc;
    };
    const int max_passes = (a.max_iterations / S) < JAC_MAX_PASSES ? (a.max_iterations / S > 0 ? a.max_iterations / S : 1) : JAC_MAX_PASSES;
    if (P > max_passes) P = max_passes;
    if (int rc = enqueue(P)) return rc;
    bool ok = false, failed = false;
    for (;;) {
        if (int rc = check()) return rc;
        bool all = true;
        double need = 0.0;      // passes still to go, from the contraction of the last pass: sizes what is enqueued next, nothing else
        for (int i = 0; i < nsys; ++i) {
            const fg_solve_info& I = s->info_pinned[i];
            if (!I.is_finite) failed = true;
            if (I.converged || !I.is_finite) continue;
            all = false;
            const double r1 = I.final_residual, r0 = s->jac_prev[i];
            if (r0 > 0.0 && r1 > 0.0 && r1 < r0) { const double m = log((double)a.tol / r1) / log(r1 / r0); need = m > need ? m : need; }
            else need = need > 1.0 ? need : 1.0;
            // GIVE UP from the residuals passes 0 and 1 measured -- a function of the system, not of how many passes the previous
            // solve of the kind made this one enqueue ahead: less than a factor 0.7 per pass is not the regime this is for, and beyond
            // ~48 sweeps in all the Krylov iteration is the cheaper one (BiCGStab takes 10-15 iterations on such systems)
            const double f0 = s->jac_prev[nsys + 2 * i], f1 = s->jac_prev[nsys + 2 * i + 1];
            if (f0 > 0.0 && f1 > 0.0) {
                const double c = f1 / f0;
                if (!(c < 0.7)) failed = true;
                else if ((2.0 + log((double)a.tol / f1) / log(c)) * S > 48.0) failed = true;
            }
        }
        if (all && !failed) { ok = true; break; }
        if (spec) {      // the speculated kernels ran on an iterate that is not the result and reset this solve's device state: once more, without
            if (int prc = fg_prof_collect(s, st)) return prc;
            *outcome = 5;
            return FG_OK;
        }
        if (failed) break;
        int more = (int)ceil(need);
        if (more < 1) more = 1;
        more += more & 1;                        // in pairs: the last pass of every batch writes the result vector
        if (passes >= max_passes) { failed = true; break; }      // out of passes (a fixed bound: a.max_iterations)
        if (passes + more > max_passes) more = ((max_passes - passes) & ~1) > 0 ? ((max_passes - passes) & ~1) : 2;
        if (int rc = enqueue(more)) return rc;
    }
    if (int prc = fg_prof_collect(s, st)) return prc;
    if (!ok) {
        // not the regime of the sweeps (or a non-finite system): the Krylov solver decides, and this kind waits before it tries again
        H.fails += 1; H.skip = H.fails > 6 ? 512 : (4 << H.fails); H.sweeps = 0;
        return FG_OK;
    }
    // where did every env's last iterate land?  pass p wrote the result vector iff (p & 1) == last_parity; an env that stopped with
    // the verdict of pass q + 1 has the output of pass q = (used + 1) / S - 1
    int used_max = 0, settle = 0;
    for (int b = 0; b < B; ++b) {
        const int used = s->info_pinned[2 * b].used_iterations + 1;
        if (used > 0) { settle |= (((used / S - 1) & 1) != last_parity) ? 1 : 0; used_max = used > used_max ? used : used_max; }
    }
    if (spec) {
        if (settle) { fg_set_error("fg_jacobi_solve: speculated corrector kernels with an iterate outside the result vector"); return FG_ERR_HIP; }
        s->jac_spec_done = 1;
    }
    if (settle)
        hipLaunchKernelGGL(k_jac_settle, dim3(32, B), dim3(256), 0, st, (const float*)work, a.x, (const fg_solve_info*)s->info_dev, S, last_parity, 2 * n);
    H.fails = 0;
    H.sweeps = used_max > 0 ? used_max : S;
    for (int i = 0; i < nsys; ++i)
        if (info_host) info_host[i] = s->info_pinned[i];
    FG_HIP_CHECK(hipGetLastError());
    fg_htrace("jac_return");
    *outcome = 1;
    return FG_OK;
}

// the streaming form's driver.  Check points at FIXED sweep counts (5, 7, ... 19), each with a verdict on the device: an env stops at
// the first one where all its systems are below the tolerance (or their fp32 floor), whatever was enqueued ahead -- the handle's
// history only decides where the first poll sits, so the iterate does not depend on it.  Every odd sweep from the fourth on sums the
// residual of the iterate it started from (a check reads the last two: the contraction per sweep for the host); sweep k writes
// the buffer that makes every check point end in the result vector.
static int jacobi_stream_solve(fg_state* s, const FgBicgArgs& a, fg_solve_info* info_host, hipStream_t st, int* outcome) {
    // (a check judges the iterate its last sweep STARTED from, so odd counts -- 5, 7, ... -- from a zero start, where sweep 0 may write the
    //  result vector; a start from the result vector has to write the work buffer first: even counts)
    // (from the block velocity -- jac_warm_start -- nothing but the measuring sweeps bounds the first check: 3, 5, ...)
    const bool warm = jac_warm_start(s, a, false);
    const bool from_x = a.use_x0 && !warm;
    const int FIRST = warm ? 3 : (from_x ? 6 : 5);
    constexpr int STEP = 2, CHECKS = 8;
    FgJacHist& H = s->jac_hist[a.kind & 3];
    const FgGrid& G = s->grid;
    const int B = G.B, n = G.n, nc = a.nc, nsys = B * nc;
    float* buf[2] = {a.x, s->w[0]};
    JacStreamArgs q = {};
    q.rA = s->rA; q.off = a.off; q.rhs = a.rhs; q.acc = s->acc; q.flags = s->flags; q.dt = a.dt;
    int sweeps = 0, checks = 0;
    // per sweep and system: rA and the 2 d off-diagonals shared by the nc systems of an env, rhs + x read, x written
    const double bytes_sys = 4.0 * n * ((1.0 + 2.0 * G.dims) / nc + 3.0), flops_sys = (double)n * (4.0 * G.dims + 2.0);
    auto run_to_check = [&](int upto, const FgPollOut& po_last) -> int {
        while (checks < upto) {
            const int target = FIRST + STEP * checks;
            for (; sweeps < target; ++sweeps) {
                const int w = (sweeps + (from_x ? 1 : 0)) & 1;
                q.xin = buf[w ^ 1]; q.xout = buf[w]; q.from_zero = (sweeps == 0 && !from_x && !warm) ? 1 : 0;
                if (warm && sweeps == 0) q.xin = s->velocity;
                q.measure_slot = (sweeps >= FIRST - 3 && ((sweeps - (FIRST - 3)) & 1) == 0) ? (sweeps - (FIRST - 3)) / 2 : -1;
                q.ax_slot = (sweeps == FIRST - 1) ? 12 : -1;
                // (the first sweep of a zero start reads no x and no off-diagonals: it is not sampled, which also keeps the sampling period
                //  from locking onto one sweep index of the solves)
                const int pslot = q.from_zero ? -1 : fg_prof_slot(s, FG_PK_JAC_STREAM, s->flags, nsys, bytes_sys, flops_sys, st);
                FG_DISPATCH(s, {
                    const FgLaunch L = fg_launch_geometry<DIMS, VEC>(s->grid);
                    FG_LAUNCH_P(s, pslot, (k_jac_stream<DIMS, VEC>), L.grid, dim3(FG_BLOCK), 0, st, s->grid, q, L.tiles_x, L.tiles_y, L.tiles);
                });
            }
            ++checks;
            if (checks == upto) fg_prof_prefetch(s, st);
            hipLaunchKernelGGL(k_jac_stream_check, dim3((B + 63) / 64), dim3(64), 0, st, s->acc, s->flags, s->info_dev, s->info_pinned, s->jac_prev, a.tol, nc,
                               (target - FIRST + 2) / 2, (target - FIRST) / 2, 12, sweeps, n, B, checks == upto ? po_last : FgPollOut{nullptr, 0});
        }
        FG_HIP_CHECK(hipGetLastError());
        return FG_OK;
    };
    int upto = 1;
    if (H.sweeps > FIRST) upto = 1 + (H.sweeps - FIRST + STEP - 1) / STEP;
    if (upto > CHECKS) upto = CHECKS;
    bool ok = false;
    for (;;) {
        const FgPollOut po = fg_poll_next(&s->poll);
        if (int rc = run_to_check(upto, po)) return rc;
        if (po.gran) {      // (the verdicts and both measured residuals arrive in the polled words: unpacked to where the mirror form leaves them)
            if (int rc = fg_poll_wait_words(&s->poll, po, 0, 12 * B, st)) return rc;
            for (int b = 0; b < B; ++b)
                for (int c = 0; c < nc; ++c) {
                    const int i = b * nc + c;
                    fg_solve_info& I = s->info_pinned[i];
                    const uint32_t wd = fg_poll_word(&s->poll, 12 * b + 4 * c + 1);
                    I.final_residual = fg_poll_word_float(&s->poll, 12 * b + 4 * c);
                    I.used_iterations = (int32_t)(wd >> 2) - 1; I.converged = (wd >> 1) & 1; I.is_finite = wd & 1;
                    s->jac_prev[2 * i] = fg_poll_word_float(&s->poll, 12 * b + 4 * c + 2);
                    s->jac_prev[2 * i + 1] = fg_poll_word_float(&s->poll, 12 * b + 4 * c + 3);
                }
        } else if (int rc = fg_poll_wait(&s->poll, po, 0, nsys, st)) return rc;
        bool all = true, bad = false;
        double need = 0.0;
        for (int i = 0; i < nsys; ++i) {
            const fg_solve_info& I = s->info_pinned[i];
            if (!I.is_finite) bad = true;
            if (I.converged || !I.is_finite) continue;
            const double r1 = s->jac_prev[2 * i], r0 = s->jac_prev[2 * i + 1];
            if (r1 < 0.0) continue;      // (an env that stopped at an earlier check point)
            all = false;                  // (a system the check left running: above the tolerance AND above its fp32 floor)
            if (r0 > 0.0 && r1 > 0.0) {
                const double c = sqrt(r1 / r0);
                if (!(c < 0.85)) bad = true;
                else { const double m = log((double)a.tol / r1) / log(c); need = m > need ? m : need; }
            } else {
                need = need > 2.0 ? need : 2.0;
            }
        }
        if (all && !bad) { ok = true; break; }
        if (bad) break;
        const int more = 1 + (int)(ceil(need > 1.0 ? need : 1.0) - 1) / STEP;
        if (checks + more > CHECKS) break;
        upto = checks + more;
    }
    if (int prc = fg_prof_collect(s, st)) return prc;
    if (!ok) {
        H.fails += 1; H.skip = H.fails > 6 ? 512 : (4 << H.fails); H.sweeps = 0;
        return FG_OK;      // *outcome stays 2: BiCGStab from a cleared start vector
    }
    int used_max = 0;
    for (int i = 0; i < nsys; ++i) {
        used_max = s->info_pinned[i].used_iterations + 1 > used_max ? s->info_pinned[i].used_iterations + 1 : used_max;
        if (info_host) info_host[i] = s->info_pinned[i];
        // released by the fp32-floor rule: converged with a measured residual at or above the tolerance (reported, not hidden: fg_config_dump)
        if (s->info_pinned[i].converged && s->info_pinned[i].used_iterations >= 0 && s->info_pinned[i].final_residual >= a.tol) s->jac_floor_released += 1;
    }
    H.fails = 0; H.sweeps = used_max > 0 ? used_max : FIRST;
    *outcome = 1;
    return FG_OK;
}
#endif
