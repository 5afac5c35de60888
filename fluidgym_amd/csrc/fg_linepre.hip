// y-line (tridiagonal) right preconditioner of the batched BiCGStab on the stencil-form advection-diffusion matrix, gfx950.
//
// On wall-refined grids (RBC 512 x 128, TCF) the stiff part of C = I/dt + advection - nu Laplacian is the y-diffusion across the
// thin wall cells: the plain recurrence of the reference (bicgstab_solver_kernel.cu:63-411) needs 20-35 iterations there.  The
// reference's own answer is a preconditioned solve (cuSPARSE ILU(0), bicgstab_solver_kernel.cu:191-226, 288-293: preconditionBiCG
// / BiCG_precondition_fallback of PISOtorch_diff.py:449-476).  Here M = the tridiagonal part of C along y (diagonal + the -y / +y
// off-diagonals; a periodic wrap is ignored), factorised once per solve and env by the Thomas recurrence
//   inv_j = 1 / (d_j - l_j c'_{j-1}),   c'_j = u_j inv_j
// and applied per system as  y_j = (r_j - l_j y_{j-1}) inv_j,  z_j = y_j - c'_j z_{j+1}.
// The recurrence is serial along y and parallel across the columns (x, z): a workgroup owns 64 consecutive columns, its four
// waves stage the column block in LDS with float4 loads (rows contiguous in x: coalesced), wave 0 sweeps out of LDS in 16-row
// register chunks (one dependent FMA per row), all waves store -- the scheme of k_tridiag_y_lds (fg_fdprecond.hip), with
// per-env, per-cell coefficients.  Needs nx % 4 == 0 and 3 x roundup(ny, 16) x 256 B of LDS; otherwise a streaming kernel with one
// thread per column.
#include <mutex>
#include "fg_internal.h"

namespace {

constexpr int LP_CH = 16;

struct LineArgs {
    const float* diag;   // [B][N]
    const float* lower;  // -y coefficient of every cell, env b at lower + b * lu_stride
    const float* upper;  // +y coefficient
    size_t lu_stride;
    float* inv;          // [B][N]
    float* cp;           // [B][N]
    int nx, ny, nz, dims, nc;
    const int32_t* flags;  // per system (apply) -- a system whose flag is not 0 is skipped
};

// coordinates of the float4 a lane handles in a 64-column block: t4 = first column, a4 / c4 its x / z index
struct LineCol { int lc, rsub; bool live; size_t col4; };
__device__ __forceinline__ LineCol line_col(int nx, int ny, int nz) {
    LineCol c;
    const int lane = threadIdx.x & 63;
    c.lc = 4 * (lane & 15);
    c.rsub = lane >> 4;
    int t4 = blockIdx.x * 64 + c.lc;
    c.live = t4 < nx * nz;
    if (!c.live) t4 = nx * nz - 4;
    const int a4 = t4 % nx, c4 = t4 / nx;
    c.col4 = (size_t)c4 * ny * nx + a4;
    return c;
}

// factorisation: one launch per solve, grid (column blocks, B); active = any system of the env still has flag 0
__global__ __launch_bounds__(256) void k_line_factor_y(LineArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tbuf[];   // d[nyp][64] | l[nyp][64] | u[nyp][64]
    const int b = blockIdx.y;
    bool any = false;
    for (int comp = 0; comp < a.nc; ++comp) any = any || (a.flags[b * a.nc + comp] == 0);
    if (!any) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nx = a.nx, ny = a.ny, last = ny - 1;
    const int nyp = (ny + LP_CH - 1) / LP_CH * LP_CH;
    const size_t N = (size_t)nx * ny * a.nz;
    float* ds = tbuf;
    float* ls = tbuf + (size_t)nyp * 64;
    float* us = tbuf + (size_t)2 * nyp * 64;
    const LineCol c = line_col(nx, ny, a.nz);
    const float* __restrict__ d4 = a.diag + (size_t)b * N + c.col4;
    const float* __restrict__ l4 = a.lower + (size_t)b * a.lu_stride + c.col4;
    const float* __restrict__ u4 = a.upper + (size_t)b * a.lu_stride + c.col4;
    for (int jb = wave * 32; jb < nyp; jb += 128) {
        float4 vd[8], vl[8], vu[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = min(jb + 4 * q + c.rsub, last);
            vd[q] = *reinterpret_cast<const float4*>(d4 + (size_t)j * nx);
            vl[q] = *reinterpret_cast<const float4*>(l4 + (size_t)j * nx);
            vu[q] = *reinterpret_cast<const float4*>(u4 + (size_t)j * nx);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = jb + 4 * q + c.rsub;
            if (j < nyp) {
                const int o = j * 64 + c.lc;
                const bool pad = j > last;   // padding rows: the identity (d = 1, l = u = 0)
                *reinterpret_cast<float4*>(ds + o) = pad ? make_float4(1.f, 1.f, 1.f, 1.f) : vd[q];
                *reinterpret_cast<float4*>(ls + o) = pad ? make_float4(0.f, 0.f, 0.f, 0.f) : vl[q];
                *reinterpret_cast<float4*>(us + o) = pad ? make_float4(0.f, 0.f, 0.f, 0.f) : vu[q];
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
        float cprev = 0.f;
        float* pd = ds + lane;
        const float* pl = ls + lane;
        float* pu = us + lane;
        for (int j0 = 0; j0 < nyp; j0 += LP_CH, pd += LP_CH * 64, pl += LP_CH * 64, pu += LP_CH * 64) {
            float ad[LP_CH], al[LP_CH], au[LP_CH];
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) { ad[q] = pd[q * 64]; al[q] = pl[q * 64]; au[q] = pu[q * 64]; }
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) {
                const float iv = 1.f / fmaf(-al[q], cprev, ad[q]);
                cprev = au[q] * iv;
                ad[q] = iv; au[q] = cprev;
            }
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) { pd[q * 64] = ad[q]; pu[q * 64] = au[q]; }
        }
    }
    __syncthreads();
    if (c.live) {
        float* __restrict__ i4 = a.inv + (size_t)b * N + c.col4;
        float* __restrict__ c4 = a.cp + (size_t)b * N + c.col4;
        for (int jb = wave * 32; jb < ny; jb += 128) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = jb + 4 * q + c.rsub;
                if (j <= last) {
                    *reinterpret_cast<float4*>(i4 + (size_t)j * nx) = *reinterpret_cast<const float4*>(ds + j * 64 + c.lc);
                    *reinterpret_cast<float4*>(c4 + (size_t)j * nx) = *reinterpret_cast<const float4*>(us + j * 64 + c.lc);
                }
            }
        }
    }
}

// z = M^-1 r for every system with flag 0; grid (column blocks, nsys)
// (r and z may be the same array: every row is staged in LDS before the first store -- no __restrict__ on the two)
__global__ __launch_bounds__(256) void k_line_apply_y(LineArgs a, const float* r, float* z) {
    extern __shared__ __attribute__((aligned(16))) float tbuf[];   // bs[nyp][64] | ms[nyp][64] | cs[nyp][64]
    const int sys = blockIdx.y;
    if (a.flags[sys] != 0) return;
    const int b = sys / a.nc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nx = a.nx, ny = a.ny, last = ny - 1;
    const int nyp = (ny + LP_CH - 1) / LP_CH * LP_CH;
    const size_t N = (size_t)nx * ny * a.nz;
    float* bs = tbuf;
    float* ms = tbuf + (size_t)nyp * 64;
    float* cs = tbuf + (size_t)2 * nyp * 64;
    const LineCol c = line_col(nx, ny, a.nz);
    const float* r4 = r + (size_t)sys * N + c.col4;
    const float* __restrict__ i4 = a.inv + (size_t)b * N + c.col4;
    const float* __restrict__ c4 = a.cp + (size_t)b * N + c.col4;
    const float* __restrict__ l4 = a.lower + (size_t)b * a.lu_stride + c.col4;
    for (int jb = wave * 32; jb < nyp; jb += 128) {
        float4 vr[8], vi[8], vc[8], vl[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = min(jb + 4 * q + c.rsub, last);
            vr[q] = *reinterpret_cast<const float4*>(r4 + (size_t)j * nx);
            vi[q] = *reinterpret_cast<const float4*>(i4 + (size_t)j * nx);
            vc[q] = *reinterpret_cast<const float4*>(c4 + (size_t)j * nx);
            vl[q] = *reinterpret_cast<const float4*>(l4 + (size_t)j * nx);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = jb + 4 * q + c.rsub;
            if (j < nyp) {
                const float m = (j > last) ? 0.f : 1.f;   // padding rows are neutral
                const int o = j * 64 + c.lc;
                *reinterpret_cast<float4*>(bs + o) =
                    make_float4(vr[q].x * vi[q].x * m, vr[q].y * vi[q].y * m, vr[q].z * vi[q].z * m, vr[q].w * vi[q].w * m);
                *reinterpret_cast<float4*>(ms + o) =
                    make_float4(vl[q].x * vi[q].x * m, vl[q].y * vi[q].y * m, vl[q].z * vi[q].z * m, vl[q].w * vi[q].w * m);
                *reinterpret_cast<float4*>(cs + o) = make_float4(vc[q].x * m, vc[q].y * m, vc[q].z * m, vc[q].w * m);
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
        float prev = 0.f;   // y_{-1} = 0: a periodic wrap of the first row's -y coefficient is dropped here
        float* px = bs + lane;
        const float* pm = ms + lane;
        for (int j0 = 0; j0 < nyp; j0 += LP_CH, px += LP_CH * 64, pm += LP_CH * 64) {
            float ax[LP_CH], am[LP_CH];
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) { ax[q] = px[q * 64]; am[q] = pm[q * 64]; }
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) { prev = fmaf(-am[q], prev, ax[q]); ax[q] = prev; }
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) px[q * 64] = ax[q];
        }
        prev = 0.f;         // z_{ny} = 0: likewise for the last row's +y coefficient
        px = bs + (size_t)(nyp - LP_CH) * 64 + lane;
        const float* pc = cs + (size_t)(nyp - LP_CH) * 64 + lane;
        for (int j0 = nyp - LP_CH; j0 >= 0; j0 -= LP_CH, px -= LP_CH * 64, pc -= LP_CH * 64) {
            float ax[LP_CH], ac[LP_CH];
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) { ax[q] = px[q * 64]; ac[q] = pc[q * 64]; }
#pragma unroll
            for (int q = LP_CH - 1; q >= 0; --q) { prev = fmaf(-ac[q], prev, ax[q]); ax[q] = prev; }
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) px[q * 64] = ax[q];
        }
    }
    __syncthreads();
    if (c.live) {
        float* z4 = z + (size_t)sys * N + c.col4;
        for (int jb = wave * 32; jb < ny; jb += 128) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = jb + 4 * q + c.rsub;
                if (j <= last)
                    *reinterpret_cast<float4*>(z4 + (size_t)j * nx) = *reinterpret_cast<const float4*>(bs + j * 64 + c.lc);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// LINE SWEEPS (round 6): the stationary iteration x <- (D + O_y)^-1 (b - O_x x) for the advection-diffusion systems of wall-refined
// 2-D grids (the RBC family), in place of the Helmholtz-preconditioned BiCGStab (bicgstabSolveGPU, bicgstab_solver_kernel.cu:63-411,
// at rbc_env_base.py:318's tolerance).  Point sweeps do not contract there (the y part of a row sums to 0.78 of its diagonal at the
// env's sub-step), line sweeps do: the x part is 0.18, measured contraction 0.23 per sweep (profiles/line_sweep_rbc.txt: 8 sweeps
// from u^n, 10 from zero, to 1e-5).  One sweep is ONE launch of the y-line solve above with the x stencil folded into its load phase:
// 36 B per cell (b, x and its two x neighbours, the two x coefficients, inv, c', l read; x written) against the ~80 us of a
// six-launch preconditioned BiCGStab iteration.  A measuring sweep also sums the residual b - C x of the iterate it STARTED from
// (diag, the +y coefficient and the rows above and below read on top).
// ---------------------------------------------------------------------------------------------------------------------------
struct LineSweepArgs {
    const float* off;      // [B][4][N]: faces -x, +x, -y, +y
    const float* rhs;      // [B][nc][N]
    const float* xin; float* xout;   // [B][nc][N]; xin == nullptr: the sweep from zero
    FgDacc* acc; int slot;           // slot >= 0: sum of squares of b - C x_in into acc[sys][slot]
};
template <bool MEASURE>
__global__ __launch_bounds__(256) void k_line_sweep_y(LineArgs a, LineSweepArgs w) {
    extern __shared__ __attribute__((aligned(16))) float tbuf[];   // bs[nyp][64] | ms[nyp][64] | cs[nyp][64]
    __shared__ float red[4];
    const int sys = blockIdx.y;
    if (a.flags[sys] != 0) return;
    const int b = sys / a.nc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nx = a.nx, ny = a.ny, last = ny - 1;
    const int nyp = (ny + LP_CH - 1) / LP_CH * LP_CH;
    const size_t N = (size_t)nx * ny;
    float* bs = tbuf;
    float* ms = tbuf + (size_t)nyp * 64;
    float* cs = tbuf + (size_t)2 * nyp * 64;
    const LineCol c = line_col(nx, ny, 1);
    const int i0 = (int)(c.col4 % (size_t)nx);                  // first of the lane's four columns
    const int il = i0 == 0 ? nx - 1 : i0 - 1, ir = i0 + 4 >= nx ? 0 : i0 + 4;      // (a periodic wrap; at a wall the coefficient is zero)
    const float* __restrict__ r4 = w.rhs + (size_t)sys * N + c.col4;
    const float* __restrict__ x0 = w.xin ? w.xin + (size_t)sys * N : nullptr;
    const float* __restrict__ i4 = a.inv + (size_t)b * N + c.col4;
    const float* __restrict__ c4 = a.cp + (size_t)b * N + c.col4;
    const float* __restrict__ l4 = a.lower + (size_t)b * a.lu_stride + c.col4;
    const float* __restrict__ u4 = a.upper + (size_t)b * a.lu_stride + c.col4;
    const float* __restrict__ d4 = a.diag + (size_t)b * N + c.col4;
    const float* __restrict__ om4 = w.off + (size_t)b * 4 * N + c.col4;
    const float* __restrict__ op4 = om4 + N;
    float part = 0.f;
    for (int jb = wave * 32; jb < nyp; jb += 128) {
        float4 vr[8], vi[8], vc[8], vl[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = min(jb + 4 * q + c.rsub, last);
            float4 rv = *reinterpret_cast<const float4*>(r4 + (size_t)j * nx);
            vi[q] = *reinterpret_cast<const float4*>(i4 + (size_t)j * nx);
            vc[q] = *reinterpret_cast<const float4*>(c4 + (size_t)j * nx);
            vl[q] = *reinterpret_cast<const float4*>(l4 + (size_t)j * nx);
            if (x0) {
                const float* xr = x0 + (size_t)j * nx;
                const float4 xc = *reinterpret_cast<const float4*>(xr + i0);
                const float xl = xr[il], xrr = xr[ir];
                const float4 om = *reinterpret_cast<const float4*>(om4 + (size_t)j * nx), op = *reinterpret_cast<const float4*>(op4 + (size_t)j * nx);
                rv.x -= om.x * xl + op.x * xc.y;
                rv.y -= om.y * xc.x + op.y * xc.z;
                rv.z -= om.z * xc.y + op.z * xc.w;
                rv.w -= om.w * xc.z + op.w * xrr;
                if (MEASURE && jb + 4 * q + c.rsub <= last && c.live) {
                    // b - C x = (b - O_x x) - (d x + l x[j - 1] + u x[j + 1]); l / u are zero at the walls
                    const float4 dv = *reinterpret_cast<const float4*>(d4 + (size_t)j * nx), uv = *reinterpret_cast<const float4*>(u4 + (size_t)j * nx);
                    const float4 xd = *reinterpret_cast<const float4*>(x0 + (size_t)(j > 0 ? j - 1 : j) * nx + i0);
                    const float4 xu = *reinterpret_cast<const float4*>(x0 + (size_t)(j < last ? j + 1 : j) * nx + i0);
                    const float e0 = rv.x - (dv.x * xc.x + vl[q].x * xd.x + uv.x * xu.x), e1 = rv.y - (dv.y * xc.y + vl[q].y * xd.y + uv.y * xu.y);
                    const float e2 = rv.z - (dv.z * xc.z + vl[q].z * xd.z + uv.z * xu.z), e3 = rv.w - (dv.w * xc.w + vl[q].w * xd.w + uv.w * xu.w);
                    part += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
                }
            } else if (MEASURE && jb + 4 * q + c.rsub <= last && c.live) {
                part += (rv.x * rv.x + rv.y * rv.y) + (rv.z * rv.z + rv.w * rv.w);      // x = 0: the residual is b
            }
            vr[q] = rv;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = jb + 4 * q + c.rsub;
            if (j < nyp) {
                const float m = (j > last) ? 0.f : 1.f;   // padding rows are neutral
                const int o = j * 64 + c.lc;
                *reinterpret_cast<float4*>(bs + o) =
                    make_float4(vr[q].x * vi[q].x * m, vr[q].y * vi[q].y * m, vr[q].z * vi[q].z * m, vr[q].w * vi[q].w * m);
                *reinterpret_cast<float4*>(ms + o) =
                    make_float4(vl[q].x * vi[q].x * m, vl[q].y * vi[q].y * m, vl[q].z * vi[q].z * m, vl[q].w * vi[q].w * m);
                *reinterpret_cast<float4*>(cs + o) = make_float4(vc[q].x * m, vc[q].y * m, vc[q].z * m, vc[q].w * m);
            }
        }
    }
    if (MEASURE) {
        const float sv = fg_wave_sum(part);
        if (lane == 0) red[wave] = sv;
    }
    __syncthreads();
    if (MEASURE && threadIdx.x == 64) acc_add(w.acc + (size_t)sys * FG_ACC_DOUBLES + w.slot, (double)((red[0] + red[1]) + (red[2] + red[3])));
    if (wave == 0) {
        float prev = 0.f;
        float* px = bs + lane;
        const float* pm = ms + lane;
        for (int j0 = 0; j0 < nyp; j0 += LP_CH, px += LP_CH * 64, pm += LP_CH * 64) {
            float ax[LP_CH], am[LP_CH];
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) { ax[q] = px[q * 64]; am[q] = pm[q * 64]; }
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) { prev = fmaf(-am[q], prev, ax[q]); ax[q] = prev; }
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) px[q * 64] = ax[q];
        }
        prev = 0.f;
        px = bs + (size_t)(nyp - LP_CH) * 64 + lane;
        const float* pc = cs + (size_t)(nyp - LP_CH) * 64 + lane;
        for (int j0 = nyp - LP_CH; j0 >= 0; j0 -= LP_CH, px -= LP_CH * 64, pc -= LP_CH * 64) {
            float ax[LP_CH], ac[LP_CH];
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) { ax[q] = px[q * 64]; ac[q] = pc[q * 64]; }
#pragma unroll
            for (int q = LP_CH - 1; q >= 0; --q) { prev = fmaf(-ac[q], prev, ax[q]); ax[q] = prev; }
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) px[q * 64] = ax[q];
        }
    }
    __syncthreads();
    if (c.live) {
        float* z4 = w.xout + (size_t)sys * N + c.col4;
        for (int jb = wave * 32; jb < ny; jb += 128) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = jb + 4 * q + c.rsub;
                if (j <= last)
                    *reinterpret_cast<float4*>(z4 + (size_t)j * nx) = *reinterpret_cast<const float4*>(bs + j * 64 + c.lc);
            }
        }
    }
}

// verdict behind a measuring sweep, per system (k_bicg_check's rule on the sweep's own sum), mirrors, and for the host the last two
// measured residuals plus those of the FIRST two measuring sweeps (its give-up rule reads fixed sweeps, whatever it enqueued ahead)
__global__ void k_line_sweep_check(FgDacc* __restrict__ acc, int32_t* __restrict__ flags, fg_solve_info* __restrict__ info, fg_solve_info* __restrict__ mirror,
                                   float* __restrict__ res, float tol, int slot_now, int slot_prev, int sweeps, int n, int nsys, FgPollOut poll) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    float now = -1.f, prev = -1.f;
    if (flag_ld(flags + s) == 0) {
        const FgDacc* A = acc + (size_t)s * FG_ACC_DOUBLES;
        now = (float)sqrt(acc_ld(A + slot_now) / (double)n);
        if (slot_prev >= 0) prev = (float)sqrt(acc_ld(A + slot_prev) / (double)n);
        info[s].final_residual = now;
        info[s].used_iterations = sweeps;
        if (!(now >= tol)) {
            const bool finite = isfinite(now);
            flag_st(flags + s, finite ? 1 : 2);
            info[s].converged = finite ? 1 : 0;
            info[s].is_finite = finite ? 1 : 0;
        }
        res[2 * nsys + 2 * s] = (float)sqrt(acc_ld(A + 0) / (double)n);
        res[2 * nsys + 2 * s + 1] = slot_now >= 1 ? (float)sqrt(acc_ld(A + 1) / (double)n) : -1.f;
    }
    res[2 * s] = now; res[2 * s + 1] = prev;
    mirror[s] = info[s];
    fg_poll_publish(poll, s);
}

// ---- streaming forms (any nx, ny): one thread per column, rows read in unrolled groups so that loads stay in flight
__global__ __launch_bounds__(256) void k_line_factor_y_stream(LineArgs a) {
    const int b = blockIdx.y;
    bool any = false;
    for (int comp = 0; comp < a.nc; ++comp) any = any || (a.flags[b * a.nc + comp] == 0);
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (!any || t >= a.nx * a.nz) return;
    const int nx = a.nx, ny = a.ny;
    const size_t N = (size_t)nx * ny * a.nz;
    const size_t col = (size_t)(t / nx) * ny * nx + (t % nx);
    const float* d = a.diag + (size_t)b * N + col;
    const float* l = a.lower + (size_t)b * a.lu_stride + col;
    const float* u = a.upper + (size_t)b * a.lu_stride + col;
    float* iv = a.inv + (size_t)b * N + col;
    float* cp = a.cp + (size_t)b * N + col;
    float cprev = 0.f;
#pragma unroll 8
    for (int j = 0; j < ny; ++j) {
        const size_t o = (size_t)j * nx;
        const float inv = 1.f / fmaf(-l[o], cprev, d[o]);
        cprev = u[o] * inv;
        iv[o] = inv; cp[o] = cprev;
    }
}
__global__ __launch_bounds__(256) void k_line_apply_y_stream(LineArgs a, const float* r, float* z) {
    const int sys = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (a.flags[sys] != 0 || t >= a.nx * a.nz) return;
    const int b = sys / a.nc;
    const int nx = a.nx, ny = a.ny;
    const size_t N = (size_t)nx * ny * a.nz;
    const size_t col = (size_t)(t / nx) * ny * nx + (t % nx);
    const float* rr = r + (size_t)sys * N + col;
    float* zz = z + (size_t)sys * N + col;
    const float* iv = a.inv + (size_t)b * N + col;
    const float* cp = a.cp + (size_t)b * N + col;
    const float* l = a.lower + (size_t)b * a.lu_stride + col;
    float prev = 0.f;
#pragma unroll 8
    for (int j = 0; j < ny; ++j) {
        const size_t o = (size_t)j * nx;
        prev = (rr[o] - l[o] * prev) * iv[o];
        zz[o] = prev;
    }
    prev = 0.f;
#pragma unroll 8
    for (int j = ny - 1; j >= 0; --j) {
        const size_t o = (size_t)j * nx;
        prev = zz[o] - cp[o] * prev;
        zz[o] = prev;
    }
}

// ---- coefficients of the separable Helmholtz operator  M = I/dt - nu (Dxx + Dyy + Dzz)  in the eigenbasis of the transform axes
// (fg_fd_helmholtz_apply, fg_fdprecond.hip): per env b, mode (a, c) and row j the tridiagonal system along y
//   diag = (1/dt_b - nu lam[c][a] + nu (sum of its y-face coefficients)) / s,   lower / upper = -nu (1/hy_j + 1/hy_{j-+1}) / (2 hy_j) / s,
// s = hx hz (the library's eigenvectors are H-orthonormal: Q^T H Q = I, so Q T^-1 Q^T r carries a factor 1/(hx hz)).  A FIXED y
// face contributes the one-sided coefficient 2 / hy_j^2 when the variable is prescribed there (velocity; Dirichlet scalar) and
// nothing for a Neumann scalar -- exactly the diffusion part of k_adv_build's matrix (PISO_multiblock_cuda_kernel.cu:3616-3880).
__global__ __launch_bounds__(256) void k_helm_coeffs(FgGrid g, const float* __restrict__ dt, const float* __restrict__ lam, float nu,
                                                      int wall_lo, int wall_hi, float* __restrict__ diag, float* __restrict__ lower,
                                                      float* __restrict__ upper) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= g.n || !(dt[b] > 0.f)) return;
    const int a = idx % g.nx, j = (idx / g.nx) % g.ny, c = idx / (g.nx * g.ny);
    const float rs = g.rh[0][0] * (g.dims == 3 ? g.rh[2][0] : 1.f);   // uniform transform axes
    // face coefficient = mean of the two cells' alpha = J / h^2 (getLaplaceCoefficientOrthogonal, K.cu:1224-1239), over the cell volume
    const float ry = g.rh[1][j];
    const float lo = j > 0 ? 0.5f * (ry + g.rh[1][j - 1]) * ry : (wall_lo ? 2.f * ry * ry : 0.f);
    const float hi = j < g.ny - 1 ? 0.5f * (ry + g.rh[1][j + 1]) * ry : (wall_hi ? 2.f * ry * ry : 0.f);
    const size_t o = (size_t)b * g.n + idx;
    diag[o] = (1.f / dt[b] - nu * lam[c * g.nx + a] + nu * (lo + hi)) * rs;
    lower[o] = j > 0 ? -nu * lo * rs : 0.f;
    upper[o] = j < g.ny - 1 ? -nu * hi * rs : 0.f;
}

// ---- Helmholtz preconditioner on 2-D grids, row form (round 5).  The tridiagonal systems of the separable operator have the SAME
// off-diagonals for every mode of an env -- lower_j, upper_j depend on the row only -- and diag_j(a) = base_j + sigma_a, so
//  * the coefficients need no arrays: k_helm_factor_y synthesises them in the wave that runs the Thomas factorisation (one launch
//    instead of k_helm_coeffs + k_line_factor_y, no [B][N] coefficient traffic), and
//  * the application reads r, inv, c' (16 B per cell and system instead of 20) with lower_j as a per-row scalar.
// CB = columns per workgroup of the application: 64 columns x ny rows x 3 arrays is 96 KB of LDS at ny = 128, i.e. ONE workgroup per
// CU and two rounds of latency-bound workgroups for RBC 512 x 128 x 32 (11.4 us per application, rocprofv3 round 5); 32 columns is
// 48 KB, three workgroups per CU, every workgroup of the launch resident at once.
struct HelmArgs {
    const float* dt; const float* lam; const float* rhy;     // [B] | [nx] eigenvalues of the periodic x operator | 1 / hy [ny]
    float nu, rs; int wall_lo, wall_hi;
    float* inv; float* cp; float* lower_row;                  // [B][N] | [B][N] | [B][ny]
    int nx, ny, nc;
    const int32_t* flags;                                     // nullptr (factorisation ahead of the solves): every env with dt > 0
    // second parameter set (blockIdx.z == 1 of the factorisation): the scalar and the velocity system of a PISO step differ in their
    // diffusivity and wall condition only, so both factorisations are ONE launch at the start of the step (fg_helm_factor_pair)
    float nu2; int wall_lo2, wall_hi2; float* inv2; float* cp2; float* lower_row2;
};
__device__ __forceinline__ void helm_row(const HelmArgs& a, int j, float& lo, float& hi) {
    // face coefficient = mean of the two cells' alpha = J / h^2 (getLaplaceCoefficientOrthogonal, K.cu:1224-1239), over the cell volume;
    // a wall where the variable is prescribed adds the one-sided coefficient (k_adv_build's matrix, K.cu:3616-3880)
    const float ry = a.rhy[j];
    lo = j > 0 ? 0.5f * (ry + a.rhy[j - 1]) * ry : (a.wall_lo ? 2.f * ry * ry : 0.f);
    hi = j < a.ny - 1 ? 0.5f * (ry + a.rhy[j + 1]) * ry : (a.wall_hi ? 2.f * ry * ry : 0.f);
}
// one wave per 64 modes and env: inv_j = 1 / (d_j - l_j c'_{j-1}), c'_j = u_j inv_j with d, l, u of k_helm_coeffs.  The per-row
// coefficients are formed once, in parallel, and parked in LDS: the serial loop then has no global load in it (a first version read
// rhy[j-1 .. j+1] inside the loop -- a dependent scalar-load round trip per row, 32 us for 128 rows; IEEE division on top).  The
// reciprocal is v_rcp_f32 (1 ulp): these are the factors of a PRECONDITIONER.
constexpr int HELM_MAX_NY = 256;
__global__ __launch_bounds__(64) void k_helm_factor_y(HelmArgs a) {
    __shared__ __attribute__((aligned(16))) float sl[HELM_MAX_NY], su[HELM_MAX_NY], sb[HELM_MAX_NY];
    const int b = blockIdx.y;
    if (blockIdx.z == 1) { a.nu = a.nu2; a.wall_lo = a.wall_lo2; a.wall_hi = a.wall_hi2; a.inv = a.inv2; a.cp = a.cp2; a.lower_row = a.lower_row2; }
    bool any = a.flags == nullptr;
    if (a.flags) for (int comp = 0; comp < a.nc; ++comp) any = any || (a.flags[b * a.nc + comp] == 0);
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (!any || !(a.dt[b] > 0.f)) return;
    const int nyp = (a.ny + 7) & ~7;
    for (int j = threadIdx.x; j < nyp; j += 64) {
        if (j < a.ny) {
            float lo, hi;
            helm_row(a, j, lo, hi);
            const float l = j > 0 ? -a.nu * lo * a.rs : 0.f;
            sl[j] = l;
            su[j] = j < a.ny - 1 ? -a.nu * hi * a.rs : 0.f;
            sb[j] = a.nu * (lo + hi);
            if (blockIdx.x == 0) a.lower_row[b * a.ny + j] = l;
        } else { sl[j] = 0.f; su[j] = 0.f; sb[j] = 0.f; }      // padding rows (never stored)
    }
    __syncthreads();
    const float sig = 1.f / a.dt[b] - a.nu * a.lam[col];       // (nx is a multiple of 64: every lane owns a mode)
    const size_t N = (size_t)a.nx * a.ny;
    float* __restrict__ iv = a.inv + (size_t)b * N + col;
    float* __restrict__ cp = a.cp + (size_t)b * N + col;
    float cprev = 0.f;
    // eight rows per trip: coefficients out of LDS in one batch, then the dependent chain fma -> rcp -> mul from registers
    for (int j0 = 0; j0 < nyp; j0 += 8) {
        float b8[8], l8[8], u8[8], pv[8], cv[8];
        {
            auto ld8 = [&](const float* a8, float (&o)[8]) {
                const float4 x0 = *reinterpret_cast<const float4*>(a8 + j0), x1 = *reinterpret_cast<const float4*>(a8 + j0 + 4);
                o[0] = x0.x; o[1] = x0.y; o[2] = x0.z; o[3] = x0.w; o[4] = x1.x; o[5] = x1.y; o[6] = x1.z; o[7] = x1.w;
            };
            ld8(sb, b8); ld8(sl, l8); ld8(su, u8);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float d = (sig + b8[q]) * a.rs;
            pv[q] = __builtin_amdgcn_rcpf(fmaf(-l8[q], cprev, d));
            cprev = u8[q] * pv[q];
            cv[q] = cprev;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (j0 + q < a.ny) { iv[(size_t)(j0 + q) * a.nx] = pv[q]; cp[(size_t)(j0 + q) * a.nx] = cv[q]; }
    }
}

// z = M^-1 r for every system with flag 0; grid (nx / CB, nsys); r and z may be the same array
template <int CB>
__global__ __launch_bounds__(256) void k_helm_apply_y(HelmArgs a, const float* r, float* z) {
    extern __shared__ __attribute__((aligned(16))) float tbuf[];   // bs[nyp][CB] | ms[nyp][CB] | cs[nyp][CB]
    constexpr int CBL = CB / 4, RPW = 64 / CBL, UQ = 32 / RPW;    // lanes per row | rows per wave access | accesses per wave and round (32 rows)
    const int sys = blockIdx.y;
    if (a.flags[sys] != 0) return;
    const int b = sys / a.nc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nx = a.nx, ny = a.ny, last = ny - 1;
    const int nyp = (ny + LP_CH - 1) / LP_CH * LP_CH;
    const size_t N = (size_t)nx * ny;
    float* bs = tbuf;
    float* ms = tbuf + (size_t)nyp * CB;
    float* cs = tbuf + (size_t)2 * nyp * CB;
    const int lc = 4 * (lane % CBL), rsub = lane / CBL;
    int t4 = blockIdx.x * CB + lc;
    const bool live = t4 < nx;
    if (!live) t4 = nx - 4;
    const float* r4 = r + (size_t)sys * N + t4;
    const float* __restrict__ i4 = a.inv + (size_t)b * N + t4;
    const float* __restrict__ c4 = a.cp + (size_t)b * N + t4;
    const float* __restrict__ lrow = a.lower_row + (size_t)b * ny;
    for (int jb = wave * 32; jb < nyp; jb += 128) {
        float4 vr[UQ], vi[UQ], vc[UQ];
        float vl[UQ];
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            const int j = min(jb + RPW * q + rsub, last);
            vr[q] = *reinterpret_cast<const float4*>(r4 + (size_t)j * nx);
            vi[q] = *reinterpret_cast<const float4*>(i4 + (size_t)j * nx);
            vc[q] = *reinterpret_cast<const float4*>(c4 + (size_t)j * nx);
            vl[q] = lrow[j];
        }
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            const int j = jb + RPW * q + rsub;
            if (j < nyp) {
                const float m = (j > last) ? 0.f : 1.f;   // padding rows are neutral
                const float l = vl[q] * m;
                const int o = j * CB + lc;
                *reinterpret_cast<float4*>(bs + o) =
                    make_float4(vr[q].x * vi[q].x * m, vr[q].y * vi[q].y * m, vr[q].z * vi[q].z * m, vr[q].w * vi[q].w * m);
                *reinterpret_cast<float4*>(ms + o) = make_float4(l * vi[q].x, l * vi[q].y, l * vi[q].z, l * vi[q].w);
                *reinterpret_cast<float4*>(cs + o) = make_float4(vc[q].x * m, vc[q].y * m, vc[q].z * m, vc[q].w * m);
            }
        }
    }
    __syncthreads();
    if (wave == 0 && lane < CB) {
        float prev = 0.f;
        float* px = bs + lane;
        const float* pm = ms + lane;
        for (int j0 = 0; j0 < nyp; j0 += LP_CH, px += LP_CH * CB, pm += LP_CH * CB) {
            float ax[LP_CH], am[LP_CH];
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) { ax[q] = px[q * CB]; am[q] = pm[q * CB]; }
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) { prev = fmaf(-am[q], prev, ax[q]); ax[q] = prev; }
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) px[q * CB] = ax[q];
        }
        prev = 0.f;
        px = bs + (size_t)(nyp - LP_CH) * CB + lane;
        const float* pc = cs + (size_t)(nyp - LP_CH) * CB + lane;
        for (int j0 = nyp - LP_CH; j0 >= 0; j0 -= LP_CH, px -= LP_CH * CB, pc -= LP_CH * CB) {
            float ax[LP_CH], ac[LP_CH];
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) { ax[q] = px[q * CB]; ac[q] = pc[q * CB]; }
#pragma unroll
            for (int q = LP_CH - 1; q >= 0; --q) { prev = fmaf(-ac[q], prev, ax[q]); ax[q] = prev; }
#pragma unroll
            for (int q = 0; q < LP_CH; ++q) px[q * CB] = ax[q];
        }
    }
    __syncthreads();
    if (live) {
        float* z4 = z + (size_t)sys * N + t4;
        for (int jb = wave * 32; jb < ny; jb += 128) {
#pragma unroll
            for (int q = 0; q < UQ; ++q) {
                const int j = jb + RPW * q + rsub;
                if (j <= last)
                    *reinterpret_cast<float4*>(z4 + (size_t)j * nx) = *reinterpret_cast<const float4*>(bs + j * CB + lc);
            }
        }
    }
}

bool line_lds_ready(size_t bytes) {   // dynamic LDS above 64 KB needs an explicit opt-in per kernel and per device
    static std::mutex mu;
    static size_t granted_dev[64] = {0};
    static bool failed_dev[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return false; }
    std::lock_guard<std::mutex> lock(mu);
    size_t& granted = granted_dev[dev];
    bool& failed = failed_dev[dev];
    if (bytes <= granted) return true;
    if (failed) return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_line_apply_y), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_line_factor_y), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        failed = true;
        return false;
    }
    granted = bytes;
    return true;
}

LineArgs line_args(const fg_state* s, const float* diag, const float* off, int nc) {
    LineArgs a;
    a.diag = diag; a.inv = s->line_inv; a.cp = s->line_cp;
    if (off) {   // the tridiagonal part of the stencil matrix itself: faces 2 / 3 of [B][2 d][N]
        a.lower = off + (size_t)2 * s->grid.n; a.upper = off + (size_t)3 * s->grid.n; a.lu_stride = (size_t)2 * s->grid.dims * s->grid.n;
    } else {     // synthesized coefficients of the Helmholtz preconditioner (k_helm_coeffs): [B][N] each
        a.lower = s->helm_lower; a.upper = s->helm_upper; a.lu_stride = (size_t)s->grid.n;
    }
    a.nx = s->grid.nx; a.ny = s->grid.ny; a.nz = s->grid.nz; a.dims = s->grid.dims; a.nc = nc;
    a.flags = s->flags;
    return a;
}

size_t line_lds_bytes(const fg_state* s) {
    const int nyp = (s->grid.ny + LP_CH - 1) / LP_CH * LP_CH;
    return (size_t)3 * nyp * 64 * sizeof(float);
}
bool line_use_lds(const fg_state* s) {
    const size_t bytes = line_lds_bytes(s);
    return (s->grid.nx & 3) == 0 && bytes <= 160 * 1024 && line_lds_ready(bytes);
}

}  // namespace

int fg_line_alloc(fg_state* s) {
    if (s->line_inv) return FG_OK;
    const size_t count = (size_t)s->grid.B * s->grid.n;
    FG_HIP_CHECK(hipMalloc(&s->line_inv, sizeof(float) * count));
    FG_HIP_CHECK(hipMalloc(&s->line_cp, sizeof(float) * count));
    return FG_OK;
}

int fg_helm_alloc(fg_state* s) {
    if (int rc = fg_line_alloc(s)) return rc;
    if (s->helm_diag) return FG_OK;
    const size_t count = (size_t)s->grid.B * s->grid.n;
    FG_HIP_CHECK(hipMalloc(&s->helm_diag, sizeof(float) * count));
    FG_HIP_CHECK(hipMalloc(&s->helm_lower, sizeof(float) * count));
    FG_HIP_CHECK(hipMalloc(&s->helm_upper, sizeof(float) * count));
    FG_HIP_CHECK(hipMalloc(&s->helm_tmp, sizeof(float) * count * s->grid.dims));
    FG_HIP_CHECK(hipMalloc(&s->helm_lower_row, sizeof(float) * (size_t)s->grid.B * s->grid.ny));
    FG_HIP_CHECK(hipMalloc(&s->helm_lower_row2, sizeof(float) * (size_t)s->grid.B * s->grid.ny));
    FG_HIP_CHECK(hipMalloc(&s->line_inv2, sizeof(float) * count));
    FG_HIP_CHECK(hipMalloc(&s->line_cp2, sizeof(float) * count));
    {   // 1 / (hx hz) of the uniform transform axes (the H-orthonormal eigenvectors carry it: k_helm_coeffs)
        float rx = 1.f, rz = 1.f;
        FG_HIP_CHECK(hipMemcpy(&rx, s->d_rh[0], sizeof(float), hipMemcpyDeviceToHost));
        if (s->grid.dims == 3) FG_HIP_CHECK(hipMemcpy(&rz, s->d_rh[2], sizeof(float), hipMemcpyDeviceToHost));
        s->helm_rs = rx * rz;
    }
    return FG_OK;
}

// row form (k_helm_factor_y / k_helm_apply_y above): 2-D, nx a multiple of 4 and of the column block, the column block in LDS
static int helm_cb(const fg_state* s) {      // columns per workgroup of the application, 0 = the array form (k_helm_coeffs + line kernels)
    if (s->grid.dims != 2 || (s->grid.nx & 63) != 0 || s->grid.ny > HELM_MAX_NY || s->helm_rowform_off) return 0;
    const int nyp = (s->grid.ny + LP_CH - 1) / LP_CH * LP_CH;
    const int cb = s->helm_cb_pref == 64 ? 64 : 32;
    const size_t bytes = (size_t)3 * nyp * cb * sizeof(float);
    if (bytes > 160 * 1024) return 0;
    static std::mutex mu;
    static size_t granted_dev[64] = {0};      // per device: the attribute belongs to the device current when it is set
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 0; }
    std::lock_guard<std::mutex> lock(mu);
    size_t& granted = granted_dev[dev];
    if (bytes > granted) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_helm_apply_y<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_helm_apply_y<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        granted = bytes;
    }
    return cb;
}

// both factorisations of a PISO step with a passive scalar (set 0: scalar, set 1: velocity) in one launch, ahead of the solves;
// fg_helm_factor then finds the record and launches nothing.  Row form only (returns FG_OK without a record otherwise).
int fg_helm_factor_pair(fg_state* s, const float* dt, const float nu[2], const int wall_lo[2], const int wall_hi[2], hipStream_t st) {
    s->helm_pre_mask = 0;
    if (!helm_cb(s) || !s->line_inv2) return FG_OK;
    HelmArgs a = {};
    a.dt = dt; a.lam = s->fd_lam; a.rhy = s->grid.rh[1]; a.rs = s->helm_rs; a.nx = s->grid.nx; a.ny = s->grid.ny; a.nc = 1; a.flags = nullptr;
    a.nu = nu[0]; a.wall_lo = wall_lo[0]; a.wall_hi = wall_hi[0]; a.inv = s->line_inv; a.cp = s->line_cp; a.lower_row = s->helm_lower_row;
    a.nu2 = nu[1]; a.wall_lo2 = wall_lo[1]; a.wall_hi2 = wall_hi[1]; a.inv2 = s->line_inv2; a.cp2 = s->line_cp2; a.lower_row2 = s->helm_lower_row2;
    hipLaunchKernelGGL(k_helm_factor_y, dim3((s->grid.nx + 63) / 64, s->grid.B, 2), dim3(64), 0, st, a);
    FG_HIP_CHECK(hipGetLastError());
    s->helm_pre_mask = 3; s->helm_pre_dt = dt;
    for (int k = 0; k < 2; ++k) { s->helm_pre_nu[k] = nu[k]; s->helm_pre_walls[k][0] = wall_lo[k]; s->helm_pre_walls[k][1] = wall_hi[k]; }
    return FG_OK;
}

// coefficients + Thomas factorisation of the Helmholtz preconditioner for this solve (dt per env, nu of the solve); kind = the
// factor set a record of fg_helm_factor_pair may already hold (0 scalar, 1 velocity)
int fg_helm_factor(fg_state* s, const float* dt, float nu, int wall_lo, int wall_hi, int nc, hipStream_t st, int kind) {
    s->helm_cb = helm_cb(s);
    s->helm_set = 0;
    if (s->helm_cb && kind >= 0 && kind < 2 && ((s->helm_pre_mask >> kind) & 1) && s->helm_pre_dt == dt && s->helm_pre_nu[kind] == nu &&
        s->helm_pre_walls[kind][0] == wall_lo && s->helm_pre_walls[kind][1] == wall_hi) {
        s->helm_pre_mask &= ~(1 << kind);
        s->helm_set = kind;
        return FG_OK;
    }
    if (kind >= 0 && kind < 2) s->helm_pre_mask &= ~1;      // (set 0 is overwritten below)
    if (s->helm_cb) {
        HelmArgs a = {};
        a.dt = dt; a.lam = s->fd_lam; a.rhy = s->grid.rh[1]; a.nu = nu; a.wall_lo = wall_lo; a.wall_hi = wall_hi;
        a.rs = s->helm_rs;
        a.inv = s->line_inv; a.cp = s->line_cp; a.lower_row = s->helm_lower_row; a.nx = s->grid.nx; a.ny = s->grid.ny; a.nc = nc; a.flags = s->flags;
        hipLaunchKernelGGL(k_helm_factor_y, dim3((s->grid.nx + 63) / 64, s->grid.B), dim3(64), 0, st, a);
        FG_HIP_CHECK(hipGetLastError());
        return FG_OK;
    }
    hipLaunchKernelGGL(k_helm_coeffs, dim3((s->grid.n + 255) / 256, s->grid.B), dim3(256), 0, st, s->grid, dt, (const float*)s->fd_lam, nu,
                       wall_lo, wall_hi, s->helm_diag, s->helm_lower, s->helm_upper);
    return fg_line_factor(s, s->helm_diag, nullptr, nc, st);
}

// z = M^-1 r (the per-mode Thomas solves of the Helmholtz operator factorised by the last fg_helm_factor), nc systems per env
int fg_helm_apply(fg_state* s, int nc, const float* r, float* z, hipStream_t st) {
    if (!s->helm_cb) return fg_line_apply(s, s->helm_diag, nullptr, nc, r, z, st);
    HelmArgs a = {};
    a.inv = s->helm_set ? s->line_inv2 : s->line_inv; a.cp = s->helm_set ? s->line_cp2 : s->line_cp;
    a.lower_row = s->helm_set ? s->helm_lower_row2 : s->helm_lower_row; a.nx = s->grid.nx; a.ny = s->grid.ny; a.nc = nc; a.flags = s->flags;
    const int nsys = s->grid.B * nc;
    // per system and cell: r, inv, c' read + z written
    const int slot = fg_prof_slot(s, FG_PK_LINE, s->flags, nsys, 16.0 * s->grid.n, 5.0 * s->grid.n, st);
    const int nyp = (s->grid.ny + LP_CH - 1) / LP_CH * LP_CH;
    if (s->helm_cb == 64)
        FG_LAUNCH_P(s, slot, k_helm_apply_y<64>, dim3(s->grid.nx / 64, nsys), dim3(256), (size_t)3 * nyp * 64 * sizeof(float), st, a, r, z);
    else
        FG_LAUNCH_P(s, slot, k_helm_apply_y<32>, dim3(s->grid.nx / 32, nsys), dim3(256), (size_t)3 * nyp * 32 * sizeof(float), st, a, r, z);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_line_factor(fg_state* s, const float* diag, const float* off, int nc, hipStream_t st) {
    const LineArgs a = line_args(s, diag, off, nc);
    const int cols = s->grid.nx * s->grid.nz;
    if (line_use_lds(s))
        hipLaunchKernelGGL(k_line_factor_y, dim3((cols + 63) / 64, s->grid.B), dim3(256), line_lds_bytes(s), st, a);
    else
        hipLaunchKernelGGL(k_line_factor_y_stream, dim3((cols + 255) / 256, s->grid.B), dim3(256), 0, st, a);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

int fg_line_apply(fg_state* s, const float* diag, const float* off, int nc, const float* r, float* z, hipStream_t st) {
    const LineArgs a = line_args(s, diag, off, nc);
    const int cols = s->grid.nx * s->grid.nz, nsys = s->grid.B * nc;
    // per system and cell: r, inv, c', l read + z written
    const int slot = fg_prof_slot(s, FG_PK_LINE, s->flags, nsys, 20.0 * s->grid.n, 5.0 * s->grid.n, st);
    if (line_use_lds(s))
        FG_LAUNCH_P(s, slot, k_line_apply_y, dim3((cols + 63) / 64, nsys), dim3(256), line_lds_bytes(s), st, a, r, z);
    else
        FG_LAUNCH_P(s, slot, k_line_apply_y_stream, dim3((cols + 255) / 256, nsys), dim3(256), 0, st, a, r, z);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

// ---- line sweeps: driver.  *outcome: 0 not tried (the kind is backing off), 1 solved, 2 given up (the caller runs its Krylov solver
// from a cleared start vector behind a fresh k_bicg_begin).  Check points at FIXED sweep counts (6, 8, ... 20) with the verdict on the
// device, per system; the host's give-up rule reads the residuals the FIRST two measuring sweeps left (sweeps 3 and 5), so which
// solver runs is a function of the system, not of where the previous solve of the kind made this one poll first.
bool fg_linesweep_ok(const fg_state* s, const FgBicgArgs& a) {
    return s->adv_linesweep && a.precond == 2 && s->grid.dims == 2 && s->grid.nz == 1 && a.nc <= 2 && s->jac_prev != nullptr && (s->grid.nx & 3) == 0 &&
           line_lds_bytes(s) <= 160 * 1024;
}
int fg_linesweep_solve(fg_state* s, const FgBicgArgs& a, fg_solve_info* info_host, hipStream_t st, int* outcome) {
    *outcome = 0;
    FgJacHist& H = s->jac_hist[a.kind & 3];
    if (H.skip > 0) { --H.skip; return FG_OK; }
    *outcome = 2;
    if (!line_use_lds(s)) return FG_OK;
    if (int rc = fg_line_alloc(s)) return rc;
    constexpr int FIRST = 6, STEP = 2, CHECKS = 8;
    const FgGrid& G = s->grid;
    const int B = G.B, n = G.n, nc = a.nc, nsys = B * nc, cols = G.nx;
    if (int rc = fg_line_factor(s, a.diag, a.off, nc, st)) return rc;      // (D + O_y) = L U per env, once per solve
    const LineArgs la = line_args(s, a.diag, a.off, nc);
    float* buf[2] = {a.x, s->w[0]};      // sweep k writes buf[(k + 1) & 1]: the check points are even counts, so each ends in a.x
    LineSweepArgs w;
    w.off = a.off; w.rhs = a.rhs; w.acc = s->acc;
    int sweeps = 0, checks = 0;
    const double bytes_sys = 36.0 * n, flops_sys = 12.0 * n;
    auto run_to_check = [&](int upto, const FgPollOut& po_last) -> int {
        while (checks < upto) {
            const int target = FIRST + STEP * checks;
            for (; sweeps < target; ++sweeps) {
                const int o = (sweeps + 1) & 1;
                w.xin = (sweeps == 0 && !a.use_x0) ? nullptr : buf[o ^ 1];
                w.xout = buf[o];
                w.slot = (sweeps >= FIRST - 3 && (sweeps & 1)) ? (sweeps - (FIRST - 3)) / 2 : -1;      // sweeps 3, 5, ... -> slots 0, 1, ...
                const int pslot = fg_prof_slot(s, FG_PK_LINE, s->flags, nsys, bytes_sys, flops_sys, st);
                if (w.slot >= 0) FG_LAUNCH_P(s, pslot, (k_line_sweep_y<true>), dim3((cols + 63) / 64, nsys), dim3(256), line_lds_bytes(s), st, la, w);
                else FG_LAUNCH_P(s, pslot, (k_line_sweep_y<false>), dim3((cols + 63) / 64, nsys), dim3(256), line_lds_bytes(s), st, la, w);
            }
            ++checks;
            const int now = (target - 1 - (FIRST - 3)) / 2;
            if (checks == upto) fg_prof_prefetch(s, st);
            hipLaunchKernelGGL(k_line_sweep_check, dim3((nsys + 63) / 64), dim3(64), 0, st, s->acc, s->flags, s->info_dev, s->info_pinned, s->jac_prev, a.tol,
                               now, now - 1, sweeps, n, nsys, checks == upto ? po_last : FgPollOut{nullptr, 0});
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
        if (int rc = fg_poll_wait(&s->poll, po, 0, nsys, st)) return rc;
        bool all = true, bad = false;
        double need = 0.0;
        for (int i = 0; i < nsys; ++i) {
            const fg_solve_info& I = s->info_pinned[i];
            if (!I.is_finite) bad = true;
            const double r1 = s->jac_prev[2 * i], r0 = s->jac_prev[2 * i + 1];
            if (I.converged || !I.is_finite || r1 < 0.0) continue;      // (settled here or at an earlier check point, or inactive)
            all = false;
            if (r0 > 0.0 && r1 > 0.0 && r1 < r0) { const double m = log((double)a.tol / r1) / log(sqrt(r1 / r0)); need = m > need ? m : need; }
            else need = need > 2.0 ? need : 2.0;
            // give up from the residuals of sweeps 3 and 5 (two sweeps apart): less than 0.7 per sweep is not the regime this is for
            const double f0 = s->jac_prev[2 * nsys + 2 * i], f1 = s->jac_prev[2 * nsys + 2 * i + 1];
            if (f0 > 0.0 && f1 > 0.0 && !(sqrt(f1 / f0) < 0.7)) bad = true;
        }
        if (all && !bad) { ok = true; break; }
        if (bad) break;
        const int more = 1 + (int)(ceil(need > 1.0 ? need : 1.0) - 1) / STEP;
        if (checks >= CHECKS) break;
        upto = checks + more > CHECKS ? CHECKS : checks + more;
    }
    if (int prc = fg_prof_collect(s, st)) return prc;
    if (!ok) {
        H.fails += 1; H.skip = H.fails > 6 ? 512 : (4 << H.fails); H.sweeps = 0;
        return FG_OK;
    }
    int used_max = 0;
    for (int i = 0; i < nsys; ++i) {
        used_max = s->info_pinned[i].used_iterations > used_max ? s->info_pinned[i].used_iterations : used_max;
        if (info_host) info_host[i] = s->info_pinned[i];
    }
    H.fails = 0; H.sweeps = used_max > 0 ? used_max : FIRST;
    *outcome = 1;
    return FG_OK;
}
