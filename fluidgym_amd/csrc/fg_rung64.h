// The reference's `solver_double_fallback` rung on the single-block path (_linear_solve_wrapper, pict/PISOtorch_diff.py:418-445): a
// solve that FAILED in single precision -- an advection-diffusion BiCGStab that ended unconverged, a pressure CG (which runs with
// returnBestResult) that ended non-finite -- is repeated in double precision on the SAME matrix and right-hand side
// (`csrMat.toType(dp)`, `rhs.to(dp)`: the fp32 entries promoted, not re-assembled), from a cleared result, with the same
// recurrence, criterion, tolerance and iteration cap; the result is cast back.  This is a rare path: one system at a time, plain
// kernels, scalars read back by the host (what the reference's solvers do on every solve).  Dot products go through FgDacc, so a
// replayed state takes the same rung to the same bits.  fp32 library only (the fp64 build has nothing to fall back to).
#pragma once
#include "fg_internal.h"

#if !FG_F64
namespace {

constexpr int R64_VECS = 8;

__global__ void k64_dot(const double* __restrict__ a, const double* __restrict__ b, int n, FgDacc* acc) {
    double v = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v += a[i] * b[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) acc_add(acc, part[0] + part[1] + part[2] + part[3]);
}
// y = a x + b z   (y may alias x or z)
__global__ void k64_axpby(double* y, double a, const double* x, double b, const double* z, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = a * x[i] + b * z[i];
}
__global__ void k64_from_f32(double* __restrict__ y, const float* __restrict__ x, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = (double)x[i];
}
__global__ void k64_to_f32(float* __restrict__ y, const double* __restrict__ x, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = (float)x[i];
}

struct R64 {
    fg_state* s; hipStream_t st; int n; double* v[R64_VECS]; int err = FG_OK;
    int init(fg_state* s_, hipStream_t st_) {
        s = s_; st = st_; n = s->grid.n;
        if (!s->r64_buf) {
            FG_HIP_CHECK(hipMalloc(&s->r64_buf, sizeof(double) * (size_t)R64_VECS * n));
            FG_HIP_CHECK(hipMalloc(&s->r64_acc, sizeof(FgDacc)));
        }
        for (int i = 0; i < R64_VECS; ++i) v[i] = s->r64_buf + (size_t)i * n;
        return FG_OK;
    }
    dim3 grid() const { int g = (n + 255) / 256; return dim3(g > 2048 ? 2048 : g); }
    double dot(const double* a, const double* b) {
        FgDacc host;
        if (hipMemsetAsync(s->r64_acc, 0, sizeof(FgDacc), st) != hipSuccess) { err = FG_ERR_HIP; return NAN; }
        hipLaunchKernelGGL(k64_dot, grid(), dim3(256), 0, st, a, b, n, s->r64_acc);
        if (hipMemcpyAsync(&host, s->r64_acc, sizeof(FgDacc), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { err = FG_ERR_HIP; return NAN; }
        return fg_dacc_host_value(host);
    }
    void axpby(double* y, double a, const double* x, double b, const double* z) {
        hipLaunchKernelGGL(k64_axpby, grid(), dim3(256), 0, st, y, a, x, b, z, n);
    }
    void load(double* y, const float* x) { hipLaunchKernelGGL(k64_from_f32, grid(), dim3(256), 0, st, y, x, n); }
    void store(float* y, const double* x) { hipLaunchKernelGGL(k64_to_f32, grid(), dim3(256), 0, st, y, x, n); }
    void zero(double* y) { (void)hipMemsetAsync(y, 0, sizeof(double) * (size_t)n, st); }
};

// neighbour values of a double vector at the thread's cell (VEC = 1 context)
template <int DIMS>
struct R64Nbr { double c, xm, xp, ym, yp, zm, zp; };
template <int DIMS>
__device__ __forceinline__ R64Nbr<DIMS> r64_gather(const double* __restrict__ q, const FgCtx<DIMS, 1>& c) {
    R64Nbr<DIMS> n;
    n.c = q[c.idx]; n.xm = q[c.ixm]; n.xp = q[c.ixp]; n.ym = q[c.iym]; n.yp = q[c.iyp];
    n.zm = DIMS == 3 ? q[c.izm] : 0.0; n.zp = DIMS == 3 ? q[c.izp] : 0.0;
    return n;
}

}  // namespace
#endif
