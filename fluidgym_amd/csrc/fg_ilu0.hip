// ILU(0) of the stencil-form advection-diffusion matrix as the right preconditioner of the BiCGStab: the reference's
// `preconditionBiCG` / `BiCG_precondition_fallback` rung is BiCGStab preconditioned by cuSPARSE's incomplete LU without fill
// (cusparseScsrilu02 + two cusparseSpSV per application, bicgstab_solver_kernel.cu:191-226, 288-293; PISOtorch_diff.py:449-476).
//
// On a 5- / 7-point stencil in natural ordering (x fastest) ILU(0) has a closed form: no position of the pattern is updated by
// an elimination step except the diagonal (an update of a_cj by row n needs (n, j) AND (c, j) in the pattern: never, for axes of at
// least four cells, periodic wraps included), so  L = I + strict_lower(C) D^-1,  U = D + strict_upper(C)  with the modified diagonal
//     d_c = a_cc - sum_{n in lower(c)} a_cn a_nc / d_n .
// "lower(c)" are the stencil neighbours with a smaller cell index: -x, -y, -z, and across a periodic wrap the +x / +y / +z
// neighbour of the last cell of that axis.  Every dependency of cell (i, j, k) lies on a hyperplane i + j + k of smaller index
// (wraps too), so the factorisation and both triangular solves are sweeps over the nx + ny + nz - 2 hyperplanes -- what cuSPARSE's
// level scheduling finds for this matrix.  One workgroup per (env, component) walks the hyperplanes with a barrier between
// them: a sequential algorithm, ~1 us per hyperplane, i.e. 0.5 - 1 ms per application on the bench grids against 60 us for a whole
// plain iteration.  It is here because the reference's rung is THIS preconditioner (tests/test_gpu_ilu0.py holds it against a generic
// CSR ILU(0) in NumPy); the y-line solve of fg_linepre.hip stays the default of the rung (fg_set_advection_preconditioner).
#include "fg_internal.h"

namespace {

struct IluGeo {
    int nx, ny, nz, n, dims;
    int fixed[6];
};

// neighbour of cell (i, j, k) across face f, or -1 (FIXED boundary: its coefficient is zero, k_adv_build)
__device__ __forceinline__ int ilu_nbr(const IluGeo& g, int i, int j, int k, int f) {
    int p[3] = {i, j, k};
    const int ext[3] = {g.nx, g.ny, g.nz};
    const int ax = f >> 1, up = f & 1;
    if (up) { if (p[ax] + 1 < ext[ax]) p[ax] += 1; else if (g.fixed[f]) return -1; else p[ax] = 0; }
    else { if (p[ax] > 0) p[ax] -= 1; else if (g.fixed[f]) return -1; else p[ax] = ext[ax] - 1; }
    return p[0] + g.nx * (p[1] + g.ny * p[2]);
}

// MODE 0: modified diagonal; 1: forward (unit lower) sweep out = L^-1 in; 2: backward sweep out = U^-1 out (in place)
template <int MODE>
__device__ __forceinline__ void ilu_cell(const IluGeo& g, int i, int j, int k, const float* __restrict__ diag, const float* __restrict__ off,
                                         const float* __restrict__ dmod, float* __restrict__ dmod_out, const float* __restrict__ in,
                                         float* __restrict__ out) {
    const int c = i + g.nx * (j + g.ny * k), F = 2 * g.dims;
    if (MODE == 0) {
        float d = diag[c];
        for (int f = 0; f < F; ++f) {
            const int n = ilu_nbr(g, i, j, k, f);
            if (n >= 0 && n < c) d -= off[(size_t)f * g.n + c] * off[(size_t)(f ^ 1) * g.n + n] / dmod_out[n];
        }
        dmod_out[c] = d;
    } else if (MODE == 1) {
        float y = in[c];
        for (int f = 0; f < F; ++f) {
            const int n = ilu_nbr(g, i, j, k, f);
            if (n >= 0 && n < c) y -= off[(size_t)f * g.n + c] / dmod[n] * out[n];
        }
        out[c] = y;
    } else {
        float z = out[c];
        for (int f = 0; f < F; ++f) {
            const int n = ilu_nbr(g, i, j, k, f);
            if (n >= 0 && n > c) z -= off[(size_t)f * g.n + c] * out[n];
        }
        out[c] = z / dmod[c];
    }
}

// hyperplane sweep; grid = (systems per env, B).  MODE 0 runs with one system per env (the matrix belongs to the env).
template <int MODE>
__global__ __launch_bounds__(1024) void k_ilu0(IluGeo g, int nc, const float* __restrict__ diag, const float* __restrict__ off,
                                               float* __restrict__ dmod, const float* __restrict__ in, float* __restrict__ out,
                                               const int32_t* __restrict__ flags) {
    const int b = blockIdx.y, comp = blockIdx.x, sys = b * nc + comp;
    if (MODE != 0 && flags && flags[sys] != 0) return;
    const float* dg = diag + (size_t)b * g.n;
    const float* of = off + (size_t)b * 2 * g.dims * g.n;
    float* dm = dmod + (size_t)b * g.n;
    const float* src = MODE == 1 ? in + (size_t)sys * g.n : nullptr;
    float* dst = MODE == 0 ? nullptr : out + (size_t)sys * g.n;
    const int planes = g.nx + g.ny + g.nz - 2, jk = g.ny * g.nz;
    for (int step = 0; step < planes; ++step) {
        const int L = (MODE == 2) ? planes - 1 - step : step;
        for (int m = threadIdx.x; m < jk; m += blockDim.x) {
            const int j = m % g.ny, k = m / g.ny, i = L - j - k;
            if (i >= 0 && i < g.nx) ilu_cell<MODE>(g, i, j, k, dg, of, dm, dm, src, dst);
        }
        __syncthreads();   // (orders the global writes of a hyperplane before the reads of the next within the workgroup)
    }
}

IluGeo geo_of(const fg_state* s) {
    IluGeo g;
    g.nx = s->grid.nx; g.ny = s->grid.ny; g.nz = s->grid.dims == 3 ? s->grid.nz : 1; g.n = s->grid.n; g.dims = s->grid.dims;
    for (int f = 0; f < 6; ++f) g.fixed[f] = s->grid.fixed[f];
    return g;
}

}  // namespace

int fg_ilu_alloc(fg_state* s) {
    if (s->ilu_d) return FG_OK;
    const FgGrid& G = s->grid;
    for (int a = 0; a < G.dims; ++a) {
        const int ext = a == 0 ? G.nx : (a == 1 ? G.ny : G.nz);
        if (ext < 4) { fg_set_error("ILU(0) preconditioner: every axis needs at least four cells (closed form of the factorisation)"); return FG_ERR_UNSUPPORTED; }
    }
    FG_HIP_CHECK(hipMalloc(&s->ilu_d, sizeof(float) * (size_t)G.B * G.n));
    return FG_OK;
}

// modified diagonal of every env's matrix (once per solve: the matrix is the solve's)
int fg_ilu_factor(fg_state* s, const float* diag, const float* off, hipStream_t st) {
    hipLaunchKernelGGL(k_ilu0<0>, dim3(1, s->grid.B), dim3(1024), 0, st, geo_of(s), 1, diag, off, s->ilu_d, (const float*)nullptr,
                       (float*)nullptr, (const int32_t*)nullptr);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

// z = U^-1 L^-1 r for the nc systems of every env (systems whose flag is set are skipped)
int fg_ilu_apply(fg_state* s, const float* diag, const float* off, int nc, const float* r, float* z, hipStream_t st) {
    const IluGeo g = geo_of(s);
    hipLaunchKernelGGL(k_ilu0<1>, dim3(nc, s->grid.B), dim3(1024), 0, st, g, nc, diag, off, s->ilu_d, r, z, (const int32_t*)s->flags);
    hipLaunchKernelGGL(k_ilu0<2>, dim3(nc, s->grid.B), dim3(1024), 0, st, g, nc, diag, off, s->ilu_d, r, z, (const int32_t*)s->flags);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}
