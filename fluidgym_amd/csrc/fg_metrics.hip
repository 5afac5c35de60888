// Cell transforms from vertex coordinates (CoordsToTransforms, grid_gen.cu:298-390):
//   M[i][k] = x_i(centre of face +k) - x_i(centre of face -k), Minv = M^-1, det = |M|
// stored per cell as [M (d*d) | Minv (d*d) | det] (domain_structs_gpu.h:160-168).
// Runs once per reset; one thread per cell, coalesced on the x-fastest vertex array.
#include "fg_internal.h"

namespace {

template <int D>
__global__ __launch_bounds__(FG_BLOCK) void k_coords_to_transforms(const float* __restrict__ coords,
                                                                    float* __restrict__ tr, int nx, int ny, int nz) {
    const int n = nx * ny * nz;
    const int vx = nx + 1, vy = ny + 1, vz = (D == 3) ? nz + 1 : 1;
    const size_t vplane = (size_t)vx * vy, vtot = vplane * vz;
    constexpr int TS = 2 * D * D + 1;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const int i = idx % nx, j = (idx / nx) % ny, k = idx / (nx * ny);
        float M[D][D];
#pragma unroll
        for (int kk = 0; kk < D; ++kk) {
            float hi[D], lo[D];
#pragma unroll
            for (int c = 0; c < D; ++c) hi[c] = lo[c] = 0.f;
#pragma unroll
            for (int v = 0; v < (1 << D); ++v) {
                const int ox = v & 1, oy = (v >> 1) & 1, oz = (D == 3) ? ((v >> 2) & 1) : 0;
                const size_t vi = (size_t)(k + oz) * vplane + (size_t)(j + oy) * vx + (i + ox);
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    const float x = coords[(size_t)c * vtot + vi];
                    if ((v >> kk) & 1) hi[c] += x; else lo[c] += x;
                }
            }
            const float norm = 1.f / (float)(1 << (D - 1));
#pragma unroll
            for (int c = 0; c < D; ++c) M[c][kk] = (hi[c] - lo[c]) * norm;
        }
        float* out = tr + (size_t)idx * TS;
        float det, Mi[D][D];
        if constexpr (D == 2) {
            det = M[0][0] * M[1][1] - M[0][1] * M[1][0];
            const float r = 1.f / det;
            Mi[0][0] = M[1][1] * r; Mi[0][1] = -M[0][1] * r;
            Mi[1][0] = -M[1][0] * r; Mi[1][1] = M[0][0] * r;
        } else {
            const float c00 = M[1][1] * M[2][2] - M[1][2] * M[2][1];
            const float c01 = M[1][2] * M[2][0] - M[1][0] * M[2][2];
            const float c02 = M[1][0] * M[2][1] - M[1][1] * M[2][0];
            det = M[0][0] * c00 + M[0][1] * c01 + M[0][2] * c02;
            const float r = 1.f / det;
            Mi[0][0] = c00 * r;
            Mi[0][1] = (M[0][2] * M[2][1] - M[0][1] * M[2][2]) * r;
            Mi[0][2] = (M[0][1] * M[1][2] - M[0][2] * M[1][1]) * r;
            Mi[1][0] = c01 * r;
            Mi[1][1] = (M[0][0] * M[2][2] - M[0][2] * M[2][0]) * r;
            Mi[1][2] = (M[0][2] * M[1][0] - M[0][0] * M[1][2]) * r;
            Mi[2][0] = c02 * r;
            Mi[2][1] = (M[0][1] * M[2][0] - M[0][0] * M[2][1]) * r;
            Mi[2][2] = (M[0][0] * M[1][1] - M[0][1] * M[1][0]) * r;
        }
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) {
                out[a * D + b] = M[a][b];
                out[D * D + a * D + b] = Mi[a][b];
            }
        out[2 * D * D] = det;
    }
}

}  // namespace

int fg_metrics_launch(const float* coords, float* transforms, int dims, int nx, int ny, int nz, hipStream_t st) {
    const int n = nx * ny * nz;
    int blocks = (n + FG_BLOCK - 1) / FG_BLOCK;
    if (blocks > 2048) blocks = 2048;
    if (dims == 2)
        hipLaunchKernelGGL(k_coords_to_transforms<2>, dim3(blocks), dim3(FG_BLOCK), 0, st, coords, transforms, nx, ny, nz);
    else
        hipLaunchKernelGGL(k_coords_to_transforms<3>, dim3(blocks), dim3(FG_BLOCK), 0, st, coords, transforms, nx, ny, nz);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}
