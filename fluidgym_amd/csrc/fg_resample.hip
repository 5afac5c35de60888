// Observation resampling: one rectilinear block -> uniform grid.
//
// Replaces SampleTransformedGridLocalToGlobalMulti + _FillEmptyCells (extensions/resampling.cu:191-609) for the
// single-block rectilinear case.  The reference SCATTERS: every source cell atomically adds value*weight to the
// 2^d (3-D: 6, see below) output cells around its centre, then a second kernel divides by the accumulated weight.
// On a rectilinear block the world -> output-index map is axis aligned and monotone, so the source cells that touch
// output cell o along one axis are a contiguous range [lo, hi): this file GATHERS -- one thread per output cell
// walks the small source box, no atomics, a fixed summation order, and the weight sum is accumulated alongside.
// The empty-cell fill is static geometry too: the step at which a cell gets filled is computed once on the host
// (fill_step), so each fill pass is one kernel over the cells of that step.
//
// 3-D quirk kept on request (quirk3d): the compiled reference loops `idx < (DIMS << 1)` corners, i.e. 6 of 8 in 3-D;
// the corners "upper in y AND upper in z" are never written (resampling.cu:320).  Its pure-torch twin
// (data/resample.py:361-548) writes all 8.  quirk3d = 1 reproduces the compiled kernel, 0 the torch form.
#include <stdlib.h>

#include <vector>

#include "fg_internal.h"

struct fg_resampler_state {
    int dims, quirk3d, device;
    int n_src[3], n_out[3];
    int max_fill_step;         // largest entry of fill_step (0: nothing to fill)
    int32_t* d_base[3];        // [n_src[a]] floor of the continuous output index of source cell i
    float* d_frac[3];          // [n_src[a]] its fractional part
    int32_t* d_lo[3];          // [n_out[a]] first source cell touching output o
    int32_t* d_hi[3];          // [n_out[a]] one past the last
    int16_t* d_fill_step;      // [n_out cells] 0 = written by the splat, k > 0 = filled in pass k, -1 = stays empty
};

namespace {

constexpr int MAXC = 8;

struct AxisTab { const int32_t* base; const float* frac; const int32_t* lo; const int32_t* hi; };

__device__ __forceinline__ float axis_w(const AxisTab& t, int i, int o, bool& upper) {
    const int b = t.base[i];
    upper = (b + 1 == o);
    return upper ? t.frac[i] : ((b == o) ? 1.f - t.frac[i] : 0.f);
}

// one thread per (env, output cell); channels looped inside so the weights are computed once
template <int DIMS>
__global__ __launch_bounds__(256) void k_resample_gather(const float* __restrict__ src, float* __restrict__ dst, AxisTab tx,
                                                          AxisTab ty, AxisTab tz, int sx, int sy, int sz, int ox, int oy,
                                                          int oz, int channels, int quirk3d) {
    const int cells = ox * oy * oz;
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= cells) return;
    const int b = blockIdx.y;
    const int ix = cell % ox, iy = (cell / ox) % oy, iz = cell / (ox * oy);
    const size_t src_n = (size_t)sx * sy * sz;
    const float* sb = src + (size_t)b * channels * src_n;
    float acc[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[c] = 0.f;
    float wsum = 0.f;
    const int k0 = (DIMS == 3) ? tz.lo[iz] : 0, k1 = (DIMS == 3) ? tz.hi[iz] : 1;
    for (int k = k0; k < k1; ++k) {
        bool upz = false;
        const float wz = (DIMS == 3) ? axis_w(tz, k, iz, upz) : 1.f;
        for (int j = ty.lo[iy]; j < ty.hi[iy]; ++j) {
            bool upy;
            const float wy = axis_w(ty, j, iy, upy);
            if (DIMS == 3 && quirk3d && upy && upz) continue;  // corners 110 / 111 are never written by the reference
            const float wyz = wy * wz;
            for (int i = tx.lo[ix]; i < tx.hi[ix]; ++i) {
                bool upx;
                const float w = axis_w(tx, i, ix, upx) * wyz;
                wsum += w;
                const size_t o = ((size_t)k * sy + j) * sx + i;
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < channels) acc[c] += w * sb[(size_t)c * src_n + o];
            }
        }
    }
    // k_NormScatteredWithWeight (resampling.cu:346-365): divide where the weight exceeds getEps<float>() = 1e-8
    const float inv = (wsum > 1e-8f) ? 1.f / wsum : 1.f;
    float* db = dst + (size_t)b * channels * cells;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < channels) db[(size_t)c * cells + cell] = acc[c] * inv;
}

// k_FillFromNeighbors (resampling.cu:191-240), pass `step`: cells whose fill_step == step take the mean of the face
// neighbours that were valid before this pass (fill_step in [0, step)).  In place: those neighbours are not written
// in this pass.
template <int DIMS>
__global__ __launch_bounds__(256) void k_resample_fill(float* __restrict__ dst, const int16_t* __restrict__ fill_step,
                                                        int step, int ox, int oy, int oz, int channels) {
    const int cells = ox * oy * oz;
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= cells || fill_step[cell] != step) return;
    const int b = blockIdx.y;
    const int p[3] = {cell % ox, (cell / ox) % oy, cell / (ox * oy)};
    const int n[3] = {ox, oy, oz};
    const int stride[3] = {1, ox, ox * oy};
    int nb[2 * DIMS], cnt = 0;
#pragma unroll
    for (int f = 0; f < 2 * DIMS; ++f) {
        const int a = f >> 1, q = p[a] + ((f & 1) * 2 - 1);
        nb[f] = -1;
        if (q >= 0 && q < n[a]) {
            const int c2 = cell + ((f & 1) * 2 - 1) * stride[a];
            const int fs = fill_step[c2];
            if (fs >= 0 && fs < step) { nb[f] = c2; ++cnt; }
        }
    }
    float* db = dst + (size_t)b * channels * cells;
    for (int c = 0; c < channels; ++c) {
        float v = 0.f;
#pragma unroll
        for (int f = 0; f < 2 * DIMS; ++f)
            if (nb[f] >= 0) v += db[(size_t)c * cells + nb[f]];
        db[(size_t)c * cells + cell] = v / (float)cnt;
    }
}

template <typename T>
hipError_t upload(T** dst, const T* src, size_t count) {
    hipError_t e = hipMalloc(dst, sizeof(T) * (count ? count : 1));
    if (e == hipSuccess && count) e = hipMemcpy(*dst, src, sizeof(T) * count, hipMemcpyHostToDevice);
    return e;
}

}  // namespace

extern "C" int fg_resampler_create(int dims, const int32_t* n_src, const int32_t* n_out, const int32_t* base_cat,
                                   const float* frac_cat, int quirk3d, int device, fg_resampler* out) {
    FG_REQUIRE(n_src && n_out && base_cat && frac_cat && out, FG_ERR_INVALID_ARG, "null argument");
    FG_REQUIRE(dims == 2 || dims == 3, FG_ERR_INVALID_ARG, "dims must be 2 or 3");
    FG_HIP_CHECK(hipSetDevice(device));
    fg_resampler_state* r = new fg_resampler_state();
    r->dims = dims; r->quirk3d = (dims == 3) ? (quirk3d != 0) : 0; r->device = device;
    for (int a = 0; a < 3; ++a) { r->n_src[a] = a < dims ? n_src[a] : 1; r->n_out[a] = a < dims ? n_out[a] : 1; }
    // per-axis gather ranges: base is non-decreasing (cell centres increase), output o is touched by the source cells
    // with base in {o - 1, o}
    std::vector<std::vector<float>> wax(3);  // per-axis weight sums split by corner side, for the static fill geometry
    std::vector<std::vector<float>> wax_up(3);
    size_t off = 0;
    for (int a = 0; a < dims; ++a) {
        const int ns = r->n_src[a], no = r->n_out[a];
        const int32_t* base = base_cat + off;
        const float* frac = frac_cat + off;
        for (int i = 1; i < ns; ++i)
            FG_REQUIRE(base[i] >= base[i - 1], FG_ERR_INVALID_ARG, "axis index map must be non-decreasing");
        std::vector<int32_t> lo(no), hi(no);
        wax[a].assign(no, 0.f); wax_up[a].assign(no, 0.f);
        for (int o = 0; o < no; ++o) {
            int l = 0;
            while (l < ns && base[l] < o - 1) ++l;
            int h = l;
            while (h < ns && base[h] <= o) ++h;
            lo[o] = l; hi[o] = h;
            for (int i = l; i < h; ++i) {
                if (base[i] == o) wax[a][o] += 1.f - frac[i];
                else if (base[i] + 1 == o) { wax[a][o] += frac[i]; wax_up[a][o] += frac[i]; }
            }
        }
        FG_HIP_CHECK(upload(&r->d_base[a], base, ns));
        FG_HIP_CHECK(upload(&r->d_frac[a], frac, ns));
        FG_HIP_CHECK(upload(&r->d_lo[a], lo.data(), no));
        FG_HIP_CHECK(upload(&r->d_hi[a], hi.data(), no));
        off += ns;
    }
    // static fill schedule (resampling.cu:242-290): written = weight > eps, then breadth-first passes over face neighbours
    const int ox = r->n_out[0], oy = r->n_out[1], oz = r->n_out[2];
    const size_t cells = (size_t)ox * oy * oz;
    std::vector<int16_t> fs(cells, -1);
    for (int k = 0; k < oz; ++k)
        for (int j = 0; j < oy; ++j)
            for (int i = 0; i < ox; ++i) {
                float w = wax[0][i] * wax[1][j];
                if (dims == 3) {
                    w *= wax[2][k];
                    if (r->quirk3d) w = wax[0][i] * (wax[1][j] * wax[2][k] - wax_up[1][j] * wax_up[2][k]);
                }
                if (w > 1e-8f) fs[((size_t)k * oy + j) * ox + i] = 0;
            }
    int step = 0;
    const int n[3] = {ox, oy, oz};
    const long stride[3] = {1, ox, (long)ox * oy};
    for (;;) {
        std::vector<size_t> newly;
        for (size_t c = 0; c < cells; ++c) {
            if (fs[c] >= 0) continue;
            const int p[3] = {(int)(c % ox), (int)((c / ox) % oy), (int)(c / ((size_t)ox * oy))};
            bool any = false;
            for (int f = 0; f < 2 * dims && !any; ++f) {
                const int a = f >> 1, q = p[a] + ((f & 1) * 2 - 1);
                if (q >= 0 && q < n[a]) any = fs[c + ((f & 1) * 2 - 1) * stride[a]] >= 0 && fs[c + ((f & 1) * 2 - 1) * stride[a]] <= step;
            }
            if (any) newly.push_back(c);
        }
        if (newly.empty() || step >= 32000) break;
        ++step;
        for (size_t c : newly) fs[c] = (int16_t)step;
    }
    r->max_fill_step = step;
    FG_HIP_CHECK(upload(&r->d_fill_step, fs.data(), cells));
    *out = r;
    return FG_OK;
}

extern "C" int fg_resampler_destroy(fg_resampler r) {
    if (!r) return FG_OK;
    for (int a = 0; a < 3; ++a) { (void)hipFree(r->d_base[a]); (void)hipFree(r->d_frac[a]); (void)hipFree(r->d_lo[a]); (void)hipFree(r->d_hi[a]); }
    (void)hipFree(r->d_fill_step);
    delete r;
    return FG_OK;
}

extern "C" int fg_resample(fg_resampler r, const float* src, int batch, int channels, float* dst, int fill_max_steps,
                           void* stream) {
    FG_REQUIRE(r && src && dst, FG_ERR_INVALID_ARG, "null argument");
    FG_REQUIRE(batch > 0 && channels > 0 && channels <= MAXC, FG_ERR_INVALID_ARG, "channels must be 1..8");
    hipStream_t st = (hipStream_t)stream;
    const int ox = r->n_out[0], oy = r->n_out[1], oz = r->n_out[2];
    const int cells = ox * oy * oz;
    AxisTab t[3];
    for (int a = 0; a < 3; ++a) t[a] = AxisTab{r->d_base[a], r->d_frac[a], r->d_lo[a], r->d_hi[a]};
    const dim3 grid((cells + 255) / 256, batch);
    if (r->dims == 2)
        hipLaunchKernelGGL(k_resample_gather<2>, grid, dim3(256), 0, st, src, dst, t[0], t[1], t[2], r->n_src[0], r->n_src[1], 1,
                           ox, oy, 1, channels, 0);
    else
        hipLaunchKernelGGL(k_resample_gather<3>, grid, dim3(256), 0, st, src, dst, t[0], t[1], t[2], r->n_src[0], r->n_src[1],
                           r->n_src[2], ox, oy, oz, channels, r->quirk3d);
    const int steps = fill_max_steps < r->max_fill_step ? fill_max_steps : r->max_fill_step;
    for (int s = 1; s <= steps; ++s) {
        if (r->dims == 2) hipLaunchKernelGGL(k_resample_fill<2>, grid, dim3(256), 0, st, dst, r->d_fill_step, s, ox, oy, 1, channels);
        else hipLaunchKernelGGL(k_resample_fill<3>, grid, dim3(256), 0, st, dst, r->d_fill_step, s, ox, oy, oz, channels);
    }
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-block resampling (curvilinear blocks -> uniform render grid / sensor pixels).  For a fixed mesh the reference's chain
// splat + normalise + FillFromNeighbors (resampling.cu:191-609; sample_multi_coords_to_uniform_grid_diff) is one linear
// operator; simulation/resample_mb.py folds it on the host once.  Applying it is a sparse gather: the two kernels below are
// the whole device side (round 1 went through torch.sparse.mm / fancy indexing).
//   k_sparse_ell: rows of equal length K (the sensors' rows: 151 x <= 9 cells in 2-D) -- one thread per (row, field)
//   k_sparse_csr: the full operator -- eight lanes per row stride over its entries (coalesced index / weight reads for
//                 the long rows that hole filling produces), then reduce with three shuffles
// field x: [M][N] (M = envs x components, row stride N); out y: [M][rows].
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_sparse_ell(const int32_t* __restrict__ idx, const float* __restrict__ w, int rows, int K,
                                                     const float* __restrict__ x, size_t n, float* __restrict__ y) {
    const int r = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (r >= rows) return;
    const float* xm = x + (size_t)m * n;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += w[(size_t)r * K + k] * xm[idx[(size_t)r * K + k]];
    y[(size_t)m * rows + r] = acc;
}
__global__ __launch_bounds__(256) void k_sparse_csr(const int32_t* __restrict__ indptr, const int32_t* __restrict__ col,
                                                     const float* __restrict__ val, int rows, const float* __restrict__ x, size_t n,
                                                     float* __restrict__ y) {
    const int r = (blockIdx.x * 256 + threadIdx.x) >> 3, sub = threadIdx.x & 7, m = blockIdx.y;
    float acc = 0.f;
    if (r < rows) {
        const float* xm = x + (size_t)m * n;
        for (int k = indptr[r] + sub; k < indptr[r + 1]; k += 8) acc += val[k] * xm[col[k]];
    }
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 1, 64);
    if (r < rows && sub == 0) y[(size_t)m * rows + r] = acc;
}
}  // namespace

extern "C" int fg_sparse_apply_ell(const int32_t* idx, const float* w, int32_t rows, int32_t K, const float* x, int64_t n, int32_t m,
                                   float* y, void* stream) {
    FG_REQUIRE(idx && w && x && y && rows > 0 && K > 0 && n > 0 && m > 0, FG_ERR_INVALID_ARG, "fg_sparse_apply_ell: bad argument");
    hipLaunchKernelGGL(k_sparse_ell, dim3((rows + 255) / 256, m), dim3(256), 0, (hipStream_t)stream, idx, w, rows, K, x, (size_t)n, y);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}
extern "C" int fg_sparse_apply_csr(const int32_t* indptr, const int32_t* col, const float* val, int32_t rows, const float* x, int64_t n,
                                   int32_t m, float* y, void* stream) {
    FG_REQUIRE(indptr && col && val && x && y && rows > 0 && n > 0 && m > 0, FG_ERR_INVALID_ARG, "fg_sparse_apply_csr: bad argument");
    hipLaunchKernelGGL(k_sparse_csr, dim3((rows * 8 + 255) / 256, m), dim3(256), 0, (hipStream_t)stream, indptr, col, val, rows, x,
                       (size_t)n, y);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}
