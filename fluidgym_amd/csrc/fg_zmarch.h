// Shared pieces of the z-marching 3-D kernels (fg_poisson3d.hip: pressure operator / sweeps / CG kernel; fg_bicgstab3d.hip: the
// two-kernel BiCGStab on the advection-diffusion matrix): buffer-resource addressing, tile geometry, the per-thread context with
// its halo duty, and the LDS tile fill.  gfx950 / wave64 only, fp32 only.
#pragma once
#include "fg_internal.h"

namespace {

// Buffer-resource addressing (cdna_hip_programming.md T8): one 128-bit SGPR descriptor per field, a per-thread
// 32-bit byte offset that never changes (vo_*) and the plane offset as the scalar soffset.  The flat-pointer
// form needed a 64-bit VGPR pair + v_lshl_add_u64 per distinct address (48 of them in the ISA) and pushed the
// kernel to 160+ VGPRs.
#ifndef Z_STORE_AUX
#define Z_STORE_AUX 0  // 2 = nt stores: round 2 measured +8 % on the one-shot apply and -12 % on the Jacobi ping-pong at 256^3; round 3 (nt for
                       // the apply mode only): 0.550 against 0.554-0.578 with plain stores, i.e. nothing -- left off
#endif
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t z_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ FgVec<4> z_bload4(rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    FgVec<4> o;
    o.v[0] = __uint_as_float(v.x); o.v[1] = __uint_as_float(v.y); o.v[2] = __uint_as_float(v.z); o.v[3] = __uint_as_float(v.w);
    return o;
}
__device__ __forceinline__ float z_bload1(rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// 128-bit buffer stores go out as inline assembly with wait states glued behind them.  Found in round 4 (fg_bicgstab3d.hip, kernel b):
// the compiler issued `buffer_store_dwordx4 v[86:89], ..., s12 offen` and, as the very next instruction, a packed-fp32 VALU write
// of v[88:89] -- and the stored element 3 (resp. element 1 after a write of v[66:67]) came out wrong in lanes 12-15 of every
// 16-lane row: the store reads its data registers over several cycles after issue (longer behind another store), and LLVM only
// pads this hazard for stores WITHOUT an SGPR offset (GCNHazardRecognizer: "only exists if the instruction is not using a
// register in the soffset field").  The s_nop belongs to the same asm statement, so nothing can be scheduled in between.
template <int AUX = Z_STORE_AUX>
__device__ __forceinline__ void z_bstore4(rsrc_t r, unsigned voff, unsigned soff, const FgVec<4>& v) {
    u32x4_t u;
    u.x = __float_as_uint(v.v[0]); u.y = __float_as_uint(v.v[1]); u.z = __float_as_uint(v.v[2]); u.w = __float_as_uint(v.v[3]);
    const unsigned so = __builtin_amdgcn_readfirstlane(soff);    // (wave-uniform by construction: plane / component offsets)
    if constexpr (AUX == 0)
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 7" : : "v"(u), "v"(voff), "s"(r), "s"(so) : "memory");
    else
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 7" : : "v"(u), "v"(voff), "s"(r), "s"(so) : "memory");
}

// Tile shape: BXL lanes (x float4) along x, 256 / BXL rows along y.  64 x 16, 128 x 8 or 256 x 4 cells: wider
// tiles make each wave-level access a longer contiguous run (256 B / 512 B / 1 KiB per row) at the price of
// more y-halo rows per cell.
template <int BXL>
struct ZT {
    static constexpr int TX = BXL * 4, TY = FG_BLOCK / BXL;
    static constexpr int LP = TX + 8;     // LDS row pitch in floats: [3] left halo, [4 .. TX+3] cells (16-B aligned), [TX+4] right halo
    static constexpr int LROWS = TY + 2;
};

struct FgCoefZ {
    float xm[4], xp[4], ym[4], yp[4], zm[4], zp[4];
};

struct ZCtx {
    int b, i0, j, k0, k1;                 // env, first cell of the vector, row, z range [k0, k1)
    int lx, ly;
    bool valid;                           // (i0, j) inside the grid
    float mxm, mxp, mym, myp;             // x/y face masks of this thread's cells
    int row_c, row_ym, row_yp;            // in-plane offsets (j * nx + i0) of the centre / y-neighbour rows
    int col_xm, col_xp;                   // in-plane offsets of the x-halo cells (row j)
};

template <int BXL>
__device__ __forceinline__ ZCtx z_make_ctx(const FgGrid& g, int tiles_x, int tiles_y, int zchunks, int ZC) {
    constexpr int TX = ZT<BXL>::TX, TY = ZT<BXL>::TY;
    ZCtx c;
    const unsigned per_env = tiles_x * tiles_y * zchunks;
    const unsigned bid = fg_xcd_remap(blockIdx.x, gridDim.x);
    c.b = bid / per_env;
    unsigned t = bid - c.b * per_env;
    const int tix = t % tiles_x; t /= tiles_x;
    const int tiy = t % tiles_y;
    const int tz = t / tiles_y;
    c.lx = threadIdx.x % BXL; c.ly = threadIdx.x / BXL;
    c.i0 = tix * TX + c.lx * 4;
    c.j = tiy * TY + c.ly;
    c.k0 = tz * ZC;
    c.k1 = min(c.k0 + ZC, g.nz);
    c.valid = (c.i0 < g.nx) && (c.j < g.ny);
    const int i0 = c.valid ? c.i0 : 0, j = c.valid ? c.j : 0;
    const bool at_xm = (i0 == 0), at_xp = (i0 + 4 == g.nx), at_ym = (j == 0), at_yp = (j == g.ny - 1);
    c.mxm = (at_xm && g.fixed[0]) ? 0.f : 1.f;
    c.mxp = (at_xp && g.fixed[1]) ? 0.f : 1.f;
    c.mym = (at_ym && g.fixed[2]) ? 0.f : 1.f;
    c.myp = (at_yp && g.fixed[3]) ? 0.f : 1.f;
    c.row_c = j * g.nx + i0;
    c.row_ym = (at_ym ? (g.fixed[2] ? j : g.ny - 1) : j - 1) * g.nx + i0;
    c.row_yp = (at_yp ? (g.fixed[3] ? j : 0) : j + 1) * g.nx + i0;
    c.col_xm = j * g.nx + (at_xm ? (g.fixed[0] ? i0 : g.nx - 1) : i0 - 1);
    c.col_xp = j * g.nx + (at_xp ? (g.fixed[1] ? i0 + 3 : 0) : i0 + 4);
    return c;
}

__device__ __forceinline__ int z_plane(const FgGrid& g, int k) {  // plane index with periodic wrap / clamp
    if (k < 0) return g.fixed[4] ? 0 : g.nz - 1;
    if (k >= g.nz) return g.fixed[5] ? g.nz - 1 : 0;
    return k;
}

// Halo duty of a thread: ly == 0 / TY-1 fetch the y-halo row segment above / below the tile, lx == 0 / 15 the
// x-halo cell left / right of their row.  Halos are PREFETCHED one plane ahead into registers (Halo), so the
// LDS fill of plane k never waits on memory.
struct Halo {
    FgVec<4> y;   // valid when ly == 0 or ly == TY-1
    float x;      // valid when lx == 0 or lx == 15
};

template <int BXL>
__device__ __forceinline__ Halo z_load_halo(const ZCtx& c, rsrc_t r, unsigned vo_hy, unsigned vo_hx, unsigned soff) {
    constexpr int TY = ZT<BXL>::TY;
    Halo h;
    h.x = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) h.y.v[e] = 0.f;
    if (c.ly == 0 || c.ly == TY - 1) h.y = z_bload4(r, vo_hy, soff);
    if (c.lx == 0 || c.lx == BXL - 1) h.x = z_bload1(r, vo_hx, soff);
    return h;
}
template <int BXL>
__device__ __forceinline__ void z_fill_tile(float* __restrict__ tile, const ZCtx& c, const FgVec<4>& ctr, const Halo& h) {
    constexpr int TY = ZT<BXL>::TY, LP = ZT<BXL>::LP;
    float* row = tile + (c.ly + 1) * LP + 4 + c.lx * 4;
    *reinterpret_cast<float4*>(row) = make_float4(ctr.v[0], ctr.v[1], ctr.v[2], ctr.v[3]);
    if (c.ly == 0) *reinterpret_cast<float4*>(row - LP) = make_float4(h.y.v[0], h.y.v[1], h.y.v[2], h.y.v[3]);
    if (c.ly == TY - 1) *reinterpret_cast<float4*>(row + LP) = make_float4(h.y.v[0], h.y.v[1], h.y.v[2], h.y.v[3]);
    if (c.lx == 0) row[-1] = h.x;
    if (c.lx == BXL - 1) row[4] = h.x;
}

}  // namespace
