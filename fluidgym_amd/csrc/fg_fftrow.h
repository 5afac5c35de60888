// Wave-level building blocks of the row transforms (fg_fdfft.hip has the stand-alone kernel k_dct_rows; this header is what the
// FUSED row kernels of fg_fftcg.hip are assembled from): one wave64 transforms a PAIR of real rows of length N as one complex FFT
// in its own two LDS buffers; between stages only the lanes of that wave exchange data, so a wave-level barrier is enough.
//   PERIODIC = false: orthonormal DCT-II / DCT-III of a uniform FIXED axis through Makhoul's mapping onto an N-point complex FFT
//   PERIODIC = true : real Fourier basis of a uniform periodic axis in FFT order (mode m <= N/2 the cosine, m > N/2 the sine of
//                     wavenumber min(m, N - m))
// -- the same bases, scales and arithmetic as k_dct_rows (see the derivation there and in simulation/fd_precond.py).
// Element <-> lane mapping: a lane owns EPL = N / 64 elements of each row, in groups of VW = min(4, EPL) CONSECUTIVE elements,
// group q at column q * 64 * VW + lane * VW: every wave-wide access of a group is one coalesced 64 * VW * 4-byte row segment
// (k_dct_rows gives a lane EPL consecutive elements: two half-coalesced accesses per row at N = 512).
#pragma once
#include "fg_internal.h"

#if !FG_F64
namespace fgfft {

template <int N> struct Map {
    static constexpr int EPL = N / 64;
    static constexpr int VW = EPL < 4 ? EPL : 4;
    static constexpr int NG = EPL / VW;          // groups per lane and row
    __device__ static __forceinline__ int col(int lane, int e) { return (e / VW) * (64 * VW) + lane * VW + (e % VW); }
};

// VW consecutive floats (VW = 1, 2, 4): p must be aligned to VW * 4 bytes
template <int VW> __device__ __forceinline__ void ldv(const float* __restrict__ p, float* out) {
    if constexpr (VW == 4) { const float4 v = *reinterpret_cast<const float4*>(p); out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w; }
    else if constexpr (VW == 2) { const float2 v = *reinterpret_cast<const float2*>(p); out[0] = v.x; out[1] = v.y; }
    else out[0] = *p;
}
template <int VW> __device__ __forceinline__ void stv(float* __restrict__ p, const float* in) {
    if constexpr (VW == 4) *reinterpret_cast<float4*>(p) = make_float4(in[0], in[1], in[2], in[3]);
    else if constexpr (VW == 2) *reinterpret_cast<float2*>(p) = make_float2(in[0], in[1]);
    else *p = in[0];
}
// a whole row (EPL elements of this lane) from / to global memory; row = pointer to column 0
template <int N> __device__ __forceinline__ void load_row(const float* __restrict__ row, int lane, float (&v)[N / 64]) {
    using M = Map<N>;
#pragma unroll
    for (int q = 0; q < M::NG; ++q) ldv<M::VW>(row + q * 64 * M::VW + lane * M::VW, &v[q * M::VW]);
}
template <int N> __device__ __forceinline__ void store_row(float* __restrict__ row, int lane, const float (&v)[N / 64]) {
    using M = Map<N>;
#pragma unroll
    for (int q = 0; q < M::NG; ++q) stv<M::VW>(row + q * 64 * M::VW + lane * M::VW, &v[q * M::VW]);
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Stockham autosort FFT of the wave's buffer x (N complex), radix 4 while the remaining length allows it, then one radix-2 stage;
// ping-pongs between x and y and returns the buffer that holds the result.  twl: W^k = (cos, sin)(2 pi k / N) in LDS.
template <int N, bool INVERSE>
__device__ __forceinline__ float2* stockham(float2* x, float2* y, const float2* __restrict__ twl, int lane) {
    auto cmul = [](float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); };
    auto twid = [&](int k) { float2 w = twl[k & (N - 1)]; if (!INVERSE) w.y = -w.y; return w; };
    int sft = 0, rem = N;
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        if (rem % 4 != 0) break;
        const int sm = (1 << sft) - 1;
        for (int t = lane; t < N / 4; t += 64) {
            const int q = t & sm, ps = t - q;
            const float2 a0 = x[t], a1 = x[t + N / 4], a2 = x[t + N / 2], a3 = x[t + 3 * N / 4];
            const float2 r1 = INVERSE ? make_float2(-a1.y, a1.x) : make_float2(a1.y, -a1.x);
            const float2 r3 = INVERSE ? make_float2(-a3.y, a3.x) : make_float2(a3.y, -a3.x);
            const float2 b0 = make_float2(a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y);
            const float2 c1 = make_float2(a0.x + r1.x - a2.x - r3.x, a0.y + r1.y - a2.y - r3.y);
            const float2 c2 = make_float2(a0.x - a1.x + a2.x - a3.x, a0.y - a1.y + a2.y - a3.y);
            const float2 c3 = make_float2(a0.x - r1.x - a2.x + r3.x, a0.y - r1.y - a2.y + r3.y);
            float2* o = y + 4 * ps + q;
            o[0] = b0;
            o[1 << sft] = cmul(c1, twid(ps));
            o[2 << sft] = cmul(c2, twid(2 * ps));
            o[3 << sft] = cmul(c3, twid(3 * ps));
        }
        wave_sync();
        float2* tmp = x; x = y; y = tmp;
        sft += 2; rem /= 4;
    }
    if (rem == 2) {
        const int sm = (1 << sft) - 1;
        for (int t = lane; t < N / 2; t += 64) {
            const int q = t & sm, ps = t - q;
            const float2 u = x[t], v = x[t + N / 2];
            y[2 * ps + q] = make_float2(u.x + v.x, u.y + v.y);
            y[2 * ps + q + (1 << sft)] = cmul(make_float2(u.x - v.x, u.y - v.y), twid(ps));
        }
        wave_sync();
        float2* tmp = x; x = y; y = tmp;
    }
    return x;
}

struct Scales { float s0, s; };   // k = 0 (and N/2 on a periodic axis) | the other modes

// FORWARD transform of the two real rows (xa, xb) held in registers into their mode coefficients (oa, ob), same element mapping.
// x, y: the wave's two LDS buffers (N float2 each); rot: (cos, sin)(pi k / 2N) in global memory (cosine basis only).
template <int N, bool PERIODIC>
__device__ __forceinline__ void forward_rows(const float (&xa)[N / 64], const float (&xb)[N / 64], float (&oa)[N / 64], float (&ob)[N / 64],
                                             float2* x, float2* y, const float2* __restrict__ twl, const float2* __restrict__ rot,
                                             Scales sc, int lane) {
    using M = Map<N>;
#pragma unroll
    for (int e = 0; e < M::EPL; ++e) {
        const int i = M::col(lane, e);
        const int j = PERIODIC ? i : ((i & 1) ? N - 1 - (i >> 1) : (i >> 1));     // v_j = x_2j, v_{n-1-j} = x_{2j+1}
        x[j] = make_float2(xa[e], xb[e]);
    }
    wave_sync();
    const float2* X = stockham<N, false>(x, y, twl, lane);
    // split the spectrum of z = a + i b:  V^a = (Z_k + conj Z_{n-k}) / 2,  V^b = (Z_k - conj Z_{n-k}) / 2i
#pragma unroll
    for (int e = 0; e < M::EPL; ++e) {
        const int m = M::col(lane, e);
        const int k = PERIODIC ? (m <= N / 2 ? m : N - m) : m;
        const float2 Z = X[k], Mz = X[(N - k) & (N - 1)];
        const float2 Va = make_float2(0.5f * (Z.x + Mz.x), 0.5f * (Z.y - Mz.y));
        const float2 Vb = make_float2(0.5f * (Z.y + Mz.y), -0.5f * (Z.x - Mz.x));
        if (PERIODIC) {   // X_k = s_k Re V_k, X_{n-k} = -s_k Im V_k
            oa[e] = (m == 0 || m == N / 2) ? sc.s0 * Va.x : (m < N / 2 ? sc.s * Va.x : -sc.s * Va.y);
            ob[e] = (m == 0 || m == N / 2) ? sc.s0 * Vb.x : (m < N / 2 ? sc.s * Vb.x : -sc.s * Vb.y);
        } else {          // X_k = s_k Re(e^{-i theta_k} V_k)
            const float2 r = rot[k];
            const float g = (k == 0) ? sc.s0 : sc.s;
            oa[e] = g * (r.x * Va.x + r.y * Va.y);
            ob[e] = g * (r.x * Vb.x + r.y * Vb.y);
        }
    }
    wave_sync();      // the buffers may be reused by the caller
}

// INVERSE transform of the two rows of mode coefficients (ya, yb) in registers; the result rows are left in LDS in NATURAL order as
// floats: row a in out[0 .. N), row b in out[N .. 2N), where out is the returned pointer (one of the wave's two buffers; the OTHER
// buffer is free afterwards and is returned through `spare`).
template <int N, bool PERIODIC>
__device__ __forceinline__ float* inverse_rows(const float (&ya)[N / 64], const float (&yb)[N / 64], float2* x, float2* y,
                                               const float2* __restrict__ twl, const float2* __restrict__ rot, Scales sc, int lane,
                                               float2** spare) {
    using M = Map<N>;
    // stage both coefficient rows so that Y_{n-k} is reachable, then Z_k = W^a_k + i W^b_k with
    //   cosine basis:  W_k = g_k (Y_k - i Y_{n-k}) e^{+i theta_k},  Y_n := 0
    //   Fourier basis: W_0 = g_0 Y_0, W_{n/2} = g_0 Y_{n/2}, W_k = g/2 (Y_k - i Y_{n-k}), W_{n-k} = conj W_k
    float* sa = reinterpret_cast<float*>(y);
    float* sb = sa + N;
#pragma unroll
    for (int q = 0; q < M::NG; ++q) {
        stv<M::VW>(sa + q * 64 * M::VW + lane * M::VW, &ya[q * M::VW]);
        stv<M::VW>(sb + q * 64 * M::VW + lane * M::VW, &yb[q * M::VW]);
    }
    wave_sync();
#pragma unroll
    for (int e = 0; e < M::EPL; ++e) {
        const int k = M::col(lane, e);
        float2 wa, wb;
        if (PERIODIC) {
            const int km = (N - k) & (N - 1);
            const float a0 = sa[k], am = sa[km], b0 = sb[k], bm = sb[km];
            if (k == 0 || k == N / 2) { wa = make_float2(sc.s0 * a0, 0.f); wb = make_float2(sc.s0 * b0, 0.f); }
            else if (k < N / 2) { wa = make_float2(0.5f * sc.s * a0, -0.5f * sc.s * am); wb = make_float2(0.5f * sc.s * b0, -0.5f * sc.s * bm); }
            else { wa = make_float2(0.5f * sc.s * am, 0.5f * sc.s * a0); wb = make_float2(0.5f * sc.s * bm, 0.5f * sc.s * b0); }
        } else {
            const float a0 = sa[k], am = (k == 0) ? 0.f : sa[N - k], b0 = sb[k], bm = (k == 0) ? 0.f : sb[N - k];
            const float2 r = rot[k];
            const float g = (k == 0) ? sc.s0 : sc.s;
            wa = make_float2(g * (a0 * r.x + am * r.y), g * (a0 * r.y - am * r.x));
            wb = make_float2(g * (b0 * r.x + bm * r.y), g * (b0 * r.y - bm * r.x));
        }
        x[k] = make_float2(wa.x - wb.y, wa.y + wb.x);
    }
    wave_sync();
    float2* X = stockham<N, true>(x, y, twl, lane);
    float2* other = (X == x) ? y : x;
    // natural order, rows apart: v_j = x_2j, v_{n-1-j} = x_{2j+1} undone for the cosine basis
    float* out = reinterpret_cast<float*>(other);
    float va[M::EPL], vb[M::EPL];
#pragma unroll
    for (int e = 0; e < M::EPL; ++e) {
        const int i = M::col(lane, e);
        const float2 v = X[PERIODIC ? i : ((i & 1) ? N - 1 - (i >> 1) : (i >> 1))];
        va[e] = v.x; vb[e] = v.y;
    }
    wave_sync();      // (`other` is the buffer the last stage read from: every lane is through with it)
#pragma unroll
    for (int q = 0; q < M::NG; ++q) {
        stv<M::VW>(out + q * 64 * M::VW + lane * M::VW, &va[q * M::VW]);
        stv<M::VW>(out + N + q * 64 * M::VW + lane * M::VW, &vb[q * M::VW]);
    }
    *spare = X;
    return out;
}

}  // namespace fgfft
#endif
