// Live per-kernel timing for bench.py's roofline (see FgProf in fg_internal.h).
//
// A sampled launch is issued with hipExtLaunchKernelGGL(start, stop): the two events carry the begin/end
// timestamps of the kernel's own dispatch, so the figure is the kernel's duration as rocprofv3 --kernel-trace
// reports it (an event bracket around a ~10 us kernel would add 1.5-2 us of dispatch gap).  Before it, a
// one-wave kernel on the same stream counts the systems whose flag is still 0: kernels skip converged systems,
// so the ALGORITHMIC bytes of a launch are (active systems) x (bytes per system), not the full batch.
#include <hip/hip_ext.h>
#include <stdlib.h>

#include <algorithm>

#include "fg_internal.h"

namespace {
__global__ void k_prof_count(const int32_t* __restrict__ flags, int nsys, int32_t* __restrict__ out) {
    int c = 0;
    for (int i = threadIdx.x; i < nsys; i += 64) c += (flags[i] == 0);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if (threadIdx.x == 0) *out = c;
}
const char* const kNames[FG_PK_COUNT] = {"k_cg_ap", "k_cg_update", "k_bicg_p", "k_bicg_v", "k_bicg_s", "k_bicg_t",
                                         "k_bicg_x", "k_gemm_f32", "k_gemm_sk", "k_tridiag_y", "k_dct_rows"};
}  // namespace

int fg_prof_slot(const fg_state* cs, int kind, const int32_t* flags, int nsys, double bytes_per_sys,
                 double flops_per_sys, hipStream_t st) {
    if (!cs->prof.on) return -1;
    FgProf& P = const_cast<fg_state*>(cs)->prof;
    const long long k = P.launches[kind]++;
    if (P.used >= FG_PROF_POOL || (k % P.period) != 0) return -1;
    const int slot = P.used++;
    P.meta[slot] = FgProfMeta{kind, nsys, bytes_per_sys, flops_per_sys};
    if (flags == FG_PROF_SELF) (void)hipMemsetAsync(P.active_dev + slot, 0, sizeof(int32_t), st);  // kernel counts
    else if (flags) hipLaunchKernelGGL(k_prof_count, dim3(1), dim3(64), 0, st, flags, nsys, P.active_dev + slot);
    else (void)hipMemsetAsync(P.active_dev + slot, 0xff, sizeof(int32_t), st);  // -1 = "all active"
    return slot;
}

int fg_prof_collect(fg_state* s, hipStream_t st) {
    FgProf& P = s->prof;
    if (!P.used) return FG_OK;
    FG_HIP_CHECK(hipMemcpyAsync(P.active_pinned, P.active_dev, sizeof(int32_t) * P.used, hipMemcpyDeviceToHost, st));
    FG_HIP_CHECK(hipStreamSynchronize(st));
    for (int i = 0; i < P.used; ++i) {
        const FgProfMeta& m = P.meta[i];
        const int act = P.active_pinned[i] < 0 ? m.nsys : P.active_pinned[i];
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, P.ev[2 * i], P.ev[2 * i + 1]) != hipSuccess) continue;
        P.all_ms[m.kind] += ms; P.all_n[m.kind]++;
        if (act <= 0) continue;  // every system had converged: the launch did no work
        P.ms[m.kind] += ms; P.n[m.kind]++;
        P.bytes[m.kind] += act * m.bytes_per_sys; P.flops[m.kind] += act * m.flops_per_sys;
        if (act == m.nsys) { P.full_ms[m.kind] += ms; P.full_bytes[m.kind] += act * m.bytes_per_sys; P.full_n[m.kind]++; }
    }
    P.used = 0;
    return FG_OK;
}

extern "C" int fg_profile_enable(fg_handle s, int on) {
    FG_REQUIRE(s, FG_ERR_INVALID_ARG, "null handle");
    FgProf& P = s->prof;
    if (on && !P.active_dev) {
        for (int i = 0; i < 2 * FG_PROF_POOL; ++i) FG_HIP_CHECK(hipEventCreate(&P.ev[i]));
        FG_HIP_CHECK(hipMalloc(&P.active_dev, sizeof(int32_t) * FG_PROF_POOL));
        FG_HIP_CHECK(hipHostMalloc(&P.active_pinned, sizeof(int32_t) * FG_PROF_POOL));
    }
    FG_HIP_CHECK(hipDeviceSynchronize());
    P.on = on; P.used = 0;
    const char* e = getenv("FG_PROF_PERIOD");
    P.period = e && atoi(e) > 0 ? atoi(e) : 8;
    for (int k = 0; k < FG_PK_COUNT; ++k) {
        P.ms[k] = P.bytes[k] = P.flops[k] = P.full_ms[k] = P.full_bytes[k] = 0.0;
        P.n[k] = P.full_n[k] = P.launches[k] = P.all_n[k] = 0;
        P.all_ms[k] = 0.0;
    }
    return FG_OK;
}

extern "C" int fg_profile_kinds(void) { return FG_PK_COUNT; }
extern "C" const char* fg_profile_kind_name(int kind) { return kind >= 0 && kind < FG_PK_COUNT ? kNames[kind] : ""; }

extern "C" int fg_profile_read(fg_handle s, int kind, double* ms_sum, int64_t* samples, double* bytes_sum,
                               double* flops_sum, double* full_ms_sum, double* full_bytes_sum, int64_t* full_samples,
                               int64_t* launches, double* all_ms_sum, int64_t* all_samples) {
    FG_REQUIRE(s && kind >= 0 && kind < FG_PK_COUNT, FG_ERR_INVALID_ARG, "bad handle or kernel kind");
    FG_HIP_CHECK(hipDeviceSynchronize());
    if (int rc = fg_prof_collect(s, nullptr)) return rc;
    const FgProf& P = s->prof;
    if (ms_sum) *ms_sum = P.ms[kind];
    if (samples) *samples = P.n[kind];
    if (bytes_sum) *bytes_sum = P.bytes[kind];
    if (flops_sum) *flops_sum = P.flops[kind];
    if (full_ms_sum) *full_ms_sum = P.full_ms[kind];
    if (full_bytes_sum) *full_bytes_sum = P.full_bytes[kind];
    if (full_samples) *full_samples = P.full_n[kind];
    if (launches) *launches = P.launches[kind];
    if (all_ms_sum) *all_ms_sum = P.all_ms[kind];
    if (all_samples) *all_samples = P.all_n[kind];
    return FG_OK;
}

void fg_prof_destroy(fg_state* s) {
    FgProf& P = s->prof;
    if (!P.active_dev) return;
    for (int i = 0; i < 2 * FG_PROF_POOL; ++i) (void)hipEventDestroy(P.ev[i]);
    (void)hipFree(P.active_dev); (void)hipHostFree(P.active_pinned);
    P.active_dev = nullptr;
}

// ---- measured practical roof: STREAM triad a = b + s c on caller-provided device arrays ("use the measured triad
// number as the practical roof too").  Four 16-byte accesses per lane and array in flight, one workgroup per 16 KiB of each array.
namespace {
__global__ __launch_bounds__(FG_BLOCK) void k_stream_triad(float4* __restrict__ a, const float4* __restrict__ b,
                                                           const float4* __restrict__ c, float s, size_t n4) {
    // one workgroup streams 4 x 256 consecutive float4 per array: all eight loads of a lane are issued before the first use
    const size_t base = (size_t)blockIdx.x * (FG_BLOCK * 4) + threadIdx.x;
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f* bv = reinterpret_cast<const v4f*>(b);
    const v4f* cv = reinterpret_cast<const v4f*>(c);
    v4f* av = reinterpret_cast<v4f*>(a);
    v4f x[4], y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = base + (size_t)k * FG_BLOCK;
        if (i < n4) { x[k] = __builtin_nontemporal_load(bv + i); y[k] = __builtin_nontemporal_load(cv + i); }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = base + (size_t)k * FG_BLOCK;
        if (i < n4) __builtin_nontemporal_store(x[k] + s * y[k], av + i);
    }
}
}  // namespace

extern "C" int fg_stream_triad(float* a, const float* b, const float* c, float scalar, int64_t n, int32_t reps, float* ms_per_launch,
                               void* stream) {
    FG_REQUIRE(a && b && c && n > 0 && n % 4 == 0 && reps > 0 && ms_per_launch, FG_ERR_INVALID_ARG, "fg_stream_triad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    FG_HIP_CHECK(hipEventCreate(&e0));
    FG_HIP_CHECK(hipEventCreate(&e1));
    const size_t n4 = (size_t)n / 4;
    const unsigned grid = (unsigned)((n4 + FG_BLOCK * 4 - 1) / (FG_BLOCK * 4));
    hipLaunchKernelGGL(k_stream_triad, dim3(grid), dim3(FG_BLOCK), 0, st, (float4*)a, (const float4*)b, (const float4*)c, scalar, n4);
    FG_HIP_CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(k_stream_triad, dim3(grid), dim3(FG_BLOCK), 0, st, (float4*)a, (const float4*)b, (const float4*)c, scalar, n4);
    FG_HIP_CHECK(hipEventRecord(e1, st));
    FG_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    FG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / (float)reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return FG_OK;
}
