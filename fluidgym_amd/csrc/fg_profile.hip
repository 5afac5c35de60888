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

#include <string.h>

#include "fg_internal.h"

namespace {
__global__ void k_prof_count(const int32_t* __restrict__ flags, int nsys, int32_t* __restrict__ out) {
    int c = 0;
    for (int i = threadIdx.x; i < nsys; i += 64) c += (flags[i] == 0);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if (threadIdx.x == 0) *out = c;
}
const char* const kNames[FG_PK_COUNT] = {"k_cg_ap", "k_cg_update", "k_bicg_p", "k_bicg_v", "k_bicg_s", "k_bicg_t",
                                         "k_bicg_x", "k_gemm_f32", "k_gemm_sk", "k_tridiag_y", "k_dct_rows", "k_line_y", "k_bicgf_a", "k_bicgf_b",
                                         "k_fcg_update_fwd", "k_fcg_inv_apply", "k_fbicg_fwd", "k_fbicg_inv", "k_jac_pass", "k_jac_stream", "k_tridiag_y_fac",
                                         "k_adv_build", "k_h", "k_div", "k_correct", "k_max_velocity"};
}  // namespace

int fg_prof_slot(const fg_state* cs, int kind, const int32_t* flags, int nsys, double bytes_per_sys,
                 double flops_per_sys, hipStream_t st) {
    if (!cs->prof.on) return -1;
    FgProf& P = const_cast<fg_state*>(cs)->prof;
    const long long k = P.launches[kind]++;
    if (P.used >= FG_PROF_POOL || (k % P.period) != 0) return -1;
    const int slot = P.used++;
    P.meta[slot] = FgProfMeta{kind, nsys, bytes_per_sys, flops_per_sys};
    // the count goes straight into the host-pinned word (no device-to-host copy per poll: those copies were 2.6 of the headline's
    // 57 launches per PISO step); only a kernel that counts itself (atomics) uses the device word, copied by fg_prof_prefetch
    P.self_counted[slot] = (flags == FG_PROF_SELF);
    if (flags == FG_PROF_SELF) (void)hipMemsetAsync(P.active_dev + slot, 0, sizeof(int32_t), st);  // kernel counts
    else if (flags) hipLaunchKernelGGL(k_prof_count, dim3(1), dim3(64), 0, st, flags, nsys, P.active_pinned + slot);
    else P.active_pinned[slot] = -1;   // "all active" (the slot's previous sample was collected before the pool index came round again)
    return slot;
}

void fg_prof_prefetch(fg_state* s, hipStream_t st) {
    FgProf& P = s->prof;
    if (!P.on || P.used <= P.prefetched) return;
    for (int i = P.prefetched; i < P.used; ++i)
        if (P.self_counted[i]) (void)hipMemcpyAsync(P.active_pinned + i, P.active_dev + i, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    P.prefetched = P.used;
}

int fg_prof_collect(fg_state* s, hipStream_t st) {
    FgProf& P = s->prof;
    if (!P.used) return FG_OK;
    if (P.prefetched < P.used) {   // samples taken after the last poll (or no poll at all): copy the self-counted ones and wait here
        for (int i = P.prefetched; i < P.used; ++i)
            if (P.self_counted[i]) FG_HIP_CHECK(hipMemcpyAsync(P.active_pinned + i, P.active_dev + i, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        FG_HIP_CHECK(hipStreamSynchronize(st));
    }
    P.prefetched = 0;
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
    P.on = on; P.used = 0; P.prefetched = 0;
    const char* e = getenv("FG_PROF_PERIOD");
    P.period = e && atoi(e) > 0 ? atoi(e) : 64;   // every 64th launch of a kind is timed (8 cost the timed region ~5 %: a count kernel + two events per sample; 32 until round 4)
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

// ---- accumulator-coherence litmus ---------------------------------------------------------------------------------
// The access pattern of the multi-kernel BiCGStab recurrence reduced to its skeleton: per iteration five launches (p, v, s, t, x)
// over `nsys` systems x `G` workgroups; every workgroup adds 1.0 to a sum slot with a device-scope atomic, the NEXT kernel reads
// the slot in every wave and expects exactly G, and the system's leader workgroup zeroes slots for later kernels -- seven sums in
// one 96-byte record per system, as in fg_mb_step.hip / fg_bicgstab.hip.  LD / ST select how the slots are read and zeroed:
// plain (round 1), agent-scope atomic load / store, or an atomic exchange for the zeroing.  A read that is not G is
// counted per slot and the first wrong value kept.
enum { L_RHO = 0, L_RV = 2, L_SS = 3, L_TS = 4, L_TT = 5, L_RR = 6, L_REC = 12 };
// LD: 0 plain load, 1 agent-scope atomic load.  ST: 0 plain store, 1 agent-scope atomic store, 2 atomic exchange (the path of the adds)
template <int LD> __device__ __forceinline__ double lit_ld(const double* p) {
    if constexpr (LD == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <int ST> __device__ __forceinline__ void lit_st(double* p, double v) {
    if constexpr (ST == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (ST == 2) (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <int LD> __device__ __forceinline__ int32_t lit_ldf(const int32_t* p) {
    if constexpr (LD == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <int ST> __device__ __forceinline__ void lit_stf(int32_t* p, int32_t v) {
    if constexpr (ST != 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
struct LitArgs { double* acc; float* vec; int64_t* bad; double* bad_value; int32_t* flag; int n; int it; };
__device__ __forceinline__ void lit_check(const LitArgs& q, int slot, double got) {
    if (threadIdx.x == 0 && got != (double)gridDim.x) {
        if (atomicAdd((unsigned long long*)(q.bad + slot), 1ull) == 0ull) q.bad_value[slot] = got;
    }
}
// the flag / scalar protocol of the same solvers: a word stored by the system's leader workgroup in kernel t (plain or atomic
// store) is read by every workgroup of the four kernels that follow; slot 11 counts reads that did not return it
__device__ __forceinline__ void lit_check_flag(const LitArgs& q, int got, int want) {
    if (threadIdx.x == 0 && got != want) {
        if (atomicAdd((unsigned long long*)(q.bad + 11), 1ull) == 0ull) q.bad_value[11] = (double)got;
    }
}
template <int LD, int ST, int K>
__global__ __launch_bounds__(FG_BLOCK) void k_litmus(LitArgs q) {
    const int sys = blockIdx.y;
    double* a = q.acc + (size_t)sys * L_REC;
    const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
    const int i = blockIdx.x * FG_BLOCK + threadIdx.x;
    float* v = q.vec + (size_t)sys * q.n;
    const int it = q.it;
    {   // systems store at different rates (every 1 / 2 / 3 iterations), as systems converge at different times in the solvers:
        // the 32 flags of a batch share one 128-byte line that is partially rewritten from different XCDs at different times
        const int m = 1 + sys % 3;
        if constexpr (K == 3) { if (leader && it % m == 0) lit_stf<ST>(q.flag + sys, it + 1); }
        else {
            const int last = (K == 4) ? it : it - 1;   // iteration of the latest t kernel before this one
            lit_check_flag(q, lit_ldf<LD>(q.flag + sys), last < 0 ? 0 : (last / m) * m + 1);
        }
    }
    if constexpr (K == 0) {          // p: reads rr, both rho; leader zeroes ss, ts, tt
        const double rr = lit_ld<LD>(a + L_RR), r0 = lit_ld<LD>(a + L_RHO + (it & 1));
        lit_check(q, L_RR, rr); lit_check(q, L_RHO + (it & 1), r0);
        if (leader) { lit_st<ST>(a + L_SS, 0.0); lit_st<ST>(a + L_TS, 0.0); lit_st<ST>(a + L_TT, 0.0); }
        if (i < q.n) v[i] = v[i] * 0.5f + (float)(rr - r0);
    } else if constexpr (K == 1) {   // v: accumulates rw.v
        if (i < q.n) v[i] += 1.f;
        if (threadIdx.x == 0) atomicAdd(a + L_RV, 1.0);
    } else if constexpr (K == 2) {   // s: reads rho, rw.v; leader zeroes next rho, rr; accumulates ss
        const double rv = lit_ld<LD>(a + L_RV), r0 = lit_ld<LD>(a + L_RHO + (it & 1));
        lit_check(q, L_RV, rv); lit_check(q, L_RHO + (it & 1), r0);
        if (leader) { lit_st<ST>(a + L_RHO + ((it + 1) & 1), 0.0); lit_st<ST>(a + L_RR, 0.0); }
        if (i < q.n) v[i] -= (float)(rv / r0);
        if (threadIdx.x == 0) atomicAdd(a + L_SS, 1.0);
    } else if constexpr (K == 3) {   // t: reads ss; accumulates ts, tt
        const double ss = lit_ld<LD>(a + L_SS);
        lit_check(q, L_SS, ss);
        if (i < q.n) v[i] += (float)ss * 1e-3f;
        if (threadIdx.x == 0) { atomicAdd(a + L_TS, 1.0); atomicAdd(a + L_TT, 1.0); }
    } else {                         // x: reads ts, tt; leader zeroes rw.v; accumulates rr and next rho
        const double ts = lit_ld<LD>(a + L_TS), tt = lit_ld<LD>(a + L_TT);
        lit_check(q, L_TS, ts); lit_check(q, L_TT, tt);
        if (leader) lit_st<ST>(a + L_RV, 0.0);
        if (i < q.n) v[i] -= (float)(ts / tt);
        if (threadIdx.x == 0) { atomicAdd(a + L_RR, 1.0); atomicAdd(a + L_RHO + ((it + 1) & 1), 1.0); }
    }
}
template <int LD, int ST>
__global__ void k_litmus_init(LitArgs q, int nsys, double g) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    for (int k = 0; k < L_REC; ++k) lit_st<ST>(q.acc + (size_t)s * L_REC + k, 0.0);
    lit_st<ST>(q.acc + (size_t)s * L_REC + L_RR, g);
    lit_st<ST>(q.acc + (size_t)s * L_REC + L_RHO, g);
}
template <int LD, int ST>
void litmus_run(LitArgs q, int nsys, int G, int iters, hipStream_t st) {
    hipLaunchKernelGGL((k_litmus_init<LD, ST>), dim3((nsys + 63) / 64), dim3(64), 0, st, q, nsys, (double)G);
    const dim3 grid(G, nsys), blk(FG_BLOCK);
    for (int it = 0; it < iters; ++it) {
        q.it = it;
        hipLaunchKernelGGL((k_litmus<LD, ST, 0>), grid, blk, 0, st, q);
        hipLaunchKernelGGL((k_litmus<LD, ST, 1>), grid, blk, 0, st, q);
        hipLaunchKernelGGL((k_litmus<LD, ST, 2>), grid, blk, 0, st, q);
        hipLaunchKernelGGL((k_litmus<LD, ST, 3>), grid, blk, 0, st, q);
        hipLaunchKernelGGL((k_litmus<LD, ST, 4>), grid, blk, 0, st, q);
    }
}
}  // namespace

// ---- order-independent accumulator (FgDacc, fg_internal.h): the host evaluation of the split / value code the kernels run, and
// the same sum done on the device by one atomic contribution per thread in whatever order the hardware schedules them
namespace {
__global__ void k_dacc_selftest(const double* __restrict__ v, long long n, FgDacc* __restrict__ acc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc_add(acc, v[i]);
}
__global__ void k_dacc_read(FgDacc* __restrict__ acc, double* __restrict__ out, double plain) {
    out[0] = acc_ld(acc);
    acc_st(acc, plain);
}
}  // namespace

extern "C" int fg_dacc_host_sum(const double* values, int64_t n, double plain, double* out_sum) {
    FG_REQUIRE(values && n >= 0 && out_sum, FG_ERR_INVALID_ARG, "fg_dacc_host_sum: bad argument");
    FgDacc a;
    memset(&a, 0, sizeof(a));
    a.plain = plain;
    for (int64_t i = 0; i < n; ++i) {
        long long k[5];
        if (!fg_dacc_split(values[i], k)) { a.poison += 1; continue; }
        for (int q = 0; q < 4; ++q) a.w[q] += (unsigned long long)k[q];
        a.w4 += (unsigned long long)k[4];
    }
    *out_sum = fg_dacc_host_value(a);
    return FG_OK;
}

extern "C" int fg_dacc_device_sum(const double* values_host, int64_t n, double plain, int32_t reps, double* out_sums_host, void* stream) {
    FG_REQUIRE(values_host && n > 0 && reps > 0 && out_sums_host, FG_ERR_INVALID_ARG, "fg_dacc_device_sum: bad argument");
    hipStream_t st = (hipStream_t)stream;
    double *v = nullptr, *out = nullptr;
    FgDacc* acc = nullptr;
    FG_HIP_CHECK(hipMalloc(&v, sizeof(double) * n));
    FG_HIP_CHECK(hipMalloc(&out, sizeof(double)));
    FG_HIP_CHECK(hipMalloc(&acc, sizeof(FgDacc)));
    FG_HIP_CHECK(hipMemcpy(v, values_host, sizeof(double) * n, hipMemcpyHostToDevice));
    FG_HIP_CHECK(hipMemset(acc, 0, sizeof(FgDacc)));
    hipLaunchKernelGGL(k_dacc_read, dim3(1), dim3(1), 0, st, acc, out, plain);   // parks `plain` through acc_st
    for (int r = 0; r < reps; ++r) {
        // different launch shapes per repetition: the arrival order of the contributions differs
        const int blk = 64 << (r % 4);
        hipLaunchKernelGGL(k_dacc_selftest, dim3((unsigned)((n + blk - 1) / blk)), dim3(blk), 0, st, v, (long long)n, acc);
        hipLaunchKernelGGL(k_dacc_read, dim3(1), dim3(1), 0, st, acc, out, plain);
        FG_HIP_CHECK(hipMemcpyAsync(out_sums_host + r, out, sizeof(double), hipMemcpyDeviceToHost, st));
    }
    FG_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(v); (void)hipFree(out); (void)hipFree(acc);
    return FG_OK;
}

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

extern "C" int fg_coherence_litmus(int32_t atomic_access, int32_t nsys, int32_t cells, int32_t iterations, int64_t* bad_reads,
                                   double* bad_value, void* stream) {
    FG_REQUIRE(nsys > 0 && cells > 0 && iterations > 0 && bad_reads && bad_value, FG_ERR_INVALID_ARG, "fg_coherence_litmus: bad argument");
    hipStream_t st = (hipStream_t)stream;
    LitArgs q{};
    q.n = cells;
    const int G = (cells + FG_BLOCK - 1) / FG_BLOCK;
    FG_HIP_CHECK(hipMalloc(&q.acc, sizeof(double) * L_REC * nsys));
    FG_HIP_CHECK(hipMalloc(&q.vec, sizeof(float) * (size_t)cells * nsys));
    FG_HIP_CHECK(hipMalloc(&q.bad, sizeof(int64_t) * L_REC));
    FG_HIP_CHECK(hipMalloc(&q.flag, sizeof(int32_t) * nsys));
    FG_HIP_CHECK(hipMemsetAsync(q.flag, 0, sizeof(int32_t) * nsys, st));
    FG_HIP_CHECK(hipMalloc(&q.bad_value, sizeof(double) * L_REC));
    FG_HIP_CHECK(hipMemsetAsync(q.vec, 0, sizeof(float) * (size_t)cells * nsys, st));
    FG_HIP_CHECK(hipMemsetAsync(q.bad, 0, sizeof(int64_t) * L_REC, st));
    FG_HIP_CHECK(hipMemsetAsync(q.bad_value, 0, sizeof(double) * L_REC, st));
    switch (atomic_access) {   // 10 * store + load
        case 0: litmus_run<0, 0>(q, nsys, G, iterations, st); break;
        case 1: litmus_run<1, 0>(q, nsys, G, iterations, st); break;
        case 10: litmus_run<0, 1>(q, nsys, G, iterations, st); break;
        case 11: litmus_run<1, 1>(q, nsys, G, iterations, st); break;
        case 20: litmus_run<0, 2>(q, nsys, G, iterations, st); break;
        case 21: litmus_run<1, 2>(q, nsys, G, iterations, st); break;
        default: (void)hipFree(q.acc); (void)hipFree(q.vec); (void)hipFree(q.bad); (void)hipFree(q.bad_value); (void)hipFree(q.flag);
                 FG_REQUIRE(false, FG_ERR_INVALID_ARG, "fg_coherence_litmus: access must be 0, 1, 10, 11, 20 or 21");
    }
    FG_HIP_CHECK(hipMemcpyAsync(bad_reads, q.bad, sizeof(int64_t) * L_REC, hipMemcpyDeviceToHost, st));
    FG_HIP_CHECK(hipMemcpyAsync(bad_value, q.bad_value, sizeof(double) * L_REC, hipMemcpyDeviceToHost, st));
    FG_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(q.acc); (void)hipFree(q.vec); (void)hipFree(q.bad); (void)hipFree(q.bad_value); (void)hipFree(q.flag);
    return FG_OK;
}
