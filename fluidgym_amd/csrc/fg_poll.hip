// Host polls on pinned sequence words instead of hipStreamSynchronize (FgPoll, fg_internal.h).
#include <chrono>
#include <cstdlib>
#include <sched.h>

#include "fg_internal.h"

int fg_poll_create(FgPoll* P, int n) {
    P->n = n; P->epoch = 0;
    const char* e = getenv("FG_POLL_SPIN");
    P->spin = !(e && atoi(e) == 0);
    FG_HIP_CHECK(hipHostMalloc(&P->seq, sizeof(int32_t) * (size_t)n));
    for (int i = 0; i < n; ++i) P->seq[i] = 0;
    P->gran = nullptr; P->n_gran = 0;
#if !FG_F64
    // FG_POLL_WORDS=0: the mirror + release form everywhere (the A/B switch of the result words)
    const char* w = getenv("FG_POLL_WORDS");
    if (!P->spin || (w && atoi(w) == 0)) return FG_OK;
    P->n_gran = 8 * n;
    FG_HIP_CHECK(hipHostMalloc(&P->gran, sizeof(unsigned long long) * (size_t)P->n_gran));
    for (int i = 0; i < P->n_gran; ++i) P->gran[i] = 0ull;
#endif
    return FG_OK;
}

void fg_poll_destroy(FgPoll* P) {
    if (P->seq) (void)hipHostFree(P->seq);
    if (P->gran) (void)hipHostFree(P->gran);
    P->seq = nullptr; P->gran = nullptr;
}

FgPollOut fg_poll_next(FgPoll* P) {
    if (!P->spin || !P->seq) return FgPollOut{nullptr, 0};
    if (++P->epoch == 0) ++P->epoch;      // 0 is what the words start with
    return FgPollOut{P->seq, P->epoch, P->gran};
}

template <typename Ready>
static int poll_spin(int first, int count, hipStream_t st, Ready ready) {
    const auto t0 = std::chrono::steady_clock::now();
    int i = first;
    unsigned spins = 0;
    while (i < first + count) {
        if (ready(i)) { ++i; continue; }
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        sched_yield();
#endif
        // the kernels polled here take 5-50 us; past ~100-200 us of spinning (a long queue in front of the polled kernel) the core is
        // handed back between looks, so that several ranks per host -- one per GPU under ParallelFluidEnv -- or a CPU-limited
        // container do not starve the threads that launch (FG_POLL_SPIN=0 takes the spin away altogether: hipStreamSynchronize)
        if (spins > 0x1000) sched_yield();
        if ((++spins & 0xfff) == 0 &&
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 50.0) {
            // a long kernel queue, or a fault: let the runtime wait (and report)
            FG_HIP_CHECK(hipStreamSynchronize(st));
            for (int k = first; k < first + count; ++k)
                if (!ready(k)) {
                    fg_set_error("fg_poll_wait: the polled kernel finished without publishing its sequence word");
                    return FG_ERR_HIP;
                }
            return FG_OK;
        }
    }
    return FG_OK;
}

int fg_poll_wait(FgPoll* P, const FgPollOut& out, int first, int count, hipStream_t st) {
    if (!out.seq) { FG_HIP_CHECK(hipStreamSynchronize(st)); return FG_OK; }
    if (first < 0 || first + count > P->n) { fg_set_error("fg_poll_wait: range outside the sequence words"); return FG_ERR_INVALID_ARG; }
    return poll_spin(first, count, st, [&](int i) { return __atomic_load_n(out.seq + i, __ATOMIC_ACQUIRE) == out.value; });
}

int fg_poll_wait_words(FgPoll* P, const FgPollOut& out, int first, int count, hipStream_t st) {
    if (!out.gran) { fg_set_error("fg_poll_wait_words: no result words (the caller must use the mirror form)"); return FG_ERR_INVALID_ARG; }
    if (first < 0 || first + count > P->n_gran) { fg_set_error("fg_poll_wait_words: range outside the result words"); return FG_ERR_INVALID_ARG; }
    const unsigned long long tag = (unsigned long long)(uint32_t)out.value;
    return poll_spin(first, count, st, [&](int i) { return (__atomic_load_n(out.gran + i, __ATOMIC_ACQUIRE) >> 32) == tag; });
}

int fg_poll_wait_infos(FgPoll* P, const FgPollOut& out, int first, int count, fg_solve_info* pinned, hipStream_t st) {
    if (!out.gran) return fg_poll_wait(P, out, first, count, st);
    if (int rc = fg_poll_wait_words(P, out, 2 * first, 2 * count, st)) return rc;
    for (int i = first; i < first + count; ++i) fg_poll_info(P, i, pinned + i);
    return FG_OK;
}

// ---- FG_HTRACE=1: host-side time stamps around the polls (diagnosis of the idle time between a polled kernel and the launch that
// follows it); the deltas between consecutive tags are summed per pair and printed when the process ends
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <map>
#include <vector>
#include <string>
namespace {
struct HTrace {
    bool on; const char* last; std::chrono::steady_clock::time_point t;
    struct Row { long long n = 0; double sum = 0, mn = 1e30; std::vector<float> v; };
    std::map<std::string, Row> sum;
    HTrace() : on(false), last(nullptr) { const char* e = getenv("FG_HTRACE"); on = e && atoi(e) != 0; }
    ~HTrace() {
        if (!on) return;
        for (auto& kv : sum) {
            Row& r = kv.second;
            std::sort(r.v.begin(), r.v.end());
            fprintf(stderr, "FG_HTRACE %-44s n %8lld  avg_us %8.2f  min %8.2f  median %8.2f\n", kv.first.c_str(), r.n, r.sum / r.n, r.mn, r.v.empty() ? 0.0 : (double)r.v[r.v.size() / 2]);
        }
    }
};
HTrace g_htrace;
}  // namespace
void fg_htrace(const char* tag) {
    HTrace& H = g_htrace;
    if (!H.on) return;
    const auto now = std::chrono::steady_clock::now();
    if (H.last) {
        auto& e = H.sum[std::string(H.last) + " -> " + tag];
        const double us = std::chrono::duration<double, std::micro>(now - H.t).count();
        e.n += 1; e.sum += us; e.mn = us < e.mn ? us : e.mn;
        if (e.v.size() < (1u << 20)) e.v.push_back((float)us);
    }
    H.last = tag; H.t = now;
}

// ---- FG_ROCTX=1: named ranges for rocprofv3 --marker-trace (in place of the reference's profiling.py scopes,
// /root/reference .../pict/util/profiling.py:48-499).  The roctx library is resolved at run time (dlopen), so the build and a run
// without the switch do not depend on it.
#include <dlfcn.h>
namespace {
struct Roctx {
    bool on = false;
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = getenv("FG_ROCTX");
        if (!(e && atoi(e) != 0)) return;
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) { on = true; return; }
        }
        fprintf(stderr, "FG_ROCTX=1 but no roctx library could be loaded: ranges are off\n");
    }
};
Roctx g_roctx;
}  // namespace
void fg_range_push(const char* name) { if (g_roctx.on) (void)g_roctx.push(name); }
void fg_range_pop() { if (g_roctx.on) (void)g_roctx.pop(); }

