// gap between a kernel that sets a word and a kernel gated on it by hipStreamWaitValue32, against no gate and against a host round trip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_work(float* p, int n, int reps) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float v = p[i]; for (int r = 0; r < reps; ++r) v = v * 1.0001f + 0.5f; p[i] = v; }
}
__global__ void k_set(unsigned* w, unsigned v, int* pinned, int pv) {
    if (threadIdx.x == 0) {
        __hip_atomic_store(w, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (pinned) __hip_atomic_store(pinned, pv, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n = 1 << 22;
    float* buf; CK(hipMalloc(&buf, sizeof(float) * n));
    unsigned* word = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&word, 8, hipMallocSignalMemory);
    printf("signal memory alloc: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) CK(hipMalloc(&word, 8));
    CK(hipMemset(word, 0, 8));
    int* pinned; CK(hipHostMalloc(&pinned, 64)); pinned[0] = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200;
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: A, set, B back to back | 1: A, set, waitvalue, B | 2: A, set(pinned), host spin, B
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (int it = 1; it <= iters; ++it) {
            hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, st, buf, n, 8);
            hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, st, word, (unsigned)(mode * 1000 + it), pinned, mode * 1000 + it);
            if (mode == 1) {
                hipError_t w = hipStreamWaitValue32(st, word, (unsigned)(mode * 1000 + it), hipStreamWaitValueEq, 0xffffffffu);
                if (w != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(w)); return 1; }
            }
            if (mode == 2) { while (__atomic_load_n(pinned, __ATOMIC_ACQUIRE) != mode * 1000 + it) __builtin_ia32_pause(); }
            hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, st, buf, n, 8);
        }
        CK(hipStreamSynchronize(st));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
        printf("mode %d: %.2f us per (work, set, [gate], work)\n", mode, us);
    }
    // a gate that is already open when reached, enqueued far ahead: pure cost of the wait packet
    CK(hipStreamSynchronize(st));
    hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, st, word, 7777u, (int*)nullptr, 0);
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, st, buf, n, 8);
        CK(hipStreamWaitValue32(st, word, 7777u, hipStreamWaitValueEq, 0xffffffffu));
    }
    CK(hipStreamSynchronize(st));
    printf("open gate: %.2f us per (work, gate)\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters);
    t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, st, buf, n, 8);
    CK(hipStreamSynchronize(st));
    printf("no gate:   %.2f us per work\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters);
    return 0;
}
