// What an env step does either side of its n sim steps, as two launches instead of ~25 elementwise torch launches: the action
// smoothing schedule + wall-jet slabs BEFORE (cylinder_env_base.py:748-753: a_k = a_{k-1} + alpha (target - a_{k-1}), applied per sim
// step; here all n controls of the env step at once), and reward + sensor observation AFTER (fluid_env.py:749-800: step() returns
// obs, reward, info from the fields the n-th sim step left).  Round 6 found the headline step bound by host time, 0.43 ms of 7.2 ms
// of it in these small launches (profiles/headline_glue.py); the kernels themselves move ~5 MB and ~4 MB.
//
// Both are deterministic per env (one workgroup per env in the observation kernel, a fixed summation tree): an env's reward does not
// depend on the batch it is stepped in.
#include "fg_internal.h"

namespace {

// jets [n][2 walls][B][2 components][1][X] = shape[c][x] * control[k][b];  control[k][b] = target[b] + (current[b] - target[b]) *
// decay[k] with separately rounded multiply and add (the values the elementwise torch expression gives);  last[b] = control[n-1][b]
__global__ __launch_bounds__(256) void k_jet_schedule(const float* __restrict__ target, const float* __restrict__ current,
                                                       const float* __restrict__ decay, const float* __restrict__ shape, int n, int B,
                                                       int X, float* __restrict__ jets, float* __restrict__ last) {
    const int x4 = X >> 2;                       // X % 4 == 0 (checked by the caller)
    const size_t total = (size_t)n * 2 * B * 2 * x4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % x4) * 4;
    const int c = (int)((i / x4) % 2);
    const int b = (int)((i / ((size_t)x4 * 2)) % B);
    const int w = (int)((i / ((size_t)x4 * 2 * B)) % 2);
    const int k = (int)(i / ((size_t)x4 * 2 * B * 2));
    const float t = target[b];
    const float diff = current[b] - t;
    float prod = diff * decay[k];
    // the torch expression rounds the product and the sum separately; the build (-ffp-contract=fast: the backend fuses whatever the
    // per-operation flags or __fmul_rn say) would make one fma of them -- the product passes through an opaque register move
    asm volatile("" : "+v"(prod));
    const float a = t + prod;
    const float4 s = *reinterpret_cast<const float4*>(shape + (size_t)c * X + x);
    float4 o;
    o.x = s.x * a; o.y = s.y * a; o.z = s.z * a; o.w = s.w * a;
    *reinterpret_cast<float4*>(jets + i * 4) = o;
    if (k == n - 1 && w == 0 && c == 0 && x == 0) last[b] = a;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// one workgroup per env: cross = mean(v^2), shear = scale * (mean_x u[y=0] + mean_x u[y=Y-1]), reward = -(shear + penalty * cross),
// obs_u[b][s][c] = u[b][c][sensor s], obs_p[b][s] = p[b][sensor s]
__global__ __launch_bounds__(256) void k_channel_observe(const float* __restrict__ u, const float* __restrict__ p,
                                                          const int64_t* __restrict__ sensor, int S, int Y, int X, float shear_scale,
                                                          float penalty, float* __restrict__ obs_u, float* __restrict__ obs_p,
                                                          float* __restrict__ cross, float* __restrict__ shear,
                                                          float* __restrict__ reward) {
    __shared__ double red[3][4];
    const int b = blockIdx.x, t = threadIdx.x;
    const size_t cells = (size_t)Y * X;
    const float* ub = u + (size_t)b * 2 * cells;
    const float* vb = ub + cells;
    float acc = 0.f;
    for (size_t i = (size_t)t * 4; i < cells; i += 1024) {       // cells % 4 == 0 (X % 4 == 0)
        const float4 q = *reinterpret_cast<const float4*>(vb + i);
        acc += q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    }
    float lo = 0.f, hi = 0.f;
    for (int x = t; x < X; x += 256) { lo += ub[x]; hi += ub[(size_t)(Y - 1) * X + x]; }
    double s0 = wave_sum((double)acc), s1 = wave_sum((double)lo), s2 = wave_sum((double)hi);
    if ((t & 63) == 0) { red[0][t >> 6] = s0; red[1][t >> 6] = s1; red[2][t >> 6] = s2; }
    for (int s = t; s < S; s += 256) {
        const int64_t cell = sensor[s];
        obs_u[((size_t)b * S + s) * 2] = ub[cell];
        obs_u[((size_t)b * S + s) * 2 + 1] = vb[cell];
        obs_p[(size_t)b * S + s] = p[(size_t)b * cells + cell];
    }
    __syncthreads();
    if (t == 0) {
        const double c = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (double)cells;
        const double l = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (double)X;
        const double h = ((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) / (double)X;
        const float cf = (float)c, sf = shear_scale * (float)(l + h);
        cross[b] = cf;
        shear[b] = sf;
        reward[b] = -(sf + penalty * cf);
    }
}

}  // namespace

extern "C" int fg_envglue_jet_schedule(const float* target, const float* current, const float* decay, const float* shape, int32_t n,
                                       int32_t batch, int32_t nx, float* jets, float* last, void* stream) {
    FG_REQUIRE(target && current && decay && shape && jets && last && n > 0 && batch > 0 && nx > 0 && nx % 4 == 0, FG_ERR_INVALID_ARG,
               "fg_envglue_jet_schedule: bad argument (nx must be a multiple of 4)");
    const size_t total = (size_t)n * 2 * batch * 2 * (nx / 4);
    hipLaunchKernelGGL(k_jet_schedule, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, target, current, decay,
                       shape, n, batch, nx, jets, last);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}

extern "C" int fg_envglue_channel_observe(const float* velocity, const float* pressure, const int64_t* sensor, int32_t n_sensors,
                                          int32_t batch, int32_t ny, int32_t nx, float shear_scale, float penalty, float* obs_velocity,
                                          float* obs_pressure, float* cross, float* shear, float* reward, void* stream) {
    FG_REQUIRE(velocity && pressure && sensor && obs_velocity && obs_pressure && cross && shear && reward && n_sensors > 0 && batch > 0 &&
                   ny > 1 && nx > 0 && nx % 4 == 0,
               FG_ERR_INVALID_ARG, "fg_envglue_channel_observe: bad argument (nx must be a multiple of 4)");
    hipLaunchKernelGGL(k_channel_observe, dim3(batch), dim3(256), 0, (hipStream_t)stream, velocity, pressure, sensor, n_sensors, ny, nx,
                       shear_scale, penalty, obs_velocity, obs_pressure, cross, shear, reward);
    FG_HIP_CHECK(hipGetLastError());
    return FG_OK;
}
