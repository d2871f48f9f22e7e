// anakin_amd/csrc/api_streams.hip - streams for SERVING: up to four per device, no two of which share a hardware queue.
//
// Role: the reference runs one Net per Worker pool thread, each on its own Context<T> streams (framework/core/worker.h,
// saber/core/context.h:38-77 - Env hands every context its data / compute streams), and the NV runtime gives every stream its own
// channel. The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4), assigned by creation
// order across the whole process: two Nets whose streams land on one queue run one after the other whatever their kernels leave idle.
// Measured (profiles/r06/multi_stream_curve.txt): the same k shared-device ResNet50 INT8 batch-8 nets reach 33 / 56 / 62 / 68k images/s
// at k = 1 .. 4 on streams picked here, 49 - 51k at k = 4 on the first four streams of another pool; more than four ACTIVE queues is worse
// than sharing (GPU_MAX_HW_QUEUES=8: 33k at k = 5), so four is the cap.
// How: a candidate stream is kept when a spin kernel on it OVERLAPS a spin kernel on every stream kept so far (two spins on one queue take
// twice one spin); ~3 ms once per device.
#include <mutex>

#include "api_internal.h"

namespace {

__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();             // constant 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

constexpr int kMaxServing = 4, kMaxDevices = 64, kCandidates = 24;
constexpr long long kSpinTicks = 12000;              // 120 us

struct Pool {
    hipStream_t s[kMaxServing];
    int n = -1;                                      // -1: not probed yet
};
Pool g_pool[kMaxDevices];
std::mutex g_pool_mu;

// elapsed ms from one start event to the LAST end of one spin per stream, issued together
hipError_t spin_ms(hipStream_t a, hipStream_t b, hipEvent_t e0, hipEvent_t ea, hipEvent_t eb, float* ms) {
    hipError_t e;
    if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
    if ((e = hipEventRecord(e0, a)) != hipSuccess) return e;
    if (b && (e = hipStreamWaitEvent(b, e0, 0)) != hipSuccess) return e;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, kSpinTicks);
    if (b) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, kSpinTicks);
    if ((e = hipEventRecord(ea, a)) != hipSuccess) return e;
    if (b && (e = hipEventRecord(eb, b)) != hipSuccess) return e;
    if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
    float t = 0.f, u = 0.f;
    if ((e = hipEventElapsedTime(&t, e0, ea)) != hipSuccess) return e;
    if (b && (e = hipEventElapsedTime(&u, e0, eb)) != hipSuccess) return e;
    *ms = t > u ? t : u;
    return hipSuccess;
}

hipError_t probe(Pool& p) {
    hipEvent_t e0, ea, eb;
    hipError_t e;
    if ((e = hipEventCreate(&e0)) != hipSuccess) return e;
    if ((e = hipEventCreate(&ea)) != hipSuccess) return e;
    if ((e = hipEventCreate(&eb)) != hipSuccess) return e;
    p.n = 0;
    std::vector<hipStream_t> rejected;
    for (int c = 0; c < kCandidates && p.n < kMaxServing; ++c) {
        hipStream_t s;
        if ((e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) break;
        bool keep = true;
        for (int i = 0; i < p.n && keep; ++i) {
            float solo = 1e30f, both = 1e30f, t;
            for (int r = 0; r < 3; ++r) {
                if ((e = spin_ms(p.s[i], nullptr, e0, ea, eb, &t)) != hipSuccess) break;
                solo = t < solo ? t : solo;
                if ((e = spin_ms(p.s[i], s, e0, ea, eb, &t)) != hipSuccess) break;
                both = t < both ? t : both;
            }
            if (e != hipSuccess) break;
            keep = both < 1.5f * solo;
        }
        if (e != hipSuccess) { rejected.push_back(s); break; }
        if (keep) p.s[p.n++] = s;
        else rejected.push_back(s);
    }
    // rejected candidates are destroyed only now: a destroyed stream's queue slot would be handed to the next candidate again
    for (auto s : rejected) (void)hipStreamDestroy(s);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    if (p.n == 0) { p.n = -1; return e != hipSuccess ? e : hipErrorUnknown; }
    return hipSuccess;
}

}  // namespace

int saber_hip_serving_streams(int n, saber_hip_stream_t* out, int* distinct) {
    if (n < 0 || (n > 0 && !out)) return fail(SABER_HIP_INVALID_VALUE, "saber_hip_serving_streams: n < 0 or null out");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(SABER_HIP_INVALID_VALUE, "saber_hip_serving_streams: device id");
    std::lock_guard<std::mutex> lk(g_pool_mu);
    Pool& p = g_pool[dev];
    if (p.n < 0) HIP_TRY(probe(p));
    for (int i = 0; i < n; ++i) out[i] = (saber_hip_stream_t)p.s[i % p.n];
    if (distinct) *distinct = p.n;
    return SABER_HIP_OK;
}
