// Measures what a device-wide barrier costs on MI355X (256 persistent workgroups, one per CU) against the
// dependent-kernel boundary of a hipGraph, with and without a cross-workgroup data hand-off per step.
// Decides whether multi-layer persistent kernels can beat one launch per layer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* count, unsigned target, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();                                   // release: this workgroup's stores visible device-wide
        __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { *err = 1; ok = false; break; }   // never hang the GPU
        }
        __threadfence();                                   // acquire
    }
    __syncthreads();
    return ok;
}

// mode 0: barrier only. mode 1: each step every workgroup writes 4 KiB, barrier, reads another workgroup's 4 KiB.
__global__ __launch_bounds__(256) void persistent(int steps, int mode, unsigned* count, int* err, unsigned* buf,
                                                  unsigned long long* sum) {
    const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    unsigned acc = 0;
    for (int s = 0; s < steps; ++s) {
        unsigned* cur = buf + (size_t)(s & 1) * nb * 1024;
        if (mode) {
            uint4 v = make_uint4(s * 3 + b, t, s, b);
            ((uint4*)(cur + (size_t)b * 1024))[t] = v;
        }
        if (!grid_barrier(count, (unsigned)(s + 1) * nb, err)) return;
        if (mode) {
            const int src = (b + 97) % nb;
            const uint4 v = ((const uint4*)(cur + (size_t)src * 1024))[t];
            acc += v.x + v.y + v.z + v.w;
            if (v.x != (unsigned)(s * 3 + src) || v.z != (unsigned)s) atomicExch(err, 2);   // stale data
        }
    }
    if (mode && t == 0) atomicAdd(sum, (unsigned long long)acc);
}
__global__ __launch_bounds__(256) void one_step(int s, int mode, unsigned* buf, int* err, unsigned long long* sum) {
    const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    unsigned* prev = buf + (size_t)((s + 1) & 1) * nb * 1024;
    unsigned* cur = buf + (size_t)(s & 1) * nb * 1024;
    if (mode) {
        if (s > 0) {
            const int src = (b + 97) % nb;
            const uint4 v = ((const uint4*)(prev + (size_t)src * 1024))[t];
            if (v.x != (unsigned)((s - 1) * 3 + src)) atomicExch(err, 2);
        }
        ((uint4*)(cur + (size_t)b * 1024))[t] = make_uint4(s * 3 + b, t, s, b);
    }
}

int main() {
    const int nb = 256, steps = 200;
    unsigned *count, *buf; int* err; unsigned long long* sum;
    CK(hipMalloc(&count, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&sum, 8)); CK(hipMalloc(&buf, (size_t)2 * nb * 4096));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemsetAsync(count, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st)); CK(hipMemsetAsync(sum, 0, 8, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(persistent, dim3(nb), dim3(256), 0, st, steps, mode, count, err, buf, sum);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("persistent kernel, %s: %.2f us per step (err=%d)\n", mode ? "4 KiB hand-off + barrier" : "barrier only",
               best * 1000.f / steps, herr);
        // the same chain as dependent kernels in a hipGraph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(one_step, dim3(nb), dim3(256), 0, st, s, mode, buf, err, sum);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("hipGraph of dependent kernels, %s: %.2f us per step (err=%d)\n", mode ? "4 KiB hand-off" : "empty", best * 1000.f / steps, herr);
    }
    return 0;
}
