// Does an XCD's L2 keep (clean) lines across a kernel boundary on MI355X? A reader kernel (256 workgroups, each
// summing its own 8 KiB slice of a 2 MiB buffer) is timed (a) cold: after 1 GiB of other traffic, (b) after a
// prefetch kernel with the SAME workgroup -> slice mapping (same XCD), (c) after a prefetch with a shifted mapping
// (lines end up in another XCD's L2 / the Infinity Cache only).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void reader(const uint4* w, int shift, unsigned* out) {
    const int b = (blockIdx.x + shift) % gridDim.x;
    const uint4* p = w + (size_t)b * 512;                  // 8 KiB per workgroup
    uint4 a = p[threadIdx.x], c = p[threadIdx.x + 256];
    unsigned s = a.x ^ a.y ^ a.z ^ a.w ^ c.x ^ c.y ^ c.z ^ c.w;
    if (s == 0x12345u) out[blockIdx.x] = s;                // keep the loads
}
__global__ __launch_bounds__(256) void flush(const uint4* big, size_t n, unsigned* out) {
    unsigned s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = big[i]; s ^= v.x ^ v.w; }
    if (s == 0x12345u) out[0] = s;
}
int main() {
    uint4 *w, *big; unsigned* out;
    const size_t nbig = (size_t)1 << 26;                   // 1 GiB
    CK(hipMalloc(&w, 2 << 20)); CK(hipMalloc(&big, nbig * 16)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(w, 1, 2 << 20)); CK(hipMemset(big, 2, nbig * 16));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"cold (after 1 GiB of other reads)", "after a prefetch with the same mapping (same XCD)", "after a prefetch with a shifted mapping"};
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 10; ++rep) {
            hipLaunchKernelGGL(flush, dim3(2048), dim3(256), 0, st, big, nbig, out);
            if (mode == 1) hipLaunchKernelGGL(reader, dim3(256), dim3(256), 0, st, w, 0, out);
            if (mode == 2) hipLaunchKernelGGL(reader, dim3(256), dim3(256), 0, st, w, 1, out);
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(reader, dim3(256), dim3(256), 0, st, w, 0, out);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; sum += ms;
        }
        printf("reader %s: min %.2f us, mean %.2f us\n", names[mode], best * 1000.f, sum * 100.f);
    }
    return 0;
}
