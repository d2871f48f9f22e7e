// scripts/probe/boundary_probe.hip — what does a dependent kernel boundary cost on this box, and what would replace it?
//
// Round-3 probe for two review questions:
//  (1) "the dependent-kernel boundary measures 2.3-2.7 us here, the guide's table says 1.45 (trivial) to 1.7-1.9 (streaming):
//      where do the 0.6-1 us go?"  -> chains of N dependent launches, eager and hipGraph, varying ONE thing at a time:
//      kernel-argument bytes (8 / 128 / 320), grid size (256 / 1024 workgroups), and the bytes the predecessor leaves DIRTY in
//      the XCD L2s (0 / 1 / 4 / 8 MB written with plain stores, then read by the successor): the guide prices the
//      end-of-kernel write-back at "+ B / 6 TB/s".
//  (2) "an XCD-resident stage kernel: 32 workgroups of ONE XCD separated by an XCD-local barrier (no agent-scope fence: the
//      hand-off never leaves that XCD's L2); go if <= 1.2 us with a 4 KB hand-off" -> a persistent kernel of 256 workgroups
//      (one per CU), 8 independent groups by HW_REG_XCC_ID; each phase: every workgroup writes 4 KB (plain or sc1 stores),
//      arrives on its XCD's counter (relaxed agent atomic add, 8 counters on separate 128-B lines), polls it with sc1 loads,
//      then READS the 4 KB of its right-hand neighbour on the same XCD with sc1 (L1-bypassing) loads and checks EVERY word.
//      Reported: us per phase (host-paired over 200 phases), and the number of stale words seen (must be 0 to be usable).
//      Variants: hand-off 0 / 4 KB / 32 KB; 'fenced' = the placement-independent protocol (release fence + acquire fence, agent).
//  (3) round 4: the same persistent kernel with a DEVICE-wide barrier (what a weight-stationary res5 kernel would need between
//      its convolutions): mode 3 = two levels (arrival on the XCD's counter, the XCD's first workgroup forwards one arrival to
//      a global counter, polls it and releases its XCD through a local "go" word), mode 4 = flat (all 256 workgroups on one counter);
//      release fence before the arrival, acquire after the release, and the 4 KB read is the buffer of a workgroup on the NEXT XCD.
// Every spin is bounded (a stuck barrier sets a timeout word and the kernel ends).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probe/boundary_probe.hip -o scripts/probe/boundary_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args8 { unsigned* out; };
struct Args128 { unsigned* out; unsigned pad[30]; };
struct Args320 { unsigned* out; unsigned pad[78]; };

template <typename A>
__global__ __launch_bounds__(256) void null_kernel(const A a) {
    if (a.out == nullptr) a.out[threadIdx.x] = 0;
}

// writes `bytes` of dst (plain 16-byte stores), after reading the same amount of src: the successor's src is our dst
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        uint4 v = src[i];
        v.x += 1;
        dst[i] = v;
    }
}

template <typename F>
static double chain_us(hipStream_t st, int n, bool graph, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double best = 1e30;
    if (graph) {
        hipGraph_t g;
        hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < n; ++i) launch(i);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ex, st));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best = std::min(best, (double)ms * 1e3 / n);
        }
        CK(hipGraphExecDestroy(ex));
        CK(hipGraphDestroy(g));
    } else {
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < n; ++i) launch(i);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best = std::min(best, (double)ms * 1e3 / n);
        }
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return best;
}

// ---------------------------------------------------------------------------------------------------------- XCD barrier
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ unsigned load_sc1(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // global_load ... sc1: bypasses L1
}

struct BarArgs {
    unsigned* counters;     // [8][32] dwords: one 128-byte line per XCD
    unsigned* slot;         // [8]: next free slot index per XCD (workgroups number themselves within their XCD)
    unsigned* data;         // [256][words] hand-off buffers
    unsigned* stale;        // [1] count of stale words seen
    unsigned* timeout;      // [1] set when a spin gave up
    unsigned* xcd_hist;     // [8] workgroups per XCD
    int phases, words, mode;   // mode 0: plain stores + vmcnt(0); 1: sc1 stores; 2: release / acquire fences (agent)
};

__global__ __launch_bounds__(256) void xcd_barrier_kernel(const BarArgs a) {
    __shared__ unsigned s_x, s_me;
    if (threadIdx.x == 0) {
        s_x = xcc_id() & 7;
        s_me = atomicAdd(&a.slot[s_x], 1u);
        atomicAdd(&a.xcd_hist[s_x], 1u);
    }
    __syncthreads();
    const unsigned x = s_x, me = s_me;
    __shared__ unsigned s_n;
    // group size: all workgroups of this XCD; wait until every workgroup of the grid has registered (bounded)
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int spin = 0; spin < 2000000; ++spin) {
            tot = 0;
            for (int i = 0; i < 8; ++i) tot += load_sc1(&a.slot[i]);
            if (tot >= gridDim.x) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (tot < gridDim.x) atomicExch(a.timeout, 1u);
        s_n = load_sc1(&a.slot[x]);
    }
    __syncthreads();
    const unsigned n = s_n;
    const bool dev = a.mode >= 3;       // device-wide barrier: read across XCDs (every XCD holds the same number of workgroups here)
    unsigned* mine = a.data + ((size_t)x * 64 + me) * a.words;
    const unsigned* theirs = dev ? a.data + ((size_t)((x + 1) & 7) * 64 + me % n) * a.words : a.data + ((size_t)x * 64 + (me + 1) % n) * a.words;
    unsigned* ctr = a.counters + x * 32;
    unsigned* gctr = a.counters + 8 * 32;           // the global lines (modes 3, 4)
    // one device-wide barrier (thread 0): `which` selects the counter set (0: data published, 1: readers done)
    auto dev_barrier = [&](int ph, int which) {
        unsigned* lc = ctr + which * 16;
        unsigned* gc = gctr + which * 32;
        unsigned* go = ctr + 8 + which * 16;
        int spin = 0;
        if (a.mode == 4) {
            __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (load_sc1(gc) < (unsigned)ph * gridDim.x) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > 4000000) { atomicExch(a.timeout, 1u); break; }
            }
            return;
        }
        __hip_atomic_fetch_add(lc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (me == 0) {
            while (load_sc1(lc) < (unsigned)ph * n) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > 4000000) { atomicExch(a.timeout, 1u); break; }
            }
            __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (load_sc1(gc) < (unsigned)ph * 8u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > 4000000) { atomicExch(a.timeout, 1u); break; }
            }
            __hip_atomic_store(go, (unsigned)ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (load_sc1(go) < (unsigned)ph) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > 4000000) { atomicExch(a.timeout, 1u); break; }
            }
        }
    };
    unsigned bad = 0;
    __shared__ unsigned s_abort;
    if (me >= 64) {      // more workgroups on one XCD than the hand-off area holds: give up cleanly (the others time out)
        if (threadIdx.x == 0) atomicExch(a.timeout, 1u);
        return;
    }
    for (int ph = 1; ph <= a.phases; ++ph) {
        if (threadIdx.x == 0) s_abort = load_sc1(a.timeout);
        __syncthreads();
        if (s_abort) break;
        // publish
        for (int w = threadIdx.x; w < a.words; w += 256) {
            const unsigned v = (unsigned)ph * 0x10001u + (unsigned)w;
            if (a.mode == 1) __hip_atomic_store(&mine[w], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else mine[w] = v;
        }
        if (a.mode >= 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (dev) {
            if (threadIdx.x == 0) {
                dev_barrier(ph, 0);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            for (int w = threadIdx.x; w < a.words; w += 256) bad += theirs[w] != (unsigned)ph * 0x10001u + (unsigned)w;
            __syncthreads();
            if (threadIdx.x == 0) dev_barrier(ph, 1);
            __syncthreads();
            continue;
        }
        // arrive + wait: monotonic counter, n arrivals per phase
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)ph * n;
            int spin = 0;
            while (load_sc1(ctr) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > 4000000) { atomicExch(a.timeout, 1u); break; }
            }
            if (a.mode == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        // consume the neighbour's buffer, every word checked
        for (int w = threadIdx.x; w < a.words; w += 256) {
            const unsigned v = (a.mode == 2) ? theirs[w] : load_sc1(&theirs[w]);
            bad += v != (unsigned)ph * 0x10001u + (unsigned)w;
        }
        __syncthreads();     // nobody overwrites its buffer before the neighbour has read it ... next phase's barrier orders that:
        // a second arrival keeps the protocol simple: readers done -> writers may overwrite
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)ph * n;
            int spin = 0;
            while (load_sc1(ctr + 16) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > 4000000) { atomicExch(a.timeout, 1u); break; }
            }
        }
        __syncthreads();
    }
    if (bad) atomicAdd(a.stale, bad);
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("# device %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    unsigned* d_out;
    CK(hipMalloc(&d_out, 4096));
    const int N = 400;
    printf("# (1) dependent-kernel boundary: us per launch of a %d-launch chain (min of 5 repetitions)\n", N);
    for (int grid : {256, 1024}) {
        Args8 a8{d_out};
        Args128 a128{d_out, {0}};
        Args320 a320{d_out, {0}};
        for (int g = 0; g < 2; ++g) {
            const double t8 = chain_us(st, N, g, [&](int) { hipLaunchKernelGGL(null_kernel<Args8>, dim3(grid), dim3(256), 0, st, a8); });
            const double t128 = chain_us(st, N, g, [&](int) { hipLaunchKernelGGL(null_kernel<Args128>, dim3(grid), dim3(256), 0, st, a128); });
            const double t320 = chain_us(st, N, g, [&](int) { hipLaunchKernelGGL(null_kernel<Args320>, dim3(grid), dim3(256), 0, st, a320); });
            printf("null kernel, %4d workgroups, %-5s: kernarg 8 B %.2f | 128 B %.2f | 320 B %.2f\n", grid, g ? "graph" : "eager", t8, t128, t320);
        }
    }
    // dirty bytes: kernel i reads buffer (i & 1), writes buffer (i & 1) ^ 1
    for (size_t mb : {(size_t)0, (size_t)1, (size_t)4, (size_t)8, (size_t)16}) {
        const size_t bytes = mb ? mb << 20 : 4096;
        uint4 *b0, *b1;
        CK(hipMalloc(&b0, bytes));
        CK(hipMalloc(&b1, bytes));
        CK(hipMemset(b0, 0, bytes));
        CK(hipMemset(b1, 0, bytes));
        for (int g = 0; g < 2; ++g) {
            const double t = chain_us(st, N, g, [&](int i) {
                hipLaunchKernelGGL(stream_kernel, dim3(1024), dim3(256), 0, st, (i & 1) ? b1 : b0, (i & 1) ? b0 : b1, bytes / 16);
            });
            printf("streaming kernel, 1024 workgroups, %-5s, %5.1f MB read + written per launch: %.2f us per launch (%.0f GB/s)\n",
                   g ? "graph" : "eager", bytes / 1048576.0, t, 2.0 * bytes / t / 1e3);
        }
        CK(hipFree(b0));
        CK(hipFree(b1));
    }
    // (2) XCD-local barrier
    printf("# (2) XCD-local barrier + hand-off, persistent kernel, one 256-thread workgroup per CU (grid %d), 200 phases\n", prop.multiProcessorCount);
    const int grid = prop.multiProcessorCount;
    for (int mode = 0; mode < 5; ++mode) {
        if (mode == 3) printf("# (3) DEVICE-wide barrier + hand-off across XCDs, same kernel: two-level / flat\n");
        for (int words : {0, 1024, 8192}) {
            BarArgs a;
            CK(hipMalloc(&a.counters, 10 * 32 * 4));
            CK(hipMalloc(&a.slot, 32));
            CK(hipMalloc(&a.stale, 4));
            CK(hipMalloc(&a.timeout, 4));
            CK(hipMalloc(&a.xcd_hist, 32));
            CK(hipMalloc(&a.data, (size_t)8 * 64 * (words ? words : 1) * 4));
            a.words = words;
            a.mode = mode;
            double us[2] = {0, 0};
            unsigned stale = 0, tmo = 0, hist[8];
            const int phases[2] = {20, 220};
            for (int r = 0; r < 2; ++r) {        // host-paired: (220 phases) - (20 phases) = 200 phases
                a.phases = phases[r];
                CK(hipMemsetAsync(a.counters, 0, 10 * 32 * 4, st));
                CK(hipMemsetAsync(a.slot, 0, 32, st));
                CK(hipMemsetAsync(a.stale, 0, 4, st));
                CK(hipMemsetAsync(a.timeout, 0, 4, st));
                CK(hipMemsetAsync(a.xcd_hist, 0, 32, st));
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0));
                CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(xcd_barrier_kernel, dim3(grid), dim3(256), 0, st, a);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                us[r] = ms * 1e3;
                unsigned v;
                CK(hipMemcpy(&v, a.stale, 4, hipMemcpyDeviceToHost));
                stale += v;
                CK(hipMemcpy(&v, a.timeout, 4, hipMemcpyDeviceToHost));
                tmo |= v;
                CK(hipMemcpy(hist, a.xcd_hist, 32, hipMemcpyDeviceToHost));
            }
            printf("%-34s hand-off %5d B: %.2f us per phase (two arrivals per phase), stale words %u%s; workgroups per XCD %u %u %u %u %u %u %u %u\n",
                   mode == 0 ? "plain stores + vmcnt(0), sc1 loads" : (mode == 1 ? "sc1 stores, sc1 loads" : (mode == 2 ? "release / acquire fences (agent)" :
                   (mode == 3 ? "device-wide, two levels, fences" : "device-wide, one counter, fences"))),
                   words * 4, (us[1] - us[0]) / 200.0, stale, tmo ? "  [TIMEOUT]" : "", hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
            CK(hipFree(a.counters)); CK(hipFree(a.slot)); CK(hipFree(a.stale)); CK(hipFree(a.timeout)); CK(hipFree(a.xcd_hist)); CK(hipFree(a.data));
        }
    }
    return 0;
}
