// scripts/probe/pw_direct_probe.hip - PROTOTYPE (not product code) of the kernel DESIGN 8 item 3 asks for: an FP32 pointwise (1x1) convolution with the
// in-place sum + relu epilogue on the bf16 matrix cores WITHOUT LDS and WITHOUT barriers - every wave takes both operands straight into MFMA registers and
// runs on its own:   y[p][k] = relu(y[p][k] + bias[k] + sum_c x[p][c] * w[k][c]),   x [M][C], w [K][C], y [M][K] f32 (NHWC).
//   workgroup = 64 pixels x 64 output channels, 4 waves of 16 pixels; a lane (pixel = lane & 15, k-group = lane >> 4) loads 8 consecutive f32 of its pixel per
//   32-deep slab, splits them into the three bf16 planes (x = h + m + l exactly) in registers; the weights come pre-split from the host in MFMA A-fragment
//   order ([64-channel block][16-row tile][slab][plane][lane] x 16 B; tile i = channels base + 16 i .. + 15 in their natural order and the k-groups
//   interleaved in fours, so that every load / store INSTRUCTION covers 64 contiguous bytes of a pixel - the 16-consecutive-channels-per-lane order of the INT8
//   kernels scatters an f32 instruction over four 16-byte pieces per pixel); six plane products per slab in mma_step3's order (conv_igemm_impl.h); the residual is requested at kernel entry.
// Checked against a plain f32 FMA kernel (max error relative to max |y|), timed back to back and after a 384 MB sweep, next to what the product kernels need for the
// same layers (profiles/r04_resnet50_fp32/sequence.txt: 24.2 / 21.2 us).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off scripts/probe/pw_direct_probe.hip -o scripts/probe/_bin/pw_direct_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const v2f x = {x0, x1};
    const v2bf hb = __builtin_convertvector(x, v2bf);
    const v2f r1 = x - __builtin_convertvector(hb, v2f);
    const v2bf mb = __builtin_convertvector(r1, v2bf);
    const v2f r2 = r1 - __builtin_convertvector(mb, v2f);
    const v2bf lb = __builtin_convertvector(r2, v2bf);
    h = __builtin_bit_cast(unsigned, hb);
    m = __builtin_bit_cast(unsigned, mb);
    l = __builtin_bit_cast(unsigned, lb);
}
__device__ __forceinline__ v4f mma3(const v4i (&a)[3], const v4i (&b)[3], v4f c) {
#define MF(x, y) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, x), __builtin_bit_cast(v8bf, y), c, 0, 0, 0)
    MF(a[2], b[0]); MF(a[0], b[2]); MF(a[1], b[1]);
    MF(a[1], b[0]); MF(a[0], b[1]); MF(a[0], b[0]);
#undef MF
    return c;
}

template <int C, bool SUM>
__global__ __launch_bounds__(256) void pw_direct(const float* __restrict__ x, const v4i* __restrict__ wfrag, const float* __restrict__ bias, float* y, int M, int K) {
    constexpr int NS = C / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int kblocks = K >> 6;
    const int ptile = blockIdx.x / kblocks, kblk = blockIdx.x - ptile * kblocks;      // neighbouring workgroups share the pixel tile (x from L2)
    const int p = ptile * 64 + wave * 16 + frow;
    const int pc = p < M ? p : M - 1;
    const int kb = kblk * 64 + fq * 4;             // tile i: channels kb + 16 i .. + 3
    float4 rs[4];
    float* yr = y + (size_t)pc * K + kb;
    if (SUM) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rs[i] = *(const float4*)(yr + 16 * i);
    }
    const float* xr = x + (size_t)pc * C + fq * 4;
    v4i bp[NS][3];
    {
        float4 xv[NS][2];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            xv[s][0] = *(const float4*)(xr + s * 32);
            xv[s][1] = *(const float4*)(xr + s * 32 + 16);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            unsigned h[4], m[4], l[4];
            split3_pair(xv[s][0].x, xv[s][0].y, h[0], m[0], l[0]);
            split3_pair(xv[s][0].z, xv[s][0].w, h[1], m[1], l[1]);
            split3_pair(xv[s][1].x, xv[s][1].y, h[2], m[2], l[2]);
            split3_pair(xv[s][1].z, xv[s][1].w, h[3], m[3], l[3]);
            bp[s][0] = v4i{(int)h[0], (int)h[1], (int)h[2], (int)h[3]};
            bp[s][1] = v4i{(int)m[0], (int)m[1], (int)m[2], (int)m[3]};
            bp[s][2] = v4i{(int)l[0], (int)l[1], (int)l[2], (int)l[3]};
        }
    }
    const v4i* wf = wfrag + (size_t)kblk * (4 * NS * 3 * 64) + lane;
    v4f acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
        v4i a[NS][3];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[s][pl] = wf[((i * NS + s) * 3 + pl) * 64];
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[i] = mma3(a[s], bp[s], acc[i]);
    }
    if (p >= M) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 b = *(const float4*)(bias + kb + 16 * i);
        float4 o = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        if (SUM) { o.x = o.x + rs[i].x; o.y = o.y + rs[i].y; o.z = o.z + rs[i].z; o.w = o.w + rs[i].w; }
        o.x = fmaxf(o.x + b.x, 0.f); o.y = fmaxf(o.y + b.y, 0.f); o.z = fmaxf(o.z + b.z, 0.f); o.w = fmaxf(o.w + b.w, 0.f);
        *(float4*)(yr + 16 * i) = o;
    }
}

// Persistent form: a wave keeps the three weight planes of its TILES x 16 output channels in REGISTERS (C = 64, 4 tiles: 96 VGPRs) for the whole launch and
// walks over 16-pixel groups with the next group's x and residual already in flight - no weight re-fetch, no workgroup structure at all (waves are independent).
template <int C, int TILES, int D, bool SUM>
__global__ __launch_bounds__(256) void pw_persist(const float* __restrict__ x, const v4i* __restrict__ wfrag, const float* __restrict__ bias, float* y, int M, int K) {
    constexpr int NS = C / 32, NB = 4 / TILES;          // NB: channel blocks per 64 channels
    constexpr int R = D + 1;                            // register buffers: D groups in flight behind the one being combined
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int cblocks = (K >> 6) * NB;
    const int cb = gw % cblocks, slot = gw / cblocks, nslots = nw / cblocks;      // (host: nw % cblocks == 0)
    const int kb64 = cb / NB, i0 = (cb % NB) * TILES;
    const int kb = kb64 * 64 + i0 * 16 + fq * 4;          // tile i of this wave: channels kb + 16 i .. + 3
    v4i a[TILES][NS][3];
    {
        const v4i* wf = wfrag + (size_t)kb64 * (4 * NS * 3 * 64) + lane;
#pragma unroll
        for (int i = 0; i < TILES; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[i][s][pl] = wf[(((i0 + i) * NS + s) * 3 + pl) * 64];
    }
    float4 bs[TILES];
#pragma unroll
    for (int i = 0; i < TILES; ++i) bs[i] = *(const float4*)(bias + kb + 16 * i);
    const int groups = (M + 15) >> 4;
    // requests are UNCONDITIONAL (a group past the end re-reads the last one and is never combined): with a branch around them the
    // compiler's s_waitcnt insertion waits for everything at the first use and the groups in flight overlap nothing
    auto request = [&](int g, float4 (&xv)[NS][2], float4 (&rs)[TILES]) {
        g = g < groups ? g : groups - 1;
        const int p = g * 16 + frow;
        const int pc = p < M ? p : M - 1;
        const float* xr = x + (size_t)pc * C + fq * 4;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            xv[s][0] = *(const float4*)(xr + s * 32);
            xv[s][1] = *(const float4*)(xr + s * 32 + 16);
        }
        if (SUM) {
            const float* yr = y + (size_t)pc * K + kb;
#pragma unroll
            for (int i = 0; i < TILES; ++i) rs[i] = *(const float4*)(yr + 16 * i);
        }
    };
    auto finish = [&](int g, const float4 (&xv)[NS][2], const float4 (&rs)[TILES]) {
        v4i bp[NS][3];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            unsigned h[4], m[4], l[4];
            split3_pair(xv[s][0].x, xv[s][0].y, h[0], m[0], l[0]);
            split3_pair(xv[s][0].z, xv[s][0].w, h[1], m[1], l[1]);
            split3_pair(xv[s][1].x, xv[s][1].y, h[2], m[2], l[2]);
            split3_pair(xv[s][1].z, xv[s][1].w, h[3], m[3], l[3]);
            bp[s][0] = v4i{(int)h[0], (int)h[1], (int)h[2], (int)h[3]};
            bp[s][1] = v4i{(int)m[0], (int)m[1], (int)m[2], (int)m[3]};
            bp[s][2] = v4i{(int)l[0], (int)l[1], (int)l[2], (int)l[3]};
        }
        const int p = g * 16 + frow;
        float* yr = y + (size_t)(p < M ? p : M - 1) * K + kb;
        // term-major: the six plane products of a slab run over the TILES accumulators before the next product, so two consecutive MFMAs never
        // touch the same accumulator (accumulator-major, each waits out its predecessor: conv_igemm_impl.h)
        v4f accs[TILES];
#pragma unroll
        for (int i = 0; i < TILES; ++i) accs[i] = v4f{0.f, 0.f, 0.f, 0.f};
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < TILES; ++i)
                    accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a[i][s][PA[t]]), __builtin_bit_cast(v8bf, bp[s][PB[t]]), accs[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TILES; ++i) {
            const v4f acc = accs[i];
            float4 o = {acc[0], acc[1], acc[2], acc[3]};
            if (SUM) { o.x = o.x + rs[i].x; o.y = o.y + rs[i].y; o.z = o.z + rs[i].z; o.w = o.w + rs[i].w; }
            o.x = fmaxf(o.x + bs[i].x, 0.f); o.y = fmaxf(o.y + bs[i].y, 0.f); o.z = fmaxf(o.z + bs[i].z, 0.f); o.w = fmaxf(o.w + bs[i].w, 0.f);
            if (p < M) *(float4*)(yr + 16 * i) = o;
        }
    };
    float4 xv[R][NS][2], rs[R][TILES];
    if (slot >= groups) return;
#pragma unroll
    for (int j = 0; j < D; ++j) request(slot + j * nslots, xv[j], rs[j]);
    for (int base = 0;; base += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int g = slot + (base + j) * nslots;
            if (g >= groups) return;
            request(g + D * nslots, xv[(j + D) % R], rs[(j + D) % R]);
            finish(g, xv[j], rs[j]);
        }
    }
}

// Fourth form = the second (weights in registers, C = 64, four tiles) with the groups in flight requested by INLINE-ASM loads and waited for with an explicit
// s_waitcnt: gfx950 has ONE counter for loads and stores, and once a store is pending hipcc's wait-count pass answers every use of a loaded value with vmcnt(0)
// (conv_stage_coop.hip's finding) - in the forms above the stores of group g therefore make group g + 1 wait for the loads of ALL groups in flight,
// and a deeper ring only adds loads to wait for. Here the compiler does not see the loads; before group g is combined, everything older than the 12 D memory
// operations issued after its own request has landed (8 loads per group + its 4 stores... counted below).
__device__ __forceinline__ v4f asm_ld16(const float* p) {
    v4f v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
    return v;
}
template <int D, bool SUM>
__global__ __launch_bounds__(256) void pw_asm(const float* __restrict__ x, const v4i* __restrict__ wfrag, const float* __restrict__ bias, float* y, int M, int K) {
    constexpr int C = 64, NS = 2, R = D + 1;
    constexpr int PER_GROUP = 4 + (SUM ? 4 : 0) + 4;      // memory operations a group issues: 4 x loads (+ 4 residual loads) when requested, 4 stores when combined
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int cblocks = K >> 6;
    const int cb = gw % cblocks, slot = gw / cblocks, nslots = nw / cblocks;
    const int kb = cb * 64 + fq * 4;
    v4i a[4][NS][3];
    {
        const v4i* wf = wfrag + (size_t)cb * (4 * NS * 3 * 64) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[i][s][pl] = wf[((i * NS + s) * 3 + pl) * 64];
    }
    float4 bs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bs[i] = *(const float4*)(bias + kb + 16 * i);
    const int groups = (M + 15) >> 4;
    v4f xv[R][4], rs[R][4];
    auto request = [&](int g, v4f (&xq)[4], v4f (&rq)[4]) {
        g = g < groups ? g : groups - 1;
        const int p = g * 16 + frow;
        const int pc = p < M ? p : M - 1;
        const float* xr = x + (size_t)pc * C + fq * 4;
        xq[0] = asm_ld16(xr); xq[1] = asm_ld16(xr + 16); xq[2] = asm_ld16(xr + 32); xq[3] = asm_ld16(xr + 48);
        if (SUM) {
            const float* yr = y + (size_t)pc * K + kb;
            rq[0] = asm_ld16(yr); rq[1] = asm_ld16(yr + 16); rq[2] = asm_ld16(yr + 32); rq[3] = asm_ld16(yr + 48);
        }
    };
    auto finish = [&](int g, v4f (&xq)[4], v4f (&rq)[4]) {
        // everything older than the D groups requested after this one (and the stores issued in between) has landed; the "+v" ties keep the uses behind the wait
        if (SUM)
            asm volatile("s_waitcnt vmcnt(%8)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]), "+v"(rq[0]), "+v"(rq[1]), "+v"(rq[2]), "+v"(rq[3]) : "n"(D * PER_GROUP) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]) : "n"(D * PER_GROUP) : "memory");
        v4i bp[NS][3];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            unsigned h[4], m[4], l[4];
            const v4f f0 = xq[2 * s], f1 = xq[2 * s + 1];
            split3_pair(f0.x, f0.y, h[0], m[0], l[0]);
            split3_pair(f0.z, f0.w, h[1], m[1], l[1]);
            split3_pair(f1.x, f1.y, h[2], m[2], l[2]);
            split3_pair(f1.z, f1.w, h[3], m[3], l[3]);
            bp[s][0] = v4i{(int)h[0], (int)h[1], (int)h[2], (int)h[3]};
            bp[s][1] = v4i{(int)m[0], (int)m[1], (int)m[2], (int)m[3]};
            bp[s][2] = v4i{(int)l[0], (int)l[1], (int)l[2], (int)l[3]};
        }
        v4f accs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) accs[i] = v4f{0.f, 0.f, 0.f, 0.f};
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a[i][s][PA[t]]), __builtin_bit_cast(v8bf, bp[s][PB[t]]), accs[i], 0, 0, 0);
        const int p = g * 16 + frow;
        float* yr = y + (size_t)(p < M ? p : M - 1) * K + kb;
        const bool ok = p < M;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v4f o = accs[i];
            if (SUM) o = o + rq[i];
            o = o + v4f{bs[i].x, bs[i].y, bs[i].z, bs[i].w};
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            // always 4 stores per group (a lane past M rewrites the last pixel's value with itself... no: it must not write) - predicate by address: such lanes store to
            // their own clamped pixel only when ok; the count of ISSUED stores must not depend on it, so the store is issued under EXEC masking by the compiler
            if (ok) *(v4f*)(yr + 16 * i) = o;
        }
    };
    if (slot >= groups) return;
#pragma unroll
    for (int j = 0; j < D; ++j) request(slot + j * nslots, xv[j], rs[j]);
    for (int base = 0;; base += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int g = slot + (base + j) * nslots;
            if (g >= groups) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                return;
            }
            request(g + D * nslots, xv[(j + D) % R], rs[(j + D) % R]);
            finish(g, xv[j], rs[j]);
        }
    }
}

// Fifth form = the fourth with every global access a FULL-LINE access: the MFMA operand / result layouts give a lane 16 bytes of a pixel and its three k-group
// neighbours the next 48, so a load or store instruction of the forms above touches sixteen 64-byte half lines (measured: 4.1 TB/s where 1 KB-contiguous instructions
// stream at 7.1, pw_stream_probe). Here a wave reads and writes whole rows (lane L, step j -> 16-byte element L + 64 j of the 16-pixel tile: 1 KB contiguous per
// instruction) and changes layout through a PRIVATE slice of LDS (pixel pitch 17 x 16 bytes: conflict-free both ways). Waves still never synchronise with each other.
template <int D, bool SUM>
__global__ __launch_bounds__(256) void pw_tr(const float* __restrict__ x, const v4i* __restrict__ wfrag, const float* __restrict__ bias, float* y, int M, int K) {
    constexpr int C = 64, NS = 2, R = D + 1;
    constexpr int PER_GROUP = 4 + (SUM ? 4 : 0) + 4;
    __shared__ v4f tx[4][16 * 17], ty[4][16 * 17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int cblocks = K >> 6;
    const int cb = gw % cblocks, slot = gw / cblocks, nslots = nw / cblocks;
    v4i a[4][NS][3];
    {
        const v4i* wf = wfrag + (size_t)cb * (4 * NS * 3 * 64) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[i][s][pl] = wf[((i * NS + s) * 3 + pl) * 64];
    }
    v4f bs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bs[i] = *(const v4f*)(bias + cb * 64 + i * 16 + fq * 4);
    const int groups = (M + 15) >> 4;
    v4f* mx = tx[wave];
    v4f* my = ty[wave];
    // row-order element e = lane + 64 j of a tile: pixel e >> 4, 16-byte chunk e & 15
    int lrow[4], px_of[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = lane + 64 * j;
        px_of[j] = e >> 4;
        lrow[j] = (e >> 4) * 17 + (e & 15);
    }
    v4f xv[R][4], rs[R][4];
    auto request = [&](int g, v4f (&xq)[4], v4f (&rq)[4]) {
        g = g < groups ? g : groups - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int p = g * 16 + px_of[j];
            p = p < M ? p : M - 1;
            xq[j] = asm_ld16(x + (size_t)p * C + ((lane + 64 * j) & 15) * 4);
        }
        if (SUM) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int p = g * 16 + px_of[j];
                p = p < M ? p : M - 1;
                rq[j] = asm_ld16(y + (size_t)p * K + cb * 64 + ((lane + 64 * j) & 15) * 4);
            }
        }
    };
    auto finish = [&](int g, v4f (&xq)[4], v4f (&rq)[4]) {
        if (SUM)
            asm volatile("s_waitcnt vmcnt(%8)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]), "+v"(rq[0]), "+v"(rq[1]), "+v"(rq[2]), "+v"(rq[3]) : "n"(D * PER_GROUP) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]) : "n"(D * PER_GROUP) : "memory");
        // rows -> LDS -> MFMA B layout (lane = pixel frow, k-group fq: chunks s * 8 + fq and s * 8 + 4 + fq of its pixel)
#pragma unroll
        for (int j = 0; j < 4; ++j) mx[lrow[j]] = xq[j];
        if (SUM) {
#pragma unroll
            for (int j = 0; j < 4; ++j) my[lrow[j]] = rq[j];
        }
        v4i bp[NS][3];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const v4f f0 = mx[frow * 17 + s * 8 + fq], f1 = mx[frow * 17 + s * 8 + 4 + fq];
            unsigned h[4], m[4], l[4];
            split3_pair(f0.x, f0.y, h[0], m[0], l[0]);
            split3_pair(f0.z, f0.w, h[1], m[1], l[1]);
            split3_pair(f1.x, f1.y, h[2], m[2], l[2]);
            split3_pair(f1.z, f1.w, h[3], m[3], l[3]);
            bp[s][0] = v4i{(int)h[0], (int)h[1], (int)h[2], (int)h[3]};
            bp[s][1] = v4i{(int)m[0], (int)m[1], (int)m[2], (int)m[3]};
            bp[s][2] = v4i{(int)l[0], (int)l[1], (int)l[2], (int)l[3]};
        }
        v4f accs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) accs[i] = v4f{0.f, 0.f, 0.f, 0.f};
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a[i][s][PA[t]]), __builtin_bit_cast(v8bf, bp[s][PB[t]]), accs[i], 0, 0, 0);
        // result layout: lane = pixel frow, tile i -> chunk i * 4 + fq of its 64-channel row; combined in place in the LDS tile, then stored row-wise
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v4f o = accs[i];
            if (SUM) o = o + my[frow * 17 + i * 4 + fq];
            o = o + bs[i];
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            my[frow * 17 + i * 4 + fq] = o;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = g * 16 + px_of[j];
            const v4f o = my[lrow[j]];
            float* dst = y + (size_t)(p < M ? p : M - 1) * K + cb * 64 + ((lane + 64 * j) & 15) * 4;
            if (p < M) *(v4f*)dst = o;
        }
    };
    if (slot >= groups) return;
#pragma unroll
    for (int j = 0; j < D; ++j) request(slot + j * nslots, xv[j], rs[j]);
    for (int base = 0;; base += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int g = slot + (base + j) * nslots;
            if (g >= groups) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                return;
            }
            request(g + D * nslots, xv[(j + D) % R], rs[(j + D) % R]);
            finish(g, xv[j], rs[j]);
        }
    }
}

// Third form: the 64-channel block's weight planes live in LDS (24 KB at C = 64, 48 KB at C = 128, copied once per workgroup) instead of in every wave's registers:
// 4 x less weight traffic from L2 in the prologue, and the registers go to groups in flight (D) instead. The four waves of a workgroup share the channel block and
// walk over different 16-pixel groups; after the one barrier behind the copy they never synchronise again.
template <int C, int D, bool SUM>
__global__ __launch_bounds__(256) void pw_lds(const float* __restrict__ x, const v4i* __restrict__ wfrag, const float* __restrict__ bias, float* y, int M, int K) {
    constexpr int NS = C / 32, R = D + 1, NF = 4 * NS * 3 * 64;      // NF: 16-byte fragments of one channel block
    __shared__ v4i wl[NF];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int cblocks = K >> 6;
    const int cb = blockIdx.x % cblocks, wslot = blockIdx.x / cblocks, nwslots = gridDim.x / cblocks;      // (host: gridDim.x % cblocks == 0)
    const int slot = wslot * 4 + wave, nslots = nwslots * 4;
    const int kb = cb * 64 + fq * 4;
    for (int i = threadIdx.x; i < NF; i += 256) wl[i] = wfrag[(size_t)cb * NF + i];
    float4 bs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bs[i] = *(const float4*)(bias + kb + 16 * i);
    const int groups = (M + 15) >> 4;
    auto request = [&](int g, float4 (&xv)[NS][2], float4 (&rs)[4]) {
        g = g < groups ? g : groups - 1;
        const int p = g * 16 + frow;
        const int pc = p < M ? p : M - 1;
        const float* xr = x + (size_t)pc * C + fq * 4;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            xv[s][0] = *(const float4*)(xr + s * 32);
            xv[s][1] = *(const float4*)(xr + s * 32 + 16);
        }
        if (SUM) {
            const float* yr = y + (size_t)pc * K + kb;
#pragma unroll
            for (int i = 0; i < 4; ++i) rs[i] = *(const float4*)(yr + 16 * i);
        }
    };
    auto finish = [&](int g, const float4 (&xv)[NS][2], const float4 (&rs)[4]) {
        asm volatile("" ::: "memory");                 // the weight fragments are re-read from LDS for every group (hoisted, they would be 96 registers again)
        v4f accs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) accs[i] = v4f{0.f, 0.f, 0.f, 0.f};
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            unsigned h[4], m[4], l[4];
            split3_pair(xv[s][0].x, xv[s][0].y, h[0], m[0], l[0]);
            split3_pair(xv[s][0].z, xv[s][0].w, h[1], m[1], l[1]);
            split3_pair(xv[s][1].x, xv[s][1].y, h[2], m[2], l[2]);
            split3_pair(xv[s][1].z, xv[s][1].w, h[3], m[3], l[3]);
            const v4i bp[3] = {v4i{(int)h[0], (int)h[1], (int)h[2], (int)h[3]}, v4i{(int)m[0], (int)m[1], (int)m[2], (int)m[3]},
                               v4i{(int)l[0], (int)l[1], (int)l[2], (int)l[3]}};
            v4i af[4][3];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) af[i][pl] = wl[((i * NS + s) * 3 + pl) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af[i][PA[t]]), __builtin_bit_cast(v8bf, bp[PB[t]]), accs[i], 0, 0, 0);
        }
        const int p = g * 16 + frow;
        float* yr = y + (size_t)(p < M ? p : M - 1) * K + kb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 o = {accs[i][0], accs[i][1], accs[i][2], accs[i][3]};
            if (SUM) { o.x = o.x + rs[i].x; o.y = o.y + rs[i].y; o.z = o.z + rs[i].z; o.w = o.w + rs[i].w; }
            o.x = fmaxf(o.x + bs[i].x, 0.f); o.y = fmaxf(o.y + bs[i].y, 0.f); o.z = fmaxf(o.z + bs[i].z, 0.f); o.w = fmaxf(o.w + bs[i].w, 0.f);
            if (p < M) *(float4*)(yr + 16 * i) = o;
        }
    };
    float4 xv[R][NS][2], rs[R][4];
#pragma unroll
    for (int j = 0; j < D; ++j) request(slot + j * nslots, xv[j], rs[j]);
    __syncthreads();                                   // the weights are in LDS
    if (slot >= groups) return;
    for (int base = 0;; base += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int g = slot + (base + j) * nslots;
            if (g >= groups) return;
            request(g + D * nslots, xv[(j + D) % R], rs[(j + D) % R]);
            finish(g, xv[j], rs[j]);
        }
    }
}

__global__ void reference(const float* x, const float* w, const float* bias, const float* y0, float* out, int M, int C, int K, int sum) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)M * K) return;
    const int p = (int)(idx / K), k = (int)(idx % K);
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += (double)x[(size_t)p * C + c] * (double)w[(size_t)k * C + c];
    float d = (float)s;
    if (sum) d = d + y0[idx];
    d = d + bias[k];
    out[idx] = d > 0.f ? d : 0.f;
}

static uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

template <int C>
static void run_case(const char* name, int M, int K, double product_us, hipStream_t st) {
    const int NS = C / 32;
    std::vector<float> hx((size_t)M * C), hw((size_t)K * C), hb(K), hy((size_t)M * K);
    srand(7);
    auto rnd = [] { return (float)(rand() % 2001 - 1000) / 1000.f; };
    for (auto& v : hx) v = rnd() > 0 ? rnd() : 0.f;       // post-relu like activations
    for (auto& v : hw) v = rnd() * 0.17f;
    for (auto& v : hb) v = rnd() * 0.3f;
    for (auto& v : hy) v = rnd() * 2.f;
    // weights -> three bf16 planes in A-fragment order
    std::vector<uint16_t> frag((size_t)(K / 64) * 4 * NS * 3 * 64 * 8);
    for (int kb = 0; kb < K / 64; ++kb)
        for (int i = 0; i < 4; ++i)
            for (int s = 0; s < NS; ++s)
                for (int L = 0; L < 64; ++L) {
                    const int r = L & 15, kg = L >> 4;
                    const int ch = kb * 64 + i * 16 + r;                                            // natural: a store instruction covers 64 contiguous bytes per pixel
                    for (int j = 0; j < 8; ++j) {
                        const int kk = s * 32 + (j < 4 ? kg * 4 + j : 16 + kg * 4 + (j - 4));         // k order: a load instruction covers 64 contiguous bytes per pixel
                        const float w = hw[(size_t)ch * C + kk];
                        const uint16_t h = bf16_rne(w);
                        const float r1 = w - bf16_f(h);
                        const uint16_t m = bf16_rne(r1);
                        const float r2 = r1 - bf16_f(m);
                        const uint16_t l = bf16_rne(r2);
                        const uint16_t pl[3] = {h, m, l};
                        for (int q = 0; q < 3; ++q) frag[(((((size_t)kb * 4 + i) * NS + s) * 3 + q) * 64 + L) * 8 + j] = pl[q];
                    }
                }
    float *x, *w, *b, *y, *y0, *ref;
    v4i* wf;
    void* big;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&b, hb.size() * 4));
    CK(hipMalloc(&y, hy.size() * 4)); CK(hipMalloc(&y0, hy.size() * 4)); CK(hipMalloc(&ref, hy.size() * 4));
    CK(hipMalloc(&wf, frag.size() * 2)); CK(hipMalloc(&big, (size_t)384 << 20));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(y0, hy.data(), hy.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wf, frag.data(), frag.size() * 2, hipMemcpyHostToDevice));
    const int grid = ((M + 63) / 64) * (K / 64);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // variants: 0 = one workgroup per 64 x 64 tile; then persistent waves <TILES, D> at 1 / 2 / 3 / 4 workgroups per CU where the registers allow
    struct V { int tiles, d, per_cu; };
    // tiles 9: the LDS form (pw_lds) with d groups in flight
    // tiles 8: pw_asm (C = 64 only)
    // tiles 7: pw_tr (full-line accesses through wave-private LDS, C = 64 only)
    const V vs[] = {{0, 0, 0}, {4, 1, 1}, {4, 1, 2}, {2, 2, 1}, {2, 2, 2}, {9, 1, 1}, {9, 2, 1}, {8, 1, 1}, {8, 2, 1}, {7, 1, 1}, {7, 2, 1}, {7, 1, 2}, {7, 2, 2}};
    for (const V& v : vs)
    for (int sum = 1; sum >= 0; --sum) {
        if (C == 128 && (v.tiles == 4 || v.tiles == 8 || v.tiles == 7)) continue;       // (192 weight registers)
        const int variant = v.tiles;
        const int cblocks = (variant == 9 || variant == 8 || variant == 7) ? K / 64 : (variant ? (K / 64) * (4 / v.tiles) : 1);
        int pgrid = 256 * v.per_cu;
        if (variant && variant != 9 && (pgrid * 4) % cblocks) continue;
        if (variant == 9) pgrid = pgrid / cblocks * cblocks;
        auto launch = [&] {
#define PWL(T, DD) do { if (sum) hipLaunchKernelGGL((pw_persist<C, T, DD, true>), dim3(pgrid), dim3(256), 0, st, x, wf, b, y, M, K); \
                        else hipLaunchKernelGGL((pw_persist<C, T, DD, false>), dim3(pgrid), dim3(256), 0, st, x, wf, b, y, M, K); } while (0)
            if (variant == 0) {
                if (sum) hipLaunchKernelGGL((pw_direct<C, true>), dim3(grid), dim3(256), 0, st, x, wf, b, y, M, K);
                else hipLaunchKernelGGL((pw_direct<C, false>), dim3(grid), dim3(256), 0, st, x, wf, b, y, M, K);
            } else if (v.tiles == 7) {
                if constexpr (C == 64) {
                    if (v.d == 1) { if (sum) hipLaunchKernelGGL((pw_tr<1, true>), dim3(pgrid), dim3(256), 0, st, x, wf, b, y, M, K);
                                    else hipLaunchKernelGGL((pw_tr<1, false>), dim3(pgrid), dim3(256), 0, st, x, wf, b, y, M, K); }
                    else { if (sum) hipLaunchKernelGGL((pw_tr<2, true>), dim3(pgrid), dim3(256), 0, st, x, wf, b, y, M, K);
                           else hipLaunchKernelGGL((pw_tr<2, false>), dim3(pgrid), dim3(256), 0, st, x, wf, b, y, M, K); }
                }
            } else if (v.tiles == 8) {
                if constexpr (C == 64) {
                    if (v.d == 1) { if (sum) hipLaunchKernelGGL((pw_asm<1, true>), dim3(pgrid), dim3(256), 0, st, x, wf, b, y, M, K);
                                    else hipLaunchKernelGGL((pw_asm<1, false>), dim3(pgrid), dim3(256), 0, st, x, wf, b, y, M, K); }
                    else { if (sum) hipLaunchKernelGGL((pw_asm<2, true>), dim3(pgrid), dim3(256), 0, st, x, wf, b, y, M, K);
                           else hipLaunchKernelGGL((pw_asm<2, false>), dim3(pgrid), dim3(256), 0, st, x, wf, b, y, M, K); }
                }
            } else if (v.tiles == 9) {
#define PLL(DD) do { if (sum) hipLaunchKernelGGL((pw_lds<C, DD, true>), dim3(pgrid), dim3(256), 0, st, x, wf, b, y, M, K); \
                     else hipLaunchKernelGGL((pw_lds<C, DD, false>), dim3(pgrid), dim3(256), 0, st, x, wf, b, y, M, K); } while (0)
                if (v.d == 1) PLL(1); else if (v.d == 2) PLL(2); else PLL(3);
#undef PLL
            } else if (v.tiles == 4) { if constexpr (C == 64) PWL(4, 1); }
            else if (v.tiles == 2 && v.d == 2) PWL(2, 2);
            else if (v.tiles == 2 && v.d == 3) PWL(2, 3);
            else if (v.tiles == 1 && v.d == 3) PWL(1, 3);
            else PWL(1, 4);
#undef PWL
        };
        char vname[32];
        if (variant == 9) snprintf(vname, sizeof vname, "lds d%d", v.d);
        else if (variant == 8) snprintf(vname, sizeof vname, "asm loads d%d", v.d);
        else if (variant == 7) snprintf(vname, sizeof vname, "full lines d%d", v.d);
        else snprintf(vname, sizeof vname, variant ? "persist t%d d%d" : "per tile", v.tiles, v.d);
        CK(hipMemcpyAsync(y, y0, hy.size() * 4, hipMemcpyDeviceToDevice, st));
        launch();
        hipLaunchKernelGGL(reference, dim3((unsigned)(((size_t)M * K + 255) / 256)), dim3(256), 0, st, x, w, b, y0, ref, M, C, K, sum);
        CK(hipStreamSynchronize(st));
        std::vector<float> got(hy.size()), want(hy.size());
        CK(hipMemcpy(got.data(), y, got.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(want.data(), ref, want.size() * 4, hipMemcpyDeviceToHost));
        double maxabs = 0, maxerr = 0;
        for (size_t i = 0; i < got.size(); ++i) {
            maxabs = std::max(maxabs, (double)std::fabs(want[i]));
            maxerr = std::max(maxerr, (double)std::fabs(got[i] - want[i]));
        }
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 30; ++i) launch();      // (the in-place sum keeps accumulating: timing only)
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double warm = ms * 1e3 / 30;
        double cold = 0;
        for (int i = 0; i < 6; ++i) {
            CK(hipMemsetAsync(big, i, (size_t)384 << 20, st));
            CK(hipEventRecord(e0, st));
            launch();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (i) cold += ms * 1e3 / 5;
        }
        const double mb = ((double)M * C * 4 + (sum ? 2.0 : 1.0) * M * K * 4 + (double)K * C * 6) / 1e6;
        printf("%-26s %-14s %-14s %5d workgroups: max error %.2e of max |y| %.2f (%s) | back to back %6.2f us = %4.1f TB/s, %5.1f TF f32-equivalent | after a 384 MB sweep %6.2f us"
               " (one event pair) | the product kernels: %.1f us\n", name, sum ? "+ sum in place" : "no sum", vname, variant ? pgrid : grid, maxerr / maxabs, maxabs,
               maxerr / maxabs < 2e-5 ? "ok" : "WRONG", warm, mb / warm, 2.0 * M * K * C / warm / 1e6, cold, sum ? product_us : 0.0);
    }
    CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(b)); CK(hipFree(y)); CK(hipFree(y0)); CK(hipFree(ref)); CK(hipFree(wf)); CK(hipFree(big));
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    run_case<64>("res2 64->256 @56 b8", 8 * 56 * 56, 256, 24.2, st);
    run_case<128>("res3 128->512 @28 b8", 8 * 28 * 28, 512, 21.2, st);
    run_case<64>("res2 64->256 @56 b1", 56 * 56, 256, 0.0, st);
    run_case<64>("ragged 64->128, 1000 px", 1000, 128, 0.0, st);
    return 0;
}
