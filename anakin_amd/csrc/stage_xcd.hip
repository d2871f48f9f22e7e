// anakin_amd/csrc/stage_xcd.hip - XCD-resident stage kernel (gfx950): a run of INT8 convolutions over small feature maps
// (ResNet res5: 7x7) as ONE persistent launch.
//
// Why: at batch 8 the res5 convolutions are 392 pixels each; as separate launches every one of them costs a dependent
// kernel boundary plus a cold start (4.6 - 7 us per op in the forward pass, profiles/r03/sequence_b8.txt) for 0.2 - 0.5 us
// of matrix-core work. What a boundary buys is visibility of one op's output to the next across the 8 L2s. Here an IMAGE
// never leaves its XCD: XCD x runs every conv of the stage for images x, x + 8, ... on its 32 CUs, the edge tensors are
// handed from phase to phase through that XCD's own L2 (plain stores, s_waitcnt vmcnt(0), one arrival counter per XCD,
// L1 invalidate, loads), and the phases are separated by an XCD-local barrier (0.9 us measured,
// profiles/r03/boundary_probe.txt) instead of a kernel boundary. Every edge is still written to its tensor (the executor's
// other ops and the parity tests read them); the results are the bits of the separate launches - int32 accumulation in any
// order is exact and the requantisation epilogues are the shared ones (epilogue_pack.h).
//
//   grid = 256 workgroups x 256 threads, one per CU (LDS > 80 KB keeps a second one off the CU). A workgroup reads its XCD
//   from HW_REG_XCC_ID and takes a number 0..31 within it from a per-XCD registration counter.
//   Phase = one convolution (1x1, or 3x3 pad 1, stride 1) on one image: GEMM  D[cout][64 pixel slots] with the weights as
//   the MFMA A operand. CU c owns output channels [c*16*NT, (c+1)*16*NT); its 4 waves split the REDUCTION (k-steps
//   [w*KQ, (w+1)*KQ)), so the whole weight slice of a wave (NT*KQ <= 20 KB-sized steps) is loaded into registers up
//   front - BEFORE the wait on the barrier, they do not depend on it - and is reused for every image of the XCD.
//   The input image (<= 64 pixels x cin) comes into LDS by DMA (pixel pitch cin + 16 bytes, a zero row for padding
//   taps and unused pixel slots), the four partial accumulators meet in LDS, wave w finishes pixel slots [16w, 16w+16).
// The barrier's spin runs on SCALAR loads (s_dcache_inv + s_load): vector loads return in order, a spinning vector load
// would queue behind the wave's weight loads.
// A workgroup that waits longer than ~20 ms (the 256 workgroups are not co-resident: another kernel holds CUs) raises the
// abort flag and everybody leaves; saber_hip_stage_status reports it.
#include "epilogue_pack.h"

namespace saber_mi355x {

namespace {

typedef const __attribute__((address_space(4))) StagePhase* PhasePtr;

__device__ __forceinline__ unsigned stage_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
// a 64-bit word as the XCD's L2 holds it now (the scalar cache is invalidated first)
__device__ __forceinline__ unsigned long long stage_sload(const unsigned long long* p) {
    unsigned long long v;
    asm volatile("s_dcache_inv\n\ts_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(p) : "memory");
    return v;
}
__device__ __forceinline__ void stage_dma16(const void* src, void* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}
// ... with sc1: the load misses this CU's L1 and reads the XCD's L2 - what another workgroup of the XCD stored in an earlier phase
__device__ __forceinline__ void stage_dma16_l2(const void* src, void* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 16);
}

struct StageCtx {
    int xcd, cu;                  // this workgroup's XCD and its number within it
    int img0, img_step;           // images this workgroup computes: img0, img0 + img_step, ... (stage: its XCD's, step 8)
    unsigned long long gen;       // launch generation of the registration counter
    unsigned long long* ctr;      // this XCD's arrival counter
    unsigned long long* abort_w;
    int bar;                      // barriers passed so far in this launch
    int hw;
    int py[4], px[4];             // this lane's pixel in each of the 4 pixel fragments (py < 0: unused slot)
    unsigned long long* tr;       // this workgroup's trace row for the current phase, or null
};

constexpr int kSpinLimit = 60000;

// NT: 16-channel tiles per CU, KQ: k-steps (64 bytes of reduction) per wave, IS3: 3x3 pad-1 conv (k-steps ordered [tap][c])
template <int NT, int KQ, bool IS3>
__device__ __forceinline__ bool stage_phase(const StageKArgs& a, PhasePtr php, StageCtx& cx, v4i* lds, unsigned* s_abort) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int cin = php->cin, cout = php->cout;
    const int HW = cx.hw;
#define STAGE_TR(i) do { if (cx.tr && tid == 0) cx.tr[i] = wall_clock64(); } while (0)
    STAGE_TR(0);
    // ---- arrive: this workgroup's stores of the earlier phases are in the L2 ------------------------------------------
    const bool barrier = php->barrier != 0;
    if (barrier) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cx.ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    STAGE_TR(1);
    // ---- this wave's weights and the epilogue constants of its pixel fragment: registers, ahead of the wait -------------
    v4i wreg[NT * KQ];
    {
        const v4i* wp = (const v4i*)a.weights + php->w_chunk + (size_t)((cx.cu * 4 + wave) * (NT * KQ)) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NT * KQ; ++i) wreg[i] = wp[i * 64];
    }
    const int ch0 = cx.cu * (16 * NT) + fq * 4;          // tile j: channels ch0 + 16 j .. + 3
    v4i prm[NT][3];
    {
        const v4i* pp = (const v4i*)a.prm + php->prm_chunk + (ch0 / 4) * 3;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 3; ++r) prm[j][r] = pp[j * 12 + r];
    }
    STAGE_TR(2);
    // ---- wait for the other 31 workgroups of the XCD ----------------------------------------------------------------------
    if (barrier) {
        cx.bar += 1;
        if (tid == 0) {
            const unsigned long long target = (cx.gen * (unsigned long long)a.n_barriers + (unsigned long long)cx.bar) * 32ull;
            int spins = 0;
            while (stage_sload(cx.ctr) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit || ((spins & 63) == 0 && stage_sload(cx.abort_w) != 0ull)) {
                    __hip_atomic_store(cx.abort_w, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *s_abort = 1u;
                    break;
                }
            }
        }
        __syncthreads();
        if (*s_abort) return false;
        // (round 3 had `buffer_inv sc1` here - a DEVICE-scope acquire, which on this multi-XCD part also drops every non-coherent
        // line of the XCD's L2: 32 workgroups x 10 phases wiped their XCD's cached weights and images continuously, which is
        // most of what profiles/r03/stage_trace.txt blamed on the fabric ports. What other workgroups stored is read with sc1
        // loads instead - stage_dma16_l2 for the image, an agent-scope atomic load for the residual - found with the cooperative
        // chain's in-kernel stamps, conv_chain_coop.hip)
    }
    STAGE_TR(3);
    const unsigned pch = php->pch, mg = php->mg_pch;
    const int chunks = cin >> 4;
    const int nA = (int)(((unsigned)(HW + 1) * pch + 63u) / 64u);       // DMA instructions (64 chunks each) for the image
    const int xmask = php->in_u8 ? (int)0x80808080u : 0;
    v4i* red = lds + php->red_chunk;
    const char* xin = (const char*)a.t[php->in_t];
    char* yout = (char*)a.t[php->out_t];
    const char* rin = php->elt ? (const char*)a.t[php->res_t] : nullptr;
    const float lo = php->relu ? 0.f : -3.0e38f;
    const float off = php->out_u8 ? 0.f : 128.f;
    const unsigned xo = php->out_u8 ? 0u : 0x80808080u;
    const float lo_s8 = php->relu ? 0.f : -128.f;
    const float res_lo = php->res_relu ? 0.f : -3.0e38f;
    struct { float coeff_conv, scale_conv, coeff_res, scale_res; } ec = {php->coeff_conv, php->scale_conv, php->coeff_res, php->scale_res};
    const int pool_t = php->pool_t;
    v4i* pool_part = red + 4 * NT * 4 * 64;             // [wave][tile][channel quad] int32 x 4, behind the partial accumulators
    const int kspt = chunks >> 2;                       // k-steps per tap
    const int my_p = wave * 16 + frow;                  // epilogue: this lane's pixel slot
    bool first = true;
    for (int img = cx.img0; img < a.n_img; img += cx.img_step) {
        // ---- the input image -> LDS ------------------------------------------------------------------------------------
        if (php->reload || !first || cx.img0 + cx.img_step < a.n_img) {       // (several images for this workgroup: LDS holds the last one)
            __syncthreads();                            // nobody still reads the previous image / partial sums
            const char* xg = xin + (size_t)img * HW * cin;
            for (int i = wave; i < nA; i += 4) {
                const unsigned L = (unsigned)(i * 64 + lane);
                const unsigned hp = __umulhi(L, mg), cc = L - hp * pch;
                const bool in = hp < (unsigned)HW && cc < (unsigned)chunks;
                const char* src = in ? xg + (size_t)hp * cin + cc * 16 : (const char*)a.zero;
                if (barrier) stage_dma16_l2(src, lds + i * 64);
                else stage_dma16(src, lds + i * 64);
            }
        }
        if (first) STAGE_TR(4);
        first = false;
        unsigned rs[NT];
        if (rin) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int ch = ch0 + 16 * j;
                // (sc1 load: the residual may be an earlier phase's output, stored by another workgroup of this XCD)
                rs[j] = (my_p < HW && ch < cout) ? __hip_atomic_load((const unsigned*)(rin + ((size_t)img * HW + my_p) * cout + ch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (img == cx.img0) STAGE_TR(5);
        // ---- this wave's quarter of the reduction, all 4 pixel fragments ---------------------------------------------------
        v4i acc[NT][4];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[j][m] = v4i{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            const int ks = wave * KQ + i;
            int kc = ks, dy = 1, dx = 1;
            if constexpr (IS3) {
                const int tap = ks / kspt;
                kc = ks - tap * kspt;
                dy = tap / 3;
                dx = tap - dy * 3;
            }
            v4i bf[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int ny = cx.py[m] + dy - 1, nx = cx.px[m] + dx - 1;
                const bool ok = cx.py[m] >= 0 && (unsigned)ny < (unsigned)a.H && (unsigned)nx < (unsigned)a.W;
                const int q = ok ? ny * a.W + nx : HW;   // HW: the zero row
                v4i b = lds[q * (int)pch + kc * 4 + fq];
                b.x ^= xmask; b.y ^= xmask; b.z ^= xmask; b.w ^= xmask;
                bf[m] = b;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[j][m] = mma_step(wreg[i * NT + j], bf[m], acc[j][m]);
        }
        if (img == cx.img0) STAGE_TR(6);
        // ---- the four partial sums meet in LDS; wave w finishes pixel fragment w ---------------------------------------------
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m) red[((wave * NT + j) * 4 + m) * 64 + lane] = acc[j][m];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            v4i s = red[((0 * NT + j) * 4 + wave) * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < 4; ++w2) {
                const v4i t = red[((w2 * NT + j) * 4 + wave) * 64 + lane];
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            const int ch = ch0 + 16 * j;
            unsigned o;
            if (rin) o = chain_elt_pack(s, prm[j][2], __builtin_bit_cast(v4f, prm[j][1]), __builtin_bit_cast(v4f, prm[j][0]), rs[j], lo_s8, res_lo, ec);
            else o = chain_out_pack(s, prm[j][2], __builtin_bit_cast(v4f, prm[j][1]), __builtin_bit_cast(v4f, prm[j][0]), lo, off, xo);
            if (my_p < HW && ch < cout) *(unsigned*)(yout + ((size_t)img * HW + my_p) * cout + ch) = o;
            if (pool_t >= 0) {      // global average pooling of the 8-bit values just stored: exact int32 sums, this wave's 16 pixels first
                int ps[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int b = (int)((o >> (8 * t)) & 0xffu);
                    ps[t] = (php->out_u8 && !rin) ? b : (int)(int8_t)b;      // (the eltwise epilogue writes s8)
                    if (my_p >= HW) ps[t] = 0;
#pragma unroll
                    for (int m = 1; m < 16; m <<= 1) ps[t] += __shfl_xor(ps[t], m, 64);
                }
                if (frow == 0) pool_part[(wave * NT + j) * 4 + fq] = v4i{ps[0], ps[1], ps[2], ps[3]};
            }
        }
        if (pool_t >= 0) {
            __syncthreads();
            if (tid < NT * 4) {          // (tile j, channel quad fq): the four waves' partial sums, then the pooling op's arithmetic
                const int j = tid >> 2, q4 = tid & 3;
                const int ch = cx.cu * (16 * NT) + 16 * j + q4 * 4;
                v4i tot = pool_part[(0 * NT + j) * 4 + q4];
#pragma unroll
                for (int w2 = 1; w2 < 4; ++w2) {
                    const v4i t2 = pool_part[(w2 * NT + j) * 4 + q4];
                    tot.x += t2.x; tot.y += t2.y; tot.z += t2.z; tot.w += t2.w;
                }
                const bool u8o = php->out_u8 && !rin;
                unsigned w = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float f = rintf(__fmul_rn((float)tot[t], php->pool_idiv));
                    const int q = u8o ? (int)fminf(fmaxf(f, 0.f), 255.f) : (int)fminf(fmaxf(f, -128.f), 127.f);
                    w |= (unsigned)(q & 0xff) << (8 * t);
                }
                if (ch < cout) *(unsigned*)((char*)a.t[pool_t] + (size_t)img * cout + ch) = w;
            }
        }
    }
    STAGE_TR(7);
#undef STAGE_TR
    return true;
}

}  // namespace

__global__ __launch_bounds__(256) void stage_xcd_kernel(const StageKArgs a) {
    extern __shared__ v4i stage_lds[];
    __shared__ unsigned s_x, s_idx, s_abort;
    __shared__ unsigned long long s_gen;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const unsigned x = stage_xcc_id();
        const unsigned long long me = __hip_atomic_fetch_add(a.sync + 16 * x, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_x = x;
        s_idx = (unsigned)(me & 31ull);
        s_gen = me >> 5;
        s_abort = 0u;
    }
    __syncthreads();
    StageCtx cx;
    cx.xcd = (int)s_x;
    cx.cu = (int)s_idx;
    cx.gen = s_gen;
    cx.ctr = a.sync + 16 * (8 + cx.xcd);
    cx.abort_w = a.sync + 16 * 16;
    cx.bar = 0;
    cx.hw = a.H * a.W;
    cx.img0 = cx.xcd;
    cx.img_step = 8;
    if (cx.xcd >= a.n_img) {      // no image for this XCD: keep its arrival counter in step with the generations and leave
        if (tid == 0 && a.n_barriers) __hip_atomic_fetch_add(cx.ctr, (unsigned long long)a.n_barriers, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int frow = tid & 15;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int p = m * 16 + frow;
        cx.py[m] = p < cx.hw ? p / a.W : -1;
        cx.px[m] = p < cx.hw ? p - (p / a.W) * a.W : 0;
    }
    PhasePtr tab = (PhasePtr)a.phases;
    for (int p = 0; p < a.n_phases; ++p) {
        PhasePtr php = tab + p;
        cx.tr = a.trace ? a.trace + ((size_t)blockIdx.x * a.n_phases + p) * 8 : nullptr;
        bool ok = false;
        switch (php->type) {
        case 0: ok = stage_phase<4, 4, false>(a, php, cx, stage_lds, &s_abort); break;    // 1024 -> 2048 (res5a branch1)
        case 1: ok = stage_phase<1, 4, false>(a, php, cx, stage_lds, &s_abort); break;    // 1024 -> 512
        case 2: ok = stage_phase<1, 18, true>(a, php, cx, stage_lds, &s_abort); break;    // 3x3 512 -> 512
        case 3: ok = stage_phase<4, 2, false>(a, php, cx, stage_lds, &s_abort); break;    // 512 -> 2048
        case 4: ok = stage_phase<1, 8, false>(a, php, cx, stage_lds, &s_abort); break;    // 2048 -> 512
        default: break;
        }
        if (!ok) return;
    }
}

// One convolution, ordinary grid: workgroup b = (image b / 32, channel group b % 32). No registration, no barrier: the phase's
// `barrier` is 0 and the counters are never touched.
__global__ __launch_bounds__(256) void img_conv_kernel(const StageKArgs a) {
    extern __shared__ v4i stage_lds[];
    __shared__ unsigned s_abort;
    StageCtx cx;
    cx.xcd = 0;
    cx.cu = (int)(blockIdx.x & 31u);
    cx.gen = 0;
    cx.ctr = nullptr;
    cx.abort_w = nullptr;
    cx.bar = 0;
    cx.hw = a.H * a.W;
    cx.img0 = (int)(blockIdx.x >> 5);
    cx.img_step = 1 << 30;
    cx.tr = nullptr;
    const int frow = threadIdx.x & 15;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int p = m * 16 + frow;
        cx.py[m] = p < cx.hw ? p / a.W : -1;
        cx.px[m] = p < cx.hw ? p - (p / a.W) * a.W : 0;
    }
    PhasePtr php = (PhasePtr)a.phases;
    switch (php->type) {
    case 0: (void)stage_phase<4, 4, false>(a, php, cx, stage_lds, &s_abort); break;
    case 1: (void)stage_phase<1, 4, false>(a, php, cx, stage_lds, &s_abort); break;
    case 2: (void)stage_phase<1, 18, true>(a, php, cx, stage_lds, &s_abort); break;
    case 3: (void)stage_phase<4, 2, false>(a, php, cx, stage_lds, &s_abort); break;
    case 4: (void)stage_phase<1, 8, false>(a, php, cx, stage_lds, &s_abort); break;
    default: break;
    }
}

bool stage_xcd_type(int tiles_per_cu, int ksteps_per_wave, int is3x3, int* type) {
    static const int tab[5][3] = {{4, 4, 0}, {1, 4, 0}, {1, 18, 1}, {4, 2, 0}, {1, 8, 0}};
    for (int i = 0; i < 5; ++i)
        if (tab[i][0] == tiles_per_cu && tab[i][1] == ksteps_per_wave && tab[i][2] == is3x3) {
            *type = i;
            return true;
        }
    return false;
}

hipError_t launch_stage_xcd(const StageKArgs& a, size_t lds_bytes, hipStream_t s) {
    if (lds_bytes > 160 * 1024 - 64 || lds_bytes <= 80 * 1024) return hipErrorInvalidValue;   // exactly one workgroup per CU
    static bool done[64] = {false};      // per device; racing threads both set the same attribute
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) return e != hipSuccess ? e : hipErrorInvalidDevice;
    if (!done[dev]) {
        e = hipFuncSetAttribute((const void*)stage_xcd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL(stage_xcd_kernel, dim3(256), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_img_conv(const StageKArgs& a, size_t lds_bytes, hipStream_t s) {
    if (lds_bytes > 160 * 1024 - 64 || a.n_img <= 0 || a.n_img > (1 << 20)) return hipErrorInvalidValue;
    static bool done[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) return e != hipSuccess ? e : hipErrorInvalidDevice;
    if (!done[dev]) {
        e = hipFuncSetAttribute((const void*)img_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    hipLaunchKernelGGL(img_conv_kernel, dim3(a.n_img * 32), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

}  // namespace saber_mi355x
