// scripts/probe/bf16_split_probe.hip — is FP32 convolution on split-bf16 MFMA worth building? (round-3 review item 5)
//
// x = h + m + l with h, m, l bf16 (3 x 8 mantissa bits = the 24 of an f32); a . b ~= hh + hm + mh + hl + lh + mm (the three
// dropped products are <= 2^-32 relative), six v_mfma_f32_16x16x32_bf16 per 32-deep slab accumulated in f32, against
// eight v_mfma_f32_16x16x4_f32 for the same slab. Peak ratio 16 : 1 per instruction-flop, so 16 / 6 = 2.7x at best.
// This probe measures, on one workgroup per CU (4 waves, each a 64 x 64 output tile = 16 accumulators, operands from LDS
// with ds_read_b128 like the product kernel):
//   (1) layout of v_mfma_f32_16x16x32_bf16 (A / B: lane l holds row l & 15, k = (l >> 4) * 8 .. + 7);
//   (2) accuracy: |split - exact| / |exact| against an f64 dot product over K = 2304 (a VGG 3x3 x 256 reduction), next to
//       the f32 MFMA's own error;
//   (3) throughput of the inner loop: f32 MFMA vs 6 x bf16 MFMA with pre-split operands in LDS (3 planes), and with the B
//       operand split on the fly from f32 while it is staged into LDS (VALU cost of the split).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probe/bf16_split_probe.hip -o scripts/probe/bf16_split_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef short v8s __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rne(float x) {       // round to nearest even, as v_cvt_pk_bf16_f32
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = bf16_rne(x);
    const float r1 = x - bf16_f(h);          // exact
    m = bf16_rne(r1);
    const float r2 = r1 - bf16_f(m);         // exact
    l = bf16_rne(r2);
}

// ---- (1) + (2): one 16 x 16 output, K deep, A [16][K], B [16][K] (B^T), f32 in memory --------------------------------
__global__ void acc_kernel(const float* A, const float* B, int K, float* D_f32, float* D_split) {
    const int l = threadIdx.x, row = l & 15, kg = l >> 4;
    v4f acc = {0, 0, 0, 0};
    for (int k = 0; k < K; k += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[row * K + k + kg], B[row * K + k + kg], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D_f32[(kg * 4 + r) * 16 + row] = acc[r];
    v4f s = {0, 0, 0, 0};
    for (int k = 0; k < K; k += 32) {
        v8s ah, am, al, bh, bm, bl;
        for (int t = 0; t < 8; ++t) {
            unsigned short h, m, lo;
            split3(A[row * K + k + kg * 8 + t], h, m, lo);
            ah[t] = (short)h; am[t] = (short)m; al[t] = (short)lo;
            split3(B[row * K + k + kg * 8 + t], h, m, lo);
            bh[t] = (short)h; bm[t] = (short)m; bl[t] = (short)lo;
        }
#define MF(a, b) s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), s, 0, 0, 0)
        MF(al, bh); MF(ah, bl); MF(am, bm); MF(am, bh); MF(ah, bm); MF(ah, bh);      // small terms first
#undef MF
    }
    for (int r = 0; r < 4; ++r) D_split[(kg * 4 + r) * 16 + row] = s[r];
}

// ---- (3) throughput: operands in LDS, each wave a 64 x 64 tile, `iters` slabs of K = 32 -------------------------------
template <int MODE>   // 0: f32 MFMA; 1: bf16 x 3 pre-split in LDS; 2: as 1, plus the B split recomputed per slab from f32 registers
__global__ __launch_bounds__(256) void tput_kernel(const float* src, float* out, int iters) {
    extern __shared__ __align__(16) unsigned char lds[];
    // LDS image per wave-independent operand fragment: [frag 0..3][plane][lane] 16 bytes
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint4* L = (uint4*)lds;
    for (int i = threadIdx.x; i < 2 * 4 * 3 * 64 * 2; i += 256) {       // A and B, 4 frags, 3 planes (f32: 2 chunks of 16 B per K = 32... )
        const float f = src[i & 1023];
        L[i] = make_uint4(__float_as_uint(f), __float_as_uint(f * 0.5f), __float_as_uint(f * 0.25f), __float_as_uint(f * 2.f));
    }
    __syncthreads();
    v4f acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0, 0, 0, 0};
    const uint4* LA = L + (size_t)w * 0;               // all waves read the same image: bank behaviour as in the product kernel
    const uint4* LB = L + 4 * 3 * 64 * 2;
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            // K = 32 = 8 k-steps of 4: two 16-byte chunks per fragment
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                uint4 a[4], b[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) { a[f] = LA[(f * 2 + half) * 64 + l]; b[f] = LB[(f * 2 + half) * 64 + l]; }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(((const unsigned*)&a[i])[t]),
                                                                            __uint_as_float(((const unsigned*)&b[j])[t]), acc[i][j], 0, 0, 0);
            }
        } else {
            uint4 a[4][3], b[4][3];
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int p = 0; p < 3; ++p) { a[f][p] = LA[(f * 3 + p) * 64 + l]; b[f][p] = LB[(f * 3 + p) * 64 + l]; }
            if constexpr (MODE == 2) {
                // the staging-side split of one lane's share of the B tile for this slab: 16 f32 -> 3 x 16 bf16
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    unsigned short h, m, lo;
                    split3(__uint_as_float(((const unsigned*)&b[e & 3][0])[e >> 2]) + (float)it, h, m, lo);
                    keep += bf16_f(h) + bf16_f(m) + bf16_f(lo);
                }
            }
#define MF(x, y) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, x), __builtin_bit_cast(v8bf, y), acc[i][j], 0, 0, 0)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    MF(a[i][2], b[j][0]); MF(a[i][0], b[j][2]); MF(a[i][1], b[j][1]);
                    MF(a[i][1], b[j][0]); MF(a[i][0], b[j][1]); MF(a[i][0], b[j][0]);
                }
#undef MF
        }
    }
    float s = keep;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
    // (1) + (2)
    const int K = 2304;
    std::vector<float> hA(16 * K), hB(16 * K);
    srand(7);
    for (auto& v : hA) v = (float)((rand() / (double)RAND_MAX) * 2 - 1) * 0.06f;          // He-scaled weights
    for (auto& v : hB) v = (float)((rand() / (double)RAND_MAX)) * 3.f;                      // relu'd activations
    float *dA, *dB, *d1, *d2;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&d1, 1024)); CK(hipMalloc(&d2, 1024));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dA, dB, K, d1, d2);
    float h1[256], h2[256];
    CK(hipMemcpy(h1, d1, 1024, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2, d2, 1024, hipMemcpyDeviceToHost));
    double e1 = 0, e2 = 0, mx = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)hA[i * K + k] * (double)hB[j * K + k];
            mx = fmax(mx, fabs(s));
            e1 = fmax(e1, fabs(h1[i * 16 + j] - s));
            e2 = fmax(e2, fabs(h2[i * 16 + j] - s));
        }
    printf("accuracy over K = %d (max |err| / max |exact|):  f32 MFMA %.3e   split-bf16 (6 products) %.3e\n", K, e1 / mx, e2 / mx);
    // (3)
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 2000;
    float* dsrc; float* dout;
    CK(hipMalloc(&dsrc, 4096)); CK(hipMalloc(&dout, 4096));
    CK(hipMemcpy(dsrc, hA.data(), 4096, hipMemcpyHostToDevice));
    const size_t lds_bytes = 2 * 4 * 3 * 64 * 2 * 16;
    hipEvent_t e0, e1v;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1v));
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(tput_kernel<0>, dim3(cus), dim3(256), lds_bytes, 0, dsrc, dout, iters);
            if (mode == 1) hipLaunchKernelGGL(tput_kernel<1>, dim3(cus), dim3(256), lds_bytes, 0, dsrc, dout, iters);
            if (mode == 2) hipLaunchKernelGGL(tput_kernel<2>, dim3(cus), dim3(256), lds_bytes, 0, dsrc, dout, iters);
            CK(hipEventRecord(e1v, 0));
            CK(hipEventSynchronize(e1v));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1v));
            if (rep) best = fminf(best, ms);
        }
        const double flop = 2.0 * 64 * 64 * 32 * (double)iters * 4 * cus;          // useful f32-equivalent flops
        printf("%-58s %8.3f ms  %7.1f TFLOP/s f32-equivalent (%.2f of the 157.3 f32 MFMA peak)\n",
               mode == 0 ? "f32 MFMA 16x16x4, operands from LDS" : (mode == 1 ? "6 x bf16 MFMA 16x16x32, 3 pre-split planes from LDS"
                                                                              : "  ... + on-the-fly 3-way split of the B share per slab"),
               best, flop / (best * 1e-3) / 1e12, flop / (best * 1e-3) / 1e12 / 157.3);
    }
    return 0;
}
