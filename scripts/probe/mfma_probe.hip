// Empirical check of the MFMA fragment layouts this repo relies on (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k_f32(const float* A, const float* B, float* D) {  // A[16][4], B[4][16], D[16][16]
    int l = threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
__global__ void k_i8(const signed char* A, const signed char* B, int* D) {  // A[16][64], B[16][64] (B^T), D[16][16]
    int l = threadIdx.x;
    v4i a = *(const v4i*)(A + (l & 15) * 64 + (l >> 4) * 16);
    v4i b = *(const v4i*)(B + (l & 15) * 64 + (l >> 4) * 16);
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}
int main() {
    float hA[64], hB[64], hD[256];
    for (int i = 0; i < 64; ++i) { hA[i] = (float)(rand() % 7 - 3); hB[i] = (float)(rand() % 5 - 2); }
    float *dA, *dB, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    k_f32<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j];
        if (s != hD[i * 16 + j]) ++bad;
    }
    printf("f32 16x16x4: %d mismatches\n", bad);
    signed char iA[1024], iB[1024]; int iD[256];
    for (int i = 0; i < 1024; ++i) { iA[i] = rand() % 255 - 127; iB[i] = rand() % 255 - 127; }
    signed char *eA, *eB; int* eD;
    hipMalloc(&eA, 1024); hipMalloc(&eB, 1024); hipMalloc(&eD, 1024);
    hipMemcpy(eA, iA, 1024, hipMemcpyHostToDevice); hipMemcpy(eB, iB, 1024, hipMemcpyHostToDevice);
    k_i8<<<1, 64>>>(eA, eB, eD);
    hipMemcpy(iD, eD, 1024, hipMemcpyDeviceToHost);
    bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        int s = 0; for (int k = 0; k < 64; ++k) s += (int)iA[i * 64 + k] * (int)iB[j * 64 + k];
        if (s != iD[i * 16 + j]) ++bad;
    }
    printf("i8 16x16x64: %d mismatches\n", bad);
    return 0;
}
