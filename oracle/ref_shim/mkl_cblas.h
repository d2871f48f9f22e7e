/* Prototype-only shim for the MKL CBLAS entry points the reference's x86 Saber
 * GEMM-path sources call (gemm_x8s8s32x_conv.cpp:244-252, saber_conv_1x1.cpp; round 6: vender_gemm.cpp's cblas_sgemv,
 * winograd_float.cpp / winograd_avx2.cpp's cblas_sgemm_batch).
 * The symbols come from the container's /opt/conda/lib/libmkl_rt.so at link time.
 * TEST INFRASTRUCTURE ONLY (oracle/_ref). */
#ifndef ORACLE_SHIM_MKL_CBLAS_H
#define ORACLE_SHIM_MKL_CBLAS_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef int MKL_INT;
typedef int8_t MKL_INT8_OR_INT;   /* ao / bo of the s8u8s32 routines */
typedef enum { CblasRowMajor = 101, CblasColMajor = 102 } CBLAS_LAYOUT;
typedef enum { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 } CBLAS_TRANSPOSE;
typedef enum { CblasRowOffset = 171, CblasColOffset = 172, CblasFixOffset = 173 } CBLAS_OFFSET;
void cblas_gemm_s8u8s32(const CBLAS_LAYOUT Layout, const CBLAS_TRANSPOSE TransA,
                        const CBLAS_TRANSPOSE TransB, const CBLAS_OFFSET OffsetC,
                        const MKL_INT M, const MKL_INT N, const MKL_INT K, const float alpha,
                        const void* A, const MKL_INT lda, const int8_t ao,
                        const void* B, const MKL_INT ldb, const int8_t bo, const float beta,
                        int32_t* C, const MKL_INT ldc, const int32_t* cb);
void cblas_sgemm(const CBLAS_LAYOUT Layout, const CBLAS_TRANSPOSE TransA,
                 const CBLAS_TRANSPOSE TransB, const MKL_INT M, const MKL_INT N, const MKL_INT K,
                 const float alpha, const float* A, const MKL_INT lda, const float* B,
                 const MKL_INT ldb, const float beta, float* C, const MKL_INT ldc);
void cblas_sgemv(const CBLAS_LAYOUT Layout, const CBLAS_TRANSPOSE TransA, const MKL_INT M, const MKL_INT N, const float alpha,
                 const float* A, const MKL_INT lda, const float* X, const MKL_INT incX, const float beta, float* Y,
                 const MKL_INT incY);
void cblas_sgemm_batch(const CBLAS_LAYOUT Layout, const CBLAS_TRANSPOSE* TransA_Array, const CBLAS_TRANSPOSE* TransB_Array,
                       const MKL_INT* M_Array, const MKL_INT* N_Array, const MKL_INT* K_Array, const float* alpha_Array,
                       const float** A_Array, const MKL_INT* lda_Array, const float** B_Array, const MKL_INT* ldb_Array,
                       const float* beta_Array, float** C_Array, const MKL_INT* ldc_Array, const MKL_INT group_count,
                       const MKL_INT* group_size);
void cblas_saxpy(const MKL_INT N, const float alpha, const float* X, const MKL_INT incX, float* Y, const MKL_INT incY);
#ifdef __cplusplus
}
#endif
#endif
