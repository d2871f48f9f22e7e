/* Prototype-only shim for "mkl.h" as included by the reference's packed INT8 / FP32 GEMM sources
 * (saber/funcs/impl/x86/mkl_gemm_int8.h:19; mkl_gemm.cpp, mkl_gemm_int8.cpp, mkl_packed_int8_gemm.cpp).
 * The symbols come from the container's /opt/conda/lib/libmkl_rt.so at link time, except cblas_sgemm_alloc /
 * cblas_sgemm_free, which newer oneMKL no longer exports: oracle/ref_driver.cpp provides them on top of
 * cblas_sgemm_pack_get_size + mkl_malloc (used by the FP32 packed paths of mkl_gemm.cpp / vender_fc.cpp).
 * TEST INFRASTRUCTURE ONLY (oracle/_ref). */
#ifndef ORACLE_SHIM_MKL_H
#define ORACLE_SHIM_MKL_H
#include "mkl_cblas.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef enum { CblasAMatrix = 161, CblasBMatrix = 162 } CBLAS_IDENTIFIER;
typedef enum { CblasPacked = 151 } CBLAS_STORAGE;
size_t cblas_gemm_s8u8s32_pack_get_size(const CBLAS_IDENTIFIER identifier, const MKL_INT M, const MKL_INT N, const MKL_INT K);
void cblas_gemm_s8u8s32_pack(const CBLAS_LAYOUT Layout, const CBLAS_IDENTIFIER identifier, const CBLAS_TRANSPOSE Trans,
                             const MKL_INT M, const MKL_INT N, const MKL_INT K, const void* src, const MKL_INT ld, void* dest);
void cblas_gemm_s8u8s32_compute(const CBLAS_LAYOUT Layout, const MKL_INT TransA, const MKL_INT TransB, const CBLAS_OFFSET offsetc,
                                const MKL_INT M, const MKL_INT N, const MKL_INT K, const float alpha, const void* A,
                                const MKL_INT lda, const MKL_INT8_OR_INT ao, const void* B, const MKL_INT ldb,
                                const MKL_INT8_OR_INT bo, const float beta, int32_t* C, const MKL_INT ldc, const int32_t* co);
size_t cblas_sgemm_pack_get_size(const CBLAS_IDENTIFIER identifier, const MKL_INT M, const MKL_INT N, const MKL_INT K);
float* cblas_sgemm_alloc(const CBLAS_IDENTIFIER identifier, const MKL_INT M, const MKL_INT N, const MKL_INT K);
void cblas_sgemm_free(float* dest);
void cblas_sgemm_pack(const CBLAS_LAYOUT Layout, const CBLAS_IDENTIFIER identifier, const CBLAS_TRANSPOSE Trans,
                      const MKL_INT M, const MKL_INT N, const MKL_INT K, const float alpha, const float* src,
                      const MKL_INT ld, float* dest);
void cblas_sgemm_compute(const CBLAS_LAYOUT Layout, const MKL_INT TransA, const MKL_INT TransB, const MKL_INT M,
                         const MKL_INT N, const MKL_INT K, const float* A, const MKL_INT lda, const float* B,
                         const MKL_INT ldb, const float beta, float* C, const MKL_INT ldc);
void* mkl_malloc(size_t size, int align);
void mkl_free(void* p);
#ifdef __cplusplus
}
#endif
#endif

/* thread control (ref_driver.cpp's ref_set_threads) */
#ifdef __cplusplus
extern "C" {
#endif
void MKL_Set_Num_Threads(int nth);
#define mkl_set_num_threads MKL_Set_Num_Threads
#ifdef __cplusplus
}
#endif
