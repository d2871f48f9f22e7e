/* Prototype-only shim for "mkl_trans.h" as included by the reference's FP32 Winograd sources
 * (saber/funcs/impl/x86/winograd_float.cpp:3,539,571; winograd_avx2.cpp:3,598,630): the one routine they call.
 * The symbol comes from the container's /opt/conda/lib/libmkl_rt.so at link time.
 * TEST INFRASTRUCTURE ONLY (oracle/_ref). */
#ifndef ORACLE_SHIM_MKL_TRANS_H
#define ORACLE_SHIM_MKL_TRANS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
void MKL_Somatcopy(char ordering, char trans, size_t rows, size_t cols, const float alpha, const float* A, size_t lda,
                   float* B, size_t ldb);
#ifdef __cplusplus
}
#endif
#endif
