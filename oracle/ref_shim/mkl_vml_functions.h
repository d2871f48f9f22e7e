/* Empty shim: saber/funcs/impl/x86/vender_fc.cpp:4 includes "mkl_vml_functions.h" but calls no VML routine.
 * TEST INFRASTRUCTURE ONLY (oracle/_ref). */
#ifndef ORACLE_SHIM_MKL_VML_FUNCTIONS_H
#define ORACLE_SHIM_MKL_VML_FUNCTIONS_H
#endif
