/* Hand-written replacement for the cmake-generated anakin_config.h
 * (reference template: cmake/config/anakin_config.h.in). TEST INFRASTRUCTURE ONLY:
 * lets oracle/Makefile compile a few unmodified reference TUs where they lie
 * under /root/reference into oracle/_ref/. Nothing in the product path includes this. */
#ifndef ANAKIN_CONFIG_H
#define ANAKIN_CONFIG_H
#define ANAKIN_TYPE_FP32
#define USE_OPENMP
#define USE_LOGGER
#define USE_X86_PLACE
#define BUILD_X86_ARCH "native"
#define PLATFORM_POSIX
#define PLATFORM_X86
#define ANAKIN_VERSION 1
#endif
