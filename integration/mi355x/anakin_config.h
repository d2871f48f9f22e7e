/* Hand-written replacement for the cmake-generated anakin_config.h (reference template: cmake/config/anakin_config.h.in)
 * for the MI355X integration build (integration/build_mi355x_test.sh): the X86 host target + the MI355X device target.
 * TEST INFRASTRUCTURE ONLY. */
#ifndef ANAKIN_CONFIG_H
#define ANAKIN_CONFIG_H
#define ANAKIN_TYPE_FP32
#define USE_OPENMP
#define USE_LOGGER
#define USE_X86_PLACE
#define USE_MI355X_PLACE
#define BUILD_X86_ARCH "native"
#define PLATFORM_POSIX
#define PLATFORM_X86
#define ANAKIN_VERSION 1
#endif
