// saber/funcs/impl/mi355x/saber_gemm.h — Gemm<MI355X, SABER_IMPL | VENDER_IMPL, float, float> (saber/funcs/gemm.h:27-66;
// pattern: the NV specialisation saber/funcs/impl/cuda/saber_gemm.h): row-major C = alpha * op(A) * op(B) + beta * C on
// raw device pointers, enqueued on the context's compute stream.
#ifndef ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_GEMM_H
#define ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_GEMM_H

#include "saber/funcs/gemm.h"
#include "saber_mi355x_adaptor.h"

namespace anakin {
namespace saber {

template <>
class Gemm<MI355X, SABER_IMPL, float, float> : public MatrixFunc<MI355X, float, float> {
public:
    SaberStatus init(const bool trans_a, const bool trans_b, const int m, const int n, const int k, Context<MI355X> ctx) {
        return _g.init(trans_a, trans_b, m, n, k, ctx);
    }
    SaberStatus dispatch(const float alpha, const float beta, const float* a, const float* b, float* c) {
        return _g.dispatch(alpha, beta, a, b, c);
    }

private:
    SaberGemmMI355X<MI355X> _g;
};
template <>
class Gemm<MI355X, VENDER_IMPL, float, float> : public Gemm<MI355X, SABER_IMPL, float, float> {};

}  // namespace saber
}  // namespace anakin
#endif
