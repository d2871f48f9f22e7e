// saber/funcs/impl/mi355x/mi355x_timer.h — SaberTimer<MI355X> on hipEvents (pattern: SaberTimer<NV>,
// saber/funcs/timer.h:95-165): start / end record events on the context's compute stream, get_average_ms etc. read
// hipEventElapsedTime. Included at the end of saber/funcs/timer.h when USE_MI355X_PLACE is defined.
#ifndef ANAKIN_SABER_FUNCS_IMPL_MI355X_TIMER_H
#define ANAKIN_SABER_FUNCS_IMPL_MI355X_TIMER_H

#include <hip/hip_runtime_api.h>
#include <list>

namespace anakin {
namespace saber {

template <>
class SaberTimer<MI355X> final {
public:
    SaberTimer() {
        MI355X_CHECK(hipEventCreate(&_e0));
        MI355X_CHECK(hipEventCreate(&_e1));
    }
    ~SaberTimer() {
        (void)hipEventDestroy(_e0);
        (void)hipEventDestroy(_e1);
    }
    void clear() { _ms.clear(); }
    void start(Context<MI355X>& ctx) { MI355X_CHECK(hipEventRecord(_e0, ctx.get_compute_stream())); }
    void end(Context<MI355X>& ctx) {
        MI355X_CHECK(hipEventRecord(_e1, ctx.get_compute_stream()));
        MI355X_CHECK(hipEventSynchronize(_e1));
        float ms = 0.f;
        MI355X_CHECK(hipEventElapsedTime(&ms, _e0, _e1));
        _ms.push_back(ms);
    }
    float get_average_ms() {
        if (_ms.empty()) return 0.f;
        float s = 0.f;
        for (float v : _ms) s += v;
        return s / _ms.size();
    }
    float get_best_ms() {
        float b = 0.f;
        bool first = true;
        for (float v : _ms) {
            if (first || v < b) b = v;
            first = false;
        }
        return b;
    }
    const std::list<float> get_time_stat() { return _ms; }

private:
    hipEvent_t _e0, _e1;
    std::list<float> _ms;
};

}  // namespace saber
}  // namespace anakin
#endif
