// saber/core/impl/mi355x/mi355x_target_wrapper.h — TargetWrapper<MI355X, __device_target> on the HIP runtime.
//
// The device half of docs/Manual/addCustomDevice.md:59-280 for the MI355X Saber target: memory, streams, events and
// copies expressed with hipMalloc / hipMemcpyAsync / hipStream_t / hipEvent_t (the member list is the one Buffer,
// Tensor, Env, Context and SaberTimer use; pattern: the NV specialisation, saber/core/target_wrapper.h:303-396).
// Included at the end of saber/core/target_wrapper.h when USE_MI355X_PLACE is defined (integration/apply_mi355x_target.py).
// New code of this repository; nothing here is copied from the reference's CUDA / OpenCL wrappers.
#ifndef ANAKIN_SABER_CORE_IMPL_MI355X_TARGET_WRAPPER_H
#define ANAKIN_SABER_CORE_IMPL_MI355X_TARGET_WRAPPER_H

#include <hip/hip_runtime_api.h>

namespace anakin {
namespace saber {

#define MI355X_CHECK(expr)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        CHECK_EQ((int)e_, (int)hipSuccess) << "MI355X target: " << #expr << ": " << hipGetErrorString(e_); \
    } while (0)

template <>
struct TargetWrapper<MI355X, __device_target> {
    typedef hipEvent_t event_t;
    typedef hipStream_t stream_t;

    static void get_device_count(int& count);
    static void set_device(int id);
    static int get_device_id();
    static void device_sync();

    static void mem_alloc(void** ptr, size_t n);
    static void mem_free(void* ptr);
    static void mem_set(void* ptr, int value, size_t n);

    static void create_event(event_t* event, bool flag = false);
    static void destroy_event(event_t event);
    static void record_event(event_t event, stream_t stream);
    static void query_event(event_t event);
    static void sync_event(event_t event);

    static void create_stream(stream_t* stream);
    static void create_stream_with_flag(stream_t* stream, unsigned int flag);   // 1: non-blocking
    static void create_stream_with_priority(stream_t* stream, unsigned int flag, int priority);
    static void destroy_stream(stream_t stream);
    static void sync_stream(event_t event, stream_t stream);   // stream waits for event
    static void sync_stream(stream_t stream);
    // additions to the reference's member list: a synchronous device-to-host copy drains every stream made through create_stream*
    // on the calling thread's device - except one whose owner declares here that it synchronises its own results before it hands them
    // out (a plan's stream); known_streams: how many streams of `dev` that drain covers (tests)
    static void owner_syncs_stream(stream_t stream);
    static int known_streams(int dev);
    // record_event on a non-timing event is lazy (mi355x_impl.cpp): how many records were only noted / how many became a hipEventRecord
    static void lazy_event_stats(long long* noted, long long* flushed);

    static void sync_memcpy(void* dst, size_t dst_offset, int dst_id, const void* src, size_t src_offset, int src_id,
                            size_t count, __DtoD);
    static void async_memcpy(void* dst, size_t dst_offset, int dst_id, const void* src, size_t src_offset, int src_id,
                             size_t count, stream_t stream, __DtoD);
    static void sync_memcpy(void* dst, size_t dst_offset, int dst_id, const void* src, size_t src_offset, int src_id,
                            size_t count, __HtoD);
    static void async_memcpy(void* dst, size_t dst_offset, int dst_id, const void* src, size_t src_offset, int src_id,
                             size_t count, stream_t stream, __HtoD);
    static void sync_memcpy(void* dst, size_t dst_offset, int dst_id, const void* src, size_t src_offset, int src_id,
                            size_t count, __DtoH);
    static void async_memcpy(void* dst, size_t dst_offset, int dst_id, const void* src, size_t src_offset, int src_id,
                             size_t count, stream_t stream, __DtoH);
    static void sync_memcpy_p2p(void* dst, size_t dst_offset, int dst_id, const void* src, size_t src_offset, int src_id,
                                size_t count);
    static void async_memcpy_p2p(void* dst, size_t dst_offset, int dst_id, const void* src, size_t src_offset, int src_id,
                                 size_t count, stream_t stream);
};

}  // namespace saber
}  // namespace anakin
#endif
