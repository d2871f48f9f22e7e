// saber/core/impl/mi355x/mi355x_impl.cpp — definitions of TargetWrapper<MI355X>, Device<MI355X> and the explicit
// instantiations of the core templates for the MI355X target (what saber/core/impl/x86/x86_impl.cpp + x86_device.cpp
// are for X86). HIP runtime API only: compiled by the host compiler, no device code.
#include "core/tensor.h"
#include "core/env.h"

#ifdef USE_MI355X_PLACE
namespace anakin {
namespace saber {

typedef TargetWrapper<MI355X, __device_target> MI355X_API;

void MI355X_API::get_device_count(int& count) { MI355X_CHECK(hipGetDeviceCount(&count)); }
void MI355X_API::set_device(int id) { MI355X_CHECK(hipSetDevice(id)); }
int MI355X_API::get_device_id() {
    int id = 0;
    MI355X_CHECK(hipGetDevice(&id));
    return id;
}
void MI355X_API::device_sync() { MI355X_CHECK(hipDeviceSynchronize()); }

void MI355X_API::mem_alloc(void** ptr, size_t n) { MI355X_CHECK(hipMalloc(ptr, n ? n : 1)); }
void MI355X_API::mem_free(void* ptr) {
    if (ptr) MI355X_CHECK(hipFree(ptr));
}
void MI355X_API::mem_set(void* ptr, int value, size_t n) { MI355X_CHECK(hipMemset(ptr, value, n)); }

void MI355X_API::create_event(event_t* event, bool flag) {
    MI355X_CHECK(hipEventCreateWithFlags(event, flag ? hipEventDefault : hipEventDisableTiming));
}
void MI355X_API::destroy_event(event_t event) { MI355X_CHECK(hipEventDestroy(event)); }
void MI355X_API::record_event(event_t event, stream_t stream) { MI355X_CHECK(hipEventRecord(event, stream)); }
void MI355X_API::query_event(event_t event) { (void)hipEventQuery(event); }
void MI355X_API::sync_event(event_t event) { MI355X_CHECK(hipEventSynchronize(event)); }

void MI355X_API::create_stream(stream_t* stream) { MI355X_CHECK(hipStreamCreate(stream)); }
void MI355X_API::create_stream_with_flag(stream_t* stream, unsigned int flag) {
    MI355X_CHECK(hipStreamCreateWithFlags(stream, flag ? hipStreamNonBlocking : hipStreamDefault));
}
void MI355X_API::create_stream_with_priority(stream_t* stream, unsigned int flag, int priority) {
    MI355X_CHECK(hipStreamCreateWithPriority(stream, flag ? hipStreamNonBlocking : hipStreamDefault, priority));
}
void MI355X_API::destroy_stream(stream_t stream) { MI355X_CHECK(hipStreamDestroy(stream)); }
void MI355X_API::sync_stream(event_t event, stream_t stream) { MI355X_CHECK(hipStreamWaitEvent(stream, event, 0)); }
void MI355X_API::sync_stream(stream_t stream) { MI355X_CHECK(hipStreamSynchronize(stream)); }

static inline void mi355x_copy(void* dst, size_t dst_offset, const void* src, size_t src_offset, size_t count,
                               hipMemcpyKind kind, hipStream_t stream, bool async) {
    if (count == 0) return;
    void* d = (char*)dst + dst_offset;
    const void* s = (const char*)src + src_offset;
    if (async) MI355X_CHECK(hipMemcpyAsync(d, s, count, kind, stream));
    else MI355X_CHECK(hipMemcpy(d, s, count, kind));
}
void MI355X_API::sync_memcpy(void* dst, size_t dst_offset, int, const void* src, size_t src_offset, int, size_t count, __DtoD) {
    mi355x_copy(dst, dst_offset, src, src_offset, count, hipMemcpyDeviceToDevice, nullptr, false);
}
void MI355X_API::async_memcpy(void* dst, size_t dst_offset, int, const void* src, size_t src_offset, int, size_t count,
                              stream_t stream, __DtoD) {
    mi355x_copy(dst, dst_offset, src, src_offset, count, hipMemcpyDeviceToDevice, stream, true);
}
void MI355X_API::sync_memcpy(void* dst, size_t dst_offset, int, const void* src, size_t src_offset, int, size_t count, __HtoD) {
    mi355x_copy(dst, dst_offset, src, src_offset, count, hipMemcpyHostToDevice, nullptr, false);
}
void MI355X_API::async_memcpy(void* dst, size_t dst_offset, int, const void* src, size_t src_offset, int, size_t count,
                              stream_t stream, __HtoD) {
    mi355x_copy(dst, dst_offset, src, src_offset, count, hipMemcpyHostToDevice, stream, true);
}
void MI355X_API::sync_memcpy(void* dst, size_t dst_offset, int, const void* src, size_t src_offset, int, size_t count, __DtoH) {
    // The target's compute streams are non-blocking (Device<MI355X>::create_stream): a null-stream hipMemcpy is not ordered after
    // them. A SYNCHRONOUS device-to-host copy is the framework's "give me the tensor now" (Tensor::copy_from into a host tensor:
    // Worker::sync_prediction, EntropyCalibrator::max_data / histgram, which on NV rely on the legacy default stream's implicit
    // ordering): wait for the device first.
    MI355X_CHECK(hipDeviceSynchronize());
    mi355x_copy(dst, dst_offset, src, src_offset, count, hipMemcpyDeviceToHost, nullptr, false);
}
void MI355X_API::async_memcpy(void* dst, size_t dst_offset, int, const void* src, size_t src_offset, int, size_t count,
                              stream_t stream, __DtoH) {
    mi355x_copy(dst, dst_offset, src, src_offset, count, hipMemcpyDeviceToHost, stream, true);
}
void MI355X_API::sync_memcpy_p2p(void* dst, size_t dst_offset, int dst_id, const void* src, size_t src_offset, int src_id,
                                 size_t count) {
    if (count) MI355X_CHECK(hipMemcpyPeer((char*)dst + dst_offset, dst_id, (const char*)src + src_offset, src_id, count));
}
void MI355X_API::async_memcpy_p2p(void* dst, size_t dst_offset, int dst_id, const void* src, size_t src_offset, int src_id,
                                  size_t count, stream_t stream) {
    if (count)
        MI355X_CHECK(hipMemcpyPeerAsync((char*)dst + dst_offset, dst_id, (const char*)src + src_offset, src_id, count, stream));
}

// ---- Device<MI355X>: properties from hipGetDeviceProperties; max_stream data + compute streams (non-blocking) ----
template <>
void Device<MI355X>::create_stream() {
    _data_stream.clear();
    _compute_stream.clear();
    for (int i = 0; i < _max_stream; ++i) {
        MI355X_API::stream_t sd, sc;
        MI355X_API::create_stream_with_flag(&sd, 1);
        MI355X_API::create_stream_with_flag(&sc, 1);
        _data_stream.push_back(sd);
        _compute_stream.push_back(sc);
    }
}
template <>
void Device<MI355X>::get_info() {
    int dev = 0;
    MI355X_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t p;
    MI355X_CHECK(hipGetDeviceProperties(&p, dev));
    _info._idx = dev;
    _info._device_name = p.name;
    _info._compute_ability = p.gcnArchName;          // "gfx950:sramecc+:xnack-"
    _info._compute_core_num = p.multiProcessorCount;   // 256 CUs
    _info._max_frequence = p.clockRate / 1000;         // MHz
    _info._min_frequence = p.memoryClockRate / 1000;
    _info._max_memory = (int)(p.totalGlobalMem / (1024 * 1024));   // MiB (288 GB HBM3E)
    _info._sharemem_size = (int)(p.sharedMemPerBlock / 1024);      // KiB of LDS per workgroup
    _info._L2_cache = p.l2CacheSize / 1024;
    _info._generate_arch = 950;
}
template void Device<MI355X>::get_info();
template void Device<MI355X>::create_stream();

template class Buffer<MI355X>;
template class Tensor<MI355X>;
template struct Env<MI355X>;

}  // namespace saber
}  // namespace anakin
#endif  // USE_MI355X_PLACE
