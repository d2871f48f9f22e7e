// integration/mock_hip/mock_hip.cpp -> libmock_hip.so — a malloc-backed stand-in for the handful of HIP runtime entry
// points the MI355X Saber target touches while a Net is being BUILT (TargetWrapper<MI355X>: memory, streams, events;
// libsaber_mi355x.so: weight repacking into device buffers). LD_PRELOADed in front of libamdhip64 it lets
// `test_net_mi355x.bin ... dry` run Graph::Optimize() and Net<MI355X>::init() on a machine WITHOUT a GPU (the build
// container, the `-m "not gpu"` test tier) and dump the op list the reference's optimiser produced. Kernel launches
// are swallowed, so nothing is ever computed through it: it cannot stand in for the device in any parity or timing
// test, and nothing in anakin_amd/ knows about it. TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>

extern "C" {

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
    memset(p, 0, sizeof *p);
    strcpy(p->name, "mock MI355X (no device)");
    strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)288 << 30;
    p->sharedMemPerBlock = 160 << 10;
    p->maxSharedMemoryPerMultiProcessor = 160 << 10;
    p->warpSize = 64;
    p->clockRate = 2400000;
    p->memoryClockRate = 2000000;
    p->memoryBusWidth = 8192;
    p->l2CacheSize = 4 << 20;
    p->major = 9;
    p->minor = 5;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipPeekAtLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "mock hip"; }

hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeer(void* d, int, const void* s, int, size_t n) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { memmove(d, s, n); return hipSuccess; }

hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t*, const void*) { return hipErrorInvalidValue; }      // everything is "pageable"
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }

hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)calloc(1, 8); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)calloc(1, 8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// stream capture / graphs: not available on the mock (callers fall back to eager launches)
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }

// kernel launches: swallowed (see the header)
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* shm, hipStream_t* s) {
    *g = dim3(1); *b = dim3(1); *shm = 0; *s = nullptr;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchCooperativeKernel(const void*, dim3, dim3, void**, unsigned int, hipStream_t) { return hipSuccess; }
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }

}  // extern "C"
