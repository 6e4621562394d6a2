// TEST INFRASTRUCTURE: a stand-in for the HIP runtime, for the AddressSanitizer / UBSan build of libqsv's HOST side.
//
// The library's sources are compiled with `clang++ -x hip --offload-host-only -fsanitize=address,undefined` (no device
// code at all) and linked against this file instead of libamdhip64: "device" memory is host memory (so every
// hipMemcpy the library issues is bounds-checked by ASan against the allocation it made), kernel launches are counted
// and otherwise ignored, events measure nothing.  What runs for real is everything the C ABI does before and after
// a launch: argument validation, qubit -> bit mapping, enumeration tables, matrix re-indexing, lookup-table
// construction, buffer management.  GPU sanitizers are not available on this pool; this is the CPU half.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdlib>
#include <cstring>

extern "C" unsigned long qsv_stub_launches = 0;
extern "C" unsigned long qsv_stub_bytes_copied = 0;

extern "C" {

hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "stub"; }

hipError_t hipMalloc(void **p, size_t bytes) {
    *p = std::calloc(bytes ? bytes : 1, 1);  // zeroed: kernels never run, reductions read what they would have written
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); qsv_stub_bytes_copied += n; return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { static int token; *s = reinterpret_cast<hipStream_t>(&token); return hipSuccess; }

struct StubEvent { std::chrono::steady_clock::time_point t; };
hipError_t hipEventCreate(hipEvent_t *e) { *e = reinterpret_cast<hipEvent_t>(new StubEvent); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<StubEvent *>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { reinterpret_cast<StubEvent *>(e)->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(reinterpret_cast<StubEvent *>(b)->t - reinterpret_cast<StubEvent *>(a)->t).count();
    return hipSuccess;
}

// what the host stubs of __global__ functions call
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *s, hipStream_t *st) {
    *g = dim3(1); *b = dim3(1); *s = 0; *st = nullptr;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t) { ++qsv_stub_launches; return hipSuccess; }
void **__hipRegisterFatBinary(const void *) { static void *handle = nullptr; return &handle; }
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void **) {}

}  // extern "C"
