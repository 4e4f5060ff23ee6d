// hip_stub.cpp -- TEST INFRASTRUCTURE ONLY (tests/test_sanitizers.py).  A stand-in for the HIP runtime so that the HOST code of
// libkws_mi355x.so -- the .kwsm parser, the plan builders, the table uploads, the argument checks of the C ABI -- can run under
// AddressSanitizer + UndefinedBehaviorSanitizer in a container without a GPU (SURVEY.md section 5 asked for a sanitizer run).
// "Device" memory is host heap (so an upload that reads past its source, or a plan table indexed out of range, is an ASan report),
// kernel launches do nothing and report success: NO arithmetic of the product runs here and nothing computes a result -- this is not
// a CPU path of the library, it is never linked into it, and nothing outside tests/ refers to it.
#include <hip/hip_runtime.h>

#include <stdlib.h>
#include <string.h>

extern "C" {
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t *p, int) { memset(p, 0, sizeof(*p)); p->multiProcessorCount = 256; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
// (device variables: the host-only build has their host shadows; no kernel runs here, so the shadow is the variable)
hipError_t hipMemcpyFromSymbol(void *d, const void *sym, size_t n, size_t off, hipMemcpyKind) { memcpy(d, (const char *)sym + off, n); return hipSuccess; }
hipError_t hipMemcpyToSymbol(const void *sym, const void *s, size_t n, size_t off, hipMemcpyKind) { memcpy((char *)const_cast<void *>(sym) + off, s, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "hip_stub"; }
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *s, hipStream_t *st) { *g = dim3(1); *b = dim3(1); *s = 0; *st = nullptr; return hipSuccess; }
void **__hipRegisterFatBinary(const void *) { static void *h; return &h; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
}
