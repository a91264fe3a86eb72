// Minimal host-memory stand-in for the HIP runtime entry points that the HOST half of libbspy_amd.so calls
// (table upload, workspaces, pinned staging, copy-thread pipeline, streams/events).  TEST INFRASTRUCTURE: it
// lets tests/test_host_logic.py run bsk_api.hip's host code under AddressSanitizer and ThreadSanitizer on the
// CPU box (GPU sanitizer runs are not available on the GPU pool).  HIPSTUB_DEVICES sets the number of fake devices
// (the multi-device entry points, bsk_multi.hip, run on 2 - 8 of them).  "Device" memory is host memory, copies are
// memcpy, streams execute immediately, kernel launches are no-ops (results are whatever calloc left: zeros).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>

extern "C" {

static std::atomic<long> g_live{0}, g_launches{0};
long hipstub_live_allocations() { return g_live.load(); }
long hipstub_kernel_launches() { return g_launches.load(); }

// HIPSTUB_DEVICES fake devices (default 1): the current device is per thread, as in the HIP runtime
static int stub_devices()
{
    static const int n = [] { const char *e = getenv("HIPSTUB_DEVICES"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 64 ? v : 1; }();
    return n;
}
static thread_local int t_device = 0;
static std::atomic<long> g_set_device{0};
long hipstub_set_device_calls() { return g_set_device.load(); }
hipError_t hipGetDeviceCount(int *n) { *n = stub_devices(); return hipSuccess; }
hipError_t hipSetDevice(int d)
{
    if (d < 0 || d >= stub_devices()) return hipErrorInvalidDevice;
    t_device = d;
    ++g_set_device;
    return hipSuccess;
}
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t *p, int)
{
    memset(p, 0, sizeof(*p));
    p->multiProcessorCount = 256;
    p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    return hipSuccess;
}
const char *hipGetErrorString(hipError_t) { return "hipstub error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); ++g_live; return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { if (p) { free(p); --g_live; } return hipSuccess; }
// pinned memory starts as all ones: the small-call path reads its out-of-domain record from there after a
// (here: no-op) kernel published it, and all ones is "no offender"
hipError_t hipHostMalloc(void **p, size_t n, unsigned)
{
    *p = malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    memset(*p, 0xff, n ? n : 1);
    ++g_live;
    return hipSuccess;
}
hipError_t hipHostFree(void *p) { if (p) { free(p); --g_live; } return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(calloc(1, 8)); ++g_live; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { if (s) { free(s); --g_live; } return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = reinterpret_cast<hipEvent_t>(calloc(1, 8)); ++g_live; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { if (e) { free(e); --g_live; } return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t) { ++g_launches; return hipSuccess; }

// fat-binary registration emitted by hipcc for every translation unit with kernels
void **__hipRegisterFatBinary(const void *) { static void *h; return &h; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *s, hipStream_t *st) { *g = dim3(1); *b = dim3(1); *s = 0; *st = nullptr; return hipSuccess; }
}
