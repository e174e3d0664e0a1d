// A stand-in for the HIP runtime in the sanitizer build of the library's host logic (make -C crfp_amd/csrc asan): every call "succeeds" and does
// nothing, kernels are never run.  The library's host code (argument checks, Layout arenas, launch-argument tables, fork / join of its side
// stream) then walks complete engine calls under AddressSanitizer + UBSan with device pointers that are never dereferenced.  Test scaffolding
// for the CPU build container: not part of the product, never linked into libcrfp_hip.so.
#include <hip/hip_runtime_api.h>
#include <cstddef>

static long g_launches = 0, g_streams = 0, g_events = 0;
extern "C" long crfp_stub_launches(void) { return g_launches; }

extern "C" {
void** __hipRegisterFatBinary(const void*) { static void* h = nullptr; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
struct StubCfg { dim3 g, b; size_t shmem; hipStream_t s; };
static thread_local StubCfg t_cfg;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t shmem, hipStream_t s) { t_cfg = {g, b, shmem, s}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* shmem, hipStream_t* s) { *g = t_cfg.g; *b = t_cfg.b; *shmem = t_cfg.shmem; *s = t_cfg.s; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3 g, dim3 b, void**, size_t, hipStream_t) {
    ++g_launches;
    // what the real runtime would refuse: an empty or oversized grid / block
    if (g.x == 0 || g.y == 0 || g.z == 0 || b.x * b.y * b.z == 0 || b.x * b.y * b.z > 1024 || g.y > 65535 || g.z > 65535) return hipErrorInvalidConfiguration;
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub HIP runtime"; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipMemsetAsync(void*, int, size_t, hipStream_t) { return hipSuccess; }
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(0x1000 + 16 * ++g_streams); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { --g_streams; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(0x100000 + 16 * ++g_events); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t) { --g_events; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
}
