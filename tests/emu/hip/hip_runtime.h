// tests/emu/hip/hip_runtime.h — TEST-ONLY stand-in for <hip/hip_runtime.h>.
//
// Lets g++ compile viamd_amd/csrc/*.hip|*.cpp unchanged into tests/emu/libviamd_emu.so, where every kernel runs
// on the CPU as cooperative fibers (one per GPU thread; wave64 collectives and __syncthreads are rendezvous points).
// Purpose: check kernel *logic* (indexing, segment enumeration, queue/flush machinery, host batching) against the
// oracle in this GPU-less container before spending GPU minutes.  It is NOT a backend: the viamd_amd Python package
// never loads it, and bench.py / smoke() / `-m gpu` tests only ever use the hipcc-built libviamd_amd.so.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)emu::dyn_shared();
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
#define VMD_UNIFORM_AS
#define VMD_SGPR_CAP(n)
#define VMD_BALLOT(pred) __ballot(pred)
#define VMD_NO_INLINE_ASM
#define VMD_LOAD_NT(ptr) (*(ptr))

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };

namespace emu {
struct Fiber;
extern Fiber* g_cur;
extern emu_uint3 g_blockIdx;
extern dim3 g_blockDim, g_gridDim;
emu_uint3 cur_tid();
int cur_lane();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void* dyn_shared();
void sync_block();
void sync_wave();
unsigned long long ballot(int pred);
uint32_t shfl_xor_bits(uint32_t v, int mask);
uint32_t readfirstlane_bits(uint32_t v);
uint32_t shfl_idx_bits(uint32_t v, int src);
}  // namespace emu

#define threadIdx (emu::cur_tid())
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

// ---- device builtins used by the kernels -----------------------------------------------------------------------
static inline void __syncthreads() { emu::sync_block(); }
static inline unsigned long long __ballot(int pred) { return emu::ballot(pred); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline float __shfl_xor(float v, int mask) {
    uint32_t b; memcpy(&b, &v, 4);
    b = emu::shfl_xor_bits(b, mask);
    float r; memcpy(&r, &b, 4); return r;
}
static inline unsigned emu_mbcnt_lo(unsigned mask, unsigned add) {
    const int lane = emu::cur_lane();
    const unsigned lt = lane >= 32 ? 0xffffffffu : ((1u << lane) - 1u);
    return add + (unsigned)__builtin_popcount(mask & lt);
}
static inline unsigned emu_mbcnt_hi(unsigned mask, unsigned add) {
    const int lane = emu::cur_lane();
    const unsigned lt = lane < 32 ? 0u : (lane == 63 ? 0x7fffffffu : ((1u << (lane - 32)) - 1u));
    return add + (unsigned)__builtin_popcount(mask & lt);
}
#define __builtin_amdgcn_mbcnt_lo(m, v) emu_mbcnt_lo((m), (v))
#define __builtin_amdgcn_mbcnt_hi(m, v) emu_mbcnt_hi((m), (v))
#define __builtin_amdgcn_readfirstlane(v) ((int)emu::readfirstlane_bits((uint32_t)(v)))
#define __builtin_amdgcn_wave_barrier() emu::sync_wave()
// ds_bpermute / v_readlane stand-ins (vmd_xtc_device.hip): every live lane of the wave takes part
#define VMD_SHFL_U32(v, src) emu::shfl_idx_bits((uint32_t)(v), (int)(src))
#define VMD_READLANE_U32(v, lane) emu::shfl_idx_bits((uint32_t)(v), (int)(lane))
#define VMD_XTC_BALLOT(pred) __ballot(pred)
#define VMD_XTC_SETPRIO() ((void)0)

// v_sqrt_f32 stand-in with a deliberate +-1 ulp error on half of the inputs: exercises the exactness fix-up of vmd_bin_add
static inline float emu_approx_sqrtf(float x) {
    float r = sqrtf(x);
    uint32_t b; memcpy(&b, &x, 4);
    if (!(r > 0.0f) || !std::isfinite(r)) return r;
    if ((b >> 3) & 1u) r = nextafterf(r, (b & 1u) ? INFINITY : 0.0f);
    return r;
}
#define __builtin_amdgcn_sqrtf(x) emu_approx_sqrtf(x)
typedef float emu_f2 __attribute__((vector_size(8)));
static inline emu_f2 emu_fma2(emu_f2 a, emu_f2 b, emu_f2 c) { emu_f2 r = {fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])}; return r; }
#define __builtin_elementwise_fma(a, b, c) emu_fma2((a), (b), (c))

template <typename T>
static inline T atomicAdd(T* p, T v) { T old = *p; *p = old + v; return old; }
template <typename T>
static inline T atomicExch(T* p, T v) { T old = *p; *p = v; return old; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned old = *p; if (v > old) *p = v; return old; }
static inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned old = *p; *p = old | v; return old; }
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { unsigned long long old = *p; if (v > old) *p = v; return old; }

// ---- host runtime -------------------------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
typedef struct emu_stream_t* hipStream_t;
typedef struct emu_event_t* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0
#define hipHostRegisterDefault 0

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)0x1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)0x1; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
// VIAMD_EMU_NO_HOST_REGISTER=1: the driver refuses to pin a file mapping (the evaluator must fall back to the load_raw copy)
static inline hipError_t hipHostRegister(void*, size_t, unsigned) { return getenv("VIAMD_EMU_NO_HOST_REGISTER") ? hipErrorInvalidValue : hipSuccess; }
static inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = (size_t)64 << 30; *total_b = (size_t)64 << 30; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)0x1; return hipSuccess; }
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)0x1; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); })
