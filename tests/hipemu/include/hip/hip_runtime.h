// TEST INFRASTRUCTURE ONLY (tests/hipemu): a minimal SIMT emulator that lets the product's .hip kernel sources be compiled with g++
// and executed on the CPU, one fiber per GPU thread, so that kernel LOGIC can be checked against the oracles without a GPU.
// It is not a CPU path of the product: nothing under seqdex_amd/ includes it; only tests/ builds it (tests/hipemu/build.py).
//
// Semantics provided (what the kernels of seqdex_amd/csrc use):
//   * blocks run one after the other; the threads of a block are fibers scheduled wave by wave (64 lanes per wave);
//   * __syncthreads(): a fiber parks until every live fiber of the block has arrived;
//   * wave collectives (__ballot, __shfl, __shfl_xor, __shfl_down): a lane parks; when no lane of its wave can run any more,
//     the parked lanes are grouped by source call site and each group is resolved as one exec-masked instruction would be - lanes that
//     took another branch simply are not part of the group (reading such a lane returns the caller's own value);
//   * LDS (extern __shared__ and static __shared__) is ordinary memory shared by the block's fibers; LDS/global atomics are plain
//     read-modify-writes (fibers are cooperative, nothing preempts them).
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <functional>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __constant__
// one OS thread runs every fiber: `thread_local` gives block-scope LDS arrays static storage and is also legal after `extern`
#define __shared__ thread_local
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__
#endif

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
// ---- the slice of the HIP runtime API the C-ABI layer (sdx_capi.hip, sdx_camera.hip) uses: "device" memory is host memory, streams
// do not exist (every launch has completed when hipLaunchKernelGGL returns), there is exactly one device
#include <stdlib.h>
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t = nullptr) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }

namespace hipemu {
struct Idx { unsigned x, y, z; };
extern Idx g_tid, g_bid, g_bdim, g_gdim;   // refreshed by the scheduler every time a fiber is resumed
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void barrier();
// barrier among a SUBSET of the block's threads (the kernel's spin-wait on an LDS counter between some of its waves): a thread parks until
// `count` threads of the block are parked here
void group_barrier(int count);
enum { K_BALLOT = 0, K_SHFL = 1, K_SHFL_XOR = 2, K_SHFL_DOWN = 3, K_SHFL_UP = 4, K_WAVE_SYNC = 5, K_DPP = 6 };
unsigned long long collective(int kind, uint32_t value, int arg, const void* site);
}  // namespace hipemu

#define threadIdx (hipemu::g_tid)
#define blockIdx (hipemu::g_bid)
#define blockDim (hipemu::g_bdim)
#define gridDim (hipemu::g_gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch(dim3(grid), dim3(block), (size_t)(shmem), [&]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::barrier(); }

// every collective carries the id of its SOURCE call site (__COUNTER__ at macro expansion): unlike a return address it survives the
// compiler duplicating or peeling the code around the call
static inline unsigned long long hipemu_ballot(int site, int pred) {
  return hipemu::collective(hipemu::K_BALLOT, pred ? 1u : 0u, 0, (const void*)(intptr_t)(site + 1));
}
static inline uint32_t hipemu_bits(float v) { uint32_t u; memcpy(&u, &v, 4); return u; }
static inline float hipemu_float(uint32_t u) { float v; memcpy(&v, &u, 4); return v; }
static inline float hipemu_shfl(int site, int kind, float v, int arg, int width = 64) {
  (void)width; return hipemu_float((uint32_t)hipemu::collective(kind, hipemu_bits(v), arg, (const void*)(intptr_t)(site + 1)));
}
static inline int hipemu_shfl(int site, int kind, int v, int arg, int width = 64) {
  (void)width; return (int)(uint32_t)hipemu::collective(kind, (uint32_t)v, arg, (const void*)(intptr_t)(site + 1));
}
static inline unsigned hipemu_shfl(int site, int kind, unsigned v, int arg, int width = 64) {
  (void)width; return (unsigned)hipemu::collective(kind, (uint32_t)v, arg, (const void*)(intptr_t)(site + 1));
}
#define __ballot(pred) hipemu_ballot(__COUNTER__, (pred))
#define __shfl(...) hipemu_shfl(__COUNTER__, hipemu::K_SHFL, __VA_ARGS__)
#define __shfl_xor(...) hipemu_shfl(__COUNTER__, hipemu::K_SHFL_XOR, __VA_ARGS__)
#define __shfl_down(...) hipemu_shfl(__COUNTER__, hipemu::K_SHFL_DOWN, __VA_ARGS__)
#define __shfl_up(...) hipemu_shfl(__COUNTER__, hipemu::K_SHFL_UP, __VA_ARGS__)
// wave-synchronous sections (LDS written by one lane, read by another lane of the SAME wave, no workgroup barrier): on the GPU the
// wave executes in lockstep and the builtins below only constrain the compiler; here they are a rendezvous of the wave's lanes
#define __builtin_amdgcn_wave_barrier() ((void)hipemu::collective(hipemu::K_WAVE_SYNC, 0u, 0, (const void*)(intptr_t)(__COUNTER__ + 1)))
// v_mov_b32 dpp (quad_perm 0x00-0xff, row_half_mirror 0x141, row_mirror 0x140): (old, src, ctrl, row_mask, bank_mask, bound_ctrl)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) ((int)(uint32_t)hipemu::collective(hipemu::K_DPP, (uint32_t)(src), (ctrl), (const void*)(intptr_t)(__COUNTER__ + 1)))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_s_sleep(p) ((void)0)

static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __float_as_int(float v) { return (int)hipemu_bits(v); }
static inline float __int_as_float(int v) { return hipemu_float((uint32_t)v); }
static inline unsigned __float_as_uint(float v) { return hipemu_bits(v); }
static inline float __uint_as_float(unsigned v) { return hipemu_float(v); }
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline long long hipemu_cycles() { return 0; }
#define __builtin_readcyclecounter() hipemu_cycles()
