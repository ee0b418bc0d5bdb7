// sdxp_exbench.hip — microbenchmark of the exchange edge the persistent PPO update kernel is bound by (sdxp_persist.hip, DESIGN.md
// section 4b): a 256-CU -> 256-CU all-gather of (value, step tag) words, with nothing else in the step.  One workgroup of 512
// threads per CU (148 KB of dynamic LDS keep it at one per CU, as the update kernel's 147 KB do), R dependent rounds per launch:
//
//   every CU publishes its share of the W floats of the round (its share depends on what it gathered in the previous round, so the
//   rounds form the same dependent chain as x1 -> x2 -> x3 -> dY1 -> dY0 of an optimiser step), then gathers all W floats.
//
// What is timed is therefore the floor of ONE exchange edge: publish -> visible on every other CU -> gathered, including the wait
// for the slowest producer, for the word counts the update kernel moves (x3: 3 072 floats = 24 KB of 8-byte words, x2 / dY1: 6 144
// = 48 KB, x1: 12 288 = 96 KB, the dY0 Gram: 30 words per CU).  Variants:
//   fmt 0: one float per 8-byte word (value, tag) - what the update kernel uses;
//   fmt 1: three floats per 16-byte word (v0, v1, v2, tag) written / read by one dwordx4 access per lane.  A 16-byte access is not
//          architecturally single-copy atomic; the benchmark CHECKS every gathered payload against the value its tag implies and
//          counts mismatches (torn words), so the figure comes with its own safety evidence;
//   src 0: gather from all 256 CUs; 1: only from the 32 CUs of the consumer's own XCD (blockIdx % 8, the dispatcher's observed
//          placement - a speed assumption, never a correctness one); 2: only from another XCD's 32 CUs ((blockIdx + 1) % 8).
//          With src 1 / 2 every CU still publishes its W / 256 floats; a consumer gathers only the W / 8 floats of the 32 producers it
//          listens to (compare with the all-producer edge of W / 8 floats).
//   pattern 0: producer g owns a contiguous block of the array; 1: producer g owns elements 4 g .. 4 g + 3 of every 1 024-float row
//          (the x1 layout of the update kernel: a consumer's coalesced 512-byte load then spans 16 producers).
// Not part of include/seqdex.h: a measuring tool (tools/bench_exchange.py -> profiles/r4_exchange_edge_floor.txt).
#include <cstddef>
#include <cstdint>

#include "sdx_common.h"

namespace {
typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int XNWG = 256, XNTH = 512;

__device__ __forceinline__ float payload(unsigned tag, unsigned idx) {   // what word `idx` of round `tag` must carry
  return __uint_as_float(0x3f800000u | ((tag * 2654435761u + idx * 40503u) & 0x007fffffu));   // in [1, 2): finite, tag- and index-dependent
}

struct ExArgs {
  u64* ll;            // exchange buffer: >= words * 8 bytes (fmt 0), >= ceil(words / 3) * 16 bytes (fmt 1)
  long long* out;     // [0] s_memtime ticks of CU 0 over all rounds, [1] payload mismatches (all CUs), [2] time-outs, [3] checksum
  int words, rounds, fmt, src, pattern;
  unsigned tag0;
};

// Batches of B words per thread, all requested before the first tag is looked at, re-requested as a whole (wave-uniform retry) until
// every lane of the wave has its B tags: the shape of ll_gather<12> in sdxp_persist.hip.  Elements a thread is not interested in
// (idx < 0) are not loaded.
constexpr int XB = 6;
// one buffer_load / buffer_store_dwordx4 ... sc1 per lane (device scope, as the 8-byte agent-scope atomics lower to): the compiler
// counts these in its s_waitcnt bookkeeping, unlike inline asm
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(void* base) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ u32x4 load16(__amdgpu_buffer_rsrc_t r, unsigned q) { return __builtin_amdgcn_raw_buffer_load_b128(r, q * 16u, 0, 16); }
__device__ __forceinline__ void store16(__amdgpu_buffer_rsrc_t r, unsigned q, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, q * 16u, 0, 16); }
}  // namespace

__global__ __launch_bounds__(XNTH, 2) void k_exchange_bench(ExArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* acc_s = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, g = blockIdx.x;
  const int W = a.words;
  const __amdgpu_buffer_rsrc_t rs = rsrc_of(a.ll);
  int bad = 0, tmo = 0;
  float carry = 0.0f;             // what the CU gathered in the previous round: the next round's publish waits for it
  long long t0 = 0;
  if (tid == 0) acc_s[0] = 0.0f;
  __syncthreads();
  if (g == 0 && tid == 0) t0 = __builtin_amdgcn_s_memtime();
  // which producers this CU listens to, and which elements those own
  const int xcd = g & 7, want = a.src == 0 ? -1 : (a.src == 1 ? xcd : ((xcd + 1) & 7));
  for (int r = 0; r < a.rounds; ++r) {
    const unsigned tag = a.tag0 + (unsigned)r + 1u;
    // Two buffers by round parity, as the update kernel's factor buffers are doubled by step parity: a CU publishes round r + 1 as soon as
    // IT has gathered round r, while a slower CU may still be reading round-r words; round r + 2 (the same buffer again) cannot be
    // published before every CU has published r + 1, i.e. has finished gathering r.
    const unsigned pb = (unsigned)(r & 1) * (unsigned)(a.fmt == 0 ? W : ((W + 2) / 3 / XNWG) * XNWG);
    // ---- publish this CU's share
    if (a.fmt == 0) {
      const int share = W / XNWG;                        // words per CU
      for (int j = tid; j < share; j += XNTH) {
        const unsigned idx = a.pattern == 0 ? (unsigned)(g * share + j) : (unsigned)((j / 4) * 1024 + 4 * g + (j & 3));
        float v = payload(tag, idx);
        if (carry == -1.0f) v = 0.0f;                   // (never true: ties the store to the previous round's gather)
        __hip_atomic_store(a.ll + pb + idx, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      const int trip = (W + 2) / 3, share = trip / XNWG;   // 16-byte words per CU (3 floats each)
      for (int j = tid; j < share; j += XNTH) {
        const unsigned q = a.pattern == 0 ? (unsigned)(g * share + j) : (unsigned)((j / 2) * 512 + 2 * g + (j & 1));
        u32x4 v;
        v.x = __float_as_uint(payload(tag, 3 * q)); v.y = __float_as_uint(payload(tag, 3 * q + 1)); v.z = __float_as_uint(payload(tag, 3 * q + 2));
        v.w = tag;
        if (carry == -1.0f) v.x = 0u;
        store16(rs, pb + q, v);
      }
    }
    // ---- gather the round: lane-consecutive elements (coalesced 512-byte / 1-KB requests per wave), XB per thread in flight
    float sum = 0.0f;
    {
      const int total = a.fmt == 0 ? W : ((W + 2) / 3 / XNWG) * XNWG;          // 8-byte or 16-byte words of the round
      const int share = total / XNWG;
      for (int base = 0; base < total; base += XNTH * XB) {
        int idx[XB];
#pragma unroll
        for (int b = 0; b < XB; ++b) {
          const int i = base + b * XNTH + tid;
          int owner = 0;
          if (i < total) owner = a.pattern == 0 ? i / share : (a.fmt == 0 ? ((i & 1023) >> 2) : ((i & 511) >> 1));
          idx[b] = (i < total && (want < 0 || (owner & 7) == want)) ? i : -1;
        }
        unsigned spins = 0;
        u64 w8[XB];
        u32x4 w16[XB];
        for (;;) {
          bool ok = true;
          if (a.fmt == 0) {
#pragma unroll
            for (int b = 0; b < XB; ++b) w8[b] = idx[b] >= 0 ? __hip_atomic_load(a.ll + pb + idx[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((u64)tag << 32);
#pragma unroll
            for (int b = 0; b < XB; ++b) ok = ok && (int)((unsigned)(w8[b] >> 32) - tag) >= 0;   // this round's word, or (a producer that ran ahead) a later one
          } else {
#pragma unroll
            for (int b = 0; b < XB; ++b) { if (idx[b] >= 0) w16[b] = load16(rs, pb + (unsigned)idx[b]); else { w16[b].w = tag; } }
#pragma unroll
            for (int b = 0; b < XB; ++b) ok = ok && (int)(w16[b].w - tag) >= 0;
          }
          if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
          if (++spins > (1u << 18)) { tmo += 1; atomicAdd(reinterpret_cast<unsigned long long*>(a.out + 2), 1ull); break; }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int b = 0; b < XB; ++b) {
          if (idx[b] < 0) continue;
          if (a.fmt == 0) {
            const float v = __uint_as_float((unsigned)w8[b]);
            if (v != payload((unsigned)(w8[b] >> 32), (unsigned)idx[b])) bad += 1;
            sum += v;
          } else {
            const unsigned q = (unsigned)idx[b];
            const float f0 = __uint_as_float(w16[b].x), f1 = __uint_as_float(w16[b].y), f2 = __uint_as_float(w16[b].z);
            const unsigned wt = w16[b].w;
            if (f0 != payload(wt, 3u * q) || f1 != payload(wt, 3u * q + 1) || f2 != payload(wt, 3u * q + 2)) bad += 1;   // a torn word shows here
            sum += f0 + f1 + f2;
          }
        }
      }
    }
    // the CU as a whole has gathered the round before it publishes the next one (as a forward phase needs all of x1 in LDS)
    sum += __shfl_xor(sum, 32, 64); sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 8, 64);
    sum += __shfl_xor(sum, 4, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 1, 64);
    if ((tid & 63) == 0) atomicAdd(&acc_s[0], sum);
    // somebody timed out -> every CU gives up (decided by one thread per workgroup, so that the barriers below stay uniform)
    if (tid == 0) acc_s[1] = __hip_atomic_load(reinterpret_cast<unsigned long long*>(a.out + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1.0f : 0.0f;
    __syncthreads();
    carry = acc_s[0];
    const bool stop = acc_s[1] != 0.0f;
    __syncthreads();
    if (stop) break;
  }
  if (g == 0 && tid == 0) { a.out[0] = __builtin_amdgcn_s_memtime() - t0; a.out[3] = (long long)carry; }
  if (bad) atomicAdd(reinterpret_cast<unsigned long long*>(a.out + 1), (unsigned long long)bad);
  (void)tmo;
}

// point-to-point latency: CU 0 and CU `peer` bounce one 8-byte word `rounds` times (peer & 7 == 0: the same XCD under the observed
// placement).  out[0] = s_memtime ticks of CU 0 for all round trips.
__global__ __launch_bounds__(64) void k_pingpong(u64* ll, long long* out, int peer, int rounds, unsigned tag0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  (void)smem;
  const int g = blockIdx.x;
  if (threadIdx.x != 0 || (g != 0 && g != peer)) return;
  long long t0 = __builtin_amdgcn_s_memtime();
  int tmo = 0;
  for (int r = 0; r < rounds; ++r) {
    const unsigned tag = tag0 + (unsigned)r + 1u;
    u64* mine = ll + (g == 0 ? 0 : 64), *theirs = ll + (g == 0 ? 64 : 0);
    if (g == 0) __hip_atomic_store(mine, ((u64)tag << 32) | r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    for (;;) {
      const u64 w = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(w >> 32) == tag) break;
      if (++spins > (1u << 24)) { tmo = 1; break; }
    }
    if (g != 0) __hip_atomic_store(mine, ((u64)tag << 32) | r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tmo) break;
  }
  if (g == 0) { out[0] = __builtin_amdgcn_s_memtime() - t0; out[2] = tmo; }
}

// host side: `ll` >= 2 buffers x words x 8 bytes of device memory, `out` 4 x int64 on the device (zeroed by the caller).  Returns 0 / -1.
extern "C" int sdxpk_exchange_bench(void* ll, long long* out, int words, int rounds, int fmt, int src, int pattern, unsigned tag0, hipStream_t st) {
  if (words % (XNWG * 12) != 0 || rounds < 1 || fmt < 0 || fmt > 1 || src < 0 || src > 2 || pattern < 0 || pattern > 1) return -1;
  static bool attr = false;
  const int lds = 148 * 1024;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_exchange_bench), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
    attr = true;
  }
  ExArgs a;
  a.ll = (u64*)ll; a.out = out; a.words = words; a.rounds = rounds; a.fmt = fmt; a.src = src; a.pattern = pattern; a.tag0 = tag0;
  hipLaunchKernelGGL(k_exchange_bench, dim3(XNWG), dim3(XNTH), lds, st, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int sdxpk_pingpong_bench(void* ll, long long* out, int peer, int rounds, unsigned tag0, hipStream_t st) {
  if (peer < 1 || peer >= XNWG || rounds < 1) return -1;
  static bool attr = false;
  const int lds = 148 * 1024;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_pingpong), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
    attr = true;
  }
  hipLaunchKernelGGL(k_pingpong, dim3(XNWG), dim3(64), lds, st, (u64*)ll, out, peer, rounds, tag0);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
