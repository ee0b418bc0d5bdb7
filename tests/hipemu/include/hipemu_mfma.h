// TEST INFRASTRUCTURE ONLY (tests/hipemu): what sdx_gemm_nt.h needs on top of hip_runtime.h - float4, the matrix-core builtins as wave
// collectives, global_load_lds as a synchronous per-lane copy.  Compiled with clang++ for x86 (ext_vector_type, __bf16).
//
// MFMA layouts as the CDNA4 ISA defines them (cdna_hip_programming.md section 3):
//   32x32x2 f32 : lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]
//   32x32x16 bf16: lane l holds A[i = l & 31][k = 8 (l >> 5) + 0..7] and B[k = 8 (l >> 5) + 0..7][j = l & 31]
//   C / D        : lane l, register r holds D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]
// f32: D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) (a k-ordered fma chain, exact fp32); bf16: exact products, fp32 sum in ascending k.
#pragma once
#include <hip/hip_runtime.h>

struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

namespace hipemu_mm {
struct Slot { float a[8], b[8]; };
static Slot g_slot[16][64];          // [wave of the block][lane]: operands published before the rendezvous
}
#define HIPEMU_WAVE_SYNC() ((void)hipemu::collective(hipemu::K_WAVE_SYNC, 0u, 0, (const void*)(intptr_t)(__COUNTER__ + 1)))

typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));

template <int SITE>
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x2(float a, float b, hipemu_f32x16 c) {
  const int tid = (int)threadIdx.x, w = tid >> 6, l = tid & 63;
  hipemu_mm::g_slot[w][l].a[0] = a; hipemu_mm::g_slot[w][l].b[0] = b;
  (void)hipemu::collective(hipemu::K_WAVE_SYNC, 0u, 0, (const void*)(intptr_t)(2 * SITE + 1));
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    float d = c[r];
    for (int k = 0; k < 2; ++k) d = fmaf(hipemu_mm::g_slot[w][row + 32 * k].a[0], hipemu_mm::g_slot[w][col + 32 * k].b[0], d);
    c[r] = d;
  }
  (void)hipemu::collective(hipemu::K_WAVE_SYNC, 0u, 0, (const void*)(intptr_t)(2 * SITE + 2));
  return c;
}
template <int SITE>
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x16_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c) {
  const int tid = (int)threadIdx.x, w = tid >> 6, l = tid & 63;
  for (int e = 0; e < 8; ++e) { hipemu_mm::g_slot[w][l].a[e] = (float)a[e]; hipemu_mm::g_slot[w][l].b[e] = (float)b[e]; }
  (void)hipemu::collective(hipemu::K_WAVE_SYNC, 0u, 0, (const void*)(intptr_t)(2 * SITE + 1));
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    float d = c[r];
    for (int k = 0; k < 16; ++k) d += hipemu_mm::g_slot[w][row + 32 * (k >> 3)].a[k & 7] * hipemu_mm::g_slot[w][col + 32 * (k >> 3)].b[k & 7];
    c[r] = d;
  }
  (void)hipemu::collective(hipemu::K_WAVE_SYNC, 0u, 0, (const void*)(intptr_t)(2 * SITE + 2));
  return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu_mfma_f32_32x32x2<__COUNTER__ + 1000>((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu_mfma_f32_32x32x16_bf16<__COUNTER__ + 1000>((a), (b), (c))
#define __builtin_amdgcn_readfirstlane(x) (x)
// global_load_lds: LDS destination = the (wave-uniform) base every lane passes + lane x size; synchronous here
static inline void hipemu_glds(const void* g, void* l, int size) { memcpy((char*)l + ((int)threadIdx.x & 63) * size, g, (size_t)size); }
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) hipemu_glds((const void*)(g), (void*)(l), (size))
