// Repro probe for the finding of round 4 (csrc/sdx_gemm_nt.h, profiles/r4_gemm_nt_diag_branch_between_mfmas.txt): a wave with TWO 32 x 32
// accumulators issuing v_mfma_f32_32x32x2_f32 in a loop, with a block-uniform branch around VALU work between the MFMAs of one k step.
// Block 0 takes the branch (sums its A fragments), block 1 skips it.  Both compute the same 64 x 32 product; the probe compares each block's
// result with a host reference, so a wrong sum on the path that SKIPS the VALU block shows up as a mismatch of block 1 only.
// Build:  hipcc --offload-arch=gfx950 -O3 -o mfma_branch mfma_branch.hip   (add -DNO_BRANCH for the two-copies form that shipped)
// ISA:    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o mfma_branch.s mfma_branch.hip   -> look at the s_nop / dependency distance
//         around the s_cbranch between the v_mfma pairs (tools/gpu/probe/README in DESIGN.md section 12).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 256;
__global__ __launch_bounds__(64) void k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, float* __restrict__ rsum) {
  const int l = threadIdx.x, blk = blockIdx.x;
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
  const bool do_rs = blk == 0;            // block-uniform: a scalar branch
  float rs0 = 0.0f, rs1 = 0.0f;
  const float* a0p = A + (size_t)(l & 31) * K + (l >> 5);
  const float* a1p = A + (size_t)(32 + (l & 31)) * K + (l >> 5);
  const float* bp = B + (size_t)(l & 31) * K + (l >> 5);
#pragma unroll 4
  for (int kk = 0; kk < K; kk += 2) {
    const float a0 = a0p[kk], a1 = a1p[kk], b = bp[kk];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
#ifndef NO_BRANCH
    if (do_rs) { asm volatile("" ::: "memory"); rs0 += a0; rs1 += a1; }  // VALU work between the two MFMAs, skipped by block 1 (the empty asm keeps the compiler from predicating it)
#endif
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
  }
  float* c = C + (size_t)blk * 64 * 32;
  for (int i = 0; i < 16; ++i) {
    const int row = 8 * (i >> 2) + 4 * (l >> 5) + (i & 3), col = l & 31;
    c[(size_t)row * 32 + col] = acc0[i];
    c[(size_t)(32 + row) * 32 + col] = acc1[i];
  }
  rsum[blk * 128 + l] = rs0;
  rsum[blk * 128 + 64 + l] = rs1;
}
int main() {
  std::vector<float> hA(64 * K), hB(32 * K), hC(2 * 64 * 32), ref(64 * 32);
  srand(1);
  for (auto& x : hA) x = (float)(rand() % 17 - 8);
  for (auto& x : hB) x = (float)(rand() % 13 - 6);
  for (int i = 0; i < 64; ++i)
    for (int j = 0; j < 32; ++j) {
      double s = 0;
      for (int kk = 0; kk < K; ++kk) s += (double)hA[i * K + kk] * hB[j * K + kk];
      ref[i * 32 + j] = (float)s;     // small integers: exact in fp32
    }
  float *dA, *dB, *dC, *dR;
  (void)hipMalloc(&dA, hA.size() * 4); (void)hipMalloc(&dB, hB.size() * 4); (void)hipMalloc(&dC, hC.size() * 4); (void)hipMalloc(&dR, 256 * 4);
  (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  int total = 0;
  for (int rep = 0; rep < 50; ++rep) {
    (void)hipMemset(dC, 0, hC.size() * 4);
    hipLaunchKernelGGL(k, dim3(2), dim3(64), 0, 0, dA, dB, dC, dR);
    (void)hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    for (int blk = 0; blk < 2; ++blk) {
      int bad0 = 0, bad1 = 0;
      for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 32; ++j)
          if (hC[blk * 2048 + i * 32 + j] != ref[i * 32 + j]) (i < 32 ? bad0 : bad1)++;
      if (rep == 0 || bad0 || bad1) printf("rep %d block %d (%s the VALU block): %d / 1024 wrong in accumulator 0, %d / 1024 in accumulator 1\n", rep, blk, blk == 0 ? "takes" : "skips", bad0, bad1);
      total += bad0 + bad1;
    }
  }
  printf("mfma branch probe: %d wrong sums in 50 launches\n", total);
  return total != 0;
}
