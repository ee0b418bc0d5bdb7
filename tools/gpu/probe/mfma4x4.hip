// layout probe of v_mfma_f32_4x4x1_16b_f32 (used by gram_mfma in csrc/sdxp_persist.hip): lane l supplies A = l + 1 and B = 1000 + l;
// expected: lane 4 b + j, register i holds A(lane 4 b + i) * B(lane 4 b + j)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), (float)(1000 + l), c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
    const int b = l / 4, j = l % 4;
    const float want = (float)(4 * b + i + 1) * (float)(1000 + 4 * b + j);
    if (h[l * 4 + i] != want) { if (bad < 8) printf("lane %d reg %d: got %.0f want %.0f\n", l, i, h[l * 4 + i], want); ++bad; }
  }
  printf("mfma 4x4x1 layout probe: %d mismatches of 256\n", bad);
  return bad != 0;
}
