// TEST INFRASTRUCTURE ONLY: what the C-ABI layer (seqdex_amd/csrc/sdx_capi.hip) links against besides the task / physics / camera
// kernel sources when the whole simulator is built for the SIMT emulator.  The one symbol that lives in the PPO library on the GPU
// (sdxpk_linear: the MFMA linear layer the Search task borrows for its RetriGraspTValue) is a plain loop here - the matrix cores
// are not emulated, and this stub is NOT the kernel under test.
#include <hip/hip_runtime.h>

extern "C" void sdxpk_linear(const float* X, const float* W, const float* b, float* Y, int M, int N, int K, int elu_flag,
                             const double* nmean, const double* nvar, hipStream_t) {
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      float acc = 0.0f;
      for (int k = 0; k < K; ++k) {
        float x = X[(size_t)m * K + k];
        if (nmean) {
          x = (x - (float)nmean[k]) / sqrtf((float)nvar[k] + 1e-5f);
          x = x < -5.0f ? -5.0f : (x > 5.0f ? 5.0f : x);
        }
        acc += x * W[(size_t)n * K + k];
      }
      float v = acc + b[n];
      if (elu_flag) v = v > 0.0f ? v : expm1f(v);
      Y[(size_t)m * N + n] = v;
    }
}
