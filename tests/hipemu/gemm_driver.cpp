// TEST INFRASTRUCTURE ONLY: seqdex_amd/csrc/sdx_gemm_nt.h (the product's kernel source) on the SIMT emulator, so that the swizzle, the
// fragment / accumulator layouts, the epilogues and the staging kernel are checked against numpy in the CPU suite.
#include <hipemu_mfma.h>

#include "sdx_common.h"
#include "sdx_gemm_nt.h"

// one product: A [M][lda], B [N][ldb] in the element type (fp32 or bf16 bit patterns as uint16), K a multiple of the chunk
extern "C" int emu_gemm_nt(int bf, int epi, const void* A, int lda, const void* B, int ldb, int M, int N, int K, int kchunk, int splits,
                           float* Cf, int ldc, long long cz, void* Cn, int ldn, void* Ct, int ldt, const float* bias, const void* H, int ldh,
                           const void* Ht, int ldht, float* rowsum) {
  NtArgs g = {A, lda, B, ldb, M, N, K, kchunk, Cf, ldc, (size_t)cz, Cn, ldn, Ct, ldt, bias, H, ldh, Ht, ldht, rowsum};
  NtArgs gs[3] = {g, g, g};
  if (bf) {
    if (epi == EPI_FWD) gemm_nt<1, EPI_FWD>(gs, 1, splits, nullptr);
    else if (epi == EPI_NN) gemm_nt<1, EPI_NN>(gs, 1, splits, nullptr);
    else gemm_nt<1, EPI_TN>(gs, 1, splits, nullptr);
  } else {
    if (epi == EPI_FWD) gemm_nt<0, EPI_FWD>(gs, 1, splits, nullptr);
    else if (epi == EPI_NN) gemm_nt<0, EPI_NN>(gs, 1, splits, nullptr);
    else gemm_nt<0, EPI_TN>(gs, 1, splits, nullptr);
  }
  return 0;
}
// the 128 x 64 tile shape (the launcher picks it for small grids) forced for a forward product
extern "C" int emu_gemm_nt_narrow(int bf, const void* A, int lda, const void* B, int ldb, int M, int N, int K, float* Cf, int ldc, void* Ct,
                                  int ldt, const float* bias) {
  NtBatch nb;
  NtArgs g = {A, lda, B, ldb, M, N, K, K, Cf, ldc, 0, nullptr, 0, Ct, ldt, bias, nullptr, 0, nullptr, 0, nullptr};
  nb.a[0] = nb.a[1] = nb.a[2] = g;
  nb.splits = 1;
  const int nx = (N + 63) / 64, ny = (M + 127) / 128;
  if (bf) hipLaunchKernelGGL((k_gemm_nt<1, EPI_FWD, 1>), dim3(nx * ny), dim3(256), 2 * 192 * 128, nullptr, nb, nx, ny);
  else hipLaunchKernelGGL((k_gemm_nt<0, EPI_FWD, 1>), dim3(nx * ny), dim3(256), 2 * 192 * 128, nullptr, nb, nx, ny);
  return 0;
}
extern "C" int emu_stage(int bf, const float* src, int lds, int R, int K, int Kp, void* dn, int ldn, void* dt, int ldt) {
  StageBatch sb;
  StageArgs a = {src, lds, R, K, Kp, dn, ldn, dt, ldt};
  for (int q = 0; q < 9; ++q) sb.a[q] = a;
  dim3 grid((Kp + 63) / 64, (R + 63) / 64, 1);
  if (bf) hipLaunchKernelGGL((k_stage<1>), grid, dim3(256), 0, nullptr, sb);
  else hipLaunchKernelGGL((k_stage<0>), grid, dim3(256), 0, nullptr, sb);
  return 0;
}

// the 128 x 128 tile shape forced (the launcher only picks it for grids of 512 workgroups and more): forward or data-gradient product
extern "C" int emu_gemm_nt_wide(int bf, int epi, const void* A, int lda, const void* B, int ldb, int M, int N, int K, float* Cf, int ldc, void* Cn,
                                int ldn, void* Ct, int ldt, const float* bias, const void* H, int ldh, const void* Ht, int ldht) {
  NtBatch nb;
  NtArgs g = {A, lda, B, ldb, M, N, K, K, Cf, ldc, 0, Cn, ldn, Ct, ldt, bias, H, ldh, Ht, ldht, nullptr};
  nb.a[0] = nb.a[1] = nb.a[2] = g;
  nb.splits = 1;
  const int nx = (N + 127) / 128, ny = (M + 127) / 128;
  if (bf && epi == EPI_FWD) hipLaunchKernelGGL((k_gemm_nt<1, EPI_FWD, 2>), dim3(nx * ny), dim3(256), 2 * 256 * 128, nullptr, nb, nx, ny);
  else if (bf) hipLaunchKernelGGL((k_gemm_nt<1, EPI_NN, 2>), dim3(nx * ny), dim3(256), 2 * 256 * 128, nullptr, nb, nx, ny);
  else if (epi == EPI_FWD) hipLaunchKernelGGL((k_gemm_nt<0, EPI_FWD, 2>), dim3(nx * ny), dim3(256), 2 * 256 * 128, nullptr, nb, nx, ny);
  else hipLaunchKernelGGL((k_gemm_nt<0, EPI_NN, 2>), dim3(nx * ny), dim3(256), 2 * 256 * 128, nullptr, nb, nx, ny);
  return 0;
}

// fp32 weight gradient from [row][feature] operands (k_gemm_tt): G[n][k] = sum_r A[r][n] B[r][k] in `splits` row ranges of kchunk rows;
// tile: 0 = the launcher's choice, 1 = 128 x 64, 2 = 128 x 128
extern "C" int emu_gemm_tt(const float* A, int lda, const float* B, int ldb, int M, int N, int K, int kchunk, int splits, float* Cf, int ldc,
                           long long cz, float* rowsum, int tile) {
  static float zeros[64] = {0};
  NtArgs g = {A, lda, B, ldb, M, N, K, kchunk, Cf, ldc, (size_t)cz, nullptr, 0, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, rowsum};
  NtArgs gs[3] = {g, g, g};
  if (tile == 0) { gemm_tt(gs, 1, splits, zeros, nullptr); return 0; }
  NtBatch nb;
  nb.a[0] = nb.a[1] = nb.a[2] = g;
  nb.splits = splits;
  const int ny = (M + 127) / 128;
  if (tile == 2) { const int nx = (N + 127) / 128; hipLaunchKernelGGL((k_gemm_tt<2>), dim3(nx * ny * splits), dim3(256), 2 * 32 * 256 * 4, nullptr, nb, nx, ny, zeros); }
  else { const int nx = (N + 63) / 64; hipLaunchKernelGGL((k_gemm_tt<1>), dim3(nx * ny * splits), dim3(256), 2 * 32 * 192 * 4, nullptr, nb, nx, ny, zeros); }
  return 0;
}
