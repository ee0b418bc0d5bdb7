// sdx_gemm.h — fp32-input MFMA GEMM building block shared by the large-minibatch PPO step (sdxp_bigmb.hip) and the T-value
// trainer (sdx_tvtrain.hip): C = op(A) op(B) on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain), LDS-tiled, with the
// epilogues the MLP forward / backward passes need.  gfx950 only.
#pragma once
#include "sdx_common.h"
#include <cstddef>
#include <cstdint>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define TB 64   // output tile

__device__ __forceinline__ float belu(float x) { return x > 0.0f ? x : expm1f(x); }
__device__ __forceinline__ float belu_grad_from_out(float h) { return h > 0.0f ? 1.0f : h + 1.0f; }

// four consecutive floats p[0..3] of which the first `valid` exist; one 16-byte load when possible
__device__ __forceinline__ float4 load4(const float* __restrict__ p, int valid) {
  if (valid >= 4 && ((reinterpret_cast<uintptr_t>(p) & 15u) == 0)) return *reinterpret_cast<const float4*>(p);
  float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (valid > 0) v.x = p[0];
  if (valid > 1) v.y = p[1];
  if (valid > 2) v.z = p[2];
  if (valid > 3) v.w = p[3];
  return v;
}

struct GemmArgs {
  const float* A; int lda;      // AT == 0: A[i][k] (k contiguous); AT == 1: stored [k][i] (i contiguous)
  const float* B; int ldb;      // BT == 0: B stored [j][k] (k contiguous); BT == 1: stored [k][j] (j contiguous)
  float* C; int ldc;            // C[i][j]; split z writes to C + z * cz
  size_t cz;
  int M, N, K, kchunk;          // reduction range of split z: [z * kchunk, min(K, (z + 1) * kchunk))
  const float* bias;            // EPI 1/2: bias[j]
  const float* H; int ldh;      // EPI 3: C = acc * ELU'(H[i][j]) with H the layer OUTPUT
  float* rowsum;                // EPI 4: rowsum[z * cz + i] = sum_k A(i, k) over the split (the bias gradient of a dY^T X product)
};

// EPI: 0 none, 1 bias + ELU, 2 bias, 3 times ELU'(H), 4 none + row sums of A.  WT: waves own (32 WT) x (32 WT) of a (64 WT)^2 tile.
// The global loads of reduction chunk c+1 are issued before the MFMAs of chunk c (register double buffering).
// One launch serves up to three independent products (the same layer of the actor, critic and central-value networks):
// blockIdx.z = problem * splits + split, so that the narrow layers still give every CU several workgroups.
struct GemmBatch { GemmArgs a[3]; int splits; };

// BF = 1: the operands are rounded to bf16 (round to nearest even, v_cvt_pk_bf16_f32) on their way into LDS and multiplied on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation; sources, results, weights and optimiser state stay fp32 (BASELINE.json
// configs[4] "bf16 policy", SURVEY 8(d) config 5: fp32 master weights / accumulate).  Lane l feeds row / column l & 31 and the eight
// reduction indices 8 (l >> 5) .. + 7 of a 16-wide step to BOTH operands, so the sum over k is complete whatever order the matrix
// core pairs them in.
template <int AT, int BT, int EPI, int WTM, int WTN, int KT, int BF = 0>
__global__ __launch_bounds__(256) void k_gemm(GemmBatch gb) {
  constexpr int TM = TB * WTM, TN = TB * WTN;       // tile: waves own (32 WTM) x (32 WTN) of it, 2 x 2 waves
  constexpr int NVA = TM * KT / 4 / 256, NVB = TN * KT / 4 / 256;   // float4 loads per thread per reduction chunk
  constexpr int KP = BF ? KT + 8 : KT + 1;          // LDS row stride: odd for 4-byte reads, 16-byte aligned rows for the 8 x bf16 reads
  typedef typename std::conditional<BF != 0, __bf16, float>::type lds_t;
  __shared__ __attribute__((aligned(16))) lds_t As[TM][KP];
  __shared__ __attribute__((aligned(16))) lds_t Bs[TN][KP];
  const GemmArgs& g = gb.a[blockIdx.z / gb.splits];
  const int zs = blockIdx.z % gb.splits;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i0 = blockIdx.y * TM, j0 = blockIdx.x * TN;
  if (i0 >= g.M || j0 >= g.N) return;               // the grid covers the largest problem of the batch
  const int kbeg = zs * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  const int wm = (wave >> 1) * 32 * WTM, wn = (wave & 1) * 32 * WTN;
  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int u = 0; u < WTM; ++u)
#pragma unroll
    for (int v = 0; v < WTN; ++v)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[u][v][i] = 0.0f;
  // element e = tid + 256 p of a chunk: k-contiguous sources -> (row e / (KT/4), k offset 4 (e % (KT/4)));
  // transposed sources -> (k row e / (T/4), i/j offset 4 (e % (T/4))): consecutive lanes read consecutive 16-byte pieces
  float4 av[NVA], bv[NVB];
  float rsum = 0.0f;
  auto fetch = [&](int k0) {
#pragma unroll
    for (int p = 0; p < NVA; ++p) {
      const int e = tid + 256 * p;
      av[p] = make_float4(0, 0, 0, 0);
      if (AT == 0) { const int r = i0 + e / (KT / 4), k = k0 + 4 * (e % (KT / 4)); if (r < g.M && k < kend) av[p] = load4(g.A + (size_t)r * g.lda + k, kend - k); }
      else         { const int k = k0 + e / (TM / 4), c = i0 + 4 * (e % (TM / 4));  if (k < kend && c < g.M) av[p] = load4(g.A + (size_t)k * g.lda + c, g.M - c); }
    }
#pragma unroll
    for (int p = 0; p < NVB; ++p) {
      const int e = tid + 256 * p;
      bv[p] = make_float4(0, 0, 0, 0);
      if (BT == 0) { const int r = j0 + e / (KT / 4), k = k0 + 4 * (e % (KT / 4)); if (r < g.N && k < kend) bv[p] = load4(g.B + (size_t)r * g.ldb + k, kend - k); }
      else         { const int k = k0 + e / (TN / 4), c = j0 + 4 * (e % (TN / 4));  if (k < kend && c < g.N) bv[p] = load4(g.B + (size_t)k * g.ldb + c, g.N - c); }
    }
  };
  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += KT) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NVA; ++p) {
      const int e = tid + 256 * p;
      if (AT == 0) { lds_t* d = &As[e / (KT / 4)][4 * (e % (KT / 4))]; d[0] = (lds_t)av[p].x; d[1] = (lds_t)av[p].y; d[2] = (lds_t)av[p].z; d[3] = (lds_t)av[p].w; }
      else         { const int k = e / (TM / 4), c = 4 * (e % (TM / 4)); As[c][k] = (lds_t)av[p].x; As[c + 1][k] = (lds_t)av[p].y; As[c + 2][k] = (lds_t)av[p].z; As[c + 3][k] = (lds_t)av[p].w; }
    }
#pragma unroll
    for (int p = 0; p < NVB; ++p) {
      const int e = tid + 256 * p;
      if (BT == 0) { lds_t* d = &Bs[e / (KT / 4)][4 * (e % (KT / 4))]; d[0] = (lds_t)bv[p].x; d[1] = (lds_t)bv[p].y; d[2] = (lds_t)bv[p].z; d[3] = (lds_t)bv[p].w; }
      else         { const int k = e / (TN / 4), c = 4 * (e % (TN / 4)); Bs[c][k] = (lds_t)bv[p].x; Bs[c + 1][k] = (lds_t)bv[p].y; Bs[c + 2][k] = (lds_t)bv[p].z; Bs[c + 3][k] = (lds_t)bv[p].w; }
    }
    __syncthreads();
    if (k0 + KT < kend) fetch(k0 + KT);
    if (EPI == 4 && blockIdx.x == 0 && tid < TM) {
#pragma unroll
      for (int kk = 0; kk < KT; ++kk) rsum += (float)As[tid][kk];
    }
    if constexpr (BF != 0) {
#pragma unroll
      for (int kk = 0; kk < KT; kk += 16) {
        bf16x8 a[WTM], b[WTN];
#pragma unroll
        for (int u = 0; u < WTM; ++u) a[u] = *reinterpret_cast<const bf16x8*>(&As[wm + 32 * u + (lane & 31)][kk + 8 * (lane >> 5)]);
#pragma unroll
        for (int v = 0; v < WTN; ++v) b[v] = *reinterpret_cast<const bf16x8*>(&Bs[wn + 32 * v + (lane & 31)][kk + 8 * (lane >> 5)]);
#pragma unroll
        for (int u = 0; u < WTM; ++u)
#pragma unroll
          for (int v = 0; v < WTN; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[v], acc[u][v], 0, 0, 0);
      }
    } else
#pragma unroll
    for (int kk = 0; kk < KT; kk += 2) {
      float a[WTM], b[WTN];
#pragma unroll
      for (int u = 0; u < WTM; ++u) a[u] = As[wm + 32 * u + (lane & 31)][kk + (lane >> 5)];
#pragma unroll
      for (int v = 0; v < WTN; ++v) b[v] = Bs[wn + 32 * v + (lane & 31)][kk + (lane >> 5)];
#pragma unroll
      for (int u = 0; u < WTM; ++u)
#pragma unroll
        for (int v = 0; v < WTN; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[v], acc[u][v], 0, 0, 0);
    }
  }
  if (EPI == 4 && blockIdx.x == 0 && tid < TM && i0 + tid < g.M) g.rowsum[(size_t)zs * g.cz + i0 + tid] = rsum;
  float* C = g.C + (size_t)zs * g.cz;
#pragma unroll
  for (int v = 0; v < WTN; ++v) {
    const int col = j0 + wn + 32 * v + (lane & 31);
    if (col >= g.N) continue;
    const float bias = (EPI == 1 || EPI == 2) ? g.bias[col] : 0.0f;
#pragma unroll
    for (int u = 0; u < WTM; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M) {
          float x = acc[u][v][r] + bias;
          if (EPI == 1) x = belu(x);
          if (EPI == 3) x *= belu_grad_from_out(g.H[(size_t)row * g.ldh + col]);
          C[(size_t)row * g.ldc + col] = x;
        }
      }
  }
}

// out[i] = sum_z part[z * pz + i] in fixed order
static __global__ __launch_bounds__(256) void k_reduce_parts(const float* __restrict__ part, size_t pz, int S, size_t n, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float a = 0.0f;
    for (int z = 0; z < S; ++z) a += part[(size_t)z * pz + i];
    out[i] = a;
  }
}
struct ReduceBatch { const float* part[3]; size_t pz[3]; size_t n[3]; float* out[3]; int S; };
static __global__ __launch_bounds__(256) void k_reduce_parts3(ReduceBatch rb) {
  const int q = blockIdx.y;
  const float* __restrict__ part = rb.part[q];
  float* __restrict__ out = rb.out[q];
  const size_t pz = rb.pz[q], n = rb.n[q];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float a = 0.0f;
    for (int z = 0; z < rb.S; ++z) a += part[(size_t)z * pz + i];
    out[i] = a;
  }
}

// 128 x 128 tiles (each wave 64 x 64: half the LDS and L2 traffic per flop) when they still give every CU a workgroup
template <int AT, int BT, int EPI>
static void gemm(const GemmArgs* gs, int count, int splits, hipStream_t st, bool bf16 = false) {
  GemmBatch gb;
  int Mx = 0, Nx = 0;
  for (int q = 0; q < 3; ++q) {
    gb.a[q] = gs[q < count ? q : 0];
    if (q < count) { Mx = gs[q].M > Mx ? gs[q].M : Mx; Nx = gs[q].N > Nx ? gs[q].N : Nx; }
  }
  gb.splits = splits;
  // tile choice: 128 x 128 (each wave 64 x 64: least LDS / L2 traffic per flop) when that fills the 256 CUs with whole rounds
  // of workgroups, else 128 x 64, else 64 x 64
  auto blocks = [&](int tm, int tn) { return (long)((Nx + tn - 1) / tn) * ((Mx + tm - 1) / tm) * splits * count; };
  auto waste = [&](long b) { const long rounds = (b + 255) / 256; return (double)(rounds * 256 - b) / (double)(rounds * 256); };
  const long b22 = blocks(128, 128), b21 = blocks(128, 64);
  if (b22 >= 256 && waste(b22) <= 0.15) {
    dim3 grid((Nx + 127) / 128, (Mx + 127) / 128, splits * count);
    if (bf16) hipLaunchKernelGGL((k_gemm<AT, BT, EPI, 2, 2, 32, 1>), grid, dim3(256), 0, st, gb);
    else hipLaunchKernelGGL((k_gemm<AT, BT, EPI, 2, 2, 32>), grid, dim3(256), 0, st, gb);
  } else if (b21 >= 256 && waste(b21) <= 0.15) {
    dim3 grid((Nx + 63) / 64, (Mx + 127) / 128, splits * count);
    if (bf16) hipLaunchKernelGGL((k_gemm<AT, BT, EPI, 2, 1, 32, 1>), grid, dim3(256), 0, st, gb);
    else hipLaunchKernelGGL((k_gemm<AT, BT, EPI, 2, 1, 32>), grid, dim3(256), 0, st, gb);
  } else {
    dim3 grid((Nx + TB - 1) / TB, (Mx + TB - 1) / TB, splits * count);
    if (bf16) hipLaunchKernelGGL((k_gemm<AT, BT, EPI, 1, 1, 16, 1>), grid, dim3(256), 0, st, gb);
    else hipLaunchKernelGGL((k_gemm<AT, BT, EPI, 1, 1, 16>), grid, dim3(256), 0, st, gb);
  }
}

