// sdx_gemm_nt.h — the large-minibatch PPO step's three trunk products (forward, data gradient, weight gradient) as ONE kernel
// shape for gfx950: C[i][j] = sum_k A[i][k] B[j][k] with BOTH operands k-contiguous in HBM ("NT"), fp32 or bf16 elements, fp32
// accumulation on v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16.  Round 4 replaces k_gemm of sdx_gemm.h on
// these products: that kernel loaded fp32 through registers, converted on the way into LDS and kept one chunk in flight.
//
//   * operands are STAGED by their producers, never converted on the load path: every array a product reads exists in the element
//     type of the run, k-contiguous, with its reduction length padded to the chunk with zeros - layer outputs and their gradients are
//     written by the producing epilogue in both orientations ([row][feature] for the forward / data-gradient products, [feature][row]
//     for the weight-gradient product, whose reduction runs over the minibatch rows), weights and their transposes by k_stage once
//     per optimiser step (3.4 M parameters), dataset rows once per epoch.  So a transposed operand is never transposed on the way into
//     LDS either, and there is no tail logic in the reduction loop.
//   * global -> LDS by global_load_lds_dwordx4 (16 B per lane, no VGPR round trip, no ds_write): a chunk is 128 B of every tile row
//     (32 floats / 64 bf16), i.e. one 1-KB wave instruction per 8 rows.  The LDS image is lane-linear (row r at r * 128 B), so the
//     bank swizzle is applied on the SOURCE side: the 16-byte piece s of row r is fetched into slot s ^ ((r >> 1) & 7), and the
//     fragment reads apply the same XOR (an involution).  With that key the four 16-lane groups a ds_read_b128 is serviced in
//     ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32; MI355X_MICROARCH.md, LDS) each touch 16 distinct (bank-row half, slot) pairs:
//     conflict-free.
//   * two LDS stages (2 x 32 KB for a 128 x 128 tile): the loads of chunk c + 1 are in flight while chunk c is multiplied; one
//     barrier per chunk; two workgroups per CU (64 KB each) overlap each other's barrier bubbles.
//   * fragments: lane l reads 16 B of row (l & 31) at piece 2 t + (l >> 5), t = 0..3.  bf16: the 8 reduction indices of one
//     32x32x16 step.  fp32: 4 consecutive indices - component c of the low / high half-wave is index 8 t + c / 8 t + 4 + c, fed to
//     four 32x32x2 steps; A and B use the same assignment, so every index meets its partner.
//   * epilogues write what the next product reads: EPI_FWD bias + ELU -> fp32 [i][j] (the ELU' of the backward pass and the fp32
//     heads read it) + element-type copy + transposed copy; EPI_NN times ELU'(layer output) -> element-type copy + transposed copy;
//     EPI_TN split partials of G = dY^T X and, on the first column block, the row sums of A (= the bias gradient) from one more
//     MFMA per step against a constant fragment of ones.
#pragma once
#include <cstdlib>

#include "sdx_gemm.h"

#define EPI_FWD 1
#define EPI_NN 3
#define EPI_TN 4

struct NtArgs {
  const void* A; int lda;        // A[i][k] (elements), k contiguous, rows 16-byte aligned
  const void* B; int ldb;        // B[j][k]
  int M, N, K, kchunk;           // K: reduction length, a multiple of the chunk (32 fp32 / 64 bf16); split z covers [z kchunk, min(K, (z+1) kchunk))
  float* Cf; int ldc; size_t cz; // fp32 result [i][j] of split z at Cf + z cz (or null)
  void* Cn; int ldn;             // element-type copy [i][j] (or null)
  void* Ct; int ldt;             // element-type transposed copy [j][i] (or null)
  const float* bias;             // EPI_FWD: bias[j]
  const float* H; int ldh;       // EPI_NN: the layer OUTPUT whose ELU' multiplies the result
  float* rowsum;                 // EPI_TN: rowsum[z cz + i] = sum_k A[i][k] over the split
};
struct NtBatch { NtArgs a[3]; int splits; };

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BF, int EPI, int WTN>
__global__ __launch_bounds__(256, 2) void k_gemm_nt(NtBatch nb) {
  constexpr int TM = 128, TN = 64 * WTN;             // tile; 2 x 2 waves, each 64 x (32 WTN)
  constexpr int ES = BF ? 2 : 4;                     // bytes per element
  constexpr int KC = 128 / ES;                       // elements per chunk: one 128-byte row piece
  constexpr int STAGE = (TM + TN) * 128;             // bytes per LDS stage
  constexpr int NA = TM / 32, NB = TN / 32;          // global_load_lds instructions per wave per chunk (8 rows each)
  extern __shared__ __attribute__((aligned(16))) char smem[];      // the ONLY LDS object of the kernel: [2][TM + TN][128 B]
  const NtArgs& g = nb.a[blockIdx.z / nb.splits];
  const int zs = blockIdx.z % nb.splits;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = blockIdx.y * TM, j0 = blockIdx.x * TN;
  if (i0 >= g.M || j0 >= g.N) return;                // the grid covers the largest problem of the batch
  const int kbeg = zs * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  const int nc = (kend - kbeg) / KC;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * (32 * WTN);
  // ---- loader: wave w fetches row blocks w NA .. w NA + NA - 1 of A and w NB .. of B; lane -> (row lane >> 3 of the block, slot lane & 7)
  const char* ga[NA];
  const char* gb[NB];
#pragma unroll
  for (int p = 0; p < NA; ++p) {
    const int row = (wave * NA + p) * 8 + (lane >> 3);
    const int src = (lane & 7) ^ ((row >> 1) & 7);   // the piece that belongs into this lane's slot
    ga[p] = static_cast<const char*>(g.A) + ((size_t)min(i0 + row, g.M - 1) * g.lda + kbeg) * ES + src * 16;   // rows past M: any valid row (never stored)
  }
#pragma unroll
  for (int p = 0; p < NB; ++p) {
    const int row = (wave * NB + p) * 8 + (lane >> 3);
    const int src = (lane & 7) ^ ((row >> 1) & 7);
    gb[p] = static_cast<const char*>(g.B) + ((size_t)min(j0 + row, g.N - 1) * g.ldb + kbeg) * ES + src * 16;
  }
  auto issue = [&](int c, int s) {
    char* sa = smem + s * STAGE;
#pragma unroll
    for (int p = 0; p < NA; ++p)
      __builtin_amdgcn_global_load_lds(SDX_AS_GLOBAL(ga[p] + (size_t)c * 128), SDX_AS_LDS(sa + (wave * NA + p) * 1024), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < NB; ++p)
      __builtin_amdgcn_global_load_lds(SDX_AS_GLOBAL(gb[p] + (size_t)c * 128), SDX_AS_LDS(sa + TM * 128 + (wave * NB + p) * 1024), 16, 0, 0);
  };
  f32x16 acc[2][WTN];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < WTN; ++v)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[u][v][i] = 0.0f;
  f32x16 accb[2];                                    // EPI_TN, first column block: A x ones = row sums of A
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int i = 0; i < 16; ++i) accb[u][i] = 0.0f;
  // EPI_TN: the workgroups of the first column block also accumulate A x ones (row sums of A).  The choice is made ONCE, outside the
  // reduction loop, between two copies of the loop: a branch between the MFMAs of a 2-accumulator wave (the 128 x 64 tile in fp32) gave
  // wrong sums in the waves that skipped the extra MFMAs on gfx950 (tools/diag_gemm_nt.py; the accumulator whose MFMA sits right before
  // the taken branch was off by a few per cent) - the loop bodies below are branch-free.
  const bool do_rs = EPI == EPI_TN && blockIdx.x == 0;
  // fragment addresses: row (l & 31) of a 32-row block, piece (2 t + (l >> 5)) ^ key, key = ((l & 31) >> 1) & 7 (block bases are multiples of 32)
  const int roff = (lane & 31) * 128;
  const int off0 = ((lane >> 5) ^ ((lane >> 1) & 7)) << 4;
  auto chunks = [&](auto rs_tag) {
    constexpr bool RS = decltype(rs_tag)::value;
    if (nc > 0) issue(0, 0);
    for (int c = 0; c < nc; ++c) {
      SDX_WAIT_VMCNT0();                                // this wave's pieces of chunk c have landed ...
      __syncthreads();                                  // ... everybody's have, and everybody is done reading the other stage
      if (c + 1 < nc) issue(c + 1, (c + 1) & 1);
      const char* sa = smem + (c & 1) * STAGE + wm * 128 + roff;
      const char* sb = smem + (c & 1) * STAGE + TM * 128 + wn * 128 + roff;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int off = off0 ^ (t << 5);
        if constexpr (BF != 0) {
          bf16x8 a[2], b[WTN];
#pragma unroll
          for (int u = 0; u < 2; ++u) a[u] = *reinterpret_cast<const bf16x8*>(sa + u * 4096 + off);
#pragma unroll
          for (int v = 0; v < WTN; ++v) b[v] = *reinterpret_cast<const bf16x8*>(sb + v * 4096 + off);
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < WTN; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[v], acc[u][v], 0, 0, 0);
          if constexpr (RS) {
            bf16x8 one;
#pragma unroll
            for (int i = 0; i < 8; ++i) one[i] = (__bf16)1.0f;
#pragma unroll
            for (int u = 0; u < 2; ++u) accb[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], one, accb[u], 0, 0, 0);
          }
        } else {
          f32x4 a[2], b[WTN];
#pragma unroll
          for (int u = 0; u < 2; ++u) a[u] = *reinterpret_cast<const f32x4*>(sa + u * 4096 + off);
#pragma unroll
          for (int v = 0; v < WTN; ++v) b[v] = *reinterpret_cast<const f32x4*>(sb + v * 4096 + off);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
              for (int v = 0; v < WTN; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][q], b[v][q], acc[u][v], 0, 0, 0);
            if constexpr (RS) {
#pragma unroll
              for (int u = 0; u < 2; ++u) accb[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][q], 1.0f, accb[u], 0, 0, 0);
            }
          }
        }
      }
    }
  };
  if (do_rs) chunks(std::true_type{}); else chunks(std::false_type{});   // (workgroup-uniform: every wave of a workgroup runs the same copy)
  // ---- epilogue.  C/D layout: lane l holds column (l & 31), rows (r & 3) + 8 (r >> 2) + 4 (l >> 5), r = 0..15, of each 32 x 32 block
  typedef typename std::conditional<BF != 0, __bf16, float>::type elem_t;
  if (EPI == EPI_TN && do_rs && (wave & 1) == 0 && (lane & 31) == 0) {   // every column of A x ones is the row sum: lanes 0 and 32 of the waves at column 0 hold all rows
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M) g.rowsum[(size_t)zs * g.cz + row] = accb[u][r];
      }
  }
  float* Cf = g.Cf ? g.Cf + (size_t)zs * g.cz : nullptr;
  elem_t* Cn = static_cast<elem_t*>(g.Cn);
  elem_t* Ct = static_cast<elem_t*>(g.Ct);
#pragma unroll
  for (int v = 0; v < WTN; ++v) {
    const int col = j0 + wn + 32 * v + (lane & 31);
    if (col >= g.N) continue;
    const float bias = EPI == EPI_FWD ? g.bias[col] : 0.0f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) {                  // four consecutive rows r0 .. r0 + 3 per q: one 8- / 16-byte piece of the transposed copy
        const int r0 = i0 + wm + 32 * u + 8 * q + 4 * (lane >> 5);
        float x[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[e] = acc[u][v][4 * q + e] + bias;
          if (EPI == EPI_FWD) x[e] = belu(x[e]);
          if (EPI == EPI_NN) x[e] *= (r0 + e < g.M) ? belu_grad_from_out(g.H[(size_t)(r0 + e) * g.ldh + col]) : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (r0 + e < g.M) {
            if (Cf) Cf[(size_t)(r0 + e) * g.ldc + col] = x[e];
            if (Cn) Cn[(size_t)(r0 + e) * g.ldn + col] = (elem_t)x[e];
          }
        if (Ct) {
          elem_t* d = Ct + (size_t)col * g.ldt + r0;
          if (r0 + 3 < g.M) {
            if constexpr (BF != 0) {
              bf16x4 p4;
              p4[0] = (__bf16)x[0]; p4[1] = (__bf16)x[1]; p4[2] = (__bf16)x[2]; p4[3] = (__bf16)x[3];
              *reinterpret_cast<bf16x4*>(d) = p4;    // ldt and r0 are multiples of 4: 8-byte aligned
            } else {
              *reinterpret_cast<float4*>(d) = make_float4(x[0], x[1], x[2], x[3]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (r0 + e < g.M) d[e] = (elem_t)x[e];
          }
        }
      }
  }
}

// ---- staging: src fp32 [R][K] (row stride lds) -> dn [R][ldn] in the element type with columns K .. Kp zeroed, and / or its transpose
// dt [Kp][ldt] (rows K .. Kp zeroed; columns R .. ldt are never written: they stay as allocated, i.e. zero).  64 x 64 tiles through LDS.
struct StageArgs { const float* src; int lds, R, K, Kp; void* dn; int ldn; void* dt; int ldt; };
struct StageBatch { StageArgs a[9]; };
template <int BF>
__global__ __launch_bounds__(256) void k_stage(StageBatch sb) {
  typedef typename std::conditional<BF != 0, __bf16, float>::type elem_t;
  __shared__ float tile[64][65];
  const StageArgs& g = sb.a[blockIdx.z];
  const int r0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  if (r0 >= g.R || k0 >= g.Kp) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  elem_t* dn = static_cast<elem_t*>(g.dn);
  elem_t* dt = static_cast<elem_t*>(g.dt);
#pragma unroll 4
  for (int rr = ty; rr < 64; rr += 4) {
    const int r = r0 + rr, k = k0 + tx;
    const float x = (r < g.R && k < g.K) ? g.src[(size_t)r * g.lds + k] : 0.0f;
    tile[rr][tx] = x;
    if (dn && r < g.R && k < g.Kp) dn[(size_t)r * g.ldn + k] = (elem_t)x;
  }
  if (!dt) return;
  __syncthreads();
#pragma unroll 4
  for (int kk = ty; kk < 64; kk += 4) {
    const int k = k0 + kk, r = r0 + tx;
    if (k < g.Kp && r < g.R) dt[(size_t)k * g.ldt + r] = (elem_t)tile[tx][kk];
  }
}

template <int BF, int EPI>
static void gemm_nt(const NtArgs* gs, int count, int splits, hipStream_t st) {
  NtBatch nb;
  int Mx = 0, Nx = 0;
  for (int q = 0; q < 3; ++q) {
    nb.a[q] = gs[q < count ? q : 0];
    if (q < count) { Mx = gs[q].M > Mx ? gs[q].M : Mx; Nx = gs[q].N > Nx ? gs[q].N : Nx; }
  }
  nb.splits = splits;
  // 128 x 128 tiles (least LDS / L2 traffic per flop) when they give every CU's two workgroup slots something to do, else 128 x 64
  const long b2 = (long)((Nx + 127) / 128) * ((Mx + 127) / 128) * splits * count;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_nt<BF, EPI, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_nt<BF, EPI, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 192 * 128);
    attr = true;
  }
  static const int forced = getenv("SDXP_NT_TILE") ? atoi(getenv("SDXP_NT_TILE")) : 0;   // 1: 128 x 64, 2: 128 x 128 (timing / diagnosis)
  if (forced == 2 || (forced == 0 && b2 >= 512)) {
    dim3 grid((Nx + 127) / 128, (Mx + 127) / 128, splits * count);
    hipLaunchKernelGGL((k_gemm_nt<BF, EPI, 2>), grid, dim3(256), 2 * 256 * 128, st, nb);
  } else {
    dim3 grid((Nx + 63) / 64, (Mx + 127) / 128, splits * count);
    hipLaunchKernelGGL((k_gemm_nt<BF, EPI, 1>), grid, dim3(256), 2 * 192 * 128, st, nb);
  }
}
