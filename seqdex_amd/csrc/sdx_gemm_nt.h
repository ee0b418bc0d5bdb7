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
//     barrier per chunk; two workgroups per CU (64 KB each) overlap each other's barrier bubbles.  All 16 fragments of a chunk are
//     requested before its first MFMA (one LDS latency per chunk).
//   * fragments: lane l reads 16 B of row (l & 31) at piece 2 t + (l >> 5), t = 0..3.  bf16: the 8 reduction indices of one
//     32x32x16 step.  fp32: 4 consecutive indices - component c of the low / high half-wave is index 8 t + c / 8 t + 4 + c, fed to
//     four 32x32x2 steps; A and B use the same assignment, so every index meets its partner.
//   * 1-D grid, workgroup -> tile through an XCD-aware remap: each of the 8 L2s serves one contiguous run of tiles (x fastest), so an
//     A row band is fetched into one L2 instead of eight.
//   * epilogues write what the next product reads, and every copy leaves through LDS as 16-byte pieces of long contiguous runs:
//     EPI_FWD bias + ELU -> fp32 [i][j] (fp32 runs, and the last trunk layer of bf16 runs: the fp32 heads read it) and / or the
//     element-type copy, + the transposed copy; EPI_NN times ELU'(layer output, element type) -> element-type copy + transposed copy;
//     EPI_TN split partials of G = dY^T X and, on the first column block, the row sums of A (= the bias gradient) by VALU adds of the
//     A fragments (v_dot2c_f32_bf16 against ones for bf16) beside the MFMAs.
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
  const void* H; int ldh;        // EPI_NN: the layer OUTPUT (element type, [i][j]) whose ELU' multiplies the result
  const void* Ht; int ldht;      // EPI_NN: its transposed copy [j][i] (the transposed result is multiplied from this one: 16-byte pieces again)
  float* rowsum;                 // EPI_TN: rowsum[z cz + i] = sum_k A[i][k] over the split
};
struct NtBatch { NtArgs a[3]; int splits; };

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// sum of the 4 / 8 elements of one operand fragment (VALU work that runs beside the MFMAs)
__device__ __forceinline__ float frag_sum(f32x4 a) { return (a[0] + a[1]) + (a[2] + a[3]); }
__device__ __forceinline__ float frag_sum(bf16x8 a) {
#ifdef HIPEMU
  float s = 0.0f;
  for (int i = 0; i < 8; ++i) s += (float)a[i];
  return s;
#else
  bf16x2 one; one[0] = (__bf16)1.0f; one[1] = (__bf16)1.0f;
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { bf16x2 t; t[0] = a[2 * i]; t[1] = a[2 * i + 1]; s = __builtin_amdgcn_fdot2_f32_bf16(t, one, s, false); }   // v_dot2c_f32_bf16
  return s;
#endif
}

template <int BF, int EPI, int WTN>
__global__ __launch_bounds__(256, 2) void k_gemm_nt(NtBatch nb, int nx, int ny) {
  typedef typename std::conditional<BF != 0, __bf16, float>::type elem_t;
  typedef typename std::conditional<BF != 0, bf16x8, f32x4>::type frag_t;
  constexpr int TM = 128, TN = 64 * WTN;             // tile; 2 x 2 waves, each 64 x (32 WTN)
  constexpr int ES = BF ? 2 : 4;                     // bytes per element
  constexpr int KC = 128 / ES;                       // elements per chunk: one 128-byte row piece
  constexpr int STAGE = (TM + TN) * 128;             // bytes per LDS stage
  constexpr int NA = TM / 32, NB = TN / 32;          // global_load_lds instructions per wave per chunk (8 rows each)
  extern __shared__ __attribute__((aligned(16))) char smem[];      // the ONLY LDS object of the kernel: [2][TM + TN][128 B], reused by the epilogue
  // ---- workgroup -> tile.  The dispatcher hands consecutive workgroup ids to the 8 XCDs in turn (MI355X_MICROARCH.md; a speed
  // assumption only); the remap below gives every XCD one contiguous run of tile indices, x fastest, so that the workgroups that
  // share an A row band (and, over a few bands, the B column bands) meet in ONE L2 instead of fetching it eight times.
  const int total = (int)gridDim.x;
  int wg;
  {
    const int id = (int)blockIdx.x, q = total >> 3, r = total & 7, xcd = id & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int bx = wg % nx, by = (wg / nx) % ny, bz = wg / (nx * ny);
  const NtArgs& g = nb.a[bz / nb.splits];
  const int zs = bz % nb.splits;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = by * TM, j0 = bx * TN;
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
  // EPI_TN: the workgroups of the first column block also sum the rows of A (the bias gradient of a dY^T X product): VALU adds of the
  // fragments the MFMAs consume anyway (v_dot2c_f32_bf16 against ones for bf16) - beside the matrix pipe, not on it.  The choice is made
  // ONCE, outside the reduction loop, between two copies of the loop: with a branch between the MFMAs of a 2-accumulator wave (the
  // 128 x 64 tile in fp32) the waves that skipped the extra work got wrong sums on gfx950 (tools/diag_gemm_nt.py,
  // profiles/r4_gemm_nt_diag_branch_between_mfmas.txt) - the loop bodies are branch-free.
  float rs[2] = {0.0f, 0.0f};
  const bool do_rs = EPI == EPI_TN && bx == 0;
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
      // all fragments of the chunk are requested before the first MFMA: one LDS latency per chunk instead of one per 16-wide step
      frag_t a[4][2], b[4][WTN];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int off = off0 ^ (t << 5);
#pragma unroll
        for (int u = 0; u < 2; ++u) a[t][u] = *reinterpret_cast<const frag_t*>(sa + u * 4096 + off);
#pragma unroll
        for (int v = 0; v < WTN; ++v) b[t][v] = *reinterpret_cast<const frag_t*>(sb + v * 4096 + off);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (BF != 0) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < WTN; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][u], b[t][v], acc[u][v], 0, 0, 0);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
              for (int v = 0; v < WTN; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][u][q], b[t][v][q], acc[u][v], 0, 0, 0);
        }
#ifdef SDX_GEMM_NT_BRANCHY          // diagnostic build only: the round-4 form with a (workgroup-uniform) branch between the MFMAs of a k step
        if (do_rs) {
#pragma unroll
          for (int u = 0; u < 2; ++u) rs[u] += frag_sum(a[t][u]);
        }
#else
        if constexpr (RS) {
#pragma unroll
          for (int u = 0; u < 2; ++u) rs[u] += frag_sum(a[t][u]);
        }
#endif
      }
    }
  };
#ifdef SDX_GEMM_NT_BRANCHY
  chunks(std::false_type{});
#else
  if (do_rs) chunks(std::true_type{}); else chunks(std::false_type{});   // (workgroup-uniform: every wave of a workgroup runs the same copy)
#endif
  // ---- epilogue.  C/D layout: lane l holds column (l & 31), rows (r & 3) + 8 (r >> 2) + 4 (l >> 5), r = 0..15, of each 32 x 32 block
  if constexpr (EPI == EPI_TN) {
    if (do_rs) {                                     // lanes l and l + 32 hold the two halves of row (l & 31)'s sum over this split's k range
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float t = rs[u] + __shfl_xor(rs[u], 32, 64);
        const int row = i0 + wm + 32 * u + (lane & 31);
        if ((wave & 1) == 0 && lane < 32 && row < g.M) g.rowsum[(size_t)zs * g.cz + row] = t;
      }
    }
    float* Cf = g.Cf + (size_t)zs * g.cz;            // split partials: small, written straight from the accumulators (128-byte runs)
#pragma unroll
    for (int v = 0; v < WTN; ++v) {
      const int col = j0 + wn + 32 * v + (lane & 31);
      if (col >= g.N) continue;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i0 + wm + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < g.M) Cf[(size_t)row * g.ldc + col] = acc[u][v][r];
        }
    }
    return;
  } else {
    // Both copies leave through LDS so that every global access of the epilogue is a 16-byte piece of a long contiguous run (straight
    // from the accumulator layout the transposed copy would be 8-byte pieces, 64 lanes on 64 lines).  EPI_FWD adds bias + ELU in
    // registers; EPI_NN stages the raw sums and multiplies by ELU'(layer output) on the way out, reading the layer output in the SAME
    // orientation as the copy being written (H for [i][j], Ht for [j][i]: one 16- / 32-byte load per piece instead of 64 two-byte loads
    // per lane, which made the short-K data-gradient products wait 72 % of their wave-cycles; profiles/r4_bigmb_*_pmc_sq.csv).
    if constexpr (EPI == EPI_FWD) {
#pragma unroll
      for (int v = 0; v < WTN; ++v) {
        const int col = j0 + wn + 32 * v + (lane & 31);
        const float bias = col < g.N ? g.bias[col] : 0.0f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[u][v][r] = belu(acc[u][v][r] + bias);
      }
    }
    const elem_t* Hm = static_cast<const elem_t*>(g.H);
    const elem_t* Htm = static_cast<const elem_t*>(g.Ht);
    float* Cf = g.Cf;
    elem_t* Cn = static_cast<elem_t*>(g.Cn);
    elem_t* Ct = static_cast<elem_t*>(g.Ct);
    constexpr int EPP = 16 / ES;                     // elements per 16-byte piece of an element-type array
    SDX_LDS_BARRIER();                               // every wave is done with the operand stages (LDS-only barriers from here on: sdx_common.h)
    if (Cf || Cn) {
      // normal orientation: fp32 tile [TM][TN] in LDS (64 KB for the 128 x 128 tile = both stages); 8 consecutive columns per thread
      float* T = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int v = 0; v < WTN; ++v)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            T[(wm + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TN + wn + 32 * v + (lane & 31)] = acc[u][v][r];
      SDX_LDS_BARRIER();
      constexpr int TPR = TN / 8, RPP = 256 / TPR, NP = TM / RPP;   // threads per tile row, rows per pass, passes
      const int c0 = (tid % TPR) * 8, col = j0 + c0, rr0 = tid / TPR;
      const bool whole = col + 8 <= g.N;
      float hm[NP][8];                               // ELU' factors of the thread's pieces, requested before the first store
      if constexpr (EPI == EPI_NN) {
        const bool hvec = whole && (g.ldh % 8) == 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int row = i0 + rr0 + p * RPP;
#pragma unroll
          for (int e = 0; e < 8; ++e) hm[p][e] = 0.0f;
          if (row < g.M && col < g.N) {
            const elem_t* hp = Hm + (size_t)row * g.ldh + col;
            if (hvec) {
              if constexpr (BF != 0) {
                const bf16x8 h8 = *reinterpret_cast<const bf16x8*>(hp);
#pragma unroll
                for (int e = 0; e < 8; ++e) hm[p][e] = (float)h8[e];
              } else {
                const float4 h0 = *reinterpret_cast<const float4*>(hp), h1 = *reinterpret_cast<const float4*>(hp + 4);
                hm[p][0] = h0.x; hm[p][1] = h0.y; hm[p][2] = h0.z; hm[p][3] = h0.w; hm[p][4] = h1.x; hm[p][5] = h1.y; hm[p][6] = h1.z; hm[p][7] = h1.w;
              }
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) if (col + e < g.N) hm[p][e] = (float)hp[e];
            }
          }
        }
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int rr = rr0 + p * RPP, row = i0 + rr;
        if (row >= g.M || col >= g.N) continue;
        const float4 x0 = *reinterpret_cast<const float4*>(&T[rr * TN + c0]), x1 = *reinterpret_cast<const float4*>(&T[rr * TN + c0 + 4]);
        float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        if constexpr (EPI == EPI_NN) {
#pragma unroll
          for (int e = 0; e < 8; ++e) xs[e] *= belu_grad_from_out(hm[p][e]);
        }
        if (whole) {
          if (Cf) {
            float* d = Cf + (size_t)row * g.ldc + col;
            if ((g.ldc & 3) == 0) { *reinterpret_cast<float4*>(d) = make_float4(xs[0], xs[1], xs[2], xs[3]); *reinterpret_cast<float4*>(d + 4) = make_float4(xs[4], xs[5], xs[6], xs[7]); }
            else {
#pragma unroll
              for (int e = 0; e < 8; ++e) d[e] = xs[e];
            }
          }
          if (Cn) {
            elem_t* d = Cn + (size_t)row * g.ldn + col;
            if constexpr (BF != 0) {
              if ((g.ldn & 7) == 0) {
                bf16x8 p8;
#pragma unroll
                for (int e = 0; e < 8; ++e) p8[e] = (__bf16)xs[e];
                *reinterpret_cast<bf16x8*>(d) = p8;
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = (elem_t)xs[e];
              }
            } else {
              if ((g.ldn & 3) == 0) { *reinterpret_cast<float4*>(d) = make_float4(xs[0], xs[1], xs[2], xs[3]); *reinterpret_cast<float4*>(d + 4) = make_float4(xs[4], xs[5], xs[6], xs[7]); }
              else {
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = xs[e];
              }
            }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (col + e < g.N) {
              if (Cf) Cf[(size_t)row * g.ldc + col + e] = xs[e];
              if (Cn) Cn[(size_t)row * g.ldn + col + e] = (elem_t)xs[e];
            }
        }
      }
      if (Ct) SDX_LDS_BARRIER();                     // the transposed image reuses the same LDS
    }
    if (Ct) {
      // transposed orientation, 64 tile columns per pass (the columns of the waves with (wave & 1) == pass for the 128-wide tile): fp32
      // image [64][TM + 4] (row stride 528 B: the 8 lanes of a 16-byte store group land on distinct banks), written from the
      // accumulators as 4 consecutive rows per register quad
      constexpr int LDT = TM + 4;
      float* TT = reinterpret_cast<float*>(smem);
      constexpr int PPR = TM / EPP, CPI = 256 / PPR, NI = 64 / CPI;   // pieces per image row, image rows (= tile columns) per iteration, iterations
      const int m0 = (tid % PPR) * EPP, row = i0 + m0, cc0 = tid / PPR;
      const bool wholer = row + EPP <= g.M && (g.ldt % EPP) == 0;
#pragma unroll
      for (int pass = 0; pass < WTN; ++pass) {
        if (WTN == 1 || (wave & 1) == pass) {
#pragma unroll
          for (int v = 0; v < WTN; ++v) {
            const int cl = (WTN == 1 ? wn : 0) + 32 * v + (lane & 31);        // column inside this pass's 64
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
              for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(TT + cl * LDT + wm + 32 * u + 8 * q + 4 * (lane >> 5)) =
                    make_float4(acc[u][v][4 * q], acc[u][v][4 * q + 1], acc[u][v][4 * q + 2], acc[u][v][4 * q + 3]);
          }
        }
        float ht[NI][EPP];
        if constexpr (EPI == EPI_NN) {                 // (requested before the barrier: the loads fly while the image is written)
          const bool hvec = wholer && (g.ldht % EPP) == 0;
#pragma unroll
          for (int it = 0; it < NI; ++it) {
            const int colg = j0 + (WTN == 1 ? 0 : 64 * pass) + cc0 + it * CPI;
#pragma unroll
            for (int e = 0; e < EPP; ++e) ht[it][e] = 0.0f;
            if (colg < g.N && row < g.M) {
              const elem_t* hp = Htm + (size_t)colg * g.ldht + row;
              if (hvec) {
                if constexpr (BF != 0) {
                  const bf16x8 h8 = *reinterpret_cast<const bf16x8*>(hp);
#pragma unroll
                  for (int e = 0; e < 8; ++e) ht[it][e] = (float)h8[e];
                } else {
                  const float4 h4 = *reinterpret_cast<const float4*>(hp);
                  ht[it][0] = h4.x; ht[it][1] = h4.y; ht[it][2] = h4.z; ht[it][3] = h4.w;
                }
              } else {
#pragma unroll
                for (int e = 0; e < EPP; ++e) if (row + e < g.M) ht[it][e] = (float)hp[e];
              }
            }
          }
        }
        SDX_LDS_BARRIER();
#pragma unroll
        for (int it = 0; it < NI; ++it) {
          const int cc = cc0 + it * CPI, colg = j0 + (WTN == 1 ? 0 : 64 * pass) + cc;
          if (colg >= g.N || row >= g.M) continue;
          const float* sp = TT + cc * LDT + m0;
          float xs[EPP];
#pragma unroll
          for (int e4 = 0; e4 < EPP; e4 += 4) {
            const float4 x4 = *reinterpret_cast<const float4*>(sp + e4);
            xs[e4] = x4.x; xs[e4 + 1] = x4.y; xs[e4 + 2] = x4.z; xs[e4 + 3] = x4.w;
          }
          if constexpr (EPI == EPI_NN) {
#pragma unroll
            for (int e = 0; e < EPP; ++e) xs[e] *= belu_grad_from_out(ht[it][e]);
          }
          elem_t* d = Ct + (size_t)colg * g.ldt + row;
          if (wholer) {
            if constexpr (BF != 0) {
              bf16x8 p8;
#pragma unroll
              for (int e = 0; e < 8; ++e) p8[e] = (__bf16)xs[e];
              *reinterpret_cast<bf16x8*>(d) = p8;
            } else {
              *reinterpret_cast<float4*>(d) = make_float4(xs[0], xs[1], xs[2], xs[3]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < EPP; ++e) if (row + e < g.M) d[e] = (elem_t)xs[e];
          }
        }
        if (pass + 1 < WTN) SDX_LDS_BARRIER();
      }
    }
  }
}

// ---- fp32 weight gradient straight from the [row][feature] arrays ("TT": both operands reduction-major).  G[n][k] = sum_r A[r][n] B[r][k]
// over the rows [z kchunk, min(K, (z + 1) kchunk)) of split z: A = dY_l [rows][N_l], B = the layer input [rows][K_l], i.e. the arrays
// the forward and data-gradient products write and read anyway.  With it an fp32 run needs NO transposed copy of an activation or
// gradient (the forward / data-gradient epilogues write one image instead of two, the data gradient reads the layer output once).
//   * a chunk is 32 rows; the tile's A part is 32 x 128 floats (512-byte row pieces), B 32 x (64 WTN) floats; global_load_lds_dwordx4,
//     lane-linear = row-major LDS image; rows past the split's end come from a page of zeros (a per-lane address select: the loop body
//     stays branch-free, see the note on branches between MFMAs above);
//   * v_mfma_f32_32x32x2_f32 takes one reduction index per half-wave: lane (i, h) reads A[2 t + h][i] and B[2 t + h][j] as single dwords -
//     32 consecutive floats per half-wave, conflict-free - 4 reads per 4 MFMAs, all 64 of a chunk requested up front;
//   * NtArgs: A / lda, B / ldb as above, M = N_l (rows of G), N = K_l (columns of G), K = number of rows, kchunk a multiple of 32;
//     Cf / ldc / cz split partials, rowsum = column sums of A (the bias gradient), as EPI_TN.
template <int WTN>
__global__ __launch_bounds__(256, 2) void k_gemm_tt(NtBatch nb, int nx, int ny, const float* zeros) {
  constexpr int TM = 128, TN = 64 * WTN, RC = 32;
  constexpr int ASTAGE = RC * TM * 4, BSTAGE = RC * TN * 4, STAGE = ASTAGE + BSTAGE;
  constexpr int NA = ASTAGE / 1024 / 4, NB = BSTAGE / 1024 / 4;          // global_load_lds instructions per wave per chunk
  constexpr int RPB = 1024 / (TN * 4);                                    // B rows per instruction (2 for the 128-wide tile, 4 for 64)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int total = (int)gridDim.x;
  int wg;
  {
    const int id = (int)blockIdx.x, q = total >> 3, r = total & 7, xcd = id & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int bx = wg % nx, by = (wg / nx) % ny, bz = wg / (nx * ny);
  const NtArgs& g = nb.a[bz / nb.splits];
  const int zs = bz % nb.splits;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = by * TM, j0 = bx * TN;
  if (i0 >= g.M || j0 >= g.N) return;
  const int rbeg = zs * g.kchunk, rend = min(g.K, rbeg + g.kchunk);
  const int nc = (rend - rbeg + RC - 1) / RC;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * (32 * WTN);
  // loader: instruction p of the A part covers rows 2 p, 2 p + 1 of the chunk (lane: row lane >> 5, 16-byte piece lane & 31); columns past
  // the array's width are clamped into it (their products belong to outputs that are never stored)
  const float* Ab = static_cast<const float*>(g.A);
  const float* Bb = static_cast<const float*>(g.B);
  int arow[NA], brow[NB];
  const int acol = min(i0 + 4 * (lane & 31), g.lda - 4);
  const int bcol = min(j0 + 4 * (lane & (TN / 4 - 1)), g.ldb - 4);
#pragma unroll
  for (int p = 0; p < NA; ++p) arow[p] = 2 * (wave * NA + p) + (lane >> 5);
#pragma unroll
  for (int p = 0; p < NB; ++p) brow[p] = RPB * (wave * NB + p) + lane / (TN / 4);
  auto issue = [&](int c, int s) {
    char* sa = smem + s * STAGE;
    const int r0 = rbeg + c * RC;
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int r = r0 + arow[p];
      const float* src = r < rend ? Ab + (size_t)r * g.lda + acol : zeros;
      __builtin_amdgcn_global_load_lds(SDX_AS_GLOBAL(src), SDX_AS_LDS(sa + (wave * NA + p) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int r = r0 + brow[p];
      const float* src = r < rend ? Bb + (size_t)r * g.ldb + bcol : zeros;
      __builtin_amdgcn_global_load_lds(SDX_AS_GLOBAL(src), SDX_AS_LDS(sa + ASTAGE + (wave * NB + p) * 1024), 16, 0, 0);
    }
  };
  f32x16 acc[2][WTN];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < WTN; ++v)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[u][v][i] = 0.0f;
  float rs[2] = {0.0f, 0.0f};
  const bool do_rs = bx == 0;
  const int foff = ((lane >> 5) * TM + wm + (lane & 31)) * 4, goff = ((lane >> 5) * TN + wn + (lane & 31)) * 4;
  auto chunks = [&](auto rs_tag) {
    constexpr bool RS = decltype(rs_tag)::value;
    if (nc > 0) issue(0, 0);
    for (int c = 0; c < nc; ++c) {
      SDX_WAIT_VMCNT0();
      __syncthreads();
      if (c + 1 < nc) issue(c + 1, (c + 1) & 1);
      const char* sa = smem + (c & 1) * STAGE + foff;
      const char* sb = smem + (c & 1) * STAGE + ASTAGE + goff;
      float a[16][2], b[16][WTN];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
#pragma unroll
        for (int u = 0; u < 2; ++u) a[t][u] = *reinterpret_cast<const float*>(sa + t * (2 * TM * 4) + u * 128);
#pragma unroll
        for (int v = 0; v < WTN; ++v) b[t][v] = *reinterpret_cast<const float*>(sb + t * (2 * TN * 4) + v * 128);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int v = 0; v < WTN; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][u], b[t][v], acc[u][v], 0, 0, 0);
        if constexpr (RS) {
#pragma unroll
          for (int u = 0; u < 2; ++u) rs[u] += a[t][u];
        }
      }
    }
  };
  if (do_rs) chunks(std::true_type{}); else chunks(std::false_type{});
  if (do_rs) {                                       // lanes l and l + 32 hold the sums over the even / odd rows of column (l & 31)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float t = rs[u] + __shfl_xor(rs[u], 32, 64);
      const int row = i0 + wm + 32 * u + (lane & 31);
      if ((wave & 1) == 0 && lane < 32 && row < g.M) g.rowsum[(size_t)zs * g.cz + row] = t;
    }
  }
  float* Cf = g.Cf + (size_t)zs * g.cz;
#pragma unroll
  for (int v = 0; v < WTN; ++v) {
    const int col = j0 + wn + 32 * v + (lane & 31);
    if (col >= g.N) continue;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M) Cf[(size_t)row * g.ldc + col] = acc[u][v][r];
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
  // 128 x 128 tiles (least LDS / L2 traffic per flop) when they give every CU's two workgroup slots something to do and do not pad the
  // columns by much more than 128 x 64 tiles would (a 396-wide weight gradient: 512 vs 448 columns of work), else 128 x 64
  const int ny = (Mx + 127) / 128, nx2 = (Nx + 127) / 128, nx1 = (Nx + 63) / 64;
  const long b2 = (long)nx2 * ny * splits * count;
  static const int forced = getenv("SDXP_NT_TILE") ? atoi(getenv("SDXP_NT_TILE")) : 0;   // 1: 128 x 64, 2: 128 x 128 (timing / diagnosis)
  const bool wide = forced ? forced == 2 : (b2 >= 512 && nx2 * 128 <= nx1 * 64 * 1.10);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_nt<BF, EPI, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_nt<BF, EPI, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 192 * 128);
    attr = true;
  }
  const int nz = splits * count;
  if (wide) hipLaunchKernelGGL((k_gemm_nt<BF, EPI, 2>), dim3(nx2 * ny * nz), dim3(256), 2 * 256 * 128, st, nb, nx2, ny);
  else hipLaunchKernelGGL((k_gemm_nt<BF, EPI, 1>), dim3(nx1 * ny * nz), dim3(256), 2 * 192 * 128, st, nb, nx1, ny);
}

// launcher of k_gemm_tt: tile choice as gemm_nt; `zeros`: 16-byte aligned device memory holding at least 16 zero bytes
static void gemm_tt(const NtArgs* gs, int count, int splits, const float* zeros, hipStream_t st) {
  NtBatch nb;
  int Mx = 0, Nx = 0;
  for (int q = 0; q < 3; ++q) {
    nb.a[q] = gs[q < count ? q : 0];
    if (q < count) { Mx = gs[q].M > Mx ? gs[q].M : Mx; Nx = gs[q].N > Nx ? gs[q].N : Nx; }
  }
  nb.splits = splits;
  const int ny = (Mx + 127) / 128, nx2 = (Nx + 127) / 128, nx1 = (Nx + 63) / 64;
  const long b2 = (long)nx2 * ny * splits * count;
  static const int forced = getenv("SDXP_NT_TILE") ? atoi(getenv("SDXP_NT_TILE")) : 0;
  const bool wide = forced ? forced == 2 : (b2 >= 512 && nx2 * 128 <= nx1 * 64 * 1.10);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tt<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * 256 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tt<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * 192 * 4);
    attr = true;
  }
  const int nz = splits * count;
  if (wide) hipLaunchKernelGGL((k_gemm_tt<2>), dim3(nx2 * ny * nz), dim3(256), 2 * 32 * 256 * 4, st, nb, nx2, ny, zeros);
  else hipLaunchKernelGGL((k_gemm_tt<1>), dim3(nx1 * ny * nz), dim3(256), 2 * 32 * 192 * 4, st, nb, nx1, ny, zeros);
}
