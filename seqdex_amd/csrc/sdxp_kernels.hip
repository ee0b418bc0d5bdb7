// sdxp_kernels.hip — PPO hot loops of rl_games' A2CAgent as hand-written HIP for gfx950.
//
// Rollout (M = num_envs rows): fp32 MFMA GEMMs  Y = ELU(X W^T + b)  (v_mfma_f32_32x32x2_f32, LDS-tiled,
// 64x64 block tile, 4 waves of 32x32), head kernel = mu / value / Gaussian sample / neglogp (RC:1697-1723,2114-2126).
//
// Update with the shipped minibatch_size 4 (YG:50,75): a minibatch gradient of a Linear layer is the rank-MB
// product dY^T X, so it is never materialised:
//   * the optimiser step of minibatch i is applied LAZILY inside the forward of minibatch i+1: the layer
//     kernel streams each weight row ONCE (w, m, v in; w, m, v out), rebuilds g[n][k] = sum_s dY[s][n] X[s][k]
//     from the rank-MB factors kept in LDS, applies clip-scale + Adam and immediately uses the new row for
//     the forward dot products (wave per row, lanes along K -> coalesced, wave reduction);
//   * the global gradient norm (clip_grad_norm_, RC:1859-1877) comes from the MB x MB Gram matrices of the
//     factors: |dY^T X|_F^2 = sum_{s,s'} (dY_s . dY_s')(X_s . X_s');
//   * actor, critic and central-value networks advance in the SAME launches (they are independent given the
//     dataset; rl_games runs train_central_value first, RC:1323-1324 - interleaving is arithmetically identical).
// Per optimiser step: L1, L2, L3, HEAD(+loss +backward of the heads), B3, B2, CTRL = 7 launches for all three nets.
#include "sdx_common.h"
#include "sdxp_types.h"
#include <cstddef>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float elu(float x) { return x > 0.0f ? x : expm1f(x); }
// wave64 sum with DPP cross-lane moves (VALU rate) instead of ds_bpermute-based __shfl_xor butterflies: quad_perm xor 1,
// xor 2, row_half_mirror, row_mirror give every lane its 16-lane row sum; row_bcast15 / row_bcast31 fold the four rows
// into lane 63, which is broadcast with v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
  return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);   // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);   // row_mirror
  v = dpp_add<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float lane0(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

// ------------------------------------------------------------------------------------------------ rollout GEMM
// Y[M,N] = act(X[M,K] W[N,K]^T + b[N]); act = ELU when elu_flag.  X may be normalised on the fly with (mean, rstd)
// (central-value running mean/std, clamp +-5; App. C of SURVEY.md).  K % 4 == 0.
// One launch serves up to TWO such products (blockIdx.z): the same layer of the actor and of the central-value trunk during a rollout.
// 64 x 64 output tile per workgroup (4 waves, one 32 x 32 v_mfma_f32_32x32x2_f32 tile each), reduction chunks of 32.  The operands of chunk
// c + 2 are in flight from HBM / L2 (registers) while chunk c is multiplied out of one LDS buffer and chunk c + 1 is written into the other:
// ONE barrier per chunk and two chunks of load latency hidden.  (Round 2's kernel loaded, stored, synchronised and multiplied one
// 16-wide chunk at a time: 25 dependent round trips for the 396-wide first layer, 35 us per launch, 6 launches per env step.)
#define GT 64
#define SDXP_NORM_K 1024   // widest input that may be normalised on the fly (checked by the launchers)
struct LinArgs { const float* X; const float* W; const float* b; float* Y; int M, N, K, elu; const double* nmean; const double* nvar; };
struct LinBatch { LinArgs a[2]; };
// WTM = 1: 64 x 64 tile (wave = 32 x 32); WTM = 2: 128 x 64 tile (wave = 64 x 32: two MFMAs share one W operand - 21 instead of 16 flops
// per operand byte fetched from L2).  GKc = reduction chunk staged per barrier.  KS = split of every chunk over KS groups of four waves
// (256 KS threads): group g multiplies the k range [g GKc / KS, (g + 1) GKc / KS) of each chunk into its own accumulators and the groups
// are summed through LDS, in group order, at the end.  At M = 1024 the middle and last layers launch only 256 / 128 workgroups, one wave
// per SIMD: every LDS read and every barrier was exposed (matrix pipe 26 % busy, profiles/r3_act_pmc.csv); with KS groups a SIMD holds
// KS waves whose reads and multiplies overlap, and each thread moves 1 / KS of the operands.
template <int WTM, int KS, int GKc>
__global__ __launch_bounds__(256 * KS) void k_linear_mfma(LinBatch lb) {
  constexpr int TM = GT * WTM, NT = 256 * KS, F4R = GKc / 4;       // rows of X per tile, threads, float4 pieces per row and chunk
  constexpr int NX = TM * F4R / NT, NW = GT * F4R / NT, KG = GKc / KS;
  static_assert(NX >= 1 && NW >= 1 && TM * F4R % NT == 0 && GT * F4R % NT == 0 && KG % 2 == 0, "chunk does not divide over the threads");
  constexpr int OPER = 2 * (TM + GT) * (GKc + 1), RED = (KS - 1) * 4 * 16 * 64;
  __shared__ float lds[OPER > RED ? OPER : RED];
  __shared__ float nmu[SDXP_NORM_K], nsd[SDXP_NORM_K];            // mean and sqrt(var + eps) of the normalised input, as floats
  float (*Xs)[TM][GKc + 1] = reinterpret_cast<float (*)[TM][GKc + 1]>(lds);
  float (*Ws)[GT][GKc + 1] = reinterpret_cast<float (*)[GT][GKc + 1]>(lds + 2 * TM * (GKc + 1));
  const LinArgs& g = lb.a[blockIdx.z];
  const int tid = threadIdx.x, wave = (tid >> 6) & 3, kg = tid >> 8, lane = tid & 63;
  // (An XCD-aware workgroup -> tile map - the tiles cut into 8 rectangles, XCD c working through rectangle c so that it pulls 1/4 of X and
  // 1/2 of W through the fabric instead of all of X - was measured at M = 1024: 87.3 us per sdxp_act against 86.2 with the plain map.
  // These layers are not bound by operand traffic.)
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * GT;
  const int M = g.M, N = g.N, K = g.K;
  if (m0 >= M || n0 >= N) return;                 // the grid covers the larger problem of the batch
  const int wm = (wave >> 1) * 32 * WTM, wn = (wave & 1) * 32;
  f32x16 acc[WTM];
#pragma unroll
  for (int u = 0; u < WTM; ++u)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[u][i] = 0.0f;
  // element e = tid + NT p of a chunk: row e / F4R, k offset 4 (e % F4R): consecutive lanes read consecutive 16-byte pieces of a row.
  // fetch() only issues loads, from clamped (always valid) addresses and without a branch, so that the compiler can count them: with
  // exec-masked loads every wait became vmcnt(0) and the chunk in flight was waited for where it was issued.  stage() zeroes the pieces
  // outside the matrix and applies the running-mean/std normalisation of X from per-k tables built once per workgroup (same arithmetic
  // as normalising at load time)
  float4 xv0[NX], xv1[NX], wv0[NW], wv1[NW];
  auto fetch = [&](int k0, float4 (&xr)[NX], float4 (&wr)[NW]) {
#pragma unroll
    for (int p = 0; p < NX; ++p) {
      const int e = tid + NT * p, r = e / F4R, k = k0 + 4 * (e % F4R);
      const int rr = m0 + r < M ? m0 + r : M - 1, kc = k < K ? k : 0;
      xr[p] = *reinterpret_cast<const float4*>(g.X + (size_t)rr * K + kc);
    }
#pragma unroll
    for (int p = 0; p < NW; ++p) {
      const int e = tid + NT * p, r = e / F4R, k = k0 + 4 * (e % F4R);
      const int rr = n0 + r < N ? n0 + r : N - 1, kc = k < K ? k : 0;
      wr[p] = *reinterpret_cast<const float4*>(g.W + (size_t)rr * K + kc);
    }
  };
  const bool norm = g.nmean != nullptr;
  auto stage = [&](int buf, int k0, const float4 (&xr)[NX], const float4 (&wr)[NW]) {
#pragma unroll
    for (int p = 0; p < NX; ++p) {
      const int e = tid + NT * p, r = e / F4R, k = 4 * (e % F4R);
      const bool in = m0 + r < M && k0 + k < K;
      float xq[4] = {xr[p].x, xr[p].y, xr[p].z, xr[p].w};
      if (norm) {
        const int kt = in ? k0 + k : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) xq[j] = clampf((xq[j] - nmu[kt + j]) / nsd[kt + j], -5.0f, 5.0f);
      }
      float* dx = &Xs[buf][r][k];
#pragma unroll
      for (int j = 0; j < 4; ++j) dx[j] = in ? xq[j] : 0.0f;
    }
#pragma unroll
    for (int p = 0; p < NW; ++p) {
      const int e = tid + NT * p, r = e / F4R, k = 4 * (e % F4R);
      const bool in = n0 + r < N && k0 + k < K;
      float* dw = &Ws[buf][r][k]; dw[0] = in ? wr[p].x : 0.0f; dw[1] = in ? wr[p].y : 0.0f; dw[2] = in ? wr[p].z : 0.0f; dw[3] = in ? wr[p].w : 0.0f;
    }
  };
  // all LDS operands of the chunk are requested before the first multiply (the rolled form read two, waited, multiplied twice)
  auto mma = [&](int buf) {
    float av[WTM][KG / 2], bv[KG / 2];
#pragma unroll
    for (int j = 0; j < KG / 2; ++j) {
      const int kk = kg * KG + 2 * j + (lane >> 5);
      bv[j] = Ws[buf][wn + (lane & 31)][kk];
#pragma unroll
      for (int u = 0; u < WTM; ++u) av[u][j] = Xs[buf][wm + 32 * u + (lane & 31)][kk];
    }
    __builtin_amdgcn_sched_barrier(0);             // (the scheduler otherwise sinks the reads back between the multiplies)
#pragma unroll
    for (int j = 0; j < KG / 2; ++j)
#pragma unroll
      for (int u = 0; u < WTM; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][j], bv[j], acc[u], 0, 0, 0);
  };
  const int nchunk = (K + GKc - 1) / GKc;
  fetch(0, xv0, wv0);
  fetch(GKc, xv1, wv1);
  if (norm) {
    for (int k = tid; k < K; k += NT) { nmu[k] = (float)g.nmean[k]; nsd[k] = sqrtf((float)g.nvar[k] + 1e-5f); }
    __syncthreads();
  }
  stage(0, 0, xv0, wv0);
  __syncthreads();
  // two chunks per trip so that the register slots and LDS buffers are fixed names: chunk c multiplies out of buffer c & 1 while chunk
  // c + 2 is in flight into the slot chunk c came from and chunk c + 1 moves from the other slot into the other buffer; one barrier a chunk
  int c = 0;
  for (; c + 1 < nchunk; c += 2) {
    fetch((c + 2) * GKc, xv0, wv0);
    mma(0);
    stage(1, (c + 1) * GKc, xv1, wv1);
    __syncthreads();
    fetch((c + 3) * GKc, xv1, wv1);
    mma(1);
    stage(0, (c + 2) * GKc, xv0, wv0);
    __syncthreads();
  }
  if (c < nchunk) mma(0);                         // odd number of chunks: the last one sits in buffer 0
  if constexpr (KS > 1) {                          // sum of the k groups, in group order (the operand buffers are reused)
    static_assert(WTM == 1 || KS == 1, "the reduction buffer holds one accumulator tile per wave");
    __syncthreads();                               // an odd last chunk is still being read out of buffer 0
    for (int gsrc = 1; gsrc < KS; ++gsrc) {
      float* red = lds + ((gsrc - 1) * 4 + wave) * 16 * 64;
      if (kg == gsrc)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[i * 64 + lane] = acc[0][i];
    }
    __syncthreads();
    if (kg != 0) return;
    for (int gsrc = 1; gsrc < KS; ++gsrc) {
      const float* red = lds + ((gsrc - 1) * 4 + wave) * 16 * 64;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][i] += red[i * 64 + lane];
    }
  }
  const int col = n0 + wn + (lane & 31);
  if (col < N) {
    const float bias = g.b[col];
#pragma unroll
    for (int u = 0; u < WTM; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M) {
          const float v = acc[u][r] + bias;
          g.Y[(size_t)row * N + col] = g.elu ? elu(v) : v;
        }
      }
  }
}
// ---- the same product with the operands staged by LDS-DMA (round 6).  k_linear_mfma above moves every 16-byte piece
// global -> VGPR -> (normalise / zero) -> four ds_write_b32 into a padded [row][33] image and reads it back one dword at a time: per 32-wide
// chunk a wave issues 2 loads, 8 LDS stores and 16 LDS loads around its 8 MFMAs, in three barrier-separated phases (read, multiply, stage), and
// a chunk takes ~0.9 us where its multiplies need 0.5 (2 waves per SIMD).  Here the pieces go straight into LDS (global_load_lds_dwordx4: no
// staging registers, no LDS stores, no stage phase), into an XOR-swizzled image of 16-byte slots - slot(row, q) = 8 row + (q ^ ((row >> 1) & 7)),
// the swizzle applied on the SOURCE address since the DMA writes base + 16 lane - that ds_read_b128 reads without bank conflicts in its four
// 16-lane groups; a lane gets four k values per read, so a wave issues 4 LDS loads per chunk instead of 16 (the k order inside a chunk is
// permuted - lanes 0-31 take piece 2t, lanes 32-63 piece 2t + 1 of the k group - identically for both operands).  NBUF image pairs rotate:
// wait (counted vmcnt) for chunk c, LDS-only barrier, request chunk c + NBUF - 1 into the pair everybody finished reading before that barrier,
// read, multiply.  The normalisation of a first layer's input is applied to the A operand after the read (same arithmetic: (x - mean) / sd,
// clamp); pieces past K of the last chunk are zeroed in LDS by the lanes that would have loaded them.  ALL LDS of the kernel is one array (a
// second __shared__ object makes hipcc wait vmcnt(0) in front of every ds_read of a DMA pipeline).
// Measured (tools/time_linear.py, profiles/r6_linear_layers_lds_dma.txt): 28.2 / 29.0 / 17.1 us for the three layers against 30.5 / 29.4 / 17.0 with
// k_linear_mfma's best shape, and the same again (28.9 / 28.6 / 16.9) with the next chunk's LDS reads requested under this chunk's multiplies
// (the form below).  Neither the staging nor the LDS reads are what these layers wait for: a chunk costs 0.74 us = its 16 multiplies per SIMD at
// 68 % of the nominal fp32 matrix rate, the ceiling the large GEMMs of sdx_gemm_nt.h also sit at; a launch costs 5.2 us beyond its chunks, three
// times per sdxp_act.  sdxp_act as a whole did not move, so the launcher keeps shape 3; this kernel is shape 7
// (SDXP_LINEAR_TILE=7), held to float64 with every other shape by tests/test_gpu_linear_kernel.py.  A fourth image pair (72 KB of LDS: the tables
// then sit above 64 KB) gave wrong sums in normalised launches - LDS addresses above 64 KB through inline-asm ds_read - and 1 % in time: dropped.
// LDS reads of the DMA pipeline as inline asm: hipcc's wait-count pass treats every ds_read it can see as a possible reader of every LDS-DMA
// in flight and puts s_waitcnt vmcnt(0) in front of it (seen in the first build of this kernel: the chunk requested a line earlier was waited
// for at once, the pipeline was serial).  What it cannot see it does not wait for; the waits are spelled out (lds_wait4).
typedef float f4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4v_t lds_read128(unsigned byte_addr) {
  f4v_t v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
  return v;
}
__device__ __forceinline__ void lds_wait4(f4v_t& a, f4v_t& b, f4v_t& c, f4v_t& d) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
template <int KS, int NBUF>
__global__ __launch_bounds__(256 * KS) void k_linear_glds(LinBatch lb) {
  static_assert(KS == 2 && NBUF == 3, "512 threads move one 16-byte piece of each operand tile per chunk; everything stays below 64 KB of LDS");
  constexpr int GK = 32, TILE = 64 * 8 * 4, OPER = NBUF * 2 * TILE, RED = (KS - 1) * 4 * 16 * 64, NTAB = SDXP_NORM_K + GK;
  static_assert(RED <= OPER, "the k-group reduction reuses the operand images");
  __shared__ __attribute__((aligned(16))) float lds[OPER + 2 * NTAB];
  float* nmu = lds + OPER;
  float* nsd = nmu + NTAB;
  const LinArgs& g = lb.a[blockIdx.z];
  const int tid = threadIdx.x, wave8 = tid >> 6, wave = wave8 & 3, kg = tid >> 8, lane = tid & 63;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int M = g.M, N = g.N, K = g.K;
  if (m0 >= M || n0 >= N) return;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const bool norm = g.nmean != nullptr;
  if (norm) {   // ordinary loads: finished (the barrier below) before the first DMA is requested
    for (int k = tid; k < NTAB; k += 256 * KS) { nmu[k] = k < K ? (float)g.nmean[k] : 0.0f; nsd[k] = k < K ? sqrtf((float)g.nvar[k] + 1e-5f) : 1.0f; }
    __syncthreads();
  }
  // loader coordinates: slot tid of an image = (row tid / 8, position tid % 8) holds piece q = position ^ swizzle(row) of that row
  const int lrow = tid >> 3, lq = (tid & 7) ^ ((lrow >> 1) & 7);
  const float* pa = g.X + (size_t)(m0 + lrow < M ? m0 + lrow : M - 1) * K + 4 * lq;
  const float* pb = g.W + (size_t)(n0 + lrow < N ? n0 + lrow : N - 1) * K + 4 * lq;
  const int nchunk = (K + GK - 1) / GK;
  auto request = [&](int c, int buf) {
    float* da = lds + buf * 2 * TILE + wave8 * 256;   // this wave's 64 slots of the A image (wave-uniform: the DMA adds 16 lane)
    float* db = da + TILE;
    if (c * GK + 4 * lq < K) {
      __builtin_amdgcn_global_load_lds(SDX_AS_GLOBAL(pa + c * GK), SDX_AS_LDS(da), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(SDX_AS_GLOBAL(pb + c * GK), SDX_AS_LDS(db), 16, 0, 0);
    } else {                                        // past K (last chunk only): zeros, by the lanes whose pieces these are
      *reinterpret_cast<float4*>(da + 4 * lane) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      *reinterpret_cast<float4*>(db + 4 * lane) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  // reader coordinates
  const int ra = wm + (lane & 31), rb = wn + (lane & 31), hf = lane >> 5;
  const int sa = ra * 8, xa = (ra >> 1) & 7, sb = rb * 8, xb = (rb >> 1) & 7;
  const unsigned lds_base = (unsigned)(uintptr_t)SDX_AS_LDS(lds);   // byte offset of the array inside the workgroup's LDS
  struct Frag { f4v_t a[2], b[2], mu[2], sd[2]; };
  auto read_chunk = [&](int c, int buf, Frag& f) {   // requests only: the LDS-only barrier of the next trip completes them
    const unsigned A = lds_base + (unsigned)buf * (2 * TILE * 4), B = A + TILE * 4;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int q = 4 * kg + 2 * t + hf;
      f.a[t] = lds_read128(A + 16u * (unsigned)(sa + (q ^ xa)));
      f.b[t] = lds_read128(B + 16u * (unsigned)(sb + (q ^ xb)));
      if (norm) {
        const unsigned k = (unsigned)(c * GK + 4 * q);
        f.mu[t] = lds_read128(lds_base + 4u * (OPER + k));
        f.sd[t] = lds_read128(lds_base + 4u * (OPER + NTAB + k));
      }
    }
  };
  // One trip = chunk c out of registers (its LDS reads were requested a trip ago and ran under the previous chunk's multiplies):
  //   wait (counted) until chunk c + 1 has landed, LDS-only barrier (everybody's reads of chunk c are complete: its image pair is free, and
  //   everybody's pieces of chunk c + 1 are there), request chunk c + NBUF into the freed pair, request the LDS reads of chunk c + 1, multiply chunk c.
  auto trip = [&](int c, int buf, Frag& cur, Frag& nxt) {
    if (c + 1 < nchunk) {
      if (c + NBUF - 1 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (NBUF - 2)) : "memory");   // chunks c + 2 .. c + NBUF - 1 may still fly
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    SDX_LDS_BARRIER();
    lds_wait4(cur.a[0], cur.a[1], cur.b[0], cur.b[1]);            // (already complete: ties the registers to the wait for the compiler)
    if (norm) lds_wait4(cur.mu[0], cur.mu[1], cur.sd[0], cur.sd[1]);
    if (c + NBUF < nchunk) request(c + NBUF, buf);
    if (c + 1 < nchunk) read_chunk(c + 1, buf + 1 == NBUF ? 0 : buf + 1, nxt);
    __builtin_amdgcn_sched_barrier(0);                              // the reads are issued in front of the dependent multiplies, not behind them
    if (norm) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) cur.a[t][i] = clampf((cur.a[t][i] - cur.mu[t][i]) / cur.sd[t][i], -5.0f, 5.0f);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[t][i], cur.b[t][i], acc, 0, 0, 0);
  };
#pragma unroll
  for (int c = 0; c < NBUF; ++c)
    if (c < nchunk) request(c, c);
  Frag f0, f1;
  // chunk 0: landed once at most NBUF - 1 younger chunks are outstanding
  if (nchunk >= NBUF) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (NBUF - 1)) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SDX_LDS_BARRIER();
  read_chunk(0, 0, f0);
  int buf = 0;
  for (int c = 0; c < nchunk; c += 2) {
    trip(c, buf, f0, f1);
    buf = buf + 1 == NBUF ? 0 : buf + 1;
    if (c + 1 < nchunk) {
      trip(c + 1, buf, f1, f0);
      buf = buf + 1 == NBUF ? 0 : buf + 1;
    }
  }
  // sum of the k groups, in group order (the operand images are reused)
  SDX_LDS_BARRIER();
  {
    float* red = lds + wave * 16 * 64;
    if (kg == 1)
#pragma unroll
      for (int i = 0; i < 16; ++i) red[i * 64 + lane] = acc[i];
    SDX_LDS_BARRIER();
    if (kg != 0) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += red[i * 64 + lane];
  }
  const int col = n0 + wn + (lane & 31);
  if (col < N) {
    const float bias = g.b[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < M) {
        const float v = acc[r] + bias;
        g.Y[(size_t)row * N + col] = g.elu ? elu(v) : v;
      }
    }
  }
}
template <int WTM, int KS, int GKc>
static void launch_linear_as(const LinBatch& lb, int count, int M, int Nx, hipStream_t st) {
  hipLaunchKernelGGL((k_linear_mfma<WTM, KS, GKc>), dim3((Nx + GT - 1) / GT, (M + WTM * GT - 1) / (WTM * GT), count), dim3(256 * KS), 0, st, lb);
}
static int g_linear_shape = 0;           // sdxpk_linear_force_shape (tests / timing tools); 0 = automatic
#define SDXP_LINEAR_SMALL_M_SHAPE 3     // the shape below M = 2048 when SDXP_LINEAR_TILE is unset
static void launch_linear(const LinBatch& lb, int count, int M, int Nx, hipStream_t st) {
  // Up to 2048 rows: 64 x 64 tiles, every chunk split over two groups of four waves; beyond: 128 x 64.  SDXP_LINEAR_TILE forces one shape
  // (timing aid): 1 = 64 x 64, 2 = 128 x 64, 3 = 64 x 64 with 2 k groups, 4 = the same with chunks of 64, 5 = 4 k groups and chunks of
  // 64, 6 = 64 x 64 with chunks of 64.  sdxp_act at M = 1024 (tools/time_act.py, profiles/r3_act_shapes.txt): 87 us with shape 3 or 4, 95 with
  // 1, 96 with 6, 95 with 5, 141 with 2 (round 3 started at 143 with shape 1: exec-masked loads made every wait vmcnt(0), and the LDS operands were
  // read two at a time).  The matrix pipe is about half busy in the middle layer; the rest is the per-chunk barrier and LDS read latency
  for (int i = 0; i < count; ++i)
    if (lb.a[i].nmean && lb.a[i].K > SDXP_NORM_K) { fprintf(stderr, "seqdex: k_linear_mfma normalises at most %d input columns (got %d)\n", SDXP_NORM_K, lb.a[i].K); abort(); }
  static const int env_forced = getenv("SDXP_LINEAR_TILE") ? atoi(getenv("SDXP_LINEAR_TILE")) : 0;
  const int forced = g_linear_shape ? g_linear_shape : env_forced;
  const int shape = forced ? forced : (M > 2048 ? 2 : SDXP_LINEAR_SMALL_M_SHAPE);
  switch (shape) {
    case 2: launch_linear_as<2, 1, 32>(lb, count, M, Nx, st); break;
    case 3: launch_linear_as<1, 2, 32>(lb, count, M, Nx, st); break;
    case 4: launch_linear_as<1, 2, 64>(lb, count, M, Nx, st); break;
    case 5: launch_linear_as<1, 4, 64>(lb, count, M, Nx, st); break;
    case 6: launch_linear_as<1, 1, 64>(lb, count, M, Nx, st); break;
    case 7: hipLaunchKernelGGL((k_linear_glds<2, 3>), dim3((Nx + GT - 1) / GT, (M + GT - 1) / GT, count), dim3(512), 0, st, lb); break;
    default: launch_linear_as<1, 1, 32>(lb, count, M, Nx, st); break;
  }
}

// counter-based standard normal (Box-Muller on two hashed uniforms)
__device__ __forceinline__ float randn(uint64_t seed, uint64_t a, uint64_t b) {
  const uint64_t h = sdx_hash(seed, a, b);
  const float u1 = ((float)((h >> 40) & 0xFFFFFF) + 1.0f) * (1.0f / 16777217.0f);
  const float u2 = (float)((h >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530718f * u2);
}

// rollout heads: one wave per env.  mu = h3a Wmu^T + b; V = h3v Wv^T + b (central value); a = mu + sigma eps;
// neglogp (RC:2114-2126); writes row (env, t) of the env-major experience buffer (PS:345-352).
__global__ __launch_bounds__(64) void k_act_heads(SdxpDev D, int t, const float* __restrict__ obs,
                                                  const float* __restrict__ states, const int64_t* __restrict__ dones,
                                                  const float* __restrict__ eps_in, float* __restrict__ actions_out,
                                                  uint64_t counter) {
  const int e = blockIdx.x, lane = threadIdx.x;
  const int A = D.act_dim;
  constexpr int U = 256;                            // units[2] (sdxp_create refuses other widths)
  __shared__ float s_h[2][U];
  const size_t row = (size_t)e * D.horizon + t;
  // Everything this workgroup reads that does not depend on its own arithmetic is requested up front - the trunk outputs, the value-head
  // weights and the observation / state rows that are copied into the dataset (16-byte pieces; obs_dim and state_dim are multiples of 4):
  // with 4 waves per CU every dependent load is a full L2 round trip (about 0.7 us), and the rolled copy loops alone were 16 of them
  constexpr int CP = 4;                             // 16-byte pieces per lane and row: rows up to 1024 floats (SDXP_NORM_K)
  float ha[U / 64], hv[U / 64], wvh[U / 64];
  float4 co[CP], cs[CP];
  const float4* osrc = reinterpret_cast<const float4*>(obs + (size_t)e * D.obs_dim);
  const float4* ssrc = reinterpret_cast<const float4*>(states + (size_t)e * D.state_dim);
  const int on4 = D.obs_dim >> 2, sn4 = D.state_dim >> 2;
#pragma unroll
  for (int i = 0; i < U / 64; ++i) {
    ha[i] = D.h_a[2][(size_t)e * U + lane + 64 * i];
    hv[i] = D.h_v[2][(size_t)e * U + lane + 64 * i];
    wvh[i] = D.cv[D.coff.v_w + lane + 64 * i];
  }
  // (lanes past the end of a row repeat its last piece, load and store alike: no branch for the optimiser to sink the loads into)
#pragma unroll
  for (int i = 0; i < CP; ++i) {
    co[i] = osrc[min(lane + 64 * i, on4 - 1)];
    cs[i] = ssrc[min(lane + 64 * i, sn4 - 1)];
  }
  __builtin_amdgcn_sched_barrier(0);               // (keeps the loads above the first use: the scheduler sinks each copy load to its store)
#pragma unroll
  for (int i = 0; i < U / 64; ++i) { s_h[0][lane + 64 * i] = ha[i]; s_h[1][lane + 64 * i] = hv[i]; }
  {
    float4* odst = reinterpret_cast<float4*>(D.mb_obs + row * D.obs_dim);
    float4* sdst = reinterpret_cast<float4*>(D.mb_states + row * D.state_dim);
#pragma unroll
    for (int i = 0; i < CP; ++i) {
      odst[min(lane + 64 * i, on4 - 1)] = co[i];
      sdst[min(lane + 64 * i, sn4 - 1)] = cs[i];
    }
  }
  __syncthreads();
  float nlp_part = 0.0f;
  if (lane < A) {
    const float* w = D.ac + D.off.mu_w + (size_t)lane * U;
    float mu = D.ac[D.off.mu_b + lane];
#pragma unroll 1
    for (int k0 = 0; k0 < U; k0 += 32) {           // 32 weight loads in flight (rolled, every k was its own L2 round trip: 13.7 us per launch);
      float wv[32];                                // the sum still runs over k in ascending order
#pragma unroll
      for (int j = 0; j < 32; ++j) wv[j] = w[k0 + j];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 32; ++j) mu += wv[j] * s_h[0][k0 + j];
    }
    const float ls = D.ac[D.off.logstd + lane], sg = expf(ls);
    const float eps = eps_in ? eps_in[(size_t)e * A + lane] : randn(D.seed, counter, (uint64_t)e * 64 + lane);
    const float a = mu + sg * eps;
    const float z = (a - mu) / sg;
    nlp_part = 0.5f * z * z + ls;
    actions_out[(size_t)e * A + lane] = a;
    D.mb_actions[row * A + lane] = a;
    D.mb_mus[row * A + lane] = mu;
    D.mb_sigmas[row * A + lane] = sg;
  }
  const float nlp = wave_sum(nlp_part) + 0.5f * 1.8378770664093453f * (float)A;
  float vpart = 0.0f;
#pragma unroll
  for (int i = 0; i < U / 64; ++i) vpart += wvh[i] * hv[i];
  const float v = wave_sum(vpart) + D.cv[D.coff.v_b];
  if (lane == 0) {
    D.mb_neglogp[row] = nlp;
    D.mb_values[row] = v;
    D.mb_dones[row] = (dones && dones[e] != 0) ? 1.0f : 0.0f;
  }
}

__global__ void k_store_rewards(SdxpDev D, int t, const float* __restrict__ rew, const int64_t* __restrict__ dones_after) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= D.N) return;
  const float r = rew[e];
  D.mb_rewards[(size_t)e * D.horizon + t] = r;   // reward_shaper scale 1 (YG:32-33)
  // episode statistics (PS:361-373): current_rewards/current_lengths, game_rewards/game_lengths on done
  const float cr = D.cur_rew[e] + r, cl = D.cur_len[e] + 1.0f;
  const bool done = dones_after && dones_after[e] != 0;
  if (done) {
    atomicAdd(&D.ctrl->games_sum_rew, cr);
    atomicAdd(&D.ctrl->games_sum_len, cl);
    atomicAdd(&D.ctrl->games_cnt, 1.0f);
  }
  D.cur_rew[e] = done ? 0.0f : cr;
  D.cur_len[e] = done ? 0.0f : cl;
}

// last value head only (get_values, RC:1725-1752)
__global__ __launch_bounds__(64) void k_value_head(SdxpDev D, float* __restrict__ out) {
  const int e = blockIdx.x, lane = threadIdx.x, U = D.units[2];
  float p = 0.0f;
  for (int k = lane; k < U; k += 64) p += D.cv[D.coff.v_w + k] * D.h_v[2][(size_t)e * U + k];
  const float v = wave_sum(p) + D.cv[D.coff.v_b];
  if (lane == 0) out[e] = v;
}

// GAE (discount_values, PS:331-336) + returns, thread per env, env-major rows
__global__ void k_gae(SdxpDev D, const float* __restrict__ last_values, const int64_t* __restrict__ last_dones) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= D.N) return;
  const int H = D.horizon;
  float lastgae = 0.0f;
  for (int t = H - 1; t >= 0; --t) {
    float nonterminal, nextv;
    if (t == H - 1) { nonterminal = (last_dones && last_dones[e] != 0) ? 0.0f : 1.0f; nextv = last_values[e]; }
    else { nonterminal = 1.0f - D.mb_dones[(size_t)e * H + t + 1]; nextv = D.mb_values[(size_t)e * H + t + 1]; }
    const float v = D.mb_values[(size_t)e * H + t];
    const float delta = D.mb_rewards[(size_t)e * H + t] + D.gamma * nextv * nonterminal - v;
    lastgae = delta + D.gamma * D.tau * nonterminal * lastgae;
    D.returns[(size_t)e * H + t] = lastgae + v;
    D.adv[(size_t)e * H + t] = lastgae;          // advantages = returns - values (RC:1639)
  }
}

// advantage normalisation (RC:1645-1651): (A - mean) / (std_unbiased + 1e-8), one block over N*H values
__global__ __launch_bounds__(1024) void k_adv_norm(SdxpDev D) {
  __shared__ double s_a[16], s_b[16];
  const int n = D.N * D.horizon, tid = threadIdx.x;
  double s = 0.0, s2 = 0.0;
  for (int i = tid; i < n; i += 1024) { const double a = D.adv[i]; s += a; }
  // two-pass for accuracy: mean first
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((tid & 63) == 0) s_a[tid >> 6] = s;
  __syncthreads();
  double tot = 0.0;
  for (int i = 0; i < 16; ++i) tot += s_a[i];
  const double mean = tot / n;
  for (int i = tid; i < n; i += 1024) { const double d = D.adv[i] - mean; s2 += d * d; }
  for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
  if ((tid & 63) == 0) s_b[tid >> 6] = s2;
  __syncthreads();
  double tot2 = 0.0;
  for (int i = 0; i < 16; ++i) tot2 += s_b[i];
  const float stdv = (float)sqrt(tot2 / (n - 1));
  const float fm = (float)mean;
  for (int i = tid; i < n; i += 1024) D.adv[i] = (D.adv[i] - fm) / (stdv + 1e-8f);
}

// ------------------------------------------------------------------------------------------------ small-minibatch update
// network ids: 0 actor trunk, 1 critic trunk, 2 central-value trunk.
//   x[net][l] (l=1..3)   [2][MB][units[l-1]]  ELU output of trunk layer l-1 == input of layer l (l=3: head input),
//                        double buffered by optimiser-step parity.  Layer-0 inputs are read straight from the dataset
//                        (obs rows; pre-normalised central-value rows, see k_cv_prenorm).
//   dy2[net]             [2][MB][units[2]]    dLoss/d(pre-activation) of trunk layer 2 (written by the HEAD kernel)
//   dxacc[net][l] l=0,1  [2][MB][units[l]]    dLoss/d(OUTPUT) of trunk layer l, accumulated with atomics by the B kernels;
//                        consumers multiply by elu'(output) on the fly: dY_l = dxacc_l * elu'(x[l+1]).
#define MAXG 64   // MB*MB for MB <= 8
#define STAMP(i) do { if (threadIdx.x == 0) D.dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)

__device__ __forceinline__ float elu_grad_from_out(float h) { return h > 0.0f ? 1.0f : h + 1.0f; }

template <int MB>
__device__ __forceinline__ const float* layer_input(const SdxpDev& D, const SdxpCtrl* ctl, int net, int l, bool cur) {
  const int par = ctl->step & 1;
  if (l > 0) return D.x[net][l] + (size_t)(cur ? par : par ^ 1) * MB * D.units[l - 1];
  const int mb = cur ? ctl->mb_index : ctl->prev_mb;
  const int me = cur ? ctl->mini_epoch : ctl->prev_mini_epoch;
  if (net == 2) return (me == 0 ? D.cvx0 : D.cvx1) + (size_t)mb * MB * D.state_dim;
  return D.mb_obs + (size_t)mb * MB * D.obs_dim;
}
// dLoss/d(pre-activation) of trunk layer l, sample s, neuron n, for step parity `par`
template <int MB>
__device__ __forceinline__ float dy_at(const SdxpDev& D, int net, int l, int par, int s, int n) {
  const int Nl = D.units[l];
  if (l == 2) return D.dy2[net][(size_t)par * MB * Nl + s * Nl + n];
  return D.dxacc[net][l][(size_t)par * MB * Nl + s * Nl + n] * elu_grad_from_out(D.x[net][l + 1][(size_t)par * MB * Nl + s * Nl + n]);
}

// block-wide symmetric Gram of MB vectors of length K living in LDS (v[s*K+k]); result (MB*MB floats) to out (global)
template <int MB>
__device__ void block_gram(const float* v, int K, float* out, float* s_scr /*>= MB*MB floats of LDS*/) {
  const int tid = threadIdx.x, nt = blockDim.x;
  float g[MB][MB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < MB; ++b) g[a][b] = 0.0f;
  for (int k = tid; k < K; k += nt) {
    float x[MB];
#pragma unroll
    for (int a = 0; a < MB; ++a) x[a] = v[a * K + k];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b) g[a][b] += x[a] * x[b];
  }
  if (tid < MB * MB) s_scr[tid] = 0.0f;
  __syncthreads();
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const float r = wave_sum(g[a][b]);
      if ((tid & 63) == 0) atomicAdd(&s_scr[a * MB + b], r);
    }
  __syncthreads();
  if (tid < MB * MB) { const int a = tid / MB, b = tid % MB; out[tid] = a >= b ? s_scr[a * MB + b] : s_scr[b * MB + a]; }
}

// Adam step of one parameter (torch.optim.Adam, betas 0.9/0.999, eps 1e-8, bias corrections bc1/bc2 precomputed)
__device__ __forceinline__ float adam1(float w, float g, float& m, float& v, float lr_bc1, float isq_bc2) {
  m = 0.9f * m + 0.1f * g;
  v = 0.999f * v + 0.001f * g * g;
  return w - lr_bc1 * m / (sqrtf(v) * isq_bc2 + 1e-8f);
}

// L-kernel: lazy Adam of the previous minibatch + forward of the current one for trunk layer `l` of all three nets.
// block = 256 threads = 4 waves; a wave owns RPW consecutive weight rows; lanes run along K with float4 accesses.
// All global loads of a wave (RPW rows x KI float4 columns x {w,m,v}) are issued before any arithmetic so that the
// kernel is bandwidth- rather than latency-bound.
template <int MB, int RPW, int KI>
__global__ __launch_bounds__(256) void k_layer(SdxpDev D, int l) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float s_scr[MAXG];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  SdxpCtrl* ctl = D.ctrl;
  const int par = ctl->step & 1;
  const int Nl = D.units[l];
  constexpr int rows_per_block = 4 * RPW;
  const int blocks_per_net = (Nl + rows_per_block - 1) / rows_per_block;
  const int net = blockIdx.x / blocks_per_net, blk = blockIdx.x % blocks_per_net;
  const int K = (l == 0) ? (net == 2 ? D.state_dim : D.obs_dim) : D.units[l - 1];
  const int K4 = K >> 2;
  const bool pend = (net == 2 ? ctl->cv_pending : ctl->ac_pending) != 0;
  float* P = net == 2 ? D.cv : D.ac;
  float* Mo = net == 2 ? D.cv_m : D.ac_m;
  float* Vo = net == 2 ? D.cv_v : D.ac_v;
  const size_t woff = net == 0 ? D.off.a_w[l] : net == 1 ? D.off.c_w[l] : D.coff.w[l];
  const size_t boff = net == 0 ? D.off.a_b[l] : net == 1 ? D.off.c_b[l] : D.coff.b[l];
  const float lr = net == 2 ? ctl->cv_lr_applied : ctl->ac_lr_applied;
  const float gs = net == 2 ? ctl->cv_gscale : ctl->ac_gscale;          // grad clip scale of the pending step
  const float bc1 = net == 2 ? ctl->cv_bc1 : ctl->ac_bc1, bc2 = net == 2 ? ctl->cv_bc2 : ctl->ac_bc2;
  const float lr_bc1 = lr / bc1, isq_bc2 = 1.0f / sqrtf(bc2);
  const int n0 = blk * rows_per_block + wave * RPW;
  // ---- issue this wave's weight / moment loads first
  float4 w[RPW][KI], m[RPW][KI], v[RPW][KI];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int k4 = lane + 64 * i, n = n0 + r;
      if (k4 < K4 && n < Nl) {
        const size_t o = woff + (size_t)n * K;
        w[r][i] = reinterpret_cast<const float4*>(P + o)[k4];
        if (pend) { m[r][i] = reinterpret_cast<const float4*>(Mo + o)[k4]; v[r][i] = reinterpret_cast<const float4*>(Vo + o)[k4]; }
      } else w[r][i] = make_float4(0, 0, 0, 0);
    }
  float dyn[RPW][MB], bias[RPW], bm[RPW], bv[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int n = n0 + r;
    bias[r] = n < Nl ? P[boff + n] : 0.0f;
    bm[r] = (pend && n < Nl) ? Mo[boff + n] : 0.0f;
    bv[r] = (pend && n < Nl) ? Vo[boff + n] : 0.0f;
#pragma unroll
    for (int s = 0; s < MB; ++s) dyn[r][s] = (pend && n < Nl) ? dy_at<MB>(D, net, l, par ^ 1, s, n) * gs : 0.0f;
  }
  // ---- stage the rank-MB factors of the inputs
  float* xc = sm;                 // [MB][K] current input
  float* xp = sm + MB * K;        // [MB][K] previous input (factor of the pending gradient)
  {
    const float4* gc = reinterpret_cast<const float4*>(layer_input<MB>(D, ctl, net, l, true));
    const float4* gp = reinterpret_cast<const float4*>(layer_input<MB>(D, ctl, net, l, false));
    float4* c4 = reinterpret_cast<float4*>(xc);
    float4* p4 = reinterpret_cast<float4*>(xp);
    for (int i = tid; i < MB * K4; i += 256) { c4[i] = gc[i]; if (pend) p4[i] = gp[i]; }
  }
  __syncthreads();
  if (blk == 0) block_gram<MB>(xc, K, ctl->gx[net][l], s_scr);   // Gram of this layer's input, for the grad norm
  float* xn = D.x[net][l + 1] + (size_t)par * MB * Nl;            // output = next layer's input
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int n = n0 + r;
    float acc[MB];
#pragma unroll
    for (int s = 0; s < MB; ++s) acc[s] = 0.0f;
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int k4 = lane + 64 * i;
      if (k4 < K4 && n < Nl) {
        float4 ww = w[r][i];
        if (pend) {
          float4 mm = m[r][i], vv = v[r][i];
          float4 g = make_float4(0, 0, 0, 0);
#pragma unroll
          for (int s = 0; s < MB; ++s) {
            const float4 x = reinterpret_cast<const float4*>(xp + s * K)[k4];
            g.x += dyn[r][s] * x.x; g.y += dyn[r][s] * x.y; g.z += dyn[r][s] * x.z; g.w += dyn[r][s] * x.w;
          }
          ww.x = adam1(ww.x, g.x, mm.x, vv.x, lr_bc1, isq_bc2); ww.y = adam1(ww.y, g.y, mm.y, vv.y, lr_bc1, isq_bc2);
          ww.z = adam1(ww.z, g.z, mm.z, vv.z, lr_bc1, isq_bc2); ww.w = adam1(ww.w, g.w, mm.w, vv.w, lr_bc1, isq_bc2);
          const size_t o = woff + (size_t)n * K;
          reinterpret_cast<float4*>(Mo + o)[k4] = mm; reinterpret_cast<float4*>(Vo + o)[k4] = vv;
          reinterpret_cast<float4*>(P + o)[k4] = ww;
        }
#pragma unroll
        for (int s = 0; s < MB; ++s) {
          const float4 x = reinterpret_cast<const float4*>(xc + s * K)[k4];
          acc[s] += ww.x * x.x + ww.y * x.y + ww.z * x.z + ww.w * x.w;
        }
      }
    }
    float b = bias[r];
    if (pend && n < Nl) {
      float g = 0.0f;
#pragma unroll
      for (int s = 0; s < MB; ++s) g += dyn[r][s];
      float mm = bm[r], vv = bv[r];
      b = adam1(b, g, mm, vv, lr_bc1, isq_bc2);
      if (lane == 0) { Mo[boff + n] = mm; Vo[boff + n] = vv; P[boff + n] = b; }
    }
#pragma unroll
    for (int s = 0; s < MB; ++s) {
      const float y = wave_sum(acc[s]) + b;
      if (lane == 0 && n < Nl) xn[s * Nl + n] = elu(y);
    }
  }
}

// HEAD kernel (one block of 1024 threads): lazy Adam of the head parameters, head forward, PPO losses (R7), gradients
// of the heads' pre-activations, backward through the heads into dY of trunk layer 2, KL + statistics, mu/sigma
// write-back (dataset.update_mu_sigma, RC:1358), and the gradient-norm contributions of the heads and of layer 2.
// flush != 0: only the lazy Adam part (end of an epoch).
template <int MB>
__global__ __launch_bounds__(1024) void k_head(SdxpDev D, int flush) {
  __shared__ float s_h[3][MB][256];      // head inputs of the current minibatch
  __shared__ float s_hp[3][MB][256];     // ... of the previous one (factor of the pending gradient)
  __shared__ float s_w[26][256];         // head weight rows after the lazy Adam (23 mu rows, critic V, central V)
  __shared__ float s_dy2[3][MB][256];
  __shared__ float s_mu[MB][32], s_dmu[MB][32], s_dmu_p[MB][32], s_z[MB][32], s_act[MB][32], s_omu[MB][32], s_osg[MB][32];
  __shared__ float s_v[2][MB], s_dv[2][MB], s_dv_p[2][MB], s_gnlp[MB], s_ls[32];
  __shared__ float s_stat[MB][8], s_gh[3][MAXG], s_gd[3][MAXG], s_n2[3], s_n2p[3], s_dls[32], s_term[3 * MAXG], s_bterm[32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __shared__ SdxpCtrl s_ctl;               // read-mostly snapshot of the control block
  SdxpCtrl* gctl = D.ctrl;
  for (int i = tid; i < (int)(sizeof(SdxpCtrl) / 4); i += 1024) reinterpret_cast<uint32_t*>(&s_ctl)[i] = reinterpret_cast<const uint32_t*>(gctl)[i];
  __syncthreads();
  const SdxpCtrl* ctl = &s_ctl;
  STAMP(0);
  const int par = ctl->step & 1, U = D.units[2], A = D.act_dim;
  const size_t r0 = (size_t)ctl->mb_index * MB;   // first dataset row of this minibatch: contiguous, unshuffled (App. C)
  const bool apend = ctl->ac_pending != 0, cpend = ctl->cv_pending != 0;
  for (int i = tid; i < 3 * MB * U; i += 1024) {
    const int net = i / (MB * U), r = i % (MB * U);
    s_h[net][r / U][r % U] = D.x[net][3][(size_t)par * MB * U + r];
    s_hp[net][r / U][r % U] = D.x[net][3][(size_t)(par ^ 1) * MB * U + r];
  }
  if (tid < MB * 32) {
    const int s = tid / 32, a = tid % 32;
    s_dmu_p[s][a] = (a < A) ? D.dhead[(size_t)(par ^ 1) * MB * 34 + s * 34 + a] : 0.0f;
    s_act[s][a] = (a < A) ? D.mb_actions[(r0 + s) * A + a] : 0.0f;
    s_omu[s][a] = (a < A) ? D.mb_mus[(r0 + s) * A + a] : 0.0f;
    s_osg[s][a] = (a < A) ? D.mb_sigmas[(r0 + s) * A + a] : 1.0f;
  }
  if (tid < 2 * MB) s_dv_p[tid / MB][tid % MB] = D.dhead[(size_t)(par ^ 1) * MB * 34 + (tid % MB) * 34 + 32 + tid / MB];
  if (tid < 3) s_n2[tid] = 0.0f;
  __syncthreads();
  STAMP(1);
  // ---- lazy Adam + forward of the head rows: rows 0..A-1 = mu, row A = critic value, row A+1 = central value.
  // U = 256: one float4 per lane and row; both rows of a wave are loaded before any arithmetic.
  {
    const float ac_lr_bc1 = ctl->ac_lr_applied / ctl->ac_bc1, ac_isq = 1.0f / sqrtf(ctl->ac_bc2), ac_gs = ctl->ac_gscale;
    const float cv_lr_bc1 = ctl->cv_lr_applied / ctl->cv_bc1, cv_isq = 1.0f / sqrtf(ctl->cv_bc2), cv_gs = ctl->cv_gscale;
    float4 w[2], m[2], v[2];
    float bias[2], bm[2], bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wave + 16 * j;
      w[j] = make_float4(0, 0, 0, 0); m[j] = w[j]; v[j] = w[j]; bias[j] = bm[j] = bv[j] = 0.0f;
      if (row < A + 2) {
        const int net = row < A ? 0 : (row == A ? 1 : 2);
        const float* P = net == 2 ? D.cv : D.ac;
        const float* Mo = net == 2 ? D.cv_m : D.ac_m;
        const float* Vo = net == 2 ? D.cv_v : D.ac_v;
        const size_t woff = row < A ? D.off.mu_w + (size_t)row * U : (row == A ? D.off.v_w : D.coff.v_w);
        const size_t boff = row < A ? D.off.mu_b + row : (row == A ? D.off.v_b : D.coff.v_b);
        const bool pend = net == 2 ? cpend : apend;
        w[j] = reinterpret_cast<const float4*>(P + woff)[lane];
        bias[j] = P[boff];
        if (pend) {
          m[j] = reinterpret_cast<const float4*>(Mo + woff)[lane]; v[j] = reinterpret_cast<const float4*>(Vo + woff)[lane];
          bm[j] = Mo[boff]; bv[j] = Vo[boff];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wave + 16 * j;
      if (row >= A + 2) continue;
      const int net = row < A ? 0 : (row == A ? 1 : 2);
      float* P = net == 2 ? D.cv : D.ac;
      float* Mo = net == 2 ? D.cv_m : D.ac_m;
      float* Vo = net == 2 ? D.cv_v : D.ac_v;
      const size_t woff = row < A ? D.off.mu_w + (size_t)row * U : (row == A ? D.off.v_w : D.coff.v_w);
      const size_t boff = row < A ? D.off.mu_b + row : (row == A ? D.off.v_b : D.coff.v_b);
      const bool pend = net == 2 ? cpend : apend;
      const float lr_bc1 = net == 2 ? cv_lr_bc1 : ac_lr_bc1, isq_bc2 = net == 2 ? cv_isq : ac_isq;
      const float gs = net == 2 ? cv_gs : ac_gs;
      float dyn[MB], acc[MB];
#pragma unroll
      for (int s = 0; s < MB; ++s) dyn[s] = pend ? gs * (row < A ? s_dmu_p[s][row] : s_dv_p[row - A][s]) : 0.0f;
      float4 ww = w[j];
      const int k = 4 * lane;
      if (pend) {
        float4 g = make_float4(0, 0, 0, 0), mm = m[j], vv = v[j];
#pragma unroll
        for (int s = 0; s < MB; ++s) {
          g.x += dyn[s] * s_hp[net][s][k]; g.y += dyn[s] * s_hp[net][s][k + 1];
          g.z += dyn[s] * s_hp[net][s][k + 2]; g.w += dyn[s] * s_hp[net][s][k + 3];
        }
        ww.x = adam1(ww.x, g.x, mm.x, vv.x, lr_bc1, isq_bc2); ww.y = adam1(ww.y, g.y, mm.y, vv.y, lr_bc1, isq_bc2);
        ww.z = adam1(ww.z, g.z, mm.z, vv.z, lr_bc1, isq_bc2); ww.w = adam1(ww.w, g.w, mm.w, vv.w, lr_bc1, isq_bc2);
        reinterpret_cast<float4*>(Mo + woff)[lane] = mm; reinterpret_cast<float4*>(Vo + woff)[lane] = vv;
        reinterpret_cast<float4*>(P + woff)[lane] = ww;
      }
      s_w[row][k] = ww.x; s_w[row][k + 1] = ww.y; s_w[row][k + 2] = ww.z; s_w[row][k + 3] = ww.w;
#pragma unroll
      for (int s = 0; s < MB; ++s)
        acc[s] = ww.x * s_h[net][s][k] + ww.y * s_h[net][s][k + 1] + ww.z * s_h[net][s][k + 2] + ww.w * s_h[net][s][k + 3];
      float b = bias[j];
      if (pend) {
        float g = 0.0f;
#pragma unroll
        for (int s = 0; s < MB; ++s) g += dyn[s];
        float mm = bm[j], vv = bv[j];
        b = adam1(b, g, mm, vv, lr_bc1, isq_bc2);
        if (lane == 0) { Mo[boff] = mm; Vo[boff] = vv; P[boff] = b; }
      }
#pragma unroll
      for (int s = 0; s < MB; ++s) {
        const float y = wave_sum(acc[s]) + b;
        if (lane == 0) { if (row < A) s_mu[s][row] = y; else s_v[row - A][s] = y; }
      }
    }
  }
  if (tid >= 512 && tid < 512 + 32) {   // logstd parameter (fixed_sigma: a free Parameter, YG:18-21)
    const int a = tid - 512;
    float ls = 0.0f;
    if (a < A) {
      const size_t o = D.off.logstd + a;
      ls = D.ac[o];
      if (apend) {
        const float g = D.dlogstd[(size_t)(par ^ 1) * 32 + a] * ctl->ac_gscale;
        float m = D.ac_m[o], v = D.ac_v[o];
        ls = adam1(ls, g, m, v, ctl->ac_lr_applied / ctl->ac_bc1, 1.0f / sqrtf(ctl->ac_bc2));
        D.ac_m[o] = m; D.ac_v[o] = v; D.ac[o] = ls;
      }
    }
    s_ls[a] = ls;
  }
  __syncthreads();
  if (flush) {
    if (tid == 0) { gctl->ac_pending = 0; gctl->cv_pending = 0; }
    return;
  }
  const float invM = 1.0f / (float)MB;
  STAMP(2);
  if (tid < MB * 32) {
    const int s = tid / 32, a = tid % 32;
    s_z[s][a] = (a < A) ? (s_act[s][a] - s_mu[s][a]) / expf(s_ls[a]) : 0.0f;
  }
  __syncthreads();
  // ---- per-sample scalars: lanes (s, a) reduce over the 32-lane half-wave of sample s
  float r_nlp = 0.0f, r_kl = 0.0f, r_bl = 0.0f, r_ent = 0.0f;
  if (tid < MB * 32) {
    const int s = tid / 32, a = tid % 32;
    if (a < A) {
      const float ls = s_ls[a], sg = expf(ls), mu = s_mu[s][a];
      r_nlp = 0.5f * s_z[s][a] * s_z[s][a] + ls;                                                  // RC:2114-2126
      const float omu = s_omu[s][a], osg = s_osg[s][a];
      r_kl = logf(osg / sg + 1e-5f) + (sg * sg + (omu - mu) * (omu - mu)) / (2.0f * (osg * osg + 1e-5f)) - 0.5f;
      const float hi = fmaxf(mu - 1.1f, 0.0f), lo = fminf(mu + 1.1f, 0.0f);
      r_bl = hi * hi + lo * lo;
      r_ent = 0.5f + 0.5f * 1.8378770664093453f + ls;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      r_nlp += __shfl_xor(r_nlp, o, 64); r_kl += __shfl_xor(r_kl, o, 64);
      r_bl += __shfl_xor(r_bl, o, 64); r_ent += __shfl_xor(r_ent, o, 64);
    }
  }
  if (tid < MB * 32 && (tid % 32) == 0) {
    const int s = tid / 32;
    const float nlp = r_nlp + 0.5f * 1.8378770664093453f * (float)A, kl = r_kl, bl = r_bl, ent = r_ent;
    const float adv = D.adv[r0 + s];
    const float ratio = expf(D.mb_neglogp[r0 + s] - nlp);
    const float L1 = -adv * ratio, L2 = -adv * clampf(ratio, 1.0f - D.e_clip, 1.0f + D.e_clip);   // RC:1813
    const bool inr = ratio >= 1.0f - D.e_clip && ratio <= 1.0f + D.e_clip;
    s_gnlp[s] = (L1 > L2 || inr) ? adv * ratio : 0.0f;     // d max(L1,L2)/d nlp (L2 is constant outside the clip range)
    const float R = D.returns[r0 + s], vo = D.mb_values[r0 + s];
    float closs[2];
    for (int j = 0; j < 2; ++j) {                                                                 // RC:1818-1822
      const float v = s_v[j][s];
      const float vc = vo + clampf(v - vo, -D.e_clip, D.e_clip);
      const float c1 = (v - R) * (v - R), c2 = (vc - R) * (vc - R);
      float d;
      if (D.clip_value) {
        closs[j] = fmaxf(c1, c2);
        const bool inv = fabsf(v - vo) <= D.e_clip;
        d = (c1 > c2 || inv) ? 2.0f * (v - R) : 0.0f;      // outside the clip range v_clipped is constant
      } else { closs[j] = c1; d = 2.0f * (v - R); }
      s_dv[j][s] = (j == 0 ? 0.5f * D.critic_coef : 1.0f) * d * invM;
    }
    s_stat[s][1] = fmaxf(L1, L2); s_stat[s][2] = closs[0]; s_stat[s][3] = bl; s_stat[s][4] = kl;
    s_stat[s][5] = closs[1]; s_stat[s][6] = ent;
  }
  __syncthreads();
  if (tid < MB * 32) {
    const int s = tid / 32, a = tid % 32;
    float dmu = 0.0f;
    if (a < A) {
      const float sg = expf(s_ls[a]), mu = s_mu[s][a];
      const float hi = fmaxf(mu - 1.1f, 0.0f), lo = fminf(mu + 1.1f, 0.0f);
      dmu = s_gnlp[s] * (-(s_z[s][a] / sg)) * invM + D.bounds_coef * (2.0f * hi + 2.0f * lo) * invM;
      D.mb_mus[(r0 + s) * A + a] = mu;                                                            // RC:1358
      D.mb_sigmas[(r0 + s) * A + a] = sg;
    }
    s_dmu[s][a] = dmu;
    D.dhead[(size_t)par * MB * 34 + s * 34 + a] = dmu;
    if (a < 2) D.dhead[(size_t)par * MB * 34 + s * 34 + 32 + a] = s_dv[a][s];
  }
  if (tid >= 512 && tid < 512 + A) {
    const int a = tid - 512;
    float dls = 0.0f;
    for (int s = 0; s < MB; ++s) dls += s_gnlp[s] * (1.0f - s_z[s][a] * s_z[s][a]) * invM;
    D.dlogstd[(size_t)par * 32 + a] = dls;
    s_dls[a] = dls;
  }
  if (tid == 0) {
    for (int j = 1; j <= 6; ++j) {
      float t = 0.0f;
      for (int s = 0; s < MB; ++s) t += s_stat[s][j];
      gctl->acc[j] = t;
    }
  }
  __syncthreads();
  STAMP(3);
  // ---- backward through the heads: dY2[net][s][k] = (sum_rows dhead * W_head[row][k]) * elu'(h[net][s][k])
  for (int i = tid; i < 3 * MB * U; i += 1024) {
    const int net = i / (MB * U), s = (i / U) % MB, k = i % U;
    float d = 0.0f;
    if (net == 0) { for (int a = 0; a < A; ++a) d += s_dmu[s][a] * s_w[a][k]; }
    else d = s_dv[net - 1][s] * s_w[A + net - 1][k];
    const float v = d * elu_grad_from_out(s_h[net][s][k]);
    s_dy2[net][s][k] = v;
    D.dy2[net][(size_t)par * MB * U + s * U + k] = v;
  }
  __syncthreads();
  STAMP(4);
  // ---- gradient-norm pieces available here: heads (dhead x h) and trunk layer 2 (dy2 x x[net][2], Gram from k_layer).
  // Six Gram matrices (h and dy2 of the three nets, K = 256): one wave each, 4 elements per lane, DPP reductions,
  // results written by lane 0 (no atomics).  Waves 6..8 do the bias-gradient norms of layer 2.
  if (wave < 6) {
    const int net = wave % 3;
    const float* v = wave < 3 ? &s_h[net][0][0] : &s_dy2[net][0][0];
    float* out = wave < 3 ? s_gh[net] : s_gd[net];
    float x[MB][4];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) x[a][j] = v[a * 256 + lane + 64 * j];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int b2 = 0; b2 <= a; ++b2) {
        const float r = wave_sum(x[a][0] * x[b2][0] + x[a][1] * x[b2][1] + x[a][2] * x[b2][2] + x[a][3] * x[b2][3]);
        if (lane == 0) { out[a * MB + b2] = r; out[b2 * MB + a] = r; }
      }
  } else if (wave < 9) {
    const int net = wave - 6;
    float bs = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float sb = 0.0f;
#pragma unroll
      for (int s2 = 0; s2 < MB; ++s2) sb += s_dy2[net][s2][lane + 64 * j];
      bs += sb * sb;
    }
    bs = wave_sum(bs);
    if (lane == 0) s_n2[net] = bs;
  }
  __syncthreads();
  // n2 of heads + layer 2: thread (net, a, b) forms one term, 3 threads add them up
  if (tid < 3 * MB * MB) {
    const int net = tid / (MB * MB), a = (tid / MB) % MB, b2 = tid % MB;
    float dd = 0.0f;
    if (net == 0) { for (int j = 0; j < A; ++j) dd += s_dmu[a][j] * s_dmu[b2][j]; }
    else dd = s_dv[net - 1][a] * s_dv[net - 1][b2];
    s_term[tid] = dd * s_gh[net][a * MB + b2] + s_gd[net][a * MB + b2] * ctl->gx[net][2][a * MB + b2];
  } else if (tid >= 256 && tid < 256 + 32) {   // head bias gradients (mu biases, the two value biases) and logstd
    const int j = tid - 256;
    float t = 0.0f;
    if (j < A) { float sb = 0; for (int a = 0; a < MB; ++a) sb += s_dmu[a][j]; t = sb * sb + s_dls[j] * s_dls[j]; }
    s_bterm[j] = t;
  }
  __syncthreads();
  if (tid < 3) {
    const int net = tid;
    float n2 = 0.0f;
    for (int i = 0; i < MB * MB; ++i) n2 += s_term[net * MB * MB + i];
    if (net == 0) { for (int j = 0; j < A; ++j) n2 += s_bterm[j]; }
    else { float sb = 0; for (int a = 0; a < MB; ++a) sb += s_dv[net - 1][a]; n2 += sb * sb; }
    s_n2p[net] = n2;
  }
  __syncthreads();
  if (tid < 3) gctl->n2_part[tid] = s_n2p[tid] + s_n2[tid];
  STAMP(5);
}

// B-kernel: backward of trunk layer l+1 into dxacc of layer l for all nets:
//   dxacc_l[s][k] += sum_{n in split} dY_{l+1}[s][n] W_{l+1}[n][k]; thread per k, block per (net, k-chunk, n-split).
template <int MB>
__global__ __launch_bounds__(256) void k_back(SdxpDev D, int l) {   // l = layer whose output gradient is produced (1 or 0)
  const SdxpCtrl* ctl = D.ctrl;
  const int par = ctl->step & 1;
  const int Nn = D.units[l + 1], K = D.units[l];
  const int kchunks = (K + 255) / 256, splits = D.bsplit;
  const int per_net = kchunks * splits;
  const int net = blockIdx.x / per_net, rem = blockIdx.x % per_net, kc = rem / splits, sp = rem % splits;
  const int k = kc * 256 + threadIdx.x;
  __shared__ float s_dy[MB][64];
  const int n_per = (Nn + splits - 1) / splits, n0 = sp * n_per, n1 = min(Nn, n0 + n_per);
  const float* P = net == 2 ? D.cv : D.ac;
  const size_t woff = net == 0 ? D.off.a_w[l + 1] : net == 1 ? D.off.c_w[l + 1] : D.coff.w[l + 1];
  float acc[MB];
#pragma unroll
  for (int s = 0; s < MB; ++s) acc[s] = 0.0f;
  for (int nb = n0; nb < n1; nb += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < MB * 64; i += 256) {
      const int s = i / 64, n = nb + (i % 64);
      s_dy[s][i % 64] = n < n1 ? dy_at<MB>(D, net, l + 1, par, s, n) : 0.0f;
    }
    __syncthreads();
    if (k < K) {
      const int cnt = min(64, n1 - nb);
      for (int j = 0; j < cnt; ++j) {
        const float w = P[woff + (size_t)(nb + j) * K + k];
#pragma unroll
        for (int s = 0; s < MB; ++s) acc[s] += s_dy[s][j] * w;
      }
    }
  }
  if (k < K) {
    float* out = D.dxacc[net][l] + (size_t)par * MB * K;
#pragma unroll
    for (int s = 0; s < MB; ++s) atomicAdd(&out[s * K + k], acc[s]);
  }
}

// CTRL kernel (one block of 1024 threads): finishes the gradient norms (layers 0 and 1 from dxacc*elu' and the stored
// input Grams; layer 2 + heads from k_head), clip scales, Adam bias corrections, legacy adaptive LR (PS:306-312),
// statistics, advances the minibatch cursor and clears the other parity's dxacc.
// advance bits: 1 = a minibatch was just back-propagated; 2 = move the cursor; 8 = explicit-gradient (multi-rank) mode.
template <int MB>
__global__ __launch_bounds__(1024) void k_ctrl(SdxpDev D, int advance) {
  __shared__ float s_gd[3][2][MAXG];
  __shared__ float s_b[3];
  __shared__ float s_part[16][MAXG + 1];
  __shared__ SdxpCtrl s_ctl;               // snapshot: the single-thread bookkeeping below runs on LDS, not on HBM
  SdxpCtrl* gctl = D.ctrl;
  const int tid = threadIdx.x;
  for (int i = tid; i < (int)(sizeof(SdxpCtrl) / 4); i += 1024) reinterpret_cast<uint32_t*>(&s_ctl)[i] = reinterpret_cast<const uint32_t*>(gctl)[i];
  if (tid < 3 * 2 * MAXG) (&s_gd[0][0][0])[tid] = 0.0f;
  if (tid < 3) s_b[tid] = 0.0f;
  __syncthreads();
  SdxpCtrl* ctl = &s_ctl;
  const int par = ctl->step & 1;
  if (threadIdx.x == 0) D.dbg[8] = (long long)__builtin_readcyclecounter();
  if (advance & 1) {
    // (net, l) groups of dY vectors: l = 0 groups (units[0] long) get 3 waves each, l = 1 groups 2 waves each; a wave
    // accumulates lane-wise over its 64-neuron chunks and reduces ONCE (11 DPP sums), partials go to LDS.
    const int wave = tid >> 6, lane = tid & 63;
    if (wave < 15) {
      const int l = wave < 9 ? 0 : 1;
      const int net = wave < 9 ? wave / 3 : (wave - 9) / 2;
      const int sub = wave < 9 ? wave % 3 : (wave - 9) % 2, nsub = wave < 9 ? 3 : 2;
      const int C = D.units[l] / 64;
      float g[MB][MB], bs = 0.0f;
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b2 = 0; b2 < MB; ++b2) g[a][b2] = 0.0f;
      for (int c = sub; c < C; c += nsub) {
        const int n = c * 64 + lane;
        float v[MB], sb = 0.0f;
#pragma unroll
        for (int a = 0; a < MB; ++a) { v[a] = dy_at<MB>(D, net, l, par, a, n); sb += v[a]; }
        bs += sb * sb;
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b2 = 0; b2 <= a; ++b2) g[a][b2] += v[a] * v[b2];
      }
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b2 = 0; b2 <= a; ++b2) {
          const float r = wave_sum(g[a][b2]);
          if (lane == 0) { s_part[wave][a * MB + b2] = r; s_part[wave][b2 * MB + a] = r; }
        }
      const float rb = wave_sum(bs);
      if (lane == 0) s_part[wave][MAXG] = rb;
    }
    __syncthreads();
    if (threadIdx.x == 0) D.dbg[9] = (long long)__builtin_readcyclecounter();
    // n2[net] = n2_part[net] + sum_l sum_ab Gd[net][l][ab] * Gx[net][l][ab] + bias terms: thread (net, l, ab)
    if (tid < 3 * 2 * MB * MB) {
      const int net = tid / (2 * MB * MB), l = (tid / (MB * MB)) % 2, ab = tid % (MB * MB);
      float gd = 0.0f;
      if (l == 0) { for (int w = 0; w < 3; ++w) gd += s_part[net * 3 + w][ab]; }
      else { for (int w = 0; w < 2; ++w) gd += s_part[9 + net * 2 + w][ab]; }
      s_gd[net][l][ab] = gd * ctl->gx[net][l][ab];
    }
    __syncthreads();
    if (tid == 0) {
      float n2[3];
      for (int net = 0; net < 3; ++net) {
        float s = ctl->n2_part[net];
        for (int w = 0; w < 3; ++w) s += s_part[net * 3 + w][MAXG];
        for (int w = 0; w < 2; ++w) s += s_part[9 + net * 2 + w][MAXG];
        for (int l = 0; l < 2; ++l)
          for (int i = 0; i < MB * MB; ++i) s += s_gd[net][l][i];
        n2[net] = s;
      }
      const float ac_norm = sqrtf(n2[0] + n2[1]), cv_norm = sqrtf(n2[2]);
      ctl->ac_gnorm = ac_norm; ctl->cv_gnorm = cv_norm;
      ctl->ac_gscale = D.truncate_grads ? fminf(1.0f, D.grad_norm / (ac_norm + 1e-6f)) : 1.0f;   // clip_grad_norm_
      ctl->cv_gscale = D.truncate_grads ? fminf(1.0f, D.grad_norm / (cv_norm + 1e-6f)) : 1.0f;
      const bool explicit_mode = (advance & 8) != 0;   // multi-rank: Adam/LR happen in sdxp_apply after the all-reduce
      if (!explicit_mode) {
        ctl->ac_t += 1; ctl->cv_t += 1;
        ctl->ac_b1pow *= 0.9; ctl->ac_b2pow *= 0.999; ctl->cv_b1pow *= 0.9; ctl->cv_b2pow *= 0.999;
        ctl->ac_bc1 = (float)(1.0 - ctl->ac_b1pow); ctl->ac_bc2 = (float)(1.0 - ctl->ac_b2pow);
        ctl->cv_bc1 = (float)(1.0 - ctl->cv_b1pow); ctl->cv_bc2 = (float)(1.0 - ctl->cv_b2pow);
        ctl->ac_lr_applied = ctl->ac_lr; ctl->cv_lr_applied = ctl->cv_lr;
        ctl->ac_pending = 1; ctl->cv_pending = 1;
      } else { ctl->gn2_ac = 0.0f; ctl->gn2_cv = 0.0f; ctl->ac_pending = 0; ctl->cv_pending = 0; }
      const float invM = 1.0f / (float)MB;
      const float kl = ctl->acc[4] * invM;
      if (explicit_mode) { D.fact[D.foff.kl] = kl; D.ac_g[D.g_tail] = kl; }   // this rank's minibatch KL rides with factors / gradients
      ctl->sum_a_loss += ctl->acc[1] * invM; ctl->sum_c_loss += ctl->acc[2] * invM; ctl->sum_b_loss += ctl->acc[3] * invM;
      ctl->sum_kl += kl; ctl->sum_cv_loss += ctl->acc[5] * invM; ctl->sum_entropy += ctl->acc[6] * invM;
      ctl->n_mb += 1; ctl->last_kl = kl;
      if (D.adaptive_lr && !explicit_mode) {   // legacy schedule: after every minibatch (PS:306-312)
        if (kl > 2.0f * D.kl_threshold) ctl->ac_lr = fmaxf(ctl->ac_lr / 1.5f, 1e-6f);
        if (kl < 0.5f * D.kl_threshold) ctl->ac_lr = fminf(ctl->ac_lr * 1.5f, 1e-2f);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) D.dbg[10] = (long long)__builtin_readcyclecounter();
  if (advance & 2) {
    if (tid == 0) {
      ctl->prev_mb = ctl->mb_index; ctl->prev_mini_epoch = ctl->mini_epoch;
      if (advance & 1) {
        int mbn = ctl->mb_index + 1;
        if (mbn >= D.num_minibatches) { mbn = 0; ctl->mini_epoch += 1; }
        ctl->mb_index = mbn;
        ctl->step += 1;
      }
    }
    // the split-N accumulators of the parity that the NEXT step will write must start from zero
    const int npar = (advance & 1) ? (par ^ 1) : par;
    for (int net = 0; net < 3; ++net)
      for (int l = 0; l < 2; ++l) {
        float* a = D.dxacc[net][l] + (size_t)npar * MB * D.units[l];
        for (int i = tid; i < MB * D.units[l]; i += 1024) a[i] = 0.0f;
      }
  }
  __syncthreads();
  // write the header of the control block back (the Gram area gx is owned by the L kernels and left untouched)
  for (int i = tid; i < (int)(offsetof(SdxpCtrl, gx) / 4); i += 1024) reinterpret_cast<uint32_t*>(gctl)[i] = reinterpret_cast<const uint32_t*>(&s_ctl)[i];
  for (int i = (int)(offsetof(SdxpCtrl, rms_count) / 4) + tid; i < (int)(sizeof(SdxpCtrl) / 4); i += 1024) reinterpret_cast<uint32_t*>(gctl)[i] = reinterpret_cast<const uint32_t*>(&s_ctl)[i];
  if (threadIdx.x == 0) D.dbg[11] = (long long)__builtin_readcyclecounter();
}

// central-value input normalisation for the whole epoch, hoisted out of the minibatch loop: thread = feature, sequential
// over the minibatches exactly like RunningMeanStd in train mode (update with the minibatch, then normalise it) during
// mini-epoch 0 -> cvx0; statistics frozen afterwards (App. C) -> cvx1 for mini-epochs >= 1.
template <int MB>
__global__ void k_cv_prenorm(SdxpDev D) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= D.state_dim) return;
  const int S = D.state_dim;
  const float* __restrict__ src = D.mb_states;
  float* __restrict__ dst = D.cvx0;
  double mean = D.rms_mean[k], var = D.rms_var[k], cnt = D.ctrl->rms_count;
  // the running statistics form a serial chain over the minibatches; the rows of the next minibatch are fetched while the
  // chain of the current one runs, so that the chain, not the HBM latency, sets the time per minibatch
  float xn[MB];
#pragma unroll
  for (int s = 0; s < MB; ++s) xn[s] = src[(size_t)s * S + k];
  for (int mb = 0; mb < D.num_minibatches; ++mb) {
    float x[MB];
#pragma unroll
    for (int s = 0; s < MB; ++s) x[s] = xn[s];
    if (mb + 1 < D.num_minibatches) {
#pragma unroll
      for (int s = 0; s < MB; ++s) xn[s] = src[((size_t)(mb + 1) * MB + s) * S + k];
    }
    if (D.cv_normalize_input) {
      double bm = 0.0, bv = 0.0;
#pragma unroll
      for (int s = 0; s < MB; ++s) bm += x[s];
      bm /= MB;
#pragma unroll
      for (int s = 0; s < MB; ++s) { const double d = x[s] - bm; bv += d * d; }
      bv = MB > 1 ? bv / (MB - 1) : 0.0;
      const double delta = bm - mean, tot = cnt + MB;
      const double m2 = var * cnt + bv * MB + delta * delta * cnt * MB / tot;
      mean = mean + delta * MB / tot;
      var = m2 / tot;
      cnt = tot;
    }
    const float fm = (float)mean, rs = sqrtf((float)var + 1e-5f);
#pragma unroll
    for (int s = 0; s < MB; ++s)
      dst[((size_t)mb * MB + s) * S + k] = D.cv_normalize_input ? clampf((x[s] - fm) / rs, -5.0f, 5.0f) : x[s];
  }
  D.rms_mean[k] = mean; D.rms_var[k] = var;
  // rms_count is advanced by k_cv_rms_count AFTER this launch: every block of this grid reads the starting count above, and no
  // ordering exists between the blocks of one launch
}
__global__ void k_cv_rms_count(SdxpDev D, int rows) { if (D.cv_normalize_input) D.ctrl->rms_count += (double)rows; }
// mini-epochs >= 1: statistics frozen at their end-of-mini-epoch-0 values -> cvx1 (one thread per element)
__global__ __launch_bounds__(256) void k_cv_prenorm_frozen(SdxpDev D) {
  const int S = D.state_dim;
  const size_t total = (size_t)D.N * D.horizon * S;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % S);
    const float x = D.mb_states[i];
    const float fm = (float)D.rms_mean[k], rs = sqrtf((float)D.rms_var[k] + 1e-5f);
    D.cvx1[i] = D.cv_normalize_input ? clampf((x - fm) / rs, -5.0f, 5.0f) : x;
  }
}

// ------------------------------------------------------------------------------------------------ explicit gradients
// world_size > 1: the minibatch gradient is materialised into the flat *_GRADS buffers (same layout as the
// parameters) so that the caller can all-reduce it with RCCL; then sq-norm, clip and Adam run elementwise.
template <int MB>
__global__ __launch_bounds__(256) void k_grad_layer(SdxpDev D, int l) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const SdxpCtrl* ctl = D.ctrl;
  const int par = ctl->step & 1, Nl = D.units[l];
  const int blocks_per_net = (Nl + 3) / 4;
  const int net = blockIdx.x / blocks_per_net, n = (blockIdx.x % blocks_per_net) * 4 + wave;
  const int K = (l == 0) ? (net == 2 ? D.state_dim : D.obs_dim) : D.units[l - 1];
  float* G = net == 2 ? D.cv_g : D.ac_g;
  const size_t woff = net == 0 ? D.off.a_w[l] : net == 1 ? D.off.c_w[l] : D.coff.w[l];
  const size_t boff = net == 0 ? D.off.a_b[l] : net == 1 ? D.off.c_b[l] : D.coff.b[l];
  const float* gx = layer_input<MB>(D, ctl, net, l, true);
  for (int i = tid; i < MB * K; i += 256) sm[i] = gx[i];
  __syncthreads();
  if (n >= Nl) return;
  float dyn[MB], sb = 0.0f;
#pragma unroll
  for (int s = 0; s < MB; ++s) { dyn[s] = dy_at<MB>(D, net, l, par, s, n); sb += dyn[s]; }
  for (int k = lane; k < K; k += 64) {
    float g = 0.0f;
#pragma unroll
    for (int s = 0; s < MB; ++s) g += dyn[s] * sm[s * K + k];
    G[woff + (size_t)n * K + k] = g;
  }
  if (lane == 0) G[boff + n] = sb;
}
template <int MB>
__global__ __launch_bounds__(256) void k_grad_heads(SdxpDev D) {
  const int tid = threadIdx.x, par = D.ctrl->step & 1, U = D.units[2], A = D.act_dim;
  const float* dh = D.dhead + (size_t)par * MB * 34;
  for (int i = tid; i < (A + 2) * U; i += 256) {
    const int row = i / U, k = i % U;
    const int net = row < A ? 0 : (row == A ? 1 : 2);
    const float* h = D.x[net][3] + (size_t)par * MB * U;
    float g = 0.0f;
    for (int s = 0; s < MB; ++s) g += (row < A ? dh[s * 34 + row] : dh[s * 34 + 32 + (row - A)]) * h[s * U + k];
    if (row < A) D.ac_g[D.off.mu_w + (size_t)row * U + k] = g;
    else if (row == A) D.ac_g[D.off.v_w + k] = g;
    else D.cv_g[D.coff.v_w + k] = g;
  }
  if (tid < A + 2) {
    float g = 0.0f;
    for (int s = 0; s < MB; ++s) g += tid < A ? dh[s * 34 + tid] : dh[s * 34 + 32 + (tid - A)];
    if (tid < A) D.ac_g[D.off.mu_b + tid] = g;
    else if (tid == A) D.ac_g[D.off.v_b] = g;
    else D.cv_g[D.coff.v_b] = g;
  }
  if (tid < A) D.ac_g[D.off.logstd + tid] = D.dlogstd[(size_t)par * 32 + tid];
}
// ---- multi-rank path, factor exchange: instead of all-reducing 13.4 MB of materialised gradients per optimiser step the ranks
// all-gather the rank-MB factors (194 KB each) and every rank rebuilds the SUM over ranks of dY^T X locally.
template <int MB>
__global__ __launch_bounds__(256) void k_pack_factors(SdxpDev D) {
  const SdxpCtrl* ctl = D.ctrl;
  const int par = ctl->step & 1, seg = blockIdx.y;
  float* F = D.fact;
  const int stride = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
  if (seg < 9) {
    const int net = seg / 3, l = seg % 3;
    const int K = (l == 0) ? (net == 2 ? D.state_dim : D.obs_dim) : D.units[l - 1];
    const float* src = layer_input<MB>(D, ctl, net, l, true);
    for (int i = t0; i < MB * K; i += stride) F[D.foff.x[net][l] + i] = src[i];
  } else if (seg < 18) {
    const int net = (seg - 9) / 3, l = (seg - 9) % 3, Nl = D.units[l];
    for (int i = t0; i < MB * Nl; i += stride) F[D.foff.dy[net][l] + i] = dy_at<MB>(D, net, l, par, i / Nl, i % Nl);
  } else if (seg < 21) {
    const int net = seg - 18, U = D.units[2];
    const float* src = D.x[net][3] + (size_t)par * MB * U;
    for (int i = t0; i < MB * U; i += stride) F[D.foff.h[net] + i] = src[i];
  } else if (seg == 21) {
    for (int i = t0; i < MB * 34; i += stride) F[D.foff.dh + i] = D.dhead[(size_t)par * MB * 34 + i];
  } else {
    for (int i = t0; i < 32; i += stride) F[D.foff.dls + i] = D.dlogstd[(size_t)par * 32 + i];
  }
}
// block-level sum of the squared-gradient contributions of this workgroup, written (no atomics) to the slot of this workgroup:
// part[slot] for the actor-critic buffer, part[SDXP_SQN_STRIDE + slot] for the central value.  Every thread of the 256 calls it.
#define SDXP_SQN_STRIDE 2048
#define SDXP_TW_OFF 4096    // D.sqn_part + SDXP_TW_OFF: 16-byte tagged words of the one-launch apply: [SDXP_SQN_STRIDE] workgroups, then [64] groups
typedef unsigned int tw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void write_sqn_partials(float ss_ac, float ss_cv, float* part, int slot) {
  __shared__ float s_sq[2][4];
  const int tid = threadIdx.x;
  ss_ac = wave_sum(ss_ac); ss_cv = wave_sum(ss_cv);
  if ((tid & 63) == 0) { s_sq[0][tid >> 6] = ss_ac; s_sq[1][tid >> 6] = ss_cv; }
  __syncthreads();
  if (tid == 0) {
    part[slot] = (s_sq[0][0] + s_sq[0][1]) + (s_sq[0][2] + s_sq[0][3]);
    part[SDXP_SQN_STRIDE + slot] = (s_sq[1][0] + s_sq[1][1]) + (s_sq[1][2] + s_sq[1][3]);
  }
}

template <int MB>
__device__ __forceinline__ void grad_layer_w_body(const SdxpDev& D, int l, int bidx, float* part = nullptr, int slot = 0) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int Nl = D.units[l];
  const int blocks_per_net = (Nl + 3) / 4;
  const int net = bidx / blocks_per_net, n = (bidx % blocks_per_net) * 4 + wave;
  const int K = (l == 0) ? (net == 2 ? D.state_dim : D.obs_dim) : D.units[l - 1];
  float* G = net == 2 ? D.cv_g : D.ac_g;
  const size_t woff = net == 0 ? D.off.a_w[l] : net == 1 ? D.off.c_w[l] : D.coff.w[l];
  const size_t boff = net == 0 ? D.off.a_b[l] : net == 1 ? D.off.c_b[l] : D.coff.b[l];
  const bool valid = n < Nl;
  float g[16], sb = 0.0f;   // K <= 1024: 16 columns per lane
#pragma unroll
  for (int j = 0; j < 16; ++j) g[j] = 0.0f;
  for (int r = 0; r < D.world; ++r) {      // ascending rank order on every rank: identical sums everywhere
    const float* F = D.fact_all + (size_t)r * D.foff.total;
    const float* gx = F + D.foff.x[net][l];
    for (int i = tid; i < MB * K; i += 256) sm[i] = gx[i];
    __syncthreads();
    if (valid) {
      float dyn[MB];
#pragma unroll
      for (int s = 0; s < MB; ++s) { dyn[s] = F[D.foff.dy[net][l] + s * Nl + n]; sb += dyn[s]; }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = lane + 64 * j;
        if (k < K) {
#pragma unroll
          for (int s = 0; s < MB; ++s) g[j] += dyn[s] * sm[s * K + k];
        }
      }
    }
    __syncthreads();
  }
  float ss = 0.0f;
  if (valid) {
    const float sc = 1.0f / (float)D.world;
#pragma unroll
    for (int j = 0; j < 16; ++j) { const int k = lane + 64 * j; if (k < K) { G[woff + (size_t)n * K + k] = g[j]; ss += (g[j] * sc) * (g[j] * sc); } }
    if (lane == 0) { G[boff + n] = sb; ss += (sb * sc) * (sb * sc); }
  }
  if (part) write_sqn_partials(net == 2 ? 0.0f : ss, net == 2 ? ss : 0.0f, part, slot);
}
template <int MB>
__global__ __launch_bounds__(256) void k_grad_layer_w(SdxpDev D, int l) { grad_layer_w_body<MB>(D, l, blockIdx.x); }
template <int MB>
__device__ __forceinline__ void grad_heads_w_body(const SdxpDev& D, int hb, int nhb, float* part = nullptr, int slot = 0) {   // block hb of nhb; the last one also does the tails
  const int tid = threadIdx.x, U = D.units[2], A = D.act_dim, W = D.world;
  const size_t T = D.foff.total;
  const float sc = 1.0f / (float)W;
  float ss_ac = 0.0f, ss_cv = 0.0f;
  for (int i = hb * 256 + tid; i < (A + 2) * U; i += nhb * 256) {
    const int row = i / U, k = i % U;
    const int net = row < A ? 0 : (row == A ? 1 : 2);
    float g = 0.0f;
    for (int r = 0; r < W; ++r) {
      const float* F = D.fact_all + r * T;
      const float* dh = F + D.foff.dh;
      const float* h = F + D.foff.h[net];
      for (int s = 0; s < MB; ++s) g += (row < A ? dh[s * 34 + row] : dh[s * 34 + 32 + (row - A)]) * h[s * U + k];
    }
    if (row < A) D.ac_g[D.off.mu_w + (size_t)row * U + k] = g;
    else if (row == A) D.ac_g[D.off.v_w + k] = g;
    else D.cv_g[D.coff.v_w + k] = g;
    if (row <= A) ss_ac += (g * sc) * (g * sc); else ss_cv += (g * sc) * (g * sc);
  }
  if (hb != nhb - 1) { if (part) write_sqn_partials(ss_ac, ss_cv, part, slot); return; }
  if (tid < A + 2) {
    float g = 0.0f;
    for (int r = 0; r < W; ++r) {
      const float* dh = D.fact_all + r * T + D.foff.dh;
      for (int s = 0; s < MB; ++s) g += tid < A ? dh[s * 34 + tid] : dh[s * 34 + 32 + (tid - A)];
    }
    if (tid < A) D.ac_g[D.off.mu_b + tid] = g;
    else if (tid == A) D.ac_g[D.off.v_b] = g;
    else D.cv_g[D.coff.v_b] = g;
    if (tid <= A) ss_ac += (g * sc) * (g * sc); else ss_cv += (g * sc) * (g * sc);
  }
  if (tid >= 64 && tid < 64 + A) {
    float g = 0.0f;
    for (int r = 0; r < W; ++r) g += D.fact_all[r * T + D.foff.dls + (tid - 64)];
    D.ac_g[D.off.logstd + (tid - 64)] = g;
    ss_ac += (g * sc) * (g * sc);
  }
  if (tid == 128) {   // SUM of the ranks' minibatch KL, where sdxp_apply(0, -INFINITY) looks for it
    float kl = 0.0f;
    for (int r = 0; r < W; ++r) kl += D.fact_all[r * T + D.foff.kl];
    D.ac_g[D.g_tail] = kl;
  }
  if (part) write_sqn_partials(ss_ac, ss_cv, part, slot);
}
template <int MB>
__global__ __launch_bounds__(256) void k_grad_heads_w(SdxpDev D) { grad_heads_w_body<MB>(D, 0, 1); }
// squared norm of the flat gradient, bit-reproducible (every rank must compute the SAME clip scale from the same gradient, or the
// replicas drift apart): fixed block -> slice mapping, in-block tree in a fixed order, second stage sums the 512 block partials
__global__ __launch_bounds__(256) void k_sqnorm(const float* __restrict__ g, size_t n, float scale, float* part) {
  __shared__ float sw[4];
  float s = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float v = g[i] * scale; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ __launch_bounds__(512) void k_sqnorm_fin(const float* __restrict__ part, int nparts, float* out) {
  __shared__ float sw[8];
  float s = threadIdx.x < nparts ? part[threadIdx.x] : 0.0f;
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = ((sw[0] + sw[1]) + (sw[2] + sw[3])) + ((sw[4] + sw[5]) + (sw[6] + sw[7]));
}
__global__ __launch_bounds__(256) void k_adam_explicit(SdxpDev D, int which) {
  const SdxpCtrl* ctl = D.ctrl;
  const size_t n = which ? D.coff.total : D.off.total;
  float* P = which ? D.cv : D.ac; float* M = which ? D.cv_m : D.ac_m; float* V = which ? D.cv_v : D.ac_v;
  const float* G = which ? D.cv_g : D.ac_g;
  const float inv_w = 1.0f / (float)ctl->world;
  const float norm = sqrtf(which ? ctl->gn2_cv : ctl->gn2_ac);
  const float clip = D.truncate_grads ? fminf(1.0f, D.grad_norm / (norm + 1e-6f)) : 1.0f;
  const int t = (which ? ctl->cv_t : ctl->ac_t) + 1;
  const float bc1 = 1.0f - powf(0.9f, (float)t), bc2 = 1.0f - powf(0.999f, (float)t);
  const float lr = which ? ctl->cv_lr : ctl->ac_lr;
  const float lr_bc1 = lr / bc1, isq_bc2 = 1.0f / sqrtf(bc2);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float m = M[i], v = V[i];
    P[i] = adam1(P[i], G[i] * inv_w * clip, m, v, lr_bc1, isq_bc2);
    M[i] = m; V[i] = v;
  }
}
__global__ void k_apply_fin(SdxpDev D, int which, float kl_host) {
  SdxpCtrl* ctl = D.ctrl;
  if (which) { ctl->cv_t += 1; ctl->cv_b1pow *= 0.9; ctl->cv_b2pow *= 0.999; ctl->cv_gnorm = sqrtf(ctl->gn2_cv); return; }
  ctl->ac_t += 1; ctl->ac_b1pow *= 0.9; ctl->ac_b2pow *= 0.999; ctl->ac_gnorm = sqrtf(ctl->gn2_ac);
  // NaN -> device value, all-reduced in place; -inf -> the KL word that travelled with the gradients in ALL_GRADS
  const float kl = (kl_host != kl_host) ? ctl->last_kl / (float)ctl->world
                 : (kl_host == -INFINITY ? D.ac_g[D.g_tail] / (float)ctl->world : kl_host);
  if (D.adaptive_lr) {   // legacy schedule after every minibatch, on the rank-averaged KL (PS:306-312)
    if (kl > 2.0f * D.kl_threshold) ctl->ac_lr = fmaxf(ctl->ac_lr / 1.5f, 1e-6f);
    if (kl < 0.5f * D.kl_threshold) ctl->ac_lr = fminf(ctl->ac_lr * 1.5f, 1e-2f);
  }
}

// ------------------------------------------------------------------------------------------------ launch helpers
extern "C" void sdxpk_linear(const float* X, const float* W, const float* b, float* Y, int M, int N, int K, int elu_flag,
                             const double* nmean, const double* nvar, hipStream_t st) {
  LinBatch lb;
  lb.a[0] = LinArgs{X, W, b, Y, M, N, K, elu_flag, nmean, nvar};
  lb.a[1] = lb.a[0];
  launch_linear(lb, 1, M, N, st);
}
extern "C" void sdxpk_linear_force_shape(int shape) { g_linear_shape = shape; }   // not part of include/*.h: kernel tests and timing tools only
// two independent products in one launch (the same layer of the actor and of the central-value trunk)
extern "C" void sdxpk_linear2(const float* X0, const float* W0, const float* b0, float* Y0, int N0, int K0, const double* nmean0, const double* nvar0,
                              const float* X1, const float* W1, const float* b1, float* Y1, int N1, int K1, const double* nmean1, const double* nvar1,
                              int M, int elu_flag, hipStream_t st) {
  LinBatch lb;
  lb.a[0] = LinArgs{X0, W0, b0, Y0, M, N0, K0, elu_flag, nmean0, nvar0};
  lb.a[1] = LinArgs{X1, W1, b1, Y1, M, N1, K1, elu_flag, nmean1, nvar1};
  const int Nx = N0 > N1 ? N0 : N1;
  launch_linear(lb, 2, M, Nx, st);
}
extern "C" void sdxpk_act_heads(const SdxpDev* D, int t, const float* obs, const float* states, const int64_t* dones,
                                const float* eps, float* actions_out, uint64_t counter, hipStream_t st) {
  hipLaunchKernelGGL(k_act_heads, dim3(D->N), dim3(64), 0, st, *D, t, obs, states, dones, eps, actions_out, counter);
}
extern "C" void sdxpk_store_rewards(const SdxpDev* D, int t, const float* rew, const int64_t* dones_after, hipStream_t st) {
  hipLaunchKernelGGL(k_store_rewards, dim3((D->N + 255) / 256), dim3(256), 0, st, *D, t, rew, dones_after);
}
extern "C" void sdxpk_value_head(const SdxpDev* D, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_value_head, dim3(D->N), dim3(64), 0, st, *D, out);
}
extern "C" void sdxpk_gae_only(const SdxpDev* D, const float* last_values, const int64_t* last_dones, hipStream_t st) {
  hipLaunchKernelGGL(k_gae, dim3((D->N + 255) / 256), dim3(256), 0, st, *D, last_values, last_dones);
}
extern "C" void sdxpk_adv_norm(const SdxpDev* D, hipStream_t st) {
  if (D->normalize_advantage) hipLaunchKernelGGL(k_adv_norm, dim3(1), dim3(1024), 0, st, *D);
}
extern "C" void sdxpk_gae(const SdxpDev* D, const float* last_values, const int64_t* last_dones, hipStream_t st) {
  hipLaunchKernelGGL(k_gae, dim3((D->N + 255) / 256), dim3(256), 0, st, *D, last_values, last_dones);
  if (D->normalize_advantage) hipLaunchKernelGGL(k_adv_norm, dim3(1), dim3(1024), 0, st, *D);
}

template <int MB, int RPW, int KI>
static void launch_layer(const SdxpDev* D, int l, int Kmax, hipStream_t st) {
  const int blocks = 3 * ((D->units[l] + 4 * RPW - 1) / (4 * RPW));
  hipLaunchKernelGGL((k_layer<MB, RPW, KI>), dim3(blocks), dim3(256), (size_t)2 * MB * Kmax * sizeof(float), st, *D, l);
}
template <int MB>
static void launch_layers(const SdxpDev* D, hipStream_t st) {
  for (int l = 0; l < 3; ++l) {
    const int Kmax = l == 0 ? (D->state_dim > D->obs_dim ? D->state_dim : D->obs_dim) : D->units[l - 1];
    const int ki = (Kmax / 4 + 63) / 64;   // float4 columns per lane
    const bool two = (3 * D->units[l]) / 4 >= 512;   // 2 rows per wave when there are plenty of rows
    if (ki <= 1) { if (two) launch_layer<MB, 2, 1>(D, l, Kmax, st); else launch_layer<MB, 1, 1>(D, l, Kmax, st); }
    else if (ki == 2) { if (two) launch_layer<MB, 2, 2>(D, l, Kmax, st); else launch_layer<MB, 1, 2>(D, l, Kmax, st); }
    else if (ki == 3) { if (two) launch_layer<MB, 2, 3>(D, l, Kmax, st); else launch_layer<MB, 1, 3>(D, l, Kmax, st); }
    else { if (two) launch_layer<MB, 2, 4>(D, l, Kmax, st); else launch_layer<MB, 1, 4>(D, l, Kmax, st); }
  }
}
template <int MB>
static void launch_fwd_bwd(const SdxpDev* D, hipStream_t st) {
  launch_layers<MB>(D, st);
  hipLaunchKernelGGL(k_head<MB>, dim3(1), dim3(1024), 0, st, *D, 0);
  for (int l = 1; l >= 0; --l) {
    const int K = D->units[l];
    hipLaunchKernelGGL(k_back<MB>, dim3(3 * ((K + 255) / 256) * D->bsplit), dim3(256), 0, st, *D, l);
  }
}
template <int MB>
static void launch_step(const SdxpDev* D, hipStream_t st) {
  launch_fwd_bwd<MB>(D, st);
  hipLaunchKernelGGL(k_ctrl<MB>, dim3(1), dim3(1024), 0, st, *D, 1 | 2);
}
#define MB_SWITCH(mb, CALL) switch (mb) { case 2: { CALL(2); return 0; } case 4: { CALL(4); return 0; } case 8: { CALL(8); return 0; } default: return -1; }
// observation rows narrower than the (4-aligned) network input: zero-padded copy
__global__ __launch_bounds__(256) void k_pad_obs(SdxpDev D, const float* __restrict__ obs) {
  const size_t total = (size_t)D.N * D.obs_dim;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % D.obs_dim);
    D.obs_pad[i] = c < D.obs_cols ? obs[(i / D.obs_dim) * D.obs_cols + c] : 0.0f;
  }
}
extern "C" void sdxpk_pad_obs(const SdxpDev* D, const float* obs, hipStream_t st) {
  hipLaunchKernelGGL(k_pad_obs, dim3(256), dim3(256), 0, st, *D, obs);
}
extern "C" int sdxpk_update_step(const SdxpDev* D, int mb_size, hipStream_t st) {
#define C_(M) launch_step<M>(D, st)
  MB_SWITCH(mb_size, C_)
#undef C_
}
// begin of an epoch's update phase: central-value pre-normalisation for all minibatches + cursor/accumulator reset
extern "C" int sdxpk_update_begin(const SdxpDev* D, int mb_size, hipStream_t st) {
#define C_(M) hipLaunchKernelGGL(k_cv_prenorm<M>, dim3((D->state_dim + 63) / 64), dim3(64), 0, st, *D); \
              hipLaunchKernelGGL(k_cv_rms_count, dim3(1), dim3(1), 0, st, *D, D->num_minibatches * M); \
              hipLaunchKernelGGL(k_cv_prenorm_frozen, dim3(1024), dim3(256), 0, st, *D); \
              hipLaunchKernelGGL(k_ctrl<M>, dim3(1), dim3(1024), 0, st, *D, 2)
  MB_SWITCH(mb_size, C_)
#undef C_
}
extern "C" int sdxpk_prenorm(const SdxpDev* D, int mb_size, hipStream_t st) {
#define C_(M) hipLaunchKernelGGL(k_cv_prenorm<M>, dim3((D->state_dim + 63) / 64), dim3(64), 0, st, *D); \
              hipLaunchKernelGGL(k_cv_rms_count, dim3(1), dim3(1), 0, st, *D, D->num_minibatches * M); \
              hipLaunchKernelGGL(k_cv_prenorm_frozen, dim3(1024), dim3(256), 0, st, *D)
  MB_SWITCH(mb_size, C_)
#undef C_
}
// flush: apply the last pending optimiser step (the L/HEAD kernels run once more on the staged minibatch; their forward
// outputs are discarded)
extern "C" int sdxpk_update_flush_layers(const SdxpDev* D, int mb_size, hipStream_t st) {
#define C_(M) launch_layers<M>(D, st); hipLaunchKernelGGL(k_head<M>, dim3(1), dim3(1024), 0, st, *D, 1)
  MB_SWITCH(mb_size, C_)
#undef C_
}
template <int MB>
static void launch_backward_explicit(const SdxpDev* D, hipStream_t st) {
  launch_fwd_bwd<MB>(D, st);
  for (int l = 0; l < 3; ++l) {
    const int Nl = D->units[l];
    const int Kmax = l == 0 ? (D->state_dim > D->obs_dim ? D->state_dim : D->obs_dim) : D->units[l - 1];
    hipLaunchKernelGGL(k_grad_layer<MB>, dim3(3 * ((Nl + 3) / 4)), dim3(256), (size_t)MB * Kmax * sizeof(float), st, *D, l);
  }
  hipLaunchKernelGGL(k_grad_heads<MB>, dim3(1), dim3(256), 0, st, *D);
  hipLaunchKernelGGL(k_ctrl<MB>, dim3(1), dim3(1024), 0, st, *D, 1 | 2 | 8);
}
extern "C" int sdxpk_backward_explicit(const SdxpDev* D, int mb_size, hipStream_t st) {
#define C_(M) launch_backward_explicit<M>(D, st)
  MB_SWITCH(mb_size, C_)
#undef C_
}
// ---- the factor path's "apply" in four launches: every gradient block of the three layers and the heads at once, the squared
// norms of both flat gradients, clip + Adam of both networks, the step counters / LR rule
template <int MB>
__global__ __launch_bounds__(256) void k_grad_all_w(SdxpDev D, int with_norm) {
  const int nb0 = 3 * ((D.units[0] + 3) / 4), nb1 = 3 * ((D.units[1] + 3) / 4), nb2 = 3 * ((D.units[2] + 3) / 4);
  float* part = with_norm ? D.sqn_part : nullptr;      // squared-norm contribution of every workgroup (the gradients are still in registers)
  int b = blockIdx.x, l;
  if (b < nb0) l = 0;
  else if (b < nb0 + nb1) { l = 1; b -= nb0; }
  else if (b < nb0 + nb1 + nb2) { l = 2; b -= nb0 + nb1; }
  else { grad_heads_w_body<MB>(D, b - (nb0 + nb1 + nb2), (int)gridDim.x - (nb0 + nb1 + nb2), part, blockIdx.x); return; }
  grad_layer_w_body<MB>(D, l, b, part, blockIdx.x);
}
__global__ __launch_bounds__(256) void k_sqnorm2(SdxpDev D, float scale) {
  __shared__ float sw[4];
  const int which = blockIdx.y;
  const float* g = which ? D.cv_g : D.ac_g;
  const size_t n = which ? D.coff.total : D.off.total;
  float s = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float v = g[i] * scale; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) D.sqn_part[which * 512 + blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ __launch_bounds__(256) void k_adam2(SdxpDev D) {
  __shared__ float sw[4];
  __shared__ float s_n2;
  SdxpCtrl* ctl = D.ctrl;
  const int which = blockIdx.y;
  {   // every block folds the 512 partials in the same fixed order -> the same clip scale everywhere, on every rank
    float s = D.sqn_part[which * 512 + threadIdx.x] + D.sqn_part[which * 512 + 256 + threadIdx.x];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      s_n2 = (sw[0] + sw[1]) + (sw[2] + sw[3]);
      if (blockIdx.x == 0) { if (which) ctl->gn2_cv = s_n2; else ctl->gn2_ac = s_n2; }
    }
    __syncthreads();
  }
  const size_t n = which ? D.coff.total : D.off.total;
  float* P = which ? D.cv : D.ac; float* M = which ? D.cv_m : D.ac_m; float* V = which ? D.cv_v : D.ac_v;
  const float* G = which ? D.cv_g : D.ac_g;
  const float inv_w = 1.0f / (float)ctl->world;
  const float norm = sqrtf(s_n2);
  const float clip = D.truncate_grads ? fminf(1.0f, D.grad_norm / (norm + 1e-6f)) : 1.0f;
  const int t = (which ? ctl->cv_t : ctl->ac_t) + 1;
  const float bc1 = 1.0f - powf(0.9f, (float)t), bc2 = 1.0f - powf(0.999f, (float)t);
  const float lr = which ? ctl->cv_lr : ctl->ac_lr;
  const float lr_bc1 = lr / bc1, isq_bc2 = 1.0f / sqrtf(bc2);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float m = M[i], v = V[i];
    P[i] = adam1(P[i], G[i] * inv_w * clip, m, v, lr_bc1, isq_bc2);
    M[i] = m; V[i] = v;
  }
}
// adam1 with every rounding spelled out (no compiler-chosen contraction): k_adam3 and the one-launch apply (k_apply_factors_fused) inline it
// in different surroundings, where hipcc picked different fused multiply-adds for `0.9 m + 0.1 g` (first moments one ulp apart after one
// step); written this way the two forms of the apply are bit-identical (tests/test_gpu_fullsize_properties.py).
__device__ __forceinline__ float adam1x(float w, float g, float& m, float& v, float lr_bc1, float isq_bc2) {
#pragma clang fp contract(off)
  m = __builtin_fmaf(0.1f, g, 0.9f * m);
  v = __builtin_fmaf(0.001f * g, g, 0.999f * v);
  const float den = __builtin_fmaf(sqrtf(v), isq_bc2, 1e-8f);
  const float q = (lr_bc1 * m) / den;
  return w - q;
}
// k_adam2 with the squared norms taken from the per-workgroup partials that k_grad_all_w left in sqn_part (nparts slots per buffer):
// every block folds them in the same fixed order -> the same clip scale in every block and on every rank
__global__ __launch_bounds__(256) void k_adam3(SdxpDev D, int nparts) {
  __shared__ float sw[4];
  __shared__ float s_n2;
  SdxpCtrl* ctl = D.ctrl;
  const int which = blockIdx.y;
  {
    float s = 0.0f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += D.sqn_part[which * SDXP_SQN_STRIDE + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      s_n2 = (sw[0] + sw[1]) + (sw[2] + sw[3]);
      if (blockIdx.x == 0) { if (which) ctl->gn2_cv = s_n2; else ctl->gn2_ac = s_n2; }
    }
    __syncthreads();
  }
  const size_t n = which ? D.coff.total : D.off.total;
  float* P = which ? D.cv : D.ac; float* M = which ? D.cv_m : D.ac_m; float* V = which ? D.cv_v : D.ac_v;
  const float* G = which ? D.cv_g : D.ac_g;
  const float inv_w = 1.0f / (float)ctl->world;
  const float norm = sqrtf(s_n2);
  const float clip = D.truncate_grads ? fminf(1.0f, D.grad_norm / (norm + 1e-6f)) : 1.0f;
  const int t = (which ? ctl->cv_t : ctl->ac_t) + 1;
  const float bc1 = 1.0f - powf(0.9f, (float)t), bc2 = 1.0f - powf(0.999f, (float)t);
  const float lr = which ? ctl->cv_lr : ctl->ac_lr;
  const float lr_bc1 = lr / bc1, isq_bc2 = 1.0f / sqrtf(bc2);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float m = M[i], v = V[i];
    P[i] = adam1x(P[i], G[i] * inv_w * clip, m, v, lr_bc1, isq_bc2);
    M[i] = m; V[i] = v;
  }
}
__global__ void k_apply_fin2(SdxpDev D) {
  SdxpCtrl* ctl = D.ctrl;
  ctl->cv_t += 1; ctl->cv_b1pow *= 0.9; ctl->cv_b2pow *= 0.999; ctl->cv_gnorm = sqrtf(ctl->gn2_cv);
  ctl->ac_t += 1; ctl->ac_b1pow *= 0.9; ctl->ac_b2pow *= 0.999; ctl->ac_gnorm = sqrtf(ctl->gn2_ac);
  const float kl = D.ac_g[D.g_tail] / (float)ctl->world;
  if (D.adaptive_lr) {   // legacy schedule after every minibatch, on the rank-averaged KL (PS:306-312)
    if (kl > 2.0f * D.kl_threshold) ctl->ac_lr = fmaxf(ctl->ac_lr / 1.5f, 1e-6f);
    if (kl < 0.5f * D.kl_threshold) ctl->ac_lr = fminf(ctl->ac_lr * 1.5f, 1e-2f);
  }
}
template <int MB>
static void launch_apply_factors(const SdxpDev* D, hipStream_t st) {
  const int nhb = ((D->act_dim + 2) * D->units[2] + 255) / 256 + 1;   // head blocks: one output per thread, + one block for the tails
  const int nb = 3 * ((D->units[0] + 3) / 4) + 3 * ((D->units[1] + 3) / 4) + 3 * ((D->units[2] + 3) / 4) + nhb;
  const int Kmax = D->units[0] > D->state_dim ? D->units[0] : D->state_dim;
  // the squared norms ride along with the gradient rebuild (one launch and one 13.4 MB read less than a separate norm pass)
  hipLaunchKernelGGL(k_grad_all_w<MB>, dim3(nb), dim3(256), (size_t)MB * Kmax * sizeof(float), st, *D, 1);
  hipLaunchKernelGGL(k_adam3, dim3(512, 2), dim3(256), 0, st, *D, nb);
  hipLaunchKernelGGL(k_apply_fin2, dim3(1), dim3(1), 0, st, *D);
}
// ---- the apply phase of the multi-rank step as ONE launch (round 6).  The three-launch form above rebuilds the summed
// gradient (k_grad_all_w: 13.4 MB written), reads it back beside weights and moments (k_adam3) and advances the control block from a
// single thread (k_apply_fin2): three dependent graph nodes of 7-10 us each.  Here every workgroup keeps the gradient elements it rebuilt
// in registers, publishes its share of the squared norm, meets the other workgroups at ONE grid-wide ticket (all of them are resident: the
// host checks the occupancy before it picks this form), folds the published shares in the fixed order of k_adam3 - the same clip scale
// in every workgroup and on every rank - and sends its elements through Adam.  The flat gradient is never materialised (only the KL word
// is); workgroup 0 advances the control block behind the ticket (every workgroup read what it needs of it BEFORE it arrived).
// Exchange discipline (MI355X guide, cross-XCD hand-off): shares and KL word leave as write-through (sc1) stores, the producer drains its
// stores (vmcnt(0)) before the relaxed agent-scope ticket increment, consumers read them with agent-scope loads after the ticket completes.
// A workgroup that waits too long raises the fail flag (bar[34]) and every later launch returns at once: sdxp_update_status reports it
// and the handle goes back to the three-launch form.
template <int MB>
__global__ __launch_bounds__(256, 6) void k_apply_factors_fused(SdxpDev D, unsigned* __restrict__ gen, unsigned* __restrict__ failflag) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float s_sq[2][4], s_n2[2];
  __shared__ int s_ok;
  __shared__ unsigned s_gen;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  SdxpCtrl* ctl = D.ctrl;
  if (tid == 0) { s_ok = __hip_atomic_load(failflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; s_gen = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __syncthreads();
  const unsigned tag_gen = s_gen;   // launches of this handle so far (device counter: launches replayed from a hipGraph see it grow); workgroup 0 advances it behind the meeting
  if (!s_ok) return;   // an earlier launch timed out at its ticket: nothing is applied until the host has looked (sdxp_update_status)
  // ---- what this launch needs of the control block (read before the ticket: workgroup 0 advances the block behind it)
  const float inv_w = 1.0f / (float)ctl->world;
  const int t_ac = ctl->ac_t + 1, t_cv = ctl->cv_t + 1;
  const float lr_ac = ctl->ac_lr, lr_cv = ctl->cv_lr;
  const float ac_lr_bc1 = lr_ac / (1.0f - powf(0.9f, (float)t_ac)), ac_isq = 1.0f / sqrtf(1.0f - powf(0.999f, (float)t_ac));
  const float cv_lr_bc1 = lr_cv / (1.0f - powf(0.9f, (float)t_cv)), cv_isq = 1.0f / sqrtf(1.0f - powf(0.999f, (float)t_cv));
  // ---- the job of this workgroup (the map of k_grad_all_w): four rows of a trunk layer of one network, or a slice of the heads
  const int nb0 = 3 * ((D.units[0] + 3) / 4), nb1 = 3 * ((D.units[1] + 3) / 4), nb2 = 3 * ((D.units[2] + 3) / 4);
  int b = blockIdx.x, l = -1;
  if (b < nb0) l = 0;
  else if (b < nb0 + nb1) { l = 1; b -= nb0; }
  else if (b < nb0 + nb1 + nb2) { l = 2; b -= nb0 + nb1; }
  else b -= nb0 + nb1 + nb2;
  const int W = D.world;
  const float sc = 1.0f / (float)W;
  float g[16], sb = 0.0f, ss_ac = 0.0f, ss_cv = 0.0f;
#pragma unroll
  for (int j = 0; j < 16; ++j) g[j] = 0.0f;
  // layer job
  int net = 0, n = 0, K = 0;
  size_t woff = 0, boff = 0;
  bool valid = false;
  // heads job: at most one element per thread (main slice, or - last heads block - a bias / logstd entry)
  float gh = 0.0f;
  int h_which = -1;     // 0: actor-critic buffer, 1: central value, -1: none
  size_t h_idx = 0;
  if (l >= 0) {
    const int Nl = D.units[l];
    const int blocks_per_net = (Nl + 3) / 4;
    net = b / blocks_per_net; n = (b % blocks_per_net) * 4 + wave;
    K = (l == 0) ? (net == 2 ? D.state_dim : D.obs_dim) : D.units[l - 1];
    woff = net == 0 ? D.off.a_w[l] : net == 1 ? D.off.c_w[l] : D.coff.w[l];
    boff = net == 0 ? D.off.a_b[l] : net == 1 ? D.off.c_b[l] : D.coff.b[l];
    valid = n < Nl;
    for (int r = 0; r < W; ++r) {      // ascending rank order on every rank: identical sums everywhere (grad_layer_w_body's arithmetic)
      const float* F = D.fact_all + (size_t)r * D.foff.total;
      const float* gx = F + D.foff.x[net][l];
      for (int i = tid; i < MB * K; i += 256) sm[i] = gx[i];
      __syncthreads();
      if (valid) {
        float dyn[MB];
#pragma unroll
        for (int s = 0; s < MB; ++s) { dyn[s] = F[D.foff.dy[net][l] + s * Nl + n]; sb += dyn[s]; }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int k = lane + 64 * j;
          if (k < K) {
#pragma unroll
            for (int s = 0; s < MB; ++s) g[j] += dyn[s] * sm[s * K + k];
          }
        }
      }
      __syncthreads();
    }
    float ss = 0.0f;
    if (valid) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { const int k = lane + 64 * j; if (k < K) ss += (g[j] * sc) * (g[j] * sc); }
      if (lane == 0) ss += (sb * sc) * (sb * sc);
    }
    if (net == 2) ss_cv = ss; else ss_ac = ss;
  } else {
    const int hb = b, nhb = (int)gridDim.x - (nb0 + nb1 + nb2);
    const int U = D.units[2], A = D.act_dim;
    const size_t T = D.foff.total;
    const int i = hb * 256 + tid;
    if (i < (A + 2) * U) {
      const int row = i / U, k = i % U;
      const int hnet = row < A ? 0 : (row == A ? 1 : 2);
      for (int r = 0; r < W; ++r) {
        const float* F = D.fact_all + r * T;
        const float* dh = F + D.foff.dh;
        const float* h = F + D.foff.h[hnet];
        for (int s = 0; s < MB; ++s) gh += (row < A ? dh[s * 34 + row] : dh[s * 34 + 32 + (row - A)]) * h[s * U + k];
      }
      if (row < A) { h_which = 0; h_idx = D.off.mu_w + (size_t)row * U + k; }
      else if (row == A) { h_which = 0; h_idx = D.off.v_w + k; }
      else { h_which = 1; h_idx = D.coff.v_w + k; }
      if (row <= A) ss_ac += (gh * sc) * (gh * sc); else ss_cv += (gh * sc) * (gh * sc);
    }
    if (hb == nhb - 1) {   // the tails: head biases, logstd, the ranks' KL sum
      if (tid < A + 2) {
        for (int r = 0; r < W; ++r) {
          const float* dh = D.fact_all + r * T + D.foff.dh;
          for (int s = 0; s < MB; ++s) gh += tid < A ? dh[s * 34 + tid] : dh[s * 34 + 32 + (tid - A)];
        }
        if (tid < A) { h_which = 0; h_idx = D.off.mu_b + tid; }
        else if (tid == A) { h_which = 0; h_idx = D.off.v_b; }
        else { h_which = 1; h_idx = D.coff.v_b; }
        if (tid <= A) ss_ac += (gh * sc) * (gh * sc); else ss_cv += (gh * sc) * (gh * sc);
      }
      if (tid >= 64 && tid < 64 + A) {
        for (int r = 0; r < W; ++r) gh += D.fact_all[r * T + D.foff.dls + (tid - 64)];
        h_which = 0; h_idx = D.off.logstd + (tid - 64);
        ss_ac += (gh * sc) * (gh * sc);
      }
      if (tid == 128) {   // SUM of the ranks' minibatch KL, where the LR rule (and sdxp_apply(0, -INFINITY)) looks for it
        float kl = 0.0f;
        for (int r = 0; r < W; ++r) kl += D.fact_all[r * T + D.foff.kl];
        __hip_atomic_store(&D.ac_g[D.g_tail], kl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // ---- this workgroup's shares of the two squared norms (write_sqn_partials' order), then the grid-wide meeting: NO atomics.
  // (First versions: one device-scope counter - 1 380 arrivals served one after the other at the memory side: 105 us per optimiser step
  // against 68 with three launches; 64 counters: 87 us, the kernel alone 46.5 us against 7.8 + 18.4 + 4.7 for the three.)  The meeting is
  // two hops of tagged 16-byte words, the persistent kernel's exchange discipline (write-through store, polling agent-scope loads, the
  // tag - spread and folded with the payload so that a torn word is polled again - in the last dword), laid out so that the sums are
  // k_adam3's to the bit: every workgroup publishes (share_ac, share_cv); workgroup w < 4 plays wave w of k_adam3's fold - lane l adds
  // the shares 64 w + l + 256 k in ascending k, then wave_sum - and publishes that wave's sum; every workgroup gathers the four sums
  // and adds them as (s0 + s1) + (s2 + s3).
  ss_ac = wave_sum(ss_ac); ss_cv = wave_sum(ss_cv);
  if (lane == 0) { s_sq[0][wave] = ss_ac; s_sq[1][wave] = ss_cv; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the KL word of thread 128 has left before the barrier that precedes this workgroup's word
  __syncthreads();
  if (wave == 0) {
    const __amdgpu_buffer_rsrc_t Q = __builtin_amdgcn_make_buffer_rsrc(D.sqn_part + SDXP_TW_OFF, 0, (SDXP_SQN_STRIDE + 64) * 16, 0x00020000);
    const unsigned nwg = gridDim.x;
    const unsigned tagm = (tag_gen + 1u) * 0x9E3779B1u;
    int ok = 1;
    auto put = [&](unsigned idx, float x, float y) {
      tw_u32x4 w; w.x = __float_as_uint(x); w.y = __float_as_uint(y); w.z = 0u; w.w = tagm ^ w.x ^ w.y;
      __builtin_amdgcn_raw_buffer_store_b128(w, Q, idx * 16u, 0, 16);
    };
    // wave-uniform retry until every word idx0 + i * stride < limit carries this launch's tag; x / y: the words' payloads added in ascending i
    auto gather_sum = [&](unsigned idx0, unsigned stride, unsigned limit, float& x, float& y) {
      unsigned spins = 0;
      for (;;) {
        tw_u32x4 w[SDXP_SQN_STRIDE / 256];
#pragma unroll
        for (int i = 0; i < SDXP_SQN_STRIDE / 256; ++i) { const unsigned idx = idx0 + i * stride; w[i] = __builtin_amdgcn_raw_buffer_load_b128(Q, (idx < limit ? idx : idx0) * 16u, 0, 16); }
        bool good = true;
        x = 0.0f; y = 0.0f;
#pragma unroll
        for (int i = 0; i < SDXP_SQN_STRIDE / 256; ++i) {
          const bool act = idx0 + i * stride < limit;
          good = good && (!act || (w[i].w ^ w[i].x ^ w[i].y ^ w[i].z) == tagm);
          if (act) { x += __uint_as_float(w[i].x); y += __uint_as_float(w[i].y); }
        }
        if (__builtin_amdgcn_ballot_w64(!good) == 0) return;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0) {
          const unsigned f = __hip_atomic_load(failflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (spins > (1u << 20) || __builtin_amdgcn_readfirstlane(f) != 0) {
            if (lane == 0) __hip_atomic_store(failflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = 0;
            return;
          }
        }
      }
    };
    if (lane == 0) put(blockIdx.x, (s_sq[0][0] + s_sq[0][1]) + (s_sq[0][2] + s_sq[0][3]), (s_sq[1][0] + s_sq[1][1]) + (s_sq[1][2] + s_sq[1][3]));
    if (blockIdx.x < 4) {   // wave blockIdx.x of k_adam3's fold
      float x, y;
      const unsigned t = 64u * blockIdx.x + lane;
      gather_sum(t < nwg ? t : 0u, 256u, t < nwg ? nwg : 0u, x, y);
      x = wave_sum(x); y = wave_sum(y);
      if (ok && lane == 0) put(SDXP_SQN_STRIDE + blockIdx.x, x, y);
    }
    float x = 0.0f, y = 0.0f;
    if (ok) gather_sum(SDXP_SQN_STRIDE + (lane < 4 ? lane : 0), 64u, SDXP_SQN_STRIDE + 4u, x, y);   // (one word per lane: 0.0f + the sum, exact)
    const float x0 = lane0(x), y0 = lane0(y);
    const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 1)), y1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), 1));
    const float x2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 2)), y2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), 2));
    const float x3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 3)), y3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), 3));
    if (lane == 0) { s_ok = ok; s_n2[0] = (x0 + x1) + (x2 + x3); s_n2[1] = (y0 + y1) + (y2 + y3); }
  }
  __syncthreads();
  if (!s_ok) return;
  const float n2_ac = s_n2[0], n2_cv = s_n2[1];
  const float clip_ac = D.truncate_grads ? fminf(1.0f, D.grad_norm / (sqrtf(n2_ac) + 1e-6f)) : 1.0f;
  const float clip_cv = D.truncate_grads ? fminf(1.0f, D.grad_norm / (sqrtf(n2_cv) + 1e-6f)) : 1.0f;
  // ---- Adam of the elements this workgroup holds (k_adam3's arithmetic: adam1x(P, G * inv_w * clip, ...))
  if (l >= 0) {
    if (valid) {
      float* P = net == 2 ? D.cv : D.ac; float* M = net == 2 ? D.cv_m : D.ac_m; float* V = net == 2 ? D.cv_v : D.ac_v;
      const float clip = net == 2 ? clip_cv : clip_ac, lr_bc1 = net == 2 ? cv_lr_bc1 : ac_lr_bc1, isq = net == 2 ? cv_isq : ac_isq;
      const size_t row0 = woff + (size_t)n * K;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {   // eight columns per pass: their 24 loads in flight together, 80 VGPRs without a spill
        if (512 * hf >= K) break;
        float p[8], m[8], v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int kk = lane + 64 * (8 * hf + j);
          const int k = kk < K ? kk : K - 1;    // clamped address instead of a branch around the load
          p[j] = P[row0 + k]; m[j] = M[row0 + k]; v[j] = V[row0 + k];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = lane + 64 * (8 * hf + j);
          if (k < K) {
            P[row0 + k] = adam1x(p[j], g[8 * hf + j] * inv_w * clip, m[j], v[j], lr_bc1, isq);
            M[row0 + k] = m[j]; V[row0 + k] = v[j];
          }
        }
      }
      if (lane == 0) {
        float bm = M[boff + n], bv = V[boff + n];
        P[boff + n] = adam1x(P[boff + n], sb * inv_w * clip, bm, bv, lr_bc1, isq);
        M[boff + n] = bm; V[boff + n] = bv;
      }
    }
  } else if (h_which >= 0) {
    float* P = h_which ? D.cv : D.ac; float* M = h_which ? D.cv_m : D.ac_m; float* V = h_which ? D.cv_v : D.ac_v;
    float hm = M[h_idx], hv = V[h_idx];
    P[h_idx] = adam1x(P[h_idx], gh * inv_w * (h_which ? clip_cv : clip_ac), hm, hv, h_which ? cv_lr_bc1 : ac_lr_bc1, h_which ? cv_isq : ac_isq);
    M[h_idx] = hm; V[h_idx] = hv;
  }
  // ---- k_apply_fin2's work, behind the ticket
  if (blockIdx.x == 0 && tid == 0) {
    __hip_atomic_store(gen, tag_gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ctl->gn2_ac = n2_ac; ctl->gn2_cv = n2_cv;
    ctl->cv_t += 1; ctl->cv_b1pow *= 0.9; ctl->cv_b2pow *= 0.999; ctl->cv_gnorm = sqrtf(n2_cv);
    ctl->ac_t += 1; ctl->ac_b1pow *= 0.9; ctl->ac_b2pow *= 0.999; ctl->ac_gnorm = sqrtf(n2_ac);
    const float kl = __hip_atomic_load(&D.ac_g[D.g_tail], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / (float)ctl->world;
    if (D.adaptive_lr) {
      if (kl > 2.0f * D.kl_threshold) ctl->ac_lr = fmaxf(ctl->ac_lr / 1.5f, 1e-6f);
      if (kl < 0.5f * D.kl_threshold) ctl->ac_lr = fminf(ctl->ac_lr * 1.5f, 1e-2f);
    }
  }
}
// 1 when every workgroup of the one-launch apply can be resident at once on this device (its grid-wide ticket needs that); decided once per MB
template <int MB>
static int apply_fused_fits(const SdxpDev* D, int nb, size_t lds) {
  static int fits = -1;
  if (fits < 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    fits = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_apply_factors_fused<MB>, 256, lds) == hipSuccess)
      fits = (long long)per_cu * prop.multiProcessorCount >= nb ? 1 : 0;
  }
  return fits;
}
extern "C" int sdxpk_apply_factors_fused(const SdxpDev* D, int mb_size, unsigned* bar, hipStream_t st) {
  const int nhb = ((D->act_dim + 2) * D->units[2] + 255) / 256 + 1;
  const int nb = 3 * ((D->units[0] + 3) / 4) + 3 * ((D->units[1] + 3) / 4) + 3 * ((D->units[2] + 3) / 4) + nhb;
  const int Kmax = D->units[0] > D->state_dim ? D->units[0] : D->state_dim;
  if (nb > SDXP_SQN_STRIDE || Kmax > 1024 || D->obs_dim > 1024) return -1;
#define C_(M) { const size_t lds = (size_t)M * Kmax * sizeof(float); if (!apply_fused_fits<M>(D, nb, lds)) return -1; \
                hipLaunchKernelGGL(k_apply_factors_fused<M>, dim3(nb), dim3(256), lds, st, *D, bar + 36, bar + 34); }
  switch (mb_size) { case 2: C_(2) return 0; case 4: C_(4) return 0; default: return -1; }
#undef C_
}
// clip_grad_norm_ + Adam + LR schedule on the flat gradients already sitting in ac_g / cv_g (KL word in ac_g[g_tail])
extern "C" void sdxpk_apply_flat(const SdxpDev* D, hipStream_t st) {
  hipLaunchKernelGGL(k_sqnorm2, dim3(512, 2), dim3(256), 0, st, *D, 1.0f / (float)D->world);
  hipLaunchKernelGGL(k_adam2, dim3(512, 2), dim3(256), 0, st, *D);
  hipLaunchKernelGGL(k_apply_fin2, dim3(1), dim3(1), 0, st, *D);
}
extern "C" int sdxpk_apply_factors(const SdxpDev* D, int mb_size, hipStream_t st) {
#define C_(M) launch_apply_factors<M>(D, st)
  MB_SWITCH(mb_size, C_)
#undef C_
}
template <int MB>
static void launch_backward_factors(const SdxpDev* D, hipStream_t st) {
  launch_fwd_bwd<MB>(D, st);
  hipLaunchKernelGGL(k_pack_factors<MB>, dim3(8, 23), dim3(256), 0, st, *D);
  hipLaunchKernelGGL(k_ctrl<MB>, dim3(1), dim3(1024), 0, st, *D, 1 | 2 | 8);
}
extern "C" int sdxpk_backward_factors(const SdxpDev* D, int mb_size, hipStream_t st) {
#define C_(M) launch_backward_factors<M>(D, st)
  MB_SWITCH(mb_size, C_)
#undef C_
}
template <int MB>
static void launch_grads_from_factors(const SdxpDev* D, hipStream_t st) {
  for (int l = 0; l < 3; ++l) {
    const int Nl = D->units[l];
    const int Kmax = l == 0 ? (D->state_dim > D->obs_dim ? D->state_dim : D->obs_dim) : D->units[l - 1];
    hipLaunchKernelGGL(k_grad_layer_w<MB>, dim3(3 * ((Nl + 3) / 4)), dim3(256), (size_t)MB * Kmax * sizeof(float), st, *D, l);
  }
  hipLaunchKernelGGL(k_grad_heads_w<MB>, dim3(1), dim3(256), 0, st, *D);
}
extern "C" int sdxpk_grads_from_factors(const SdxpDev* D, int mb_size, hipStream_t st) {
#define C_(M) launch_grads_from_factors<M>(D, st)
  MB_SWITCH(mb_size, C_)
#undef C_
}
extern "C" void sdxpk_apply_explicit(const SdxpDev* D, int which, float kl, int world, hipStream_t st) {
  const size_t n = which ? D->coff.total : D->off.total;
  float* acc = which ? &D->ctrl->gn2_cv : &D->ctrl->gn2_ac;
  hipLaunchKernelGGL(k_sqnorm, dim3(512), dim3(256), 0, st, which ? D->cv_g : D->ac_g, n, 1.0f / (float)world, D->sqn_part);
  hipLaunchKernelGGL(k_sqnorm_fin, dim3(1), dim3(512), 0, st, D->sqn_part, 512, acc);
  hipLaunchKernelGGL(k_adam_explicit, dim3(1024), dim3(256), 0, st, *D, which);
  hipLaunchKernelGGL(k_apply_fin, dim3(1), dim3(1), 0, st, *D, which, kl);
}
