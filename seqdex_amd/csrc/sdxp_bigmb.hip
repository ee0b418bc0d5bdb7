// sdxp_bigmb.hip — large-minibatch PPO optimiser step (minibatch_size > 8) for gfx950.
//
// The reference trains the insert policy with minibatch_size 4096 (cfg/lego/ppo_continuous_insert.yaml:50,75) and BASELINE.md asks
// for a labelled large-minibatch variant next to the shipped minibatch of 4.  With thousands of samples per step the update is
// GEMM-shaped, not the rank-MB weight streaming of sdxp_kernels.hip / sdxp_persist.hip: forward Y = ELU(X W^T + b), data gradient
// dX = (dY W) * ELU'(H), weight gradient G = dY^T X, all three on the f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// k-ordered fma chain), LDS-tiled 128x128x32 per 256-thread workgroup (4 waves of 64x64; 64x64x16 for small problems), the
// same layer of the three networks in one launch.  rl_games' calc_gradients
// (a2c_continuous.py / RC:1796-1877) + CentralValueTrain.train_net for all three networks of one minibatch:
//
//   forward 3 nets x 3 trunk layers (NT GEMM, bias + ELU fused) -> mu head (NT GEMM) + two value heads (row dots)
//   k_big_head: per-sample PPO losses and their head gradients (same formulas as k_head of sdxp_kernels.hip), block partials
//   k_big_fin:  statistics, d logstd, KL word, minibatch bookkeeping of the control block
//   backward: head data gradients, then per net  G_l = dY_l^T X_l (TN GEMM, split over the minibatch rows, partials reduced in a
//   fixed order: deterministic), b_l = column sums of dY_l, dY_{l-1} = (dY_l W_l) * ELU'(H_{l-1}) (NN GEMM, fused)
//   -> flat gradients in ac_g / cv_g (parameter layout), then the explicit clip + Adam of sdxp_kernels.hip (k_sqnorm2 / k_adam2).
//
// The central-value input normalisation (RunningMeanStd in train mode: update with the minibatch, then normalise it, during
// mini-epoch 0; frozen afterwards) is hoisted out of the loop as in the small-minibatch path, with column statistics reduced in fp64.
#include "sdx_common.h"
#include "sdxp_types.h"
#include <cstddef>
#include <cstdint>
#include <cstdlib>

#include "sdx_gemm.h"
#include "sdx_gemm_nt.h"

// column sums of Y[M][N] (ld) over the row range of split blockIdx.y -> out[blockIdx.y * oz + n]
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ Y, int ld, int M, int N, int rchunk, float* __restrict__ out, size_t oz) {
  __shared__ float s[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const int rbeg = blockIdx.y * rchunk, rend = min(M, rbeg + rchunk);
  float a = 0.0f;
  if (c < N)
    for (int r = rbeg + rg; r < rend; r += 4) a += Y[(size_t)r * ld + c];
  s[rg][threadIdx.x & 63] = a;
  __syncthreads();
  if (rg == 0 && c < N) out[(size_t)blockIdx.y * oz + c] = (s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]);
}
// value heads: v[m] = H[m] . w + b (wave per row, U = 256: one float4 per lane)
__global__ __launch_bounds__(256) void k_rowdot(const float* __restrict__ H, int U, int M, const float* __restrict__ w, const float* __restrict__ b,
                                                float* __restrict__ v) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float a = 0.0f;
  for (int k = lane; k < U; k += 64) a += H[(size_t)row * U + k] * w[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0) v[row] = a + b[0];
}
// dY2[m][k] = dv[m] * w[k] * ELU'(H[m][k])
__global__ __launch_bounds__(256) void k_vhead_back(const float* __restrict__ dv, const float* __restrict__ w, const float* __restrict__ H, int U,
                                                    int M, float* __restrict__ dY) {
  const size_t n = (size_t)M * U;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / U), k = (int)(i % U);
    dY[i] = dv[m] * w[k] * belu_grad_from_out(H[i]);
  }
}
// value-head weight gradient partials: out[z][k] = sum_{m in split z} dv[m] H[m][k]; out[z][U] = sum dv
__global__ __launch_bounds__(256) void k_vhead_wgrad(const float* __restrict__ dv, const float* __restrict__ H, int U, int M, int rchunk,
                                                     float* __restrict__ out, size_t oz) {
  const int k = threadIdx.x;            // U == 256
  const int rbeg = blockIdx.x * rchunk, rend = min(M, rbeg + rchunk);
  float a = 0.0f, sb = 0.0f;
  for (int m = rbeg; m < rend; ++m) { const float d = dv[m]; a += d * H[(size_t)m * U + k]; sb += d; }
  out[(size_t)blockIdx.x * oz + k] = a;
  if (k == 0) out[(size_t)blockIdx.x * oz + U] = sb;
}

// out[q][i] = sum_z part[q][z * pz + i] for the two value heads (q = 0, 1): one wave per output element, lanes stride over the VS
// partials, fixed-order wave reduction (deterministic).  Round 3 used 2 blocks whose threads walked 128 partials one after the other
// (23 us of dependent loads per head for 257 numbers).
__global__ __launch_bounds__(256) void k_vhead_reduce(const float* __restrict__ p0, const float* __restrict__ p1, size_t pz, int VS, int n,
                                                      float* __restrict__ o0, float* __restrict__ o1) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= 2 * n) return;
  const float* part = w < n ? p0 : p1;
  const int i = w < n ? w : w - n;
  float a = 0.0f;
  for (int z = lane; z < VS; z += 64) a += part[(size_t)z * pz + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0) (w < n ? o0 : o1)[i] = a;
}

// out[i] = sum_z part[z * pz + i], i < n: one wave per output element, lanes stride over the S partials (fixed order: deterministic)
__global__ __launch_bounds__(256) void k_wave_reduce(const float* __restrict__ part, size_t pz, int S, int n, float* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  float a = 0.0f;
  for (int z = lane; z < S; z += 64) a += part[(size_t)z * pz + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0) out[i] = a;
}

// ------------------------------------------------------------------------------------------------ losses
// thread = sample s of the minibatch (dataset row r0 + s).  Formulas as k_head of sdxp_kernels.hip (RC:1796-1830, 2114-2126):
// Gaussian neglogp, clipped surrogate, (clipped) value losses of the critic and the central value, bound loss, KL to the stored
// mu/sigma.  Writes dmu [MB][24] (column 23 = 0), dv [2][MB], the refreshed mu/sigma rows (RC:1358) and block partials
// part[block][40]: 0..22 d logstd, 32..37 the six loss sums.
#define BIGP 40
__global__ __launch_bounds__(256) void k_big_head(SdxpDev D, size_t r0, int MB, const float* __restrict__ mu, const float* __restrict__ vc,
                                                  const float* __restrict__ vcv, float* __restrict__ dmu, float* __restrict__ dv,
                                                  float* __restrict__ part) {
  __shared__ float s_red[4][BIGP];
  const int s = blockIdx.x * 256 + threadIdx.x, A = D.act_dim, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = s < MB;
  const float invM = 1.0f / (float)MB;
  float nlp = 0.0f, kl = 0.0f, bl = 0.0f, ent = 0.0f, gnlp = 0.0f;
  float stat[6] = {0, 0, 0, 0, 0, 0};
  const float* ls_p = D.ac + D.off.logstd;
  if (live) {
    const size_t r = r0 + s;
    for (int a = 0; a < A; ++a) {
      const float ls = ls_p[a], sg = expf(ls), m = mu[(size_t)s * 24 + a];
      const float z = (D.mb_actions[r * A + a] - m) / sg;
      nlp += 0.5f * z * z + ls;
      const float omu = D.mb_mus[r * A + a], osg = D.mb_sigmas[r * A + a];
      kl += logf(osg / sg + 1e-5f) + (sg * sg + (omu - m) * (omu - m)) / (2.0f * (osg * osg + 1e-5f)) - 0.5f;
      const float hi = fmaxf(m - 1.1f, 0.0f), lo = fminf(m + 1.1f, 0.0f);
      bl += hi * hi + lo * lo;
      ent += 0.5f + 0.5f * 1.8378770664093453f + ls;
    }
    nlp += 0.5f * 1.8378770664093453f * (float)A;
    const float adv = D.adv[r];
    const float ratio = expf(D.mb_neglogp[r] - nlp);
    const float L1 = -adv * ratio, L2 = -adv * clampf(ratio, 1.0f - D.e_clip, 1.0f + D.e_clip);
    const bool inr = ratio >= 1.0f - D.e_clip && ratio <= 1.0f + D.e_clip;
    gnlp = (L1 > L2 || inr) ? adv * ratio : 0.0f;
    const float R = D.returns[r], vo = D.mb_values[r];
    float closs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float v = j == 0 ? vc[s] : vcv[s];
      const float vcl = vo + clampf(v - vo, -D.e_clip, D.e_clip);
      const float c1 = (v - R) * (v - R), c2 = (vcl - R) * (vcl - R);
      float d;
      if (D.clip_value) {
        closs[j] = fmaxf(c1, c2);
        const bool inv = fabsf(v - vo) <= D.e_clip;
        d = (c1 > c2 || inv) ? 2.0f * (v - R) : 0.0f;
      } else { closs[j] = c1; d = 2.0f * (v - R); }
      dv[(size_t)j * MB + s] = (j == 0 ? 0.5f * D.critic_coef : 1.0f) * d * invM;
    }
    stat[0] = fmaxf(L1, L2); stat[1] = closs[0]; stat[2] = bl; stat[3] = kl; stat[4] = closs[1]; stat[5] = ent;
  }
  // head gradients + d logstd partial sums (one wave reduction per action)
  for (int a = 0; a < 24; ++a) {
    float dls = 0.0f;
    if (live) {
      float d = 0.0f;
      if (a < A) {
        const size_t r = r0 + s;
        const float ls = ls_p[a], sg = expf(ls), m = mu[(size_t)s * 24 + a];
        const float z = (D.mb_actions[r * A + a] - m) / sg;
        const float hi = fmaxf(m - 1.1f, 0.0f), lo = fminf(m + 1.1f, 0.0f);
        d = gnlp * (-(z / sg)) * invM + D.bounds_coef * (2.0f * hi + 2.0f * lo) * invM;
        dls = gnlp * (1.0f - z * z) * invM;
        D.mb_mus[r * A + a] = m;
        D.mb_sigmas[r * A + a] = sg;
      }
      dmu[(size_t)s * 24 + a] = d;
    }
    if (a < A) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) dls += __shfl_xor(dls, o, 64);
      if (lane == 0) s_red[wave][a] = dls;
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float t = stat[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane == 0) s_red[wave][32 + j] = t;
  }
  __syncthreads();
  if (threadIdx.x < BIGP) {
    const int j = threadIdx.x;
    const bool used = j < A || (j >= 32 && j < 38);
    part[(size_t)blockIdx.x * BIGP + j] = used ? (s_red[0][j] + s_red[1][j]) + (s_red[2][j] + s_red[3][j]) : 0.0f;
  }
}
// one block: fold the block partials, write d logstd into the flat gradient, the KL word, the loss sums and advance the minibatch
// cursor of the control block (what k_ctrl does in explicit mode for the small-minibatch path)
__global__ __launch_bounds__(64) void k_big_fin(SdxpDev D, int nblocks, int MB, const float* __restrict__ part) {
  __shared__ float s_t[BIGP];
  const int j = threadIdx.x;
  if (j < BIGP) {
    float t = 0.0f;
    for (int b = 0; b < nblocks; ++b) t += part[(size_t)b * BIGP + j];
    s_t[j] = t;
  }
  __syncthreads();
  if (j < D.act_dim) D.ac_g[D.off.logstd + j] = s_t[j] - D.entropy_coef;   // d(-coef * mean entropy)/d logstd = -coef
  if (j == 0) {
    SdxpCtrl* ctl = D.ctrl;
    const float invM = 1.0f / (float)MB;
    const float kl = s_t[35] * invM;
    for (int q = 0; q < 6; ++q) ctl->acc[1 + q] = s_t[32 + q];
    ctl->sum_a_loss += s_t[32] * invM; ctl->sum_c_loss += s_t[33] * invM; ctl->sum_b_loss += s_t[34] * invM;
    ctl->sum_kl += kl; ctl->sum_cv_loss += s_t[36] * invM; ctl->sum_entropy += s_t[37] * invM;
    ctl->n_mb += 1; ctl->last_kl = kl;
    D.ac_g[D.g_tail] = kl;                        // rides with the gradients (multi-rank all-reduce), read by k_apply_fin2
    ctl->gn2_ac = 0.0f; ctl->gn2_cv = 0.0f; ctl->ac_pending = 0; ctl->cv_pending = 0;
    ctl->prev_mb = ctl->mb_index; ctl->prev_mini_epoch = ctl->mini_epoch;
    int mbn = ctl->mb_index + 1;
    if (mbn >= D.num_minibatches) { mbn = 0; ctl->mini_epoch += 1; }
    ctl->mb_index = mbn;
    ctl->step += 1;
  }
}

// ------------------------------------------------------------------------------------------------ fused heads (round 6)
// Everything between the last trunk layer's outputs and the trunk's backward pass in ONE launch (+ one reduction): the policy head
// mu = H_a Wmu^T + b and the two value heads, the per-sample losses of k_big_head, the head gradients dmu / dv, the data gradients
// dY_2 = (dmu Wmu) * ELU'(H_a), dv w * ELU'(H) of the three networks, and the heads' weight-gradient partials.  Round 5 ran this as 13
// launches (one 23-column GEMM forward, two row-dot kernels, the loss kernel, its fold, a 23-deep GEMM backward, two value-head backward
// kernels, a split GEMM + two split kernels for the weight gradients, two reductions): 126 us of a 630 us optimiser step at 2 048
// rows for 1 % of its flops (profiles/r6_bigmb_kernel_stats_mb2048_before.csv).  A workgroup owns HR = 32 consecutive minibatch rows (times
// `nrb` such blocks, one after the other, for large minibatches): the three trunk outputs of those rows (96 KB), Wmu and the value
// weights live in LDS; rows are padded to 260 floats so that 16-byte reads of 8 different rows cover the 32 banks once.
//   phase 1  thread = (row, 3 actions): mu, 16-byte LDS reads along k; wave 0 also does the two value heads
//   phase 2  thread = row (32 lanes): the loss formulas of k_big_head, dmu / dv into LDS, loss sums folded over the 32 lanes
//   phase 3  thread = trunk column k: dY_2 of the three networks for the 32 rows (Wmu's column in registers, dmu broadcast from LDS) and
//            the column's weight-gradient sums, which stay in registers across the workgroup's row blocks
// Partials per workgroup: [40: d logstd sums 0 .. 22, loss sums 24 .. 29 | G_mu [A][256] | b_mu [A] | g_v [256] b_v | g_cv [256] b_cv]; k_big_heads_reduce folds them over
// the workgroups in index order (deterministic) into the flat gradients and does k_big_fin's bookkeeping.
#define HR 32
#define HU 256              // trunk output width the fused kernel is written for (units[2] of both shipped YAMLs); else the 13-launch path
#define HLD 260             // padded LDS row
#define HPZ (BIGP + 23 * HU + 23 + 2 * (HU + 1))
struct HeadsLds {
  float H[3][HR][HLD];
  float Wm[23][HLD];
  float wv[2][HU];
  float bmu[24], bv[4];   // (bv padded: mu / dmu rows stay 16-byte aligned)
  float mu[HR][24], dmu[HR][24];
  float v[2][HR], dv[2][HR];
  float red[BIGP], redw[4][BIGP];
};
__global__ __launch_bounds__(512) void k_big_heads(SdxpDev D, size_t r0, int MB, int nrb, const float* __restrict__ Ha, const float* __restrict__ Hc,
                                                   const float* __restrict__ Hcv, float* __restrict__ dYa, float* __restrict__ dYc,
                                                   float* __restrict__ dYcv, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char heads_smem[];
  HeadsLds& L = *reinterpret_cast<HeadsLds*>(heads_smem);
  const int t = threadIdx.x, A = D.act_dim;
  const float* Hsrc[3] = {Ha, Hc, Hcv};
  // weights of the heads (once per workgroup)
  constexpr int NTH = 512;   // 8 waves: two per SIMD (one workgroup per CU: 130 KB of LDS)
  for (int i = t; i < 23 * (HU / 4); i += NTH) {
    const int a = i / (HU / 4), k4 = i % (HU / 4);
    float4 w = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (a < A) w = *reinterpret_cast<const float4*>(D.ac + D.off.mu_w + (size_t)a * HU + 4 * k4);
    *reinterpret_cast<float4*>(&L.Wm[a][4 * k4]) = w;
  }
  if (t < HU) { L.wv[0][t] = D.ac[D.off.v_w + t]; L.wv[1][t] = D.cv[D.coff.v_w + t]; }
  if (t < 24) L.bmu[t] = t < A ? D.ac[D.off.mu_b + t] : 0.0f;
  if (t == 32) L.bv[0] = D.ac[D.off.v_b];
  if (t == 33) L.bv[1] = D.cv[D.coff.v_b];
  if (t < BIGP) L.red[t] = 0.0f;
  // phase-3 accumulators of column (t & 255) over the row half (t >> 8) of every block, kept across the row blocks
  float G[23], gv0 = 0.0f, gv1 = 0.0f, bsum = 0.0f;   // bsum: thread a < 23: sum of dmu[.][a]; threads 23, 24: sum of dv[0 / 1][.]
#pragma unroll
  for (int a = 0; a < 23; ++a) G[a] = 0.0f;
  const float invM = 1.0f / (float)MB;
  const float* ls_p = D.ac + D.off.logstd;
  for (int rb = 0; rb < nrb; ++rb) {
    const int s0 = (blockIdx.x * nrb + rb) * HR;
    if (s0 >= MB) break;                                  // block-uniform
    __syncthreads();                                      // the previous block's phase 3 is done with H / dmu / dv
    // ---- phase 0: the three trunk outputs of rows s0 .. s0 + 31 (coalesced 16-byte loads; rows past the minibatch read as zeros)
    for (int i = t; i < 3 * HR * (HU / 4); i += NTH) {
      const int net = i / (HR * (HU / 4)), rr = (i / (HU / 4)) % HR, k4 = i % (HU / 4);
      float4 h = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (s0 + rr < MB) h = *reinterpret_cast<const float4*>(Hsrc[net] + (size_t)(s0 + rr) * HU + 4 * k4);
      *reinterpret_cast<float4*>(&L.H[net][rr][4 * k4]) = h;
    }
    __syncthreads();
    // ---- phase 1: heads forward
    {
      const int rr = (t & 255) >> 3, g = t & 7, kh = t >> 8;   // the two halves of k on the two halves of the workgroup
      float acc[3] = {0.0f, 0.0f, 0.0f};
      const int a0 = g, a1 = g + 8, a2 = g + 16 < 23 ? g + 16 : 22;
#pragma unroll 4
      for (int k4 = kh * (HU / 8); k4 < (kh + 1) * (HU / 8); ++k4) {
        const float4 h = *reinterpret_cast<const float4*>(&L.H[0][rr][4 * k4]);
        const float4 w0 = *reinterpret_cast<const float4*>(&L.Wm[a0][4 * k4]), w1 = *reinterpret_cast<const float4*>(&L.Wm[a1][4 * k4]),
                     w2 = *reinterpret_cast<const float4*>(&L.Wm[a2][4 * k4]);
        acc[0] += h.x * w0.x; acc[0] += h.y * w0.y; acc[0] += h.z * w0.z; acc[0] += h.w * w0.w;
        acc[1] += h.x * w1.x; acc[1] += h.y * w1.y; acc[1] += h.z * w1.z; acc[1] += h.w * w1.w;
        acc[2] += h.x * w2.x; acc[2] += h.y * w2.y; acc[2] += h.z * w2.z; acc[2] += h.w * w2.w;
      }
      if (kh == 1) { L.dmu[rr][a0] = acc[0]; L.dmu[rr][a1] = acc[1]; if (g + 16 < 23) L.dmu[rr][a2] = acc[2]; }   // (dmu as staging: phase 2 rewrites it)
      // value heads: (net, row, 8 slices of k) on the upper half's first 512 ... = all 512 threads: 2 x 32 x 8
      float av = 0.0f;
      {
        const int net = t >> 8, r2 = (t & 255) >> 3, ks = t & 7;
#pragma unroll
        for (int k4 = ks * (HU / 32); k4 < (ks + 1) * (HU / 32); ++k4) {
          const float4 h = *reinterpret_cast<const float4*>(&L.H[1 + net][r2][4 * k4]);
          const float4 w = *reinterpret_cast<const float4*>(&L.wv[net][4 * k4]);
          av += h.x * w.x; av += h.y * w.y; av += h.z * w.z; av += h.w * w.w;
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) av += __shfl_xor(av, o, 64);
        if (ks == 0) L.v[net][r2] = av + L.bv[net];
      }
      __syncthreads();
      if (kh == 0) {
        L.mu[rr][a0] = (acc[0] + L.dmu[rr][a0]) + L.bmu[a0];
        L.mu[rr][a1] = (acc[1] + L.dmu[rr][a1]) + L.bmu[a1];
        if (g + 16 < 23) L.mu[rr][a2] = (acc[2] + L.dmu[rr][a2]) + L.bmu[a2];
        if (g == 7) L.mu[rr][23] = 0.0f;
      }
    }
    __syncthreads();
    // ---- phase 2: losses and head gradients (formulas: k_big_head; RC:1796-1830, 2114-2126).  Thread = (row rr, action group g): the
    // actions g, g + 8, g + 16 of row s0 + rr; the per-row sums over the actions are folded over the 8 lanes of the row (xor 1, 2, 4),
    // the per-action sums over the rows over the wave's 8 rows (xor 8, 16, 32), then over the 4 waves in index order
    if (t < 256) {
      const int rr = t >> 3, g = t & 7, s = s0 + rr, wave = t >> 6;
      const bool live = s < MB;
      const size_t r = r0 + (live ? s : 0);
      float nlp = 0.0f, kl = 0.0f, bl = 0.0f, ent = 0.0f;
      float zq[3], sgq[3], mq[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int a = g + 8 * j;
        zq[j] = 0.0f; sgq[j] = 1.0f; mq[j] = 0.0f;
        if (live && a < A) {
          const float ls = ls_p[a], sg = expf(ls), m = L.mu[rr][a];
          const float z = (D.mb_actions[r * A + a] - m) / sg;
          nlp += 0.5f * z * z + ls;
          const float omu = D.mb_mus[r * A + a], osg = D.mb_sigmas[r * A + a];
          kl += logf(osg / sg + 1e-5f) + (sg * sg + (omu - m) * (omu - m)) / (2.0f * (osg * osg + 1e-5f)) - 0.5f;
          const float hi = fmaxf(m - 1.1f, 0.0f), lo = fminf(m + 1.1f, 0.0f);
          bl += hi * hi + lo * lo;
          ent += 0.5f + 0.5f * 1.8378770664093453f + ls;
          zq[j] = z; sgq[j] = sg; mq[j] = m;
        }
      }
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) { nlp += __shfl_xor(nlp, o, 64); kl += __shfl_xor(kl, o, 64); bl += __shfl_xor(bl, o, 64); ent += __shfl_xor(ent, o, 64); }
      nlp += 0.5f * 1.8378770664093453f * (float)A;
      float gnlp = 0.0f;
      float stat[6] = {0, 0, 0, 0, 0, 0};
      if (live) {
        const float adv = D.adv[r];
        const float ratio = expf(D.mb_neglogp[r] - nlp);
        const float L1 = -adv * ratio, L2 = -adv * clampf(ratio, 1.0f - D.e_clip, 1.0f + D.e_clip);
        const bool inr = ratio >= 1.0f - D.e_clip && ratio <= 1.0f + D.e_clip;
        gnlp = (L1 > L2 || inr) ? adv * ratio : 0.0f;
        if (g == 0) {
          const float R = D.returns[r], vo = D.mb_values[r];
          float closs[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float v = L.v[j][rr];
            const float vcl = vo + clampf(v - vo, -D.e_clip, D.e_clip);
            const float c1 = (v - R) * (v - R), c2 = (vcl - R) * (vcl - R);
            float d;
            if (D.clip_value) {
              closs[j] = fmaxf(c1, c2);
              const bool inv = fabsf(v - vo) <= D.e_clip;
              d = (c1 > c2 || inv) ? 2.0f * (v - R) : 0.0f;
            } else { closs[j] = c1; d = 2.0f * (v - R); }
            L.dv[j][rr] = (j == 0 ? 0.5f * D.critic_coef : 1.0f) * d * invM;
          }
          stat[0] = fmaxf(L1, L2); stat[1] = closs[0]; stat[2] = bl; stat[3] = kl; stat[4] = closs[1]; stat[5] = ent;
        }
      } else if (g == 0) { L.dv[0][rr] = 0.0f; L.dv[1][rr] = 0.0f; }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int a = g + 8 * j;
        float d = 0.0f, dls = 0.0f;
        if (live && a < A) {
          const float z = zq[j], sg = sgq[j], m = mq[j];
          const float hi = fmaxf(m - 1.1f, 0.0f), lo = fminf(m + 1.1f, 0.0f);
          d = gnlp * (-(z / sg)) * invM + D.bounds_coef * (2.0f * hi + 2.0f * lo) * invM;
          dls = gnlp * (1.0f - z * z) * invM;
          D.mb_mus[r * A + a] = m;
          D.mb_sigmas[r * A + a] = sg;
        }
        if (a < 24) L.dmu[rr][a] = d;
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) dls += __shfl_xor(dls, o, 64);
        if (a < 24 && (t & 56) == 0) L.redw[wave][a] = dls;   // lanes 0 .. 7 of the wave hold the sums over its 8 rows
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        float x = stat[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        if ((t & 63) == 0) L.redw[wave][32 + j] = x;
      }
    }
    __syncthreads();
    if (t < BIGP && (t < 24 || (t >= 32 && t < 38))) L.red[t] += (L.redw[0][t] + L.redw[1][t]) + (L.redw[2][t] + L.redw[3][t]);
    // ---- phase 3: thread = trunk column t
    {
      const int col = t & 255, rh = t >> 8;
      float wcol[23];
#pragma unroll
      for (int a = 0; a < 23; ++a) wcol[a] = L.Wm[a][col];
      const float wv0 = L.wv[0][col], wv1 = L.wv[1][col];
      const int nrow = MB - s0 < HR ? MB - s0 : HR;
      const int rbeg = rh * (HR / 2), rend = nrow < (rh + 1) * (HR / 2) ? nrow : (rh + 1) * (HR / 2);
      for (int rr = rbeg; rr < rend; ++rr) {
        float d[24];
#pragma unroll
        for (int q4 = 0; q4 < 6; ++q4) {
          const float4 x = *reinterpret_cast<const float4*>(&L.dmu[rr][4 * q4]);   // every lane reads the same row: LDS broadcast
          d[4 * q4] = x.x; d[4 * q4 + 1] = x.y; d[4 * q4 + 2] = x.z; d[4 * q4 + 3] = x.w;
        }
        const float ha = L.H[0][rr][col], hc = L.H[1][rr][col], hcv = L.H[2][rr][col];
        float x = 0.0f;
#pragma unroll
        for (int a = 0; a < 23; ++a) { x += d[a] * wcol[a]; G[a] += d[a] * ha; }
        const size_t o = (size_t)(s0 + rr) * HU + col;
        dYa[o] = x * belu_grad_from_out(ha);
        const float d0 = L.dv[0][rr], d1 = L.dv[1][rr];
        dYc[o] = d0 * wv0 * belu_grad_from_out(hc);
        dYcv[o] = d1 * wv1 * belu_grad_from_out(hcv);
        gv0 += d0 * hc; gv1 += d1 * hcv;
      }
      if (t < 23) for (int rr = 0; rr < nrow; ++rr) bsum += L.dmu[rr][t];
      else if (t < 25) for (int rr = 0; rr < nrow; ++rr) bsum += L.dv[t - 23][rr];
    }
  }
  __syncthreads();
  // the upper row half's column sums through LDS (the H rows are dead), added to the lower half's in a fixed order
  float* X = &L.H[0][0][0];   // [25][256]
  if (t >= 256) {
#pragma unroll
    for (int a = 0; a < 23; ++a) X[a * HU + (t & 255)] = G[a];
    X[23 * HU + (t & 255)] = gv0; X[24 * HU + (t & 255)] = gv1;
  }
  __syncthreads();
  if (t >= 256) return;
  float* P = part + (size_t)blockIdx.x * HPZ;
  if (t < BIGP) P[t] = t < 24 ? L.red[t] : (t < 30 ? L.red[t + 8] : 0.0f);   // [0..22] d logstd, [24..29] the six loss sums: all inside the reduction's first block of 32 elements
#pragma unroll
  for (int a = 0; a < 23; ++a) P[BIGP + a * HU + t] = G[a] + X[a * HU + t];
  if (t < 23) P[BIGP + 23 * HU + t] = bsum;
  P[BIGP + 23 * HU + 23 + t] = gv0 + X[23 * HU + t];
  P[BIGP + 23 * HU + 23 + HU + 1 + t] = gv1 + X[24 * HU + t];
  if (t == 23) P[BIGP + 23 * HU + 23 + HU] = bsum;
  if (t == 24) P[BIGP + 23 * HU + 23 + HU + 1 + HU] = bsum;
}
// fold of the heads' partials over the workgroups (index order: deterministic) into the flat gradients + what k_big_fin does
__global__ __launch_bounds__(256) void k_big_heads_reduce(SdxpDev D, int nblocks, int MB, const float* __restrict__ part) {
  __shared__ float s_t[BIGP];
  // eight lanes per output element: lane l sums the workgroups b = l, l + 8, ... in order, the eight sums are added in a fixed tree
  const int i = blockIdx.x * 32 + (threadIdx.x >> 3), l = threadIdx.x & 7, A = D.act_dim;
  float x = 0.0f;
  if (i < HPZ) for (int b = l; b < nblocks; b += 8) x += part[(size_t)b * HPZ + i];
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) x += __shfl_xor(x, o, 64);
  if (l != 0) { if (blockIdx.x != 0) return; }   // (block 0's helper lanes stay for its barrier)
  const int gi = i - BIGP;
  if (l == 0 && i >= BIGP && i < HPZ) {
    if (gi < 23 * HU + 23) {   // [mu_w | mu_b] is contiguous in the flat layout; rows a >= act_dim of the partials are zeros and have no slot
      const int a = gi < 23 * HU ? gi / HU : gi - 23 * HU;
      if (a < A) D.ac_g[D.off.mu_w + (gi < 23 * HU ? (size_t)gi : (size_t)A * HU + a)] = x;
    } else if (gi < 23 * HU + 23 + HU + 1) D.ac_g[D.off.v_w + (gi - (23 * HU + 23))] = x;
    else D.cv_g[D.coff.v_w + (gi - (23 * HU + 23 + HU + 1))] = x;
  }
  if (blockIdx.x != 0) return;
  if (l == 0 && i < 32) s_t[i] = x;    // block 0 folds elements 0 .. 31: the d logstd sums 0 .. 22 and the six loss sums 24 .. 29
  __syncthreads();
  const int j = threadIdx.x;
  if (j < A) D.ac_g[D.off.logstd + j] = s_t[j] - D.entropy_coef;   // d(-coef * mean entropy)/d logstd = -coef
  if (j == 0) {
    SdxpCtrl* ctl = D.ctrl;
    const float invM = 1.0f / (float)MB;
    const float kl = s_t[27] * invM;
    for (int q = 0; q < 6; ++q) ctl->acc[1 + q] = s_t[24 + q];
    ctl->sum_a_loss += s_t[24] * invM; ctl->sum_c_loss += s_t[25] * invM; ctl->sum_b_loss += s_t[26] * invM;
    ctl->sum_kl += kl; ctl->sum_cv_loss += s_t[28] * invM; ctl->sum_entropy += s_t[29] * invM;
    ctl->n_mb += 1; ctl->last_kl = kl;
    D.ac_g[D.g_tail] = kl;
    ctl->gn2_ac = 0.0f; ctl->gn2_cv = 0.0f; ctl->ac_pending = 0; ctl->cv_pending = 0;
    ctl->prev_mb = ctl->mb_index; ctl->prev_mini_epoch = ctl->mini_epoch;
    int mbn = ctl->mb_index + 1;
    if (mbn >= D.num_minibatches) { mbn = 0; ctl->mini_epoch += 1; }
    ctl->mb_index = mbn;
    ctl->step += 1;
  }
}
static int heads_nrb(int MB) { int n = MB / (HR * 512); return n < 1 ? 1 : n; }
static int heads_blocks(int MB) { const int nrb = heads_nrb(MB); return (MB + HR * nrb - 1) / (HR * nrb); }
static bool heads_fused_ok(const SdxpDev& D) {
  static const int off = getenv("SDXP_BIGMB_FUSED_HEADS") ? atoi(getenv("SDXP_BIGMB_FUSED_HEADS")) == 0 : 0;   // =0: the 13-launch head section (diagnosis)
  return !off && D.units[2] == HU && D.act_dim <= 23;
}

// ------------------------------------------------------------------------------------------------ central-value input statistics
// partial column sums / sums of squares (fp64) of rows [r0 + z * rchunk, ...) of mb_states
__global__ __launch_bounds__(256) void k_big_colstats(SdxpDev D, size_t r0, int MB, int rchunk, double* __restrict__ part) {
  __shared__ double s1[4][64], s2[4][64];
  const int S = D.state_dim, c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const int rbeg = blockIdx.y * rchunk, rend = min(MB, rbeg + rchunk);
  double a = 0.0, q = 0.0;
  if (c < S)
    for (int r = rbeg + rg; r < rend; r += 4) { const double x = D.mb_states[(r0 + r) * S + c]; a += x; q += x * x; }
  s1[rg][threadIdx.x & 63] = a; s2[rg][threadIdx.x & 63] = q;
  __syncthreads();
  if (rg == 0 && c < S) {
    const int t = threadIdx.x;
    part[((size_t)blockIdx.y * S + c) * 2 + 0] = (s1[0][t] + s1[1][t]) + (s1[2][t] + s1[3][t]);
    part[((size_t)blockIdx.y * S + c) * 2 + 1] = (s2[0][t] + s2[1][t]) + (s2[2][t] + s2[3][t]);
  }
}
// RunningMeanStd.update with the minibatch statistics (unbiased batch variance, parallel-variance merge), thread = feature
__global__ __launch_bounds__(256) void k_big_rms_update(SdxpDev D, int MB, int nsplit, const double* __restrict__ part) {
  const int k = blockIdx.x * 256 + threadIdx.x, S = D.state_dim;
  if (k >= S) return;
  double a = 0.0, q = 0.0;
  for (int z = 0; z < nsplit; ++z) { a += part[((size_t)z * S + k) * 2]; q += part[((size_t)z * S + k) * 2 + 1]; }
  const double bm = a / MB;
  double bv = MB > 1 ? (q - MB * bm * bm) / (MB - 1) : 0.0;
  if (bv < 0.0) bv = 0.0;
  const double cnt = D.ctrl->rms_count, mean = D.rms_mean[k], var = D.rms_var[k];
  const double delta = bm - mean, tot = cnt + MB;
  const double m2 = var * cnt + bv * MB + delta * delta * cnt * MB / tot;
  D.rms_mean[k] = mean + delta * MB / tot;
  D.rms_var[k] = m2 / tot;
}
__global__ void k_big_rms_count(SdxpDev D, int MB) { D.ctrl->rms_count += (double)MB; }
// dst rows = clamp((x - mean) / sqrt(var + 1e-5), +-5) of mb_states rows [r0, r0 + rows)
__global__ __launch_bounds__(256) void k_big_normalise(SdxpDev D, size_t r0, size_t rows, float* __restrict__ dst) {
  const int S = D.state_dim;
  const size_t base = r0 * S, total = rows * S;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)((base + i) % S);
    const float x = D.mb_states[base + i];
    const float fm = (float)D.rms_mean[k], rs = sqrtf((float)D.rms_var[k] + 1e-5f);
    dst[base + i] = D.cv_normalize_input ? clampf((x - fm) / rs, -5.0f, 5.0f) : x;
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int big_splits(int MB) { int s = (MB + 511) / 512; return s < 1 ? 1 : (s > 16 ? 16 : s); }

extern "C" size_t sdxpk_big_part_floats(const SdxpDev* D, int MB) {
  const int S = big_splits(MB);
  const size_t in0 = D->obs_dim > D->state_dim ? D->obs_dim : D->state_dim;
  size_t mx = (size_t)D->units[0] * in0 + D->units[0];
  const size_t l1 = (size_t)D->units[1] * D->units[0] + D->units[1], l2 = (size_t)D->units[2] * D->units[1] + D->units[2];
  if (l1 > mx) mx = l1;
  if (l2 > mx) mx = l2;
  int Sh = MB / 256;                 // splits of the policy head's weight gradient (sdxpk_big_step)
  Sh = Sh < S ? S : (Sh > 128 ? 128 : Sh);
  const size_t hp = (size_t)((MB + 255) / 256) * BIGP + (size_t)2 * ((MB + 63) / 64) * (D->units[2] + 1) + (size_t)Sh * (32 * (D->units[2] + 1));
  const size_t need = (size_t)S * mx;
  const size_t hf = ((size_t)heads_blocks(MB) * HPZ + 2) / 3;   // fused heads: their partials span the three regions
  const size_t m1 = need > hp ? need : hp;
  return m1 > hf ? m1 : hf;          // per network; the workspace holds three such regions
}
extern "C" int sdxpk_big_nsplit(int MB) { return big_splits(MB); }
// The NT path (sdx_gemm_nt.h: staged operands, global_load_lds, two LDS stages) serves the trunk products unless SDXP_BIGMB_NT=0 asks for
// the round-1 kernel (k_gemm of sdx_gemm.h; kept for A/B timing).  It needs 16-byte aligned minibatch windows in the transposed inputs.
extern "C" int sdxpk_big_nt_enabled(const SdxpDev* D, int MB) {
  const char* e = getenv("SDXP_BIGMB_NT");
  if (e && e[0] == '0') return 0;
  return MB % 8 == 0 && D->units[0] % 128 == 0 && D->units[1] % 128 == 0 && D->units[2] % 128 == 0 && D->obs_dim % 4 == 0 && D->state_dim % 4 == 0;
}
template <int BF>
static void stage_launch(const StageArgs* a, int count, hipStream_t st) {
  StageBatch sb;
  int Rx = 0, Kx = 0;
  for (int q = 0; q < 9; ++q) {
    sb.a[q] = a[q < count ? q : 0];
    if (q < count) { Rx = a[q].R > Rx ? a[q].R : Rx; Kx = a[q].Kp > Kx ? a[q].Kp : Kx; }
  }
  hipLaunchKernelGGL((k_stage<BF>), dim3((Kx + 63) / 64, (Rx + 63) / 64, count), dim3(256), 0, st, sb);
}
static void stage(const SdxpDev& D, const StageArgs* a, int count, hipStream_t st) {
  if (D.bf16) stage_launch<1>(a, count, st); else stage_launch<0>(a, count, st);
}

// central-value inputs of the whole epoch: cvx0 (statistics updated minibatch by minibatch, mini-epoch 0), cvx1 (frozen)
extern "C" void sdxpk_big_prenorm(const SdxpDev* D, const SdxpBigWs* ws, hipStream_t st) {
  const int MB = ws->MB, S = D->state_dim, nsplit = ws->nsplit, rchunk = (MB + nsplit - 1) / nsplit;
  for (int mb = 0; mb < D->num_minibatches; ++mb) {
    const size_t r0 = (size_t)mb * MB;
    if (D->cv_normalize_input) {
      hipLaunchKernelGGL(k_big_colstats, dim3((S + 63) / 64, nsplit), dim3(256), 0, st, *D, r0, MB, rchunk, ws->dpart);
      hipLaunchKernelGGL(k_big_rms_update, dim3((S + 255) / 256), dim3(256), 0, st, *D, MB, nsplit, ws->dpart);
      hipLaunchKernelGGL(k_big_rms_count, dim3(1), dim3(1), 0, st, *D, MB);
    }
    hipLaunchKernelGGL(k_big_normalise, dim3(1024), dim3(256), 0, st, *D, r0, (size_t)MB, D->cvx0);
  }
  hipLaunchKernelGGL(k_big_normalise, dim3(1024), dim3(256), 0, st, *D, (size_t)0, (size_t)D->N * D->horizon, D->cvx1);
  if (ws->nt) {   // the epoch's layer-0 inputs in the element type of the run, both orientations (sdx_gemm_nt.h): once per epoch
    const int R = D->N * D->horizon;
    StageArgs a[3] = {{D->mb_obs, D->obs_dim, R, D->obs_dim, ws->kp[0][0], ws->xn[0], ws->kp[0][0], ws->xt[0], ws->Rp},
                      {D->cvx0, D->state_dim, R, D->state_dim, ws->kp[2][0], ws->xn[1], ws->kp[2][0], ws->xt[1], ws->Rp},
                      {D->cvx1, D->state_dim, R, D->state_dim, ws->kp[2][0], ws->xn[2], ws->kp[2][0], ws->xt[2], ws->Rp}};
    stage(*D, a, 3, st);
  }
}

// forward + losses + backward of minibatch `mb` (mini-epoch `me`) for all three networks; leaves the flat gradients in ac_g / cv_g,
// the KL word in ac_g[g_tail] and the control block advanced.  The caller follows with the explicit clip + Adam.
extern "C" void sdxpk_big_step(const SdxpDev* Dp, const SdxpBigWs* ws, int mb, int me, hipStream_t st) {
  const SdxpDev& D = *Dp;
  const int MB = ws->MB, A = D.act_dim, U2 = D.units[2];
  const size_t r0 = (size_t)mb * MB;
  const int S = ws->nsplit, rchunk = (MB + S - 1) / S;
  const float* P[3] = {D.ac, D.ac, D.cv};
  float* G[3] = {D.ac_g, D.ac_g, D.cv_g};
  const int in0[3] = {D.obs_dim, D.obs_dim, D.state_dim};
  const float* X0[3] = {D.mb_obs + r0 * D.obs_dim, D.mb_obs + r0 * D.obs_dim, (me == 0 ? D.cvx0 : D.cvx1) + r0 * D.state_dim};
  auto woff = [&](int net, int l) { return net == 0 ? D.off.a_w[l] : (net == 1 ? D.off.c_w[l] : D.coff.w[l]); };
  auto boff = [&](int net, int l) { return net == 0 ? D.off.a_b[l] : (net == 1 ? D.off.c_b[l] : D.coff.b[l]); };
  const size_t region = ws->part_region;                                  // floats of split partials per network
  const size_t ES = D.bf16 ? 2 : 4;
  const int xsel[3] = {0, 0, me == 0 ? 1 : 2};                            // which staged dataset input a network reads
  auto Kof = [&](int net, int l) { return l == 0 ? in0[net] : D.units[l - 1]; };
  // ---- forward: layer l of the three networks in one launch
  if (ws->nt) {
    {   // this step's weights in the element type, [out][in padded] and (layers 1, 2) transposed [in][out]
      StageArgs a[9];
      for (int net = 0; net < 3; ++net)
        for (int l = 0; l < 3; ++l)
          a[net * 3 + l] = {P[net] + woff(net, l), Kof(net, l), D.units[l], Kof(net, l), ws->kp[net][l], ws->wn[net][l], ws->kp[net][l],
                            ws->wt[net][l], D.units[l]};
      stage(D, a, 9, st);
    }
    for (int l = 0; l < 3; ++l) {
      NtArgs g[3];
      for (int net = 0; net < 3; ++net) {
        const int kp = ws->kp[net][l];
        const void* A = l == 0 ? (const void*)((const char*)ws->xn[xsel[net]] + r0 * kp * ES) : (const void*)ws->hn[net][l - 1];
        // bf16 runs keep the outputs of layers 0 and 1 in bf16 only (both orientations): the next layer, the weight gradient and the
        // ELU' of the backward pass read those; the last trunk layer stays fp32 for the fp32 heads
        g[net] = {A, l == 0 ? kp : D.units[l - 1], ws->wn[net][l], kp, MB, D.units[l], kp, kp, (D.bf16 && l < 2) ? nullptr : ws->h[net][l], D.units[l], 0,
                  (D.bf16 && l < 2) ? ws->hn[net][l] : nullptr, D.units[l], (l < 2 && !ws->tt) ? ws->ht[net][l] : nullptr, ws->MBp,
                  P[net] + boff(net, l), nullptr, 0, nullptr, 0, nullptr};
      }
      if (D.bf16) gemm_nt<1, EPI_FWD>(g, 3, 1, st); else gemm_nt<0, EPI_FWD>(g, 3, 1, st);
    }
  } else
  for (int l = 0; l < 3; ++l) {
    GemmArgs g[3];
    for (int net = 0; net < 3; ++net) {
      const int in = l == 0 ? in0[net] : D.units[l - 1];
      const float* X = l == 0 ? X0[net] : ws->h[net][l - 1];
      g[net] = {X, in, P[net] + woff(net, l), in, ws->h[net][l], D.units[l], 0, MB, D.units[l], in, in, P[net] + boff(net, l), nullptr, 0, nullptr};
    }
    gemm<0, 0, 1>(g, 3, 1, st, D.bf16 != 0);
  }
  if (heads_fused_ok(D)) {
    static bool attr = false;
    if (!attr) { attr = true; (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_big_heads), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(HeadsLds)); }
    const int nrb = heads_nrb(MB), nb = heads_blocks(MB);
    hipLaunchKernelGGL(k_big_heads, dim3(nb), dim3(512), sizeof(HeadsLds), st, D, r0, MB, nrb, ws->h[0][2], ws->h[1][2], ws->h[2][2], ws->dy[0][2], ws->dy[1][2],
                       ws->dy[2][2], ws->part);
    hipLaunchKernelGGL(k_big_heads_reduce, dim3((HPZ + 31) / 32), dim3(256), 0, st, D, nb, MB, ws->part);
  } else {
  {  // heads
    GemmArgs g = {ws->h[0][2], U2, D.ac + D.off.mu_w, U2, ws->mu, 24, 0, MB, A, U2, U2, D.ac + D.off.mu_b, nullptr, 0, nullptr};
    gemm<0, 0, 2>(&g, 1, 1, st);
    hipLaunchKernelGGL(k_rowdot, dim3((MB + 3) / 4), dim3(256), 0, st, ws->h[1][2], U2, MB, D.ac + D.off.v_w, D.ac + D.off.v_b, ws->v);
    hipLaunchKernelGGL(k_rowdot, dim3((MB + 3) / 4), dim3(256), 0, st, ws->h[2][2], U2, MB, D.cv + D.coff.v_w, D.cv + D.coff.v_b, ws->v + MB);
  }
  const int hb = (MB + 255) / 256;
  hipLaunchKernelGGL(k_big_head, dim3(hb), dim3(256), 0, st, D, r0, MB, ws->mu, ws->v, ws->v + MB, ws->dmu, ws->dv, ws->part);
  hipLaunchKernelGGL(k_big_fin, dim3(1), dim3(64), 0, st, D, hb, MB, ws->part);
  // ---- head backward: data gradients into dy[net][2], weight gradients into the flat buffers
  {
    GemmArgs g = {ws->dmu, 24, D.ac + D.off.mu_w, U2, ws->dy[0][2], U2, 0, MB, U2, A, A, nullptr, ws->h[0][2], U2, nullptr};
    gemm<0, 1, 3>(&g, 1, 1, st);                                         // dY2 = (dmu Wmu) * ELU'(h3)
    hipLaunchKernelGGL(k_vhead_back, dim3(512), dim3(256), 0, st, ws->dv, D.ac + D.off.v_w, ws->h[1][2], U2, MB, ws->dy[1][2]);
    hipLaunchKernelGGL(k_vhead_back, dim3(512), dim3(256), 0, st, ws->dv + MB, D.cv + D.coff.v_w, ws->h[2][2], U2, MB, ws->dy[2][2]);
    // mu head: G[A][U2] = dmu^T h3, bias = row sums of dmu^T; contiguous [mu_w | mu_b] in the flat layout
    // (a 23-row product: its only parallelism is the reduction over the minibatch rows, so it is split 256 rows at a time - 32 .. 128
    // splits instead of the trunk's 8 .. 16: 94 us -> about 15 at 32 768 rows - and the partials are summed by one wave per output)
    const size_t pz = (size_t)A * U2 + A;
    int Sh = MB / 256;
    Sh = Sh < S ? S : (Sh > 128 ? 128 : Sh);
    GemmArgs gw = {ws->dmu, 24, ws->h[0][2], U2, ws->part, U2, pz, A, U2, MB, (MB + Sh - 1) / Sh, nullptr, nullptr, 0, ws->part + (size_t)A * U2};
    gemm<1, 1, 4>(&gw, 1, Sh, st);
    // value heads: [v_w | v_b] contiguous; 64-row splits
    const int VS = (MB + 63) / 64;
    float* vp0 = ws->part + (size_t)Sh * pz;
    float* vp1 = vp0 + (size_t)VS * (U2 + 1);
    hipLaunchKernelGGL(k_vhead_wgrad, dim3(VS), dim3(256), 0, st, ws->dv, ws->h[1][2], U2, MB, 64, vp0, (size_t)U2 + 1);
    hipLaunchKernelGGL(k_vhead_wgrad, dim3(VS), dim3(256), 0, st, ws->dv + MB, ws->h[2][2], U2, MB, 64, vp1, (size_t)U2 + 1);
    hipLaunchKernelGGL(k_wave_reduce, dim3(((int)pz + 3) / 4), dim3(256), 0, st, ws->part, pz, Sh, (int)pz, D.ac_g + D.off.mu_w);
    hipLaunchKernelGGL(k_vhead_reduce, dim3((2 * (U2 + 1) + 3) / 4), dim3(256), 0, st, vp0, vp1, (size_t)U2 + 1, VS, U2 + 1, D.ac_g + D.off.v_w,
                       D.cv_g + D.coff.v_w);
  }
  }
  // ---- trunk backward, layer by layer for the three networks at once
  if (ws->nt) {
    if (!ws->tt) {   // the head kernels left dLoss/d(pre-activation of trunk layer 2) in fp32: element-type copy + transpose
      StageArgs a[3];
      for (int net = 0; net < 3; ++net)
        a[net] = {ws->dy[net][2], U2, MB, U2, U2, D.bf16 ? ws->dyn[net][2] : nullptr, U2, ws->dyt[net][2], ws->MBp};
      stage(D, a, 3, st);
    }
    for (int l = 2; l >= 0; --l) {
      const int Nl = D.units[l];
      // splits of this layer's weight-gradient product over the minibatch rows: as few as still give the 512 workgroup slots about a
      // round and a half of 128 x 64-or-wider tiles (every split costs a pass over [W | b] in the partial reduction), at least 4 chunks each
      int Sl;
      {
        int Kmax = 0;
        for (int net = 0; net < 3; ++net) Kmax = Kof(net, l) > Kmax ? Kof(net, l) : Kmax;
        const int tiles = 3 * ((Nl + 127) / 128) * ((Kmax + 127) / 128);
        Sl = (768 + tiles - 1) / tiles;
        const int smax = ws->MBp / (4 * ws->KC) > 0 ? ws->MBp / (4 * ws->KC) : 1;
        if (Sl > smax) Sl = smax;
        if (Sl > S) Sl = S;
        if (Sl < 1) Sl = 1;
      }
      const int kc = (((ws->MBp + Sl - 1) / Sl) + ws->KC - 1) / ws->KC * ws->KC;        // reduction rows per split, a whole number of chunks
      NtArgs gw[3];
      ReduceBatch rb;
      rb.S = Sl;
      for (int net = 0; net < 3; ++net) {
        const int Kl = Kof(net, l);
        const size_t pz = (size_t)Nl * Kl + Nl;                           // [W_l | b_l] contiguous in the flat layout
        float* part = ws->part + (size_t)net * region;
        if (ws->tt) {   // fp32: dY_l [MB][N_l] and the layer input [MB][K_l] as the other products leave them (k_gemm_tt)
          const int kp = ws->kp[net][0];
          const void* X = l == 0 ? (const void*)((const char*)ws->xn[xsel[net]] + r0 * kp * ES) : (const void*)ws->h[net][l - 1];
          gw[net] = {ws->dy[net][l], Nl, X, l == 0 ? kp : D.units[l - 1], Nl, Kl, MB, kc, part, Kl, pz, nullptr, 0, nullptr, 0,
                     nullptr, nullptr, 0, nullptr, 0, part + (size_t)Nl * Kl};
        } else {
          const void* Xt = l == 0 ? (const void*)((const char*)ws->xt[xsel[net]] + r0 * ES) : (const void*)ws->ht[net][l - 1];
          gw[net] = {ws->dyt[net][l], ws->MBp, Xt, l == 0 ? ws->Rp : ws->MBp, Nl, Kl, ws->MBp, kc, part, Kl, pz, nullptr, 0, nullptr, 0,
                     nullptr, nullptr, 0, nullptr, 0, part + (size_t)Nl * Kl};
        }
        rb.part[net] = part; rb.pz[net] = pz; rb.n[net] = pz; rb.out[net] = G[net] + woff(net, l);
      }
      if (ws->tt) gemm_tt(gw, 3, Sl, ws->zeros, st);
      else if (D.bf16) gemm_nt<1, EPI_TN>(gw, 3, Sl, st); else gemm_nt<0, EPI_TN>(gw, 3, Sl, st);   // G_l = dY_l^T X_l, b_l = row sums of dY_l^T
      hipLaunchKernelGGL(k_reduce_parts3, dim3(256, 3), dim3(256), 0, st, rb);
      if (l > 0) {
        NtArgs gx[3];
        const int Kl = D.units[l - 1];
        for (int net = 0; net < 3; ++net) {
          if (ws->tt)   // one image: dY_{l-1} [MB][K_l] (layer 0 included: its weight gradient reads it), ELU' from the layer output itself
            gx[net] = {ws->dy[net][l], Nl, ws->wt[net][l], Nl, MB, Kl, Nl, Nl, nullptr, 0, 0, ws->dy[net][l - 1], Kl,
                       nullptr, 0, nullptr, ws->h[net][l - 1], Kl, nullptr, 0, nullptr};
          else
            gx[net] = {ws->dyn[net][l], Nl, ws->wt[net][l], Nl, MB, Kl, Nl, Nl, nullptr, 0, 0, l - 1 >= 1 ? ws->dyn[net][l - 1] : nullptr, Kl,
                       ws->dyt[net][l - 1], ws->MBp, nullptr, ws->hn[net][l - 1], Kl, ws->ht[net][l - 1], ws->MBp, nullptr};
        }
        if (D.bf16) gemm_nt<1, EPI_NN>(gx, 3, 1, st); else gemm_nt<0, EPI_NN>(gx, 3, 1, st);   // dY_{l-1} = (dY_l W_l) * ELU'(H_{l-1})
      }
    }
    return;
  }
  for (int l = 2; l >= 0; --l) {
    const int Nl = D.units[l];
    GemmArgs gw[3];
    ReduceBatch rb;
    rb.S = S;
    for (int net = 0; net < 3; ++net) {
      const int Kl = l == 0 ? in0[net] : D.units[l - 1];
      const float* Xl = l == 0 ? X0[net] : ws->h[net][l - 1];
      const size_t pz = (size_t)Nl * Kl + Nl;                             // [W_l | b_l] contiguous in the flat layout
      float* part = ws->part + (size_t)net * region;
      gw[net] = {ws->dy[net][l], Nl, Xl, Kl, part, Kl, pz, Nl, Kl, MB, rchunk, nullptr, nullptr, 0, part + (size_t)Nl * Kl};
      rb.part[net] = part; rb.pz[net] = pz; rb.n[net] = pz; rb.out[net] = G[net] + woff(net, l);
    }
    gemm<1, 1, 4>(gw, 3, S, st, D.bf16 != 0);                             // G_l = dY_l^T X_l, b_l = row sums of dY_l^T (fused)
    hipLaunchKernelGGL(k_reduce_parts3, dim3(256, 3), dim3(256), 0, st, rb);
    if (l > 0) {
      GemmArgs gx[3];
      const int Kl = D.units[l - 1];
      for (int net = 0; net < 3; ++net)
        gx[net] = {ws->dy[net][l], Nl, P[net] + woff(net, l), Kl, ws->dy[net][l - 1], Kl, 0, MB, Kl, Nl, Nl, nullptr, ws->h[net][l - 1], Kl, nullptr};
      gemm<0, 1, 3>(gx, 3, 1, st, D.bf16 != 0);                           // dY_{l-1} = (dY_l W_l) * ELU'(H_{l-1})
    }
  }
}

// ---- timing / test hook (not part of include/seqdex.h): launch one batched NT product on caller-owned operands (tools/time_gemm_nt.py)
extern "C" int sdxpk_gemm_nt_launch(int bf, int epi, const NtArgs* gs, int count, int splits, hipStream_t st) {
  if (count < 1 || count > 3 || splits < 1) return -1;
  if (bf) {
    if (epi == EPI_FWD) gemm_nt<1, EPI_FWD>(gs, count, splits, st);
    else if (epi == EPI_NN) gemm_nt<1, EPI_NN>(gs, count, splits, st);
    else if (epi == EPI_TN) gemm_nt<1, EPI_TN>(gs, count, splits, st);
    else return -1;
  } else {
    if (epi == EPI_FWD) gemm_nt<0, EPI_FWD>(gs, count, splits, st);
    else if (epi == EPI_NN) gemm_nt<0, EPI_NN>(gs, count, splits, st);
    else if (epi == EPI_TN) gemm_nt<0, EPI_TN>(gs, count, splits, st);
    else return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
