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

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float elu(float x) { return x > 0.0f ? x : expm1f(x); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------ rollout GEMM
// Y[M,N] = act(X[M,K] W[N,K]^T + b[N]); act = ELU when elu_flag.  X may be normalised on the fly with (mean, rstd)
// (central-value running mean/std, clamp +-5; App. C of SURVEY.md).  K % 4 == 0.
#define GT 64
#define GK 16
__global__ __launch_bounds__(256) void k_linear_mfma(const float* __restrict__ X, const float* __restrict__ W,
                                                     const float* __restrict__ b, float* __restrict__ Y, int M, int N,
                                                     int K, int elu_flag, const double* __restrict__ nmean,
                                                     const double* __restrict__ nvar) {
  __shared__ float Xs[GT][GK + 1];
  __shared__ float Ws[GT][GK + 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  const int lr = tid >> 2, lk = (tid & 3) * 4;  // this thread stages row lr, k-offset lk of both tiles
  for (int k0 = 0; k0 < K; k0 += GK) {
    float4 xv = make_float4(0, 0, 0, 0), wv = make_float4(0, 0, 0, 0);
    if (m0 + lr < M && k0 + lk < K) {
      xv = *reinterpret_cast<const float4*>(X + (size_t)(m0 + lr) * K + k0 + lk);
      if (nmean) {
        float* xp = reinterpret_cast<float*>(&xv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float mu = (float)nmean[k0 + lk + j], var = (float)nvar[k0 + lk + j];
          xp[j] = clampf((xp[j] - mu) / sqrtf(var + 1e-5f), -5.0f, 5.0f);
        }
      }
    }
    if (n0 + lr < N && k0 + lk < K) wv = *reinterpret_cast<const float4*>(W + (size_t)(n0 + lr) * K + k0 + lk);
    __syncthreads();
    Xs[lr][lk] = xv.x; Xs[lr][lk + 1] = xv.y; Xs[lr][lk + 2] = xv.z; Xs[lr][lk + 3] = xv.w;
    Ws[lr][lk] = wv.x; Ws[lr][lk + 1] = wv.y; Ws[lr][lk + 2] = wv.z; Ws[lr][lk + 3] = wv.w;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; kk += 2) {
      const float a = Xs[wm + (lane & 31)][kk + (lane >> 5)];
      const float bb = Ws[wn + (lane & 31)][kk + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
    }
  }
  const int col = n0 + wn + (lane & 31);
  if (col < N) {
    const float bias = b[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < M) {
        float v = acc[r] + bias;
        Y[(size_t)row * N + col] = elu_flag ? elu(v) : v;
      }
    }
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
  const int A = D.act_dim, U = D.units[2];
  __shared__ float s_h[2][256];
  for (int i = lane; i < U; i += 64) {
    s_h[0][i] = D.h_a[2][(size_t)e * U + i];
    s_h[1][i] = D.h_v[2][(size_t)e * U + i];
  }
  __syncthreads();
  const size_t row = (size_t)e * D.horizon + t;
  float nlp_part = 0.0f;
  if (lane < A) {
    const float* w = D.ac + D.off.mu_w + (size_t)lane * U;
    float mu = D.ac[D.off.mu_b + lane];
    for (int k = 0; k < U; ++k) mu += w[k] * s_h[0][k];
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
  {
    const float* w = D.cv + D.coff.v_w;
    for (int k = lane; k < U; k += 64) vpart += w[k] * s_h[1][k];
  }
  const float v = wave_sum(vpart) + D.cv[D.coff.v_b];
  if (lane == 0) {
    D.mb_neglogp[row] = nlp;
    D.mb_values[row] = v;
    D.mb_dones[row] = (dones && dones[e] != 0) ? 1.0f : 0.0f;
  }
  for (int i = lane; i < D.obs_dim; i += 64) D.mb_obs[row * D.obs_dim + i] = obs[(size_t)e * D.obs_dim + i];
  for (int i = lane; i < D.state_dim; i += 64) D.mb_states[row * D.state_dim + i] = states[(size_t)e * D.state_dim + i];
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
// network ids: 0 actor trunk, 1 critic trunk, 2 central-value trunk.  Per net and layer l (0..2):
//   x[net][l][par][MB][K_l]   layer input of the minibatch with parity `par` (l=0: obs / normalised states)
//   dy[net][l][par][MB][N_l]  dLoss/d(pre-activation) of layer l
// Head layers (mu, value, cv value) live in the HEAD kernel.

// L-kernel: lazy Adam of the previous minibatch + forward of the current one for trunk layer `l` of all nets.
// grid = ceil(total_rows / ROWS_PER_BLOCK); block = 256 threads = 4 waves, wave per weight row.
template <int MB>
__global__ __launch_bounds__(256) void k_layer(SdxpDev D, int l) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const SdxpCtrl* ctl = D.ctrl;
  const int par = ctl->step & 1;                 // parity of the CURRENT minibatch
  const int Nl = D.units[l];
  // which net does this block serve?  rows of (actor, critic, cv) are laid out consecutively
  const int rows_per_block = 4 * D.rows_per_wave;
  const int blocks_per_net = (Nl + rows_per_block - 1) / rows_per_block;
  const int net = blockIdx.x / blocks_per_net, blk = blockIdx.x % blocks_per_net;
  const int K = (l == 0) ? (net == 2 ? D.state_dim : D.obs_dim) : D.units[l - 1];
  const bool pend = (net == 2 ? ctl->cv_pending : ctl->ac_pending) != 0;
  float* P = net == 2 ? D.cv : D.ac;
  float* Mo = net == 2 ? D.cv_m : D.ac_m;
  float* Vo = net == 2 ? D.cv_v : D.ac_v;
  const size_t woff = net == 0 ? D.off.a_w[l] : net == 1 ? D.off.c_w[l] : D.coff.w[l];
  const size_t boff = net == 0 ? D.off.a_b[l] : net == 1 ? D.off.c_b[l] : D.coff.b[l];
  const float lr = net == 2 ? ctl->cv_lr_applied : ctl->ac_lr_applied;
  const float gs = net == 2 ? ctl->cv_gscale : ctl->ac_gscale;          // grad clip scale of the pending step
  const float bc1 = net == 2 ? ctl->cv_bc1 : ctl->ac_bc1, bc2 = net == 2 ? ctl->cv_bc2 : ctl->ac_bc2;
  float* xc = sm;                 // [MB][K] current input
  float* xp = sm + MB * K;        // [MB][K] previous input (factor of the pending gradient)
  const float* gx_c = D.x[net][l] + (size_t)par * MB * K;
  const float* gx_p = D.x[net][l] + (size_t)(par ^ 1) * MB * K;
  for (int i = tid; i < MB * K; i += 256) { xc[i] = gx_c[i]; xp[i] = gx_p[i]; }
  __syncthreads();
  const float* dyp = D.dy[net][l] + (size_t)(par ^ 1) * MB * Nl;
  float* xn = D.x[net][l + 1] + (size_t)par * MB * Nl;    // output = next layer's input
  for (int rr = 0; rr < D.rows_per_wave; ++rr) {
    const int n = blk * rows_per_block + wave * D.rows_per_wave + rr;
    if (n >= Nl) break;
    float dyn[MB];
#pragma unroll
    for (int s = 0; s < MB; ++s) dyn[s] = pend ? dyp[s * Nl + n] * gs : 0.0f;
    float acc[MB];
#pragma unroll
    for (int s = 0; s < MB; ++s) acc[s] = 0.0f;
    float* wrow = P + woff + (size_t)n * K;
    float* mrow = Mo + woff + (size_t)n * K;
    float* vrow = Vo + woff + (size_t)n * K;
    for (int k = lane; k < K; k += 64) {
      float w = wrow[k];
      if (pend) {
        float g = 0.0f;
#pragma unroll
        for (int s = 0; s < MB; ++s) g += dyn[s] * xp[s * K + k];
        const float m = 0.9f * mrow[k] + 0.1f * g;
        const float v = 0.999f * vrow[k] + 0.001f * g * g;
        mrow[k] = m; vrow[k] = v;
        w -= (lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + 1e-8f);
        wrow[k] = w;
      }
#pragma unroll
      for (int s = 0; s < MB; ++s) acc[s] += w * xc[s * K + k];
    }
    float bias = P[boff + n];
    if (pend && lane == 0) {
      float g = 0.0f;
#pragma unroll
      for (int s = 0; s < MB; ++s) g += dyn[s];
      const float m = 0.9f * Mo[boff + n] + 0.1f * g;
      const float v = 0.999f * Vo[boff + n] + 0.001f * g * g;
      Mo[boff + n] = m; Vo[boff + n] = v;
      bias -= (lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + 1e-8f);
      P[boff + n] = bias;
    }
    bias = __shfl(bias, 0, 64);
#pragma unroll
    for (int s = 0; s < MB; ++s) {
      const float y = wave_sum(acc[s]) + bias;
      if (lane == 0) xn[s * Nl + n] = elu(y);
    }
  }
}

// HEAD kernel (one block): lazy Adam of the head parameters, head forward, PPO losses (R7), gradients of the
// heads' pre-activations, backward through the heads into dY of trunk layer 2, KL + statistics, mu/sigma
// write-back (dataset.update_mu_sigma, RC:1358).  flush != 0: only the lazy Adam part (end of an epoch).
template <int MB>
__global__ __launch_bounds__(256) void k_head(SdxpDev D, int flush) {
  __shared__ float s_h[3][MB][256];
  __shared__ float s_hp[3][MB][256];
  __shared__ float s_mu[MB][32], s_dmu[MB][32], s_dmu_p[MB][32], s_z[MB][32];
  __shared__ float s_v[2][MB], s_dv[2][MB], s_dv_p[2][MB], s_gnlp[MB];
  __shared__ float s_stat[MB][8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  SdxpCtrl* ctl = D.ctrl;
  const int par = ctl->step & 1, U = D.units[2], A = D.act_dim;
  const int mb = ctl->mb_index;
  const bool apend = ctl->ac_pending != 0, cpend = ctl->cv_pending != 0;
  for (int i = tid; i < 3 * MB * U; i += 256) {
    const int net = i / (MB * U), r = i % (MB * U);
    s_h[net][r / U][r % U] = D.x[net][3][(size_t)par * MB * U + r];
    s_hp[net][r / U][r % U] = D.x[net][3][(size_t)(par ^ 1) * MB * U + r];
  }
  if (tid < MB * 32) {
    const int s = tid / 32, a = tid % 32;
    s_dmu_p[s][a] = (a < A) ? D.dhead[(size_t)(par ^ 1) * MB * 34 + s * 34 + a] : 0.0f;
  }
  if (tid < 2 * MB) s_dv_p[tid / MB][tid % MB] = D.dhead[(size_t)(par ^ 1) * MB * 34 + (tid % MB) * 34 + 32 + tid / MB];
  __syncthreads();
  // ---- lazy Adam + forward of the head rows: rows 0..A-1 = mu, row A = critic value, row A+1 = central value
  for (int row = wave; row < A + 2; row += 4) {
    const int net = row < A ? 0 : (row == A ? 1 : 2);
    float* P = net == 2 ? D.cv : D.ac;
    float* Mo = net == 2 ? D.cv_m : D.ac_m;
    float* Vo = net == 2 ? D.cv_v : D.ac_v;
    const size_t woff = row < A ? D.off.mu_w + (size_t)row * U : (row == A ? D.off.v_w : D.coff.v_w);
    const size_t boff = row < A ? D.off.mu_b + row : (row == A ? D.off.v_b : D.coff.v_b);
    const bool pend = net == 2 ? cpend : apend;
    const float lr = net == 2 ? ctl->cv_lr_applied : ctl->ac_lr_applied;
    const float gs = net == 2 ? ctl->cv_gscale : ctl->ac_gscale;
    const float bc1 = net == 2 ? ctl->cv_bc1 : ctl->ac_bc1, bc2 = net == 2 ? ctl->cv_bc2 : ctl->ac_bc2;
    float dyn[MB], acc[MB];
#pragma unroll
    for (int s = 0; s < MB; ++s) {
      dyn[s] = pend ? gs * (row < A ? s_dmu_p[s][row] : s_dv_p[row - A][s]) : 0.0f;
      acc[s] = 0.0f;
    }
    for (int k = lane; k < U; k += 64) {
      float w = P[woff + k];
      if (pend) {
        float g = 0.0f;
#pragma unroll
        for (int s = 0; s < MB; ++s) g += dyn[s] * s_hp[net][s][k];
        const float m = 0.9f * Mo[woff + k] + 0.1f * g;
        const float v = 0.999f * Vo[woff + k] + 0.001f * g * g;
        Mo[woff + k] = m; Vo[woff + k] = v;
        w -= (lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + 1e-8f);
        P[woff + k] = w;
      }
#pragma unroll
      for (int s = 0; s < MB; ++s) acc[s] += w * s_h[net][s][k];
    }
    float bias = P[boff];
    if (pend && lane == 0) {
      float g = 0.0f;
#pragma unroll
      for (int s = 0; s < MB; ++s) g += dyn[s];
      const float m = 0.9f * Mo[boff] + 0.1f * g;
      const float v = 0.999f * Vo[boff] + 0.001f * g * g;
      Mo[boff] = m; Vo[boff] = v;
      bias -= (lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + 1e-8f);
      P[boff] = bias;
    }
    bias = __shfl(bias, 0, 64);
#pragma unroll
    for (int s = 0; s < MB; ++s) {
      const float y = wave_sum(acc[s]) + bias;
      if (lane == 0) { if (row < A) s_mu[s][row] = y; else s_v[row - A][s] = y; }
    }
  }
  if (tid < A && apend) {   // logstd parameter (fixed_sigma: a free Parameter, YG:18-21)
    const float g = D.dlogstd[(size_t)(par ^ 1) * 32 + tid] * ctl->ac_gscale;
    const size_t o = D.off.logstd + tid;
    const float m = 0.9f * D.ac_m[o] + 0.1f * g;
    const float v = 0.999f * D.ac_v[o] + 0.001f * g * g;
    D.ac_m[o] = m; D.ac_v[o] = v;
    D.ac[o] -= (ctl->ac_lr_applied / ctl->ac_bc1) * m / (sqrtf(v) / sqrtf(ctl->ac_bc2) + 1e-8f);
  }
  __syncthreads();
  if (flush) {
    if (tid == 0) { ctl->ac_pending = 0; ctl->cv_pending = 0; }
    return;
  }
  const size_t r0 = (size_t)mb * MB;   // first dataset row of this minibatch: contiguous, unshuffled (App. C)
  const float invM = 1.0f / (float)MB;
  // ---- per (sample, action) terms
  if (tid < MB * 32) {
    const int s = tid / 32, a = tid % 32;
    float z = 0.0f;
    if (a < A) {
      const float sg = expf(D.ac[D.off.logstd + a]);
      z = (D.mb_actions[(r0 + s) * A + a] - s_mu[s][a]) / sg;
    }
    s_z[s][a] = z;
  }
  __syncthreads();
  // ---- per-sample scalars: one thread per sample (A = 23 terms each)
  if (tid < MB) {
    const int s = tid;
    float nlp = 0.5f * 1.8378770664093453f * (float)A, kl = 0.0f, bl = 0.0f, ent = 0.0f;
    for (int a = 0; a < A; ++a) {
      const float ls = D.ac[D.off.logstd + a], sg = expf(ls), mu = s_mu[s][a];
      nlp += 0.5f * s_z[s][a] * s_z[s][a] + ls;                                                  // RC:2114-2126
      const float omu = D.mb_mus[(r0 + s) * A + a], osg = D.mb_sigmas[(r0 + s) * A + a];
      kl += logf(osg / sg + 1e-5f) + (sg * sg + (omu - mu) * (omu - mu)) / (2.0f * (osg * osg + 1e-5f)) - 0.5f;
      const float hi = fmaxf(mu - 1.1f, 0.0f), lo = fminf(mu + 1.1f, 0.0f);
      bl += hi * hi + lo * lo;
      ent += 0.5f + 0.5f * 1.8378770664093453f + ls;
    }
    const float adv = D.adv[r0 + s];
    const float ratio = expf(D.mb_neglogp[r0 + s] - nlp);
    const float L1 = -adv * ratio, L2 = -adv * clampf(ratio, 1.0f - D.e_clip, 1.0f + D.e_clip);   // RC:1813
    // d max(L1,L2)/d nlp: L1' = adv*ratio; L2' = adv*ratio inside the clip range, else 0; ties share the same value
    const bool inr = ratio >= 1.0f - D.e_clip && ratio <= 1.0f + D.e_clip;
    s_gnlp[s] = (L1 > L2 || inr) ? adv * ratio : 0.0f;
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
        d = (c1 > c2 || inv) ? 2.0f * (v - R) : 0.0f;   // outside the clip range v_clipped is constant
        if (c2 > c1 && inv) d = 2.0f * (vc - R);
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
      const float ls = D.ac[D.off.logstd + a], sg = expf(ls), mu = s_mu[s][a];
      const float hi = fmaxf(mu - 1.1f, 0.0f), lo = fminf(mu + 1.1f, 0.0f);
      dmu = s_gnlp[s] * (-(s_z[s][a] / sg)) * invM + D.bounds_coef * (2.0f * hi + 2.0f * lo) * invM;
      D.mb_mus[(r0 + s) * A + a] = mu;                                                            // RC:1358
      D.mb_sigmas[(r0 + s) * A + a] = sg;
    }
    s_dmu[s][a] = dmu;
    D.dhead[(size_t)par * MB * 34 + s * 34 + a] = dmu;
    if (a < 2) D.dhead[(size_t)par * MB * 34 + s * 34 + 32 + a] = s_dv[a][s];
  }
  if (tid < A) {
    float g = 0.0f;
    for (int s = 0; s < MB; ++s) g += s_gnlp[s] * (1.0f - s_z[s][tid] * s_z[s][tid]) * invM;
    D.dlogstd[(size_t)par * 32 + tid] = g;
  }
  if (tid == 0) {
    for (int j = 1; j <= 6; ++j) {
      float t = 0.0f;
      for (int s = 0; s < MB; ++s) t += s_stat[s][j];
      ctl->acc[j] = t;
    }
  }
  __syncthreads();
  // ---- backward through the heads: dY2[net][s][k] = (sum_rows dhead * W_head[row][k]) * elu'(h[net][s][k])
  for (int i = tid; i < 3 * MB * U; i += 256) {
    const int net = i / (MB * U), s = (i / U) % MB, k = i % U;
    float d = 0.0f;
    if (net == 0) { for (int a = 0; a < A; ++a) d += s_dmu[s][a] * D.ac[D.off.mu_w + (size_t)a * U + k]; }
    else if (net == 1) d = s_dv[0][s] * D.ac[D.off.v_w + k];
    else d = s_dv[1][s] * D.cv[D.coff.v_w + k];
    const float h = s_h[net][s][k];
    D.dy[net][2][(size_t)par * MB * U + s * U + k] = d * (h > 0.0f ? 1.0f : h + 1.0f);
  }
}

// B-kernel: backward of trunk layer l+1 into layer l of all nets:
//   dY_l[s][k] = (sum_n dY_{l+1}[s][n] W_{l+1}[n][k]) * elu'(h_l[s][k]),  thread per k, block per (net, k-chunk, n-split)
// partial sums over the n-splits are accumulated with atomics into a zeroed buffer; elu' is applied by the consumer.
template <int MB>
__global__ __launch_bounds__(256) void k_back(SdxpDev D, int l) {   // l = layer whose dY is produced (1 or 0)
  const SdxpCtrl* ctl = D.ctrl;
  const int par = ctl->step & 1;
  const int Nn = D.units[l + 1], K = D.units[l];
  const int kchunks = (K + 255) / 256, splits = D.bsplit;
  const int per_net = kchunks * splits;
  const int net = blockIdx.x / per_net, rem = blockIdx.x % per_net, kc = rem / splits, sp = rem % splits;
  const int k = kc * 256 + threadIdx.x;
  __shared__ float s_dy[MB][128];
  const int n_per = (Nn + splits - 1) / splits, n0 = sp * n_per, n1 = min(Nn, n0 + n_per);
  const float* P = net == 2 ? D.cv : D.ac;
  const size_t woff = net == 0 ? D.off.a_w[l + 1] : net == 1 ? D.off.c_w[l + 1] : D.coff.w[l + 1];
  const float* dyn = D.dy[net][l + 1] + (size_t)par * MB * Nn;
  float acc[MB];
#pragma unroll
  for (int s = 0; s < MB; ++s) acc[s] = 0.0f;
  for (int nb = n0; nb < n1; nb += 128) {
    __syncthreads();
    for (int i = threadIdx.x; i < MB * 128; i += 256) {
      const int s = i / 128, n = nb + (i % 128);
      s_dy[s][i % 128] = n < n1 ? dyn[s * Nn + n] : 0.0f;
    }
    __syncthreads();
    if (k < K) {
      const int cnt = min(128, n1 - nb);
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

// elu' + move: dY_l = dxacc_l * elu'(h_l); also zero dxacc of the other parity for the next step
template <int MB>
__global__ void k_back_fin(SdxpDev D, int l) {
  const SdxpCtrl* ctl = D.ctrl;
  const int par = ctl->step & 1, K = D.units[l];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * MB * K) return;
  const int net = i / (MB * K), r = i % (MB * K);
  const float h = D.x[net][l + 1][(size_t)par * MB * K + r];
  float* acc = D.dxacc[net][l] + (size_t)par * MB * K;
  D.dy[net][l][(size_t)par * MB * K + r] = acc[r] * (h > 0.0f ? 1.0f : h + 1.0f);
  acc[r] = 0.0f;
}

// CTRL kernel (one block): gradient norms from the Gram matrices of the rank-MB factors, clip scale, Adam bias
// corrections, legacy adaptive LR (PS:306-312), statistics, central-value running mean/std of the NEXT minibatch and
// staging of the next minibatch's layer-0 inputs.
template <int MB>
__global__ __launch_bounds__(256) void k_ctrl(SdxpDev D, int advance) {
  __shared__ float s_g[3][MB][MB];   // accumulated sum_{layers} (dY_s.dY_s')(X_s.X_s') per optimiser group (ac uses [0]+[1])
  __shared__ float s_b[3];
  __shared__ float s_tmp[2][MB][MB];
  SdxpCtrl* ctl = D.ctrl;
  const int tid = threadIdx.x, par = ctl->step & 1, A = D.act_dim, U = D.units[2];
  if (tid < 3 * MB * MB) (&s_g[0][0][0])[tid] = 0.0f;
  if (tid < 3) s_b[tid] = 0.0f;
  __syncthreads();
  if (advance & 1) {
    // ---- gradient norm of the minibatch that was just back-propagated (parity `par`)
    for (int net = 0; net < 3; ++net) {
      for (int l = 0; l < 3; ++l) {
        const int Nl = D.units[l];
        const int K = (l == 0) ? (net == 2 ? D.state_dim : D.obs_dim) : D.units[l - 1];
        const float* dy = D.dy[net][l] + (size_t)par * MB * Nl;
        const float* x = D.x[net][l] + (size_t)par * MB * K;
        // Gram matrices with all 256 threads: each thread strides over the vector, then block reduce via atomics
        float gd[MB][MB], gx[MB][MB], bs = 0.0f;
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b = 0; b < MB; ++b) { gd[a][b] = 0.0f; gx[a][b] = 0.0f; }
        for (int n = tid; n < Nl; n += 256) {
          float v[MB], sb = 0.0f;
#pragma unroll
          for (int a = 0; a < MB; ++a) { v[a] = dy[a * Nl + n]; sb += v[a]; }
          bs += sb * sb;
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) gd[a][b] += v[a] * v[b];
        }
        for (int k = tid; k < K; k += 256) {
          float v[MB];
#pragma unroll
          for (int a = 0; a < MB; ++a) v[a] = x[a * K + k];
#pragma unroll
          for (int a = 0; a < MB; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) gx[a][b] += v[a] * v[b];
        }
        if (tid < 2 * MB * MB) (&s_tmp[0][0][0])[tid] = 0.0f;
        __syncthreads();
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b) {
            const float r1 = wave_sum(gd[a][b]), r2 = wave_sum(gx[a][b]);
            if ((tid & 63) == 0) { atomicAdd(&s_tmp[0][a][b], r1); atomicAdd(&s_tmp[1][a][b], r2); }
          }
        const float rb = wave_sum(bs);
        if ((tid & 63) == 0) atomicAdd(&s_b[net], rb);
        __syncthreads();
        if (tid < MB * MB) {
          const int a = tid / MB, b = tid % MB;
          const float d = a >= b ? s_tmp[0][a][b] : s_tmp[0][b][a];
          const float xx = a >= b ? s_tmp[1][a][b] : s_tmp[1][b][a];
          s_g[net][a][b] += d * xx;
        }
        __syncthreads();
      }
    }
    // heads: factors are dhead (MB x 34) and h = x[net][3] (MB x U); h Gram in parallel, the rest is tiny
    __shared__ float s_hh[3][MB][MB];
    if (tid < 3 * MB * MB) (&s_hh[0][0][0])[tid] = 0.0f;
    __syncthreads();
    for (int net = 0; net < 3; ++net) {
      const float* h = D.x[net][3] + (size_t)par * MB * U;
      float gx[MB][MB];
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < MB; ++b) gx[a][b] = 0.0f;
      for (int k = tid; k < U; k += 256) {
        float v[MB];
#pragma unroll
        for (int a = 0; a < MB; ++a) v[a] = h[a * U + k];
#pragma unroll
        for (int a = 0; a < MB; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b) gx[a][b] += v[a] * v[b];
      }
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          const float r = wave_sum(gx[a][b]);
          if ((tid & 63) == 0) { atomicAdd(&s_hh[net][a][b], r); if (a != b) atomicAdd(&s_hh[net][b][a], r); }
        }
    }
    __syncthreads();
    if (tid == 0) {
      float n2[3] = {0, 0, 0};
      for (int net = 0; net < 3; ++net) {
        float s = 0.0f;
        for (int a = 0; a < MB; ++a) for (int b = 0; b < MB; ++b) s += s_g[net][a][b];
        n2[net] = s + s_b[net];
      }
      const float* dh = D.dhead + (size_t)par * MB * 34;
      for (int net = 0; net < 3; ++net) {
        for (int a = 0; a < MB; ++a)
          for (int b = 0; b < MB; ++b) {
            float dd = 0.0f;
            if (net == 0) { for (int j = 0; j < A; ++j) dd += dh[a * 34 + j] * dh[b * 34 + j]; }
            else dd = dh[a * 34 + 32 + (net - 1)] * dh[b * 34 + 32 + (net - 1)];
            n2[net] += dd * s_hh[net][a][b];
          }
        if (net == 0) { for (int j = 0; j < A; ++j) { float sb = 0; for (int a = 0; a < MB; ++a) sb += dh[a * 34 + j]; n2[0] += sb * sb; } }
        else { float sb = 0; for (int a = 0; a < MB; ++a) sb += dh[a * 34 + 32 + (net - 1)]; n2[net] += sb * sb; }
      }
      for (int j = 0; j < A; ++j) { const float g = D.dlogstd[(size_t)par * 32 + j]; n2[0] += g * g; }
      const float ac_norm = sqrtf(n2[0] + n2[1]), cv_norm = sqrtf(n2[2]);
      ctl->ac_gnorm = ac_norm; ctl->cv_gnorm = cv_norm;
      ctl->ac_gscale = D.truncate_grads ? fminf(1.0f, D.grad_norm / (ac_norm + 1e-6f)) : 1.0f;   // clip_grad_norm_
      ctl->cv_gscale = D.truncate_grads ? fminf(1.0f, D.grad_norm / (cv_norm + 1e-6f)) : 1.0f;
      // Adam step counters / bias corrections of the step that is now pending
      ctl->ac_t += 1; ctl->cv_t += 1;
      ctl->ac_bc1 = 1.0f - powf(0.9f, (float)ctl->ac_t); ctl->ac_bc2 = 1.0f - powf(0.999f, (float)ctl->ac_t);
      ctl->cv_bc1 = 1.0f - powf(0.9f, (float)ctl->cv_t); ctl->cv_bc2 = 1.0f - powf(0.999f, (float)ctl->cv_t);
      ctl->ac_lr_applied = ctl->ac_lr; ctl->cv_lr_applied = ctl->cv_lr;
      const bool explicit_mode = (advance & 8) != 0;   // multi-rank: Adam/LR happen in sdxp_apply after the all-reduce
      if (explicit_mode) { ctl->ac_t -= 1; ctl->cv_t -= 1; ctl->gn2_ac = 0.0f; ctl->gn2_cv = 0.0f; }
      ctl->ac_pending = explicit_mode ? 0 : 1; ctl->cv_pending = explicit_mode ? 0 : 1;
      // statistics (means over the minibatch) + legacy adaptive LR from this minibatch's KL (PS:306-312)
      const float invM = 1.0f / (float)MB;
      const float kl = ctl->acc[4] * invM;
      ctl->sum_a_loss += ctl->acc[1] * invM; ctl->sum_c_loss += ctl->acc[2] * invM; ctl->sum_b_loss += ctl->acc[3] * invM;
      ctl->sum_kl += kl; ctl->sum_cv_loss += ctl->acc[5] * invM; ctl->sum_entropy += ctl->acc[6] * invM;
      ctl->n_mb += 1; ctl->last_kl = kl;
      if (D.adaptive_lr && !explicit_mode) {
        if (kl > 2.0f * D.kl_threshold) ctl->ac_lr = fmaxf(ctl->ac_lr / 1.5f, 1e-6f);
        if (kl < 0.5f * D.kl_threshold) ctl->ac_lr = fminf(ctl->ac_lr * 1.5f, 1e-2f);
      }
      for (int i = 0; i < 8; ++i) ctl->acc[i] = 0.0f;
    }
    __syncthreads();
  }
  if (advance & 2) {
    // ---- move on to the next minibatch: parity flips, inputs of layer 0 are staged (obs rows; normalised states)
    __shared__ int s_next;
    if (tid == 0) {
      int mbn = ctl->mb_index + ((advance & 1) ? 1 : 0);
      if (mbn >= D.num_minibatches) { mbn = 0; ctl->mini_epoch += 1; }
      ctl->mb_index = mbn;
      ctl->step += (advance & 1) ? 1 : 0;
      s_next = mbn;
    }
    __syncthreads();
    const int mbn = s_next, np = ctl->step & 1;
    const size_t r0 = (size_t)mbn * MB;
    const bool upd_rms = D.cv_normalize_input && ctl->mini_epoch == 0 && (advance & 4) == 0;
    // running mean/std update with this minibatch's raw states BEFORE normalising it (train mode, App. C)
    for (int k = tid; k < D.state_dim; k += 256) {
      double mean = D.rms_mean[k], var = D.rms_var[k];
      const double cnt = ctl->rms_count;
      if (upd_rms) {
        double bm = 0.0, bv = 0.0;
        for (int s = 0; s < MB; ++s) bm += D.mb_states[(r0 + s) * D.state_dim + k];
        bm /= MB;
        for (int s = 0; s < MB; ++s) { const double d = D.mb_states[(r0 + s) * D.state_dim + k] - bm; bv += d * d; }
        bv = MB > 1 ? bv / (MB - 1) : 0.0;
        const double delta = bm - mean, tot = cnt + MB;
        const double m2 = var * cnt + bv * MB + delta * delta * cnt * MB / tot;
        mean = mean + delta * MB / tot;
        var = m2 / tot;
        D.rms_mean[k] = mean; D.rms_var[k] = var;
      }
      for (int s = 0; s < MB; ++s) {
        float x = D.mb_states[(r0 + s) * D.state_dim + k];
        if (D.cv_normalize_input) x = clampf((x - (float)mean) / sqrtf((float)var + 1e-5f), -5.0f, 5.0f);
        D.x[2][0][(size_t)np * MB * D.state_dim + s * D.state_dim + k] = x;
      }
    }
    for (int i = tid; i < MB * D.obs_dim; i += 256) {
      const float v = D.mb_obs[r0 * D.obs_dim + i];
      D.x[0][0][(size_t)np * MB * D.obs_dim + i] = v;
      D.x[1][0][(size_t)np * MB * D.obs_dim + i] = v;
    }
    __syncthreads();
    if (tid == 0 && upd_rms) ctl->rms_count += MB;
  }
}


// ------------------------------------------------------------------------------------------------ launch helpers
extern "C" void sdxpk_linear(const float* X, const float* W, const float* b, float* Y, int M, int N, int K, int elu_flag,
                             const double* nmean, const double* nvar, hipStream_t st) {
  dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT);
  hipLaunchKernelGGL(k_linear_mfma, grid, dim3(256), 0, st, X, W, b, Y, M, N, K, elu_flag, nmean, nvar);
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
extern "C" void sdxpk_gae(const SdxpDev* D, const float* last_values, const int64_t* last_dones, hipStream_t st) {
  hipLaunchKernelGGL(k_gae, dim3((D->N + 255) / 256), dim3(256), 0, st, *D, last_values, last_dones);
  if (D->normalize_advantage) hipLaunchKernelGGL(k_adv_norm, dim3(1), dim3(1024), 0, st, *D);
}

template <int MB>
static void launch_step(const SdxpDev* D, hipStream_t st) {
  const int rpb = 4 * D->rows_per_wave;
  for (int l = 0; l < 3; ++l) {
    const int Nl = D->units[l];
    const int Kmax = l == 0 ? (D->state_dim > D->obs_dim ? D->state_dim : D->obs_dim) : D->units[l - 1];
    const int blocks = 3 * ((Nl + rpb - 1) / rpb);
    hipLaunchKernelGGL(k_layer<MB>, dim3(blocks), dim3(256), (size_t)2 * MB * Kmax * sizeof(float), st, *D, l);
  }
  hipLaunchKernelGGL(k_head<MB>, dim3(1), dim3(256), 0, st, *D, 0);
  for (int l = 1; l >= 0; --l) {
    const int K = D->units[l];
    const int blocks = 3 * ((K + 255) / 256) * D->bsplit;
    hipLaunchKernelGGL(k_back<MB>, dim3(blocks), dim3(256), 0, st, *D, l);
    hipLaunchKernelGGL(k_back_fin<MB>, dim3((3 * MB * K + 255) / 256), dim3(256), 0, st, *D, l);
  }
  hipLaunchKernelGGL(k_ctrl<MB>, dim3(1), dim3(256), 0, st, *D, 1 | 2);
}
extern "C" int sdxpk_update_step(const SdxpDev* D, int mb_size, hipStream_t st) {
  switch (mb_size) {
    case 2: launch_step<2>(D, st); return 0;
    case 4: launch_step<4>(D, st); return 0;
    case 8: launch_step<8>(D, st); return 0;
    default: return -1;
  }
}
// stage minibatch 0 (advance = 2: no gradient bookkeeping), bit 2 (=4) suppresses the RMS update
extern "C" int sdxpk_update_begin(const SdxpDev* D, int mb_size, hipStream_t st) {
  switch (mb_size) {
    case 2: hipLaunchKernelGGL(k_ctrl<2>, dim3(1), dim3(256), 0, st, *D, 2); return 0;
    case 4: hipLaunchKernelGGL(k_ctrl<4>, dim3(1), dim3(256), 0, st, *D, 2); return 0;
    case 8: hipLaunchKernelGGL(k_ctrl<8>, dim3(1), dim3(256), 0, st, *D, 2); return 0;
    default: return -1;
  }
}
// flush: apply the last pending optimiser step (the L/HEAD kernels run once more on the last staged minibatch; their
// forward outputs are discarded)
template <int MB>
static void launch_flush(const SdxpDev* D, hipStream_t st) {
  const int rpb = 4 * D->rows_per_wave;
  for (int l = 0; l < 3; ++l) {
    const int Nl = D->units[l];
    const int Kmax = l == 0 ? (D->state_dim > D->obs_dim ? D->state_dim : D->obs_dim) : D->units[l - 1];
    hipLaunchKernelGGL(k_layer<MB>, dim3(3 * ((Nl + rpb - 1) / rpb)), dim3(256), (size_t)2 * MB * Kmax * sizeof(float), st, *D, l);
  }
  hipLaunchKernelGGL(k_head<MB>, dim3(1), dim3(256), 0, st, *D, 1);
}
extern "C" int sdxpk_update_flush_layers(const SdxpDev* D, int mb_size, hipStream_t st) {
  switch (mb_size) {
    case 2: launch_flush<2>(D, st); return 0;
    case 4: launch_flush<4>(D, st); return 0;
    case 8: launch_flush<8>(D, st); return 0;
    default: return -1;
  }
}

// ------------------------------------------------------------------------------------------------ explicit gradients
// world_size > 1: the minibatch gradient is materialised into the flat *_GRADS buffers (same layout as the
// parameters) so that the caller can all-reduce it with RCCL; then sq-norm, clip and Adam run elementwise.
template <int MB>
__global__ __launch_bounds__(256) void k_grad_layer(SdxpDev D, int l) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int par = D.ctrl->step & 1, Nl = D.units[l];
  const int blocks_per_net = (Nl + 3) / 4;
  const int net = blockIdx.x / blocks_per_net, n = (blockIdx.x % blocks_per_net) * 4 + wave;
  const int K = (l == 0) ? (net == 2 ? D.state_dim : D.obs_dim) : D.units[l - 1];
  float* G = net == 2 ? D.cv_g : D.ac_g;
  const size_t woff = net == 0 ? D.off.a_w[l] : net == 1 ? D.off.c_w[l] : D.coff.w[l];
  const size_t boff = net == 0 ? D.off.a_b[l] : net == 1 ? D.off.c_b[l] : D.coff.b[l];
  const float* gx = D.x[net][l] + (size_t)par * MB * K;
  for (int i = tid; i < MB * K; i += 256) sm[i] = gx[i];
  __syncthreads();
  if (n >= Nl) return;
  const float* dy = D.dy[net][l] + (size_t)par * MB * Nl;
  float dyn[MB], sb = 0.0f;
#pragma unroll
  for (int s = 0; s < MB; ++s) { dyn[s] = dy[s * Nl + n]; sb += dyn[s]; }
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
__global__ __launch_bounds__(256) void k_sqnorm(const float* __restrict__ g, size_t n, float scale, float* out) {
  float s = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float v = g[i] * scale; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
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
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float g = G[i] * inv_w * clip;
    const float m = 0.9f * M[i] + 0.1f * g, v = 0.999f * V[i] + 0.001f * g * g;
    M[i] = m; V[i] = v;
    P[i] -= (lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + 1e-8f);
  }
}
__global__ void k_apply_fin(SdxpDev D, int which, float kl_host) {
  SdxpCtrl* ctl = D.ctrl;
  if (which) { ctl->cv_t += 1; ctl->cv_gnorm = sqrtf(ctl->gn2_cv); return; }
  ctl->ac_t += 1; ctl->ac_gnorm = sqrtf(ctl->gn2_ac);
  const float kl = (kl_host == kl_host) ? kl_host : ctl->last_kl / (float)ctl->world;   // NaN -> device value, all-reduced in place
  if (D.adaptive_lr) {   // legacy schedule after every minibatch, on the rank-averaged KL (PS:306-312)
    if (kl > 2.0f * D.kl_threshold) ctl->ac_lr = fmaxf(ctl->ac_lr / 1.5f, 1e-6f);
    if (kl < 0.5f * D.kl_threshold) ctl->ac_lr = fminf(ctl->ac_lr * 1.5f, 1e-2f);
  }
}
template <int MB>
static void launch_backward_explicit(const SdxpDev* D, hipStream_t st) {
  const int rpb = 4 * D->rows_per_wave;
  for (int l = 0; l < 3; ++l) {
    const int Nl = D->units[l];
    const int Kmax = l == 0 ? (D->state_dim > D->obs_dim ? D->state_dim : D->obs_dim) : D->units[l - 1];
    hipLaunchKernelGGL(k_layer<MB>, dim3(3 * ((Nl + rpb - 1) / rpb)), dim3(256), (size_t)2 * MB * Kmax * sizeof(float), st, *D, l);
  }
  hipLaunchKernelGGL(k_head<MB>, dim3(1), dim3(256), 0, st, *D, 0);
  for (int l = 1; l >= 0; --l) {
    const int K = D->units[l];
    hipLaunchKernelGGL(k_back<MB>, dim3(3 * ((K + 255) / 256) * D->bsplit), dim3(256), 0, st, *D, l);
    hipLaunchKernelGGL(k_back_fin<MB>, dim3((3 * MB * K + 255) / 256), dim3(256), 0, st, *D, l);
  }
  for (int l = 0; l < 3; ++l) {
    const int Nl = D->units[l];
    const int Kmax = l == 0 ? (D->state_dim > D->obs_dim ? D->state_dim : D->obs_dim) : D->units[l - 1];
    hipLaunchKernelGGL(k_grad_layer<MB>, dim3(3 * ((Nl + 3) / 4)), dim3(256), (size_t)MB * Kmax * sizeof(float), st, *D, l);
  }
  hipLaunchKernelGGL(k_grad_heads<MB>, dim3(1), dim3(256), 0, st, *D);
  hipLaunchKernelGGL(k_ctrl<MB>, dim3(1), dim3(256), 0, st, *D, 1 | 2 | 8);
}
extern "C" int sdxpk_backward_explicit(const SdxpDev* D, int mb_size, hipStream_t st) {
  switch (mb_size) {
    case 2: launch_backward_explicit<2>(D, st); return 0;
    case 4: launch_backward_explicit<4>(D, st); return 0;
    case 8: launch_backward_explicit<8>(D, st); return 0;
    default: return -1;
  }
}
extern "C" void sdxpk_apply_explicit(const SdxpDev* D, int which, float kl, int world, hipStream_t st) {
  const size_t n = which ? D->coff.total : D->off.total;
  float* acc = which ? &D->ctrl->gn2_cv : &D->ctrl->gn2_ac;
  hipLaunchKernelGGL(k_sqnorm, dim3(512), dim3(256), 0, st, which ? D->cv_g : D->ac_g, n, 1.0f / (float)world, acc);
  hipLaunchKernelGGL(k_adam_explicit, dim3(1024), dim3(256), 0, st, *D, which);
  hipLaunchKernelGGL(k_apply_fin, dim3(1), dim3(1), 0, st, *D, which, kl);
}
