// sdxp_types.h — device-side tables of the PPO engine (shared by sdxp_kernels.hip and sdxp_capi.hip)
#pragma once
#include <stdint.h>

struct SdxpOff {   // offsets (in floats) into the flat actor-critic parameter buffer, torch layout W[out][in]
  size_t a_w[3], a_b[3], mu_w, mu_b, logstd, c_w[3], c_b[3], v_w, v_b, total;
};
struct SdxpCOff {  // central-value network
  size_t w[3], b[3], v_w, v_b, total;
};

// control block living in HBM, advanced by the CTRL kernel (no host round trips inside an epoch)
struct SdxpCtrl {
  int32_t step;          // optimiser steps issued so far (parity selects the factor buffers)
  int32_t mb_index;      // minibatch currently staged
  int32_t mini_epoch;
  int32_t ac_pending, cv_pending;
  int32_t ac_t, cv_t;    // Adam step counters
  int32_t n_mb;
  float ac_lr, cv_lr, ac_lr_applied, cv_lr_applied;
  float ac_gscale, cv_gscale, ac_gnorm, cv_gnorm;
  float ac_bc1, ac_bc2, cv_bc1, cv_bc2;
  float last_kl;
  float sum_a_loss, sum_c_loss, sum_b_loss, sum_kl, sum_cv_loss, sum_entropy;
  float acc[8];          // per-minibatch sums written by the HEAD kernel: [1]a [2]c [3]b [4]kl [5]cv [6]entropy
  float games_sum_rew, games_sum_len, games_cnt, pad0;
  float gn2_ac, gn2_cv;  // explicit-gradient path: sum of squares of the (all-reduced) flat gradients
  int32_t world;
  uint32_t ll_tag;       // last exchange tag handed out to a persistent launch: lives on the DEVICE so that launches captured in a hipGraph
                         // still see tags that only grow (the exchange buffer is never cleared)
  int32_t prev_mb, prev_mini_epoch;
  float n2_part[4];      // grad-norm^2 contributions of heads + trunk layer 2 per net (written by the HEAD kernel)
  float gx[3][3][64];    // Gram matrices (MB x MB) of the inputs of trunk layers 0..2 per net (written by the L kernels)
  double rms_count;
  double ac_b1pow, ac_b2pow, cv_b1pow, cv_b2pow;   // running beta^t of the fused path (bias corrections)
};

#define SDXP_LL_WORDS 65536

// rank-MB factors of one minibatch packed for the multi-rank exchange (floats): per net the inputs X_l [MB][K_l] and the
// pre-activation gradients dY_l [MB][N_l] of the three trunk layers, the head inputs [MB][units[2]], the head gradients
// [MB][34], dlogstd [32] and the minibatch KL
struct SdxpFactOff { uint32_t x[3][3], dy[3][3], h[3], dh, dls, kl, total; };

struct SdxpDev {
  int32_t N, horizon, obs_dim, state_dim, act_dim, units[3];
  int32_t num_minibatches, rows_per_wave, bsplit;
  int32_t clip_value, truncate_grads, normalize_advantage, cv_normalize_input, adaptive_lr;
  float gamma, tau, e_clip, grad_norm, critic_coef, entropy_coef, bounds_coef, kl_threshold;
  uint64_t seed;
  SdxpOff off;
  SdxpCOff coff;
  float *ac, *ac_g, *ac_m, *ac_v, *cv, *cv_g, *cv_m, *cv_v;
  // rollout activations [N, units[l]] of the actor / (unused critic) / central value trunks
  float *h_a[3], *h_v[3];
  // experience buffer, env-major rows r = env*horizon + t (== swap_and_flatten01 order, PS:338-339)
  float *mb_obs, *mb_states, *mb_actions, *mb_mus, *mb_sigmas, *mb_neglogp, *mb_values, *mb_rewards, *mb_dones;
  float *returns, *adv, *last_values, *cur_rew, *cur_len;
  double *rms_mean, *rms_var;
  // small-minibatch update: rank-MB factors, double buffered by step parity
  float* x[3][4];        // x[net][l], l=1..3: [2][MB][units[l-1]] output of trunk layer l-1 (l=3: head input); x[.][0] unused
  float* dy2[3];         // dy2[net]: [2][MB][units[2]] dLoss/d(pre-activation) of trunk layer 2
  float* dxacc[3][2];    // dxacc[net][l]: [2][MB][units[l]] dLoss/d(output of layer l), split-N accumulators
  float *cvx0, *cvx1;    // [N*H, state_dim] normalised central-value inputs (mini-epoch 0 / later mini-epochs)
  float* dhead;          // [2][MB][34]: dmu (cols 0..act_dim-1), dV critic (32), dV central value (33)
  float* dlogstd;        // [2][32]
  SdxpCtrl* ctrl;
  long long* dbg;        // [64] phase timestamps (s_memtime) written by thread 0 of the single-block kernels
  SdxpFactOff foff;
  float *fact, *fact_all; // [foff.total] this rank's factors; [world][foff.total] all ranks' (all-gathered by the caller)
  int32_t world;
  int32_t obs_cols;      // columns of the caller's observation rows (<= obs_dim); obs_pad [N, obs_dim] holds the zero-padded copy
  float* obs_pad;
  float* sqn_part;       // [512] block partials of the deterministic gradient-norm reduction
  size_t g_tail;         // ALL_GRADS = [ac_g | pad | cv_g | pad | kl word | pad]: offset (floats, from ac_g) of the kl word
  int32_t bf16;          // large-minibatch path: trunk GEMMs on bf16 MFMA (fp32 sources, accumulation, weights and optimiser state)
  unsigned long long* ll; // [SDXP_LL_WORDS] (value, step tag) words exchanged between the CUs of the persistent update kernel
};

// workspace of the large-minibatch update path (sdxp_bigmb.hip), allocated by sdxp_capi.hip
struct SdxpBigWs {
  float* h[3][3];             // trunk outputs  [MB][units[l]] in fp32 (the heads read l = 2; bf16 NT runs do not write l = 0, 1: see hn)
  float* dy[3][3];            // dLoss/d(pre-activation) [MB][units[l]] (NT path: only l = 2, written by the head kernels)
  float* mu;                  // [MB][24]
  float* dmu;                 // [MB][24]
  float* v;                   // [2][MB] critic / central value
  float* dv;                  // [2][MB]
  float* part;                // split partials (max over layers of S * (N*K + N)), also head partials
  double* dpart;              // [nsplit][state_dim][2]
  int MB, nsplit;
  size_t part_region;         // floats of split partials per network inside `part`
  // ---- NT path (sdx_gemm_nt.h): every operand staged k-contiguous in the element type of the run (fp32, or bf16 with mixed_precision)
  int nt;                     // 1: the trunk products run on k_gemm_nt
  int KC, MBp, Rp;            // chunk elements (32 fp32 / 64 bf16); MB rounded up to KC; dataset rows + 64 (row stride of the transposed inputs)
  int kp[3][3];               // padded reduction length of forward layer l of net
  void* xn[3];                // dataset inputs [R][kp0]: [0] observations (actor + critic), [1] cvx0, [2] cvx1
  void* xt[3];                // their transposes [kp0][Rp]
  void* wn[3][3];             // weights [units[l]][kp[net][l]]
  void* wt[3][3];             // transposed weights [K_l][units[l]], l = 1, 2
  void* hn[3][2];             // layer outputs l = 0, 1 in the element type [MB][units[l]] (fp32 runs: the h arrays themselves)
  void* ht[3][2];             // ... transposed [units[l]][MBp]
  void* dyn[3][3];            // gradients [MB][units[l]], l = 1, 2 (fp32 runs, l = 2: dy[net][2] itself)
  void* dyt[3][3];            // ... transposed [units[l]][MBp], l = 0, 1, 2
  // ---- fp32 runs: weight gradients read the [row][feature] arrays themselves (k_gemm_tt): no transposed copy of an activation or a
  // gradient is written or read (ht / dyt / xt stay unused); SDXP_BIGMB_TT=0 keeps the transposed-copy form
  int tt;
  float* zeros;               // 64 zero floats: the rows a ragged last chunk lacks
};
