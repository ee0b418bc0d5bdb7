// sdxp_capi.hip — host side of the sdxp_* C ABI (include/seqdex.h): the rl_games A2CAgent inner loops.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "sdx_common.h"
#include "sdxp_types.h"

extern "C" {
void sdxpk_linear(const float*, const float*, const float*, float*, int, int, int, int, const double*, const double*, hipStream_t);
void sdxpk_linear2(const float*, const float*, const float*, float*, int, int, const double*, const double*,
                   const float*, const float*, const float*, float*, int, int, const double*, const double*, int, int, hipStream_t);
void sdxpk_act_heads(const SdxpDev*, int, const float*, const float*, const int64_t*, const float*, float*, uint64_t, hipStream_t);
void sdxpk_store_rewards(const SdxpDev*, int, const float*, const int64_t*, hipStream_t);
void sdxpk_value_head(const SdxpDev*, float*, hipStream_t);
void sdxpk_gae(const SdxpDev*, const float*, const int64_t*, hipStream_t);
int sdxpk_update_step(const SdxpDev*, int, hipStream_t);
int sdxpk_update_begin(const SdxpDev*, int, hipStream_t);
int sdxpk_update_flush_layers(const SdxpDev*, int, hipStream_t);
int sdxpk_backward_explicit(const SdxpDev*, int, hipStream_t);
int sdxpk_persist_supported(const SdxpDev*, int, int);
void sdxpk_pad_obs(const SdxpDev*, const float*, hipStream_t);
int sdxpk_backward_factors(const SdxpDev*, int, hipStream_t);
int sdxpk_grads_from_factors(const SdxpDev*, int, hipStream_t);
int sdxpk_apply_factors(const SdxpDev*, int, hipStream_t);
int sdxpk_apply_factors_fused(const SdxpDev*, int, unsigned*, hipStream_t);
int sdxpk_update_persistent(const SdxpDev*, int, unsigned*, hipStream_t);
int sdxpk_fwd_bwd_persistent(const SdxpDev*, unsigned*, hipStream_t);
int sdxpk_prenorm(const SdxpDev*, int, hipStream_t);
void sdxpk_apply_explicit(const SdxpDev*, int, float, int, hipStream_t);
}

extern "C" size_t sdxpk_big_part_floats(const SdxpDev* D, int MB);
extern "C" int sdxpk_big_nsplit(int MB);
extern "C" int sdxpk_big_nt_enabled(const SdxpDev* D, int MB);
extern "C" void sdxpk_big_prenorm(const SdxpDev* D, const SdxpBigWs* ws, hipStream_t st);
extern "C" void sdxpk_big_step(const SdxpDev* D, const SdxpBigWs* ws, int mb, int me, hipStream_t st);
extern "C" void sdxpk_apply_flat(const SdxpDev* D, hipStream_t st);

struct sdxp_agent {
  int device = 0;
  sdxp_config cfg;
  SdxpDev D;
  std::vector<void*> allocs;
  struct TensorInfo { void* ptr; int64_t shape[4]; int ndim; int dtype; } tinfo[SDXP_T_COUNT];
  float* stats_dev = nullptr;
  uint64_t act_counter = 0;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  int graph_chunk = 0;
  bool use_persist = false;      // persistent register-resident update kernel (sdxp_persist.hip)
  unsigned* bar_dev = nullptr;   // [64] grid-barrier counter (+ fail flags at [32] persistent kernels, [34] one-launch apply; [36] its launch counter = tag of its exchange words)
  bool last_was_step = false;    // the most recent persistent launch was a single forward/backward (no restore possible on failure)
  bool use_persist_step = false; // multi-rank path: forward/backward of one minibatch as one persistent-style launch
  bool use_fused_apply = false;  // multi-rank path: gradient rebuild + norm + clip + Adam + control block as one launch (SDXP_APPLY_IMPL=fused; default: three launches)
  unsigned* fail_host = nullptr; // pinned mirror of the fail flag, refreshed after every persistent update
  // what a persistent update touches before it can fail (old mu/sigma rows, running mean/std, control block): saved at the start
  // of the call so that sdxp_update_status can put it back and the caller can repeat the epoch on the hipGraph path
  float *mus_bak = nullptr, *sig_bak = nullptr;
  double* rms_bak = nullptr;
  SdxpCtrl* ctrl_bak = nullptr;
  bool big = false;              // minibatch_size > 8: GEMM-shaped update path (sdxp_bigmb.hip)
  SdxpBigWs bigws;
  int big_me = 0, big_next = 0;  // multi-rank big path: mini-epoch / expected minibatch of the next sdxp_backward call
  long max_steps = 0;            // SDXP_MAX_STEPS read ONCE at sdxp_create (0: no limit): debug limit of optimiser steps per sdxp_update
  std::string err;
};

static thread_local std::string gp_create_err = "";
static_assert(sizeof(SdxpCtrl) == 2544, "keep seqdex_amd/ppo.py::Ctrl in sync with SdxpCtrl");

#define PCHK(h, call)                                                                                  \
  do {                                                                                                 \
    hipError_t _e = (call);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      char _b[512];                                                                                    \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      if (h) (h)->err = _b; else gp_create_err = _b;                                                   \
      return SDX_ERR_HIP;                                                                              \
    }                                                                                                  \
  } while (0)

template <typename T>
static int palloc(sdxp_agent* h, T** p, size_t count) {
  void* q = nullptr;
  if (count == 0) count = 1;
  hipError_t e = hipMalloc(&q, count * sizeof(T));
  if (e != hipSuccess) { h->err = std::string("hipMalloc failed: ") + hipGetErrorString(e); return SDX_ERR_NOMEM; }
  e = hipMemset(q, 0, count * sizeof(T));
  if (e != hipSuccess) { h->err = std::string("hipMemset failed: ") + hipGetErrorString(e); return SDX_ERR_HIP; }
  h->allocs.push_back(q);
  *p = (T*)q;
  return SDX_OK;
}

static void pset(sdxp_agent* h, int id, void* p, int dtype, std::initializer_list<int64_t> shape) {
  auto& t = h->tinfo[id];
  t.ptr = p; t.dtype = dtype; t.ndim = (int)shape.size();
  int i = 0;
  for (auto s : shape) t.shape[i++] = s;
  for (; i < 4; ++i) t.shape[i] = 1;
}

// torch.nn.Linear default init: kaiming_uniform_(a=sqrt(5)) -> U(-1/sqrt(fan_in), 1/sqrt(fan_in)); rl_games then
// zeroes every bias (App. C).  Host-side, seeded.
static void init_linear(std::vector<float>& p, size_t woff, int out, int in, std::mt19937_64& rng) {
  const float bound = 1.0f / std::sqrt((float)in);
  std::uniform_real_distribution<float> U(-bound, bound);
  for (size_t i = 0; i < (size_t)out * in; ++i) p[woff + i] = U(rng);
}

extern "C" int sdxp_create(const sdxp_config* cfg, int32_t device, uint64_t seed, sdxp_handle* out) {
  if (!cfg || !out) { gp_create_err = "sdxp_create: bad argument"; return SDX_ERR_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    gp_create_err = "sdxp_create: no HIP device visible; libseqdex_hip has no CPU fallback";
    return SDX_ERR_NO_DEVICE;
  }
  const int B = cfg->horizon * cfg->num_actors;
  if (cfg->minibatch <= 0 || B % cfg->minibatch != 0) { gp_create_err = "sdxp_create: batch_size % minibatch_size != 0"; return SDX_ERR_INVALID; }
  if (cfg->minibatch != cfg->cv_minibatch || cfg->mini_epochs != cfg->cv_mini_epochs) {
    gp_create_err = "sdxp_create: the fused update needs equal minibatch_size / mini_epochs for actor-critic and central value";
    return SDX_ERR_INVALID;
  }
  if (cfg->obs_cols < 0 || cfg->obs_cols > cfg->obs_dim) { gp_create_err = "sdxp_create: obs_cols must be in [0, obs_dim]"; return SDX_ERR_INVALID; }
  if (cfg->units[2] != 256 || cfg->units[0] % 4 || cfg->units[1] % 4 || cfg->act_dim > 32 || cfg->obs_dim % 4 || cfg->state_dim % 4) {
    gp_create_err = "sdxp_create: unsupported network shape"; return SDX_ERR_INVALID;
  }
  if (cfg->obs_dim < 4 || cfg->state_dim < 4) { gp_create_err = "sdxp_create: obs_dim and state_dim must be at least 4"; return SDX_ERR_INVALID; }
  if (cfg->obs_dim > 1024 || cfg->state_dim > 1024) {     // SDXP_NORM_K of k_linear_mfma: the per-k normalisation tables of the first layer
    gp_create_err = "sdxp_create: obs_dim / state_dim above 1024 are not supported by the rollout's first-layer kernel"; return SDX_ERR_INVALID;
  }
  sdxp_agent* h = new sdxp_agent();
  if (const char* ms = getenv("SDXP_MAX_STEPS")) {
    h->max_steps = atol(ms);
    if (h->max_steps > 0) fprintf(stderr, "libseqdex_hip: SDXP_MAX_STEPS=%ld - DEBUG LIMIT: every sdxp_update of this handle stops after %ld optimiser steps\n", h->max_steps, h->max_steps);
  }
  h->device = device;
  h->cfg = *cfg;
  PCHK(h, hipSetDevice(device));
  SdxpDev& D = h->D;
  memset(&D, 0, sizeof(D));
  D.N = cfg->num_actors; D.horizon = cfg->horizon; D.obs_dim = cfg->obs_dim; D.state_dim = cfg->state_dim;
  D.act_dim = cfg->act_dim;
  for (int i = 0; i < 3; ++i) D.units[i] = cfg->units[i];
  D.num_minibatches = B / cfg->minibatch;
  D.rows_per_wave = 1;
  D.bsplit = 8;
  D.clip_value = cfg->clip_value; D.truncate_grads = cfg->truncate_grads; D.normalize_advantage = cfg->normalize_advantage;
  D.cv_normalize_input = cfg->cv_normalize_input; D.adaptive_lr = cfg->adaptive_lr;
  D.gamma = cfg->gamma; D.tau = cfg->tau; D.e_clip = cfg->e_clip; D.grad_norm = cfg->grad_norm;
  D.critic_coef = cfg->critic_coef; D.entropy_coef = cfg->entropy_coef; D.bounds_coef = cfg->bounds_loss_coef;
  D.kl_threshold = cfg->kl_threshold; D.seed = seed;
  D.bf16 = cfg->mixed_precision != 0;
  // ---- flat parameter layouts
  {
    size_t o = 0;
    int in = cfg->obs_dim;
    for (int l = 0; l < 3; ++l) { D.off.a_w[l] = o; o += (size_t)cfg->units[l] * in; D.off.a_b[l] = o; o += cfg->units[l]; in = cfg->units[l]; }
    D.off.mu_w = o; o += (size_t)cfg->act_dim * in; D.off.mu_b = o; o += cfg->act_dim;
    D.off.logstd = o; o += cfg->act_dim;
    in = cfg->obs_dim;
    for (int l = 0; l < 3; ++l) { D.off.c_w[l] = o; o += (size_t)cfg->units[l] * in; D.off.c_b[l] = o; o += cfg->units[l]; in = cfg->units[l]; }
    D.off.v_w = o; o += in; D.off.v_b = o; o += 1;
    D.off.total = o;
    o = 0; in = cfg->state_dim;
    for (int l = 0; l < 3; ++l) { D.coff.w[l] = o; o += (size_t)cfg->units[l] * in; D.coff.b[l] = o; o += cfg->units[l]; in = cfg->units[l]; }
    D.coff.v_w = o; o += in; D.coff.v_b = o; o += 1;
    D.coff.total = o;
  }
  int rc;
#define PAL(ptr, count) if ((rc = palloc(h, &(ptr), (size_t)(count))) != SDX_OK) { gp_create_err = h->err; sdxp_destroy(h); return rc; }
  PAL(D.ac, D.off.total); PAL(D.ac_m, D.off.total); PAL(D.ac_v, D.off.total);
  // both flat gradients and the KL word live in ONE allocation so that the multi-rank path needs a single all-reduce per step
  const size_t g_ac = (D.off.total + 63) / 64 * 64, g_cv = (D.coff.total + 63) / 64 * 64;
  PAL(D.ac_g, g_ac + g_cv + 64);
  D.cv_g = D.ac_g + g_ac; D.g_tail = g_ac + g_cv;
  PAL(D.cv, D.coff.total); PAL(D.cv_m, D.coff.total); PAL(D.cv_v, D.coff.total);
  const size_t N = D.N, H = D.horizon, R = N * H;
  for (int l = 0; l < 3; ++l) { PAL(D.h_a[l], N * cfg->units[l]); PAL(D.h_v[l], N * cfg->units[l]); }
  PAL(D.mb_obs, R * cfg->obs_dim); PAL(D.mb_states, R * cfg->state_dim); PAL(D.mb_actions, R * cfg->act_dim);
  PAL(D.mb_mus, R * cfg->act_dim); PAL(D.mb_sigmas, R * cfg->act_dim); PAL(D.mb_neglogp, R); PAL(D.mb_values, R);
  PAL(D.mb_rewards, R); PAL(D.mb_dones, R); PAL(D.returns, R); PAL(D.adv, R); PAL(D.last_values, N);
  PAL(D.cur_rew, N); PAL(D.cur_len, N);
  PAL(D.rms_mean, cfg->state_dim); PAL(D.rms_var, cfg->state_dim);
  h->big = cfg->minibatch > 8;
  const int MB = h->big ? 1 : cfg->minibatch;   // the rank-MB factor buffers below belong to the small-minibatch paths
  for (int net = 0; net < 3; ++net) {
    for (int l = 0; l < 3; ++l) {
      PAL(D.x[net][l + 1], (size_t)2 * MB * cfg->units[l]);
      if (l < 2) PAL(D.dxacc[net][l], (size_t)2 * MB * cfg->units[l]);
    }
    PAL(D.dy2[net], (size_t)2 * MB * cfg->units[2]);
  }
  PAL(D.cvx0, R * cfg->state_dim); PAL(D.cvx1, R * cfg->state_dim);
  PAL(D.dbg, 64); PAL(D.dhead, (size_t)2 * MB * 34); PAL(D.dlogstd, 64); PAL(D.ctrl, 1); PAL(h->stats_dev, 16);
  PAL(h->bar_dev, 64); PAL(D.ll, SDXP_LL_WORDS);
  D.obs_cols = cfg->obs_cols > 0 ? cfg->obs_cols : cfg->obs_dim;
  PAL(D.obs_pad, (size_t)N * cfg->obs_dim);
  PAL(h->mus_bak, R * cfg->act_dim); PAL(h->sig_bak, R * cfg->act_dim); PAL(h->rms_bak, (size_t)2 * cfg->state_dim); PAL(h->ctrl_bak, 1);
  {   // factor exchange buffers of the multi-rank path
    uint32_t o = 0;
    for (int net = 0; net < 3; ++net)
      for (int l = 0; l < 3; ++l) { D.foff.x[net][l] = o; o += MB * (l == 0 ? (net == 2 ? cfg->state_dim : cfg->obs_dim) : cfg->units[l - 1]); }
    for (int net = 0; net < 3; ++net)
      for (int l = 0; l < 3; ++l) { D.foff.dy[net][l] = o; o += MB * cfg->units[l]; }
    for (int net = 0; net < 3; ++net) { D.foff.h[net] = o; o += MB * cfg->units[2]; }
    D.foff.dh = o; o += MB * 34; D.foff.dls = o; o += 32; D.foff.kl = o; o += 1;
    D.foff.total = (o + 63) / 64 * 64;
    D.world = cfg->world_size > 0 ? cfg->world_size : 1;
    PAL(D.fact, D.foff.total); PAL(D.fact_all, (size_t)D.foff.total * D.world); PAL(D.sqn_part, 16384);   // 2 x SDXP_SQN_STRIDE partials, then (SDXP_TW_OFF) the tagged 16-byte words of the one-launch apply (sdxp_kernels.hip)
  }
  if (h->big) {   // activations and pre-activation gradients of one large minibatch, split partials of the weight gradients
    SdxpBigWs& w = h->bigws;
    const size_t BM = (size_t)cfg->minibatch;
    w.MB = cfg->minibatch;
    w.nsplit = sdxpk_big_nsplit(cfg->minibatch);
    for (int net = 0; net < 3; ++net)
      for (int l = 0; l < 3; ++l) { PAL(w.h[net][l], BM * cfg->units[l]); PAL(w.dy[net][l], BM * cfg->units[l]); }
    PAL(w.mu, BM * 24); PAL(w.dmu, BM * 24); PAL(w.v, 2 * BM); PAL(w.dv, 2 * BM);
    w.part_region = sdxpk_big_part_floats(&D, cfg->minibatch);
    PAL(w.part, 3 * w.part_region);
    PAL(w.dpart, (size_t)w.nsplit * cfg->state_dim * 2);
    w.nt = sdxpk_big_nt_enabled(&D, cfg->minibatch);
    w.tt = 0; w.zeros = nullptr;
    {
      const char* e = getenv("SDXP_BIGMB_TT");
      if (w.nt && !D.bf16 && !(e && e[0] == '0')) { w.tt = 1; PAL(w.zeros, 64); }
    }
    if (w.nt) {   // staged operands of the NT products (sdx_gemm_nt.h); hipMemset by palloc: the padding the kernels never write stays zero
      const size_t ES = D.bf16 ? 2 : 4;
      char* q;
      w.KC = D.bf16 ? 64 : 32;
      w.MBp = (cfg->minibatch + w.KC - 1) / w.KC * w.KC;
      w.Rp = (int)R + 64;
      for (int net = 0; net < 3; ++net)
        for (int l = 0; l < 3; ++l) {
          const int K = l == 0 ? (net == 2 ? cfg->state_dim : cfg->obs_dim) : cfg->units[l - 1];
          w.kp[net][l] = (K + w.KC - 1) / w.KC * w.KC;
        }
      for (int i = 0; i < 3; ++i) {
        const int kp0 = w.kp[i == 0 ? 0 : 2][0];
        PAL(q, R * kp0 * ES); w.xn[i] = q;
        PAL(q, (size_t)kp0 * w.Rp * ES); w.xt[i] = q;
      }
      for (int net = 0; net < 3; ++net)
        for (int l = 0; l < 3; ++l) {
          PAL(q, (size_t)cfg->units[l] * w.kp[net][l] * ES); w.wn[net][l] = q;
          w.wt[net][l] = nullptr;
          if (l > 0) { PAL(q, (size_t)w.kp[net][l] * cfg->units[l] * ES); w.wt[net][l] = q; }
          PAL(q, (size_t)cfg->units[l] * w.MBp * ES); w.dyt[net][l] = q;
          w.dyn[net][l] = nullptr;
          if (l == 2 && !D.bf16) w.dyn[net][l] = w.dy[net][l];
          else if (l > 0) { PAL(q, BM * cfg->units[l] * ES); w.dyn[net][l] = q; }
          if (l < 2) {
            PAL(q, (size_t)cfg->units[l] * w.MBp * ES); w.ht[net][l] = q;
            if (D.bf16) { PAL(q, BM * cfg->units[l] * ES); w.hn[net][l] = q; } else w.hn[net][l] = w.h[net][l];
          }
        }
    }
  }
#undef PAL
  // ---- parameter init
  {
    std::mt19937_64 rng(seed * 0x9E3779B97F4A7C15ull + 12345);
    std::vector<float> p(D.off.total, 0.0f), c(D.coff.total, 0.0f);
    int in = cfg->obs_dim;
    for (int l = 0; l < 3; ++l) { init_linear(p, D.off.a_w[l], cfg->units[l], in, rng); in = cfg->units[l]; }
    init_linear(p, D.off.mu_w, cfg->act_dim, in, rng);
    in = cfg->obs_dim;
    for (int l = 0; l < 3; ++l) { init_linear(p, D.off.c_w[l], cfg->units[l], in, rng); in = cfg->units[l]; }
    init_linear(p, D.off.v_w, 1, in, rng);
    in = cfg->state_dim;
    for (int l = 0; l < 3; ++l) { init_linear(c, D.coff.w[l], cfg->units[l], in, rng); in = cfg->units[l]; }
    init_linear(c, D.coff.v_w, 1, in, rng);
    PCHK(h, hipMemcpy(D.ac, p.data(), p.size() * 4, hipMemcpyHostToDevice));
    PCHK(h, hipMemcpy(D.cv, c.data(), c.size() * 4, hipMemcpyHostToDevice));
    std::vector<double> ones(cfg->state_dim, 1.0);
    PCHK(h, hipMemcpy(D.rms_var, ones.data(), ones.size() * 8, hipMemcpyHostToDevice));
    SdxpCtrl ctl;
    memset(&ctl, 0, sizeof(ctl));
    ctl.ac_lr = cfg->lr; ctl.cv_lr = cfg->cv_lr; ctl.ac_gscale = 1.0f; ctl.cv_gscale = 1.0f;
    ctl.ac_bc1 = ctl.ac_bc2 = ctl.cv_bc1 = ctl.cv_bc2 = 1.0f;
    ctl.rms_count = 1.0;
    ctl.ac_b1pow = ctl.ac_b2pow = ctl.cv_b1pow = ctl.cv_b2pow = 1.0;
    ctl.world = cfg->world_size > 0 ? cfg->world_size : 1;
    PCHK(h, hipMemcpy(D.ctrl, &ctl, sizeof(ctl), hipMemcpyHostToDevice));
  }
  {
    hipDeviceProp_t prop;
    PCHK(h, hipGetDeviceProperties(&prop, device));
    const char* impl = getenv("SDXP_UPDATE_IMPL");   // "persist" (default when supported) | "graph"
    const bool supported = sdxpk_persist_supported(&D, cfg->minibatch, prop.multiProcessorCount);
    h->use_persist = supported && !(impl && std::string(impl) == "graph");
    const char* simpl = getenv("SDXP_STEP_IMPL");   // multi-rank path: "kernels" forces the multi-kernel forward/backward
    h->use_persist_step = supported && !(simpl && std::string(simpl) == "kernels");
    const char* aimpl = getenv("SDXP_APPLY_IMPL");  // multi-rank path: "fused" = the one-launch apply (opt-in: its grid-wide meeting needs every workgroup resident, i.e. the GPU to itself)
    h->use_fused_apply = aimpl && std::string(aimpl) == "fused";
    PCHK(h, hipHostMalloc((void**)&h->fail_host, sizeof(unsigned), hipHostMallocDefault));
    *h->fail_host = 0;
  }
  pset(h, SDXP_T_AC_PARAMS, D.ac, SDX_F32, {(int64_t)D.off.total});
  pset(h, SDXP_T_AC_GRADS, D.ac_g, SDX_F32, {(int64_t)D.off.total});
  pset(h, SDXP_T_CV_PARAMS, D.cv, SDX_F32, {(int64_t)D.coff.total});
  pset(h, SDXP_T_CV_GRADS, D.cv_g, SDX_F32, {(int64_t)D.coff.total});
  pset(h, SDXP_T_MB_OBS, D.mb_obs, SDX_F32, {(int64_t)N, (int64_t)H, cfg->obs_dim});
  pset(h, SDXP_T_MB_STATES, D.mb_states, SDX_F32, {(int64_t)N, (int64_t)H, cfg->state_dim});
  pset(h, SDXP_T_MB_ACTIONS, D.mb_actions, SDX_F32, {(int64_t)N, (int64_t)H, cfg->act_dim});
  pset(h, SDXP_T_MB_MUS, D.mb_mus, SDX_F32, {(int64_t)N, (int64_t)H, cfg->act_dim});
  pset(h, SDXP_T_MB_SIGMAS, D.mb_sigmas, SDX_F32, {(int64_t)N, (int64_t)H, cfg->act_dim});
  pset(h, SDXP_T_MB_NEGLOGP, D.mb_neglogp, SDX_F32, {(int64_t)N, (int64_t)H});
  pset(h, SDXP_T_MB_VALUES, D.mb_values, SDX_F32, {(int64_t)N, (int64_t)H});
  pset(h, SDXP_T_MB_REWARDS, D.mb_rewards, SDX_F32, {(int64_t)N, (int64_t)H});
  pset(h, SDXP_T_MB_DONES, D.mb_dones, SDX_F32, {(int64_t)N, (int64_t)H});
  pset(h, SDXP_T_RETURNS, D.returns, SDX_F32, {(int64_t)R});
  pset(h, SDXP_T_ADVANTAGES, D.adv, SDX_F32, {(int64_t)R});
  pset(h, SDXP_T_CV_RMS_MEAN, D.rms_mean, SDX_F64, {cfg->state_dim});
  pset(h, SDXP_T_CV_RMS_VAR, D.rms_var, SDX_F64, {cfg->state_dim});
  pset(h, SDXP_T_STATS, D.ctrl, SDX_F32, {(int64_t)(sizeof(SdxpCtrl) / 4)});
  pset(h, SDXP_T_LAST_VALUES, D.last_values, SDX_F32, {(int64_t)N});
  pset(h, SDXP_T_DEBUG, D.dbg, SDX_I64, {64});
  pset(h, SDXP_T_ALL_GRADS, D.ac_g, SDX_F32, {(int64_t)(D.g_tail + 64)});
  pset(h, SDXP_T_FACTORS, D.fact, SDX_F32, {(int64_t)D.foff.total});
  pset(h, SDXP_T_FACTORS_ALL, D.fact_all, SDX_F32, {(int64_t)D.world, (int64_t)D.foff.total});
  pset(h, SDXP_T_AC_ADAM_M, D.ac_m, SDX_F32, {(int64_t)D.off.total});
  pset(h, SDXP_T_AC_ADAM_V, D.ac_v, SDX_F32, {(int64_t)D.off.total});
  pset(h, SDXP_T_CV_ADAM_M, D.cv_m, SDX_F32, {(int64_t)D.coff.total});
  pset(h, SDXP_T_CV_ADAM_V, D.cv_v, SDX_F32, {(int64_t)D.coff.total});
  *out = h;
  return SDX_OK;
}

extern "C" int sdxp_destroy(sdxp_handle h) {
  if (!h) return SDX_ERR_INVALID;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
  if (h->graph) (void)hipGraphDestroy(h->graph);
  if (h->fail_host) (void)hipHostFree(h->fail_host);
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return SDX_OK;
}

extern "C" int sdxp_tensor(sdxp_handle h, int32_t id, void** dev_ptr, int64_t shape[4], int32_t* ndim, int32_t* dtype) {
  if (!h) return SDX_ERR_INVALID;
  if (id < 0 || id >= SDXP_T_COUNT || !dev_ptr || !shape || !ndim || !dtype) { h->err = "sdxp_tensor: bad argument"; return SDX_ERR_INVALID; }
  const auto& t = h->tinfo[id];
  *dev_ptr = t.ptr;
  for (int i = 0; i < 4; ++i) shape[i] = t.shape[i];
  *ndim = t.ndim;
  *dtype = t.dtype;
  return SDX_OK;
}

extern "C" int64_t sdxp_param_count(sdxp_handle h, int32_t which) {
  if (!h) return -1;
  return which == 0 ? (int64_t)h->D.off.total : (int64_t)h->D.coff.total;
}

static int plaunch_ok(sdxp_handle h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { h->err = std::string(what) + ": " + hipGetErrorString(e); return SDX_ERR_HIP; }
  return SDX_OK;
}

__global__ void k_ctrl_begin_epoch(SdxpCtrl* c) {
  c->mb_index = 0; c->mini_epoch = 0; c->n_mb = 0; c->prev_mb = 0; c->prev_mini_epoch = 0;
  c->sum_a_loss = c->sum_c_loss = c->sum_b_loss = c->sum_kl = c->sum_cv_loss = c->sum_entropy = 0.0f;
  for (int i = 0; i < 8; ++i) c->acc[i] = 0.0f;
}
__global__ void k_ctrl_begin_rollout(SdxpCtrl* c) { c->games_sum_rew = c->games_sum_len = c->games_cnt = 0.0f; }

// trunk forward of `net` (0 actor, 2 central value) over M rows into the h_* activation buffers
static void trunk_forward(sdxp_agent* h, int net, const float* x, int M, hipStream_t st) {
  const SdxpDev& D = h->D;
  const float* P = net == 2 ? D.cv : D.ac;
  float* const* hb = net == 2 ? D.h_v : D.h_a;
  int in = net == 2 ? D.state_dim : D.obs_dim;
  const float* cur = x;
  for (int l = 0; l < 3; ++l) {
    const size_t wo = net == 0 ? D.off.a_w[l] : D.coff.w[l], bo = net == 0 ? D.off.a_b[l] : D.coff.b[l];
    const bool norm = (net == 2 && l == 0 && D.cv_normalize_input);
    sdxpk_linear(cur, P + wo, P + bo, hb[l], M, D.units[l], in, 1, norm ? D.rms_mean : nullptr, norm ? D.rms_var : nullptr, st);
    cur = hb[l];
    in = D.units[l];
  }
}

// actor and central-value trunks side by side: the same layer of both networks is ONE launch (3 launches per env step instead of 6)
static void trunk_forward2(sdxp_agent* h, const float* obs, const float* states, int M, hipStream_t st) {
  const SdxpDev& D = h->D;
  const float* xa = obs; const float* xv = states;
  int ina = D.obs_dim, inv = D.state_dim;
  for (int l = 0; l < 3; ++l) {
    const bool norm = (l == 0 && D.cv_normalize_input);
    sdxpk_linear2(xa, D.ac + D.off.a_w[l], D.ac + D.off.a_b[l], D.h_a[l], D.units[l], ina, nullptr, nullptr,
                  xv, D.cv + D.coff.w[l], D.cv + D.coff.b[l], D.h_v[l], D.units[l], inv, norm ? D.rms_mean : nullptr, norm ? D.rms_var : nullptr,
                  M, 1, st);
    xa = D.h_a[l]; xv = D.h_v[l];
    ina = inv = D.units[l];
  }
}

extern "C" int sdxp_act(sdxp_handle h, int32_t t, const float* obs_dev, const float* states_dev, const int64_t* dones_dev,
                        const float* eps_dev, float* actions_out_dev, void* stream) {
  if (!h || !obs_dev || !states_dev || !actions_out_dev || t < 0 || t >= h->D.horizon) { if (h) h->err = "sdxp_act: bad argument"; return SDX_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  if (t == 0) hipLaunchKernelGGL(k_ctrl_begin_rollout, dim3(1), dim3(1), 0, st, h->D.ctrl);
  if (h->D.obs_cols != h->D.obs_dim) { sdxpk_pad_obs(&h->D, obs_dev, st); obs_dev = h->D.obs_pad; }
  trunk_forward2(h, obs_dev, states_dev, h->D.N, st);
  sdxpk_act_heads(&h->D, t, obs_dev, states_dev, dones_dev, eps_dev, actions_out_dev, h->act_counter++, st);
  return plaunch_ok(h, "sdxp_act");
}

extern "C" int sdxp_store_rewards(sdxp_handle h, int32_t t, const float* rew_dev, const int64_t* dones_after_dev, void* stream) {
  if (!h || !rew_dev || t < 0 || t >= h->D.horizon) return SDX_ERR_INVALID;
  sdxpk_store_rewards(&h->D, t, rew_dev, dones_after_dev, (hipStream_t)stream);
  return plaunch_ok(h, "sdxp_store_rewards");
}

extern "C" int sdxp_finish_rollout(sdxp_handle h, const float* last_states_dev, const int64_t* last_dones_dev, void* stream) {
  if (!h || !last_states_dev) return SDX_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  trunk_forward(h, 2, last_states_dev, h->D.N, st);
  sdxpk_value_head(&h->D, h->D.last_values, st);
  sdxpk_gae(&h->D, h->D.last_values, last_dones_dev, st);
  return plaunch_ok(h, "sdxp_finish_rollout");
}

// The three stages of sdxp_finish_rollout as separate entry points, for callers that drive rl_games-style code themselves
// (policy_sequencing/policy_seq_runner.py:329-343 calls get_values, discount_values and prepare_dataset one by one).
extern "C" int sdxp_get_values(sdxp_handle h, const float* states_dev, float* values_out_dev, void* stream) {
  if (!h || !states_dev || !values_out_dev) return SDX_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  trunk_forward(h, 2, states_dev, h->D.N, st);
  sdxpk_value_head(&h->D, values_out_dev, st);
  return plaunch_ok(h, "sdxp_get_values");
}
extern "C" void sdxpk_gae_only(const SdxpDev*, const float*, const int64_t*, hipStream_t);
extern "C" void sdxpk_adv_norm(const SdxpDev*, hipStream_t);
extern "C" int sdxp_discount_values(sdxp_handle h, const float* last_values_dev, const int64_t* last_dones_dev, void* stream) {
  if (!h || !last_values_dev) return SDX_ERR_INVALID;
  sdxpk_gae_only(&h->D, last_values_dev, last_dones_dev, (hipStream_t)stream);
  return plaunch_ok(h, "sdxp_discount_values");
}
extern "C" int sdxp_prepare_dataset(sdxp_handle h, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  sdxpk_adv_norm(&h->D, (hipStream_t)stream);
  return plaunch_ok(h, "sdxp_prepare_dataset");
}

extern "C" int sdxp_update(sdxp_handle h, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int MB = h->cfg.minibatch;
  if (h->big) {   // minibatch_size > 8 (ppo_continuous_insert.yaml: 4096): GEMM-shaped step, explicit gradients, clip + Adam
    hipLaunchKernelGGL(k_ctrl_begin_epoch, dim3(1), dim3(1), 0, st, h->D.ctrl);
    sdxpk_big_prenorm(&h->D, &h->bigws, st);
    for (int me = 0; me < h->cfg.mini_epochs; ++me)
      for (int mb = 0; mb < h->D.num_minibatches; ++mb) {
        sdxpk_big_step(&h->D, &h->bigws, mb, me, st);
        sdxpk_apply_flat(&h->D, st);
      }
    return plaunch_ok(h, "sdxp_update(large minibatch)");
  }
  if (MB != 2 && MB != 4 && MB != 8) {
    h->err = "sdxp_update: minibatch_size must be 2/4/8 (rank-MB paths) or > 8 (GEMM path)";
    return SDX_ERR_INVALID;
  }
  long total = (long)h->cfg.mini_epochs * h->D.num_minibatches;
  // debug step limit (tests/test_gpu_fullsize_properties.py pins the N = 1024 persistent update to the oracle step for step): the
  // update phase stops after SDXP_MAX_STEPS optimiser steps, in minibatch order, on whichever path the handle uses
  // (read once, at sdxp_create, and announced on stderr there: a stray variable cannot silently truncate the epochs of a production run)
  if (h->max_steps > 0 && h->max_steps < total) total = h->max_steps;
  if (h->fail_host && *h->fail_host) {
    h->use_persist = false;   // a grid barrier of the persistent kernel timed out earlier: fall back for good
    *h->fail_host = 0;
    h->err = "sdxp_update: the persistent update kernel of the previous call timed out waiting for an exchange word (not all 256 "
             "workgroups co-resident?); that epoch's update was NOT applied (call sdxp_update_status after sdxp_update to catch "
             "this in time to repeat it); this handle now uses the hipGraph path";
    return SDX_ERR_STATE;
  }
  if (h->use_persist) {
    const size_t ra = (size_t)h->D.N * h->D.horizon * h->D.act_dim * sizeof(float), sd = (size_t)h->D.state_dim * sizeof(double);
    PCHK(h, hipMemcpyAsync(h->mus_bak, h->D.mb_mus, ra, hipMemcpyDeviceToDevice, st));
    PCHK(h, hipMemcpyAsync(h->sig_bak, h->D.mb_sigmas, ra, hipMemcpyDeviceToDevice, st));
    PCHK(h, hipMemcpyAsync(h->rms_bak, h->D.rms_mean, sd, hipMemcpyDeviceToDevice, st));
    PCHK(h, hipMemcpyAsync(h->rms_bak + h->D.state_dim, h->D.rms_var, sd, hipMemcpyDeviceToDevice, st));
    PCHK(h, hipMemcpyAsync(h->ctrl_bak, h->D.ctrl, sizeof(SdxpCtrl), hipMemcpyDeviceToDevice, st));
  }
  hipLaunchKernelGGL(k_ctrl_begin_epoch, dim3(1), dim3(1), 0, st, h->D.ctrl);
  if (h->use_persist) {
    sdxpk_prenorm(&h->D, MB, st);
    h->last_was_step = false;
    if (sdxpk_update_persistent(&h->D, (int)total, h->bar_dev + 32, st) != 0) { h->err = "persistent update launch failed"; return SDX_ERR_HIP; }
    PCHK(h, hipMemcpyAsync(h->fail_host, h->bar_dev + 32, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    return plaunch_ok(h, "sdxp_update(persistent)");
  }
  sdxpk_update_begin(&h->D, MB, st);
  // a chunk of optimiser steps is captured once into a hipGraph (no step-dependent kernel arguments: all state
  // lives in the device control block) and replayed; the remainder is launched eagerly
  const int chunk = 32;
  long done = 0;
  if (total >= chunk) {
    if (!h->graph_exec) {
      hipStream_t cs;
      PCHK(h, hipStreamCreate(&cs));
      PCHK(h, hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < chunk; ++i) sdxpk_update_step(&h->D, MB, cs);
      PCHK(h, hipStreamEndCapture(cs, &h->graph));
      PCHK(h, hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
      PCHK(h, hipStreamDestroy(cs));
      h->graph_chunk = chunk;
    }
    for (; done + chunk <= total; done += chunk) PCHK(h, hipGraphLaunch(h->graph_exec, st));
  }
  for (; done < total; ++done) sdxpk_update_step(&h->D, MB, st);
  sdxpk_update_flush_layers(&h->D, MB, st);
  return plaunch_ok(h, "sdxp_update");
}

// Multi-rank path.  which: 0 = runs forward/backward of ALL THREE networks for minibatch `mb` and materialises both flat
// gradients (the networks advance in the same launches); 1 = no-op kept for symmetry with sdxp_apply.  mb == -1 begins an
// epoch's update phase (resets the control block, stages minibatch 0); otherwise minibatches must come in order.
extern "C" int sdxp_backward(sdxp_handle h, int32_t which, int32_t mb, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int MB = h->cfg.minibatch;
  if (h->big) {
    if (which == 1) return SDX_OK;
    if (mb < 0) {
      hipLaunchKernelGGL(k_ctrl_begin_epoch, dim3(1), dim3(1), 0, st, h->D.ctrl);
      sdxpk_big_prenorm(&h->D, &h->bigws, st);
      h->big_me = 0; h->big_next = 0;
      return plaunch_ok(h, "sdxp_backward(begin, large minibatch)");
    }
    if (mb != h->big_next) { h->err = "sdxp_backward: minibatches must come in order"; return SDX_ERR_INVALID; }
    sdxpk_big_step(&h->D, &h->bigws, mb, h->big_me, st);
    h->big_next = mb + 1;
    if (h->big_next >= h->D.num_minibatches) { h->big_next = 0; h->big_me += 1; }
    return plaunch_ok(h, "sdxp_backward(large minibatch)");
  }
  if (MB != 2 && MB != 4 && MB != 8) { h->err = "sdxp_backward: minibatch_size must be 2/4/8"; return SDX_ERR_INVALID; }
  if (which == 1) return SDX_OK;
  if (mb < 0) {
    hipLaunchKernelGGL(k_ctrl_begin_epoch, dim3(1), dim3(1), 0, st, h->D.ctrl);
    sdxpk_update_begin(&h->D, MB, st);
    return plaunch_ok(h, "sdxp_backward(begin)");
  }
  if (mb >= h->D.num_minibatches) { h->err = "sdxp_backward: minibatch index out of range"; return SDX_ERR_INVALID; }
  sdxpk_backward_explicit(&h->D, MB, st);
  return plaunch_ok(h, "sdxp_backward");
}
// Factor exchange variant of the multi-rank step: sdxp_backward_factors leaves this rank's rank-MB factors in SDXP_T_FACTORS,
// the caller all-gathers them into SDXP_T_FACTORS_ALL [world, F] (RCCL all_gather, 194 KB per rank instead of a 13.4 MB
// all-reduce), sdxp_grads_from_factors rebuilds the SUM over ranks of the minibatch gradients (and of the KL) in the *_GRADS
// buffers, then sdxp_apply(0, -INFINITY); sdxp_apply(1) as usual.
extern "C" int sdxp_backward_factors(sdxp_handle h, int32_t mb, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int MB = h->cfg.minibatch;
  if (MB != 2 && MB != 4 && MB != 8) { h->err = "sdxp_backward_factors: minibatch_size must be 2/4/8"; return SDX_ERR_INVALID; }
  if (mb < 0) return sdxp_backward(h, 0, -1, stream);
  if (mb >= h->D.num_minibatches) { h->err = "sdxp_backward_factors: minibatch index out of range"; return SDX_ERR_INVALID; }
  if (h->use_persist_step) {   // one launch: forward + backward on 256 CUs with tagged-word exchange, factors straight into D.fact
    h->last_was_step = true;
    if (sdxpk_fwd_bwd_persistent(&h->D, h->bar_dev + 32, st) != 0) { h->err = "persistent forward/backward launch failed"; return SDX_ERR_HIP; }
    return plaunch_ok(h, "sdxp_backward_factors(persistent)");
  }
  sdxpk_backward_factors(&h->D, MB, st);
  return plaunch_ok(h, "sdxp_backward_factors");
}
extern "C" int sdxp_grads_from_factors(sdxp_handle h, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  sdxpk_grads_from_factors(&h->D, h->cfg.minibatch, (hipStream_t)stream);
  return plaunch_ok(h, "sdxp_grads_from_factors");
}
// sdxp_grads_from_factors + sdxp_apply(0, -INFINITY) + sdxp_apply(1) in four launches (the multi-rank step is launch-latency bound)
extern "C" int sdxp_apply_factors(sdxp_handle h, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  if (h->use_fused_apply) {   // one launch with a grid-wide ticket; -1: shape or occupancy does not allow it on this device
    if (sdxpk_apply_factors_fused(&h->D, h->cfg.minibatch, h->bar_dev, (hipStream_t)stream) == 0) return plaunch_ok(h, "sdxp_apply_factors(fused)");
    h->use_fused_apply = false;
  }
  sdxpk_apply_factors(&h->D, h->cfg.minibatch, (hipStream_t)stream);
  return plaunch_ok(h, "sdxp_apply_factors");
}
// clip_grad_norm_ + Adam on the (caller-all-reduced, SUM) flat gradient of network `which`; gradients are divided by
// world_size here.  kl: rank-averaged KL for the legacy LR schedule, or NaN to use SdxpCtrl.last_kl that the caller
// all-reduced in place (SUM) through the SDXP_T_STATS view.
extern "C" int sdxp_apply(sdxp_handle h, int32_t which, float kl, void* stream) {
  if (!h || which < 0 || which > 1) return SDX_ERR_INVALID;
  sdxpk_apply_explicit(&h->D, which, kl, h->cfg.world_size > 0 ? h->cfg.world_size : 1, (hipStream_t)stream);
  return plaunch_ok(h, "sdxp_apply");
}
// Blocks until the update launched by sdxp_update on `stream` has finished.  SDX_OK, or SDX_ERR_STATE if the persistent kernel gave
// up (an exchange word never arrived): nothing of that epoch was applied, the inputs it had already touched (old mu/sigma rows,
// running mean/std, control block) are restored, and the handle is switched to the hipGraph path - calling sdxp_update again
// repeats the epoch there.
extern "C" int sdxp_update_status(sdxp_handle h, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  PCHK(h, hipStreamSynchronize(st));
  if (!h->fail_host) return SDX_OK;
  if (h->use_fused_apply) {   // the one-launch apply gave up at its ticket: from that step on nothing was applied
    PCHK(h, hipMemcpy(h->fail_host, h->bar_dev + 34, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (*h->fail_host) {
      *h->fail_host = 0;
      h->use_fused_apply = false;
      PCHK(h, hipMemset(h->bar_dev + 34, 0, sizeof(unsigned)));
      h->err = "sdxp_apply_factors: the one-launch apply timed out at its grid-wide meeting (not all workgroups resident?); optimiser steps of "
               "this epoch were skipped from that launch on; this handle now uses the three-launch apply - restore a checkpoint";
      return SDX_ERR_STATE;
    }
  }
  PCHK(h, hipMemcpy(h->fail_host, h->bar_dev + 32, sizeof(unsigned), hipMemcpyDeviceToHost));
  if (!*h->fail_host) return SDX_OK;
  if (h->last_was_step) {   // multi-rank path: a forward/backward launch gave up; its factors were garbage and may have been applied
    *h->fail_host = 0;
    h->use_persist_step = false;
    PCHK(h, hipMemset(h->bar_dev, 0, 36 * sizeof(unsigned)));
    {   // tags of the failed launch must never be handed out again
      SdxpCtrl c;
      PCHK(h, hipMemcpy(&c, h->D.ctrl, sizeof(c), hipMemcpyDeviceToHost));
      c.ll_tag += 1024u;
      PCHK(h, hipMemcpy(h->D.ctrl, &c, sizeof(c), hipMemcpyHostToDevice));
    }
    h->err = "sdxp_backward_factors: a persistent forward/backward launch timed out waiting for an exchange word (not all 256 "
             "workgroups co-resident?); at least one optimiser step of this epoch used invalid factors; this handle now uses the "
             "multi-kernel forward/backward - restore a checkpoint";
    return SDX_ERR_STATE;
  }
  *h->fail_host = 0;
  h->use_persist = false;
  const size_t ra = (size_t)h->D.N * h->D.horizon * h->D.act_dim * sizeof(float), sd = (size_t)h->D.state_dim * sizeof(double);
  PCHK(h, hipMemsetAsync(h->bar_dev, 0, 36 * sizeof(unsigned), st));
  PCHK(h, hipMemcpyAsync(h->D.mb_mus, h->mus_bak, ra, hipMemcpyDeviceToDevice, st));
  PCHK(h, hipMemcpyAsync(h->D.mb_sigmas, h->sig_bak, ra, hipMemcpyDeviceToDevice, st));
  PCHK(h, hipMemcpyAsync(h->D.rms_mean, h->rms_bak, sd, hipMemcpyDeviceToDevice, st));
  PCHK(h, hipMemcpyAsync(h->D.rms_var, h->rms_bak + h->D.state_dim, sd, hipMemcpyDeviceToDevice, st));
  PCHK(h, hipMemcpyAsync(h->D.ctrl, h->ctrl_bak, sizeof(SdxpCtrl), hipMemcpyDeviceToDevice, st));
  PCHK(h, hipStreamSynchronize(st));
  {   // the restored block carries the tag from BEFORE the failed launch: move it past every word that launch may have written
    SdxpCtrl c;
    PCHK(h, hipMemcpy(&c, h->D.ctrl, sizeof(c), hipMemcpyDeviceToHost));
    c.ll_tag += (uint32_t)((long)h->cfg.mini_epochs * h->D.num_minibatches) + 1024u;
    PCHK(h, hipMemcpy(h->D.ctrl, &c, sizeof(c), hipMemcpyHostToDevice));
  }
  h->err = "sdxp_update: the persistent update kernel timed out waiting for an exchange word (not all 256 workgroups co-resident?); "
           "nothing was applied, inputs restored; this handle now uses the hipGraph path - call sdxp_update again";
  return SDX_ERR_STATE;
}
// Optimiser state that lives in the device control block and that a checkpoint must carry besides the parameters and Adam moments
// (rl_games restores all of it: optimizer.state_dict() holds the step counters, running_mean_std.count is part of the central-value
// state_dict, last_lr is re-applied to the param groups).  Blocking.
extern "C" int sdxp_get_state(sdxp_handle h, sdxp_opt_state* out, void* stream) {
  if (!h || !out) return SDX_ERR_INVALID;
  PCHK(h, hipStreamSynchronize((hipStream_t)stream));
  SdxpCtrl c;
  PCHK(h, hipMemcpy(&c, h->D.ctrl, sizeof(c), hipMemcpyDeviceToHost));
  out->rms_count = c.rms_count; out->ac_t = c.ac_t; out->cv_t = c.cv_t; out->ac_lr = c.ac_lr; out->cv_lr = c.cv_lr;
  return SDX_OK;
}
extern "C" int sdxp_set_state(sdxp_handle h, const sdxp_opt_state* in, void* stream) {
  if (!h || !in) return SDX_ERR_INVALID;
  if (in->ac_t < 0 || in->cv_t < 0 || !(in->rms_count >= 0.0) || !(in->ac_lr > 0.0f) || !(in->cv_lr > 0.0f)) {
    h->err = "sdxp_set_state: negative step counter / count or non-positive learning rate"; return SDX_ERR_INVALID;
  }
  PCHK(h, hipStreamSynchronize((hipStream_t)stream));
  SdxpCtrl c;
  PCHK(h, hipMemcpy(&c, h->D.ctrl, sizeof(c), hipMemcpyDeviceToHost));
  c.rms_count = in->rms_count;
  c.ac_t = in->ac_t; c.cv_t = in->cv_t;
  c.ac_lr = c.ac_lr_applied = in->ac_lr; c.cv_lr = c.cv_lr_applied = in->cv_lr;
  c.ac_b1pow = std::pow(0.9, (double)in->ac_t); c.ac_b2pow = std::pow(0.999, (double)in->ac_t);
  c.cv_b1pow = std::pow(0.9, (double)in->cv_t); c.cv_b2pow = std::pow(0.999, (double)in->cv_t);
  c.ac_bc1 = in->ac_t ? (float)(1.0 - c.ac_b1pow) : 1.0f; c.ac_bc2 = in->ac_t ? (float)(1.0 - c.ac_b2pow) : 1.0f;
  c.cv_bc1 = in->cv_t ? (float)(1.0 - c.cv_b1pow) : 1.0f; c.cv_bc2 = in->cv_t ? (float)(1.0 - c.cv_b2pow) : 1.0f;
  PCHK(h, hipMemcpy(h->D.ctrl, &c, sizeof(c), hipMemcpyHostToDevice));
  return SDX_OK;
}
extern "C" int sdxp_update_impl(sdxp_handle h) { return !h ? 0 : (h->big ? 2 : (h->use_persist ? 1 : 0)); }

extern "C" const char* sdxp_last_error(sdxp_handle h) { return h ? h->err.c_str() : gp_create_err.c_str(); }
