// sdx_tvtrain.hip — the transition-value trainer of the policy chain on gfx950.
// Reference: policy_sequencing/transition_value_trainer.py:180-248 (TValue_Trainer.init_TValue_function / train_rollout) for the
// network of policy_sequencing/terminal_value_function.py:30-46 (GraspInsertTValue: 4-256-128-64-2, ELU after EVERY layer):
//   batch = 512 success + 512 failure camera-frame quaternions drawn at random, + U(-1, 1) * 0.05 noise, renormalised (TT:213-222)
//   loss  = BCEWithLogitsLoss(net(batch), one-hot [failure, success]) (TT:225-228), Adam(lr 1e-3) (TT:187,229-231)
// One iteration = sample kernel, 4 forward GEMMs (bias + ELU fused), the loss kernel, 4 weight-gradient GEMMs (bias row sums fused)
// + 3 data-gradient GEMMs (x ELU' fused) on the fp32 matrix cores (sdx_gemm.h), one Adam kernel.  No CPU path.
#include "sdx_gemm.h"
#include "../../include/seqdex.h"
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <math.h>

static const int TV_IN = 4, TV_U[4] = {256, 128, 64, 2};

struct sdxtv_trainer {
  int device = 0, B = 0;
  uint64_t seed = 0;
  uint64_t iter = 0;     // batches drawn so far (stream of the sampler)
  int t = 0;             // Adam steps taken
  float *p = nullptr, *g = nullptr, *m = nullptr, *v = nullptr;   // [SDX_TV_PARAMS] torch layout: W1 b1 W2 b2 W3 b3 W4 b4
  float* x = nullptr;    // [B, 4]
  float* h[4] = {nullptr, nullptr, nullptr, nullptr};    // layer outputs [B, U_l]
  float* dy[4] = {nullptr, nullptr, nullptr, nullptr};   // dLoss/d(pre-activation) [B, U_l]
  float* loss = nullptr; // [1]
  size_t woff[4], boff[4];
  std::vector<void*> allocs;
  std::string err;
};
static thread_local std::string g_tv_err;

#define TVCHK(h, call) do { hipError_t _e = (call); if (_e != hipSuccess) { std::string _m = std::string(#call) + ": " + hipGetErrorString(_e); \
  if (h) (h)->err = _m; else g_tv_err = _m; return SDX_ERR_HIP; } } while (0)

// rows [0, B/2): success samples, rows [B/2, B): failure samples (TT:213-222); sampling is with replacement (counter-based hash) where
// the reference uses random.sample without replacement - the same distribution for datasets much larger than the batch
static __global__ void k_tv_sample(const float* __restrict__ succ, int ns, const float* __restrict__ fail, int nf, int B, uint64_t seed,
                                   uint64_t iter, float* __restrict__ x) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B) return;
  const bool ok = r < B / 2;
  const float* src = ok ? succ : fail;
  const int n = ok ? ns : nf;
  const uint64_t hsh = sdx_hash(seed, iter, (uint64_t)r);
  const float* q = src + (size_t)(hsh % (uint64_t)n) * 4;
  float v[4], nn = 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t u = sdx_hash(seed ^ 0x7A11ull, iter * 4 + j, (uint64_t)r);
    const float uni = (float)((u >> 40) & 0xFFFFFFull) * (2.0f / 16777216.0f) - 1.0f;    // U[-1, 1)
    v[j] = q[j] + uni * 0.05f;
    nn += v[j] * v[j];
  }
  nn = sqrtf(nn);
#pragma unroll
  for (int j = 0; j < 4; ++j) x[(size_t)r * 4 + j] = v[j] / nn;
}
// BCEWithLogitsLoss (mean over B x 2) on the ELU outputs y; dY = dLoss/dy * ELU'(y) = gradient w.r.t. the last pre-activation
static __global__ __launch_bounds__(1024) void k_tv_loss(const float* __restrict__ y, int B, float* __restrict__ dy, float* __restrict__ loss) {
  __shared__ float s[16];
  float acc = 0.0f;
  const float inv = 1.0f / (float)(2 * B);
  for (int i = threadIdx.x; i < 2 * B; i += 1024) {
    const int r = i >> 1, j = i & 1;
    const float t = (r < B / 2) ? (j == 1 ? 1.0f : 0.0f) : (j == 0 ? 1.0f : 0.0f);     // success_buf, TT:203-205
    const float z = y[i];
    acc += fmaxf(z, 0.0f) - z * t + log1pf(expf(-fabsf(z)));
    const float sg = 1.0f / (1.0f + expf(-z));
    dy[i] = (sg - t) * inv * belu_grad_from_out(z);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tsum = 0.0f;
    for (int w = 0; w < 16; ++w) tsum += s[w];
    loss[0] = tsum * inv;
  }
}
// torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay
static __global__ void k_tv_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int n, float lr,
                                 float bc1, float bc2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = 0.9f * m[i] + 0.1f * gi;
  const float vi = 0.999f * v[i] + 0.001f * gi * gi;
  m[i] = mi; v[i] = vi;
  p[i] -= (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + 1e-8f);
}

extern "C" int sdxtv_create(int32_t batch, int32_t device, uint64_t seed, sdxtv_handle* out) {
  if (!out || batch < 2 || batch % 2 || batch > 1024 * 64) { g_tv_err = "sdxtv_create: batch must be even, 2..65536"; return SDX_ERR_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_tv_err = "sdxtv_create: no HIP device visible; libseqdex_hip has no CPU fallback"; return SDX_ERR_NO_DEVICE; }
  sdxtv_trainer* h = new sdxtv_trainer();
  h->device = device; h->B = batch; h->seed = seed;
  TVCHK(h, hipSetDevice(device));
  auto al = [&](float** p, size_t n) -> int {
    hipError_t e = hipMalloc((void**)p, n * sizeof(float));
    if (e != hipSuccess) { h->err = std::string("hipMalloc: ") + hipGetErrorString(e); return SDX_ERR_HIP; }
    h->allocs.push_back(*p);
    return hipMemset(*p, 0, n * sizeof(float)) == hipSuccess ? SDX_OK : SDX_ERR_HIP;
  };
  int rc = SDX_OK;
  size_t o = 0;
  int in = TV_IN;
  for (int l = 0; l < 4; ++l) { h->woff[l] = o; o += (size_t)TV_U[l] * in; h->boff[l] = o; o += TV_U[l]; in = TV_U[l]; }
  if (o != SDX_TV_PARAMS) { g_tv_err = "sdxtv_create: parameter layout mismatch"; delete h; return SDX_ERR_INVALID; }
  float** four[4] = {&h->p, &h->g, &h->m, &h->v};
  for (int i = 0; i < 4 && rc == SDX_OK; ++i) rc = al(four[i], SDX_TV_PARAMS);
  if (rc == SDX_OK) rc = al(&h->x, (size_t)batch * 4);
  for (int l = 0; l < 4 && rc == SDX_OK; ++l) { rc = al(&h->h[l], (size_t)batch * TV_U[l]); if (rc == SDX_OK) rc = al(&h->dy[l], (size_t)batch * TV_U[l]); }
  if (rc == SDX_OK) rc = al(&h->loss, 4);
  if (rc != SDX_OK) { g_tv_err = h->err; for (void* p : h->allocs) (void)hipFree(p); delete h; return rc; }
  *out = h;
  return SDX_OK;
}
extern "C" int sdxtv_destroy(sdxtv_handle h) {
  if (!h) return SDX_ERR_INVALID;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return SDX_OK;
}
extern "C" const char* sdxtv_last_error(sdxtv_handle h) { return h ? h->err.c_str() : g_tv_err.c_str(); }

extern "C" int sdxtv_tensor(sdxtv_handle h, int32_t id, void** dev_ptr, int64_t shape[4], int32_t* ndim, int32_t* dtype) {
  if (!h || !dev_ptr || !shape || !ndim || !dtype) return SDX_ERR_INVALID;
  for (int i = 0; i < 4; ++i) shape[i] = 1;
  *dtype = SDX_F32; *ndim = 1;
  switch (id) {
    case SDXTV_T_PARAMS: *dev_ptr = h->p; shape[0] = SDX_TV_PARAMS; break;
    case SDXTV_T_GRADS: *dev_ptr = h->g; shape[0] = SDX_TV_PARAMS; break;
    case SDXTV_T_ADAM_M: *dev_ptr = h->m; shape[0] = SDX_TV_PARAMS; break;
    case SDXTV_T_ADAM_V: *dev_ptr = h->v; shape[0] = SDX_TV_PARAMS; break;
    case SDXTV_T_BATCH: *dev_ptr = h->x; shape[0] = h->B; shape[1] = 4; *ndim = 2; break;
    case SDXTV_T_LOSS: *dev_ptr = h->loss; shape[0] = 1; break;
    case SDXTV_T_OUTPUT: *dev_ptr = h->h[3]; shape[0] = h->B; shape[1] = 2; *ndim = 2; break;
    default: h->err = "sdxtv_tensor: unknown id"; return SDX_ERR_INVALID;
  }
  return SDX_OK;
}

static void tv_forward(sdxtv_trainer* h, const float* x, int rows, float* const* outs, hipStream_t st) {
  const float* cur = x;
  int in = TV_IN;
  for (int l = 0; l < 4; ++l) {
    GemmArgs g = {cur, in, h->p + h->woff[l], in, outs[l], TV_U[l], 0, rows, TV_U[l], in, in, h->p + h->boff[l], nullptr, 0, nullptr};
    gemm<0, 0, 1>(&g, 1, 1, st);                              // ELU after every layer, the output layer included (TV:45)
    cur = outs[l];
    in = TV_U[l];
  }
}

extern "C" int sdxtv_sample(sdxtv_handle h, const float* succ_dev, int32_t n_succ, const float* fail_dev, int32_t n_fail, void* stream) {
  if (!h || !succ_dev || !fail_dev || n_succ <= 0 || n_fail <= 0) { if (h) h->err = "sdxtv_sample: bad argument"; return SDX_ERR_INVALID; }
  hipLaunchKernelGGL(k_tv_sample, dim3((h->B + 255) / 256), dim3(256), 0, (hipStream_t)stream, succ_dev, n_succ, fail_dev, n_fail, h->B, h->seed,
                     h->iter, h->x);
  h->iter += 1;
  return SDX_OK;
}
// forward on SDXTV_T_BATCH, BCE-with-logits loss -> SDXTV_T_LOSS, backward -> SDXTV_T_GRADS, one Adam step
extern "C" int sdxtv_step(sdxtv_handle h, float lr, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int B = h->B;
  tv_forward(h, h->x, B, h->h, st);
  hipLaunchKernelGGL(k_tv_loss, dim3(1), dim3(1024), 0, st, h->h[3], B, h->dy[3], h->loss);
  for (int l = 3; l >= 0; --l) {
    const int Nl = TV_U[l], Kl = l == 0 ? TV_IN : TV_U[l - 1];
    const float* Xl = l == 0 ? h->x : h->h[l - 1];
    GemmArgs gw = {h->dy[l], Nl, Xl, Kl, h->g + h->woff[l], Kl, 0, Nl, Kl, B, B, nullptr, nullptr, 0, h->g + h->boff[l]};
    gemm<1, 1, 4>(&gw, 1, 1, st);                             // G_l = dY_l^T X_l, bias gradient = row sums
    if (l > 0) {
      GemmArgs gx = {h->dy[l], Nl, h->p + h->woff[l], Kl, h->dy[l - 1], Kl, 0, B, Kl, Nl, Nl, nullptr, h->h[l - 1], Kl, nullptr};
      gemm<0, 1, 3>(&gx, 1, 1, st);
    }
  }
  h->t += 1;
  const float bc1 = 1.0f - powf(0.9f, (float)h->t), bc2 = 1.0f - powf(0.999f, (float)h->t);
  hipLaunchKernelGGL(k_tv_adam, dim3((SDX_TV_PARAMS + 255) / 256), dim3(256), 0, st, h->p, h->g, h->m, h->v, (int)SDX_TV_PARAMS, lr, bc1, bc2);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { h->err = std::string("sdxtv_step: ") + hipGetErrorString(e); return SDX_ERR_HIP; }
  return SDX_OK;
}
extern "C" int sdxtv_train(sdxtv_handle h, const float* succ_dev, int32_t n_succ, const float* fail_dev, int32_t n_fail, int32_t iters, float lr,
                           void* stream) {
  if (!h || iters < 0) return SDX_ERR_INVALID;
  for (int i = 0; i < iters; ++i) {
    int rc = sdxtv_sample(h, succ_dev, n_succ, fail_dev, n_fail, stream);
    if (rc != SDX_OK) return rc;
    rc = sdxtv_step(h, lr, stream);
    if (rc != SDX_OK) return rc;
  }
  return SDX_OK;
}
// net(x) for n <= batch rows (validation, TT:235-246): out_dev [n, 2] = the ELU outputs
extern "C" int sdxtv_predict(sdxtv_handle h, const float* x_dev, int32_t n, float* out_dev, void* stream) {
  if (!h || !x_dev || !out_dev || n <= 0 || n > h->B) { if (h) h->err = "sdxtv_predict: 1 <= n <= batch"; return SDX_ERR_INVALID; }
  float* outs[4] = {h->h[0], h->h[1], h->h[2], out_dev};
  tv_forward(h, x_dev, n, outs, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? SDX_OK : SDX_ERR_HIP;
}
