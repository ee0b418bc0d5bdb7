// sdx_capi.hip — host side of the sdx_* C ABI declared in include/seqdex.h (simulator + task seam).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sdx_common.h"
#include "sdx_const_build.h"

extern "C" {
void sdxk_pre_physics(const SdxConst*, const SdxBuf*, const float*, const uint8_t*, const int32_t*, int, hipStream_t);
void sdxk_post_physics(const SdxConst*, const SdxBuf*, int, hipStream_t);
void sdxk_physics(const SdxConst*, const SdxBuf*, hipStream_t);
void sdxk_kinematics(const SdxConst*, const SdxBuf*, hipStream_t);
extern "C" void sdxk_seg_camera(const SdxConst*, const SdxBuf*, hipStream_t);
void sdxk_orient_pregrasp(const SdxConst*, const SdxBuf*, const uint8_t*, int, int, hipStream_t);
void sdxk_orient_post_reset(const SdxConst*, const SdxBuf*, const uint8_t*, hipStream_t);
}

struct sdx_sim {
  int device = 0;
  SdxConst* d_const = nullptr;
  SdxConst h_const;
  SdxBuf buf;
  uint8_t* orient_mask = nullptr;   // device copy of the reset flags of an Orient reset event
  std::vector<void*> allocs;
  struct TensorInfo { void* ptr; int64_t shape[4]; int ndim; int dtype; } tinfo[SDX_T_COUNT];
  bool has_piles = false;
  std::string err;
};

static thread_local std::string g_create_err = "";

#define HIPCHK(h, call)                                                                              \
  do {                                                                                               \
    hipError_t _e = (call);                                                                          \
    if (_e != hipSuccess) {                                                                          \
      char _b[512];                                                                                  \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      if (h) (h)->err = _b; else g_create_err = _b;                                                  \
      return SDX_ERR_HIP;                                                                            \
    }                                                                                                \
  } while (0)

template <typename T>
static int dalloc(sdx_sim* h, T** p, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, count * sizeof(T));
  if (e != hipSuccess) { h->err = std::string("hipMalloc failed: ") + hipGetErrorString(e); return SDX_ERR_NOMEM; }
  e = hipMemset(q, 0, count * sizeof(T));
  if (e != hipSuccess) { h->err = std::string("hipMemset failed: ") + hipGetErrorString(e); return SDX_ERR_HIP; }
  h->allocs.push_back(q);
  *p = (T*)q;
  return SDX_OK;
}

static void set_tensor(sdx_sim* h, int id, void* p, int dtype, std::initializer_list<int64_t> shape) {
  auto& t = h->tinfo[id];
  t.ptr = p;
  t.dtype = dtype;
  t.ndim = (int)shape.size();
  int i = 0;
  for (auto s : shape) t.shape[i++] = s;
  for (; i < 4; ++i) t.shape[i] = 1;
}

static void qrot_host(const float q[4], const float v[3], float out[3]) {
  float u[3] = {q[0], q[1], q[2]};
  float t[3] = {2 * (u[1] * v[2] - u[2] * v[1]), 2 * (u[2] * v[0] - u[0] * v[2]), 2 * (u[0] * v[1] - u[1] * v[0])};
  out[0] = v[0] + q[3] * t[0] + (u[1] * t[2] - u[2] * t[1]);
  out[1] = v[1] + q[3] * t[1] + (u[2] * t[0] - u[0] * t[2]);
  out[2] = v[2] + q[3] * t[2] + (u[0] * t[1] - u[1] * t[0]);
}

extern "C" int sdx_create(const sdx_scene_desc* scene, int32_t num_envs, int32_t device, uint64_t seed, sdx_handle* out) {
  if (!scene || !out || num_envs <= 0) { g_create_err = "sdx_create: bad argument"; return SDX_ERR_INVALID; }
  if (scene->abi_version != SDX_ABI_VERSION) { g_create_err = "sdx_create: scene.abi_version mismatch"; return SDX_ERR_INVALID; }
  {   // the shape tables k_physics indexes without further checks
    const int ns = scene->n_static, nr = scene->n_rbox;
    bool ok = ns >= 0 && ns <= SDX_MAX_STATIC && nr >= 0 && nr <= SDX_MAX_RBOX && scene->n_static_sub >= 0 && scene->n_static_sub <= SDX_MAX_STATIC_SUB;
    // candidate body pairs of the broadphase: <= 16 per lane of the 512-thread workgroup (the pair rank's 13 bits)
    ok = ok && SDX_NFREE * SDX_MAX_STATIC + SDX_NFREE * (SDX_NFREE - 1) / 2 + nr * (SDX_NFREE + SDX_MAX_STATIC) <= 16 * 512;   // (the enumeration runs over all static slots)
    for (int t = 0; ok && t < SDX_NBRICK_TYPES; ++t)
      ok = scene->brick_nsub[t] >= 1 && scene->brick_nsub[t] <= SDX_MAX_SUB && scene->hollow_nsub[t] >= 0 && scene->hollow_nsub[t] <= SDX_MAX_SUB_HOLLOW &&
           (!scene->seg_hollow || scene->hollow_nsub[t] >= 1);
    for (int r = 0; ok && r < SDX_MAX_STATIC_TAB; ++r) {
      const bool used = r < ns || (scene->static_var_slot >= 0 && (r == scene->static_var_row[0] || r == scene->static_var_row[1] || r == scene->static_var_row[2]));
      if (used) ok = scene->static_sub_n[r] >= 1 && scene->static_sub_n[r] <= 63 && scene->static_sub_first[r] >= 0 &&
                     scene->static_sub_first[r] + scene->static_sub_n[r] <= scene->n_static_sub;
    }
    if (scene->static_var_slot >= ns) ok = false;
    for (int k = 0; ok && scene->static_var_slot >= 0 && k < 3; ++k) ok = scene->static_var_row[k] >= 0 && scene->static_var_row[k] < SDX_MAX_STATIC_TAB;
    if (!ok) { g_create_err = "sdx_create: scene shape tables out of range (n_static, n_rbox, brick / static compounds)"; return SDX_ERR_INVALID; }
    // the warm start's contact age is a 4-bit saturating counter in the cache key (k_physics solve()): a ramp longer than 16 solves would never end
    if (!(scene->warm_age >= 0.0f && scene->warm_age <= 16.0f)) { g_create_err = "sdx_create: warm_age must lie in [0, 16] (the contact age saturates at 16 solves)"; return SDX_ERR_INVALID; }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_err = "sdx_create: no HIP device visible; libseqdex_hip has no CPU fallback";
    return SDX_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) { g_create_err = "sdx_create: bad device index"; return SDX_ERR_INVALID; }
  sdx_sim* h = new sdx_sim();
  h->device = device;
  sdx_sim* none = nullptr;
  (void)none;
  HIPCHK(h, hipSetDevice(device));
  const int N = num_envs;
  // ---- constants + derived tables
  SdxConst& K = h->h_const;
  sdx_build_const(scene, &K);
  int rc;
  if ((rc = dalloc(h, &h->d_const, 1)) != SDX_OK) { g_create_err = h->err; delete h; return rc; }
  HIPCHK(h, hipMemcpy(h->d_const, &K, sizeof(K), hipMemcpyHostToDevice));

  // ---- buffers
  SdxBuf& B = h->buf;
  memset(&B, 0, sizeof(B));
  B.N = N;
  B.task_kind = scene->task_kind;
  B.orient_gate = scene->orient_tvalue_gate;
  B.obs_w = (scene->task_kind == 1 || scene->task_kind == 3) ? 186 : (scene->task_kind == 2 ? 75 : SDX_NUM_OBS);
  B.K = 1;
  B.seed = seed;
#define ALLOC(field, count) if ((rc = dalloc(h, &B.field, (size_t)(count))) != SDX_OK) { g_create_err = h->err; sdx_destroy(h); return rc; }
  ALLOC(root, (size_t)N * SDX_ACTORS * 13);
  ALLOC(dof, (size_t)N * SDX_NDOF * 2);
  ALLOC(rb, (size_t)N * SDX_BODIES * 13);
  ALLOC(contact, (size_t)N * SDX_BODIES * 3);
  ALLOC(jac, (size_t)N * 42);
  ALLOC(jac_full, (size_t)N * (SDX_NLINK - 1) * 6 * SDX_NDOF);
  ALLOC(targets, (size_t)N * SDX_NDOF);
  ALLOC(prev_targets, (size_t)N * SDX_NDOF);
  ALLOC(obs, (size_t)N * SDX_NUM_OBS);
  ALLOC(states, (size_t)N * SDX_NUM_STATES);
  ALLOC(obs_c, (size_t)N * SDX_NUM_OBS);
  ALLOC(states_c, (size_t)N * SDX_NUM_STATES);
  ALLOC(rew, N);
  ALLOC(reset, N);
  ALLOC(progress, N);
  ALLOC(randomize, N);
  ALLOC(actions, (size_t)N * SDX_NDOF);
  ALLOC(init_pos, (size_t)N * 3);
  ALLOC(init_rot, (size_t)N * 4);
  ALLOC(successes, N);
  ALLOC(meta_rew, N);
  ALLOC(cons, 1);
  ALLOC(finger_dist, N);
  ALLOC(tvalue, N);
  ALLOC(arm_contacts, (size_t)N * 6);
  ALLOC(student_obs, (size_t)N * 30);
  ALLOC(success_buf, N);
  ALLOC(pile_choice, N);
  ALLOC(ncontacts, N);
  ALLOC(piles, (size_t)8 * SDX_NBRICK * 13);
  ALLOC(tv_w, SDX_TV_PARAMS);
  ALLOC(cam_rot, (size_t)N * 4);
  ALLOC(cscratch, 4);   // (contact rows now live in LDS / registers)
  ALLOC(stat, 4);
  ALLOC(step_count, 1);
  ALLOC(dbg, 64 + 2 * (size_t)N);
  { const char* de = getenv("SDX_DEBUG_ENV"); B.dbg_env = de ? atoi(de) : 0; }
  ALLOC(harvest_hand, (size_t)8 * SDX_HARVEST_SLOTS * SDX_NDOF * 2);
  ALLOC(harvest_obj, (size_t)8 * SDX_HARVEST_SLOTS * 13);
  ALLOC(harvest_count, 8);
  ALLOC(insert_aux, (size_t)N * 8);
  ALLOC(tv_succ, (size_t)SDX_TV_LOG_SLOTS * 4);
  ALLOC(tv_fail, (size_t)SDX_TV_LOG_SLOTS * 4);
  ALLOC(tv_count, 2);
  ALLOC(tv_key, (size_t)2 * SDX_TV_LOG_SLOTS);
  ALLOC(harvest_key, (size_t)8 * SDX_HARVEST_SLOTS);
  B.pile_slots = (scene->task_kind == 1 || scene->task_kind == 3) ? SDX_PILE_HARVEST_SLOTS : 1;   // Orient and Search harvest piles
  if (B.pile_slots > 1) {   // SDX_PILE_SLOTS=10000: the reference's ring length (OR:1485; 549 MB of the 288 GB); default SDX_PILE_HARVEST_SLOTS
    const char* ps = getenv("SDX_PILE_SLOTS");
    const long v = ps ? atol(ps) : 0;
    if (v >= 16 && v <= 10000) B.pile_slots = (int32_t)v;
  }
  ALLOC(pile_harvest, (size_t)8 * B.pile_slots * SDX_NBRICK * 13);
  ALLOC(pile_harvest_count, 8);
  ALLOC(pile_key, (size_t)8 * B.pile_slots);
  ALLOC(seg_stats, (size_t)N * 4);
  ALLOC(seg_image, scene->task_kind == 3 ? (size_t)N * 128 * 128 : 1);
  ALLOC(seg_pix, (size_t)N * 4);
  ALLOC(emergence, N);
  ALLOC(cstats, 4);
  ALLOC(order, N);
  ALLOC(cost, N);
  { std::vector<int32_t> iota((size_t)N); for (int i = 0; i < N; ++i) iota[i] = i; HIPCHK(h, hipMemcpy(B.order, iota.data(), (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice)); }
  ALLOC(wcount, N);
  // the solver's impulse cache (24.5 KB per env) only exists when the scene asks for the warm start; k_physics<.., false> never reads it
  ALLOC(wkey, scene->warm_start > 0.0f ? (size_t)N * SDX_MAXC : 1);
  ALLOC(wlam, scene->warm_start > 0.0f ? (size_t)N * 3 * SDX_MAXC : 1);
  if (scene->task_kind == 3) {
    ALLOC(tvt_buf, (size_t)N * 652);
    ALLOC(tvt_w, (size_t)1024 * 652 + 1024 + 512 * 1024 + 512 + 128 * 512 + 128 + 2 * 128 + 2);
    ALLOC(tvt_h, (size_t)N * (1024 + 512 + 128 + 4));
  }
#undef ALLOC
  set_tensor(h, SDX_T_ROOT, B.root, SDX_F32, {(int64_t)N * SDX_ACTORS, 13});
  set_tensor(h, SDX_T_DOF, B.dof, SDX_F32, {(int64_t)N * SDX_NDOF, 2});
  set_tensor(h, SDX_T_RB, B.rb, SDX_F32, {N, SDX_BODIES, 13});
  set_tensor(h, SDX_T_CONTACT, B.contact, SDX_F32, {N, SDX_BODIES * 3});
  set_tensor(h, SDX_T_JAC_EEF, B.jac, SDX_F32, {N, 6, 7});
  set_tensor(h, SDX_T_TARGETS, B.targets, SDX_F32, {N, SDX_NDOF});
  set_tensor(h, SDX_T_PREV_TARGETS, B.prev_targets, SDX_F32, {N, SDX_NDOF});
  set_tensor(h, SDX_T_OBS, B.obs, SDX_F32, {N, B.obs_w});
  set_tensor(h, SDX_T_STATES, B.states, SDX_F32, {N, SDX_NUM_STATES});
  set_tensor(h, SDX_T_OBS_CLAMPED, B.obs_c, SDX_F32, {N, B.obs_w});
  set_tensor(h, SDX_T_STATES_CLAMPED, B.states_c, SDX_F32, {N, SDX_NUM_STATES});
  set_tensor(h, SDX_T_REW, B.rew, SDX_F32, {N});
  set_tensor(h, SDX_T_RESET, B.reset, SDX_I64, {N});
  set_tensor(h, SDX_T_PROGRESS, B.progress, SDX_I64, {N});
  set_tensor(h, SDX_T_RANDOMIZE, B.randomize, SDX_I64, {N});
  set_tensor(h, SDX_T_ACTIONS, B.actions, SDX_F32, {N, SDX_NDOF});
  set_tensor(h, SDX_T_INIT_POS, B.init_pos, SDX_F32, {N, 3});
  set_tensor(h, SDX_T_INIT_ROT, B.init_rot, SDX_F32, {N, 4});
  set_tensor(h, SDX_T_SUCCESSES, B.successes, SDX_F32, {N});
  set_tensor(h, SDX_T_META_REW, B.meta_rew, SDX_F32, {N});
  set_tensor(h, SDX_T_CONS_SUCCESSES, B.cons, SDX_F32, {1});
  set_tensor(h, SDX_T_FINGER_DIST, B.finger_dist, SDX_F32, {N});
  set_tensor(h, SDX_T_TVALUE, B.tvalue, SDX_F32, {N});
  set_tensor(h, SDX_T_ARM_CONTACTS, B.arm_contacts, SDX_F32, {N, 6});
  set_tensor(h, SDX_T_STUDENT_OBS, B.student_obs, SDX_F32, {N, 30});
  set_tensor(h, SDX_T_SUCCESS_BUF, B.success_buf, SDX_I64, {N});
  set_tensor(h, SDX_T_PILE_CHOICE, B.pile_choice, SDX_I32, {N});
  set_tensor(h, SDX_T_NCONTACTS, B.ncontacts, SDX_I32, {N});
  set_tensor(h, SDX_T_DEBUG, B.dbg, SDX_I64, {64 + 2 * (int64_t)N});
  set_tensor(h, SDX_T_HARVEST_HAND, B.harvest_hand, SDX_F32, {8, SDX_HARVEST_SLOTS, SDX_NDOF, 2});
  set_tensor(h, SDX_T_HARVEST_OBJ, B.harvest_obj, SDX_F32, {8, SDX_HARVEST_SLOTS, 13});
  set_tensor(h, SDX_T_HARVEST_COUNT, B.harvest_count, SDX_I32, {8});
  set_tensor(h, SDX_T_INSERT_AUX, B.insert_aux, SDX_F32, {N, 8});
  set_tensor(h, SDX_T_TV_SUCCESS, B.tv_succ, SDX_F32, {SDX_TV_LOG_SLOTS, 4});
  set_tensor(h, SDX_T_TV_FAILURE, B.tv_fail, SDX_F32, {SDX_TV_LOG_SLOTS, 4});
  set_tensor(h, SDX_T_TV_COUNT, B.tv_count, SDX_I32, {2});
  set_tensor(h, SDX_T_PILE_HARVEST, B.pile_harvest, SDX_F32, {8, B.pile_slots, SDX_NBRICK, 13});
  set_tensor(h, SDX_T_PILE_HARVEST_COUNT, B.pile_harvest_count, SDX_I32, {8});
  set_tensor(h, SDX_T_TV_KEYS, B.tv_key, SDX_I64, {2, SDX_TV_LOG_SLOTS});
  set_tensor(h, SDX_T_HARVEST_KEYS, B.harvest_key, SDX_I64, {8, SDX_HARVEST_SLOTS});
  set_tensor(h, SDX_T_PILE_HARVEST_KEYS, B.pile_key, SDX_I64, {8, B.pile_slots});
  if (scene->task_kind == 3) set_tensor(h, SDX_T_SEG_IMAGE, B.seg_image, SDX_I16, {N, 128, 128});
  else set_tensor(h, SDX_T_SEG_IMAGE, B.seg_image, SDX_I16, {1, 1, 1});   // placeholder: the camera belongs to Search
  set_tensor(h, SDX_T_SEG_PIXELS, B.seg_pix, SDX_F32, {N, 4});
  set_tensor(h, SDX_T_EMERGENCE, B.emergence, SDX_F32, {N});
  set_tensor(h, SDX_T_CONTACT_STATS, B.cstats, SDX_I32, {4});
  set_tensor(h, SDX_T_WARM_COUNT, B.wcount, SDX_I32, {N});
  if (scene->warm_start > 0.0f) {
    set_tensor(h, SDX_T_WARM_KEYS, B.wkey, SDX_I32, {N, SDX_MAXC});
    set_tensor(h, SDX_T_WARM_LAMBDA, B.wlam, SDX_F32, {N, 3, SDX_MAXC});
  } else {
    set_tensor(h, SDX_T_WARM_KEYS, B.wkey, SDX_I32, {1});
    set_tensor(h, SDX_T_WARM_LAMBDA, B.wlam, SDX_F32, {1});
  }
  set_tensor(h, SDX_T_CAM_ROT, B.cam_rot, SDX_F32, {N, 4});
  set_tensor(h, SDX_T_JACOBIAN, B.jac_full, SDX_F32, {N, SDX_NLINK - 1, 6, SDX_NDOF});
  if (scene->task_kind == 3) set_tensor(h, SDX_T_TVALUE_OBS, B.tvt_buf, SDX_F32, {N, 652});
  else set_tensor(h, SDX_T_TVALUE_OBS, B.seg_pix, SDX_F32, {1, 1});   // placeholder: the temporal buffer belongs to Search

  // ---- initial actor states (what create_actor's start poses give, GS:897-1000)
  std::vector<float> root((size_t)N * SDX_ACTORS * 13, 0.0f), rbv((size_t)N * SDX_BODIES * 13, 0.0f);
  std::vector<float> pile0((size_t)SDX_NBRICK * 13, 0.0f);
  for (int i = 0; i < SDX_NBRICK; ++i) {
    float* s = &pile0[(size_t)i * 13];
    if (i < SDX_NFREE) {
      memcpy(s, scene->free_spawn_pos[i], 12);
      memcpy(s + 3, scene->free_spawn_quat, 16);
    } else {
      memcpy(s, scene->fixed_brick_pos[i - SDX_NFREE], 12);
      s[6] = 1.0f;
    }
  }
  for (int e = 0; e < N; ++e) {
    float* r = &root[(size_t)e * SDX_ACTORS * 13];
    for (int a = 0; a < SDX_ACTORS; ++a) r[a * 13 + 6] = 1.0f;
    memcpy(r, scene->base_pos, 12);
    memcpy(r + 3, scene->base_quat, 16);
    memcpy(r + 13, scene->object_init_state, 13 * 4);
    memcpy(r + 26, scene->goal_reset_pos, 12);
    for (int s = 0; s < 6; ++s) memcpy(r + (3 + s) * 13, scene->static_actor_pos[s], 12);
    memcpy(r + SDX_ACTOR_BRICK0 * 13, pile0.data(), pile0.size() * 4);
    memcpy(r + 141 * 13, scene->base_plate_pos, 12);
    float* b = &rbv[(size_t)e * SDX_BODIES * 13];
    for (int a = 1; a < SDX_ACTORS; ++a) memcpy(b + (SDX_NLINK + a - 1) * 13, r + a * 13, 13 * 4);
  }
  HIPCHK(h, hipMemcpy(B.root, root.data(), root.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(B.rb, rbv.data(), rbv.size() * 4, hipMemcpyHostToDevice));
  {  // default saved piles: K=1, the spawn lattice for every type group (replaced by sdx_load_initial_states)
    std::vector<float> p8;
    for (int t = 0; t < 8; ++t) p8.insert(p8.end(), pile0.begin(), pile0.end());
    HIPCHK(h, hipMemcpy(B.piles, p8.data(), p8.size() * 4, hipMemcpyHostToDevice));
  }
  {
    std::vector<int64_t> ones(N, 1);  // reset_buf = ONES: every env resets on the first step (BT:63)
    HIPCHK(h, hipMemcpy(B.reset, ones.data(), (size_t)N * 8, hipMemcpyHostToDevice));
  }
  sdxk_kinematics(h->d_const, &h->buf, 0);  // first refresh (GS:243-246)
  HIPCHK(h, hipDeviceSynchronize());
  *out = h;
  return SDX_OK;
}

extern "C" int sdx_destroy(sdx_handle h) {
  if (!h) return SDX_ERR_INVALID;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return SDX_OK;
}

extern "C" int sdx_tensor(sdx_handle h, int32_t id, void** dev_ptr, int64_t shape[4], int32_t* ndim, int32_t* dtype) {
  if (!h) return SDX_ERR_INVALID;
  if (id < 0 || id >= SDX_T_COUNT || !dev_ptr || !shape || !ndim || !dtype) { h->err = "sdx_tensor: bad argument"; return SDX_ERR_INVALID; }
  const auto& t = h->tinfo[id];
  *dev_ptr = t.ptr;
  for (int i = 0; i < 4; ++i) shape[i] = t.shape[i];
  *ndim = t.ndim;
  *dtype = t.dtype;
  return SDX_OK;
}

extern "C" int sdx_load_initial_states(sdx_handle h, const float* piles_host, int32_t K) {
  if (!h) return SDX_ERR_INVALID;
  if (!piles_host || K <= 0) { h->err = "sdx_load_initial_states: bad argument"; return SDX_ERR_INVALID; }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipDeviceSynchronize());
  float* p = nullptr;
  const size_t count = (size_t)8 * K * SDX_NBRICK * 13;
  int rc = dalloc(h, &p, count);
  if (rc != SDX_OK) return rc;
  HIPCHK(h, hipMemcpy(p, piles_host, count * 4, hipMemcpyHostToDevice));
  h->buf.piles = p;  // the previous table stays allocated until destroy (cheap, avoids a free under a live stream)
  h->buf.K = K;
  h->has_piles = true;
  return SDX_OK;
}

extern "C" int sdx_set_tvalue_weights(sdx_handle h, const float* w, int32_t n) {
  if (!h) return SDX_ERR_INVALID;
  if (!w || n != SDX_TV_PARAMS) { h->err = "sdx_set_tvalue_weights: expected SDX_TV_PARAMS floats"; return SDX_ERR_INVALID; }
  // host layout (torch): W[out][in] then b; device layout: W^T[in][out] then b
  std::vector<float> t(SDX_TV_PARAMS);
  const int dims[5] = {4, 256, 128, 64, 2};
  size_t o = 0;
  for (int l = 0; l < 4; ++l) {
    const int in = dims[l], out = dims[l + 1];
    for (int i = 0; i < in; ++i)
      for (int j = 0; j < out; ++j) t[o + (size_t)i * out + j] = w[o + (size_t)j * in + i];
    o += (size_t)in * out;
    for (int j = 0; j < out; ++j) t[o + j] = w[o + j];
    o += out;
  }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy(h->buf.tv_w, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  return SDX_OK;
}

extern "C" int sdx_set_retri_tvalue_weights(sdx_handle h, const float* w, int32_t n) {
  if (!h) return SDX_ERR_INVALID;
  if (h->h_const.sc.task_kind != 3) { h->err = "sdx_set_retri_tvalue_weights: RetriGraspTValue belongs to BlockAssemblySearch (task_kind 3)"; return SDX_ERR_STATE; }
  if (!w || n != SDX_RETRI_TV_PARAMS) { h->err = "sdx_set_retri_tvalue_weights: expected SDX_RETRI_TV_PARAMS floats"; return SDX_ERR_INVALID; }
  // device layout = host layout with the rows of W1 padded from 650 to 652 columns (16-byte rows for the float4 loads of the GEMM)
  std::vector<float> t((size_t)1024 * 652 + (n - (size_t)1024 * 650), 0.0f);
  for (int r = 0; r < 1024; ++r) memcpy(&t[(size_t)r * 652], w + (size_t)r * 650, 650 * sizeof(float));
  memcpy(&t[(size_t)1024 * 652], w + (size_t)1024 * 650, (n - (size_t)1024 * 650) * sizeof(float));
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(h->buf.tvt_w, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  return SDX_OK;
}

static int check_launch(sdx_handle h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { h->err = std::string(what) + ": " + hipGetErrorString(e); return SDX_ERR_HIP; }
  return SDX_OK;
}

extern "C" int sdx_pre_physics(sdx_handle h, const float* actions_dev, void* stream) {
  if (!h || !actions_dev) return SDX_ERR_INVALID;
  sdxk_pre_physics(h->d_const, &h->buf, actions_dev, nullptr, nullptr, 1 | 4, (hipStream_t)stream);
  return check_launch(h, "sdx_pre_physics");
}
extern "C" int sdx_simulate(sdx_handle h, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  sdxk_physics(h->d_const, &h->buf, (hipStream_t)stream);
  return check_launch(h, "sdx_simulate");
}
extern "C" int sdx_post_physics(sdx_handle h, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  sdxk_post_physics(h->d_const, &h->buf, 1, (hipStream_t)stream);
  return check_launch(h, "sdx_post_physics");
}
extern "C" int sdx_compute_observations(sdx_handle h, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  sdxk_post_physics(h->d_const, &h->buf, 0, (hipStream_t)stream);
  return check_launch(h, "sdx_compute_observations");
}
// BlockAssemblyOrient reset (OR:1390-1695).  Like the reference (reset_buf.nonzero()) this reads the reset flags on the host; a reset
// event costs 50 (only when steps have been taken) + 2 + 1 + 50 simulator steps of ALL envs, during which the resetting envs' arms
// are scripted by the tracking IK.  The shipped task only resets on time-out, so all envs reset together every episodeLength steps.
static int orient_reset_if_needed(sdx_handle h, hipStream_t st) {
  const int N = h->buf.N;
  std::vector<int64_t> flags(N);
  HIPCHK(h, hipMemcpyAsync(flags.data(), h->buf.reset, sizeof(int64_t) * N, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  std::vector<uint8_t> mask(N);
  int any = 0;
  for (int i = 0; i < N; ++i) { mask[i] = flags[i] != 0; any |= mask[i]; }
  if (!any) return SDX_OK;
  if (!h->orient_mask) { HIPCHK(h, hipMalloc((void**)&h->orient_mask, N)); h->allocs.push_back(h->orient_mask); }
  HIPCHK(h, hipMemcpyAsync(h->orient_mask, mask.data(), N, hipMemcpyHostToDevice, st));
  uint32_t steps = 0;
  HIPCHK(h, hipMemcpyAsync(&steps, h->buf.step_count, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (steps > 0)
    for (int i = 0; i < 50; ++i) { sdxk_orient_pregrasp(h->d_const, &h->buf, h->orient_mask, 0, i, st); sdxk_physics(h->d_const, &h->buf, st); }
  if (steps > 0) sdxk_post_physics(h->d_const, &h->buf, 0, st);                        // self.compute_observations() before the harvest, OR:1464-1465
  sdxk_pre_physics(h->d_const, &h->buf, nullptr, h->orient_mask, nullptr, 2, st);      // harvest, restore piles, hand to the prepare pose, counters
  sdxk_physics(h->d_const, &h->buf, st);
  sdxk_physics(h->d_const, &h->buf, st);                                               // OR:1618-1620
  sdxk_orient_post_reset(h->d_const, &h->buf, h->orient_mask, st);
  sdxk_physics(h->d_const, &h->buf, st);                                               // OR:1653
  for (int i = 0; i < 50; ++i) { sdxk_orient_pregrasp(h->d_const, &h->buf, h->orient_mask, 1, i, st); sdxk_physics(h->d_const, &h->buf, st); }
  return check_launch(h, "sdx_step(orient reset)");
}

extern "C" void sdxk_search_set_hand(const SdxConst*, const SdxBuf*, const uint8_t*, int, hipStream_t);

// BlockAssemblySearch: reset_idx + post_reset (SE:1274-1538) for the envs whose reset flag is set (read on the host, as
// reset_buf.nonzero() does): outcome / harvest / lattice restore on the device, 60 settling steps of all envs, a segmentation render
// (emergence bookkeeping), hand to the prepare pose.  *progress0 = progress_buf[0] before this step (the end-of-episode render keys
// on it, SE:992).
static int search_reset_if_needed(sdx_handle h, hipStream_t st, int64_t* progress0) {
  const int N = h->buf.N;
  std::vector<int64_t> flags(N);
  HIPCHK(h, hipMemcpyAsync(flags.data(), h->buf.reset, sizeof(int64_t) * N, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(progress0, h->buf.progress, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  std::vector<uint8_t> mask(N);
  int any = 0;
  for (int i = 0; i < N; ++i) { mask[i] = flags[i] != 0; any |= mask[i]; }
  if (!any) return SDX_OK;
  if (!h->orient_mask) { HIPCHK(h, hipMalloc((void**)&h->orient_mask, N)); h->allocs.push_back(h->orient_mask); }
  HIPCHK(h, hipMemcpyAsync(h->orient_mask, mask.data(), N, hipMemcpyHostToDevice, st));
  sdxk_pre_physics(h->d_const, &h->buf, nullptr, h->orient_mask, nullptr, 2, st);      // outcome, harvest, lattice + noise, target drop, default pose
  for (int i = 0; i < 60; ++i) sdxk_physics(h->d_const, &h->buf, st);                  // SE:1437-1439
  sdxk_seg_camera(h->d_const, &h->buf, st);                                            // SE:1444-1455
  sdxk_search_set_hand(h->d_const, &h->buf, h->orient_mask, 0, st);                    // SE:1482-1495
  if (mask[0]) *progress0 = 0;
  return check_launch(h, "sdx_step(search reset)");
}

extern "C" int sdx_step(sdx_handle h, const float* actions_dev, void* stream) {
  if (!h || !actions_dev) return SDX_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (h->h_const.sc.task_kind == 3) {
    int64_t p0 = 0;
    const int rc = search_reset_if_needed(h, st, &p0);
    if (rc != SDX_OK) return rc;
    sdxk_pre_physics(h->d_const, &h->buf, actions_dev, nullptr, nullptr, 4, st);
    sdxk_physics(h->d_const, &h->buf, st);
    if ((float)(p0 + 1) >= h->h_const.sc.max_episode_length - 1.0f) {                  // SE:992-1019: park the hand, one more step, render
      sdxk_search_set_hand(h->d_const, &h->buf, nullptr, 1, st);
      sdxk_physics(h->d_const, &h->buf, st);
      sdxk_seg_camera(h->d_const, &h->buf, st);
    }
    sdxk_post_physics(h->d_const, &h->buf, 1, st);
    return check_launch(h, "sdx_step");
  }
  if (h->h_const.sc.task_kind == 1) {
    const int rc = orient_reset_if_needed(h, st);
    if (rc != SDX_OK) return rc;
    sdxk_pre_physics(h->d_const, &h->buf, actions_dev, nullptr, nullptr, 4, st);
    sdxk_physics(h->d_const, &h->buf, st);
    sdxk_post_physics(h->d_const, &h->buf, 1, st);
    return check_launch(h, "sdx_step");
  }
  sdxk_pre_physics(h->d_const, &h->buf, actions_dev, nullptr, nullptr, 1 | 4, st);
  sdxk_physics(h->d_const, &h->buf, st);   // controlFrequencyInv = 1 (EG:18)
  sdxk_post_physics(h->d_const, &h->buf, 1, st);
  return check_launch(h, "sdx_step");
}
// Not part of include/seqdex.h: the scripted stand-in for a trained grasp policy that the chain benchmark and its tests drive the grasp
// stage with (seqdex_amd/scripts/evaluation.py), as one launch instead of ~40 torch operations per env step.
extern "C" void sdxk_scripted_grasp(const SdxConst*, const SdxBuf*, float*, float*, hipStream_t);
extern "C" int sdxk_scripted_grasp_actions(sdx_handle h, float* close_state_dev, float* actions_out_dev, void* stream) {
  if (!h || !close_state_dev || !actions_out_dev) return SDX_ERR_INVALID;
  sdxk_scripted_grasp(h->d_const, &h->buf, close_state_dev, actions_out_dev, (hipStream_t)stream);
  return check_launch(h, "sdxk_scripted_grasp_actions");
}
extern "C" int sdx_reset_idx(sdx_handle h, const uint8_t* env_mask_dev, const int32_t* pile_choice_dev, void* stream) {
  if (!h || !env_mask_dev) return SDX_ERR_INVALID;
  sdxk_pre_physics(h->d_const, &h->buf, nullptr, env_mask_dev, pile_choice_dev, 2, (hipStream_t)stream);
  return check_launch(h, "sdx_reset_idx");
}
extern "C" int sdx_render_segmentation(sdx_handle h, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  if (h->h_const.sc.task_kind != 3) { h->err = "sdx_render_segmentation: the segmentation camera belongs to BlockAssemblySearch (task_kind 3)"; return SDX_ERR_STATE; }
  sdxk_seg_camera(h->d_const, &h->buf, (hipStream_t)stream);
  return check_launch(h, "sdx_render_segmentation");
}
extern "C" int sdx_refresh_kinematics(sdx_handle h, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  sdxk_kinematics(h->d_const, &h->buf, (hipStream_t)stream);
  return check_launch(h, "sdx_refresh_kinematics");
}
// one thread per (listed actor, column)
// src may BE the library's own tensor (the documented use: edit SDX_T_ROOT / SDX_T_DOF in place, then name the rows that changed): no __restrict__ on it
__global__ void k_set_indexed(SdxBuf B, int id, const float* src, const int32_t* __restrict__ ids, int n) {
  const int i = blockIdx.x, t = threadIdx.x;
  if (i >= n) return;
  const int actor = ids[i];
  if (actor < 0 || actor >= B.N * SDX_ACTORS) return;
  const int e = actor / SDX_ACTORS, slot = actor % SDX_ACTORS;
  if (id == SDX_T_ROOT) {
    if (slot == 0 || t >= 13) return;                       // the hand actor has a fixed base: its root state is not settable
    const float v = src[(size_t)actor * 13 + t];
    B.root[(size_t)actor * 13 + t] = v;
    B.rb[((size_t)e * SDX_BODIES + SDX_NLINK + slot - 1) * 13 + t] = v;
    if (t == 0) B.wcount[e] = 0;                            // a teleported body invalidates the env's cached contact impulses (warm start)
  } else if (slot == 0) {
    if (id == SDX_T_DOF) { if (t < SDX_NDOF * 2) B.dof[(size_t)e * SDX_NDOF * 2 + t] = src[(size_t)e * SDX_NDOF * 2 + t]; }
    else if (t < SDX_NDOF) B.targets[(size_t)e * SDX_NDOF + t] = src[(size_t)e * SDX_NDOF + t];
  }
}
extern "C" int sdx_set_indexed(sdx_handle h, int32_t id, const float* src_dev, const int32_t* actor_ids_dev, int32_t n, void* stream) {
  if (!h) return SDX_ERR_INVALID;
  if (!src_dev || !actor_ids_dev || n < 0 || (id != SDX_T_ROOT && id != SDX_T_DOF && id != SDX_T_TARGETS)) {
    h->err = "sdx_set_indexed: id must be SDX_T_ROOT, SDX_T_DOF or SDX_T_TARGETS, src / ids non-NULL"; return SDX_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  if (n > 0) hipLaunchKernelGGL(k_set_indexed, dim3(n), dim3(64), 0, st, h->buf, id, src_dev, actor_ids_dev, n);
  if (id == SDX_T_DOF && n > 0) sdxk_kinematics(h->d_const, &h->buf, st);
  return check_launch(h, "sdx_set_indexed");
}
extern "C" int sdx_num_envs(sdx_handle h) { return h ? h->buf.N : SDX_ERR_INVALID; }
extern "C" const char* sdx_last_error(sdx_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }
