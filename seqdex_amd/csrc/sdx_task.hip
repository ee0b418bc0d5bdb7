// sdx_task.hip — the reference-owned per-step tensor code of BlockAssemblyGraspSim as HIP kernels,
// one wavefront (64 lanes) per env.  SURVEY.md §8(a) rows T2-T9; reference GS = tasks/block_assembly/
// allegro_hand_block_assembly_grasp_sim.py, VR = tasks/hand_base/vec_task_rlgames.py.
//
//   k_pre_physics   GS:1555-1638 pre_physics_step: device-side masked reset_idx (GS:1361-1553, no
//                   reset_buf.nonzero() host sync), action -> joint targets incl. the 6x6 damped-least-
//                   squares IK solve (control_ik GS:1796-1804)
//   k_post_physics  GS:1640-1645 post_physics_step: progress++, compute_observations (GS:1090-1218,
//                   1299-1332, 1220-1280) with 3-frame stacking, compute_hand_reward (GS:1706-1776),
//                   +-5 clamped copies (VR:171-172)
//   k_tvalue        GraspInsertTValue MLP 4-256-128-64-2, ELU on every layer (terminal_value_function.py:30-46),
//                   sigmoid(.)[:,1] (GS:1200-1201), batched over envs
//
// HBM-bound by design: every per-env row is loaded once with lane-strided (coalesced) accesses into LDS,
// derived quantities are computed once per wave, rows are written back lane-strided.
#include "sdx_common.h"

// orientation_error(quat_from_euler_xyz(euler), current) (OR:1922-1925; quat_from_euler_xyz of isaacgym.torch_utils: half-angle
// products): vector part of desired * conj(current), flipped into the w >= 0 hemisphere
__device__ __forceinline__ f3 wrist_error(const float* euler, f4 current) {
  float sr, cr, sp, cp, sy, cy;
  sincosf(0.5f * euler[0], &sr, &cr);
  sincosf(0.5f * euler[1], &sp, &cp);
  sincosf(0.5f * euler[2], &sy, &cy);
  f4 qd;
  qd.x = cy * sr * cp - sy * cr * sp; qd.y = cy * cr * sp + sy * sr * cp; qd.z = sy * cr * cp - cy * sr * sp;
  qd.w = cy * cr * cp + sy * sr * sp;
  const f4 qr = qmul(qd, qconj(current));
  const float sg = qr.w > 0.0f ? 1.0f : (qr.w < 0.0f ? -1.0f : 0.0f);             // torch.sign
  return F3(qr.x * sg, qr.y * sg, qr.z * sg);
}
// control_ik (GS:1796-1804 / OR:1927-1935): y = (J J^T + 0.05^2 I)^-1 dpose by a 6x6 Cholesky solve; the caller forms u = J^T y.
// J: [6][7] row-major in LDS; every lane computes the same y (wave-uniform).
__device__ __forceinline__ void control_ik_solve(const float* J, const float* dp, float* y) {
  float A[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      float s = (r == c) ? 0.05f * 0.05f : 0.0f;
#pragma unroll
      for (int k = 0; k < 7; ++k) s += J[r * 7 + k] * J[c * 7 + k];
      A[r][c] = s;
    }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
    d = sqrtf(d);
    A[j][j] = d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float s = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= A[i][k] * A[j][k];
      A[i][j] = s / d;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s = dp[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= A[i][k] * y[k];
    y[i] = s / A[i][i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float s = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= A[k][i] * y[k];
    y[i] = s / A[i][i];
  }
}

// T-value datasets (the reference's HDF5 groups data/success_dataset, data/failure_dataset, GS:470-480): the camera-frame
// quaternion of the target brick (camera_view_segmentation_target_rot of the last compute_observations) of a finished episode goes to
// the success or the failure ring.  Wave-uniform call; lane 0 claims the slot.
__device__ __forceinline__ unsigned long long ring_key(const SdxBuf& B, int e) { return ((unsigned long long)B.step_count[0] << 24) | (unsigned)e; }
__device__ __forceinline__ void tv_log(const SdxBuf& B, int e, int lane, bool success) {
  int slot = 0;
  if (lane == 0) slot = atomicAdd(&B.tv_count[success ? 0 : 1], 1) % SDX_TV_LOG_SLOTS;
  slot = __shfl(slot, 0, SDX_WAVE);
  float* dst = (success ? B.tv_succ : B.tv_fail) + (size_t)slot * 4;
  if (lane < 4) dst[lane] = B.cam_rot[(size_t)e * 4 + lane];
  if (lane == 4) B.tv_key[(size_t)(success ? 0 : 1) * SDX_TV_LOG_SLOTS + slot] = ring_key(B, e);
}

// ------------------------------------------------------------------------------------------------ K1 + K2
// flags: bit0 reset envs with reset_buf != 0; bit1 reset envs with ext_mask != 0; bit2 compute targets
__global__ __launch_bounds__(SDX_WAVE) void k_pre_physics(const SdxConst* __restrict__ C, SdxBuf B,
                                                          const float* __restrict__ actions_in,
                                                          const uint8_t* __restrict__ ext_mask,
                                                          const int32_t* __restrict__ ext_choice, int flags) {
  const int e = blockIdx.x, lane = threadIdx.x;
  const sdx_scene_desc& sc = C->sc;
  __shared__ float s_J[42];

  bool do_reset = false;
  if (flags & 1) do_reset = B.reset[e] != 0;
  if (flags & 2) do_reset = do_reset || (ext_mask[e] != 0);
  float* root_e = B.root + (size_t)e * SDX_ACTORS * 13;

  if (do_reset) {  // wave-uniform branch
    // terminal-state harvesting (GS:1398-1442): a finished episode whose target brick was carried over the base plate
    // (y < 0) with the fingers still around it and an accepting T-value is stored in the ring buffer of its type group
    if (B.step_count[0] > 0 && sc.task_kind == 0) {                                // `if self.total_steps > 0`; GraspSim's rule only
      const float* tg = root_e + seg_actor(e) * 13;
      const bool good = tg[1] < 0.0f && B.finger_dist[e] < 0.6f && B.tvalue[e] > sc.grasp_tvalue_gate;   // GS:1404-1406 (0.8)
      tv_log(B, e, lane, good);                                                    // the save_hdf5 datasets, GS:1407-1438
      if (good) {
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&B.harvest_count[e & 7], 1) % SDX_HARVEST_SLOTS;   // GS:1417,1440-1441
        slot = __shfl(slot, 0, SDX_WAVE);
        const size_t o = (size_t)(e & 7) * SDX_HARVEST_SLOTS + slot;
        if (lane < 46) B.harvest_hand[o * 46 + lane] = B.dof[(size_t)e * 46 + lane];      // GS:1415
        if (lane < 13) B.harvest_obj[o * 13 + lane] = tg[lane];                            // GS:1416
        if (lane == 63) B.harvest_key[o] = ring_key(B, e);
      }
    }
    // BlockAssemblyOrient, OR:1463-1488: an episode that ends with the hand withdrawn (finger distance > 0.3), the target brick still
    // in the bin half (0 < y < 0.5) and an accepting T-value hands its WHOLE brick pile on: these are the saved piles that
    // BlockAssemblyGraspSim starts its episodes from (GS:412-413,1507-1513); the camera-frame quaternion goes to the T-value datasets
    if (B.step_count[0] > 0 && sc.task_kind == 1) {
      const float* tg = root_e + seg_actor(e) * 13;
      const bool good = B.finger_dist[e] > 0.3f && tg[1] > 0.0f && tg[1] < 0.5f && B.tvalue[e] > 0.6f;   // OR:1468-1470
      tv_log(B, e, lane, good);
      if (good) {
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&B.pile_harvest_count[e & 7], 1) % B.pile_slots;          // OR:1483-1486
        slot = __shfl(slot, 0, SDX_WAVE);
        float* dst = B.pile_harvest + ((size_t)(e & 7) * B.pile_slots + slot) * SDX_NBRICK * 13;
        const float* srcb = root_e + SDX_ACTOR_BRICK0 * 13;
        for (int i = lane; i < SDX_NBRICK * 13; i += SDX_WAVE) dst[i] = srcb[i];
        if (lane == 0) B.pile_key[(size_t)(e & 7) * B.pile_slots + slot] = ring_key(B, e);
      }
    }
    // BlockAssemblySearch, SE:1289,1306-1343: the episode succeeded when enough pixels of the target brick are visible to the fixed
    // camera (threshold by brick type); successes hand their whole pile on (the saved piles BlockAssemblyOrient starts from)
    if (B.step_count[0] > 0 && sc.task_kind == 3) {
      const int thr[8] = {20, 20, 15, 20, 20, 30, 30, 20};
      const bool good = B.seg_pix[(size_t)e * 4] > (float)thr[e & 7];
      tv_log(B, e, lane, good);
      if (lane == 0) B.success_buf[e] = good ? 1 : 0;
      if (good) {
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&B.pile_harvest_count[e & 7], 1) % B.pile_slots;          // SE:1323-1328
        slot = __shfl(slot, 0, SDX_WAVE);
        float* dst = B.pile_harvest + ((size_t)(e & 7) * B.pile_slots + slot) * SDX_NBRICK * 13;
        const float* srcb = root_e + SDX_ACTOR_BRICK0 * 13;
        for (int i = lane; i < SDX_NBRICK * 13; i += SDX_WAVE) dst[i] = srcb[i];
        if (lane == 0) B.pile_key[(size_t)(e & 7) * B.pile_slots + slot] = ring_key(B, e);
      }
    }
    __builtin_amdgcn_wave_barrier();   // the harvests above read rows that the restore below rewrites with another lane assignment
    int choice;
    if (ext_choice) choice = ext_choice[e];
    else choice = (int)(sdx_hash(B.seed, (uint64_t)e, (uint64_t)B.step_count[0]) % (uint64_t)B.K);
    // restore the 132 bricks from a saved pile state of this env's brick-type group (GS:1507-1513), zero velocities
    const float* src = B.piles + ((size_t)(e & 7) * B.K + choice) * SDX_NBRICK * 13;
    float* dst = root_e + SDX_ACTOR_BRICK0 * 13;
    for (int i = lane; i < SDX_NBRICK * 13; i += SDX_WAVE) {
      float v = src[i];
      dst[i] = (i % 13 >= 7) ? 0.0f : v;
    }
    if (lane == 0 && B.wcount) B.wcount[e] = 0;                                  // a restored pile has no contact history (warm start)
    if (lane < 13) root_e[1 * 13 + lane] = sc.object_init_state[lane];           // GS:1475-1482
    if (lane < 3) root_e[2 * 13 + lane] = sc.goal_reset_pos[lane];               // GS:1348
    if (lane >= 7 && lane < 13) root_e[2 * 13 + lane] = 0.0f;                    // GS:1350
    if (lane < SDX_NDOF) {
      float hp = C->hand_reset_pose[lane];
      B.dof[((size_t)e * SDX_NDOF + lane) * 2 + 0] = hp;                         // GS:1526,1531
      B.dof[((size_t)e * SDX_NDOF + lane) * 2 + 1] = 0.0f;                       // GS:1529
      B.prev_targets[(size_t)e * SDX_NDOF + lane] = hp;                          // GS:1527,1533
      B.targets[(size_t)e * SDX_NDOF + lane] = hp;                               // GS:1528,1535
    }
    const int seg = seg_actor(e) - SDX_ACTOR_BRICK0;
    if (lane < 3) B.init_pos[e * 3 + lane] = src[seg * 13 + lane];               // GS:1547
    if (lane < 4) B.init_rot[e * 4 + lane] = src[seg * 13 + 3 + lane];           // GS:1548
    if (sc.task_kind == 3) {
      // BlockAssemblySearch reset_idx, SE:1367-1421: bricks back on the spawn lattice (the saved "pile" of this task) with +-0.02 of
      // x / y noise on the free ones, the target brick dropped from z = 0.9 at a random spot over the bin, hand parked at its
      // default pose; the host then lets the pile settle for 60 steps (post_reset)
      __syncthreads();
      float* bricks = root_e + SDX_ACTOR_BRICK0 * 13;
      for (int i = lane; i < SDX_NFREE * 2; i += SDX_WAVE) {
        const uint64_t hsh = sdx_hash(B.seed ^ 0x5EA7ull, (uint64_t)e * 1024 + i, (uint64_t)B.step_count[0]);
        const float uni = (float)((hsh >> 40) & 0xFFFFFFull) * (2.0f / 16777216.0f) - 1.0f;
        bricks[(i >> 1) * 13 + (i & 1)] += uni * 0.02f;                               // SE:1395-1396
      }
      __syncthreads();
      const uint64_t hr = sdx_hash(B.seed ^ 0x7A96ull, (uint64_t)e, (uint64_t)B.step_count[0]);
      const float r = (float)((hr >> 40) & 0xFFFFFFull) * (2.0f / 16777216.0f) - 1.0f;   // ONE draw moves x and y together, SE:1399-1400
      float* tg = root_e + seg_actor(e) * 13;
      if (lane == 0) { tg[0] = 0.25f + r * 0.2f; tg[1] = 0.19f + r * 0.15f; tg[2] = 0.9f; }
      if (lane < SDX_NDOF) {
        const float qh = lane < 7 ? sc.search_default_arm[lane] : sc.search_finger_pose[lane - 7];   // SE:1416-1421
        B.dof[((size_t)e * SDX_NDOF + lane) * 2 + 0] = qh;
        B.dof[((size_t)e * SDX_NDOF + lane) * 2 + 1] = 0.0f;
        B.prev_targets[(size_t)e * SDX_NDOF + lane] = qh;
        B.targets[(size_t)e * SDX_NDOF + lane] = qh;
      }
    }
    if (sc.task_kind == 2) {
      // BlockAssemblyInsertSim reset_idx, IS:1328-1494.  Episode outcome first (IS:1345-1354, from the quantities of the last
      // compute_observations): inserted = within 2 cm and 0.2 rad of the site or of its 180-degree twin
      __syncthreads();                                                              // the pile / hand writes above land first
      float* aux = B.insert_aux + (size_t)e * 8;
      if (B.step_count[0] > 0) {
        const bool inserted = aux[3] < 0.02f && aux[4] < 0.2f;
        if (lane == 0) B.success_buf[e] = inserted ? 1 : 0;
        tv_log(B, e, lane, inserted);                                               // train_t_value datasets, IS:1392-1410
      }
      // base plate back to its place with a 0 / 90 degree yaw drawn ONCE per reset event (random.sample([0, 1], 1), IS:1435-1445)
      const float yaw_half = 0.785f * (float)(sdx_hash(B.seed, 0xA11CEull, (uint64_t)B.step_count[0]) & 1ull);
      if (lane < 13) {
        float v = 0.0f;
        if (lane < 3) v = sc.base_plate_pos[lane];
        else if (lane == 5) v = sinf(yaw_half);
        else if (lane == 6) v = cosf(yaw_half);
        root_e[141 * 13 + lane] = v;
      }
      // the target brick and the hand start from a grasp terminal state harvested by BlockAssemblyGraspSim (IS:1449-1456):
      // random slot of this env's brick-type ring, velocities zeroed, PD targets = the restored joint positions (IS:1478-1479)
      int cnt = B.harvest_count[e & 7];
      if (cnt > SDX_HARVEST_SLOTS - 1) cnt = SDX_HARVEST_SLOTS - 1;                // range(0, 5000)
      if (cnt > 0) {
        const int slot = (int)(sdx_hash(B.seed ^ 0x5EEDull, (uint64_t)e, (uint64_t)B.step_count[0]) % (uint64_t)cnt);
        const size_t o = (size_t)(e & 7) * SDX_HARVEST_SLOTS + slot;
        float* tg = root_e + seg_actor(e) * 13;
        if (lane < 13) tg[lane] = lane < 7 ? B.harvest_obj[o * 13 + lane] : 0.0f;  // IS:1452,1455
        if (lane < SDX_NDOF) {
          const float qh = B.harvest_hand[o * 46 + 2 * lane];                      // IS:1453,1456
          B.dof[((size_t)e * SDX_NDOF + lane) * 2 + 0] = qh;
          B.dof[((size_t)e * SDX_NDOF + lane) * 2 + 1] = 0.0f;
          B.prev_targets[(size_t)e * SDX_NDOF + lane] = qh;
          B.targets[(size_t)e * SDX_NDOF + lane] = qh;
        }
        if (lane < 3) B.init_pos[e * 3 + lane] = B.harvest_obj[o * 13 + lane];      // IS:1485
        if (lane < 4) B.init_rot[e * 4 + lane] = B.harvest_obj[o * 13 + 3 + lane];  // IS:1486
      }
    }
    if (lane == 0) {
      B.progress[e] = 0;                                                          // GS:1550-1553
      B.reset[e] = 0;
      B.successes[e] = 0.0f;
      B.meta_rew[e] = 0.0f;
      B.pile_choice[e] = choice;
    }
    __syncthreads();
  }
  if (!(flags & 4)) return;

  // ---- action -> targets (GS:1570-1638); actions are clamped to +-clip_actions first (VR:166)
  float a = 0.0f;
  if (lane < SDX_NDOF) {
    a = clampf(actions_in[(size_t)e * SDX_NDOF + lane], -sc.clip_actions, sc.clip_actions);
    B.actions[(size_t)e * SDX_NDOF + lane] = a;                                   // GS:1570
  }
  if (lane < 42) s_J[lane] = B.jac[(size_t)e * 42 + lane];
  const long prog = (long)B.progress[e];
  const bool m0 = prog > 75, m1 = prog > 100, m2 = prog > 125;                    // GS:1590-1592
  __syncthreads();
  // dpose (GS:1594-1600), every lane computes it (uniform)
  float dp[6];
  const bool search = sc.task_kind == 3;
  const bool orient = sc.task_kind == 1 || search, insert = sc.task_kind == 2;    // Search drives the arm like Orient (tracking IK)
  const bool hold = m0 && !insert && !search;                                     // GraspSim / Orient freeze the fingers after step 75
  if (!orient && !insert) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      float ak = __shfl(a, k, SDX_WAVE);
      dp[k] = ak * (k < 3 ? 0.64f : 0.2f);
    }
    if (m0) {
      const float hz = B.rb[((size_t)e * SDX_BODIES + sc.hand_base_body) * 13 + 2];
      dp[2] = 0.2f + 0.22f + (B.init_pos[e * 3 + 2] - hz);                          // GS:1596
      dp[0] = 0.0f;
      dp[1] = 0.0f;
    }
  } else {
    // BlockAssemblyOrient, OR:1733-1743: object-centric tracking - the hand base is held 0.22 above / 0.18 behind the target
    // brick with a fixed wrist orientation; after step 75 it lifts towards z_init + 0.39.
    // BlockAssemblyInsertSim, IS:1537-1539: the policy moves the hand base (a[0:3] * 0.64) and the wrist orientation is servoed.
    const float* hb = B.rb + ((size_t)e * SDX_BODIES + sc.hand_base_body) * 13;
    const float* tg = B.root + ((size_t)e * SDX_ACTORS + seg_actor(e)) * 13;
    if (insert) {
#pragma unroll
      for (int k = 0; k < 3; ++k) dp[k] = __shfl(a, k, SDX_WAVE) * 0.64f;
    } else {
      dp[0] = tg[0] - hb[0] - 0.18f;
      dp[1] = tg[1] - hb[1];
      dp[2] = tg[2] - hb[2] + (search ? 0.24f : 0.22f);                             // SE:1566 / OR:1735
      if (m0 && !search) dp[2] = B.init_pos[e * 3 + 2] - hb[2] + 0.15f + 0.24f;     // OR:1737
    }
    const f3 re = wrist_error(sc.target_euler, ld4(hb + 3));                        // OR:1740-1741, IS:1538-1539
    dp[3] = re.x; dp[4] = re.y; dp[5] = re.z;
    if (insert && lane == 0) {                                                      // self.rot_err feeds the reward's reset rule, IS:1539,1675
      float* aux = B.insert_aux + (size_t)e * 8;
      aux[0] = dp[3]; aux[1] = dp[4]; aux[2] = dp[5];
    }
  }
  float y[6];
  control_ik_solve(s_J, dp, y);                                                   // GS:1796-1804
  if (lane < SDX_NDOF) {
    const float lo = sc.lower[lane], hi = sc.upper[lane];
    const float prev = B.prev_targets[(size_t)e * SDX_NDOF + lane];
    float cur;
    if (lane >= 7) {
      cur = 0.5f * (a + 1.0f) * (hi - lo) + lo;                                   // scale(), GS:1585-1587
      cur = sc.act_moving_average * cur + (1.0f - sc.act_moving_average) * prev;  // GS:1588-1589
      if (hold) cur = prev;                                                       // GS:1606
    } else {
      float u = 0.0f;
#pragma unroll
      for (int r = 0; r < 6; ++r) u += s_J[r * 7 + lane] * y[r];
      cur = B.dof[((size_t)e * SDX_NDOF + lane) * 2] + u;                         // GS:1602
      if (m1 && !orient && !insert) cur = sc.insert_pose_a[lane];                 // GS:1604
      if (m2 && !orient && !insert) cur = sc.insert_pose_b[lane];                 // GS:1605
    }
    cur = fmaxf(fminf(cur, hi), lo);                                              // tensor_clamp GS:1633-1635
    B.targets[(size_t)e * SDX_NDOF + lane] = cur;
    B.prev_targets[(size_t)e * SDX_NDOF + lane] = cur;                            // GS:1636
  }
}

// ------------------------------------------------------------------------------------------------ K6
// flags: bit0 = post_physics_step (progress++, reward, resets); otherwise compute_observations only
__global__ __launch_bounds__(SDX_WAVE) void k_post_physics(const SdxConst* __restrict__ C, SdxBuf B, int flags) {
  const int e = blockIdx.x, lane = threadIdx.x;
  const sdx_scene_desc& sc = C->sc;
  __shared__ float s_in[5 * 13 + 13 + 7 + 46 + 18 + 23 + 7];   // hb, ff, mf, rf, th | target | base | dof | cf | act | init
  __shared__ float s_o[SDX_OBS_FRAME];
  __shared__ float s_s[SDX_STATE_FRAME];
  float* s_hb = s_in;
  float* s_ff = s_in + 13;
  float* s_mf = s_in + 26;
  float* s_rf = s_in + 39;
  float* s_th = s_in + 52;
  float* s_tg = s_in + 65;
  float* s_base = s_in + 78;
  float* s_dof = s_in + 85;
  float* s_cf = s_in + 131;
  float* s_act = s_in + 149;
  float* s_init = s_in + 172;

  long prog = (long)B.progress[e];
  if (flags & 1) {
    prog += 1;                                                                    // GS:1641
    if (lane == 0) {
      B.progress[e] = prog;
      B.randomize[e] += 1;                                                        // GS:1642
    }
  }
  const float* rb_e = B.rb + (size_t)e * SDX_BODIES * 13;
  const float* root_e = B.root + (size_t)e * SDX_ACTORS * 13;
  {  // gathers (GS:1097-1152): 5 rigid-body rows, the target brick's root row, robot base pose, dof, contacts
    for (int i = lane; i < 65; i += SDX_WAVE) {
      int which = i / 13, c = i % 13;
      int body = which == 0 ? sc.hand_base_body : sc.fingertip_body[which - 1];
      s_in[i] = rb_e[body * 13 + c];
    }
    if (lane < 13) s_tg[lane] = root_e[seg_actor(e) * 13 + lane];
    if (lane < 7) s_base[lane] = root_e[lane];
    if (lane < 46) s_dof[lane] = B.dof[(size_t)e * 46 + lane];
    if (lane < 18) s_cf[lane] = B.contact[(size_t)e * SDX_BODIES * 3 + 3 + lane];  // bodies 1..6, GS:1032-1033,1159-1160
    if (lane < 23) s_act[lane] = B.actions[(size_t)e * 23 + lane];
    if (lane < 3) s_init[lane] = B.init_pos[e * 3 + lane];
    if (lane >= 3 && lane < 7) s_init[lane] = B.init_rot[e * 4 + lane - 3];
  }
  __syncthreads();

  // ---- derived quantities, computed by every lane (wave-uniform, no divergence)
  const f3 off = F3(0.0f, 0.0f, 0.04f);                                           // GS:1154-1157
  const f3 tpos = ld3(s_tg);
  const f4 trot = ld4(s_tg + 3);
  const f3 hpos = ld3(s_hb);
  const f4 hrot = ld4(s_hb + 3);
  const f3 ffp = ld3(s_ff) + qrot(ld4(s_ff + 3), off);
  const f3 mfp = ld3(s_mf) + qrot(ld4(s_mf + 3), off);
  const f3 rfp = ld3(s_rf) + qrot(ld4(s_rf + 3), off);
  const f3 thp = ld3(s_th) + qrot(ld4(s_th + 3), off);
  const f3 dff = tpos - ffp, dmf = tpos - mfp, drf = tpos - rfp, dth = tpos - thp;
  const float nff = sqrtf(dot(dff, dff)), nmf = sqrtf(dot(dmf, dmf)), nrf = sqrtf(dot(drf, drf)),
              nth = sqrtf(dot(dth, dth));
  const float finger_dist = nff + nmf + nrf + nth;                                // GS:1164-1165
  // hand pose in the robot-base frame (GS:1172-1173)
  const f4 qbi = qconj(ld4(s_base + 3));
  const f3 pbi = qrot(qbi, ld3(s_base)) * -1.0f;
  const f4 hv_rot = qmul(qbi, hrot);
  const f3 hv_pos = qrot(qbi, hpos) + pbi;
  // wrist camera frame = link7 o (q_off, p_off); target pose in that frame (GS:1176-1182)
  const f4 qc = qmul(hrot, ld4(sc.camera_offset_quat));
  const f3 pc = qrot(hrot, ld3(sc.camera_offset_pos)) + hpos;
  const f4 qci = qconj(qc);
  const f3 pci = qrot(qci, pc) * -1.0f;
  const f4 ct_rot = qmul(qci, trot);
  const f3 ct_pos = qrot(qci, tpos) + pci;
  const f4 hq_rel = qmul(hrot, qconj(trot));                                      // GS:1263
  // BlockAssemblyInsertSim: the insertion site = base plate pose shifted in the plate frame by 0.0375 (1 + env % 3) in z and one
  // stud pitch in y (1xn bricks) or in x and y (the 1x1 brick, env % 8 == 5), IS:779-812,1119-1130; its 180-degree twin IS:1167
  const bool insert = sc.task_kind == 2;
  f3 epos = F3(0.0f, 0.0f, 0.0f);
  f4 erot = {0.0f, 0.0f, 0.0f, 1.0f};
  float gap = 0.0f, rot_dist = 0.0f;
  if (insert) {
    const float* ex = root_e + 141 * 13;
    erot = ld4(ex + 3);
    const bool one = (e & 7) == 5;
    epos = ld3(ex) + qrot(erot, F3(0.0f, 0.0f, 1.0f)) * (0.0375f * (float)(1 + e % 3));
    if (!one) epos = epos + qrot(erot, F3(0.0f, 1.0f, 0.0f)) * 0.015f;
    else epos = epos + qrot(erot, F3(1.0f, 0.0f, 0.0f)) * 0.015f + qrot(erot, F3(0.0f, 1.0f, 0.0f)) * 0.015f;
    const f4 zq = {0.0f, 0.0f, 1.0f, 0.0f};
    const f4 esym = qmul(erot, zq);
    const f4 d1 = qmul(trot, qconj(erot)), d2 = qmul(trot, qconj(esym));
    const float r1 = 2.0f * asinf(fminf(sqrtf(d1.x * d1.x + d1.y * d1.y + d1.z * d1.z), 1.0f));   // IS:1656-1660
    const float r2 = 2.0f * asinf(fminf(sqrtf(d2.x * d2.x + d2.y * d2.y + d2.z * d2.z), 1.0f));
    rot_dist = fminf(r1, r2);
    const f3 dg = tpos - epos;
    gap = sqrtf(dot(dg, dg));
  }

  // ---- frames in LDS: bulk copies lane-parallel, derived values by lane 0
  if (lane < 16) {
    const int j = 7 + lane;
    const float lo = sc.lower[j], hi = sc.upper[j];
    s_o[lane] = (2.0f * s_dof[2 * j] - hi - lo) / (hi - lo);                      // unscale, GS:1300-1302
    s_o[30 + lane] = 0.2f * s_dof[2 * j + 1];                                     // GS:1310
  }
  if (lane < 13) {
    s_o[46 + lane] = s_ff[lane];                                                  // GS:1312-1315 (ff, rf, mf, th)
    s_o[59 + lane] = s_rf[lane];
    s_o[72 + lane] = s_mf[lane];
    s_o[85 + lane] = s_th[lane];
    s_o[98 + lane] = s_tg[lane];                                                  // GS:1317
  }
  if (lane < 7) {
    s_o[111 + lane] = s_hb[lane];                                                 // GS:1319-1320
    s_o[118 + lane] = s_init[lane];                                               // GS:1322-1323
    s_s[81 + lane] = s_hb[lane];                                                  // GS:1232
    s_s[88 + lane] = s_tg[lane];                                                  // GS:1234
  }
  if (lane < 23) {
    const float lo = sc.lower[lane], hi = sc.upper[lane];
    s_s[lane] = (2.0f * s_dof[2 * lane] - hi - lo) / (hi - lo);                   // GS:1221-1223
    s_s[23 + lane] = 0.2f * s_dof[2 * lane + 1];                                  // GS:1224
    s_s[58 + lane] = s_act[lane];                                                 // GS:1231
  }
  if (lane < 6) s_s[95 + lane] = s_hb[7 + lane];                                  // GS:1236-1237
  if (lane < 40) {                                                                // GS:1239-1253: ff, mf, rf, th (rot, linvel, angvel)
    const int f = lane / 10, c = lane % 10;
    s_s[101 + lane] = s_in[13 * (f + 1) + 3 + c];
  }
  if (lane < 6) s_s[142 + lane] = s_tg[7 + lane];                                 // GS:1255-1256
  __builtin_amdgcn_wave_barrier();   // lane 0 overwrites some of the bulk copies above (InsertSim's frame): program order across the lanes of the wave
  if (lane == 0) {
    st3(s_o + 16, hv_pos);  st4(s_o + 19, hv_rot);                                // GS:1304-1305
    st3(s_o + 23, ct_pos);  st4(s_o + 26, ct_rot);                                // GS:1307-1308
    st3(s_o + 125, tpos - ld3(s_init));                                           // GS:1325
    st3(s_o + 128, hpos - tpos);                                                  // GS:1326
    s_o[131] = 0.0f;
    st3(s_s + 46, ffp); st3(s_s + 49, rfp); st3(s_s + 52, mfp); st3(s_s + 55, thp);   // GS:1226-1229
    s_s[141] = 0.0f;
    st3(s_s + 148, ld3(s_init));                                                  // GS:1258
    st3(s_s + 151, tpos - ld3(s_init));                                           // GS:1259
    st3(s_s + 154, hpos - tpos);                                                  // GS:1262
    st4(s_s + 157, hq_rel);
    st3(s_s + 161, dff); st3(s_s + 164, drf); st3(s_s + 167, dmf); st3(s_s + 170, dth);   // GS:1265-1268
    s_s[173] = finger_dist;                                                       // GS:1270
    st3(s_s + 174, ct_pos); st4(s_s + 177, ct_rot);                               // GS:1272-1276
    st3(s_s + 181, ct_pos); st4(s_s + 184, ct_rot);
    if (insert) {
      s_s[141] = (float)prog / sc.max_episode_length;                             // IS:1255
      st3(s_s + 181, epos); st4(s_s + 184, erot);                                 // IS:1277-1278
      float* aux = B.insert_aux + (size_t)e * 8;
      aux[3] = gap; aux[4] = rot_dist;
      // compute_contact_observations IS:1280-1298, staged behind the GraspSim frame (s_o[0:16] already holds the finger joints)
      st3(s_o + 46, hpos - epos);  st4(s_o + 49, qmul(hrot, qconj(erot)));
      st3(s_o + 53, hpos - tpos);  st4(s_o + 56, hq_rel);
      s_o[60] = 0.0f;
      st3(s_o + 61, epos);         st4(s_o + 64, erot);
      st3(s_o + 68, tpos - epos);  st4(s_o + 71, qmul(trot, qconj(erot)));
    }
    st4(B.cam_rot + (size_t)e * 4, ct_rot);
    B.finger_dist[e] = finger_dist;
  }
  if (lane < 6) {                                                                 // GS:1159-1162
    const f3 f = ld3(s_cf + 3 * lane);
    B.arm_contacts[(size_t)e * 6 + lane] = sqrtf(dot(f, f)) >= 0.1f ? 1.0f : 0.0f;
  }
  __syncthreads();

  // ---- 3-frame stacking (GS:1330-1332, 1278-1280): new = [frame, old[0:w], old[w:2w]]; rows are read fully
  // before they are written (one wave owns the row), then written lane-strided together with the clamped copies
  {
    float* o = B.obs + (size_t)e * B.obs_w;
    float* oc = B.obs_c + (size_t)e * B.obs_w;
    if (sc.task_kind == 1 || sc.task_kind == 3) {
      // BlockAssemblyOrient, compute_real_observations OR:1308-1326 (= Search's compute_contact_observations SE:1220-1230): 62 numbers,
      // NOT stacked (columns 62..185 are never written)
      if (lane < 62) {
        float v = 0.0f;
        if (lane < 16) v = s_o[lane];                                             // unscaled finger joint positions
        else if (lane >= 30 && lane < 46) v = s_act[7 + lane - 30] - s_o[lane - 30];   // action - unscaled position, OR:1322-1324
        else if (lane >= 46) v = s_act[7 + lane - 46];                            // OR:1326
        o[lane] = v;
        oc[lane] = clampf(v, -sc.clip_obs, sc.clip_obs);
      }
    } else if (insert) {
      // BlockAssemblyInsertSim: 75 numbers, one frame (stack_obs = 1, IS:172); columns 16..22 and 60 are never written
      for (int c = lane; c < 75; c += SDX_WAVE) {
        float v = s_o[c];
        if (c >= 16 && c < 23) v = 0.0f;
        else if (c >= 23 && c < 46) v = s_act[c - 23];                            // IS:1285
        o[c] = v;
        oc[c] = clampf(v, -sc.clip_obs, sc.clip_obs);
      }
    } else {
    float hist[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int c = lane + r * SDX_WAVE;
      hist[r] = (c < 2 * SDX_OBS_FRAME) ? o[c] : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();   // every lane has read its part of the row before any lane overwrites it (no instruction: the wave runs in lockstep)
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int c = lane + r * SDX_WAVE;
      if (c < 2 * SDX_OBS_FRAME) {
        o[SDX_OBS_FRAME + c] = hist[r];
        oc[SDX_OBS_FRAME + c] = clampf(hist[r], -sc.clip_obs, sc.clip_obs);
      }
    }
    for (int c = lane; c < SDX_OBS_FRAME; c += SDX_WAVE) {
      const float v = s_o[c];
      o[c] = v;
      oc[c] = clampf(v, -sc.clip_obs, sc.clip_obs);
    }
    }
    float* s = B.states + (size_t)e * SDX_NUM_STATES;
    float* stc = B.states_c + (size_t)e * SDX_NUM_STATES;
    const bool search = sc.task_kind == 3;
    if (search) {
      // Search's own asymmetric frame, SE:1168-1218, built from the GraspSim frame staged above: joints, fingertips, actions, hand and
      // target poses stay where they are; the eight hand-position history means are zero (they are only ever computed from a zeroed
      // buffer, SE:1458-1466), then the pixel statistics, the hand twist, the fingertip (rot, linvel, angvel) blocks, the target twist
      __syncthreads();
      float keep[3] = {0.0f, 0.0f, 0.0f};
      for (int r = 0; r < 3; ++r) { const int c = lane + r * SDX_WAVE; if (c < SDX_STATE_FRAME) keep[r] = s_s[c]; }
      __syncthreads();
      for (int r = 0; r < 3; ++r) {
        const int c = lane + r * SDX_WAVE;
        if (c >= 95 && c < SDX_STATE_FRAME) s_s[c] = 0.0f;
      }
      __syncthreads();
      for (int r = 0; r < 3; ++r) {                                                 // old column -> new column
        const int c = lane + r * SDX_WAVE;
        if (c >= 95 && c < 101) s_s[c + 28] = keep[r];                              // hand twist 95:101 -> 123:129
        else if (c >= 101 && c < 141) s_s[c + 28] = keep[r];                        // fingertips 101:141 -> 129:169
        else if (c >= 142 && c < 148) s_s[c + 27] = keep[r];                        // target twist 142:148 -> 169:175
      }
      if (lane == 0) {
        const float* px = B.seg_pix + (size_t)e * 4;
        s_s[120] = px[1] / 128.0f; s_s[121] = px[2] / 128.0f; s_s[122] = px[0] / 100.0f;   // SE:1192-1194
      }
      __syncthreads();
    }
    if (!insert && !search) {                                                     // InsertSim's / Search's 188 states are one frame:
      float hs[6];                                                                // columns 188.. of its rows stay zero
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const int c = lane + r * SDX_WAVE;
        hs[r] = (c < 2 * SDX_STATE_FRAME) ? s[c] : 0.0f;
      }
      __builtin_amdgcn_wave_barrier();   // as for the observation row: all reads of the row precede its writes
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const int c = lane + r * SDX_WAVE;
        if (c < 2 * SDX_STATE_FRAME) {
          s[SDX_STATE_FRAME + c] = hs[r];
          stc[SDX_STATE_FRAME + c] = clampf(hs[r], -sc.clip_obs, sc.clip_obs);
        }
      }
    }
    for (int c = lane; c < SDX_STATE_FRAME; c += SDX_WAVE) {
      const float v = s_s[c];
      s[c] = v;
      stc[c] = clampf(v, -sc.clip_obs, sc.clip_obs);
    }
  }
  if (!(flags & 1)) return;

  // ---- compute_hand_reward (GS:1706-1776 / OR:1843-1907)
  if (lane == 0) {
    const float d = nff + nmf + nrf + 3.0f * nth;                                 // GS:1740-1741, OR:1853-1854
    long resets = (long)B.reset[e];                                               // GS:1727 (d <= -1 never holds)
    const bool timed_out = (float)prog >= sc.max_episode_length - 1.0f;           // GS:1729
    if (timed_out) resets = 1;
    float reward;
    if (sc.task_kind == 3) {
      // Search, SE:1660-1711: min(-0.2 d, -0.06) - arm contacts - 0.005 |a|^2 + lift term (unweighted fingertip distance); the camera's
      // emergence reward does not enter; time-out is the only reset
      const float d4 = nff + nmf + nrf + nth;
      float asq = 0.0f, ac = 0.0f;
      for (int j = 0; j < SDX_NDOF; ++j) asq += s_act[j] * s_act[j];
      for (int j = 0; j < 6; ++j) { const f3 f = ld3(s_cf + 3 * j); ac += sqrtf(dot(f, f)) >= 0.1f ? 1.0f : 0.0f; }
      const float up = clampf(tpos.z - s_init[2], 0.0f, 0.1f) * 1000.0f - clampf(tpos.x - s_init[0], 0.0f, 0.1f) * 1000.0f -
                       clampf(tpos.y - s_init[1], 0.0f, 0.1f) * 1000.0f;
      reward = fminf(-0.2f * d4, -0.06f) - ac - asq * 0.005f + up;
    } else if (insert) {
      // InsertSim, IS:1640-1695: exp(-rot_dist - 20 |brick - site|) + 1 once seated; reset when the hand lets go, when the wrist
      // servo error (the rot_err of this step's pre_physics_step) grows, or on time-out
      const float* aux = B.insert_aux + (size_t)e * 8;
      reward = expf(-rot_dist - 20.0f * gap) + ((gap < 0.02f && rot_dist < 0.2f) ? 1.0f : 0.0f);   // IS:1664-1668,1680
      resets = (long)B.reset[e];
      if (d >= 0.6f) resets = 1;                                                  // IS:1673
      if (aux[0] * aux[0] + aux[1] * aux[1] + aux[2] * aux[2] >= 0.03f) resets = 1;   // IS:1675
      if (timed_out) resets = 1;                                                  // IS:1677-1678
    } else if (sc.task_kind == 1) {
      // Orient: exp(-5 (1 - (z_align + 1) / 2) - 5 max(d - 0.4, 0)), distance term dropped after step 175; time-out is the only
      // reset (max_consecutive_successes = 0 in the shipped config, so the fall-penalty term OR:1900-1901 is inactive)
      const float dot1 = qrot(trot, F3(0.0f, 0.0f, 1.0f)).z;                      // OR:1856-1859
      const float z_align = (dot1 > 0.0f ? 1.0f : (dot1 < 0.0f ? -1.0f : 0.0f)) * dot1 * dot1;
      const float d_rew = prog > 175 ? 0.0f : fmaxf(d - 0.4f, 0.0f);              // OR:1878-1879
      reward = expf(-(5.0f * (1.0f - (z_align + 1.0f) * 0.5f) + 5.0f * d_rew));   // OR:1884-1886
    } else {
      const float dist_rew = expf(-2.0f * fmaxf(d - 0.5f, 0.0f)) * 0.1f;          // GS:1742
      float up = clampf(tpos.z - s_init[2], 0.0f, 0.2f) * 100.0f;                 // GS:1744
      up = fminf(d < 0.5f ? up : 0.0f, 20.0f);                                    // GS:1745
      reward = dist_rew + up;                                                     // GS:1751
      if (prog >= 75 && d >= 0.6f) resets = 1;                                    // GS:1754-1755
    }
    B.rew[e] = reward;
    B.reset[e] = resets;
    B.meta_rew[e] += reward;                                                      // GS:1069
    if (resets) {                                                                 // GS:1771-1772 (summed by k_tvalue block 0)
      const int par = (int)(B.step_count[0] & 1u);
      atomicAdd(&B.stat[par * 2 + 0], 1.0f);
      atomicAdd(&B.stat[par * 2 + 1], B.successes[e]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ T-value MLP
// tv_w layout (device, transposed for coalesced neuron-major reads): W1T[4][256] b1[256] W2T[256][128] b2[128]
// W3T[128][64] b3[64] W4T[64][2] b4[2].  One block = 16 envs x 256 threads.
#define TV_ENVS 16
#define TV_LOADS 32
__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : expm1f(x); }

// Batches of TV_LOADS weight loads.  TV_LOAD's index passes through an empty asm that also takes `after`, a value of the arithmetic the batch
// must stay behind: the optimiser can then neither merge the batches into one 192-register group nor sink a load to its use, and
// sched_barrier holds the order in the machine scheduler.
#define TV_LOAD(w, expr, after) do { int o_ = 0; SDX_OPAQUE_AFTER(o_, after); _Pragma("unroll") for (int j = 0; j < TV_LOADS; ++j) { const int jj = j + o_; w[j] = (expr); } \
                                     __builtin_amdgcn_sched_barrier(0); } while (0)
__global__ __launch_bounds__(256) void k_tvalue(SdxBuf B, int finalize_stats) {
  __shared__ float s_x[TV_ENVS][4];
  __shared__ float s_h1[TV_ENVS][256];
  __shared__ float s_h2[TV_ENVS][128];
  __shared__ float s_h3[TV_ENVS][64];
  const int t = threadIdx.x, e0 = blockIdx.x * TV_ENVS;
  const float* W1 = B.tv_w;
  const float* b1 = W1 + 4 * 256;
  const float* W2 = b1 + 256;
  const float* b2 = W2 + 256 * 128;
  const float* W3 = b2 + 128;
  const float* b3 = W3 + 128 * 64;
  const float* W4 = b3 + 64;
  const float* b4 = W4 + 64 * 2;
  // The launch is 64 workgroups on an otherwise idle chip: its time is the chain of dependent L2 round trips (about 0.7 us each), not
  // arithmetic.  Rolled, every k of every layer was one (32 us per launch).  Now every input that depends on nothing is requested at the top,
  // the weights move in batches of TV_LOADS, and batch i + 1 (or the next layer's first) is in flight while batch i is multiplied.
  // The sums still run over k in ascending order: same values bit for bit.
  const int n2 = t & 127, eh = (t >> 7) * 8;        // layer 2: neuron = t % 128, env half = t / 128
  const int n3 = t & 63, eq = (t >> 6) * 4;         // layer 3: neuron = t % 64, env quarter = t / 64
  float wa[TV_LOADS], wb[TV_LOADS], w1[4];
  const float xin = t < TV_ENVS * 4 && e0 + t / 4 < B.N ? B.cam_rot[(size_t)(e0 + t / 4) * 4 + (t % 4)] : 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) w1[k] = W1[k * 256 + t];
  const float bias1 = b1[t], bias2 = b2[n2], bias3 = b3[n3], bias4 = b4[1];
  TV_LOAD(wa, W2[jj * 128 + n2], 0.0f);
  if (finalize_stats && blockIdx.x == 0 && t == 255) {
    // cons_successes EMA (GS:1771-1774) from the per-step sums gathered by k_post_physics (independent of the T-value: up here its three
    // dependent round trips overlap the MLP instead of trailing it)
    const uint32_t step = B.step_count[0];
    const int par = (int)(step & 1u);
    const float num_resets = B.stat[par * 2 + 0], fin = B.stat[par * 2 + 1];
    if (num_resets > 0.0f) B.cons[0] = 0.1f * fin / num_resets + 0.9f * B.cons[0];
    B.stat[par * 2 + 0] = 0.0f;
    B.stat[par * 2 + 1] = 0.0f;
    B.step_count[0] = step + 1;
  }
  if (t < TV_ENVS * 4) s_x[t / 4][t % 4] = xin;
  __syncthreads();
  float h1keep = 0.0f;
#pragma unroll
  for (int e = 0; e < TV_ENVS; ++e) {                // layer 1: thread = neuron, all 16 envs
    h1keep = elu1(bias1 + w1[0] * s_x[e][0] + w1[1] * s_x[e][1] + w1[2] * s_x[e][2] + w1[3] * s_x[e][3]);
    s_h1[e][t] = h1keep;
  }
  __syncthreads();
  static_assert(TV_LOADS == 32, "the batches below are written out for 256 / 128 / 64 inputs in batches of 32");
  // SDX_PIN after every 4 k: without it acc[0] (which the next TV_LOAD names) is finished first, the other accumulators are deferred to the end
  // of the block and the LDS operands already read for them are spilled (6.5 KB of scratch per lane)
#define TV_MAC(acc, NE, w, H, e0_, k0) do { _Pragma("unroll") for (int j4 = 0; j4 < TV_LOADS; j4 += 4) { _Pragma("unroll") for (int j = j4; j < j4 + 4; ++j) { \
                                             _Pragma("unroll") for (int e = 0; e < NE; ++e) acc[e] += w[j] * H[e0_ + e][(k0) + j]; } \
                                             SDX_PIN##NE(acc); } __builtin_amdgcn_sched_barrier(0); } while (0)
  {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias2;
    TV_LOAD(wb, W2[(32 + jj) * 128 + n2], h1keep);   TV_MAC(acc, 8, wa, s_h1, eh, 0);
    TV_LOAD(wa, W2[(64 + jj) * 128 + n2], acc[0]);   TV_MAC(acc, 8, wb, s_h1, eh, 32);
    TV_LOAD(wb, W2[(96 + jj) * 128 + n2], acc[0]);   TV_MAC(acc, 8, wa, s_h1, eh, 64);
    TV_LOAD(wa, W2[(128 + jj) * 128 + n2], acc[0]);  TV_MAC(acc, 8, wb, s_h1, eh, 96);
    TV_LOAD(wb, W2[(160 + jj) * 128 + n2], acc[0]);  TV_MAC(acc, 8, wa, s_h1, eh, 128);
    TV_LOAD(wa, W2[(192 + jj) * 128 + n2], acc[0]);  TV_MAC(acc, 8, wb, s_h1, eh, 160);
    TV_LOAD(wb, W2[(224 + jj) * 128 + n2], acc[0]);  TV_MAC(acc, 8, wa, s_h1, eh, 192);
    TV_LOAD(wa, W3[jj * 64 + n3], acc[0]);           TV_MAC(acc, 8, wb, s_h1, eh, 224);       // (layer 3's first batch)
#pragma unroll
    for (int e = 0; e < 8; ++e) s_h2[eh + e][n2] = elu1(acc[e]);
    h1keep = acc[0];
  }
  __syncthreads();
  {
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = bias3;
    TV_LOAD(wb, W3[(32 + jj) * 64 + n3], h1keep);    TV_MAC(acc, 4, wa, s_h2, eq, 0);
    TV_LOAD(wa, W3[(64 + jj) * 64 + n3], acc[0]);    TV_MAC(acc, 4, wb, s_h2, eq, 32);
    TV_LOAD(wb, W3[(96 + jj) * 64 + n3], acc[0]);    TV_MAC(acc, 4, wa, s_h2, eq, 64);
    TV_LOAD(wa, W4[jj * 2 + 1], acc[0]);             TV_MAC(acc, 4, wb, s_h2, eq, 96);        // (layer 4: one output, the same weights for every thread)
    TV_LOAD(wb, W4[(32 + jj) * 2 + 1], acc[0]);
#pragma unroll
    for (int e = 0; e < 4; ++e) s_h3[eq + e][n3] = elu1(acc[e]);
  }
  __syncthreads();
#undef TV_MAC
  if (t < TV_ENVS) {  // layer 4, output 1 only feeds the sigmoid (GS:1201)
    float y = bias4;
#pragma unroll
    for (int j = 0; j < TV_LOADS; ++j) y += wa[j] * s_h3[t][j];
#pragma unroll
    for (int j = 0; j < TV_LOADS; ++j) y += wb[j] * s_h3[t][32 + j];
    y = elu1(y);
    if (e0 + t < B.N) {
      float tvv = 1.0f / (1.0f + expf(-y));
      if (B.task_kind == 1) tvv = tvv > B.orient_gate ? 1.0f : 0.0f;             // Orient gates the T-value at 0.99, OR:1203 (sdx_scene_desc.orient_tvalue_gate)
      B.tvalue[e0 + t] = tvv;
    }
  }
}

// ------------------------------------------------------------------------------------------------ Search: temporal T-value (SE:1133-1166)
#define TVT_STRIDE 652
#define TVT_FRAME 65
// B.tvalue = sigmoid(out[:, 1]) of the RetriGraspTValue forward that the host launcher ran on the buffer of the PREVIOUS step (SE:1133-1134)
__global__ void k_search_tvalue_out(SdxBuf B) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B.N) return;
  const float y = B.tvt_h[(size_t)B.N * (1024 + 512 + 128) + (size_t)e * 2 + 1];   // ELU already applied by the last layer (the reference activates the output layer too)
  B.tvalue[e] = 1.0f / (1.0f + expf(-y));
}
// shift the ten frames by one and append this step's (SE:1155-1166): obs_buf[:, 0:62] with [26:30] = camera-frame target quaternion,
// centroid x / 128, centroid y / 128, pixel count / 100.  One wave per env; the shift reads a frame before anything overwrites it.
__global__ __launch_bounds__(SDX_WAVE) void k_search_tvalue_append(SdxBuf B) {
  const int e = blockIdx.x, lane = threadIdx.x;
  float* row = B.tvt_buf + (size_t)e * TVT_STRIDE;
  for (int f = 0; f < 9; ++f) {
    float a = 0.0f, b = 0.0f;
    if (lane < TVT_FRAME) a = row[(f + 1) * TVT_FRAME + lane];
    if (lane == 0) b = row[(f + 1) * TVT_FRAME + 64];
    __syncthreads();
    if (lane < 64) row[f * TVT_FRAME + lane] = a;
    if (lane == 0) row[f * TVT_FRAME + 64] = b;
    __syncthreads();
  }
  float v = 0.0f;
  if (lane < 62) v = B.obs[(size_t)e * B.obs_w + lane];
  if (lane >= 26 && lane < 30) v = B.cam_rot[(size_t)e * 4 + lane - 26];
  if (lane == 62) v = B.seg_pix[(size_t)e * 4 + 1] / 128.0f;       // segmentation_object_center_point_x
  if (lane == 63) v = B.seg_pix[(size_t)e * 4 + 2] / 128.0f;
  row[9 * TVT_FRAME + lane] = v;
  if (lane == 0) row[9 * TVT_FRAME + 64] = B.seg_pix[(size_t)e * 4 + 0] / 100.0f;
}
extern "C" void sdxpk_linear(const float* X, const float* W, const float* b, float* Y, int M, int N, int K, int elu_flag,
                             const double* nmean, const double* nvar, hipStream_t st);

// ------------------------------------------------------------------------------------------------ host launchers
// ------------------------------------------------------------------------------------------------ BlockAssemblyOrient reset
// The Orient task scripts the arm with the tracking IK of its pre_physics_step while the simulator runs 50 steps, twice per reset:
//   mode 0 (OR:1427-1461, before the states are restored, only when total_steps > 0): hand base to 0.42 above / 0.18 behind the
//          target brick's CURRENT position, targets clamped to the joint limits, finger targets = their values at loop entry - 0.01;
//   mode 1 (OR:1655-1695, after the restore): hand base to 0.22 (+0.2 during the first 20 iterations) above / 0.18 behind the
//          target brick's INITIAL position, no clamp, fingers at the reset pose.
// Only envs with mask != 0 receive targets (set_dof_position_target_tensor_indexed with the resetting hands).
__global__ __launch_bounds__(SDX_WAVE) void k_orient_pregrasp(const SdxConst* __restrict__ C, SdxBuf B, const uint8_t* __restrict__ mask,
                                                              int mode, int iter) {
  const int e = blockIdx.x, lane = threadIdx.x;
  if (!mask[e]) return;
  const sdx_scene_desc& sc = C->sc;
  __shared__ float s_J[42];
  if (lane < 42) s_J[lane] = B.jac[(size_t)e * 42 + lane];
  __syncthreads();
  const float* hb = B.rb + ((size_t)e * SDX_BODIES + sc.hand_base_body) * 13;
  float dp[6];
  if (mode == 0) {
    const float* tg = B.root + ((size_t)e * SDX_ACTORS + seg_actor(e)) * 13;
    dp[0] = tg[0] - hb[0] - 0.18f; dp[1] = tg[1] - hb[1]; dp[2] = tg[2] - hb[2] + 0.42f;          // OR:1436-1438
  } else {
    dp[0] = B.init_pos[e * 3 + 0] - hb[0] - 0.18f; dp[1] = B.init_pos[e * 3 + 1] - hb[1];
    dp[2] = B.init_pos[e * 3 + 2] - hb[2] + 0.22f + (iter < 20 ? 0.2f : 0.0f);                   // OR:1662-1667
  }
  const f3 re = wrist_error(sc.target_euler, ld4(hb + 3));
  dp[3] = re.x; dp[4] = re.y; dp[5] = re.z;
  float y[6];
  control_ik_solve(s_J, dp, y);                                                                  // OR:1927-1935
  if (lane < SDX_NDOF) {
    float cur;
    if (lane < 7) {
      float u = 0.0f;
#pragma unroll
      for (int r = 0; r < 6; ++r) u += s_J[r * 7 + lane] * y[r];
      cur = B.dof[((size_t)e * SDX_NDOF + lane) * 2] + u;
      if (mode == 0) cur = fmaxf(fminf(cur, sc.upper[lane]), sc.lower[lane]);                    // OR:1448-1450
    } else if (mode == 0) {
      // prev_targets holds the finger targets at loop entry (cur_targets_clone, OR:1429,1452-1453)
      cur = B.prev_targets[(size_t)e * SDX_NDOF + lane] - 0.01f;
    } else {
      cur = C->hand_reset_pose[lane];                                                            // OR:1678-1683
    }
    B.targets[(size_t)e * SDX_NDOF + lane] = cur;
    if (mode == 1 || lane < 7) B.prev_targets[(size_t)e * SDX_NDOF + lane] = cur;               // OR:1685 / OR:1453 keeps the clone
  }
}
// after the restore and two simulator steps (OR:1618-1646): the target brick's settled pose becomes the episode's initial pose and
// the hand is put back to the prepare pose with zero velocity
__global__ __launch_bounds__(SDX_WAVE) void k_orient_post_reset(const SdxConst* __restrict__ C, SdxBuf B, const uint8_t* __restrict__ mask) {
  const int e = blockIdx.x, lane = threadIdx.x;
  if (!mask[e]) return;
  const float* tg = B.root + ((size_t)e * SDX_ACTORS + seg_actor(e)) * 13;
  if (lane < 3) B.init_pos[e * 3 + lane] = tg[lane];                                            // OR:1627
  if (lane < 4) B.init_rot[e * 4 + lane] = tg[3 + lane];                                        // OR:1628
  if (lane < SDX_NDOF) {
    const float hp = C->hand_reset_pose[lane];
    B.dof[((size_t)e * SDX_NDOF + lane) * 2 + 0] = hp;                                          // OR:1635-1646
    B.dof[((size_t)e * SDX_NDOF + lane) * 2 + 1] = 0.0f;
    B.prev_targets[(size_t)e * SDX_NDOF + lane] = hp;
    B.targets[(size_t)e * SDX_NDOF + lane] = hp;
  }
}
// ------------------------------------------------------------------------------------------------ BlockAssemblySearch helpers
// mode 0 (SE:1482-1493, end of post_reset, masked envs): hand to the prepare pose (joint state set directly, zero velocity), PD targets
//         there too, and the settled pose of the target brick becomes the episode's initial pose;
// mode 1 (SE:992-1003, last step of an episode, all envs): hand parked at the default pose so that it does not hide the pile from
//         the camera; the caller then simulates one step and renders
__global__ __launch_bounds__(SDX_WAVE) void k_search_set_hand(const SdxConst* __restrict__ C, SdxBuf B, const uint8_t* __restrict__ mask, int mode) {
  const int e = blockIdx.x, lane = threadIdx.x;
  if (mask && !mask[e]) return;
  const sdx_scene_desc& sc = C->sc;
  if (lane < SDX_NDOF) {
    const float arm = mode == 0 ? sc.arm_prepare_pose[lane < 7 ? lane : 0] : sc.search_default_arm[lane < 7 ? lane : 0];
    const float qh = lane < 7 ? arm : sc.search_finger_pose[lane - 7];
    B.dof[((size_t)e * SDX_NDOF + lane) * 2 + 0] = qh;
    B.dof[((size_t)e * SDX_NDOF + lane) * 2 + 1] = 0.0f;
    B.prev_targets[(size_t)e * SDX_NDOF + lane] = qh;
    B.targets[(size_t)e * SDX_NDOF + lane] = qh;
  }
  if (mode == 0) {
    const float* tg = B.root + ((size_t)e * SDX_ACTORS + seg_actor(e)) * 13;
    if (lane < 3) B.init_pos[e * 3 + lane] = tg[lane];                                          // SE:1494
    if (lane < 4) B.init_rot[e * 4 + lane] = tg[3 + lane];                                      // SE:1495
  }
}
// STAND-IN for a trained BlockAssemblyGraspSim policy (the chain benchmark / tests; seqdex_amd/scripts/evaluation.py::scripted_grasp_controller
// documents why): a reach - descend - pinch - hold sequence on the task's own action interface (GS:1586-1609).  One thread per env; `state`
// [2 N + 8] is the controller's only memory, owned by the caller: per env [0] the progress value at which the env's fingers started to close,
// [1] the progress value at which the hand stopped following the brick (1e9: not yet); then eight parameters: the finger closure the pinch
// ends at (fraction of the joint range), the number of steps the pinch takes, the hand's rise per step while it holds (action units), the
// height of the hand base above the brick's origin at the pinch, the pinch point's x / y offset from the hand base, two spare.
// Round 5: once the pinch is complete the hand stops following the brick - a gripped brick moves with the hand, following it was a positive
// feedback that carried hand and brick away - and raises it a little; after step 75 the task itself lifts the hand (GS:1600-1609).
__global__ __launch_bounds__(256) void k_scripted_grasp(const SdxConst* __restrict__ C, SdxBuf B, float* __restrict__ state, float* __restrict__ act) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B.N) return;
  const sdx_scene_desc& sc = C->sc;
  const float* par = state + 2 * (size_t)B.N;
  const float frac_end = par[0], pinch_steps = par[1], rise = par[2], zpinch = par[3], xo = par[4], yo = par[5];
  const float prog = (float)B.progress[e];
  const float* hb = B.rb + ((size_t)e * SDX_BODIES + sc.hand_base_body) * 13;
  const float* br = B.root + ((size_t)e * SDX_ACTORS + seg_actor(e)) * 13;
  float cl = prog < 2.0f ? 1e9f : state[2 * e];                                 // a new episode
  float hold = prog < 2.0f ? 1e9f : state[2 * e + 1];
  // the pinch point between thumb and fingers sits (0.125, 0.02, -0.2) from the hand base in the prepare orientation (FK of the scene):
  // the defaults of (xo, yo, zpinch) = (0.125, 0.02, 0.195)
  const float rx = hb[0] - br[0], ry = hb[1] - br[1], rz = hb[2] - br[2];
  const float horiz = sqrtf((rx + xo) * (rx + xo) + (ry + yo) * (ry + yo));
  const float above = horiz > 0.05f ? 0.25f : zpinch;                           // stay above the pile while travelling
  float* a = act + (size_t)e * SDX_NDOF;
  a[0] = clampf(2.5f * (br[0] - xo - hb[0]) / 0.64f, -1.0f, 1.0f);
  a[1] = clampf(2.5f * (br[1] - yo - hb[1]) / 0.64f, -1.0f, 1.0f);
  a[2] = clampf(2.5f * (br[2] + above - hb[2]) / 0.64f, -1.0f, 1.0f);
  // hold the wrist at the prepare pose's orientation (palm down): a[3:6] x 0.2 = orientation error for the IK (GS:1596, OR:1922-1925)
  const float q0x = 0.7107f, q0y = -0.7033f, q0z = 0.0113f, q0w = -0.0091f;
  const float qx = hb[3], qy = hb[4], qz = hb[5], qw = hb[6];
  const float rw = q0w * qw + (q0x * qx + q0y * qy + q0z * qz);                 // q0 * conj(q)
  const float cx = q0y * qz - q0z * qy, cy = q0z * qx - q0x * qz, cz = q0x * qy - q0y * qx;
  const float sg = rw > 0.0f ? 1.0f : (rw < 0.0f ? -1.0f : 0.0f);
  a[3] = clampf(2.0f * (-q0w * qx + qw * q0x - cx) * sg / 0.2f, -1.0f, 1.0f);
  a[4] = clampf(2.0f * (-q0w * qy + qw * q0y - cy) * sg / 0.2f, -1.0f, 1.0f);
  a[5] = clampf(2.0f * (-q0w * qz + qw * q0z - cz) * sg / 0.2f, -1.0f, 1.0f);
  a[6] = 0.0f;
  const bool arrived = horiz < 0.012f && fabsf(rz - zpinch) < 0.012f;
  if (arrived || prog >= 58.0f) cl = fminf(cl, prog);                           // at the latest at step 58
  const float frac = clampf(0.3f + (prog - cl) / pinch_steps * (frac_end - 0.3f), 0.3f, frac_end);
  if (prog >= cl + pinch_steps) hold = fminf(hold, prog);
  if (prog >= hold) { a[0] = 0.0f; a[1] = 0.0f; a[2] = rise; }                  // stop following the brick; raise the hand
  state[2 * e] = cl;
  state[2 * e + 1] = hold;
  for (int j = 7; j < SDX_NDOF; ++j) a[j] = 2.0f * frac - 1.0f;
  a[7] = 0.0f; a[11] = 0.0f; a[15] = 0.0f;                                      // abduction joints of the three fingers stay centred
}
extern "C" void sdxk_scripted_grasp(const SdxConst* C, const SdxBuf* B, float* state, float* act, hipStream_t st) {
  hipLaunchKernelGGL(k_scripted_grasp, dim3((B->N + 255) / 256), dim3(256), 0, st, C, *B, state, act);
}
extern "C" void sdxk_search_set_hand(const SdxConst* C, const SdxBuf* B, const uint8_t* mask, int mode, hipStream_t st) {
  hipLaunchKernelGGL(k_search_set_hand, dim3(B->N), dim3(SDX_WAVE), 0, st, C, *B, mask, mode);
}
extern "C" void sdxk_orient_pregrasp(const SdxConst* C, const SdxBuf* B, const uint8_t* mask, int mode, int iter, hipStream_t st) {
  hipLaunchKernelGGL(k_orient_pregrasp, dim3(B->N), dim3(SDX_WAVE), 0, st, C, *B, mask, mode, iter);
}
extern "C" void sdxk_orient_post_reset(const SdxConst* C, const SdxBuf* B, const uint8_t* mask, hipStream_t st) {
  hipLaunchKernelGGL(k_orient_post_reset, dim3(B->N), dim3(SDX_WAVE), 0, st, C, *B, mask);
}
extern "C" void sdxk_pre_physics(const SdxConst* C, const SdxBuf* B, const float* actions, const uint8_t* mask,
                                 const int32_t* choice, int flags, hipStream_t st) {
  hipLaunchKernelGGL(k_pre_physics, dim3(B->N), dim3(SDX_WAVE), 0, st, C, *B, actions, mask, choice, flags);
}
extern "C" void sdxk_post_physics(const SdxConst* C, const SdxBuf* B, int flags, hipStream_t st) {
  hipLaunchKernelGGL(k_post_physics, dim3(B->N), dim3(SDX_WAVE), 0, st, C, *B, flags);
  hipLaunchKernelGGL(k_tvalue, dim3((B->N + TV_ENVS - 1) / TV_ENVS), dim3(256), 0, st, *B, flags & 1);
  if (B->task_kind == 3 && B->tvt_buf) {   // Search's own T-value: RetriGraspTValue on the ten-frame buffer as it stood before this step
    const int N = B->N;
    const float* W1 = B->tvt_w;
    const float* b1 = W1 + (size_t)1024 * TVT_STRIDE;
    const float* W2 = b1 + 1024;
    const float* b2 = W2 + (size_t)512 * 1024;
    const float* W3 = b2 + 512;
    const float* b3 = W3 + (size_t)128 * 512;
    const float* W4 = b3 + 128;
    const float* b4 = W4 + 2 * 128;
    float* h1 = B->tvt_h;
    float* h2 = h1 + (size_t)N * 1024;
    float* h3 = h2 + (size_t)N * 512;
    float* o = h3 + (size_t)N * 128;
    sdxpk_linear(B->tvt_buf, W1, b1, h1, N, 1024, TVT_STRIDE, 1, nullptr, nullptr, st);
    sdxpk_linear(h1, W2, b2, h2, N, 512, 1024, 1, nullptr, nullptr, st);
    sdxpk_linear(h2, W3, b3, h3, N, 128, 512, 1, nullptr, nullptr, st);
    sdxpk_linear(h3, W4, b4, o, N, 2, 128, 1, nullptr, nullptr, st);
    hipLaunchKernelGGL(k_search_tvalue_out, dim3((N + 255) / 256), dim3(256), 0, st, *B);
    hipLaunchKernelGGL(k_search_tvalue_append, dim3(N), dim3(SDX_WAVE), 0, st, *B);
  }
}
