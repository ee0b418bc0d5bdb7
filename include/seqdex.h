/* seqdex.h — C ABI of libseqdex_hip.so: the MI355X-native replacement for the two
 * third-party engines on the BlockAssemblyGraspSim hot path of sequential-dexterity/SeqDex.
 *
 *   SIM  (sdx_*) : what the task calls on Isaac Gym's tensor API          (SURVEY.md §8(b) seam 2)
 *   TASK (sdx_*) : the reference-owned per-step tensor code, fused on GPU  (§8(a) rows T1-T9)
 *   PPO  (sdxp_*): what rl_games' A2CAgent does per epoch                 (§8(a) rows R1-R9)
 *
 * Citations are file:line under /root/reference/dexteroushandenvs/:
 *   GS = tasks/block_assembly/allegro_hand_block_assembly_grasp_sim.py
 *   BT = tasks/hand_base/base_task.py     VR = tasks/hand_base/vec_task_rlgames.py
 *   PS = policy_sequencing/policy_seq_runner.py   RC = utils/rl_games_custom.py
 *   YG = cfg/lego/ppo_continuous_grasp.yaml
 *
 * Conventions: every function returns 0 on success or a negative sdx_status; no exceptions cross the
 * boundary; handles are opaque; every buffer returned by *_tensor() is device memory owned by the
 * library, stable until *_destroy(); all device work is enqueued on the caller's hipStream_t (passed as
 * void* so that this header needs no HIP include) and is stream-ordered with no hidden synchronisation
 * unless the function's comment says "blocking".  One host thread per handle.  Quaternions are xyzw.
 */
#ifndef SEQDEX_H
#define SEQDEX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDX_ABI_VERSION 8

/* ---- fixed scene dimensions of BlockAssemblyGraspSim (GS:523-1058) ---- */
#define SDX_NLINK 24        /* robot bodies after collapse_fixed_joints (GS:543); body 0 is the fixed base  */
#define SDX_NDOF 23         /* 7 arm + 16 hand revolute DOF (GS:561, 580-590)                               */
#define SDX_MAX_RBOX 40     /* robot collision boxes (URDF boxes, slab compounds of the fingertip / palm hulls, bounding boxes of the arm) */
#define SDX_NBRICK 132      /* 72 free (GS:717-746) + 60 fixed floor bricks (GS:748-808)                     */
#define SDX_NFREE 72
#define SDX_NBRICK_TYPES 8  /* GS:706                                                                        */
#define SDX_MAX_STATIC 8    /* static bodies one env sees: table, 5 bin walls, merged brick floor, base plate                        */
#define SDX_MAX_STATIC_TAB 10 /* rows of the static-body table: the 8 above + the two other base-plate variants of InsertSim          */
#define SDX_MAX_STATIC_SUB 112 /* boxes of all static bodies: 7 single boxes + up to 3 base plates of 1 + 16 x 2 boxes (body, studs)  */
#define SDX_MAX_SUB 2       /* boxes of a free brick's collision compound: slabs of its convex hull (GS:717-731: one hull per brick) */
#define SDX_MAX_SUB_HOLLOW 8 /* boxes of the hollow compound (4 walls + the slabs above the cavity) of the brick being inserted    */
#define SDX_ACTORS 142      /* hand, object, goal, table, 5 bin boxes, 132 bricks, base plate                */
#define SDX_BODIES 165      /* 24 hand links + one body per other actor                                      */
#define SDX_ACTOR_BRICK0 9  /* first brick actor inside an env                                               */
#define SDX_BODY_BRICK0 32  /* first brick rigid body inside an env                                          */
#define SDX_NUM_OBS 396     /* 132 x 3 stacked frames (GS:207-209)                                           */
#define SDX_NUM_STATES 564  /* 188 x 3 (GS:204-210)                                                          */
#define SDX_NUM_ACTIONS 23  /* GS:211                                                                        */
#define SDX_OBS_FRAME 132
#define SDX_STATE_FRAME 188
#define SDX_PILE_HARVEST_SLOTS 512 /* Orient's ring of pile states per brick-type group; SDX_PILE_SLOTS=10000 in the environment of
                                       sdx_create gives the reference's length (OR:1485: 10 000 = 549 MB)                          */
#define SDX_TV_LOG_SLOTS 1048576 /* rows of each T-value dataset ring (success / failure): large enough that the runs of the chain
                                  * never wrap it (a wrapped ring's surviving rows depend on the order the slots were claimed in)      */
#define SDX_HARVEST_SLOTS 5001 /* ring of grasp terminal states per brick-type group (GS:1440: index wraps after 5000) */
#define SDX_TV_PARAMS 42562 /* GraspInsertTValue 4-256-128-64-2 weights+biases (terminal_value_function.py:30-46) */
#define SDX_RETRI_TV_PARAMS 1257346 /* RetriGraspTValue 650-1024-512-128-2 weights+biases (terminal_value_function.py:12-28) */
#define SDX_RETRI_TV_IN 650    /* 65 x 10 (SE:397) */

typedef enum {
  SDX_OK = 0,
  SDX_ERR_INVALID = -1,     /* bad argument / null pointer / bad id                     */
  SDX_ERR_HIP = -2,         /* a HIP runtime call failed; see sdx_last_error()          */
  SDX_ERR_NO_DEVICE = -3,   /* no gfx950 device visible: the product path has NO CPU fallback */
  SDX_ERR_STATE = -4,       /* call order violated (e.g. step before initial states)    */
  SDX_ERR_NOMEM = -5
} sdx_status;

typedef enum { SDX_F32 = 0, SDX_I64 = 1, SDX_I32 = 2, SDX_U8 = 3, SDX_F64 = 4, SDX_I16 = 5 } sdx_dtype;

/* Tensor ids for sdx_tensor().  Shapes use N = num_envs. */
typedef enum {
  SDX_T_ROOT = 0,        /* f32 [N*142,13]  acquire_actor_root_state_tensor      GS:237,321              */
  SDX_T_DOF = 1,         /* f32 [N*23,2]    acquire_dof_state_tensor             GS:238,313-316          */
  SDX_T_RB = 2,          /* f32 [N,165,13]  acquire_rigid_body_state_tensor      GS:239,318              */
  SDX_T_CONTACT = 3,     /* f32 [N,165*3]   acquire_net_contact_force_tensor     GS:240,322              */
  SDX_T_JAC_EEF = 4,     /* f32 [N,6,7]     jacobian_tensor[:, 7-1, :, :7]       GS:241,1601             */
  SDX_T_TARGETS = 5,     /* f32 [N,23]      cur_targets == PD targets            GS:329,1638             */
  SDX_T_PREV_TARGETS = 6,/* f32 [N,23]      prev_targets                         GS:328,1636             */
  SDX_T_OBS = 7,         /* f32 [N,396]     task.obs_buf (unclamped)             BT:57, GS:1299-1332     */
  SDX_T_STATES = 8,      /* f32 [N,564]     task.states_buf (unclamped)          BT:59, GS:1220-1280     */
  SDX_T_OBS_CLAMPED = 9, /* f32 [N,396]     clamp(obs_buf,+-5) as VR:171 returns                         */
  SDX_T_STATES_CLAMPED = 10, /* f32 [N,564] clamp(states_buf,+-5) as VR:172 returns                      */
  SDX_T_REW = 11,        /* f32 [N]         task.rew_buf                         BT:61, GS:1061          */
  SDX_T_RESET = 12,      /* i64 [N]         task.reset_buf (initialised to 1)    BT:63                   */
  SDX_T_PROGRESS = 13,   /* i64 [N]         task.progress_buf                    BT:65                   */
  SDX_T_RANDOMIZE = 14,  /* i64 [N]         task.randomize_buf                   BT:67, GS:1642          */
  SDX_T_ACTIONS = 15,    /* f32 [N,23]      task.actions (clamped copy)          GS:1570, VR:166         */
  SDX_T_INIT_POS = 16,   /* f32 [N,3]       segmentation_target_init_pos         GS:453,1547             */
  SDX_T_INIT_ROT = 17,   /* f32 [N,4]       segmentation_target_init_rot         GS:454,1548             */
  SDX_T_SUCCESSES = 18,  /* f32 [N]         successes                            GS:337,1552             */
  SDX_T_META_REW = 19,   /* f32 [N]         meta_rew_buf                         GS:367,1069,1553        */
  SDX_T_CONS_SUCCESSES = 20, /* f32 [1]     consecutive_successes                GS:338,1771-1774        */
  SDX_T_FINGER_DIST = 21,/* f32 [N]         arm_hand_finger_dist                 GS:1164-1165            */
  SDX_T_TVALUE = 22,     /* f32 [N]         tvalue                               GS:1200-1201            */
  SDX_T_ARM_CONTACTS = 23,/* f32 [N,6]      contacts (bodies 1..6 >= 0.1 N)      GS:1159-1162            */
  SDX_T_STUDENT_OBS = 24,/* f32 [N,30]      extras["student_obs_buf"]            GS:458                  */
  SDX_T_SUCCESS_BUF = 25,/* i64 [N]         extras["success_buf"]                GS:459                  */
  SDX_T_PILE_CHOICE = 26,/* i32 [N]         saved-pile index drawn at the last reset of each env  GS:1510 */
  SDX_T_NCONTACTS = 27,  /* i32 [N]         contact points generated in the last substep (diagnostic)    */
  SDX_T_DEBUG = 28,      /* i64 [64 + 2N]   profiling build only: [0, 64) phase time stamps (s_memtime) of env SDX_DEBUG_ENV in the last k_physics, then (entry, exit) stamps of every env */
  SDX_T_HARVEST_HAND = 29, /* f32 [8,5001,23,2] saved_grasp_hand_ternimal_states per brick-type group   GS:391-417 */
  SDX_T_HARVEST_OBJ = 30,  /* f32 [8,5001,13]   saved_grasp_object_ternimal_states                       GS:391-417 */
  SDX_T_HARVEST_COUNT = 31,/* i32 [8]           terminal states harvested so far (ring index = count % 5001) GS:1417,1440 */
  SDX_T_INSERT_AUX = 32,   /* f32 [N,8]         InsertSim: [0:3] rot_err (IS:1539), [3] |brick - site|, [4] rot_dist (IS:1656-1660) */
  SDX_T_TV_SUCCESS = 33,   /* f32 [SDX_TV_LOG_SLOTS,4] camera-frame target quaternions of successful episode ends (ring)   GS:1404-1412, IS:1392-1400 */
  SDX_T_TV_FAILURE = 34,   /* f32 [SDX_TV_LOG_SLOTS,4] ... of failed ones                                                  GS:1420-1438, IS:1401-1410 */
  SDX_T_TV_COUNT = 35,     /* i32 [2]           rows logged so far: [success, failure] (ring index = count % SDX_TV_LOG_SLOTS)                      */
  SDX_T_PILE_HARVEST = 36, /* f32 [8,S,132,13]  Orient: brick states of finished episodes that left the target brick reachable (ring per
                            *                    brick-type group; S = 512 for task_kind 1 (Orient) and 3 (Search, SE:1398-1420), else 1) = the saved piles the next task starts from    OR:1463-1488, GS:412-413 */
  SDX_T_PILE_HARVEST_COUNT = 37, /* i32 [8]     pile states harvested so far (ring index = count % S)                                    */
  SDX_T_SEG_IMAGE = 38,    /* i16 [N,128,128]   Search: segmentation image of the last render (0 = background / robot / bin, i+1 = brick i)  SE:877 */
  SDX_T_SEG_PIXELS = 39,   /* f32 [N,4]         Search: target pixel count, centroid row, centroid column, count of the render before   SE:1232-1241 */
  SDX_T_EMERGENCE = 40,    /* f32 [N]           Search: extras["emergence_reward"] = 5 x change of the pixel count                    SE:1640-1646 */
  SDX_T_JACOBIAN = 41,     /* f32 [N,23,6,23]   acquire_jacobian_tensor(sim, "hand") of a fixed-base articulation: link k+1 is row k, rows 0..2 linear,
                            *                    3..5 angular; the task reads [:, 7-1, :, :7] (GS:241,1601).  Refreshed by sdx_refresh_kinematics()
                            *                    (= gym.refresh_jacobian_tensors, GS:1095), NOT by sdx_step/sdx_simulate, which keep SDX_T_JAC_EEF current */
  SDX_T_TVALUE_OBS = 42,   /* f32 [N,652]       Search: t_value_obs_buf, ten 65-number frames (SE:375,1155-1166), newest last; columns 650, 651 are row
                            *                    padding (zeros).  Frame = obs_buf[:, 0:62] with [26:30] = camera-frame target quaternion, then the target's
                            *                    pixel centroid / 128 and pixel count / 100 */
  SDX_T_CONTACT_STATS = 43, /* i32 [4]           since create, in env-SUBSTEPS: [0] the largest number of contact points one env generated in one substep;
                            *                    [1] substeps in which an env still exceeded the per-env capacity (1536) after the rebuild of [2] and lost
                            *                    the excess in enumeration order - must stay 0 for results to mean anything, bench.py and the full-size
                            *                    tests check it; [2] substeps whose contact list was rebuilt without its speculative contacts (samples
                            *                    that neither touch nor penetrate) because it would not fit; [3] substeps in which a pair list overflowed
                            *                    (body pairs > 1535, candidate box pairs > 16384, surviving box pairs > 3072: the excess was not tested, its contacts are MISSING) - like [1] it
                            *                    must stay 0, the full-size tests assert it and bench.py flags it */
  SDX_T_WARM_COUNT = 44,   /* i32 [N]           contacts in each env's warm-start cache (scene.warm_start); the engine clears an
                            *                    env's entry when it resets the env; a caller that teleports bodies by hand may zero it too */
  SDX_T_CAM_ROT = 45,      /* f32 [N,4]         camera_view_segmentation_target_rot: the target brick's quaternion in the camera frame, the input of
                            *                    GraspInsertTValue (GS:1196-1201, OR:1201); written by sdx_compute_observations / sdx_post_physics */
  SDX_T_TV_KEYS = 46,      /* i64 [2,SDX_TV_LOG_SLOTS]  (step << 24 | env) of the append that filled each slot of the success / failure ring.  Ring
                            *                    slots are claimed with atomics, i.e. in hardware order; sorting a ring's rows by these keys gives the
                            *                    order a serial loop over steps and envs produces (what seqdex_amd.sim.ring_rows does) */
  SDX_T_HARVEST_KEYS = 47, /* i64 [8,SDX_HARVEST_SLOTS] the same for the grasp terminal-state rings */
  SDX_T_PILE_HARVEST_KEYS = 48, /* i64 [8,slots] the same for the pile rings of Orient / Search */
  SDX_T_WARM_KEYS = 49,    /* i32 [N,1536]     diagnostic view of the warm-start cache: identity of each cached contact of the last solve (body-pair rank
                            *                    13 bits | box pair 9 | direction 1 | sample 5, age in the 4 bits above), ascending; rows are valid up to
                            *                    SDX_T_WARM_COUNT; [1] when scene.warm_start == 0 (the cold solver keeps no cache) */
  SDX_T_WARM_LAMBDA = 50,  /* f32 [N,3,1536]   the accumulated impulses (normal, two tangents) of those contacts */
  SDX_T_COUNT = 51
} sdx_tensor_id;

/* Compact scene constants (row A0/A1 of SURVEY.md §8(a)); produced by tools/compile_scene.py from the
 * reference's URDF/STL/OBJ assets and shipped as seqdex_amd/scene_data/grasp_sim_scene.json. */
typedef struct {
  int32_t abi_version;                 /* = SDX_ABI_VERSION */
  /* robot kinematic tree */
  float base_pos[3];                   /* GS:624-626 */
  float base_quat[4];
  int32_t parent[SDX_NLINK];           /* -1 for body 0 */
  float joint_pos[SDX_NLINK][3];       /* joint frame origin in the parent body frame */
  float joint_quat[SDX_NLINK][4];      /* joint frame rotation in the parent body frame */
  float joint_axis[SDX_NLINK][3];      /* revolute axis in the child frame */
  float lower[SDX_NDOF], upper[SDX_NDOF];
  float kp[SDX_NDOF], kd[SDX_NDOF], effort[SDX_NDOF], vel_limit[SDX_NDOF], armature[SDX_NDOF]; /* GS:580-590 */
  float link_mass[SDX_NLINK];
  float link_com[SDX_NLINK][3];        /* body frame */
  float link_inertia[SDX_NLINK][6];    /* about COM, body frame: xx yy zz xy xz yz */
  int32_t n_rbox;
  int32_t rbox_link[SDX_MAX_RBOX];
  float rbox_center[SDX_MAX_RBOX][3];
  float rbox_quat[SDX_MAX_RBOX][4];
  float rbox_half[SDX_MAX_RBOX][3];
  /* bricks.  Collision shape of a brick = a COMPOUND of axis-aligned boxes in the brick's frame: the slabs of
   * the mesh's convex hull for a free brick (the reference loads every brick as ONE convex hull, GS:717-731), or - for each env's target
   * brick when seg_hollow != 0 - the hollow compound of the mesh itself (walls, roof, upper part: what V-HACD resolves, IS:698-709),
   * whose underside takes the studs of a base plate.  brick_half / brick_center = the bounding box of either compound (broadphase,
   * segmentation camera); the body's reference point is its centre of mass brick_com. */
  float brick_half[SDX_NBRICK_TYPES][3];
  float brick_center[SDX_NBRICK_TYPES][3];   /* centre of the bounding box in the brick's mesh frame */
  float brick_com[SDX_NBRICK_TYPES][3];      /* centre of mass (centroid of the convex hull) in the mesh frame */
  float brick_mass[SDX_NBRICK_TYPES];        /* 567 x hull volume */
  float brick_inertia[SDX_NBRICK_TYPES][3];  /* diagonal of the hull's inertia tensor about the centre of mass, mesh axes (the xz products
                                              * of the four wedge-shaped types, <= 0.37 Ixx, are dropped) */
  int32_t brick_nsub[SDX_NBRICK_TYPES];
  float brick_sub_center[SDX_NBRICK_TYPES][SDX_MAX_SUB][3];   /* mesh frame */
  float brick_sub_half[SDX_NBRICK_TYPES][SDX_MAX_SUB][3];
  int32_t seg_hollow;                        /* != 0: the target brick of every env collides as its hollow compound */
  int32_t hollow_nsub[SDX_NBRICK_TYPES];
  float hollow_sub_center[SDX_NBRICK_TYPES][SDX_MAX_SUB_HOLLOW][3];
  float hollow_sub_half[SDX_NBRICK_TYPES][SDX_MAX_SUB_HOLLOW][3];
  int32_t brick_type[SDX_NBRICK];
  /* static bodies (world frame, axis aligned).  Rows 0..n_static-1 of the table are what an env sees; a body is a compound of the
   * boxes static_sub_*[first .. first + n) and static_center / static_half is its bounding box (a single-box body: the box itself).
   * The studs of a compound (boxes 1.. of a body) are SAMPLED like a brick's boxes (both directions of a pair), a body box only
   * receives the other shape's samples. */
  int32_t n_static;
  float static_center[SDX_MAX_STATIC_TAB][3];
  float static_half[SDX_MAX_STATIC_TAB][3];
  int32_t static_sub_first[SDX_MAX_STATIC_TAB];
  int32_t static_sub_n[SDX_MAX_STATIC_TAB];
  int32_t n_static_sub;
  float static_sub_center[SDX_MAX_STATIC_SUB][3];
  float static_sub_half[SDX_MAX_STATIC_SUB][3];
  /* default actor states for everything that is not a brick (written into ROOT at create) */
  float object_init_state[13];         /* GS:687-689,927-929 */
  float goal_reset_pos[3];             /* goal_init_state + goal_displacement, GS:694-700,1348 */
  float static_actor_pos[6][3];        /* table + 5 bin walls, actor slots 3..8 */
  float base_plate_pos[3];             /* GS:838 */
  float fixed_brick_pos[SDX_NBRICK - SDX_NFREE][3];
  float free_spawn_pos[SDX_NFREE][3];  /* GS:737-742 */
  float free_spawn_quat[4];
  /* task constants */
  int32_t hand_base_body;              /* 7, GS:355 */
  int32_t fingertip_body[4];           /* link_3.0, link_7.0, link_11.0, link_15.0: ff, mf, rf, th  GS:183-186 */
  float camera_offset_quat[4];         /* GS:887-888 */
  float camera_offset_pos[3];          /* GS:889 */
  float arm_prepare_pose[7];           /* GS:267 */
  float finger_reset_unscaled[16];     /* GS:1531 */
  float insert_pose_a[7], insert_pose_b[7]; /* GS:278,281 */
  float max_episode_length;            /* 150, EG:6 */
  float act_moving_average;            /* 1.0, EG:16 */
  float av_factor;                     /* 0.1, GS:151 */
  float clip_obs, clip_actions;        /* 5, 1: VR:18 */
  /* simulation parameters (CF:188, EG:155-167) and our solver constants */
  float dt;
  int32_t substeps;
  int32_t solver_iters;
  float contact_offset;
  float gravity[3];
  float friction;
  float baumgarte;                     /* position-error feedback factor  */
  float max_depenetration_vel;
  float jacobi_relax;                  /* relaxation on the mass-split Jacobi update */
  float warm_start;                    /* every solve starts from this fraction of the impulses the same contacts
                                        * (pair, direction, sample) ended the previous solve with; 0 = start from zero */
  float warm_age;                      /* the fraction ramps up linearly with the number of consecutive solves a contact has existed and
                                        * reaches warm_start after warm_age of them (0: no ramp; at most 16: the age is a 4-bit saturating counter, sdx_create
                                        * rejects larger values); DESIGN.md section 3.E */
  float robot_angular_damping;         /* asset_options.angular_damping of the arm-hand asset (GS:546: 0.01 1/s): every substep scales the joint
                                        * velocities by (1 - h x damping) - for a chain of revolute joints the joint-space image of PhysX's
                                        * per-link angular damping */
  float grasp_tvalue_gate;             /* BlockAssemblyGraspSim harvests a terminal state only when its transition value exceeds this: 0.8 (GS:1406) */
  float orient_tvalue_gate;            /* BlockAssemblyOrient binarises its transition value at this threshold before anything reads it: 0.99
                                        * (OR:1203); a chain run with an early, not yet confident T-value may lower it (say so when you do) */
  /* which task's per-step tensor code the pre/post-physics kernels run: 0 = BlockAssemblyGraspSim (GS),
   * 1 = BlockAssemblyOrient (OR = tasks/block_assembly/allegro_hand_block_assembly_orient.py; targets/IK OR:1720-1778),
   * 2 = BlockAssemblyInsertSim (IS; position action + fixed wrist orientation IS:1526-1572, 75-number observation IS:1280-1298,
   *     insertion reward IS:1640-1695, reset from harvested grasp states IS:1416-1494),
   * 3 = BlockAssemblySearch (SE; tracking IK 0.24 above the target SE:1565-1575, 62-number observation SE:1220-1230, its own asymmetric
   *     state SE:1168-1218, reward SE:1660-1711, reset with 60 settling steps and a segmentation render SE:1274-1538) */
  int32_t task_kind;
  float target_euler[3];               /* Orient: fixed wrist orientation of the tracking IK, OR:477 */
  float seg_mass_scale;                /* mass (and inertia) factor of each env's target brick: 1 (GS:980-981), 50 in Orient (OR:977) */
  /* task_kind 2 = BlockAssemblyInsertSim (IS = tasks/block_assembly/allegro_hand_block_assembly_insert_sim.py): the base plate is
   * one of 4x4x{1,2,4} chosen by env % 3 (IS:971-977): env e sees row static_var_row[e % 3] of the static-body table in slot
   * `static_var_slot`; -1 = every env sees rows 0..n_static-1 as they are. */
  int32_t static_var_slot;
  int32_t static_var_row[3];
  /* task_kind 3 = BlockAssemblySearch (SE = tasks/block_assembly/allegro_hand_block_assembly_search.py): the fixed segmentation camera
   * (128 x 128; set_camera_location SE:875; Isaac Gym's default horizontal field of view 90 degrees) */
  float seg_cam_pos[3];
  float seg_cam_target[3];
  float seg_cam_hfov_deg;
  float search_default_arm[7];         /* arm_hand_default_dof_pos[:7]: hand parked out of the camera's view, SE:203 */
  float search_finger_pose[16];        /* finger joints (radians) of the default AND the prepare pose, SE:205-206,220-222 */
} sdx_scene_desc;

typedef struct sdx_sim* sdx_handle;

/* Scene build: replaces create_sim/add_ground/load_asset/create_env/create_actor/prepare_sim
 * (GS:505-1058, BT:83-84).  Blocking.  Fails with SDX_ERR_NO_DEVICE when no GPU is present. */
int sdx_create(const sdx_scene_desc* scene, int32_t num_envs, int32_t device, uint64_t seed, sdx_handle* out);
int sdx_destroy(sdx_handle h);

/* acquire_*_tensor + gymtorch.wrap_tensor (GS:237-241,313-322) and the BaseTask buffers (BT:57-68). */
int sdx_tensor(sdx_handle h, int32_t id, void** dev_ptr, int64_t shape[4], int32_t* ndim, int32_t* dtype);

/* Saved pile states: replaces pickle.load(".../saved_searching_ternimal_states_good_mo_tvalue.pkl")
 * (GS:412-413): host float32 [8, K, 132, 13].  Blocking copy to HBM. */
int sdx_load_initial_states(sdx_handle h, const float* piles_host, int32_t K);

/* GraspInsertTValue parameters (GS:417-419): host float32[SDX_TV_PARAMS] packed as
 * W1[256,4] b1[256] W2[128,256] b2[128] W3[64,128] b3[64] W4[2,64] b4[2].  Blocking. */
int sdx_set_tvalue_weights(sdx_handle h, const float* params_host, int32_t n);

/* BlockAssemblySearch only (task_kind 3): RetriGraspTValue(650, 2) parameters (SE:395-410): host float32[SDX_RETRI_TV_PARAMS] packed as
 * W1[1024,650] b1[1024] W2[512,1024] b2[512] W3[128,512] b3[128] W4[2,128] b4[2] (torch layout W[out][in]).  Every step the task
 * evaluates it on SDX_T_TVALUE_OBS as it stood BEFORE this step's frame is appended and stores sigmoid(out)[:, 1] in SDX_T_TVALUE
 * (SE:1133-1134); like the reference's, the value reaches compute_hand_reward but does not enter the reward (SE:1687).  Blocking. */
int sdx_set_retri_tvalue_weights(sdx_handle h, const float* params_host, int32_t n);

/* BaseTask.step(actions) (BT:130-150) fused: pre_physics_step (GS:1555-1638, device-side masked reset
 * instead of reset_buf.nonzero()) -> simulate x controlFreqInv (BT:138-140) -> post_physics_step
 * (GS:1640-1658).  actions: device f32 [N,23], clamped to +-clip_actions inside (VR:166). */
int sdx_step(sdx_handle h, const float* actions_dev, void* stream);

/* The three stages of sdx_step as separate entry points (used by the parity tests and by callers that keep
 * reference-style task code on top of the simulator). */
int sdx_pre_physics(sdx_handle h, const float* actions_dev, void* stream);   /* GS:1555-1638 */
int sdx_simulate(sdx_handle h, void* stream);                                /* BT:140 + refresh_* GS:1091-1095 */
int sdx_post_physics(sdx_handle h, void* stream);                            /* GS:1640-1645 */

/* compute_observations() only (GS:1090-1218): obs/states frames + stacking, no progress++/reward. */
int sdx_compute_observations(sdx_handle h, void* stream);

/* reset_idx(env_ids) (GS:1361-1553) for envs with env_mask_dev[i] != 0 (device u8 [N]).
 * pile_choice_dev: device i32 [N] saved-pile index per env, or NULL to draw it from the counter RNG
 * (the reference uses python random.sample, GS:1510). */
int sdx_reset_idx(sdx_handle h, const uint8_t* env_mask_dev, const int32_t* pile_choice_dev, void* stream);

/* refresh_rigid_body_state / jacobian after the caller overwrote DOF/ROOT through the tensor views
 * (set_dof_state_tensor_indexed / set_actor_root_state_tensor_indexed, GS:1514,1543): recomputes FK,
 * link velocities and the end-effector Jacobian from SDX_T_DOF. */
/* Search: render the segmentation camera of every env from the current ROOT / RB states -> SDX_T_SEG_IMAGE, SDX_T_SEG_PIXELS,
 * SDX_T_EMERGENCE (gym.render_all_camera_sensors + the pixel statistics of SE:1232-1241,1640-1646).  task_kind 3 only. */
int sdx_render_segmentation(sdx_handle h, void* stream);
int sdx_refresh_kinematics(sdx_handle h, void* stream);

/* set_actor_root_state_tensor_indexed / set_dof_state_tensor_indexed / set_dof_position_target_tensor_indexed
 * (GS:1355,1514,1539,1543; gymtorch.unwrap_tensor(...) + int32 sim-domain actor indices): applies the rows of `src_dev` that belong to
 * the listed actors.  id selects the state: SDX_T_ROOT (src f32 [N*142,13]: rows of the listed actors -> SDX_T_ROOT and the actor's
 * body row of SDX_T_RB; the robot's fixed base ignores it), SDX_T_DOF (src f32 [N*23,2]) or SDX_T_TARGETS (src f32 [N,23]): the
 * listed actors must be hand actors (slot 0 of an env, index env*142) and all 23 rows of that env are taken; SDX_T_DOF also refreshes
 * the link states and Jacobians of every env (sdx_refresh_kinematics).  src_dev may be the library's own tensor (the caller edited
 * it in place, as the reference does with root_state_tensor / dof_state) or a tensor of the caller's (as the reference's cur_targets).
 * actor_ids_dev: device i32 [n]; actor index = env * SDX_ACTORS + slot.  Out-of-range ids are ignored. */
int sdx_set_indexed(sdx_handle h, int32_t id, const float* src_dev, const int32_t* actor_ids_dev, int32_t n, void* stream);

int sdx_num_envs(sdx_handle h);
const char* sdx_last_error(sdx_handle h);   /* never NULL; h may be NULL for create-time errors */

/* ======================================================================================= PPO */

typedef struct {
  int32_t num_actors;        /* N envs (TR:81-85 injects num_actors)                     */
  int32_t horizon;           /* 8    YG:49 */
  int32_t minibatch;         /* 4    YG:50 (must divide horizon*num_actors)              */
  int32_t mini_epochs;       /* 5    YG:51 */
  int32_t cv_minibatch;      /* 4    YG:75 */
  int32_t cv_mini_epochs;    /* 5    YG:76 */
  int32_t obs_dim;           /* 396 */
  int32_t state_dim;         /* 564 */
  int32_t act_dim;           /* 23  */
  int32_t units[3];          /* 1024, 512, 256   YG:23 */
  float gamma, tau;          /* 0.99, 0.95  YG:35-36 */
  float lr, cv_lr;           /* 3e-4 YG:38, 1e-3 YG:77 */
  float e_clip;              /* 0.1  YG:46 */
  float grad_norm;           /* 1.0  YG:42 */
  float critic_coef;         /* 1    YG:52 */
  float entropy_coef;        /* 0    YG:43 */
  float bounds_loss_coef;    /* 1e-3 YG:63 */
  float kl_threshold;        /* 0.02 YG:54 */
  int32_t clip_value;        /* 1    YG:47 */
  int32_t truncate_grads;    /* 1    YG:44 */
  int32_t normalize_advantage; /* 1  YG:34 */
  int32_t cv_normalize_input;  /* 1  YG:82 */
  int32_t adaptive_lr;       /* 1: lr_schedule adaptive, schedule_type legacy (YG:53, PS:306-312) */
  int32_t world_size;        /* data-parallel ranks; gradients are averaged by the CALLER (RCCL) between
                                sdxp_backward_* and sdxp_apply_* when world_size > 1 */
  int32_t obs_cols;          /* columns of the caller's observation rows if fewer than obs_dim (0 = obs_dim): the network input is
                                zero-padded to obs_dim, which must be a multiple of 4 (BlockAssemblyOrient: 186 -> 188) */
  int32_t mixed_precision;   /* rl_games' `mixed_precision` (App. C; False in YG): 1 = the trunk GEMMs of the large-minibatch update path
                                (minibatch_size > 8) run on bf16 MFMA with fp32 accumulation; weights, Adam state, activations in HBM, losses,
                                heads and the rollout stay fp32 (BASELINE.json configs[4]: "bf16 policy").  Ignored by the rank-MB paths */
} sdxp_config;

typedef enum {
  SDXP_T_AC_PARAMS = 0,      /* f32 [P_ac]   actor MLP | mu | logstd | critic MLP | value  (flat)   */
  SDXP_T_AC_GRADS = 1,       /* f32 [P_ac]   flat gradient: the buffer RCCL all-reduces             */
  SDXP_T_CV_PARAMS = 2,      /* f32 [P_cv]   central value MLP (flat)                               */
  SDXP_T_CV_GRADS = 3,       /* f32 [P_cv]                                                          */
  /* experience buffer, stored env-major [N,H,...] == swap_and_flatten01 order (PS:338-339) */
  SDXP_T_MB_OBS = 4,         /* f32 [N,H,obs]     experience_buffer 'obses'   PS:346                */
  SDXP_T_MB_STATES = 5,      /* f32 [N,H,state]   'states'                    PS:352                */
  SDXP_T_MB_ACTIONS = 6,     /* f32 [N,H,act]                                                       */
  SDXP_T_MB_MUS = 7,         /* f32 [N,H,act]     overwritten by update_mu_sigma (RC:1358)          */
  SDXP_T_MB_SIGMAS = 8,      /* f32 [N,H,act]                                                       */
  SDXP_T_MB_NEGLOGP = 9,     /* f32 [N,H]                                                           */
  SDXP_T_MB_VALUES = 10,     /* f32 [N,H]                                                           */
  SDXP_T_MB_REWARDS = 11,    /* f32 [N,H]                                                           */
  SDXP_T_MB_DONES = 12,      /* f32 [N,H]         dones stored BEFORE the step, PS:347              */
  SDXP_T_RETURNS = 13,       /* f32 [N*H]   env-major after swap_and_flatten01, PS:338-339          */
  SDXP_T_ADVANTAGES = 14,    /* f32 [N*H]   normalised, RC:1645-1651                                */
  SDXP_T_CV_RMS_MEAN = 15,   /* f64 [state] running mean of the central-value input normalisation   */
  SDXP_T_CV_RMS_VAR = 16,    /* f64 [state]                                                         */
  SDXP_T_STATS = 17,         /* raw control block (struct SdxpCtrl, csrc/sdxp_types.h) as f32 words  */
  SDXP_T_LAST_VALUES = 18,   /* f32 [N]                                                             */
  SDXP_T_AC_ADAM_M = 19, SDXP_T_AC_ADAM_V = 20, SDXP_T_CV_ADAM_M = 21, SDXP_T_CV_ADAM_V = 22,
  SDXP_T_DEBUG = 23,         /* i64 [64]    phase time stamps of the last HEAD / CTRL kernels (profiling aid)      */
  SDXP_T_ALL_GRADS = 24,     /* f32 [..]    one buffer over AC_GRADS | pad | CV_GRADS | pad | KL word | pad (multiples of 64): a
                              * multi-rank caller all-reduces THIS once per optimiser step and calls sdxp_apply(which, -INFINITY) */
  SDXP_T_FACTORS = 25,       /* f32 [F]     this rank's rank-MB factors of the current minibatch (sdxp_backward_factors)       */
  SDXP_T_FACTORS_ALL = 26,   /* f32 [world,F] all ranks' factors: the caller all-gathers SDXP_T_FACTORS into it              */
  SDXP_T_COUNT = 27
} sdxp_tensor_id;

typedef struct sdxp_agent* sdxp_handle;

/* A2CAgent.__init__ + network build (R1,R2): torch-default Linear init (kaiming_uniform a=sqrt5) drawn from a
 * counter RNG under `seed`, biases zero, logstd zero (YG:8-29, App. C).  Blocking.
 * Shapes accepted (SDX_ERR_INVALID with a message otherwise): obs_dim and state_dim multiples of 4 in [4, 1024] (the rollout's first
 * trunk layer normalises through 1 024-entry tables, its dataset copy moves a row as at most 256 16-byte pieces), units[0], units[1]
 * multiples of 4, units[2] == 256, act_dim <= 32, batch_size % minibatch == 0, equal minibatch / mini_epochs for both optimisers. */
int sdxp_create(const sdxp_config* cfg, int32_t device, uint64_t seed, sdxp_handle* out);
int sdxp_destroy(sdxp_handle h);
int sdxp_tensor(sdxp_handle h, int32_t id, void** dev_ptr, int64_t shape[4], int32_t* ndim, int32_t* dtype);
int64_t sdxp_param_count(sdxp_handle h, int32_t which /*0 actor-critic, 1 central value*/);

/* get_action_values (RC:1697-1723): actor forward, a = mu + sigma*eps (eps from the counter RNG, or from
 * eps_dev f32 [N,act] when non-NULL), neglogp (RC:2114-2126), central value on states; stores row t of the
 * experience buffer (PS:345-352): obs, states, actions, mus, sigmas, neglogp, values, dones_dev (the task's i64
 * reset_buf as returned by the PREVIOUS env step, PS:347; NULL = 0).
 * actions_out_dev f32 [N,act] receives the sampled (unclamped) actions. */
int sdxp_act(sdxp_handle h, int32_t t, const float* obs_dev, const float* states_dev, const int64_t* dones_dev,
             const float* eps_dev, float* actions_out_dev, void* stream);
/* post_step (PS:355-373): rewards row t (reward_shaper scale 1, YG:32-33) + episode statistics
 * (current_rewards/current_lengths, game_rewards/game_lengths) from the i64 dones returned by THIS env step. */
int sdxp_store_rewards(sdxp_handle h, int32_t t, const float* rew_dev, const int64_t* dones_after_dev, void* stream);
/* play_steps tail (PS:329-339) + prepare_dataset (RC:1639-1651): last_values = V(states), GAE (R5), returns,
 * flatten env-major, advantage normalisation with unbiased std (R6); updates the CV running mean/std. */
int sdxp_finish_rollout(sdxp_handle h, const float* last_states_dev, const int64_t* last_dones_dev, void* stream);
/* The three stages of sdxp_finish_rollout one by one, for callers that keep rl_games-style driver code (PS:329-343):
 * get_values (RC:1725-1745; the central value of `states`, f32 [N] out), discount_values (PS:331-336; raw advantages ->
 * SDXP_T_ADVANTAGES, returns -> SDXP_T_RETURNS, both env-major), prepare_dataset (RC:1645-1651; advantage normalisation in place). */
int sdxp_get_values(sdxp_handle h, const float* states_dev, float* values_out_dev, void* stream);
int sdxp_discount_values(sdxp_handle h, const float* last_values_dev, const int64_t* last_dones_dev, void* stream);
int sdxp_prepare_dataset(sdxp_handle h, void* stream);
/* train_central_value + the actor-critic minibatch loop of train_epoch (PS:294-326, RC:1339-1365): all
 * mini-epochs, contiguous unshuffled minibatches, loss R7, grad-norm clip, Adam, legacy adaptive LR after
 * every minibatch (R8).  Single-rank fast path: everything stays on the device. */
int sdxp_update(sdxp_handle h, void* stream);
/* Which implementation sdxp_update runs on this handle: 1 = persistent kernel (one launch per epoch, weights and Adam moments
 * resident in the VGPR files of 256 CUs; needs the shipped shapes, minibatch_size 4 and a 256-CU device), 0 = hipGraph of the
 * multi-kernel optimiser step (any minibatch_size in {2,4,8}; forced with SDXP_UPDATE_IMPL=graph), 2 = GEMM-shaped step for
 * minibatch_size > 8 (cfg/lego/ppo_continuous_insert.yaml: 4096): fp32-MFMA forward / data-gradient / weight-gradient GEMMs,
 * explicit flat gradients, clip_grad_norm_ + Adam (sdxp_bigmb.hip). */
int sdxp_update_impl(sdxp_handle h);
/* Waits for the update launched on `stream`.  SDX_OK, or SDX_ERR_STATE when the persistent kernel timed out (its 256 workgroups were
 * not co-resident): nothing was applied, the inputs it had touched are restored and the handle has switched to the hipGraph
 * path - call sdxp_update again to repeat the epoch.  Optional after the hipGraph path (always SDX_OK there). */
int sdxp_update_status(sdxp_handle h, void* stream);
/* Multi-rank path, one minibatch at a time so that the caller can all-reduce SDXP_T_*_GRADS in between:
 * which = 0 actor-critic, 1 central value; mb = minibatch index within the epoch. */
int sdxp_backward(sdxp_handle h, int32_t which, int32_t mb, void* stream);
/* Factor-exchange form of the multi-rank step (preferred: 194 KB all-gather instead of a 13.4 MB all-reduce per step):
 * sdxp_backward_factors(mb) -> all_gather(SDXP_T_FACTORS_ALL, SDXP_T_FACTORS) -> sdxp_grads_from_factors -> sdxp_apply(0, -INFINITY),
 * sdxp_apply(1).  The *_GRADS buffers then hold the same rank SUM an all-reduce would have produced. */
int sdxp_backward_factors(sdxp_handle h, int32_t mb, void* stream);
int sdxp_grads_from_factors(sdxp_handle h, void* stream);
/* = sdxp_grads_from_factors + sdxp_apply(0, -INFINITY) + sdxp_apply(1), fused into three launches */
int sdxp_apply_factors(sdxp_handle h, void* stream);
/* kl: the rank-averaged KL for the LR schedule; NaN = take SdxpCtrl.last_kl that the caller all-reduced (SUM) in place through
 * SDXP_T_STATS; -INFINITY = take the KL word of SDXP_T_ALL_GRADS that the caller all-reduced (SUM) with the gradients. */
int sdxp_apply(sdxp_handle h, int32_t which, float kl_allreduced_or_nan, void* stream);
/* Optimiser state kept in the device control block, for checkpoints (rl_games restores the same items: the Adam step counters of
 * optimizer.state_dict(), running_mean_std.count of the central value, last_lr; A2CBase.set_full_state_weights).  Both calls block.
 * sdxp_set_state also re-derives the bias-correction powers 0.9^t / 0.999^t. */
typedef struct {
  double rms_count;          /* RunningMeanStd.count of the central-value input normalisation (1.0 after create) */
  int32_t ac_t, cv_t;        /* Adam step counters of the actor-critic / central-value optimisers                 */
  float ac_lr, cv_lr;        /* current learning rates (ac_lr moves with the adaptive schedule, PS:306-312)       */
} sdxp_opt_state;
int sdxp_get_state(sdxp_handle h, sdxp_opt_state* out, void* stream);
int sdxp_set_state(sdxp_handle h, const sdxp_opt_state* in, void* stream);
const char* sdxp_last_error(sdxp_handle h);

/* ------------------------------------------------------------------------------------------------------------------------------
 * Transition-value trainer (policy_sequencing/transition_value_trainer.py:127-248 = TT): fits GraspInsertTValue
 * (terminal_value_function.py:30-46; parameter packing as sdx_set_tvalue_weights) to success / failure camera-frame quaternions.
 * The datasets are device arrays [n, 4] - e.g. the SDX_T_TV_SUCCESS / SDX_T_TV_FAILURE rings the task kernels fill at episode ends
 * (the reference writes them to HDF5 groups data/success_dataset, data/failure_dataset: GS:470-480,1404-1432, IS:1392-1410). */
typedef struct sdxtv_trainer* sdxtv_handle;
typedef enum {
  SDXTV_T_PARAMS = 0,    /* f32 [42562]  W1 b1 W2 b2 W3 b3 W4 b4, torch layout W[out][in]                      TT:181      */
  SDXTV_T_GRADS = 1,     /* f32 [42562]  gradient of the last sdxtv_step                                                    */
  SDXTV_T_ADAM_M = 2,    /* f32 [42562]                                                                         TT:187      */
  SDXTV_T_ADAM_V = 3,    /* f32 [42562]                                                                                     */
  SDXTV_T_BATCH = 4,     /* f32 [B, 4]   t_value_obs_buf: rows [0, B/2) success, [B/2, B) failure                TT:207,213-222 */
  SDXTV_T_LOSS = 5,      /* f32 [1]      BCEWithLogitsLoss of the last step                                     TT:228      */
  SDXTV_T_OUTPUT = 6,    /* f32 [B, 2]   predict_success_confident of the last step                             TT:225      */
  SDXTV_T_COUNT = 7
} sdxtv_tensor_id;
int sdxtv_create(int32_t batch /* 1024, TT:190 */, int32_t device, uint64_t seed, sdxtv_handle* out);
int sdxtv_destroy(sdxtv_handle h);
int sdxtv_tensor(sdxtv_handle h, int32_t id, void** dev_ptr, int64_t shape[4], int32_t* ndim, int32_t* dtype);
/* draw a batch (TT:211-222): B/2 random success rows and B/2 random failure rows, + U(-1,1) * 0.05, renormalised -> SDXTV_T_BATCH */
int sdxtv_sample(sdxtv_handle h, const float* succ_dev, int32_t n_succ, const float* fail_dev, int32_t n_fail, void* stream);
/* forward, BCEWithLogitsLoss against the one-hot [failure, success] labels, backward, Adam step (TT:225-231) on SDXTV_T_BATCH */
int sdxtv_step(sdxtv_handle h, float lr /* 1e-3 */, void* stream);
/* iters x (sdxtv_sample, sdxtv_step): train_rollout (TT:209-231) */
int sdxtv_train(sdxtv_handle h, const float* succ_dev, int32_t n_succ, const float* fail_dev, int32_t n_fail, int32_t iters, float lr,
                void* stream);
/* net(x) for n <= batch rows -> out_dev [n, 2] (validation pass TT:235-246) */
int sdxtv_predict(sdxtv_handle h, const float* x_dev, int32_t n, float* out_dev, void* stream);
const char* sdxtv_last_error(sdxtv_handle h);

#ifdef __cplusplus
}
#endif
#endif /* SEQDEX_H */
