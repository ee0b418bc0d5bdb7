"""ctypes mirror of include/seqdex.h (struct layouts, enums, function prototypes) and the loader of
libseqdex_hip.so.  There is NO CPU fallback: if the library is missing, or no GPU is visible when a handle
is created, the product path raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDX_LIB_PATH") or os.path.join(HERE, "lib", "libseqdex_hip.so")   # SDX_LIB_PATH: an experimental build of the same library (kernel A/B timing)

SDX_ABI_VERSION = 8
NLINK, NDOF, MAX_RBOX, NBRICK, NFREE, NBRICK_TYPES, MAX_STATIC = 24, 23, 40, 132, 72, 8, 8
MAX_STATIC_TAB, MAX_STATIC_SUB, MAX_SUB, MAX_SUB_HOLLOW = 10, 112, 2, 8
ACTORS, BODIES, ACTOR_BRICK0, BODY_BRICK0 = 142, 165, 9, 32
NUM_OBS, NUM_STATES, NUM_ACTIONS, OBS_FRAME, STATE_FRAME = 396, 564, 23, 132, 188
HARVEST_SLOTS = 5001      # SDX_HARVEST_SLOTS
TV_PARAMS = 42562
RETRI_TV_PARAMS = 1257346   # RetriGraspTValue 650-1024-512-128-2

f32, i32 = C.c_float, C.c_int32


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", i32),
        ("base_pos", f32 * 3), ("base_quat", f32 * 4),
        ("parent", i32 * NLINK),
        ("joint_pos", (f32 * 3) * NLINK), ("joint_quat", (f32 * 4) * NLINK), ("joint_axis", (f32 * 3) * NLINK),
        ("lower", f32 * NDOF), ("upper", f32 * NDOF),
        ("kp", f32 * NDOF), ("kd", f32 * NDOF), ("effort", f32 * NDOF), ("vel_limit", f32 * NDOF),
        ("armature", f32 * NDOF),
        ("link_mass", f32 * NLINK), ("link_com", (f32 * 3) * NLINK), ("link_inertia", (f32 * 6) * NLINK),
        ("n_rbox", i32), ("rbox_link", i32 * MAX_RBOX),
        ("rbox_center", (f32 * 3) * MAX_RBOX), ("rbox_quat", (f32 * 4) * MAX_RBOX), ("rbox_half", (f32 * 3) * MAX_RBOX),
        ("brick_half", (f32 * 3) * NBRICK_TYPES), ("brick_center", (f32 * 3) * NBRICK_TYPES), ("brick_com", (f32 * 3) * NBRICK_TYPES),
        ("brick_mass", f32 * NBRICK_TYPES), ("brick_inertia", (f32 * 3) * NBRICK_TYPES),
        ("brick_nsub", i32 * NBRICK_TYPES), ("brick_sub_center", ((f32 * 3) * MAX_SUB) * NBRICK_TYPES),
        ("brick_sub_half", ((f32 * 3) * MAX_SUB) * NBRICK_TYPES),
        ("seg_hollow", i32), ("hollow_nsub", i32 * NBRICK_TYPES), ("hollow_sub_center", ((f32 * 3) * MAX_SUB_HOLLOW) * NBRICK_TYPES),
        ("hollow_sub_half", ((f32 * 3) * MAX_SUB_HOLLOW) * NBRICK_TYPES),
        ("brick_type", i32 * NBRICK),
        ("n_static", i32), ("static_center", (f32 * 3) * MAX_STATIC_TAB), ("static_half", (f32 * 3) * MAX_STATIC_TAB),
        ("static_sub_first", i32 * MAX_STATIC_TAB), ("static_sub_n", i32 * MAX_STATIC_TAB), ("n_static_sub", i32),
        ("static_sub_center", (f32 * 3) * MAX_STATIC_SUB), ("static_sub_half", (f32 * 3) * MAX_STATIC_SUB),
        ("object_init_state", f32 * 13), ("goal_reset_pos", f32 * 3),
        ("static_actor_pos", (f32 * 3) * 6), ("base_plate_pos", f32 * 3),
        ("fixed_brick_pos", (f32 * 3) * (NBRICK - NFREE)), ("free_spawn_pos", (f32 * 3) * NFREE),
        ("free_spawn_quat", f32 * 4),
        ("hand_base_body", i32), ("fingertip_body", i32 * 4),
        ("camera_offset_quat", f32 * 4), ("camera_offset_pos", f32 * 3),
        ("arm_prepare_pose", f32 * 7), ("finger_reset_unscaled", f32 * 16),
        ("insert_pose_a", f32 * 7), ("insert_pose_b", f32 * 7),
        ("max_episode_length", f32), ("act_moving_average", f32), ("av_factor", f32),
        ("clip_obs", f32), ("clip_actions", f32),
        ("dt", f32), ("substeps", i32), ("solver_iters", i32), ("contact_offset", f32), ("gravity", f32 * 3),
        ("friction", f32), ("baumgarte", f32), ("max_depenetration_vel", f32), ("jacobi_relax", f32), ("warm_start", f32), ("warm_age", f32), ("robot_angular_damping", f32), ("grasp_tvalue_gate", f32), ("orient_tvalue_gate", f32),
        ("task_kind", i32), ("target_euler", f32 * 3), ("seg_mass_scale", f32),
        ("static_var_slot", i32), ("static_var_row", i32 * 3),
        ("seg_cam_pos", f32 * 3), ("seg_cam_target", f32 * 3), ("seg_cam_hfov_deg", f32),
        ("search_default_arm", f32 * 7), ("search_finger_pose", f32 * 16),
    ]


class PPOConfig(C.Structure):
    _fields_ = [
        ("num_actors", i32), ("horizon", i32), ("minibatch", i32), ("mini_epochs", i32),
        ("cv_minibatch", i32), ("cv_mini_epochs", i32), ("obs_dim", i32), ("state_dim", i32), ("act_dim", i32),
        ("units", i32 * 3), ("gamma", f32), ("tau", f32), ("lr", f32), ("cv_lr", f32), ("e_clip", f32),
        ("grad_norm", f32), ("critic_coef", f32), ("entropy_coef", f32), ("bounds_loss_coef", f32),
        ("kl_threshold", f32), ("clip_value", i32), ("truncate_grads", i32), ("normalize_advantage", i32),
        ("cv_normalize_input", i32), ("adaptive_lr", i32), ("world_size", i32), ("obs_cols", i32), ("mixed_precision", i32),
    ]


class OptState(C.Structure):
    """sdxp_opt_state"""
    _fields_ = [("rms_count", C.c_double), ("ac_t", i32), ("cv_t", i32), ("ac_lr", f32), ("cv_lr", f32)]


# tensor ids (sdx_tensor_id)
T = dict(ROOT=0, DOF=1, RB=2, CONTACT=3, JAC_EEF=4, TARGETS=5, PREV_TARGETS=6, OBS=7, STATES=8, OBS_CLAMPED=9,
         STATES_CLAMPED=10, REW=11, RESET=12, PROGRESS=13, RANDOMIZE=14, ACTIONS=15, INIT_POS=16, INIT_ROT=17,
         SUCCESSES=18, META_REW=19, CONS_SUCCESSES=20, FINGER_DIST=21, TVALUE=22, ARM_CONTACTS=23, STUDENT_OBS=24,
         SUCCESS_BUF=25, PILE_CHOICE=26, NCONTACTS=27, DEBUG=28, HARVEST_HAND=29, HARVEST_OBJ=30,
         HARVEST_COUNT=31, INSERT_AUX=32, TV_SUCCESS=33, TV_FAILURE=34, TV_COUNT=35, PILE_HARVEST=36, PILE_HARVEST_COUNT=37, SEG_IMAGE=38, SEG_PIXELS=39, EMERGENCE=40, JACOBIAN=41, TVALUE_OBS=42, CONTACT_STATS=43, WARM_COUNT=44, CAM_ROT=45, TV_KEYS=46, HARVEST_KEYS=47, PILE_HARVEST_KEYS=48, WARM_KEYS=49, WARM_LAMBDA=50)
# sdxp_tensor_id
TP = dict(AC_PARAMS=0, AC_GRADS=1, CV_PARAMS=2, CV_GRADS=3, MB_OBS=4, MB_STATES=5, MB_ACTIONS=6, MB_MUS=7,
          MB_SIGMAS=8, MB_NEGLOGP=9, MB_VALUES=10, MB_REWARDS=11, MB_DONES=12, RETURNS=13, ADVANTAGES=14,
          CV_RMS_MEAN=15, CV_RMS_VAR=16, STATS=17, LAST_VALUES=18, AC_ADAM_M=19, AC_ADAM_V=20, CV_ADAM_M=21,
          CV_ADAM_V=22, DEBUG=23, ALL_GRADS=24, FACTORS=25, FACTORS_ALL=26)

SDX_EXPORTS = ["sdx_create", "sdx_destroy", "sdx_tensor", "sdx_load_initial_states", "sdx_set_tvalue_weights", "sdx_set_retri_tvalue_weights",
               "sdx_step", "sdx_pre_physics", "sdx_simulate", "sdx_post_physics", "sdx_compute_observations",
               "sdx_reset_idx", "sdx_set_indexed", "sdx_refresh_kinematics", "sdx_render_segmentation", "sdx_num_envs", "sdx_last_error",
               "sdxp_create", "sdxp_destroy", "sdxp_tensor", "sdxp_param_count", "sdxp_act", "sdxp_store_rewards",
               "sdxp_finish_rollout", "sdxp_get_values", "sdxp_discount_values", "sdxp_prepare_dataset", "sdxp_update", "sdxp_update_impl", "sdxp_update_status", "sdxp_backward", "sdxp_apply", "sdxp_backward_factors",
               "sdxp_grads_from_factors", "sdxp_apply_factors", "sdxp_get_state", "sdxp_set_state", "sdxp_last_error",
               "sdxtv_create", "sdxtv_destroy", "sdxtv_tensor", "sdxtv_sample", "sdxtv_step", "sdxtv_train", "sdxtv_predict",
               "sdxtv_last_error"]

_lib = None


def load_library():
    """dlopen libseqdex_hip.so (built by `make -C seqdex_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("seqdex_amd: %s is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i64p, i32p = C.c_void_p, C.POINTER(C.c_int64), C.POINTER(i32)
    lib.sdx_create.argtypes = [C.POINTER(SceneDesc), i32, i32, C.c_uint64, C.POINTER(vp)]
    lib.sdx_destroy.argtypes = [vp]
    lib.sdx_tensor.argtypes = [vp, i32, C.POINTER(vp), i64p, i32p, i32p]
    lib.sdx_load_initial_states.argtypes = [vp, vp, i32]
    lib.sdx_set_tvalue_weights.argtypes = [vp, vp, i32]
    lib.sdx_set_retri_tvalue_weights.argtypes = [vp, vp, i32]
    for n in ["sdx_step", "sdx_pre_physics"]:
        getattr(lib, n).argtypes = [vp, vp, vp]
    for n in ["sdx_simulate", "sdx_post_physics", "sdx_compute_observations", "sdx_refresh_kinematics", "sdx_render_segmentation"]:
        getattr(lib, n).argtypes = [vp, vp]
    lib.sdx_reset_idx.argtypes = [vp, vp, vp, vp]
    lib.sdx_set_indexed.argtypes = [vp, i32, vp, vp, i32, vp]
    lib.sdx_num_envs.argtypes = [vp]
    lib.sdx_last_error.argtypes = [vp]
    lib.sdx_last_error.restype = C.c_char_p
    missing = [n for n in SDX_EXPORTS if not hasattr(lib, n)]
    if missing:
        raise RuntimeError("libseqdex_hip.so is stale, missing %s; rebuild with __graft_entry__.build()" % missing)
    lib.sdxp_create.argtypes = [C.POINTER(PPOConfig), i32, C.c_uint64, C.POINTER(vp)]
    lib.sdxp_destroy.argtypes = [vp]
    lib.sdxp_tensor.argtypes = [vp, i32, C.POINTER(vp), i64p, i32p, i32p]
    lib.sdxp_param_count.argtypes = [vp, i32]
    lib.sdxp_param_count.restype = C.c_int64
    lib.sdxp_act.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    lib.sdxp_store_rewards.argtypes = [vp, i32, vp, vp, vp]
    lib.sdxp_finish_rollout.argtypes = [vp, vp, vp, vp]
    lib.sdxp_get_values.argtypes = [vp, vp, vp, vp]
    lib.sdxp_discount_values.argtypes = [vp, vp, vp, vp]
    lib.sdxp_prepare_dataset.argtypes = [vp, vp]
    lib.sdxtv_create.argtypes = [i32, i32, C.c_uint64, C.POINTER(vp)]
    lib.sdxtv_destroy.argtypes = [vp]
    lib.sdxtv_tensor.argtypes = [vp, i32, C.POINTER(vp), i64p, i32p, i32p]
    lib.sdxtv_sample.argtypes = [vp, vp, i32, vp, i32, vp]
    lib.sdxtv_step.argtypes = [vp, C.c_float, vp]
    lib.sdxtv_train.argtypes = [vp, vp, i32, vp, i32, i32, C.c_float, vp]
    lib.sdxtv_predict.argtypes = [vp, vp, i32, vp, vp]
    lib.sdxtv_last_error.argtypes = [vp]
    lib.sdxtv_last_error.restype = C.c_char_p
    lib.sdxp_update.argtypes = [vp, vp]
    lib.sdxp_update_impl.argtypes = [vp]
    lib.sdxp_update_status.argtypes = [vp, vp]
    lib.sdxp_backward.argtypes = [vp, i32, i32, vp]
    lib.sdxp_apply.argtypes = [vp, i32, f32, vp]
    lib.sdxp_backward_factors.argtypes = [vp, i32, vp]
    lib.sdxp_grads_from_factors.argtypes = [vp, vp]
    lib.sdxp_apply_factors.argtypes = [vp, vp]
    lib.sdxp_get_state.argtypes = [vp, C.POINTER(OptState), vp]
    lib.sdxp_set_state.argtypes = [vp, C.POINTER(OptState), vp]
    lib.sdxp_last_error.argtypes = [vp]
    lib.sdxp_last_error.restype = C.c_char_p
    for n in SDX_EXPORTS:
        if n not in ("sdx_last_error", "sdxp_last_error", "sdxp_param_count", "sdxtv_last_error"):
            getattr(lib, n).restype = i32
    _lib = lib
    return lib
