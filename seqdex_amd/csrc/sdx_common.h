// sdx_common.h — device-side constants, buffer table and small math for libseqdex_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/seqdex.h"

#define SDX_WAVE 64
#define SDX_MAXC 1536        // contact points per env = 3 rows per lane x 512 lanes of k_physics
#define SDX_MAXP 1535        // candidate body pairs per env (one impulse row of LDS; 1 024 through round 4: a trained grasp policy overflowed it in 20 of 3e8 env-substeps)
#define SDX_CFIELDS 17       // ab, p3, n3, sep, lam3, wA3, wB3
#define SDX_NSAMP 28
#define SDX_BODY_STATIC 255

// scene constants + derived tables, one copy in HBM (read through the scalar / vector caches)
struct SdxConst {
  sdx_scene_desc sc;
  uint32_t anc[SDX_NLINK];       // bit j: dof j lies on the path base -> link
  int32_t depth[SDX_NLINK];
  int32_t max_depth;
  float brick_radius[SDX_NBRICK_TYPES];
  float rbox_radius[SDX_MAX_RBOX];
  float hand_reset_pose[SDX_NDOF];  // arm prepare pose + scale(finger_reset_unscaled)   GS:1526-1536
};

struct SdxBuf {
  int32_t N;
  int32_t K;               // saved piles per brick type
  int32_t obs_w;           // row width of obs / obs_c: 396 (GraspSim, 3 x 132), 186 (Orient, 62 + 124 never-written zeros), 75 (InsertSim)
  int32_t task_kind;       // copy of sdx_scene_desc.task_kind for kernels that do not take the constants
  uint64_t seed;
  float *root, *dof, *rb, *contact, *jac, *targets, *prev_targets;
  float *obs, *states, *obs_c, *states_c, *rew;
  int64_t *reset, *progress, *randomize;
  float *actions, *init_pos, *init_rot, *successes, *meta_rew, *cons;
  float *finger_dist, *tvalue, *arm_contacts, *student_obs;
  int64_t* success_buf;
  int32_t *pile_choice, *ncontacts;
  float* piles;            // [8,K,132,13]
  float* tv_w;             // packed + transposed T-value weights
  float* cam_rot;          // [N,4] camera-frame target quaternion (input of the T-value MLP)
  float* cscratch;         // [N, SDX_CFIELDS, SDX_MAXC]
  float* stat;             // [2][2] double-buffered (num_resets, finished successes) for cons_successes
  uint32_t* step_count;    // device counter, incremented by the post-physics kernel
  long long* dbg;          // [64] phase time stamps of env dbg_env (profiling aid)
  int32_t dbg_env;         // SDX_DEBUG_ENV at sdx_create (default 0)
  float orient_gate;       // copy of sdx_scene_desc.orient_tvalue_gate
  int32_t *order, *cost;   // [N] k_physics launch order (envs by the cost of their previous step, longest first) / that cost; nullptr: env order
  float *harvest_hand, *harvest_obj;   // [8, SDX_HARVEST_SLOTS, 23*2] / [8, SDX_HARVEST_SLOTS, 13]
  int32_t* harvest_count;  // [8]
  int32_t* seg_stats;      // [N,4] Search camera: accumulators (count, sum rows, sum cols)
  int16_t* seg_image;      // [N,128,128] or nullptr
  float* seg_pix;          // [N,4] count, centroid row, centroid col, previous count
  float* emergence;        // [N]
  float* pile_harvest;     // [8, pile_slots, 132, 13] Orient terminal pile states
  int32_t* pile_harvest_count;   // [8]
  int32_t pile_slots;
  float *tv_succ, *tv_fail;   // [SDX_TV_LOG_SLOTS,4] camera-frame target quaternions logged at episode ends (T-value datasets)
  int32_t* tv_count;       // [2] rows logged: success, failure
  // (step << 24 | env) of the append that filled a ring slot: the slots are claimed with atomics, i.e. in hardware scheduling order; the
  // hosts that consume a ring sort its rows by these keys, which is the order a serial loop over steps and envs would have produced
  unsigned long long *tv_key, *harvest_key, *pile_key;   // [2, SDX_TV_LOG_SLOTS], [8, SDX_HARVEST_SLOTS], [8, pile_slots]
  float* jac_full;         // [N,23,6,23] or nullptr: the whole-hand Jacobian (GS:241), written by k_kinematics only
  int32_t* cstats;         // [4] since create: largest contact count of one env-substep; env-substeps that lost contacts (still over SDX_MAXC
                           // after the rebuild); env-substeps whose list was rebuilt without speculative contacts; env-substeps whose pair list overflowed
  float* tvt_buf;          // Search: [N,652] temporal T-value input (ten 65-number frames, 2 padding columns) or nullptr
  float* tvt_w;            // Search: RetriGraspTValue parameters, W1 rows padded to 652 columns
  float* tvt_h;            // Search: activations [N, 1024 + 512 + 128 + 4]
  // warm start of the contact solver: what the last solve of every env ended with
  int32_t* wcount;         // [N] contacts in the cache
  uint32_t* wkey;          // [N, SDX_MAXC] contact identity (pair rank << 6 | direction << 5 | sample), ascending pair rank
  float* wlam;             // [N, 3, SDX_MAXC] accumulated impulses (normal, two tangents)
  float* insert_aux;       // [N,8] InsertSim: 0..2 rot_err of the last pre_physics_step (IS:1539), 3 |brick - site|, 4 rot_dist
};

// An empty asm the compiler must assume rewrites x: values derived from x afterwards (LDS addresses of the same rows in every
// solver iteration) are recomputed where they are used instead of being kept in registers across the loop.  (tests/hipemu compiles
// the kernels with g++ for the CPU, where the constraint letter does not exist.)
#ifdef HIPEMU
#define SDX_OPAQUE(x) ((void)0)
#define SDX_OPAQUE_S(x) ((void)0)
#define SDX_OPAQUE_AFTER(x, after) ((void)0)
#define SDX_PIN4(a) ((void)0)
#define SDX_PIN8(a) ((void)0)
#define SDX_PIN3x4(a, b, c, d) ((void)0)
#define SDX_PIN3(a) ((void)0)
#define SDX_PIN1(a) ((void)0)
#define SDX_RCP(x) (1.0f / (x))
#define SDX_SQRT_FAST(x) sqrtf(x)
#define SDX_READLANE(x, lane) __shfl((x), (lane), 64)
#define SDX_READLANE_I(x, lane) __shfl((x), (lane), 64)
#define SDX_UNIFORM(x) (x)
#define SDX_WAIT_VMCNT0() ((void)0)
#define SDX_LDS_BARRIER() __syncthreads()
#define SDX_AS_GLOBAL(p) (p)
#define SDX_AS_LDS(p) (p)
#else
#define SDX_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")   // this wave's outstanding global_load_lds pieces have landed
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's outstanding GLOBAL stores (s_waitcnt vmcnt(0)),
// which turns an epilogue of [stores, barrier, stores, ...] into a chain of write round trips.  The asm statements are compiler fences for
// memory accesses ("memory" clobber); s_waitcnt lgkmcnt(0) completes this wave's ds_write / ds_read before the others pass the barrier.
#define SDX_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define SDX_AS_GLOBAL(p) ((const __attribute__((address_space(1))) void*)(p))
#define SDX_AS_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define SDX_OPAQUE(x) asm volatile("" : "+v"(x))
#define SDX_OPAQUE_S(x) asm volatile("" : "+s"(x))   // the same for a wave-uniform value (scalar register)
// x becomes unknown at a point that is ordered after the arithmetic producing `after`: loads addressed through x cannot be hoisted above it
#define SDX_OPAQUE_AFTER(x, after) asm volatile("" : "+v"(x) : "v"(after))
// every element of a[0..3] / a[0..7] passes through one empty asm: pins the order of the arithmetic on them.  Instruction selection orders
// only what hangs on a chain of side effects; unpinned accumulators of an unrolled loop are deferred to the end of the block and the
// operands already loaded for them are spilled
#define SDX_PIN4(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]))
#define SDX_PIN8(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
// four f3 values pinned together: all of them are loaded before anything after this point is computed (one wait for four rows in flight)
#define SDX_PIN3x4(a, b, c, d) asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(d.x), "+v"(d.y), "+v"(d.z))
#define SDX_PIN3(a) asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z))
#define SDX_PIN1(a) asm volatile("" : "+v"(a))
#define SDX_SQRT_FAST(x) __builtin_amdgcn_sqrtf(x)   // v_sqrt_f32, 1 ulp: for integer results that are corrected afterwards
#define SDX_RCP(x) __builtin_amdgcn_rcpf(x)   // v_rcp_f32, 1 ulp: the solver's step lengths do not need IEEE division (12 instructions)
// value of x in a lane known at compile time (v_readlane_b32: the result is wave-uniform, no LDS crossbar); every lane of the wave must be active
#define SDX_READLANE(x, lane) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), (lane)))
#define SDX_READLANE_I(x, lane) __builtin_amdgcn_readlane((x), (lane))   // the same for an integer
#define SDX_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)   // a wave-uniform integer the compiler could not prove uniform -> scalar register
#endif

// sums over aligned groups of 4 / 8 lanes with DPP moves (VALU speed; __shfl_xor goes through ds_bpermute and its LDS latency):
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror.  Every lane of the group ends up with the group's sum; the order of the
// additions is fixed (deterministic).  All lanes of the group must be active.
__device__ __forceinline__ float sum4(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  return v;
}
__device__ __forceinline__ float sum8(float v) {
  v = sum4(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  return v;
}

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

__device__ __forceinline__ f3 F3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
  return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ f4 qmul(f4 a, f4 b) {
  f4 r;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  return r;
}
__device__ __forceinline__ f4 qconj(f4 a) { f4 r = {-a.x, -a.y, -a.z, a.w}; return r; }
// quat_apply of isaacgym.torch_utils: v + w t + u x t, t = 2 u x v
__device__ __forceinline__ f3 qrot(f4 q, f3 v) {
  f3 u = F3(q.x, q.y, q.z);
  f3 t = cross(u, v) * 2.0f;
  return v + t * q.w + cross(u, t);
}
__device__ __forceinline__ f4 qnormalize(f4 a) {
  float n = 1.0f / sqrtf(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
  f4 r = {a.x * n, a.y * n, a.z * n, a.w * n};
  return r;
}
__device__ __forceinline__ f4 qaxis(f3 ax, float ang) {
  float s, c;
  sincosf(0.5f * ang, &s, &c);
  f4 r = {ax.x * s, ax.y * s, ax.z * s, c};
  return r;
}
// a 16-byte aligned row read with ONE 128-bit access (ds_read_b128 for LDS rows)
struct __attribute__((aligned(16))) f4v { float x, y, z, w; };
__device__ __forceinline__ f3 ld3v(const float* p) { const f4v r = *reinterpret_cast<const f4v*>(p); return F3(r.x, r.y, r.z); }
__device__ __forceinline__ f3 ld3(const float* p) { return F3(p[0], p[1], p[2]); }
__device__ __forceinline__ f4 ld4(const float* p) { f4 r = {p[0], p[1], p[2], p[3]}; return r; }
__device__ __forceinline__ void st3(float* p, f3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ void st4(float* p, f4 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; p[3] = a.w; }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// target brick actor index inside the env: brick (i % 8) with {3,4,7} -> 0   (GS:962-965,974-975)
__device__ __forceinline__ int seg_actor(int env) {
  int b = env & 7;
  return SDX_ACTOR_BRICK0 + ((b == 3 || b == 4 || b == 7) ? 0 : b);
}

// counter-based RNG (splitmix-style hash of (seed, stream, counter)); documented in DESIGN.md §6
__device__ __forceinline__ uint64_t sdx_hash(uint64_t seed, uint64_t a, uint64_t b) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (a + 1) + 0xBF58476D1CE4E5B9ull * (b + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
