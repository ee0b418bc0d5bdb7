// sdx_physics.hip — the per-env physics step: one workgroup per env, TWO workgroups resident per CU.
//
// Replaces gym.simulate()/fetch_results() (BT:138-144) and the refresh_* calls (GS:1091-1095) of the reference, whose arithmetic
// lives in the closed Isaac Gym / PhysX binary.  The step is OUR definition ("SDX-1"), restated independently in
// plain C in oracle/physics_oracle.c:
//   A FK  B joint-space inertia + implicit-PD matrix, Cholesky inverse  C implicit PD drive, velocity-product terms, gravity
//   D sampled-SDF box/box contacts (<= 4 per pair of boxes; every shape a compound of boxes)  E active-set mass-split Jacobi on accumulated impulses
//   F semi-implicit Euler;  then outputs: rigid-body states, end-effector Jacobian, net arm contact forces.
//
// Shape of the kernel (rewritten in round 2, restructured in round 3):
//   * 512 threads (8 waves, <= 128 VGPRs) per env, two workgroups per CU (LDS <= 80 KiB each), so one env's barrier / latency phases overlap
//     the other env's arithmetic; launch order = envs by the cost of their previous step, longest first (k_order);
//   * contact geometry (point, normal) stays in LDS; each lane keeps only what changes or divides per iteration for its 3 contacts in
//     registers (accumulated impulses, un-split inverse masses of both sides, target velocity, body ids);
//   * per solver iteration: [AC] relative velocity, active flag -> counted for the NEXT iteration (integer LDS atomics), Jacobi update with
//     the counts of the previous one, impulse to LDS | barrier | [D] CSR gather per body (8 lanes per link, 4 per brick); links project
//     their wrench on the dofs of their path | barrier | robot only: every 8-lane group sums the generalised impulses, applies Hinv and
//     rebuilds its link's twist | barrier;
//   * everything serial (FK levels, Cholesky in registers, drive, twists) runs on wave 0 with wave-synchronous hand-offs instead of
//     workgroup barriers; the other waves meet it at the next barrier;
//   * broadphase: every lane tests its strided candidate BODY pairs (bounding boxes) into a bit mask, ONE block scan places all hits; the
//     surviving body pairs are expanded into the box pairs of the two compounds (consecutive candidates per lane, separating-axis test);
//     narrowphase: lane = box pair picks the samples, lane = contact computes their geometry.
#include <stdlib.h>

#include "sdx_common.h"

#define NL SDX_NLINK
#define ND SDX_NDOF
#define NF SDX_NFREE
#define HP 24                 // padded row stride of the 23x23 matrices in LDS
#define MAXC SDX_MAXC
#define MAXP SDX_MAXP
#define GL 4                  // gather lanes per brick
#define NBODY (NF + NL + 1)   // bricks, links, the static world
#define BODY_W (NF + NL)      // body id of the static world inside the kernel (SDX_BODY_STATIC = 255 outside)
// link frame origins / twists are rows NF.. of the body table: S.bp[NF + k], S.bv[NF + k], S.bw[NF + k]

// LDS written by one lane and read by another lane of the SAME wave without a workgroup barrier: the wave runs in lockstep and the
// LDS executes one wave's operations in order, so only the compiler must be kept from moving accesses across this point
#define WAVE_SYNC()                                         \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
  } while (0)

// Barrier among the waves that hold threads 0 .. nthreads-1 only (whole waves; every one of them calls it): a wave announces itself on an
// LDS counter once its LDS writes have completed and waits until `target` arrivals have been counted.  The counter is monotonic - the
// caller passes (waves) x (number of calls so far) - and the LDS executes the operations of a CU in issue order, so a wave that has seen
// the count reads what the others wrote before they signalled.  Used where a few waves have a dependent stage of their own while the rest
// of the block is busy with independent work (solver loop: link gather -> robot section beside the brick gather); the waves of a
// workgroup are resident together, so the wait cannot deadlock.
#ifdef HIPEMU
#define WAVES_BARRIER(ctr, target, nthreads) hipemu::group_barrier(nthreads)
#else
__device__ __forceinline__ void waves_barrier(int* ctr, int target) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
#define WAVES_BARRIER(ctr, target, nthreads) waves_barrier((ctr), (target))
#endif
// One wave tells the others that its LDS results up to here are complete (flag = a value that only grows); the others wait for it.
#ifdef HIPEMU
// (the emulator runs wave 0 up to its next workgroup barrier before any other wave: the flag is always set when a waiter looks)
#define WAVE_SIGNAL(flag, value) (*(flag) = (value))
#define WAVES_WAIT(flag, value) do { if (*(flag) < (value)) __builtin_trap(); } while (0)
#else
__device__ __forceinline__ void wave_signal(int* flag, int value) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void waves_wait(int* flag, int value) {
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < value) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
#define WAVE_SIGNAL(flag, value) wave_signal((flag), (value))
#define WAVES_WAIT(flag, value) waves_wait((flag), (value))
#endif

__constant__ float c_samp[SDX_NSAMP][3] = {
    {1, 1, 1}, {1, -1, -1}, {-1, 1, -1}, {-1, -1, 1}, {-1, -1, -1}, {-1, 1, 1}, {1, -1, 1}, {1, 1, -1},
    {0, -1, -1}, {0, 1, -1}, {0, -1, 1}, {0, 1, 1}, {-1, 0, -1}, {1, 0, -1}, {-1, 0, 1}, {1, 0, 1},
    {-1, -1, 0}, {1, -1, 0}, {-1, 1, 0}, {1, 1, 0},
    {-0.5f, -1, -1}, {0.5f, -1, -1}, {-0.5f, 1, -1}, {0.5f, 1, -1}, {-0.5f, -1, 1}, {0.5f, -1, 1}, {-0.5f, 1, 1}, {0.5f, 1, 1}};

struct PhysLds {
  // robot
  float q[ND], qd[ND + 1], qdb[ND + 1], tgt[ND], tau[ND];   // qd[ND] = 0: the padding dof of exhausted paths; qdb: the solver's second copy (read one, write the other)
  float lq[NL][4], la[NL + 1][3], lc[NL][3], lI[NL][6], lmass[NL];   // lmass: link masses (a per-lane global load inside the mass-matrix loop cost it 8 k cycles)
  float lal[NL + 1][3], lao[NL][3], lF[NL][3], lN[NL][3];   // velocity-product terms: angular / origin accelerations at zero qdd, inertial wrenches
  alignas(16) float A[ND][HP];      // H -> L -> Hinv
  uint32_t anc[NL];     // bit j: dof j lies on the path base -> link
  int par[NL + 1];      // parent link; [NL] = NL: the padding link of exhausted paths (zero rows of the body table)
  float cf[NL][3];
  // bricks (centre-of-box frame) and their per-brick constants
  // position / linear / angular velocity of EVERY body in one table: bricks 0..71 (centre of the box), links 72..95 (frame origin),
  // entry 96 = the static world (zeros): the solver's point velocities need no case distinction
  // bv / bw are 16-byte rows (one ds_read_b128 each).  INSIDE the solver loop row i of bv holds u_i = v_i - w_i x x_i, the velocity of the
  // body's material point at the world origin, so that a point velocity is u + w x p without the reference point (solve() converts on entry and exit)
  float bp[NBODY][3];
  alignas(16) float bv[NBODY][4];
  alignas(16) float bw[NBODY][4];
  float bq[NF][4];
  float bim[NF];               // inverse mass (target brick already scaled by 1 / seg_mass_scale)
  float bK[NF][6];             // world inverse inertia R diag(1/I) R^T of the substep (xx yy zz xy xz yz)
  // collision compounds by brick TYPE, in the brick's centre-of-mass frame: tsc / tsh the boxes (slabs of the convex
  // hull), tbc / tbh the bounding box, trad a radius about the centre of mass that contains it; hsc / hsh / hn: the hollow compound of
  // THIS env's target brick (walls, roof, upper part) when the scene asks for it (seg_hollow), else hn = 0
  unsigned char btype[NF];
  int tn[SDX_NBRICK_TYPES], hn;
  float tsc[SDX_NBRICK_TYPES][SDX_MAX_SUB][3], tsh[SDX_NBRICK_TYPES][SDX_MAX_SUB][3];
  float tbc[SDX_NBRICK_TYPES][3], tbh[SDX_NBRICK_TYPES][3], trad[SDX_NBRICK_TYPES], tii[SDX_NBRICK_TYPES][3];
  float hsc[SDX_MAX_SUB_HOLLOW][3], hsh[SDX_MAX_SUB_HOLLOW][3];
  float samp[SDX_NSAMP][3];    // copy of c_samp for per-lane sample indices (the manifold selection of pair_contacts)
  float qid[4];                // (0, 0, 0, 1): the orientation row a static body's box reads in load_box2
  // robot collision boxes in the world
  float rc[SDX_MAX_RBOX][3], rq[SDX_MAX_RBOX][4], rh[SDX_MAX_RBOX][3], rrad[SDX_MAX_RBOX];
  int rbl[SDX_MAX_RBOX];
  float sth[SDX_MAX_STATIC][3], stc[SDX_MAX_STATIC][3];   // bounding boxes of the static bodies as THIS env sees them
  int ssf[SDX_MAX_STATIC], ssn[SDX_MAX_STATIC];           // their boxes: rows [ssf, ssf + ssn) of sdx_scene_desc.static_sub_* (read from HBM: only the studded base plate has more than one)
  int acount[2][NF + 2];   // ACTIVE contacts per brick, [NF] on the whole robot, [NF + 1] = 0 (static world): one set is read while the other is counted; integer atomics
  int ecount[NF + NL];  // CSR build: contact sides per body
  float Qc[NL][12];     // per iteration: generalised impulse the contacts of link k apply to the s-th dof of its path (<= 11 dofs)
  uint32_t desc[ND];    // bit k: link k lies below dof j
  int nc, np, overflow, seg_brick, rebuilt, nrob;
  int rsync;            // arrivals at the robot waves' own barrier (solver loop), monotonic within a solve
  uint32_t gtab[NF];    // brick gather lanes of this solve: first lane | lanes << 9 (solve(): the balanced gather's row packing)
  int fkflag;           // substep whose forward kinematics wave 0 has finished (the other waves' broadphase of the robot boxes waits for it)
  int eoff[NF + NL + 1], efill[NF + NL];
  int wsum[16];
  // contacts: geometry in LDS for the whole solve
  alignas(16) float cp[3][MAXC], cn[3][MAXC];
  // (before the narrowphase has produced them, cp / cn hold the list of candidate BOX pairs and the body pairs' offsets into it: S_SP0 ...)
  // three rows that are, in turn: the narrowphase's staging of (separation, body ids) + the candidate body-pair list; L^-1 of the mass
  // matrix; the unsorted CSR fill order during the solver set-up; and the per-contact impulse P of the current iteration
  float P[3][MAXC];
  unsigned short ent[2 * MAXC];   // CSR entries: contact index | side << 15, grouped by body (bricks, then links), ascending contact index
};
#define S_T(S) (reinterpret_cast<float (*)[HP]>(&(S).P[0][0]))                       // L^-1 (mass matrix phase)
#define S_PAIRS(S) (reinterpret_cast<uint32_t*>(&(S).P[2][0]))                       // candidate body pairs (collide)
#define MAXSP (2 * MAXC)                                                             // candidate box pairs per env
#define S_SP0(S) (reinterpret_cast<uint32_t*>(&(S).cp[0][0]))                        // box pair: box a | sub a << 7 | box b << 11 | sub b << 19
#define S_SP1(S) (reinterpret_cast<uint32_t*>(&(S).cn[0][0]))                        // ... its identity: rank of the body pair << 9 | index of the box pair inside it
#define S_OFF(S) (reinterpret_cast<int*>(&(S).cn[0][0]) + MAXSP)                     // body pair -> first candidate box pair (exclusive prefix sum)
#define S_BMASK(S) (reinterpret_cast<uint32_t*>(&(S).cp[0][0]))                       // broadphase hits of lane tid's candidates (bit = trip), NT words; zero between substeps
#define S_LANEMAP(S) (reinterpret_cast<unsigned short*>(&(S).P[2][0]))                // solver set-up: lane -> (brick, lane of the brick) of the balanced gather, [NT]
#define S_ENT2(S) (reinterpret_cast<unsigned short*>(&(S).P[0][0]))                  // unsorted CSR entries (solver set-up)
#define S_EBODY2(S) (reinterpret_cast<unsigned char*>(&(S).P[1][0]))                 // ... and the body each one belongs to
static_assert(ND * HP <= MAXC, "L^-1 must fit one row");
static_assert(MAXP <= MAXC, "the pair list must fit one row");
static_assert(MAXSP + MAXP + 1 <= 3 * MAXC, "box pairs and body-pair offsets fit the rows of cn");
static_assert(3 * 8 * 33 < 1024, "box-pair counts of three body pairs per lane in 10 bits");
static_assert(2 * MAXC * sizeof(unsigned short) == MAXC * sizeof(uint32_t), "contact keys alias the CSR entries");
static_assert(sizeof(PhysLds) <= 80 * 1024, "two workgroups per CU need <= 80 KiB of LDS each");

struct Box { f3 c; f4 q; f3 h; };
// phase clock of env B.dbg_env (tools/time_physics.py): compiled in only with -DSDX_PHASE_CLOCK (make prof -> lib/libseqdex_prof.so, selected
// with SDX_LIB_PATH); the stamps keep a zero VGPR and the debug pointer alive through the whole kernel, which the production build does not pay for
#ifdef SDX_PHASE_CLOCK
#define PSTAMP(i) do { if (threadIdx.x == 0 && e == B.dbg_env && sub == 0) B.dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define SSTAMP(i) do { if (threadIdx.x == 0 && dbg) dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define SCOUNT(i, v) do { if (threadIdx.x == 0 && dbg) dbg[i] = (long long)(v); } while (0)
// per-lane measurements of the first solver pass of the debug env: maximum over the block's lanes (slots 55..62 hold maxima since sdx_create)
#ifdef SDX_LANE_CLOCK   // (its global atomics distort the phase stamps of the passes they measure: a build of its own, make prof VFLAGS=-DSDX_LANE_CLOCK)
#define LCLOCK(var) const long long var = (long long)__builtin_readcyclecounter()
#define LMAX(i, v) do { if (dbg) atomicMax(reinterpret_cast<unsigned long long*>(&dbg[i]), (unsigned long long)(v)); } while (0)
#else
#define LCLOCK(var) ((void)0)
#define LMAX(i, v) ((void)0)
#endif
// timing ablations of the profiling build (tools/ablate_physics.py): bits of SDX_T_DEBUG[63], read once per workgroup.  A set bit REMOVES a
// piece of work (the results are then meaningless; the tool restores the state before every launch): 1 the gather loop of [D], 2 the body of
// [AC], 4 all solver iterations but one, 8 the robot section, 16 the sample classification (no contacts), 32 the broadphase (no pairs),
// 64 the solver set-up's rank pass, 128 the row weights, 256 FK + drive of the second substep, 512 the mass matrix
#define ABL(bit) (abl_bits & (bit))
#else
#define PSTAMP(i) ((void)0)
#define SSTAMP(i) ((void)0)
#define SCOUNT(i, v) ((void)0)
#define LCLOCK(var) ((void)0)
#define LMAX(i, v) ((void)0)
#define ABL(bit) 0
#endif

__device__ __forceinline__ float box_sdf(f3 p, f3 h, f3* g) {
  f3 d = F3(fabsf(p.x) - h.x, fabsf(p.y) - h.y, fabsf(p.z) - h.z);
  f3 s = F3(p.x < 0 ? -1.0f : 1.0f, p.y < 0 ? -1.0f : 1.0f, p.z < 0 ? -1.0f : 1.0f);
  float mx = fmaxf(d.x, fmaxf(d.y, d.z));
  if (mx <= 0) {
    if (d.x >= d.y && d.x >= d.z) *g = F3(s.x, 0, 0);
    else if (d.y >= d.z) *g = F3(0, s.y, 0);
    else *g = F3(0, 0, s.z);
    return mx;
  }
  f3 o = F3(fmaxf(d.x, 0.0f), fmaxf(d.y, 0.0f), fmaxf(d.z, 0.0f));
  float len = sqrtf(dot(o, o));
  *g = F3(s.x * o.x / len, s.y * o.y / len, s.z * o.z / len);
  return len;
}
__device__ __forceinline__ float box_sdf_val(f3 p, f3 h) {
  f3 d = F3(fabsf(p.x) - h.x, fabsf(p.y) - h.y, fabsf(p.z) - h.z);
  float mx = fmaxf(d.x, fmaxf(d.y, d.z));
  if (mx <= 0) return mx;
  f3 o = F3(fmaxf(d.x, 0.0f), fmaxf(d.y, 0.0f), fmaxf(d.z, 0.0f));
  return sqrtf(dot(o, o));
}

// orthonormal tangents of a unit normal without a square root or a branch (Duff et al. 2017); n = (0, 0, +-1) gives t1 = (+-1, 0, 0)... the
// friction pyramid's axes are the world's x and y for a vertical normal.  Recomputed in every solver iteration (12 instructions; the round-2
// construction - cross product with the least-aligned axis, normalised - cost 40).  oracle: tangents()
__device__ __forceinline__ void tangents(f3 n, f3* t1, f3* t2) {
  const float sg = n.z < 0.0f ? -1.0f : 1.0f;
  const float a = -1.0f / (sg + n.z);
  const float b = n.x * n.y * a;
  *t1 = F3(1.0f + sg * n.x * n.x * a, sg * b, -sg * n.x);
  *t2 = F3(b, sg + n.y * n.y * a, -n.y);
}

// ---- shapes.  Box id: 0..71 brick, 72..111 robot box, 128..135 static body; a brick and a static body are COMPOUNDS of boxes (sub index),
// a robot box is one box.  load_bound: the bounding box of a body (broadphase, first separating-axis pass); load_box2: one box of each body of a pair.
#define RBOX0 NF
#define STATIC0 128
static_assert(RBOX0 + SDX_MAX_RBOX <= STATIC0 && STATIC0 + SDX_MAX_STATIC <= 256, "box ids fit 8 bits (7 for the first box of a pair)");
__device__ __forceinline__ bool brick_hollow(const PhysLds& S, int i) { return S.hn > 0 && i == S.seg_brick; }
__device__ __forceinline__ int box_nsub(const PhysLds& S, int id) {
  if (id < NF) return brick_hollow(S, id) ? S.hn : S.tn[S.btype[id]];
  if (id < STATIC0) return 1;
  return S.ssn[id - STATIC0];
}
__device__ __forceinline__ Box load_bound(const PhysLds& S, int id) {
  Box b;
  if (id < NF) {
    const int t = S.btype[id];
    b.q = ld4(S.bq[id]); b.c = ld3(S.bp[id]) + qrot(b.q, ld3(S.tbc[t])); b.h = ld3(S.tbh[t]);
  } else if (id < STATIC0) {
    const int r = id - RBOX0;
    b.c = ld3(S.rc[r]); b.q = ld4(S.rq[r]); b.h = ld3(S.rh[r]);
  } else {
    const int st = id - STATIC0;
    b.c = ld3(S.stc[st]); b.h = ld3(S.sth[st]); b.q.x = 0; b.q.y = 0; b.q.z = 0; b.q.w = 1;
  }
  return b;
}
// Both boxes of a pair with TWO LDS round trips (round 6).  One box at a time (a brick: orientation, position, type -> compound row; a
// robot box: its world rows; a static body: its row, or a row of the HBM table for a compound) compiled to a chain of about ten: type ->
// compound table -> rows, one box after the other, each class of box in its own branch.  Here the rows every class needs are addressed without branches
// (orientation, position, half extents come from the brick / robot-box / static tables by pointer selection; a static body reads the
// identity row S.qid), all of them for both boxes are in flight together, then the bricks' compound rows (which depend on the type) for both.
// A brick's box: centre = position + R x (compound row - centre of mass), half extents = the compound row.  A box of a static COMPOUND
// (the studded base plate, HBM table) is the slow path behind a branch, as before.
__device__ __forceinline__ void load_box2(const SdxConst* C, const PhysLds& S, int ida, int suba, int idb, int subb, Box* A, Box* B) {
  const int id_[2] = {ida, idb}, sub_[2] = {suba, subb};
  f4 q_[2];
  f3 p_[2], h_[2];
  int t_[2], sn_[2], sf_[2];
  const bool hol_any = S.hn > 0;
  const int segb = S.seg_brick;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int id = id_[k];
    const bool brick = id < NF, rob = !brick && id < STATIC0;
    const int r = rob ? id - RBOX0 : 0, st = (!brick && !rob) ? id - STATIC0 : 0, ib = brick ? id : 0;
    const float* qp = brick ? S.bq[ib] : (rob ? S.rq[r] : S.qid);
    const float* pp = brick ? S.bp[ib] : (rob ? S.rc[r] : S.stc[st]);
    const float* hp = rob ? S.rh[r] : S.sth[st];
    q_[k] = ld4(qp); p_[k] = ld3(pp); h_[k] = ld3(hp);
    t_[k] = S.btype[ib]; sn_[k] = S.ssn[st]; sf_[k] = S.ssf[st];
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    SDX_PIN1(q_[k].x); SDX_PIN1(q_[k].y); SDX_PIN1(q_[k].z); SDX_PIN1(q_[k].w); SDX_PIN3(p_[k]); SDX_PIN3(h_[k]);
    SDX_PIN1(t_[k]); SDX_PIN1(sn_[k]); SDX_PIN1(sf_[k]);
  }
  f3 o_[2], hb_[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const bool hol = hol_any && id_[k] == segb;
    const int sb = id_[k] < NF ? sub_[k] : 0;
    o_[k] = ld3(hol ? S.hsc[sb] : S.tsc[t_[k]][sb]);
    hb_[k] = ld3(hol ? S.hsh[sb] : S.tsh[t_[k]][sb]);
  }
  SDX_PIN3x4(o_[0], hb_[0], o_[1], hb_[1]);
  Box* out_[2] = {A, B};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int id = id_[k];
    Box b;
    b.q = q_[k];
    if (id < NF) { b.c = p_[k] + qrot(q_[k], o_[k]); b.h = hb_[k]; }
    else { b.c = p_[k]; b.h = h_[k]; }
    if (id >= STATIC0 && sn_[k] > 1) {
      const int r = sf_[k] + sub_[k];
      b.c = ld3(C->sc.static_sub_center[r]); b.h = ld3(C->sc.static_sub_half[r]);
    }
    *out_[k] = b;
  }
}
__device__ __forceinline__ int box_body(const PhysLds& S, int id) {
  if (id < NF) return id;
  if (id < STATIC0) return NF + S.rbl[id - RBOX0];
  return BODY_W;
}
// the second direction of a pair (samples of B against A) exists unless B is the body box of a static (the studs
// of a static compound, boxes 1.., are sampled like a brick's boxes)
__device__ __forceinline__ bool samples_b(int bid, int bsub) { return bid < STATIC0 || bsub > 0; }

// ---- contact manifold of one direction (samples of A against box B); oracle: dir_setup / sample_contact
// Reference face = the face axis of B with the smallest overlap of the two boxes' extents (separating-axis test over B's three face
// normals), on A's side of B's centre.  Samples whose projection falls on that face (within FACE_TOL of its outline) are FACE samples:
// normal = face normal, separation along it; the others use their own signed distance to B.  <= 4 samples per direction: face samples
// in table order first, then the others.  Shallow overlaps only (s_k >= -FACE_DEPTH x contact offset): a deeply interpenetrating pair
// has no meaningful meeting face (kax = -1) and every sample keeps its own signed distance.
#define FACE_TOL 1e-4f
#define FACE_DEPTH 4.0f
#define WARM_SPEED 0.25f   // m/s
#define WARM_DEPTH 1.5f    // contact offsets
struct Dir { f3 t, ex, ey, ez; int kax; float sgn, smax; };   // ex, ey, ez: A's half edges in B's frame: sample = t + sx ex + sy ey + sz ez
__device__ __forceinline__ Dir dir_setup(const Box& A, const Box& B, float off) {
  Dir D;
  const f4 qbi = qconj(B.q);
  D.t = qrot(qbi, A.c - B.c);
  const f4 qrel = qmul(qbi, A.q);
  D.ex = qrot(qrel, F3(A.h.x, 0, 0));
  D.ey = qrot(qrel, F3(0, A.h.y, 0));
  D.ez = qrot(qrel, F3(0, 0, A.h.z));
  const f3 ex = D.ex, ey = D.ey, ez = D.ez;
  const float sx = fabsf(D.t.x) - B.h.x - (fabsf(ex.x) + fabsf(ey.x) + fabsf(ez.x));
  const float sy = fabsf(D.t.y) - B.h.y - (fabsf(ex.y) + fabsf(ey.y) + fabsf(ez.y));
  const float sz = fabsf(D.t.z) - B.h.z - (fabsf(ex.z) + fabsf(ey.z) + fabsf(ez.z));
  if (sx >= sy && sx >= sz) { D.kax = 0; D.sgn = D.t.x < 0 ? -1.0f : 1.0f; }
  else if (sy >= sz) { D.kax = 1; D.sgn = D.t.y < 0 ? -1.0f : 1.0f; }
  else { D.kax = 2; D.sgn = D.t.z < 0 ? -1.0f : 1.0f; }
  D.smax = fmaxf(sx, fmaxf(sy, sz));   // >= off: a face axis of B separates the boxes by the whole contact offset, no sample can be inside it
  if (D.smax < -FACE_DEPTH * off) D.kax = -1;
  return D;
}
__device__ __forceinline__ float sample_contact(const Dir& D, f3 pb, f3 h, f3* g) {
  const f3 d = F3(fabsf(pb.x) - h.x, fabsf(pb.y) - h.y, fabsf(pb.z) - h.z);
  const float lat = D.kax == 0 ? fmaxf(d.y, d.z) : D.kax == 1 ? fmaxf(d.x, d.z) : fmaxf(d.x, d.y);
  if (D.kax >= 0 && lat <= FACE_TOL) {
    const float pk = D.kax == 0 ? pb.x : D.kax == 1 ? pb.y : pb.z, hk = D.kax == 0 ? h.x : D.kax == 1 ? h.y : h.z;
    *g = F3(D.kax == 0 ? D.sgn : 0.0f, D.kax == 1 ? D.sgn : 0.0f, D.kax == 2 ? D.sgn : 0.0f);
    return D.sgn * pk - hk;
  }
  return box_sdf(pb, h, g);
}

constexpr float kSamp[SDX_NSAMP][3] = {   // c_samp as compile-time constants: immediates of the fully unrolled classification loop
    {1, 1, 1}, {1, -1, -1}, {-1, 1, -1}, {-1, -1, 1}, {-1, -1, -1}, {-1, 1, 1}, {1, -1, 1}, {1, 1, -1},
    {0, -1, -1}, {0, 1, -1}, {0, -1, 1}, {0, 1, 1}, {-1, 0, -1}, {1, 0, -1}, {-1, 0, 1}, {1, 0, 1},
    {-1, -1, 0}, {1, -1, 0}, {-1, 1, 0}, {1, 1, 0},
    {-0.5f, -1, -1}, {0.5f, -1, -1}, {-0.5f, 1, -1}, {0.5f, 1, -1}, {-0.5f, -1, 1}, {0.5f, -1, 1}, {-0.5f, 1, 1}, {0.5f, 1, 1}};
__device__ __forceinline__ f3 face_frame(f3 v, int kax) {   // component kax moves to z
  return kax == 0 ? F3(v.y, v.z, v.x) : kax == 1 ? F3(v.x, v.z, v.y) : v;
}
// ---- the <= 4 contacts of a pair of boxes (oracle: collide_pair).  classify_dir: the 28 samples of box A against box
// T in the face frame of that direction -> face samples / other samples inside `incl` as bit masks (incl: the contact offset, or 0 when
// the list is rebuilt after a capacity overflow).  Direction 1 also returns what the selection below needs to place a sample in the
// PAIR's reference frame: the reference axis kref and the face-frame rows (x, y) of the sample basis.
struct FaceBasis { float tx, ty, exx, exy, eyx, eyy, ezx, ezy; };
template <bool DIR2>
__device__ __forceinline__ bool classify_dir(const PhysLds& S, const Box& A, const Box& T, float off, float incl, int* kref, FaceBasis* fb,
                                             uint32_t* mface_out, uint32_t* mother_out) {
  const Dir D = dir_setup(A, T, off);
  if (D.smax >= incl) return false;   // separated: no sample of either direction can be inside the threshold
  const f3 t = face_frame(D.t, D.kax), ex = face_frame(D.ex, D.kax), ey = face_frame(D.ey, D.kax), ez = face_frame(D.ez, D.kax);
  if (!DIR2) {
    *kref = D.kax >= 0 ? D.kax : 2;
    fb->tx = t.x; fb->ty = t.y; fb->exx = ex.x; fb->exy = ex.y; fb->eyx = ey.x; fb->eyy = ey.y; fb->ezx = ez.x; fb->ezy = ez.y;
  }
  const f3 h = face_frame(T.h, D.kax);
  const float ftol = D.kax >= 0 ? FACE_TOL : -1e30f, off2 = incl * incl;
  uint32_t mface = 0, mother = 0;
  // Fully unrolled with the table entries as immediates: 18 instructions per sample (rolled four at a time with the entries from
  // scalar loads: 56, and this loop is VALU-issue bound: 30 k cycles per chunk of 512 box pairs against 11 k).  The loop body must stay
  // this small: with the running extremes of an earlier manifold rule inside it the two unrolled instances cost the WHOLE kernel 220
  // spilled VGPRs, 18 scratch operations inside the solver's iteration loop among them; as it is: 24 spills, none in a loop
  // (python tools/isa_check.py; -DSDX_CLASSIFY_UNROLL=4 is the rolled form)
#ifndef SDX_CLASSIFY_UNROLL
#define SDX_CLASSIFY_UNROLL SDX_NSAMP
#endif
  constexpr int CU = SDX_CLASSIFY_UNROLL;
#pragma unroll CU
  for (int s = 0; s < SDX_NSAMP; ++s) {
    const float k0 = CU == SDX_NSAMP ? kSamp[s][0] : c_samp[s][0], k1 = CU == SDX_NSAMP ? kSamp[s][1] : c_samp[s][1], k2 = CU == SDX_NSAMP ? kSamp[s][2] : c_samp[s][2];
    const f3 pb = ((t + ex * k0) + ey * k1) + ez * k2;
    const float dx = fabsf(pb.x) - h.x, dy = fabsf(pb.y) - h.y, dz = fabsf(pb.z) - h.z;
    const float lat = fmaxf(dx, dy);
    const float sdf = D.sgn * pb.z - h.z;
    const float ox = fmaxf(dx, 0.0f), oy = fmaxf(dy, 0.0f), oz = fmaxf(dz, 0.0f);
    const bool near = fmaxf(lat, dz) <= 0.0f || ox * ox + oy * oy + oz * oz < off2;
    const bool face = lat <= ftol;
    if (face && sdf < incl) mface |= 1u << s;
    if (!face && near) mother |= 1u << s;
  }
  *mface_out = mface;
  *mother_out = mother;
  return true;
}
// The 4 slots of the pair.  Face samples first, chosen to SPAN the patch the boxes meet on: every face sample has lateral coordinates
// (a, b) in B's frame (B's axes without the axis kref of direction 1's reference face; a sample of B: its own table entry x hB);
// p1 = the sample furthest along (1, 0.1), p2 = the one furthest from p1, p3 / p4 = the ones furthest to the left / right of the line
// p1 p2; ties: the first in enumeration order (direction 1 in table order, then direction 2) - a later candidate replaces the incumbent
// only when it beats it by more than MANIFOLD_TIE_L (1 um, p1's score) / MANIFOLD_TIE_A (1e-8 m^2, squared distance and line offset): samples
// of one box edge are all equally far from a line parallel to it, and coincident samples of the two directions score the same, so that
// without the margin rounding decides (seen on the device: one such pair per 8 golden envs chose another sample than the oracle, and
// 16 Jacobi iterations spread that over eight bricks).  Then the remaining face samples in that
// order, then the other samples (edge / corner regions, speculative contacts) the same way.  Returns the number of contacts; sel1 / sel2:
// the chosen samples of the two directions.
#define MANIFOLD_EPS 1e-7f
#define MANIFOLD_TIE_L 1e-6f
#define MANIFOLD_TIE_A 1e-8f
__device__ __forceinline__ int pair_contacts(const PhysLds& S, const Box& A, const Box& B, bool second, float off, float incl, uint32_t* sel1, uint32_t* sel2) {
  uint32_t f1 = 0, o1 = 0, f2 = 0, o2 = 0;
  int kref = 2;
  FaceBasis fb;
  *sel1 = 0; *sel2 = 0;
  if (!classify_dir<false>(S, A, B, off, incl, &kref, &fb, &f1, &o1)) return 0;
  if (second && !classify_dir<true>(S, B, A, off, incl, &kref, &fb, &f2, &o2)) return 0;
  uint32_t s1 = 0, s2 = 0;
  int n = 0;
  const uint64_t fm = (uint64_t)f1 | ((uint64_t)f2 << 32);
  if (fm) {
    const float ha = kref == 0 ? B.h.y : B.h.x, hb = kref == 2 ? B.h.y : B.h.z;
    // (a, b) of candidate id (direction << 5 | sample); the table entries come from the LDS copy of the sample table (per-lane index)
#define SDX_COORDS(id, a, b)                                                                                    \
    {                                                                                                           \
      const float* ks = S.samp[(id) & 31];                                                                      \
      const float kx = ks[0], ky = ks[1], kz = ks[2];                                                           \
      if ((id) < 32) { a = ((fb.tx + fb.exx * kx) + fb.eyx * ky) + fb.ezx * kz; b = ((fb.ty + fb.exy * kx) + fb.eyy * ky) + fb.ezy * kz; } \
      else { a = (kref == 0 ? ky : kx) * ha; b = (kref == 2 ? ky : kz) * hb; }                                  \
    }
    int p1 = -1, p2 = -1, p3 = -1, p4 = -1;
    float a1 = 0.0f, b1 = 0.0f, a2 = 0.0f, b2 = 0.0f, best = -1e30f;
    uint64_t m = fm;
#pragma unroll 1
    while (m) {
      const int id = __ffsll((unsigned long long)m) - 1;
      m &= m - 1;
      float a, b;
      SDX_COORDS(id, a, b)
      const float k = a + 0.1f * b;
      if (k > best + MANIFOLD_TIE_L) { best = k; p1 = id; a1 = a; b1 = b; }
    }
    best = 0.0f;
    m = fm;
#pragma unroll 1
    while (m) {
      const int id = __ffsll((unsigned long long)m) - 1;
      m &= m - 1;
      float a, b;
      SDX_COORDS(id, a, b)
      const float k = (a - a1) * (a - a1) + (b - b1) * (b - b1);
      if (k > best + MANIFOLD_TIE_A) { best = k; p2 = id; a2 = a; b2 = b; }
    }
    if (p2 >= 0) {
      float hi = MANIFOLD_EPS, lo = -MANIFOLD_EPS;
      m = fm;
#pragma unroll 1
      while (m) {
        const int id = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        float a, b;
        SDX_COORDS(id, a, b)
        const float k = (a2 - a1) * (b - b1) - (b2 - b1) * (a - a1);
        if (k > hi + MANIFOLD_TIE_A) { hi = k; p3 = id; }
        if (k < lo - MANIFOLD_TIE_A) { lo = k; p4 = id; }
      }
    }
#undef SDX_COORDS
    const int pk[4] = {p1, p2, p3, p4};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (pk[k] >= 0) { if (pk[k] < 32) s1 |= 1u << pk[k]; else s2 |= 1u << (pk[k] - 32); ++n; }
  }
  uint32_t m;
#define SDX_FILL(mask, sel)                                   \
  m = (mask) & ~(sel);                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)               \
    if (m && n < 4) { (sel) |= m & (0u - m); m &= m - 1; ++n; }
  SDX_FILL(f1, s1) SDX_FILL(f2, s2) SDX_FILL(o1, s1) SDX_FILL(o2, s2)
#undef SDX_FILL
  *sel1 = s1; *sel2 = s2;
  return n;
}

#define LOAD_BOX2(ba, sa, bb, sb) Box A, Bx; load_box2(C, S, ba, sa, bb, sb, &A, &Bx);   // (both boxes of a pair: A, Bx)
#define S_CKEY(S) (reinterpret_cast<uint32_t*>(&(S).ent[0]))   // identity of contact c until the solver's set-up has read it (the CSR lives here later)
__device__ __forceinline__ f3 point_vel(const PhysLds& S, int id, f3 p) {   // id: row of the body table (static world = zeros)
  return ld3(S.bv[id]) + cross(ld3(S.bw[id]), p - ld3(S.bp[id]));
}

__device__ __forceinline__ float brick_w(const PhysLds& S, int i, f3 p, f3 d) {
  const f3 rxd = cross(p - ld3(S.bp[i]), d);
  const f3 l = qrot(qconj(ld4(S.bq[i])), rxd);
  const float* ii = S.tii[S.btype[i]];   // mass / principal inertia of the type: inverse inertia = inverse mass x that
  return S.bim[i] * (1.0f + l.x * l.x * ii[0] + l.y * l.y * ii[1] + l.z * l.z * ii[2]);
}

// ---------------------------------------------------------------- A: FK, on wave 0 only (lane = link; levels by wave-synchronous hand-off)
// world inertia times vector: R I R^T x with I = (xx yy zz xy xz yz) in the link frame
__device__ __forceinline__ f3 inertia_mul(f4 q, const float* I, f3 x) {
  const f3 l = qrot(qconj(q), x);
  return qrot(q, F3(I[0] * l.x + I[3] * l.y + I[4] * l.z, I[3] * l.x + I[1] * l.y + I[5] * l.z, I[4] * l.x + I[5] * l.y + I[2] * l.z));
}
// forward declaration (the velocity phases of FK use it)
__device__ __forceinline__ void twists_wave0(PhysLds& S, int tid);

// called by every lane of wave 0 (tid < 64).  Only the link POSES form a serial chain over the tree depth (one quaternion product and
// one rotation per level); everything else is one lane per link in parallel: joint axes / centres of mass, then the twists and the
// velocity-product terms as sums over the (<= 11) dofs of each link's path - the same terms in the same order as the recursions
// w_k = w_p + a_k qd_k, al_k = al_p + w_p x (a_k qd_k), ao_k = ao_p + al_p x r + w_p x (w_p x r) (recursive Newton-Euler at zero joint
// acceleration, fixed base, no gravity on the robot).  with_bias: the velocity-product wrenches the drive needs; with_inertia: the world
// inertia tensors the mass matrix needs.
__device__ __forceinline__ void fk_wave0(const SdxConst* C, PhysLds& S, int tid, bool with_inertia, bool with_bias, long long* dbg = nullptr) {
  const sdx_scene_desc& sc = C->sc;
  const bool link = tid > 0 && tid < NL;
  // this lane's link constants, fetched once (all loads in flight together)
  int par = 0, dep = -1;
  f4 ql = {0, 0, 0, 1};
  f3 jp = F3(0, 0, 0), axl = F3(0, 0, 1), com = F3(0, 0, 0);
  float mass = 0.0f, I6[6] = {0, 0, 0, 0, 0, 0};
  if (link) {
    par = sc.parent[tid]; dep = C->depth[tid];
    const f4 jq = ld4(sc.joint_quat[tid]);
    const f3 ax = ld3(sc.joint_axis[tid]);
    jp = ld3(sc.joint_pos[tid]); com = ld3(sc.link_com[tid]);
    mass = sc.link_mass[tid];
#pragma unroll
    for (int i = 0; i < 6; ++i) I6[i] = sc.link_inertia[tid][i];
    float sn, cs;
    sincosf(0.5f * S.q[tid - 1], &sn, &cs);
    f4 qa; qa.x = ax.x * sn; qa.y = ax.y * sn; qa.z = ax.z * sn; qa.w = cs;
    ql = qmul(jq, qa);            // rotation parent frame -> link frame
    axl = qrot(jq, ax);           // joint axis in the parent frame
  }
  if (tid == 0) {
    st4(S.lq[0], ld4(sc.base_quat));
    st3(S.bp[NF + 0], ld3(sc.base_pos));
    st3(S.la[0], F3(0, 0, 1));
    st3(S.bv[NF + 0], F3(0, 0, 0));
    st3(S.bw[NF + 0], F3(0, 0, 0));
    st3(S.lal[0], F3(0, 0, 0)); st3(S.lao[0], F3(0, 0, 0)); st3(S.lF[0], F3(0, 0, 0)); st3(S.lN[0], F3(0, 0, 0));
    st3(S.lc[0], ld3(sc.base_pos) + qrot(ld4(sc.base_quat), ld3(sc.link_com[0])));
    com = ld3(sc.link_com[0]);
#pragma unroll
    for (int i = 0; i < 6; ++i) I6[i] = sc.link_inertia[0][i];
  }
  WAVE_SYNC();
  SSTAMP(7);
  // ---- the serial part: poses, level by level
  const int max_depth = C->max_depth;
  for (int d = 1; d <= max_depth; ++d) {
    if (link && dep == d) {
      f4 qp = ld4(S.lq[par]);
      f3 pp = ld3(S.bp[NF + par]);
      SDX_PIN1(qp.x); SDX_PIN1(qp.y); SDX_PIN1(qp.z); SDX_PIN1(qp.w); SDX_PIN3(pp);   // (the parent's two rows in one LDS round trip per level)
      st4(S.lq[tid], qnormalize(qmul(qp, ql)));
      st3(S.bp[NF + tid], pp + qrot(qp, jp));
    }
    WAVE_SYNC();
  }
  SSTAMP(8);
  // ---- one lane per link: world joint axis, centre of mass
  f4 qk = {0, 0, 0, 1};
  f3 dk = F3(0, 0, 0);
  if (link) {
    qk = ld4(S.lq[tid]);
    st3(S.la[tid], qrot(ld4(S.lq[par]), axl));
    dk = qrot(qk, com);
    st3(S.lc[tid], ld3(S.bp[NF + tid]) + dk);
  }
  WAVE_SYNC();
  SSTAMP(9);
  twists_wave0(S, tid);            // w_k, v_k from the dofs on the path
  SSTAMP(10);
  if (with_bias) {
    WAVE_SYNC();
    // al_k = sum over the links m on the path of w_par(m) x (a_m qd_m)
    f3 alk = F3(0, 0, 0);
    if (link) {
      uint32_t m = S.anc[tid];
      constexpr int FB = 4;
#pragma unroll
      for (int t0 = 0; t0 < 11; t0 += FB) {
        int pm_[FB];
        f3 la_[FB], wp_[FB];
        float qd_[FB];
#pragma unroll
        for (int u = 0; u < FB; ++u) if (t0 + u < 11) {
          const int j = m ? __ffs(m) - 1 : ND;
          m &= m - 1;
          pm_[u] = S.par[j + 1]; la_[u] = ld3(S.la[j + 1]); qd_[u] = S.qd[j];
        }
#pragma unroll
        for (int u = 0; u < FB; ++u) if (t0 + u < 11) SDX_PIN1(pm_[u]);
#pragma unroll
        for (int u = 0; u < FB; ++u) if (t0 + u < 11) wp_[u] = ld3(S.bw[NF + pm_[u]]);
#pragma unroll
        for (int u = 0; u < FB; ++u) if (t0 + u < 11) { SDX_PIN3(wp_[u]); SDX_PIN3(la_[u]); SDX_PIN1(qd_[u]); }
#pragma unroll
        for (int u = 0; u < FB; ++u) if (t0 + u < 11) alk = alk + cross(wp_[u], la_[u] * qd_[u]);
      }
      st3(S.lal[tid], alk);
    }
    WAVE_SYNC();
    SSTAMP(11);
    // ao_k = sum over the links m on the path of al_par(m) x r_m + w_par(m) x (w_par(m) x r_m), r_m = p_m - p_par(m)
    if (link) {
      f3 aok = F3(0, 0, 0);
      uint32_t m = S.anc[tid];
      constexpr int FB = 4;
#pragma unroll
      for (int t0 = 0; t0 < 11; t0 += FB) {
        int pm_[FB];
        f3 pj_[FB], pp_[FB], wp_[FB], al_[FB];
#pragma unroll
        for (int u = 0; u < FB; ++u) if (t0 + u < 11) {
          const int j = m ? __ffs(m) - 1 : ND;
          m &= m - 1;
          pm_[u] = S.par[j + 1]; pj_[u] = ld3(S.bp[NF + j + 1]);
        }
#pragma unroll
        for (int u = 0; u < FB; ++u) if (t0 + u < 11) SDX_PIN1(pm_[u]);
#pragma unroll
        for (int u = 0; u < FB; ++u) if (t0 + u < 11) { pp_[u] = ld3(S.bp[NF + pm_[u]]); wp_[u] = ld3(S.bw[NF + pm_[u]]); al_[u] = ld3(S.lal[pm_[u]]); }
#pragma unroll
        for (int u = 0; u < FB; ++u) if (t0 + u < 11) { SDX_PIN3(pj_[u]); SDX_PIN3(pp_[u]); SDX_PIN3(wp_[u]); SDX_PIN3(al_[u]); }
#pragma unroll
        for (int u = 0; u < FB; ++u) if (t0 + u < 11) {
          const f3 r = pj_[u] - pp_[u], wp = wp_[u];
          aok = aok + cross(al_[u], r) + cross(wp, cross(wp, r));
        }
      }
      st3(S.lao[tid], aok);
      const f3 wk = ld3(S.bw[NF + tid]);
      const f3 acom = aok + cross(alk, dk) + cross(wk, cross(wk, dk));
      st3(S.lF[tid], acom * mass);
      st3(S.lN[tid], inertia_mul(qk, I6, alk) + cross(wk, inertia_mul(qk, I6, wk)));
    }
  }
  SSTAMP(12);
  if (tid < sc.n_rbox) {
    const int k = S.rbl[tid];
    const f4 q = ld4(S.lq[k]);
    st3(S.rc[tid], ld3(S.bp[NF + k]) + qrot(q, ld3(sc.rbox_center[tid])));
    st4(S.rq[tid], qmul(q, ld4(sc.rbox_quat[tid])));
  }
  if (with_inertia && tid < NL) {   // world inertia R I R^T of each link (xx yy zz xy xz yz)
    const f4 q = ld4(S.lq[tid]);
    const f3 ex = qrot(q, F3(1, 0, 0)), ey = qrot(q, F3(0, 1, 0)), ez = qrot(q, F3(0, 0, 1));  // columns of R
    const f3 c0 = ex * I6[0] + ey * I6[3] + ez * I6[4];
    const f3 c1 = ex * I6[3] + ey * I6[1] + ez * I6[5];
    const f3 c2 = ex * I6[4] + ey * I6[5] + ez * I6[2];
    S.lI[tid][0] = c0.x * ex.x + c1.x * ey.x + c2.x * ez.x;
    S.lI[tid][1] = c0.y * ex.y + c1.y * ey.y + c2.y * ez.y;
    S.lI[tid][2] = c0.z * ex.z + c1.z * ey.z + c2.z * ez.z;
    S.lI[tid][3] = c0.x * ex.y + c1.x * ey.y + c2.x * ez.y;
    S.lI[tid][4] = c0.x * ex.z + c1.x * ey.z + c2.x * ez.z;
    S.lI[tid][5] = c0.y * ex.z + c1.y * ey.z + c2.y * ez.z;
  }
  WAVE_SYNC();
}

// link twists from qd, on wave 0: w_k = sum_j a_j qd_j, v_k = sum_j (a_j qd_j) x (p_k - p_j) over the dofs j on the path (<= 11 of them:
// 7 arm joints + 4 of one finger).  Fixed trip count, no branches: an exhausted path reads dof ND, whose velocity slot is zero
__device__ __forceinline__ void twists_wave0(PhysLds& S, int tid) {
  if (tid > 0 && tid < NL) {
    f3 w = F3(0, 0, 0), v = F3(0, 0, 0);
    const f3 pk = ld3(S.bp[NF + tid]);
    uint32_t m = S.anc[tid];
    // Operands of FOUR path terms at a time are loaded before their arithmetic (SDX_PIN*): the plain loop below compiles to two dependent
    // LDS round trips per term - 22 in a row on the one wave everybody else is waiting for (round 6, ISA reading).  Same terms, same order.
    constexpr int FB = 4;
#pragma unroll
    for (int t0 = 0; t0 < 11; t0 += FB) {
      f3 la_[FB], pj_[FB];
      float qd_[FB];
#pragma unroll
      for (int u = 0; u < FB; ++u) if (t0 + u < 11) {
        const int j = m ? __ffs(m) - 1 : ND;
        m &= m - 1;
        la_[u] = ld3(S.la[j + 1]); qd_[u] = S.qd[j]; pj_[u] = ld3(S.bp[NF + j + 1]);
      }
#pragma unroll
      for (int u = 0; u < FB; ++u) if (t0 + u < 11) { SDX_PIN3(la_[u]); SDX_PIN3(pj_[u]); SDX_PIN1(qd_[u]); }
#pragma unroll
      for (int u = 0; u < FB; ++u) if (t0 + u < 11) {
        const f3 aj = la_[u] * qd_[u];
        w = w + aj;
        v = v + cross(aj, pk - pj_[u]);
      }
    }
    st3(S.bw[NF + tid], w);
    st3(S.bv[NF + tid], v);
  }
}

__device__ __forceinline__ void tri_index(int idx, int* i, int* j) {   // idx -> (i, j), j <= i, row-major lower triangle
  int r = (int)((SDX_SQRT_FAST(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);   // within one of the exact row: two branch-free corrections
  r += (r + 1) * (r + 2) / 2 <= idx;
  r -= r * (r + 1) / 2 > idx;
  *i = r;
  *j = idx - r * (r + 1) / 2;
}

// ---------------------------------------------------------------- B: H = M + implicit PD terms, Hinv (once per step)
// (defined with the broadphase below) part 0 / 1 of the candidate enumeration, tested by the lanes of waves 1.. while wave 0 is busy alone
template <int NT>
__device__ __forceinline__ void broad_mask_part(const SdxConst* C, PhysLds& S, int tid, int part);

template <int NT>
__device__ __forceinline__ void mass_matrix(const SdxConst* C, PhysLds& S, int tid, float h, long long* dbg) {
  const sdx_scene_desc& sc = C->sc;
  for (int idx = tid; idx < ND * (ND + 1) / 2; idx += NT) {
    int i, j;
    tri_index(idx, &i, &j);
    float s = 0.0f;
    if ((S.anc[i + 1] >> j) & 1u) {
      const f3 ai = ld3(S.la[i + 1]), aj = ld3(S.la[j + 1]);
      const f3 pi = ld3(S.bp[NF + i + 1]), pj = ld3(S.bp[NF + j + 1]);
      {   // links below dof i = bits of desc[i] (all of them > i); four links' operands in flight at a time, same terms, ascending k
        const uint32_t below = S.desc[i];
        constexpr int FB = 4;
#pragma unroll 1
        for (int k0 = i + 1; k0 < NL; k0 += FB) {
          f3 ck_[FB];
          float I_[FB][6], mk_[FB];
#pragma unroll
          for (int u = 0; u < FB; ++u) {
            const int k = k0 + u < NL ? k0 + u : NL - 1;
            ck_[u] = ld3(S.lc[k]); mk_[u] = S.lmass[k];
#pragma unroll
            for (int r = 0; r < 6; ++r) I_[u][r] = S.lI[k][r];
          }
#pragma unroll
          for (int u = 0; u < FB; ++u) { SDX_PIN3(ck_[u]); SDX_PIN1(mk_[u]); SDX_PIN4(I_[u]); SDX_PIN1(I_[u][4]); SDX_PIN1(I_[u][5]); }
#pragma unroll
          for (int u = 0; u < FB; ++u) {
            const f3 li = cross(ai, ck_[u] - pi), lj = cross(aj, ck_[u] - pj);
            const float* I = I_[u];
            const f3 Ia = F3(I[0] * aj.x + I[3] * aj.y + I[4] * aj.z, I[3] * aj.x + I[1] * aj.y + I[5] * aj.z,
                             I[4] * aj.x + I[5] * aj.y + I[2] * aj.z);
            const float term = mk_[u] * dot(li, lj) + dot(ai, Ia);
            if (k0 + u < NL && ((below >> (k0 + u)) & 1u)) s += term;
          }
        }
      }
    }
    if (i == j) s += sc.armature[i] + h * sc.kd[i] + h * h * sc.kp[i];
    S.A[i][j] = s;
    S.A[j][i] = s;
  }
  __syncthreads();
  SSTAMP(34);
  if (tid < 64) {
    // Cholesky on wave 0 with the matrix in registers: lane i holds row i of H; right-looking - column j is scaled, then every lane
    // subtracts l_ij l_kj from its entries k > j, with l_kj read from lane k (v_readlane: wave-uniform, no LDS, no hand-off).  The
    // upper triangle carries don't-care values; lanes >= ND carry zeros.  (The left-looking LDS version took 35 k of the 58 k cycles of
    // this phase: 23 dependent columns with a wave-synchronous hand-off each.)
    float a[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) a[k] = tid < ND ? S.A[tid][k] : 0.0f;
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      const float d = sqrtf(SDX_READLANE(a[j], j));
      const float lij = tid == j ? d : a[j] / d;
      a[j] = lij;
#pragma unroll
      for (int k = j + 1; k < ND; ++k) a[k] -= lij * SDX_READLANE(lij, k);
    }
    SSTAMP(35);
    // T = L^-1, lane = column c: t[i] = ([i == c] - sum_{k < i} L[i][k] t[k]) / L[i][i], t[k] = 0 above the diagonal.  L[i][k] is the same
    // for every lane: row i of L goes through LDS (lane i writes its row into the free second impulse row, every lane reads it back as
    // 16-byte broadcast loads that do not depend on the recurrence and run ahead of it).  Round 5 fetched each L[i][k] with a v_readlane
    // in front of its fma: 253 readlane -> fma pairs in one dependent chain, 9 k cycles; the sums and their order are the same.
    const int c = tid;
    float* Lrow = &S.P[1][0];   // [ND][HP]
    if (tid < ND) {
#pragma unroll
      for (int k = 0; k < ND; ++k) Lrow[tid * HP + k] = a[k];
    }
    WAVE_SYNC();
    float t[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      float li[HP];
#pragma unroll
      for (int k4 = 0; k4 <= i / 4; ++k4) {
        const f4v x = *reinterpret_cast<const f4v*>(Lrow + i * HP + 4 * k4);
        li[4 * k4] = x.x; li[4 * k4 + 1] = x.y; li[4 * k4 + 2] = x.z; li[4 * k4 + 3] = x.w;
      }
      float sacc = (i == c) ? 1.0f : 0.0f;
#pragma unroll
      for (int k = 0; k < i; ++k) sacc -= li[k] * t[k];
      t[i] = i >= c ? sacc / li[i] : 0.0f;
    }
    if (tid < ND) {
#pragma unroll
      for (int i = 0; i < ND; ++i) S_T(S)[i][c] = t[i];
    }
  } else {
    broad_mask_part<NT>(C, S, tid, 1);   // the robot boxes' candidates (their poses are this substep's: FK is done) beside the factorisation
  }
  __syncthreads();
  SSTAMP(36);
  // Hinv = T^T T
  for (int idx = tid; idx < ND * (ND + 1) / 2; idx += NT) {
    int i, j;
    tri_index(idx, &i, &j);
    float s = 0.0f;
    for (int k = i; k < ND; ++k) s += S_T(S)[k][i] * S_T(S)[k][j];
    S.A[i][j] = s;
    S.A[j][i] = s;
  }
  __syncthreads();
}

// ---------------------------------------------------------------- block-level exclusive scan of a small count k (NBITS bits), thread order
template <int NT, int NBITS = 4>
__device__ __forceinline__ int block_scan_small(PhysLds& S, int k, int tid, int* total) {
  constexpr int NW = NT / 64;
  const int lane = tid & 63, wave = tid >> 6;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int pre = 0, wt = 0;
#pragma unroll
  for (int b = 0; b < NBITS; ++b) {
    const uint64_t bal = __ballot((k >> b) & 1);
    pre += __popcll(bal & lt) << b;
    wt += __popcll(bal) << b;
  }
  if (lane == 0) S.wsum[wave] = wt;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) { const int c = S.wsum[w]; if (w < wave) off += c; tot += c; }
  __syncthreads();
  *total = tot;
  return off + pre;
}

// ---------------------------------------------------------------- D: contacts
// candidate test of pair index idx in the fixed enumeration (brick/static, brick/brick i < j, then per robot box: bricks, statics).
// Conservative: a pair can only produce a contact when a sample point of one box lies within `off` of the other box, and the SDF is
// 1-Lipschitz, so |sdf_B(centre of A)| <= radius_A + off (or the same with A and B exchanged) must hold.
__device__ __forceinline__ bool box_near(const PhysLds& S, f3 ca, float ra, f3 cb, f4 qb, f3 hb, float off) {
  return box_sdf_val(qrot(qconj(qb), ca - cb), hb) <= ra + off;
}
// pair index -> (box a | box b << 8); box ids: 0..71 brick, 72..111 robot box, 128.. static body
// (ns, per are compile-time constants at the call sites - the enumeration runs over all SDX_MAX_STATIC slots, unused ones never are
// candidates - so that the divisions below are multiplications: a division by a run-time integer is ~30 instructions, twice per candidate)
__device__ __forceinline__ uint32_t pair_code(int idx, int n1, int n2, int ns, int per) {
  if (idx < n1) return (uint32_t)(idx / ns) | ((uint32_t)(STATIC0 + idx % ns) << 8);
  if (idx < n1 + n2) {
    int i, j;
    tri_index(idx - n1, &i, &j);   // lower triangle without the diagonal: bricks (j, i + 1), j <= i
    return (uint32_t)j | ((uint32_t)(i + 1) << 8);
  }
  const int t = idx - n1 - n2, r = t / per, u = t % per;
  return (uint32_t)(RBOX0 + r) | ((uint32_t)(u < NF ? u : STATIC0 + u - NF) << 8);
}
// (bricks: sphere of radius trad about the centre of mass - it contains the bounding box, whose centre lies a few millimetres off)
__device__ __forceinline__ bool brick_near(const PhysLds& S, f3 ca, float ra, int j, float off) {   // centre / radius against brick j's bounding box
  const f4 qj = ld4(S.bq[j]);
  const int tj = S.btype[j];
  return box_sdf_val(qrot(qconj(qj), ca - ld3(S.bp[j])) - ld3(S.tbc[tj]), ld3(S.tbh[tj])) <= ra + off;
}
__device__ __forceinline__ bool candidate(const PhysLds& S, int idx, int n1, int n2, int ns, int per, int ns_used, float off) {
  if (idx < n1) {
    const int i = idx / ns, st = idx % ns;
    if (st >= ns_used) return false;
    return box_sdf_val(ld3(S.bp[i]) - ld3(S.stc[st]), ld3(S.sth[st])) <= S.trad[S.btype[i]] + off;
  }
  if (idx < n1 + n2) {
    int i, j;
    tri_index(idx - n1, &i, &j);
    i += 1;
    const f3 ci = ld3(S.bp[i]), cj = ld3(S.bp[j]);
    const f3 d = ci - cj;
    const float ri = S.trad[S.btype[i]], rj = S.trad[S.btype[j]];
    const float rr = ri + rj + off;
    if (dot(d, d) > rr * rr) return false;
    return brick_near(S, ci, ri, j, off) || brick_near(S, cj, rj, i, off);
  }
  const int t = idx - n1 - n2, r = t / per, u = t % per;
  if (S.rbl[r] == 0) return false;   // the fixed base never generates contacts
  const f3 rc = ld3(S.rc[r]);
  const float rr0 = S.rrad[r];
  if (u < NF) {
    const f3 cb = ld3(S.bp[u]);
    const f3 d = rc - cb;
    const float rb = S.trad[S.btype[u]];
    const float rr = rr0 + rb + off;
    if (dot(d, d) > rr * rr) return false;
    return brick_near(S, rc, rr0, u, off) || box_near(S, cb, rb, rc, ld4(S.rq[r]), ld3(S.rh[r]), off);
  }
  const int st = u - NF;
  if (st >= ns_used) return false;
  return box_sdf_val(rc - ld3(S.stc[st]), ld3(S.sth[st])) <= rr0 + off;
}

// part 0: candidates [0, n1 + n2) - bricks against static bodies and against each other, which need nothing of this substep's robot
// state; part 1: [n1 + n2, ntot) - the robot boxes, after the forward kinematics.  Lanes 64.. of the block share a part's candidates
// (stride NT - 64) and set bit idx / NT of word idx % NT for a hit (integer atomics on a word that is zero between substeps).
template <int NT>
__device__ __forceinline__ void broad_mask_part(const SdxConst* C, PhysLds& S, int tid, int part) {
  const sdx_scene_desc& sc = C->sc;
  constexpr int ns = SDX_MAX_STATIC, per = NF + SDX_MAX_STATIC;
  constexpr int n1 = NF * ns, n2 = NF * (NF - 1) / 2;
  const int ntot = n1 + n2 + sc.n_rbox * per;
  const int lo = part == 0 ? 0 : n1 + n2, hi = part == 0 ? n1 + n2 : ntot;
#pragma unroll 1
  for (int idx = lo + tid - 64; idx < hi; idx += NT - 64)
    if (candidate(S, idx, n1, n2, ns, per, sc.n_static, sc.contact_offset)) atomicOr(&S_BMASK(S)[idx % NT], 1u << (idx / NT));
}

template <int NT>
__device__ __forceinline__ void collide(const SdxConst* C, PhysLds& S, int tid, long long* dbg, int abl_bits) {
  const sdx_scene_desc& sc = C->sc;
  const float off = sc.contact_offset;
  constexpr int ns = SDX_MAX_STATIC, per = NF + SDX_MAX_STATIC;
  // ---- broadphase over BODY pairs: lane tid owns candidates tid, tid + NT, ... (ntot <= 16 NT: sdx_create checks); hits as a bit mask; ONE block scan places them (lane-major order)
  constexpr int n1 = NF * ns, n2 = NF * (NF - 1) / 2;
  // (round 6) the candidate tests themselves have run BEFORE this function, on the lanes of waves 1..7 while wave 0 was alone with the
  // forward kinematics, the factorisation and the drive (broad_mask_part; 19 k cycles per substep when they ran here): the hits of lane
  // tid's candidates tid, tid + NT, ... are the bits of S_BMASK[tid] - the same mask, whoever computed it
  const uint32_t mask = ABL(32) ? 0u : S_BMASK(S)[tid];
  SSTAMP(32);
  int np;
  {
    // at most 16 candidates per lane: (72 * 8 + 72 * 71 / 2 + 40 * 80) / 512 < 13 (5 bits of the count, 4 bits of the pair rank)
    int pos = block_scan_small<NT, 5>(S, __popc(mask), tid, &np);
    uint32_t m = mask;
    while (m) {
      const int it = __ffs(m) - 1;
      m &= m - 1;
      // bits 16..28: rank of the candidate in this list's order (lane-major: tid * 16 + trip), the same from solve to solve: the
      // major part of the warm-start key, ascending along the pair list and therefore along the contact list
      if (pos < MAXP) S_PAIRS(S)[pos] = pair_code(tid + it * NT, n1, n2, ns, per) | ((uint32_t)(tid * 16 + it) << 16);
      ++pos;
    }
  }
  int pairs_lost = np > MAXP;   // more candidate pairs than the list holds: the excess (lane-major order) is not tested
  if (np > MAXP) np = MAXP;
  __syncthreads();
  SSTAMP(40);
  SCOUNT(48, np);
  // ---- separating-axis pass over the bounding boxes: a body pair that a face axis of either bounding box separates by the whole contact
  // offset cannot produce a contact (its boxes lie inside the bounding boxes); about half of the sphere-vs-box candidates leave the list
  // here.  Lane tid looks at the consecutive pairs tid * q .. tid * q + q - 1 (order preserved).
  {
    constexpr int QMAX = (MAXP + NT - 1) / NT;
    static_assert(QMAX <= 3, "three survivor slots per lane");
    const int q = (np + NT - 1) / NT;
    uint32_t c0 = 0, c1 = 0, c2 = 0;
    int kept = 0;
#pragma unroll
    for (int k = 0; k < QMAX; ++k) {
      const int pi = tid * q + k;
      if (k < q && pi < np) {
        const uint32_t pr = S_PAIRS(S)[pi];
        const int bb = (pr >> 8) & 0xff;
        const Box A = load_bound(S, pr & 0xff), Bx = load_bound(S, bb);
        bool sep = dir_setup(A, Bx, off).smax >= off;
        if (!sep && bb < STATIC0) sep = dir_setup(Bx, A, off).smax >= off;
        if (!sep) {
          if (kept == 0) c0 = pr; else if (kept == 1) c1 = pr; else c2 = pr;
          ++kept;
        }
      }
    }
    int np2;
    const int pos = block_scan_small<NT, 2>(S, kept, tid, &np2);   // its barrier comes after every lane's reads of the old list
    if (kept > 0) S_PAIRS(S)[pos] = c0;
    if (kept > 1) S_PAIRS(S)[pos + 1] = c1;
    if (kept > 2) S_PAIRS(S)[pos + 2] = c2;
    np = np2;
    __syncthreads();
  }
  SSTAMP(41);
  SCOUNT(49, np);
  // ---- box pairs of the surviving body pairs, FOUR lanes per body pair.  A survivor is either
  //   CONVEX - both sides stand for one convex shape (a brick as the slab compound of its hull, a robot box, a single-box static): it
  //   contributes ONE box pair, the one with the smallest separation bound sigma = the largest face-axis separation of its directions
  //   (oracle: collide_bodies; ties: the first).  Lane `sub` of the quad computes sigma of box pair `sub`, a DPP minimum over the quad finds
  //   the winner, whose index goes to bits 29..30 of the pair word, bit 31 set (no box pair below the contact offset: the pair gets 0
  //   candidates); or
  //   COMPOUND - a hollow brick or the studded base plate on either side: every (box of A) x (box of B) is a candidate.
  // The exclusive prefix sum S_OFF of the candidate counts maps a body pair to its first candidate box pair.
  int nbp;   // candidate box pairs
  {
    static_assert(SDX_MAX_SUB * SDX_MAX_SUB <= 4, "the box pairs of a convex body pair: one per lane of a quad, the winner in two bits");
    const int sub = tid & 3;
#pragma unroll 1
    for (int base = 0; base < np; base += NT / 4) {
      const int pi = base + (tid >> 2);
      const bool on = pi < np;
      const uint32_t pr = on ? S_PAIRS(S)[pi] : 0u;
      const int ba = pr & 0xff, bb = (pr >> 8) & 0xff;
      const int na = on ? box_nsub(S, ba) : 1, nbx = on ? box_nsub(S, bb) : 1;
      int cnt = na * nbx;
      const bool convex = on && !(ba < NF && brick_hollow(S, ba)) && !(bb < NF && brick_hollow(S, bb)) && !(bb >= STATIC0 && nbx > 1);
      float sg = 1e30f;
      if (convex && sub < cnt) {
        const int sa = sub / nbx, sb = sub - sa * nbx;
        LOAD_BOX2(ba, sa, bb, sb)
        sg = dir_setup(A, Bx, off).smax;
        if (samples_b(bb, sb)) sg = fmaxf(sg, dir_setup(Bx, A, off).smax);
      }
      // minimum of (sigma, sub) over the quad (every lane of the wave executes the two DPP steps)
      int wi = sub;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const float og = st == 0 ? __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sg), 0xB1, 0xF, 0xF, true))
                                 : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sg), 0x4E, 0xF, 0xF, true));
        const int oi = st == 0 ? __builtin_amdgcn_update_dpp(0, wi, 0xB1, 0xF, 0xF, true) : __builtin_amdgcn_update_dpp(0, wi, 0x4E, 0xF, 0xF, true);
        const bool take = og < sg || (og == sg && oi < wi);
        sg = take ? og : sg;
        wi = take ? oi : wi;
      }
      if (on && sub == 0) {
        if (convex) {
          cnt = sg < off ? 1 : 0;
          S_PAIRS(S)[pi] = pr | ((uint32_t)(wi & 3) << 29) | 0x80000000u;
        }
        S_OFF(S)[pi] = cnt;
      }
    }
    __syncthreads();
    SSTAMP(42);
    // exclusive prefix sum of the counts, three body pairs per lane
    static_assert(MAXP <= 3 * NT, "three body pairs per lane");
    const int i0 = 3 * tid, i1 = 3 * tid + 1, i2 = 3 * tid + 2;
    const int n0 = i0 < np ? S_OFF(S)[i0] : 0, n1 = i1 < np ? S_OFF(S)[i1] : 0, n2 = i2 < np ? S_OFF(S)[i2] : 0;
    const int boff = block_scan_small<NT, 10>(S, n0 + n1 + n2, tid, &nbp);   // <= 3 x (8 boxes of a hollow brick x 33 of a base plate) per lane: 10 bits
    if (i0 < np) S_OFF(S)[i0] = boff;
    if (i1 < np) S_OFF(S)[i1] = boff + n0;
    if (i2 < np) S_OFF(S)[i2] = boff + n0 + n1;
    if (tid == 0) S_OFF(S)[np] = nbp;
    __syncthreads();
  }
  SSTAMP(33);
  SCOUNT(50, nbp);
  // ---- expansion: candidate box pair t = S_OFF[p] + (sub a * nsub(b) + sub b) of body pair p.  Lane tid tests the CONSECUTIVE candidates
  // tid * q2 .. (so the survivors come out in ascending (pair rank, box pair) order: the order of the warm-start keys) with the
  // separating-axis test of the two boxes; survivors as a bit mask, one block scan, then the lane walks its range again and writes
  // (box a | sub a << 7 | box b << 11 | sub b << 19) and (pair rank << 9 | box pair index) per survivor.
  int nsp;
  {
    if (nbp > 32 * NT) { nbp = 32 * NT; pairs_lost = 1; }
    const int q2 = (nbp + NT - 1) / NT;
    const int t0 = tid * q2, t1 = min(t0 + q2, nbp);
    int p = 0;
    if (t0 < t1) {   // the body pair that holds candidate t0: last p with S_OFF[p] <= t0
      int lo = 0, hi = np;
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (S_OFF(S)[mid] <= t0) lo = mid; else hi = mid; }
      p = lo;
    }
    uint32_t keep = 0;
    {
      int pp = p, pend = t0 < t1 ? S_OFF(S)[pp + 1] : 0;
      uint32_t pr = t0 < t1 ? S_PAIRS(S)[pp] : 0;
      int nbs = t0 < t1 ? box_nsub(S, (pr >> 8) & 0xff) : 1;
#pragma unroll 1
      for (int t = t0; t < t1; ++t) {
        while (t >= pend) { ++pp; pend = S_OFF(S)[pp + 1]; pr = S_PAIRS(S)[pp]; nbs = box_nsub(S, (pr >> 8) & 0xff); }
        bool sep = false;
        if (!(pr >> 31)) {   // (a convex pair's only candidate has passed its test in the pass above)
          const int sidx = t - S_OFF(S)[pp];
          const int ba = pr & 0xff, bb = (pr >> 8) & 0xff, sa = sidx / nbs, sb = sidx - sa * nbs;
          LOAD_BOX2(ba, sa, bb, sb)
          sep = dir_setup(A, Bx, off).smax >= off;
          if (!sep && samples_b(bb, sb)) sep = dir_setup(Bx, A, off).smax >= off;
        }
        if (!sep) keep |= 1u << (t - t0);
      }
    }
    const int pos0 = block_scan_small<NT, 6>(S, __popc(keep), tid, &nsp);
    {
      int pos = pos0, pp = p, pend = t0 < t1 ? S_OFF(S)[pp + 1] : 0;
      uint32_t pr = t0 < t1 ? S_PAIRS(S)[pp] : 0;
      int nbs = t0 < t1 ? box_nsub(S, (pr >> 8) & 0xff) : 1;
      uint32_t m = keep;
#pragma unroll 1
      while (m) {
        const int t = t0 + __ffs(m) - 1;
        m &= m - 1;
        while (t >= pend) { ++pp; pend = S_OFF(S)[pp + 1]; pr = S_PAIRS(S)[pp]; nbs = box_nsub(S, (pr >> 8) & 0xff); }
        const int sidx = (pr >> 31) ? (int)((pr >> 29) & 3u) : t - S_OFF(S)[pp];
        const int ba = pr & 0xff, bb = (pr >> 8) & 0xff, sa = sidx / nbs, sb = sidx - sa * nbs;
        if (pos < MAXSP) {
          S_SP0(S)[pos] = (uint32_t)ba | ((uint32_t)sa << 7) | ((uint32_t)bb << 11) | ((uint32_t)sb << 19);
          S_SP1(S)[pos] = (((pr >> 16) & 0x1fffu) << 9) | (uint32_t)sidx;
        }
        ++pos;
      }
    }
    if (nsp > MAXSP) { nsp = MAXSP; pairs_lost = 1; }
    __syncthreads();
  }
  SSTAMP(37);
  SCOUNT(51, nsp);
  // ---- narrowphase in two parts.  (1) lane = candidate box pair: the <= 4 samples of the two directions that become contacts
  // (pair_contacts); a block prefix sum of the counts gives every contact its place in pair order, and the lane leaves TWO words per
  // contact there: (boxes, direction, sample) and the contact's identity.  (2) lane = contact (below): geometry of that sample.
  // (Rounds 1-2 emitted from the pair lanes, which kept both boxes alive across the scan and ran up to eight emissions in a row on one lane
  // while its neighbours idled: 36 spilled registers, 120 MB of scratch writes per launch.)
  // Capacity rule (oracle: collide()): a list that would exceed MAXC contacts is rebuilt without its speculative
  // part - only samples that touch or penetrate (inclusion threshold 0 instead of the contact offset); what still does not fit is
  // dropped in enumeration order and counted.  The second pass is the same code in a run-time loop (block-uniform trip count).
  int nc = 0, rebuilt = 0;
  float incl = off;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    nc = 0;
#pragma unroll 1
    for (int base = 0; base < nsp; base += NT) {
      const int pi = base + tid;
      uint32_t s1 = 0, s2 = 0, w0 = 0, w1 = 0;
      int k = 0;
      if (pi < nsp && !ABL(16)) {
        w0 = S_SP0(S)[pi]; w1 = S_SP1(S)[pi];
        const int ba = w0 & 0x7f, sa = (w0 >> 7) & 0xf, bb = (w0 >> 11) & 0xff, sb = (w0 >> 19) & 0x3f;
        LOAD_BOX2(ba, sa, bb, sb)
        k = pair_contacts(S, A, Bx, samples_b(bb, sb), off, incl, &s1, &s2);
      }
      if (pass == 0 && base == 0) SSTAMP(38);
      int tot;
      int c = nc + block_scan_small<NT, 3>(S, k, tid, &tot);
      // (the descriptors go to the impulse rows P[0] / P[1]: the box pair list in cp / cn stays intact for the second pass)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (s1) {
          const int sm = __ffs(s1) - 1;
          s1 &= s1 - 1;
          if (c < MAXC) { S.P[0][c] = __int_as_float((int)(w0 | ((uint32_t)sm << 26))); S.P[1][c] = __int_as_float((int)((w1 << 6) | (uint32_t)sm)); }
          ++c;
        } else if (s2) {
          const int sm = __ffs(s2) - 1;
          s2 &= s2 - 1;
          if (c < MAXC) { S.P[0][c] = __int_as_float((int)(w0 | (1u << 25) | ((uint32_t)sm << 26))); S.P[1][c] = __int_as_float((int)((w1 << 6) | 0x20u | (uint32_t)sm)); }
          ++c;
        }
      }
      nc += tot;
    }
    if (nc <= MAXC || pass == 1) break;
    incl = 0.0f;      // (the scans' barriers order this pass's LDS writes before the next pass's)
    rebuilt = 1;
  }
  __syncthreads();   // the box pair list in cp / cn is dead from here on
  SSTAMP(39);
  // ---- (2) lane = contact: point, normal, separation of sample s of box A against box B (oracle: emit_mask); the descriptor of contact c
  // sits in P[0][c] / P[1][c], which the same lane overwrites with the results
  {
    const int ncc = nc < MAXC ? nc : MAXC;
#pragma unroll 1
    for (int c = tid; c < ncc; c += NT) {
      const uint32_t d = (uint32_t)__float_as_int(S.P[0][c]), key = (uint32_t)__float_as_int(S.P[1][c]);
      const int dirb = (d >> 25) & 1, sidx = d >> 26;
      const int b0 = d & 0x7f, s0 = (d >> 7) & 0xf, b1 = (d >> 11) & 0xff, sb1 = (d >> 19) & 0x3f;
      const int ba = dirb ? b1 : b0, bb = dirb ? b0 : b1, sa = dirb ? sb1 : s0, sb = dirb ? s0 : sb1;
      LOAD_BOX2(ba, sa, bb, sb)
      const Dir D = dir_setup(A, Bx, off);
      const f3 pb = ((D.t + D.ex * S.samp[sidx][0]) + D.ey * S.samp[sidx][1]) + D.ez * S.samp[sidx][2];   // (the LDS copy: a per-lane index into __constant__ memory is a global load)
      f3 g;
      const float sd = sample_contact(D, pb, Bx.h, &g);
      const f3 n = qrot(Bx.q, g);
      const f3 pw = Bx.c + qrot(Bx.q, pb);
      const f3 p = pw - n * (0.5f * sd);
      S.cp[0][c] = p.x; S.cp[1][c] = p.y; S.cp[2][c] = p.z;
      S.cn[0][c] = n.x; S.cn[1][c] = n.y; S.cn[2][c] = n.z;
      S.P[0][c] = sd;                                       // staged for the owner lane of contact c (solver set-up)
      S.P[1][c] = __int_as_float(box_body(S, ba) | (box_body(S, bb) << 8));
      S_CKEY(S)[c] = key;   // (pair rank << 9 | box pair) << 6 | direction << 5 | sample: 28 bits
    }
  }
  if (tid == 0) {
    S.overflow = nc > MAXC ? nc - MAXC : 0;
    S.nc = nc > MAXC ? MAXC : nc;
    S.np = np;
    S.rebuilt = rebuilt | (pairs_lost << 1);
  }
  __syncthreads();
}

// ---------------------------------------------------------------- E: solver
// row weight of a robot-side contact: w = J Hinv J^T with J[j] = (a_j x (p - p_j)) . d over the <= 11 dofs on the link's path.  Hinv is
// symmetric: every off-diagonal entry is loaded once.  ONE instance per kernel (the caller loops over the three row directions at run
// time): three unrolled copies side by side cost the solver loop its registers (33 scratch reloads per iteration).
__device__ __forceinline__ float robot_w(const PhysLds& S, int k, f3 p, f3 d) {
  float Jp[11];
  int idx[11];
  uint32_t m = S.anc[k];
#pragma unroll
  for (int q = 0; q < 11; ++q) {
    const int j = m ? __ffs(m) - 1 : ND;      // exhausted path: the padding dof (zero axis)
    m &= m - 1;
    idx[q] = j < ND ? j : 0;
    Jp[q] = dot(cross(ld3(S.la[j + 1]), p - ld3(S.bp[NF + j + 1])), d);
  }
  float acc = 0.0f;
#pragma unroll
  for (int q = 0; q < 11; ++q) {
    float t = 0.5f * S.A[idx[q]][idx[q]] * Jp[q];
#pragma unroll
    for (int r = q + 1; r < 11; ++r) t += S.A[idx[q]][idx[r]] * Jp[r];
    acc += Jp[q] * t;
  }
  return 2.0f * acc;
}

template <int NT, bool WARM>
// Warm start (oracle: solve()): wcount / wkey / wlam are THIS env's impulse cache in HBM.  A contact that existed
// in the previous solve (same pair, direction and sample) starts from sc.warm_start x the impulses it ended with; iteration -1 of the
// loop below spreads those impulses over the bodies with the gather machinery of a normal iteration.  WARM is a template parameter:
// the default (cold) solver carries none of this code (it cost 2.6 % of the kernel as a run-time switch).
__device__ __forceinline__ void solve(const SdxConst* C, PhysLds& S, int tid, float h, bool last_substep, long long* dbg,
                                      int32_t* wcount, uint32_t* wkey, float* wlam, int abl_bits) {
  constexpr int CPT = MAXC / NT;   // contact rows owned by one lane
  constexpr int NB = NF + NL;      // bodies with a CSR list: bricks 0..71, links 72..95
  static_assert(CPT * NT == MAXC, "NT must divide SDX_MAXC");
  static_assert(CPT * 11 <= 64 && MAXC < 0x7ff, "11-bit cache positions of a lane's contacts in one 64-bit word");
  const sdx_scene_desc& sc = C->sc;
  const int nc = S.nc;
  const float mu = sc.friction, relax = sc.jacobi_relax;
  SSTAMP(16);
  // ---- what stays in registers for the whole solve: body ids (rows of the body table), target normal velocity, accumulated
  // impulses, un-split inverse masses of both sides
  int ab[CPT];
  float vtgt[CPT];
  uint32_t ckey[CPT];
  const float beta = sc.warm_start;
  const float inv_age = sc.warm_age > 0.0f ? 1.0f / sc.warm_age : 1e30f;
  // depth gate expressed on the velocity target the lane keeps anyway: sep < -WARM_DEPTH * offset <=> vtgt > baumgarte * depth / h
  const float wdeep = sc.baumgarte * (WARM_DEPTH * sc.contact_offset) / h;
  int nold = WARM ? *wcount : 0;   // block-uniform
  if (nold > MAXC) nold = MAXC;
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const int c = tid + q * NT;
    ab[q] = BODY_W | (BODY_W << 8);
    vtgt[q] = 0.0f;
    ckey[q] = 0xffffffffu;
    if (c < nc) {
      const float sep = S.P[0][c];
      ab[q] = __float_as_int(S.P[1][c]);
      vtgt[q] = sep > 0 ? -sep / h : fminf(sc.baumgarte * (-sep) / h, sc.max_depenetration_vel);
      if (WARM) ckey[q] = S_CKEY(S)[c];
    }
  }
  // the previous solve's keys into the free third impulse row (the pair list that lived there is dead)
  if (WARM) for (int i = tid; i < nold; i += NT) S.P[2][i] = __int_as_float((int)wkey[i]);
  SSTAMP(23);
  // ---- CSR of contact sides per body (bricks AND robot links), ascending contact index inside a body
  for (int i = tid; i < NB; i += NT) { S.ecount[i] = 0; S.efill[i] = 0; }
  if (last_substep) for (int i = tid; i < NL * 3; i += NT) (&S.cf[0][0])[i] = 0.0f;
  __syncthreads();   // also: every lane has read its (separation, ids, key) out of the staging rows, which the fill list reuses below
  // warm-start match: the old keys ascend in their pair part (bits 6..27: body pair rank, box pair), so the pair's first old contact is a lower bound away; the
  // <= 4 contacts of a pair are then compared exactly.  wmatch packs the matched positions (11 bits each, 0x7ff = none); the
  // impulses themselves are fetched where the lam registers are born.  The new keys replace the old ones in HBM right away (the
  // old ones are in LDS now; the old impulses stay untouched until the end of this solve).
  uint64_t wmatch = ~0ull;
  uint32_t wage = 0;   // 8 bits per contact: consecutive solves it has existed (0: new in this solve)
  // lower bound of every slot's pair part among the old keys: a branch-free bisection whose trip count depends on nold only, so the lane's
  // CPT searches advance together - one LDS round trip per halving for all of them (round 6; one search after the other was 10 dependent
  // round trips per slot)
  int wlo[CPT];
#pragma unroll
  for (int q = 0; q < CPT; ++q) wlo[q] = 0;
  if (WARM && nold > 0) {
    int nrem = nold;
    while (nrem > 1) {
      const int half = nrem >> 1;
      uint32_t kv[CPT];
#pragma unroll
      for (int q = 0; q < CPT; ++q) kv[q] = (uint32_t)__float_as_int(S.P[2][wlo[q] + half - 1]);
#pragma unroll
      for (int q = 0; q < CPT; ++q) SDX_PIN1(kv[q]);
#pragma unroll
      for (int q = 0; q < CPT; ++q) wlo[q] = ((kv[q] & 0x0fffffffu) >> 6) < (ckey[q] >> 6) ? wlo[q] + half : wlo[q];
      nrem -= half;
    }
#pragma unroll
    for (int q = 0; q < CPT; ++q) wlo[q] += (((uint32_t)__float_as_int(S.P[2][wlo[q]]) & 0x0fffffffu) >> 6) < (ckey[q] >> 6) ? 1 : 0;
  }
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const int c = tid + q * NT;
    if (WARM && c < nc) {
      const uint32_t key = ckey[q];
      uint32_t age = 0;
      if (nold > 0) {
        const int lo = wlo[q];
        int found = 0x7ff;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = lo + u;
          const uint32_t ok = i < nold ? (uint32_t)__float_as_int(S.P[2][i]) : 0xffffffffu;
          if ((ok & 0x0fffffffu) == key) { found = i; age = (ok >> 28) + 1u; }
        }
        wmatch = (wmatch & ~(0x7ffull << (11 * q))) | ((uint64_t)found << (11 * q));
      }
      wage |= age << (8 * q);
      wkey[c] = key | ((age < 15u ? age : 15u) << 28);   // bits 28..31: the number of consecutive solves this contact has existed before (saturating at 15: the ramp is over after warm_age <= 16 solves)
    }
  }
#pragma unroll
  for (int q = 0; q < CPT; ++q)
    if (tid + q * NT < nc) {
      const int a = ab[q] & 0xff, b = (ab[q] >> 8) & 0xff;
      if (a != BODY_W) atomicAdd(&S.ecount[a], 1);
      if (b != BODY_W) atomicAdd(&S.ecount[b], 1);
    }
  __syncthreads();
  SSTAMP(24);
  if (tid < 64) {   // exclusive prefix over the 96 bodies on wave 0: two per lane, shuffle scan
    const int i0 = 2 * tid, i1 = 2 * tid + 1;
    const int c0 = i0 < NB ? S.ecount[i0] : 0, c1 = i1 < NB ? S.ecount[i1] : 0;
    int incl = c0 + c1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (tid >= o) incl += up;
    }
    const int excl = incl - (c0 + c1);
    if (i0 < NB) S.eoff[i0] = excl;
    if (i1 < NB) S.eoff[i1] = excl + c0;
    if (i1 == NB - 1 || i0 == NB - 1) S.eoff[NB] = incl;
  } else if (tid < 128) {
    // ---- balanced brick gather (round 6): wave 1, beside the prefix sum.  The per-pass gather [D] used to give every brick 4 lanes: the
    // brick with the longest list (40 - 60 sides in a settled pile, 19 on average) kept its lanes for 3 - 4 trips while most quads idled -
    // 4 - 5 k of the pass's 9 k cycles.  Now a brick with n sides gets L = 1, 2, 4, 8 or 16 CONSECUTIVE lanes - the power of two at or
    // above ceil(n / K) - each summing a contiguous slice of ceil(n / L) <= K sides; the slices meet in a segmented row scan (DPP).
    // Layout = counting sort by size, largest first: every brick then starts at a multiple of its size and never straddles a 16-lane
    // row.  K is the smallest of 4, 6, 8, 12, ... 96 whose layout fits the brick lanes (all 512 when the hand touches nothing, else
    // those of waves 3..7); lane t holds bricks t and t + 64.
    constexpr int GKT = 10;
    static_assert(NF <= 128, "two bricks per lane of wave 1");
    const int t = tid - 64;
    const bool robot = __ballot(t < NL && S.ecount[NF + (t < NL ? t : 0)] > 0) != 0ull;
    const int nlanes = robot ? NT - NL * 8 : NT, base = robot ? NL * 8 : 0;
    const int n0 = S.ecount[t < NF ? t : 0], n1 = t + 64 < NF ? S.ecount[t + 64 < NF ? t + 64 : 0] : 0;
    const uint64_t lt = t == 0 ? 0ull : (~0ull >> (64 - t));
    int P0 = 0, P1 = 0, start0 = 0, start1 = 0;
#pragma unroll 1
    for (int tk = 0; tk < GKT; ++tk) {
      const int K = (4 + 2 * (tk & 1)) << (tk >> 1), Kmul = (65536 + K - 1) / K;   // ceil(n / K) = ((n + K - 1) * Kmul) >> 16 for n < 2048
      int L0 = ((n0 + K - 1) * Kmul) >> 16, L1 = ((n1 + K - 1) * Kmul) >> 16;
      // size class: 0 (no sides), else the power of two >= min(L, 16)
      P0 = L0 == 0 ? 0 : (L0 > 8 ? 16 : (L0 > 4 ? 8 : (L0 > 2 ? 4 : L0)));
      P1 = L1 == 0 ? 0 : (L1 > 8 ? 16 : (L1 > 4 ? 8 : (L1 > 2 ? 4 : L1)));
      int run = 0;   // lanes of the classes placed so far (wave-uniform)
      start0 = start1 = 0;
#pragma unroll
      for (int c = 16; c >= 1; c >>= 1) {
        const uint64_t b0 = __ballot(P0 == c), b1 = __ballot(P1 == c);
        const int c0 = __popcll(b0);
        if (P0 == c) start0 = run + c * __popcll(b0 & lt);
        if (P1 == c) start1 = run + c * (c0 + __popcll(b1 & lt));
        run += c * (c0 + __popcll(b1));
      }
      if (run <= nlanes) break;   // (K = 96 always fits: at most 16 bricks can have more than 96 sides)
    }
    if (t < NF) S.gtab[t] = (uint32_t)(start0 + base) | ((uint32_t)P0 << 9);
    if (t + 64 < NF) S.gtab[t + 64] = (uint32_t)(start1 + base) | ((uint32_t)P1 << 9);
  }
  S_LANEMAP(S)[tid] = 0;   // (wave 0 and 1 after their work: the row of the old warm-start keys is free since the barrier above)
  __syncthreads();
  SSTAMP(25);
  if (tid < NF) {   // lane -> (brick, lane of the brick, last lane?) of the chosen packing
    const uint32_t w = S.gtab[tid];
    const int start = w & 511, L = w >> 9;
    for (int j = 0; j < L; ++j) S_LANEMAP(S)[start + j] = (unsigned short)(0x8000 | tid | (j << 7) | ((j == L - 1) << 11));
  }
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const int c = tid + q * NT;
    if (c < nc) {
      const int a = ab[q] & 0xff, b = (ab[q] >> 8) & 0xff;
      if (a != BODY_W) { const int i = S.eoff[a] + atomicAdd(&S.efill[a], 1); S_ENT2(S)[i] = (unsigned short)c; S_EBODY2(S)[i] = (unsigned char)a; }
      if (b != BODY_W) { const int i = S.eoff[b] + atomicAdd(&S.efill[b], 1); S_ENT2(S)[i] = (unsigned short)(c | 0x8000); S_EBODY2(S)[i] = (unsigned char)b; }
    }
  }
  __syncthreads();
  SSTAMP(26);
  const int rbeg = S.eoff[NF], nrob = S.eoff[NB] - rbeg;   // the robot's sides: entries [rbeg, rbeg + nrob), grouped by link
  const bool has_robot = nrob > 0;                          // block-uniform
  // this lane's slice of a brick's list (balanced gather): brick | lane of the brick << 7 | last lane << 11 | first entry << 12 | entries << 24;
  // 0 = none.  (Read here: the row that holds the lane map is rewritten by the link-inertia stages below.)
  uint32_t cdesc = 0;
  {
    const uint32_t lm = S_LANEMAP(S)[tid];
    if (lm & 0x8000u) {
      const int b = lm & 0x7f, j = (lm >> 7) & 15;
      const int L = (int)(S.gtab[b] >> 9), n = S.ecount[b];
      const int c = (n + L - 1) / L;                      // entries per lane of this brick
      int cnt = n - j * c;
      cnt = cnt < 0 ? 0 : (cnt > c ? c : cnt);
      cdesc = (lm & 0xfffu) | ((uint32_t)(S.eoff[b] + j * c) << 12) | ((uint32_t)cnt << 24);
    }
  }
  if (tid == 0) S.nrob = nrob;
  SCOUNT(52, nc); SCOUNT(53, S.eoff[NB]); SCOUNT(54, nrob);
  // rank pass: entry -> position = number of entries of the same body with a smaller contact index (a contact touches a body at
  // most once, so indices are distinct) => every body's list is in ascending contact order, deterministically
  {
    const int total = S.eoff[NB];
    for (int i = tid; i < total; i += NT) {
      const int body = S_EBODY2(S)[i];
      const int o = S.eoff[body], n = S.eoff[body + 1] - o;
      const unsigned short v = S_ENT2(S)[i];
      const int key = v & 0x7fff;
      int rank = 0;
      if (ABL(64)) rank = i - o;
      else
#pragma unroll 4
      for (int j = 0; j < n; ++j) rank += (S_ENT2(S)[o + j] & 0x7fff) < key;
      S.ent[o + rank] = v;
    }
  }
  __syncthreads();   // rank pass complete (ent final); the fill list in the P rows is dead from here on
  SSTAMP(27);
  SSTAMP(28);
  // ---- robot: operational-space inverse inertia of every link that carries a contact, Lam_k = J_k Hinv J_k^T (6 x 6, symmetric, about
  // the link origin o_k; J_k = the link's geometric Jacobian over the <= 11 dofs of its path), built by the whole block in three stages
  // through the (free) impulse rows.  A robot-side row weight is then g^T Lam_k g with g = (d, (p - o_k) x d): no per-contact walk over
  // the path, no exchange between the lanes that own a contact and the lanes that own a link.
  uint32_t touched = 0;   // links that carry contacts in this substep (block-uniform)
  float* const W = &S.P[0][0];
  constexpr int W_T = 0, W_U = NL * 66, W_L = 2 * NL * 66, W_J = 2 * NL * 66 + NL * 21;   // T = J_k [6][11], U = J_k Hinv_sub [6][11], Lam [21], path dofs [11]
  static_assert(W_J + NL * 11 <= 3 * MAXC, "the three stages fit the impulse rows");
  if (has_robot) {
    for (int k = 0; k < NL; ++k) if (S.eoff[NF + k + 1] > S.eoff[NF + k]) touched |= 1u << k;
    // stage 1: lane = (link k, dof j on its path): column of J_k and the dof's slot
    for (int idx = tid; idx < NL * ND; idx += NT) {
      const int k = idx / ND, j = idx % ND;
      const uint32_t ak = S.anc[k];
      if (((touched >> k) & 1u) && ((ak >> j) & 1u)) {
        const int slot = __popc(ak & ((1u << j) - 1u));
        const f3 a = ld3(S.la[j + 1]);
        const f3 lin = cross(a, ld3(S.bp[NF + k]) - ld3(S.bp[NF + j + 1]));
        float* t = W + W_T + k * 66 + slot;
        t[0] = lin.x; t[11] = lin.y; t[22] = lin.z; t[33] = a.x; t[44] = a.y; t[55] = a.z;
        reinterpret_cast<int*>(W + W_J)[k * 11 + slot] = j;
      }
    }
    __syncthreads();
    // stage 2: lane = (link, row r, slot q): U[r][q] = sum_s T[r][s] Hinv[j_s][j_q]
    for (int idx = tid; idx < NL * 66; idx += NT) {
      const int k = idx / 66, rq = idx % 66, r = rq / 11, q = rq % 11;
      const int m = __popc(S.anc[k]);
      if (((touched >> k) & 1u) && q < m) {
        const int* pj = reinterpret_cast<const int*>(W + W_J) + k * 11;
        const float* t = W + W_T + k * 66 + r * 11;
        // (all 11 slots' operands in flight together - slots past the path read slot 0 and are not added; the run-time loop over the m
        // slots was two dependent LDS round trips per slot)
        int pjq = pj[q], pjv[11];
        float tv[11], av[11];
#pragma unroll
        for (int sidx = 0; sidx < 11; ++sidx) { pjv[sidx] = pj[sidx < m ? sidx : 0]; tv[sidx] = t[sidx < m ? sidx : 0]; }
        SDX_PIN1(pjq);
#pragma unroll
        for (int sidx = 0; sidx < 11; ++sidx) SDX_PIN1(pjv[sidx]);
        const float* Aq = S.A[pjq];
#pragma unroll
        for (int sidx = 0; sidx < 11; ++sidx) av[sidx] = Aq[pjv[sidx]];   // Hinv is symmetric: row j_q instead of column j_q
#pragma unroll
        for (int sidx = 0; sidx < 11; ++sidx) { SDX_PIN1(av[sidx]); SDX_PIN1(tv[sidx]); }
        float acc = 0.0f;
#pragma unroll
        for (int sidx = 0; sidx < 11; ++sidx) if (sidx < m) acc += tv[sidx] * av[sidx];
        W[W_U + idx] = acc;
      }
    }
    __syncthreads();
    // stage 3: lane = (link, entry (i, j <= i) of the lower triangle): Lam[i][j] = sum_s U[i][s] T[j][s]
    for (int idx = tid; idx < NL * 21; idx += NT) {
      const int k = idx / 21, e = idx % 21;
      if ((touched >> k) & 1u) {
        int i, j;
        tri_index(e, &i, &j);
        const int m = __popc(S.anc[k]);
        const float* u = W + W_U + k * 66 + i * 11;
        const float* t = W + W_T + k * 66 + j * 11;
        float uv[11], tv[11];
#pragma unroll
        for (int sidx = 0; sidx < 11; ++sidx) { uv[sidx] = u[sidx < m ? sidx : 0]; tv[sidx] = t[sidx < m ? sidx : 0]; }
#pragma unroll
        for (int sidx = 0; sidx < 11; ++sidx) { SDX_PIN1(uv[sidx]); SDX_PIN1(tv[sidx]); }
        float acc = 0.0f;
#pragma unroll
        for (int sidx = 0; sidx < 11; ++sidx) if (sidx < m) acc += uv[sidx] * tv[sidx];
        W[W_L + idx] = acc;
      }
    }
    __syncthreads();
  }
  SSTAMP(29);
  // (the per-lane impulse / weight registers are born only here)
  float lam[CPT][3], wA[CPT][3], wB[CPT][3];
  // ---- un-split inverse effective masses of both sides (owner lanes); zero accumulated impulses
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const int c = tid + q * NT;
    const bool on = c < nc && !ABL(128);
    const int a = ab[q] & 0xff, b = (ab[q] >> 8) & 0xff;
    f3 p = F3(0, 0, 0), n = F3(0, 0, 1);
    if (on) { p = F3(S.cp[0][c], S.cp[1][c], S.cp[2][c]); n = F3(S.cn[0][c], S.cn[1][c], S.cn[2][c]); }
    f3 t1, t2;
    tangents(n, &t1, &t2);
    const f3 dir[3] = {n, t1, t2};
    {
      // brick_w() x 6 compiled to ~7 dependent LDS round trips per call (position, orientation, type -> inertia ratios, inverse mass, one
      // after the other: 40 in a row per contact).  Here both sides' rows are loaded together, then the type rows, then the six weights are
      // plain arithmetic - the same expressions as brick_w.  A side that is not a brick reads brick 0 and is discarded.
      const bool ba = on && a < NF, bb = on && b < NF;
      const int ia = ba ? a : 0, ib = bb ? b : 0;
      f3 xa = ld3(S.bp[ia]), xb = ld3(S.bp[ib]);
      f4 qa = ld4(S.bq[ia]), qb = ld4(S.bq[ib]);
      float ima = S.bim[ia], imb = S.bim[ib];
      int ta = S.btype[ia], tb = S.btype[ib];
      SDX_PIN1(ta); SDX_PIN1(tb);
      f3 iia = ld3(S.tii[ta]), iib = ld3(S.tii[tb]);
      SDX_PIN3x4(xa, xb, iia, iib);
      SDX_PIN1(qa.x); SDX_PIN1(qa.y); SDX_PIN1(qa.z); SDX_PIN1(qa.w); SDX_PIN1(qb.x); SDX_PIN1(qb.y); SDX_PIN1(qb.z); SDX_PIN1(qb.w);
      SDX_PIN1(ima); SDX_PIN1(imb);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const f3 la = qrot(qconj(qa), cross(p - xa, dir[r])), lb = qrot(qconj(qb), cross(p - xb, dir[r]));
        const float va = ima * (1.0f + la.x * la.x * iia.x + la.y * la.y * iia.y + la.z * la.z * iia.z);
        const float vb = imb * (1.0f + lb.x * lb.x * iib.x + lb.y * lb.y * iib.y + lb.z * lb.z * iib.z);
        wA[q][r] = ba ? va : 0.0f;
        wB[q][r] = bb ? vb : 0.0f;
        lam[q][r] = 0.0f;
      }
    }
    if (has_robot && on) {
#pragma unroll 1
      for (int side = 0; side < 2; ++side) {
        const int id = side ? b : a;
        if (id >= NF && id != BODY_W) {
          const int k = id - NF;
          const float* L = W + W_L + k * 21;   // lower triangle, row-major: (0,0) (1,0) (1,1) (2,0) ...
          const f3 rr = p - ld3(S.bp[id]);
          float w3[3];
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const f3 d = dir[r], md = cross(rr, d);
            const float g[6] = {d.x, d.y, d.z, md.x, md.y, md.z};
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              float t = 0.5f * L[i * (i + 1) / 2 + i] * g[i];
#pragma unroll
              for (int j = 0; j < i; ++j) t += L[i * (i + 1) / 2 + j] * g[j];
              acc += g[i] * t;
            }
            w3[r] = 2.0f * acc;
          }
          if (side) { wB[q][0] = w3[0]; wB[q][1] = w3[1]; wB[q][2] = w3[2]; }
          else { wA[q][0] = w3[0]; wA[q][1] = w3[1]; wA[q][2] = w3[2]; }
        }
      }
    }
    // warm start only where the last solve's impulse still means something: the contact is not in deep penetration (that is recovery,
    // not rest) and the two bodies are nearly at rest relative to each other at the contact point (not an impact, not sliding)
    const int wm = (int)((wmatch >> (11 * q)) & 0x7ffull);
    if (WARM && on && wm != 0x7ff && vtgt[q] <= wdeep) {
      const f3 vrel = point_vel(S, a, p) - point_vel(S, b, p);
      // the cached impulse is trusted in proportion to the age of the contact: a contact that has existed for warm_age solves starts
      // from the full fraction, a contact the previous solve saw for the first time (an impact) from nothing
      const float bq = beta * fminf(1.0f, (float)((wage >> (8 * q)) & 0xffu) * inv_age);
      const float l0 = bq * wlam[wm];
      if (l0 > 0.0f && dot(vrel, vrel) <= WARM_SPEED * WARM_SPEED) {
        lam[q][0] = l0; lam[q][1] = bq * wlam[MAXC + wm]; lam[q][2] = bq * wlam[2 * MAXC + wm];
      }
    }
  }
  SSTAMP(30);
  // gather lanes.  Links: LL = 8 per link (tid = 8 * link + sub: the same lanes run the robot section; only when the hand touches something),
  // lane sub sums entries sub, sub + 8, ... of its link's list.  Bricks: the slice of cdesc (balanced gather, above) - kept in gbeg.
  // (The old fixed map - GL = 4 lanes per brick after the link lanes - still decides which lane computes a brick's bK below.)
  constexpr int LL = 8;
#ifndef SDX_GU
#define SDX_GU 4
#endif
  constexpr int GU = SDX_GU;   // entries per lane and trip of the gather
  static_assert(NF * GL + NL * LL <= NT, "gather lanes");
  const bool llane = tid < NL * LL;
  const int gbody = llane ? NF + tid / LL : (tid - NL * LL) / GL, gsub = llane ? tid % LL : (tid - NL * LL) % GL;
  const bool glane = llane || gbody < NF;
  int gbeg = 0, gend = 0;
  if (glane) {
    gbeg = S.eoff[gbody] + gsub; gend = S.eoff[gbody + 1];
    if (gbody < NF && gsub == 0) {
      const f4 q = ld4(S.bq[gbody]);
      const f3 ex = qrot(q, F3(1, 0, 0)), ey = qrot(q, F3(0, 1, 0)), ez = qrot(q, F3(0, 0, 1));   // columns of R
      const float* ii = S.tii[S.btype[gbody]];
      const float i0 = S.bim[gbody] * ii[0], i1 = S.bim[gbody] * ii[1], i2 = S.bim[gbody] * ii[2];
      S.bK[gbody][0] = i0 * ex.x * ex.x + i1 * ey.x * ey.x + i2 * ez.x * ez.x;
      S.bK[gbody][1] = i0 * ex.y * ex.y + i1 * ey.y * ey.y + i2 * ez.y * ez.y;
      S.bK[gbody][2] = i0 * ex.z * ex.z + i1 * ey.z * ey.z + i2 * ez.z * ez.z;
      S.bK[gbody][3] = i0 * ex.x * ex.y + i1 * ey.x * ey.y + i2 * ez.x * ez.y;
      S.bK[gbody][4] = i0 * ex.x * ex.z + i1 * ey.x * ey.z + i2 * ez.x * ez.z;
      S.bK[gbody][5] = i0 * ex.y * ex.z + i1 * ey.y * ey.z + i2 * ez.y * ez.z;
    }
  }
  if (!(has_robot && llane)) { gbeg = (int)cdesc; gend = (int)(((cdesc >> 12) & 0xfffu) + (cdesc >> 24)); }
  // robot section lanes (8 per link): the rs-th and (rs + 8)-th dof on the path base -> link rj (ND = none)
  int tjp = ND | (ND << 8);
  {
    const int rj = tid / 8, rs = tid % 8;
    if (rj < NL) {
      int tj0 = ND, tj1 = ND;
      uint32_t m = S.anc[rj];
      for (int t = 0; m; ++t) {
        const int j = __ffs(m) - 1;
        m &= m - 1;
        if (t == rs) tj0 = j;
        if (t == rs + 8) tj1 = j;
      }
      tjp = tj0 | (tj1 << 8);
    }
  }
  // mass splitting: the first iteration splits every body over ALL its contacts, iteration i > 0 over the contacts that were active in
  // iteration i - 1 (counted into the other set while this one is read); slots 0..71 bricks, NF the whole robot, NF + 1 the static world
  for (int i = tid; i < NF + 2; i += NT) {
    S.acount[0][i] = i < NF ? S.ecount[i] : (i == NF ? nrob : 0);
    S.acount[1][i] = 0;
  }
  if (tid == 0) S.rsync = 0;
  __syncthreads();   // every lane has read what it needs of v / w in the body table (warm-start gate above)
  // body table -> (u, w): u = v - w x x.  Link rows too (their twists are those of the drive phase until the robot section rewrites them)
  for (int i = tid; i < NBODY; i += NT) st3(S.bv[i], ld3(S.bv[i]) - cross(ld3(S.bw[i]), ld3(S.bp[i])));
  int qpar = 0;   // robot section: joint velocities are read from S.qd (0) / S.qdb (1) and written to the other
  __syncthreads();
  SSTAMP(17);

#pragma unroll
  for (int q = 0; q < CPT; ++q) {   // the sides' active-count slots into bits 16..31 of ab: a brick's own, NF = the whole robot, NF + 1 = the static world
    const int a = ab[q] & 0xff, b = (ab[q] >> 8) & 0xff;
    const int sa = a < NF ? a : (a != BODY_W ? NF : NF + 1), sb = b < NF ? b : (b != BODY_W ? NF : NF + 1);
    ab[q] = (ab[q] & 0xffff) | (sa << 16) | (sb << 24);
  }
  // fresh values for the loop: whatever the set-up phases above did to these registers, inside the loop they are plain registers again
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    SDX_OPAQUE(ab[q]); SDX_OPAQUE(vtgt[q]);
#pragma unroll
    for (int r = 0; r < 3; ++r) { SDX_OPAQUE(lam[q][r]); SDX_OPAQUE(wA[q][r]); SDX_OPAQUE(wB[q][r]); }
  }
  SDX_OPAQUE(gbeg); SDX_OPAQUE(gend); SDX_OPAQUE(tjp);
  const int it0 = (WARM && nold > 0) ? -1 : 0;
  for (int it = it0; it < (ABL(4) ? 1 : sc.solver_iters); ++it) {   // it = -1: only the gather of the warm-start impulses
#ifdef SDX_PHASE_CLOCK
    if (it == 1) dbg = nullptr;
#endif
    SSTAMP(18);
    const int cur = it & 1, nxt = cur ^ 1;
    LCLOCK(lc_ac0);
    // ---- [AC] lane = contact: relative velocity from the current body table, active flag -> counted for the NEXT iteration (integer
    // atomics), Jacobi update with the counts of the previous iteration; impulse P (on body A) to LDS, zero when inactive
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
      int abq = ab[q], ct = tid;
      SDX_OPAQUE(abq); SDX_OPAQUE(ct);
      const int c = ct + q * NT;
      f3 P = F3(0, 0, 0);
      if (c < nc && !ABL(2)) {
        const f3 n = F3(S.cn[0][c], S.cn[1][c], S.cn[2][c]);
        f3 t1, t2;
        tangents(n, &t1, &t2);
        if (WARM && it < 0) {                      // warm start: the whole initial impulse
          if (lam[q][0] > 0.0f) P = n * lam[q][0] + t1 * lam[q][1] + t2 * lam[q][2];
        } else {
          // (bits 16..31 of ab: the two sides' slots in the active-count sets, fixed for the solve - decoding them per iteration cost 28
          // VALU instructions per contact; the counts of the previous iteration are loaded with the geometry, not after the velocity test:
          // one LDS round trip less on the path of every active contact)
          const int a = abq & 0xff, b = (abq >> 8) & 0xff, sa = (abq >> 16) & 0xff, sb = (abq >> 24) & 0xff;
          const f3 p = F3(S.cp[0][c], S.cp[1][c], S.cp[2][c]);
          // (the four body rows in flight together: the plain expression loads side A, waits, then side B into the same registers)
          f3 ua = ld3v(S.bv[a]), wa = ld3v(S.bw[a]), ub = ld3v(S.bv[b]), wb = ld3v(S.bw[b]);
          SDX_PIN3x4(ua, wa, ub, wb);
#ifdef EXP_EXTRA_ROWS   // sensitivity probe: the four body rows a second time (results unused)
          { f3 xa = ld3v(S.bv[b]), xb = ld3v(S.bw[b]), xc = ld3v(S.bv[a]), xd = ld3v(S.bw[a]); SDX_PIN3x4(xa, xb, xc, xd); }
#endif
          const f3 vr = (ua + cross(wa, p)) - (ub + cross(wb, p));
          const float vn = dot(vr, n);
          const float lam0 = lam[q][0], lam1 = lam[q][1], lam2 = lam[q][2];
          if (lam0 > 0.0f || vn < vtgt[q]) {
            if (a != BODY_W) atomicAdd(&S.acount[nxt][sa], 1);
            if (b != BODY_W) atomicAdd(&S.acount[nxt][sb], 1);
#ifdef EXP_EXTRA_ATOMICS   // sensitivity probe: the same two atomics again on a dead array
            if (a != BODY_W) atomicAdd(&S.efill[sa], 1);
            if (b != BODY_W) atomicAdd(&S.efill[sb], 1);
#endif
            const int ca = S.acount[cur][sa], cb = S.acount[cur][sb];
            const float na = a != BODY_W ? (float)(ca > 1 ? ca : 1) : 0.0f;
            const float nb = b != BODY_W ? (float)(cb > 1 ? cb : 1) : 0.0f;
            const float w0 = na * wA[q][0] + nb * wB[q][0];
            const float w1 = na * wA[q][1] + nb * wB[q][1];
            const float w2 = na * wA[q][2] + nb * wB[q][2];
            const float ln = fmaxf(0.0f, lam0 - relax * (vn - vtgt[q]) * SDX_RCP(w0));
            const float lim = mu * ln;
            float l1 = lam1 - relax * dot(vr, t1) * SDX_RCP(w1);
            l1 = fminf(lim, fmaxf(-lim, l1));
            float l2 = lam2 - relax * dot(vr, t2) * SDX_RCP(w2);
            l2 = fminf(lim, fmaxf(-lim, l2));
            lam[q][0] = ln; lam[q][1] = l1; lam[q][2] = l2;
            P = n * (ln - lam0) + t1 * (l1 - lam1) + t2 * (l2 - lam2);
          }
        }
        S.P[0][c] = P.x; S.P[1][c] = P.y; S.P[2][c] = P.z;
      }
#ifdef SDX_AC_SCHED_BARRIER
      __builtin_amdgcn_sched_barrier(0);   // one contact after the other: interleaving the three costs the loop its registers
#endif
    }
#ifdef SDX_LANE_CLOCK
    { LCLOCK(lc_ac1); LMAX(55, lc_ac1 - lc_ac0); if (tid == NT - 1) LMAX(56, lc_ac1 - lc_ac0); }
#endif
    __syncthreads();
    SSTAMP(20);
    LCLOCK(lc_d0);
    // ---- [D] gather (LL / GL lanes per body): F = sum(+-P), M = sum(+-(p - x) x P) about the body's reference point x; each lane sums its
    // slice in ascending contact order, the partial sums are combined in a fixed order (deterministic); inactive contacts carry P = 0.
    // Bricks: w += Iw^-1 M, u += F / m - dw x x.  Links: the wrench (F, M about the link origin) is projected on the dofs of the link's
    // path right away (8 lanes, <= 2 dofs each) -> Qc for the robot section below.
    int td = tid;
    SDX_OPAQUE(td);   // lane coordinates re-derived per iteration instead of living in registers across the loop
    const bool d_link = has_robot && td < NL * LL;   // waves 0..2 of an env whose hand touches something: link lanes; every other lane is a brick lane
    const uint32_t cd = (uint32_t)gbeg;              // (brick lanes: the slice descriptor)
    const int d_body = d_link ? NF + td / LL : (int)(cd & 0x7fu), d_sub = d_link ? td % LL : 0;
    const int gstride = d_link ? LL : 1;
    const int ibeg = d_link ? gbeg : (int)((cd >> 12) & 0xfffu);
    {
      float acc[6] = {0, 0, 0, 0, 0, 0};
      const f3 x = ld3(S.bp[d_body]);
      // four entries per trip (the index loads, then the payloads, in flight together); not unrolled further: the decoded
      // addresses of a longer window would be kept in registers across the whole iteration loop
      // two entries at a time: index loads pinned in front of the 12 payload loads, those in front of the arithmetic (four dependent LDS
      // round trips per trip of four entries instead of eight; all four at once - SDX_D_BATCH - spills inside the loop)
#pragma unroll 1
      for (int i = ibeg; i < (ABL(1) ? ibeg : gend); i += GU * gstride) {
#pragma unroll
        for (int h2 = 0; h2 < GU; h2 += 2) {
          const int i0 = i + h2 * gstride, i1 = i0 + gstride;
          int e0 = S.ent[i0 < gend ? i0 : i], e1 = S.ent[i1 < gend ? i1 : i];
          SDX_PIN1(e0); SDX_PIN1(e1);
          const int c0 = e0 & 0x7fff, c1 = e1 & 0x7fff;
          f3 P0 = F3(S.P[0][c0], S.P[1][c0], S.P[2][c0]), C0 = F3(S.cp[0][c0], S.cp[1][c0], S.cp[2][c0]);
          f3 P1 = F3(S.P[0][c1], S.P[1][c1], S.P[2][c1]), C1 = F3(S.cp[0][c1], S.cp[1][c1], S.cp[2][c1]);
          SDX_PIN3x4(P0, C0, P1, C1);
          {
            const float sg = i0 < gend ? ((e0 & 0x8000) ? -1.0f : 1.0f) : 0.0f;
            const f3 P = P0 * sg;
            const f3 M = cross(C0 - x, P);
            acc[0] += P.x; acc[1] += P.y; acc[2] += P.z; acc[3] += M.x; acc[4] += M.y; acc[5] += M.z;
          }
          {
            const float sg = i1 < gend ? ((e1 & 0x8000) ? -1.0f : 1.0f) : 0.0f;
            const f3 P = P1 * sg;
            const f3 M = cross(C1 - x, P);
            acc[0] += P.x; acc[1] += P.y; acc[2] += P.z; acc[3] += M.x; acc[4] += M.y; acc[5] += M.z;
          }
        }
      }
#ifdef SDX_LANE_CLOCK
      { LCLOCK(lc_d1); LMAX(57, lc_d1 - lc_d0); LMAX(58, (gend - ibeg + gstride - 1) / gstride); }
#endif
      // the brick update's operands (inverse inertia, inverse mass, u, w: 13 numbers) are requested BEFORE the lane reductions and pinned
      // after them: their LDS round trip runs under the DPP adds instead of after them (every lane loads: the branch comes later)
      const int tb = d_link ? 0 : d_body;
      float K_[6], imb_ = S.bim[tb];
#pragma unroll
      for (int r = 0; r < 6; ++r) K_[r] = S.bK[tb][r];
      f3 u_ = ld3(S.bv[tb]), w_ = ld3(S.bw[tb]);
      if (d_link) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          acc[r] = sum4(acc[r]);
          acc[r] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[r]), 0x141, 0xF, 0xF, true));
        }
      } else {
        // the slices of a brick sit on consecutive lanes of one 16-lane row: inclusive segmented scan (row_shr 1, 2, 4, 8; lane `so` of the
        // brick adds the value `sh` lanes back while that lane still belongs to the brick), after which the brick's LAST lane holds the
        // sums - slices in ascending contact order, combined in the scan's fixed tree (deterministic)
        const int so = (int)((cd >> 7) & 15u);
#define SDX_SEG_STEP(sh)                                                                                                  \
        {                                                                                                                 \
          const bool take = so >= (sh);                                                                                   \
          _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                                                 \
            const float t = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[r]), 0x110 + (sh), 0xF, 0xF, true)); \
            acc[r] += take ? t : 0.0f;                                                                                    \
          }                                                                                                               \
        }
        SDX_SEG_STEP(1) SDX_SEG_STEP(2) SDX_SEG_STEP(4) SDX_SEG_STEP(8)
#undef SDX_SEG_STEP
      }
      if (!d_link) {
        if ((cd >> 11) & 1u) {
          SDX_PIN4(K_); SDX_PIN1(K_[4]); SDX_PIN1(K_[5]); SDX_PIN1(imb_); SDX_PIN3(u_); SDX_PIN3(w_);
          const float* K = K_;
          const f3 dw = F3(K[0] * acc[3] + K[3] * acc[4] + K[4] * acc[5], K[3] * acc[3] + K[1] * acc[4] + K[5] * acc[5],
                           K[4] * acc[3] + K[5] * acc[4] + K[2] * acc[5]);
          st3(S.bv[d_body], (u_ + F3(acc[0], acc[1], acc[2]) * imb_) - cross(dw, x));
          st3(S.bw[d_body], w_ + dw);
          S.acount[cur][d_body] = 0;   // read by [AC] before the barrier above; it is the set [AC] of the next iteration counts into
        }
      } else {
        const int k = d_body - NF;
        const int tj0 = tjp & 0xff, tj1 = tjp >> 8;
        const f3 F = F3(acc[0], acc[1], acc[2]), M = F3(acc[3], acc[4], acc[5]);
        // generalised impulse on the d_sub-th (and (d_sub + 8)-th) dof of the path: a_j . (M + (o_k - o_j) x F); la[ND + 1] = 0 for "none"
        S.Qc[k][d_sub] = dot(ld3(S.la[tj0 + 1]), M + cross(x - ld3(S.bp[NF + tj0 + 1]), F));
        if (d_sub < 4) S.Qc[k][d_sub + 8] = dot(ld3(S.la[tj1 + 1]), M + cross(x - ld3(S.bp[NF + tj1 + 1]), F));
        if (last_substep && d_sub == 0) { S.cf[k][0] += acc[0]; S.cf[k][1] += acc[1]; S.cf[k][2] += acc[2]; }   // net impulse on the link so far
      }
      if (td == NL * LL) { S.acount[cur][NF] = 0; }
    }
#ifdef SDX_LANE_CLOCK
    { LCLOCK(lc_d2); LMAX(59, lc_d2 - lc_d0); }
#endif
    if (has_robot && !ABL(8)) {
      // robot section, ONE stage on 8 lanes per link = waves 0..2, which are also the link gather's: they meet at their OWN barrier (an LDS
      // counter) while waves 3..7 are still gathering the bricks - the section touches nothing those read or write (Qc, qd / qdb, the link
      // rows of the body table) - and the pass closes with the one workgroup barrier below (round 6; a workgroup barrier in front of this
      // section made every pass of an env whose hand touches bricks brick gather + robot section long instead of the longer of the two).
      // Every group sums the generalised impulses of all 23 dofs from the Qc rows of the touched links below each dof (3 dofs per lane),
      // shares them inside the group lane to lane (ds_bpermute: no LDS space - the impulse rows are still being read by the brick
      // gather), then qd += Hinv dQ for the (<= 2) path dofs of this lane and the link's twist
      static_assert((NL * 8) % 64 == 0, "the robot section's lanes are whole waves");
      int tr = tid;
      SDX_OPAQUE(tr);
      if (tr < NL * 8) {
        // (requested before the waves meet: row `lane` of Hinv as six 16-byte loads and the joint velocity it belongs to)
        const int li = tr & 63, ri = li < ND ? li : 0;
        const float* qsrc = qpar ? S.qdb : S.qd;
        float* qdst = qpar ? S.qd : S.qdb;
        f4v arow[HP / 4];
#pragma unroll
        for (int k4 = 0; k4 < HP / 4; ++k4) arow[k4] = *reinterpret_cast<const f4v*>(&S.A[ri][4 * k4]);
        const float qi = qsrc[li < ND ? li : ND];   // (qd[ND] = 0)
        WAVES_BARRIER(&S.rsync, (NL * 8 / 64) * (it - it0 + 1), NL * 8);
        SSTAMP(21);
        const int rj = tr / 8, rs = tr % 8;
        float qa[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int j = rs + 8 * u;
          float acc = 0.0f;
          if (j < ND) {
            // the position of dof j in the path of a link below it = the number of dofs above j: the same for every such link
            const int slot = __popc(S.anc[j + 1] & ((1u << j) - 1u));
            uint32_t m = S.desc[j] & touched;
#pragma unroll
            for (int t = 0; t < 6; ++t) {          // (independent loads: in flight together)
              const int k = m ? __ffs(m) - 1 : 0;
              acc += m ? S.Qc[k][slot] : 0.0f;
              m &= m - 1;
            }
            while (m) {
              const int k = __ffs(m) - 1;
              m &= m - 1;
              acc += S.Qc[k][slot];
            }
          }
          qa[u] = acc;
        }
        // qd += Hinv dQ, ONCE per wave: lane i < 23 multiplies row i of Hinv with Q, whose entry j every 8-lane group holds in lane j % 8
        // (v_readlane from the wave's first group: no LDS).  Round 5 had every lane multiply the rows of its (<= 2) path dofs: 46 + 23 LDS
        // loads per lane in three dependent batches; now 6 + 2.
        float dq = 0.0f;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
          const f4v a4 = arow[j >> 2];
          const float aij = (j & 3) == 0 ? a4.x : ((j & 3) == 1 ? a4.y : ((j & 3) == 2 ? a4.z : a4.w));
          dq += aij * SDX_READLANE(qa[j >> 3], j & 7);
        }
        const float qn = li < ND ? qi + dq : 0.0f;
        if (tr < ND) qdst[tr] = qn;               // the other copy: groups still read the source copy of this pass
        const int tj0 = tjp & 0xff, tj1 = tjp >> 8;
        const float q0 = __shfl(qn, tj0, 64), q1 = __shfl(qn, tj1, 64);   // (tj = ND, "no dof": lane ND of the wave holds 0)
        const f3 pk = ld3(S.bp[NF + rj]);
        const f3 a0 = ld3(S.la[tj0 + 1]) * q0, a1 = ld3(S.la[tj1 + 1]) * q1;
        f3 w = a0 + a1;
        f3 v = cross(a0, pk - ld3(S.bp[NF + tj0 + 1])) + cross(a1, pk - ld3(S.bp[NF + tj1 + 1]));
        w.x = sum8(w.x); w.y = sum8(w.y); w.z = sum8(w.z);
        v.x = sum8(v.x); v.y = sum8(v.y); v.z = sum8(v.z);
        if (rs == 0 && rj > 0) { st3(S.bw[NF + rj], w); st3(S.bv[NF + rj], v - cross(w, pk)); }
      }
      qpar ^= 1;
    }
    __syncthreads();
    SSTAMP(22);
  }
  if (qpar && tid < ND) S.qd[tid] = S.qdb[tid];   // (lane tid is also the one that integrates dof tid)
  // body table back to (v, w) for the bricks (the link twists are rebuilt from qd by the next FK pass)
  for (int i = tid; i < NF; i += NT) st3(S.bv[i], ld3(S.bv[i]) + cross(ld3(S.bw[i]), ld3(S.bp[i])));
  // ---- the cache for the next solve (keys were written during the set-up)
  if (WARM) {
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
      const int c = tid + q * NT;
      if (c < nc) { wlam[c] = lam[q][0]; wlam[MAXC + c] = lam[q][1]; wlam[2 * MAXC + c] = lam[q][2]; }
    }
    if (tid == 0) *wcount = nc;
    // no agent-scope fence: the next reader is this workgroup's next substep, many workgroup barriers later on the same CU (the
    // vector L1 is shared by the waves of a workgroup), or the next launch
  }
}

// ---------------------------------------------------------------- outputs shared by step and refresh
template <int NT>
__device__ __forceinline__ void write_kinematics(const SdxConst* C, PhysLds& S, const SdxBuf& B, int e, int tid) {
  const sdx_scene_desc& sc = C->sc;
  float* rb_e = B.rb + (size_t)e * SDX_BODIES * 13;
  for (int i = tid; i < NL * 13; i += NT) {
    const int k = i / 13, c = i % 13;
    float v;
    if (c < 3) v = S.bp[NF + k][c];
    else if (c < 7) v = S.lq[k][c - 3];
    else if (c < 10) v = S.bv[NF + k][c - 7];
    else v = S.bw[NF + k][c - 10];
    rb_e[i] = v;
  }
  if (tid < 42) {  // geometric Jacobian of the hand-base body origin wrt the 7 arm dofs (GS:1601)
    const int r = tid / 7, j = tid % 7, ee = sc.hand_base_body;
    const f3 a = ld3(S.la[j + 1]);
    const f3 lin = cross(a, ld3(S.bp[NF + ee]) - ld3(S.bp[NF + j + 1]));
    const float v = r == 0 ? lin.x : r == 1 ? lin.y : r == 2 ? lin.z : r == 3 ? a.x : r == 4 ? a.y : a.z;
    B.jac[(size_t)e * 42 + tid] = v;
  }
}

// per-env constants into LDS (bricks, robot boxes, statics, ancestor masks)
template <int NT>
__device__ __forceinline__ void load_constants(const SdxConst* C, PhysLds& S, int e, int tid) {
  const sdx_scene_desc& sc = C->sc;
  const int segb = seg_actor(e) - SDX_ACTOR_BRICK0;
  if (tid == 0) {
    S.seg_brick = segb;
    st3(S.bp[BODY_W], F3(0, 0, 0)); st3(S.bv[BODY_W], F3(0, 0, 0)); st3(S.bw[BODY_W], F3(0, 0, 0));   // the static world
  }
  if (tid < NL) { S.anc[tid] = C->anc[tid]; S.lmass[tid] = sc.link_mass[tid]; }
  if (tid < ND) {
    uint32_t d = 0;
    for (int k = 1; k < NL; ++k) d |= ((C->anc[k] >> tid) & 1u) << k;
    S.desc[tid] = d;
  }
  if (tid <= NL) S.par[tid] = tid < NL ? (tid == 0 ? 0 : sc.parent[tid]) : NL;
  if (tid == 0) { S.qd[ND] = 0.0f; st3(S.la[NL], F3(0, 0, 0)); st3(S.lal[NL], F3(0, 0, 0)); }
  const int segt = sc.brick_type[segb];
  for (int i = tid; i < NF; i += NT) {
    const int t = sc.brick_type[i];
    S.btype[i] = (unsigned char)t;
    S.bim[i] = (i == segb ? 1.0f / sc.seg_mass_scale : 1.0f) / sc.brick_mass[t];
  }
  // collision compounds of the 8 brick types (centre-of-mass frame) and the hollow compound of this env's target brick
  if (tid < SDX_NBRICK_TYPES) {
    const int t = tid;
    const f3 com = ld3(sc.brick_com[t]), bc = ld3(sc.brick_center[t]) - com, bh = ld3(sc.brick_half[t]);
    S.tn[t] = sc.brick_nsub[t];
    st3(S.tbc[t], bc); st3(S.tbh[t], bh);
    S.trad[t] = sqrtf(dot(bc, bc)) + C->brick_radius[t];
    S.tii[t][0] = sc.brick_mass[t] / sc.brick_inertia[t][0]; S.tii[t][1] = sc.brick_mass[t] / sc.brick_inertia[t][1];
    S.tii[t][2] = sc.brick_mass[t] / sc.brick_inertia[t][2];
    for (int k = 0; k < SDX_MAX_SUB; ++k) { st3(S.tsc[t][k], ld3(sc.brick_sub_center[t][k]) - com); st3(S.tsh[t][k], ld3(sc.brick_sub_half[t][k])); }
  }
  if (tid >= 128 && tid < 128 + SDX_NSAMP * 3) (&S.samp[0][0])[tid - 128] = (&c_samp[0][0])[tid - 128];
  if (tid >= 256 && tid < 260) S.qid[tid - 256] = tid == 259 ? 1.0f : 0.0f;
  if (tid >= 64 && tid < 64 + SDX_MAX_SUB_HOLLOW) {
    const int k = tid - 64;
    st3(S.hsc[k], ld3(sc.hollow_sub_center[segt][k]) - ld3(sc.brick_com[segt])); st3(S.hsh[k], ld3(sc.hollow_sub_half[segt][k]));
    if (k == 0) S.hn = sc.seg_hollow ? sc.hollow_nsub[segt] : 0;
  }
  if (tid < SDX_MAX_RBOX) {
    const bool on = tid < sc.n_rbox;
    S.rbl[tid] = on ? sc.rbox_link[tid] : 0;
    st3(S.rh[tid], on ? ld3(sc.rbox_half[tid]) : F3(0, 0, 0));
    S.rrad[tid] = on ? C->rbox_radius[tid] : 0.0f;
  }
  if (tid < SDX_MAX_STATIC) {   // InsertSim's base plate is one of three by env % 3: a row of the static-body table each (sdx_scene_desc.static_var_*)
    const int row = tid == sc.static_var_slot ? sc.static_var_row[e % 3] : tid;
    st3(S.stc[tid], ld3(sc.static_center[row])); st3(S.sth[tid], ld3(sc.static_half[row]));
    S.ssf[tid] = sc.static_sub_first[row]; S.ssn[tid] = tid < sc.n_static ? sc.static_sub_n[row] : 1;
  }
}

// ---------------------------------------------------------------- the step kernel
template <int NT>
__global__ __launch_bounds__(NT, 2 * NT / 256) void k_physics(const SdxConst* __restrict__ C, SdxBuf B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PhysLds& S = *reinterpret_cast<PhysLds*>(smem);
  // launch order: B.order lists the envs by the cost of their previous step, longest first (k_order below)
  // (the loaded index is wave-uniform, but only readfirstlane tells the compiler: every address derived from it stays in scalar registers)
  const int e = B.order ? SDX_UNIFORM(B.order[blockIdx.x]) : (int)blockIdx.x, tid = threadIdx.x;
  const sdx_scene_desc& sc = C->sc;
  float* root_e = B.root + (size_t)e * SDX_ACTORS * 13;
  const float h = sc.dt / (float)sc.substeps;
#ifdef SDX_PHASE_CLOCK
  const int abl_bits = (int)B.dbg[63];
#else
  constexpr int abl_bits = 0;
#endif

  // this env's cycle count decides its place in the NEXT launch's order (k_order below): the clock is read by thread 0 at entry and exit
  const long long t_in = (long long)__builtin_readcyclecounter();
#ifdef SDX_PHASE_CLOCK
  if (threadIdx.x == 0) {
    B.dbg[64 + 2 * e] = t_in;                 // (entry, exit) of every env
    if (e == B.dbg_env) B.dbg[13] = t_in;     // kernel entry of the debug env
  }
#endif
  // ---- load per-env state (coalesced rows) into LDS.  The state rows are requested first and stored last: their HBM / L2 round trips run
  // beside the (dependent) loads of the scene constants instead of after them
  static_assert(NF <= NT, "one brick per lane");
  float r_q = 0.0f, r_qd = 0.0f, r_tg = 0.0f;
  if (tid < ND) {
    r_q = B.dof[((size_t)e * ND + tid) * 2];
    r_qd = B.dof[((size_t)e * ND + tid) * 2 + 1];
    r_tg = B.targets[(size_t)e * ND + tid];
  }
  const float* srow = root_e + (SDX_ACTOR_BRICK0 + (tid < NF ? tid : 0)) * 13;
  const f3 r_p = ld3(srow), r_v = ld3(srow + 7), r_w = ld3(srow + 10);
  const f4 r_o = ld4(srow + 3);
  const f3 r_com = ld3(sc.brick_com[sc.brick_type[tid < NF ? tid : 0]]);
  load_constants<NT>(C, S, e, tid);
  if (tid < ND) { S.q[tid] = r_q; S.qd[tid] = r_qd; S.tgt[tid] = r_tg; }
  if (tid < NF) {
    const f4 q = qnormalize(r_o);
    st4(S.bq[tid], q);
    st3(S.bp[tid], r_p + qrot(q, r_com));
    st3(S.bv[tid], r_v);
    st3(S.bw[tid], r_w);
  }
  for (int i = tid; i < NT; i += NT) S_BMASK(S)[i] = 0u;
  if (tid == 0) S.fkflag = 0;
  __syncthreads();

  const int nsub = sc.substeps;
  for (int sub = 0; sub < nsub; ++sub) {
    // per-substep opaque copies of the constants pointer and the thread index: without them every loop-invariant load and address
    // of the substep body (scene constants, per-lane offsets) is hoisted in front of the loop and parked in scratch - 48 spills per
    // lane, 100 MB of scratch writes per launch at N = 1024 (profiles/r2_kphysics_pmc_write.csv before this change)
    const SdxConst* Cs = C;
    SDX_OPAQUE_S(Cs);
    int tl = threadIdx.x;
    SDX_OPAQUE(tl);
    const sdx_scene_desc& scl = Cs->sc;
    PSTAMP(0);
    if (sub == 0) {   // M(q) is evaluated once per step (frozen over the substeps)
      // wave 0: forward kinematics + inertias; waves 1..7 meanwhile: the broadphase tests of the brick / brick and brick / static
      // candidates (part 0), which depend on nothing the robot does in this substep
      if (tl < 64) fk_wave0(Cs, S, tl, true, true, e == B.dbg_env ? B.dbg : nullptr);
      else if (!ABL(32)) broad_mask_part<NT>(Cs, S, tl, 0);
      __syncthreads();
      PSTAMP(1);
      if (!ABL(512)) mass_matrix<NT>(Cs, S, tl, h, e == B.dbg_env ? B.dbg : nullptr);   // (part 1 of the mask beside its wave-0 section)
      else { if (tl >= 64) broad_mask_part<NT>(Cs, S, tl, 1); __syncthreads(); }
      PSTAMP(2);
    }
    // A + C on wave 0 (FK, implicit PD drive (P1), velocity-product bias torques, twists); the other waves: gravity on the free bricks
    if (ABL(256) && sub != 0) {
      if (tl >= 64) { broad_mask_part<NT>(Cs, S, tl, 0); broad_mask_part<NT>(Cs, S, tl, 1); }
    } else if (tl < 64) {
      if (sub != 0) { fk_wave0(Cs, S, tl, false, true); WAVE_SIGNAL(&S.fkflag, sub); }   // (the robot boxes are placed: part 1 of the mask may start)
      if (tl < ND) {
        const float t = scl.kp[tl] * (S.tgt[tl] - S.q[tl]) - (scl.kd[tl] + h * scl.kp[tl]) * S.qd[tl];
        // velocity-product bias torque of dof tl: inertial wrenches of the links below it, projected on its axis
        float tc = 0.0f;
        const f3 aj = ld3(S.la[tl + 1]), oj = ld3(S.bp[NF + tl + 1]);
        {   // links below dof tl = bits of desc[tl]; four links' operands in flight at a time, the same terms added in the same (ascending) order
          const uint32_t below = S.desc[tl];
          constexpr int FB = 4;
#pragma unroll
          for (int k0 = 1; k0 < NL; k0 += FB) {
            f3 lc_[FB], lF_[FB], lN_[FB];
#pragma unroll
            for (int u = 0; u < FB; ++u) if (k0 + u < NL) { lc_[u] = ld3(S.lc[k0 + u]); lF_[u] = ld3(S.lF[k0 + u]); lN_[u] = ld3(S.lN[k0 + u]); }
#pragma unroll
            for (int u = 0; u < FB; ++u) if (k0 + u < NL) { SDX_PIN3(lc_[u]); SDX_PIN3(lF_[u]); SDX_PIN3(lN_[u]); }
#pragma unroll
            for (int u = 0; u < FB; ++u) if (k0 + u < NL) {
              const float term = dot(aj, cross(lc_[u] - oj, lF_[u]) + lN_[u]);
              if ((below >> (k0 + u)) & 1u) tc += term;
            }
          }
        }
        S.tau[tl] = fminf(scl.effort[tl], fmaxf(-scl.effort[tl], t)) - tc;   // the effort limit applies to the drive only
      }
      WAVE_SYNC();
      if (tl < ND) {
        float s = 0.0f;
        for (int j = 0; j < ND; ++j) s += S.A[tl][j] * S.tau[j];
        S.qd[tl] += h * s;
      }
      WAVE_SYNC();
      twists_wave0(S, tl);
    } else {
      for (int i = tl - 64; i < NF; i += NT - 64) {
        S.bv[i][0] += scl.gravity[0] * h; S.bv[i][1] += scl.gravity[1] * h; S.bv[i][2] += scl.gravity[2] * h;
      }
      // later substeps: both parts of the broadphase mask beside wave 0's FK + drive (the first substep had them beside FK and the factorisation)
      if (sub != 0 && !ABL(32)) {
        broad_mask_part<NT>(Cs, S, tl, 0);
        WAVES_WAIT(&S.fkflag, sub);
        broad_mask_part<NT>(Cs, S, tl, 1);
      }
    }
    __syncthreads();
    PSTAMP(3);
    collide<NT>(Cs, S, tl, (sub == 0 && e == B.dbg_env) ? B.dbg : nullptr, abl_bits);
    if (tl == 0 && B.cstats) {   // capacity statistics of this substep (integer atomics: order-independent)
      atomicMax(&B.cstats[0], S.nc + S.overflow);
      if (S.overflow) atomicAdd(&B.cstats[1], 1);
      if (S.rebuilt & 1) atomicAdd(&B.cstats[2], 1);
      if (S.rebuilt & 2) atomicAdd(&B.cstats[3], 1);
    }
    PSTAMP(4);
    if (scl.warm_start > 0.0f)
      solve<NT, true>(Cs, S, tl, h, sub == nsub - 1, (sub == 0 && e == B.dbg_env) ? B.dbg : nullptr, B.wcount + e, B.wkey + (size_t)e * MAXC, B.wlam + (size_t)e * 3 * MAXC, abl_bits);
    else
      solve<NT, false>(Cs, S, tl, h, sub == nsub - 1, (sub == 0 && e == B.dbg_env) ? B.dbg : nullptr, nullptr, nullptr, nullptr, abl_bits);
    PSTAMP(5);
    // F: integrate
    if (tl < ND) {
      float v = S.qd[tl] * (1.0f - h * scl.robot_angular_damping);   // GS:546
      v = fminf(scl.vel_limit[tl], fmaxf(-scl.vel_limit[tl], v));
      float qn = S.q[tl] + h * v;
      if (qn < scl.lower[tl]) { qn = scl.lower[tl]; v = fmaxf(v, 0.0f); }
      if (qn > scl.upper[tl]) { qn = scl.upper[tl]; v = fminf(v, 0.0f); }
      S.q[tl] = qn;
      S.qd[tl] = v;
    }
    for (int i = tl; i < NF; i += NT) {
      const f3 v = ld3(S.bv[i]), w = ld3(S.bw[i]);
      st3(S.bp[i], ld3(S.bp[i]) + v * h);
      const f4 q = ld4(S.bq[i]);
      f4 wq; wq.x = w.x; wq.y = w.y; wq.z = w.z; wq.w = 0.0f;
      const f4 dq = qmul(wq, q);
      f4 nq; nq.x = q.x + 0.5f * h * dq.x; nq.y = q.y + 0.5f * h * dq.y; nq.z = q.z + 0.5f * h * dq.z;
      nq.w = q.w + 0.5f * h * dq.w;
      st4(S.bq[i], qnormalize(nq));
    }
    for (int i = tl; i < NT; i += NT) S_BMASK(S)[i] = 0u;   // (the contact rows are dead between the solve and the next substep's collide)
    __syncthreads();
    PSTAMP(6);
  }

#ifdef SDX_PHASE_CLOCK
  if (threadIdx.x == 0 && e == B.dbg_env) B.dbg[15] = (long long)__builtin_readcyclecounter();   // end of the last substep
#endif
  // ---- outputs (refresh_* of GS:1091-1095): wave 0 refreshes the kinematics at the integrated joint positions; the brick rows, the contact
  // forces and the step's statistics depend on none of that and are written by waves 1..7 meanwhile
  float* rb_e = B.rb + (size_t)e * SDX_BODIES * 13;
  if (tid < 64) {
    fk_wave0(C, S, tid, false, false);
  } else {
    for (int i = tid - 64; i < NF * 13; i += NT - 64) {
      const int k = i / 13, c = i % 13;
      float v;
      if (c < 3) {
        const f3 o = ld3(S.bp[k]) - qrot(ld4(S.bq[k]), ld3(sc.brick_com[sc.brick_type[k]]));
        v = c == 0 ? o.x : c == 1 ? o.y : o.z;
      } else if (c < 7) v = S.bq[k][c - 3];
      else if (c < 10) v = S.bv[k][c - 7];
      else v = S.bw[k][c - 10];
      root_e[SDX_ACTOR_BRICK0 * 13 + i] = v;
      rb_e[SDX_BODY_BRICK0 * 13 + i] = v;
    }
    for (int i = tid - 64; i < NL * 3; i += NT - 64) B.contact[(size_t)e * SDX_BODIES * 3 + i] = (&S.cf[0][0])[i] * (1.0f / h);   // net impulse of the last substep / h
  }
  __syncthreads();
  if (tid < ND) {
    B.dof[((size_t)e * ND + tid) * 2] = S.q[tid];
    B.dof[((size_t)e * ND + tid) * 2 + 1] = S.qd[tid];
  }
  write_kinematics<NT>(C, S, B, e, tid);
  if (tid == 0) {
    B.ncontacts[e] = S.nc + S.overflow;
    // what this step cost: the next launch's order.  Round 6: the measured cycles of the env (units of 4096) instead of (hand touches
    // something, contact count), whose correlation with the cycles was 0.5 - a fifth of the launch was slots waiting for the slowest pair
    if (B.cost) B.cost[e] = (int)(((long long)__builtin_readcyclecounter() - t_in) >> 12);
  }
#ifdef SDX_PHASE_CLOCK
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long t_out = (long long)__builtin_readcyclecounter();
    B.dbg[64 + 2 * e + 1] = t_out;
    if (e == B.dbg_env) B.dbg[14] = t_out;    // all outputs issued
  }
#endif
}

// ---- launch order of the next step: the envs by the cycles their last step took (k_physics measures them), longest first, in 256 buckets of
// 4096 cycles.  With 2 workgroups per CU and N = 1024 every CU slot runs two envs back to back; in env order two slow ones can meet on one
// slot (makespan 2 x slow), longest-first pairs the slowest with the fastest.  (Rounds 3-5 sorted by "hand touches something" and the contact
// count; round 6 measured a correlation of 0.5 between that key and the cycles: env 589 k cycles on average, 505 k / 656 k at the 10th / 90th
// percentile, 854 k the slowest - and the launch took 1.42 M cycles where the slots' average load was 1.18 M.)
// The order inside a bucket is whatever the atomics give, and the buckets depend on clocks: neither can change any result (envs are independent).
__global__ __launch_bounds__(1024) void k_order(SdxBuf B) {
  __shared__ int hist[256], base[256];
  const int tid = threadIdx.x;
  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  for (int e = tid; e < B.N; e += 1024) {
    const int c = B.cost[e];
    atomicAdd(&hist[c < 0 ? 0 : (c > 255 ? 255 : c)], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 255; k >= 0; --k) { base[k] = run; run += hist[k]; }
  }
  __syncthreads();
  for (int e = tid; e < B.N; e += 1024) {
    const int c = B.cost[e];
    B.order[atomicAdd(&base[c < 0 ? 0 : (c > 255 ? 255 : c)], 1)] = e;
  }
}

// kinematics only: rigid-body states of the 24 links + end-effector Jacobian from SDX_T_DOF (one wave per env)
__global__ __launch_bounds__(64) void k_kinematics(const SdxConst* __restrict__ C, SdxBuf B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PhysLds& S = *reinterpret_cast<PhysLds*>(smem);
  const int e = blockIdx.x, tid = threadIdx.x;
  if (tid < ND) {
    S.q[tid] = B.dof[((size_t)e * ND + tid) * 2];
    S.qd[tid] = B.dof[((size_t)e * ND + tid) * 2 + 1];
  }
  if (tid < NL) S.anc[tid] = C->anc[tid];
  if (tid < SDX_MAX_RBOX) S.rbl[tid] = tid < C->sc.n_rbox ? C->sc.rbox_link[tid] : 0;
  if (tid <= NL) S.par[tid] = tid < NL ? (tid == 0 ? 0 : C->sc.parent[tid]) : NL;
  if (tid == 0) {
    S.qd[ND] = 0.0f; st3(S.la[NL], F3(0, 0, 0)); st3(S.lal[NL], F3(0, 0, 0));
    st3(S.bp[BODY_W], F3(0, 0, 0)); st3(S.bv[BODY_W], F3(0, 0, 0)); st3(S.bw[BODY_W], F3(0, 0, 0));
  }
  WAVE_SYNC();
  fk_wave0(C, S, tid, false, false);
  write_kinematics<64>(C, S, B, e, tid);
  if (B.jac_full) {   // acquire_jacobian_tensor(sim, "hand") (GS:241): [23 links (fixed base excluded), 6, 23 dofs]
    float* J = B.jac_full + (size_t)e * (NL - 1) * 6 * ND;
    for (int i = tid; i < (NL - 1) * 6 * ND; i += 64) {
      const int k = i / (6 * ND) + 1, r = (i / ND) % 6, j = i % ND;
      float v = 0.0f;
      if ((S.anc[k] >> j) & 1u) {
        const f3 a = ld3(S.la[j + 1]);
        const f3 lin = cross(a, ld3(S.bp[NF + k]) - ld3(S.bp[NF + j + 1]));
        v = r == 0 ? lin.x : r == 1 ? lin.y : r == 2 ? lin.z : r == 3 ? a.x : r == 4 ? a.y : a.z;
      }
      J[i] = v;
    }
  }
}

extern "C" size_t sdxk_physics_lds_bytes() { return sizeof(PhysLds); }
// threads per env: 512 (8 waves, <= 128 VGPRs), two envs per CU
static void physics_init() {
  static bool done = false;
  if (!done) {
    done = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_physics<512>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PhysLds));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_kinematics), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PhysLds));
  }
}
extern "C" int sdxk_physics_threads() { return 512; }
extern "C" void sdxk_physics(const SdxConst* C, const SdxBuf* B, hipStream_t st) {
  physics_init();
  hipLaunchKernelGGL(k_physics<512>, dim3(B->N), dim3(512), sizeof(PhysLds), st, C, *B);
  if (B->order && B->cost) hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, st, *B);
}
extern "C" void sdxk_kinematics(const SdxConst* C, const SdxBuf* B, hipStream_t st) {
  physics_init();
  hipLaunchKernelGGL(k_kinematics, dim3(B->N), dim3(64), sizeof(PhysLds), st, C, *B);
}
