// sdx_physics.hip — the per-env physics step (kernels K3/K4/K5), one wavefront
// per env, state tiled in LDS, contact rows in an L2/MALL-resident SoA scratch (coalesced, lane = contact).
//
// Replaces gym.simulate()/fetch_results() (BT:138-144) and the refresh_* calls (GS:1091-1095) of the
// reference, whose arithmetic lives in the closed Isaac Gym / PhysX binary.  The step is OUR definition
// ("SDX-1"), restated independently in plain C in oracle/physics_oracle.c:
//   A FK  B joint-space inertia + implicit-PD matrix, Cholesky inverse  C implicit PD drive, gravity
//   D sampled-SDF box/box contacts (<= 4 per pair)  E active-set mass-split Jacobi on accumulated impulses
//   F semi-implicit Euler;  then outputs: rigid-body states, end-effector Jacobian, net arm contact forces.
//
// Workgroup structure: one workgroup of 4 wavefronts (256 lanes) per env; "lane = link / dof / brick / matrix entry /
// candidate pair / contact" in turn; every phase is a lane-strided loop separated by workgroup barriers.  Each lane
// OWNS up to CPT = 5 contact rows (contact c belongs to lane c % 256): their geometry, effective masses and
// accumulated impulses stay in VGPRs across all solver iterations, so the iteration loop touches only LDS.
#include "sdx_common.h"

#define NT 512
#define NWAVE (NT / 64)
#define CPT ((SDX_MAXC + NT - 1) / NT)   // contact rows owned by one lane
#define NL SDX_NLINK
#define ND SDX_NDOF
#define NF SDX_NFREE
#define HP 24  // padded row stride of the 23x23 matrices in LDS

__constant__ float c_samp[SDX_NSAMP][3] = {
    {-1, -1, -1}, {1, -1, -1}, {-1, 1, -1}, {1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {-1, 1, 1}, {1, 1, 1},
    {0, -1, -1}, {0, 1, -1}, {0, -1, 1}, {0, 1, 1}, {-1, 0, -1}, {1, 0, -1}, {-1, 0, 1}, {1, 0, 1},
    {-1, -1, 0}, {1, -1, 0}, {-1, 1, 0}, {1, 1, 0},
    {-0.5f, -1, -1}, {0.5f, -1, -1}, {-0.5f, 1, -1}, {0.5f, 1, -1}, {-0.5f, -1, 1}, {0.5f, -1, 1}, {-0.5f, 1, 1}, {0.5f, 1, 1}};

struct PhysLds {
  float q[ND], qd[ND], tgt[ND], qds[ND], Q[ND], tau[ND];
  float lq[NL][4], lp[NL][3], la[NL][3], lc[NL][3], lv[NL][3], lw[NL][3], lI[NL][6];
  float lal[NL][3], lao[NL][3], lF[NL][3], lN[NL][3];   // velocity-product terms: angular / origin accelerations at zero qdd, inertial wrenches
  float A[ND][HP];   // H -> L -> Hinv
  float bp[NF][3], bq[NF][4], bv[NF][3], bw[NF][3], dv[NF][3], dw[NF][3];
  int bcount[NF];
  int rcount, nc, np, overflow;
  float rc[SDX_MAX_RBOX][3], rq[SDX_MAX_RBOX][4];
  uint32_t pairs[SDX_MAXP];
  int wsum[NWAVE];       // per-wave totals for block-level scans
  float cf[NL][3];
  int nrobot;            // contact rows touching the robot in this substep
  int seg_brick;         // this env's target brick (its mass and inertia carry sc.seg_mass_scale)
  float sincos[NL][2];   // sin/cos of half the joint angle, computed for all joints at once
  // contact staging [8][SDX_MAXC] written by the narrowphase (ab, p3, n3, sep); after the rows are in registers the
  // same area is reused by the solver for the per-contact impulse P (3) and moment p x P (3)
  float stage[8][SDX_MAXC];
  unsigned char act[SDX_MAXC];
  unsigned short ent[2 * SDX_MAXC];   // CSR entries: contact index | side << 15, grouped by brick, ascending contact index
  int eoff[NF + 1];
  int efill[NF];
  // robot-side contact sides: contact index | side << 15, ascending (contact, side) order, and the link each one touches
  unsigned short rent[SDX_MAXC];
  unsigned char rlink[SDX_MAXC];
  int rfill;
};

// Scratch that lives inside the contact staging area while those rows are dead (keeps PhysLds under 80 KiB, the LDS half of what two
// workgroups per CU would need; the register half does not fit yet: at the 128-VGPR budget of 4 waves/SIMD the solver's contact
// rows spill 640 B/lane and a launch gets 5 % SLOWER (2.15 vs 2.05 ms at N = 1024), so the kernel stays at 256 VGPRs, 1 workgroup/CU):
//   L^-1 of the mass-matrix inversion (substep 0, before the narrowphase writes the staging rows)            -> stage[0]
//   unsorted CSR fill order of the brick sides / robot sides (rank pass; rows 6 (n.z) and 7 (sep) are in registers by then and
//   the solver reuses only rows 0..5)                                                                        -> stage[6], stage[7]
#define S_T(S) (reinterpret_cast<float (*)[HP]>(&(S).stage[0][0]))
#define S_ENT2(S) (reinterpret_cast<unsigned short*>(&(S).stage[6][0]))
#define S_RENT2(S) (reinterpret_cast<unsigned short*>(&(S).stage[7][0]))
static_assert(ND * HP <= SDX_MAXC, "L^-1 must fit one staging row");

struct Box { f3 c; f4 q; f3 h; };
#define PSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && sub == 0) B.dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define SSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && dbg) dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)

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

__device__ __forceinline__ void tangents(f3 n, f3* t1, f3* t2) {
  f3 a = fabsf(n.x) < 0.57735f ? F3(1, 0, 0) : F3(0, 1, 0);
  f3 t = cross(n, a);
  t = t * (1.0f / sqrtf(dot(t, t)));
  *t1 = t;
  *t2 = cross(n, t);
}

// static box s of env blockIdx.x: InsertSim's base plate has one of three heights by env % 3 (sdx_scene_desc.static_var_*)
__device__ __forceinline__ void static_box(const sdx_scene_desc& sc, int s, f3* c, f3* h) {
  *c = ld3(sc.static_center[s]);
  *h = ld3(sc.static_half[s]);
  if (s == sc.static_var_slot) {
    const int k = (int)(blockIdx.x % 3u);
    c->z = sc.static_var_center_z[k];
    h->z = sc.static_var_half_z[k];
  }
}
// box id: 0..71 brick, 72..103 robot box, 128.. static
__device__ __forceinline__ Box load_box(const SdxConst* C, const PhysLds& S, int id) {
  const sdx_scene_desc& sc = C->sc;
  Box b;
  if (id < NF) {
    b.c = ld3(S.bp[id]); b.q = ld4(S.bq[id]); b.h = ld3(sc.brick_half[sc.brick_type[id]]);
  } else if (id < 128) {
    const int r = id - NF;
    b.c = ld3(S.rc[r]); b.q = ld4(S.rq[r]); b.h = ld3(sc.rbox_half[r]);
  } else {
    const int s = id - 128;
    static_box(sc, s, &b.c, &b.h); b.q.x = 0; b.q.y = 0; b.q.z = 0; b.q.w = 1;
  }
  return b;
}
__device__ __forceinline__ int box_body(const SdxConst* C, int id) {
  if (id < NF) return id;
  if (id < 128) return NF + C->sc.rbox_link[id - NF];
  return SDX_BODY_STATIC;
}

// samples of A against the SDF of B; returns count (<=4), indices packed 8 bits each
__device__ __forceinline__ int sample_dir(const Box& A, const Box& B, float off, uint32_t* packed) {
  const f4 qbi = qconj(B.q);
  const f3 t = qrot(qbi, A.c - B.c);
  const f4 qrel = qmul(qbi, A.q);
  int cnt = 0;
  uint32_t pk = 0;
  for (int s = 0; s < SDX_NSAMP && cnt < 4; ++s) {
    const f3 l = F3(A.h.x * c_samp[s][0], A.h.y * c_samp[s][1], A.h.z * c_samp[s][2]);
    const f3 pb = t + qrot(qrel, l);
    if (box_sdf_val(pb, B.h) < off) { pk |= (uint32_t)s << (8 * cnt); ++cnt; }
  }
  *packed = pk;
  return cnt;
}

__device__ __forceinline__ void cwrite(float (*cs)[SDX_MAXC], int c, int a, int b, f3 p, f3 n, float sep) {
  cs[0][c] = __int_as_float(a | (b << 8));
  cs[1][c] = p.x; cs[2][c] = p.y; cs[3][c] = p.z;
  cs[4][c] = n.x; cs[5][c] = n.y; cs[6][c] = n.z;
  cs[7][c] = sep;
}

__device__ __forceinline__ void emit_dir(const Box& A, const Box& B, int ida, int idb, uint32_t packed, int k,
                                         float (*cs)[SDX_MAXC], int base) {
  const f4 qbi = qconj(B.q);
  const f3 t = qrot(qbi, A.c - B.c);
  const f4 qrel = qmul(qbi, A.q);
  for (int i = 0; i < k; ++i) {
    const int c = base + i;
    if (c >= SDX_MAXC) break;
    const int s = (packed >> (8 * i)) & 0xff;
    const f3 l = F3(A.h.x * c_samp[s][0], A.h.y * c_samp[s][1], A.h.z * c_samp[s][2]);
    const f3 pb = t + qrot(qrel, l);
    f3 g;
    const float sd = box_sdf(pb, B.h, &g);
    const f3 n = qrot(B.q, g);
    const f3 pw = B.c + qrot(B.q, pb);
    cwrite(cs, c, ida, idb, pw - n * (0.5f * sd), n, sd);
  }
}

__device__ __forceinline__ f3 point_vel(const PhysLds& S, int id, f3 p) {
  if (id == SDX_BODY_STATIC) return F3(0, 0, 0);
  if (id < NF) return ld3(S.bv[id]) + cross(ld3(S.bw[id]), p - ld3(S.bp[id]));
  const int k = id - NF;
  return ld3(S.lv[k]) + cross(ld3(S.lw[k]), p - ld3(S.lp[k]));
}

__device__ __forceinline__ float brick_w(const SdxConst* C, const PhysLds& S, int i, f3 p, f3 d) {
  const sdx_scene_desc& sc = C->sc;
  const int t = sc.brick_type[i];
  const f3 rxd = cross(p - ld3(S.bp[i]), d);
  const f3 l = qrot(qconj(ld4(S.bq[i])), rxd);
  const float* I = sc.brick_inertia[t];
  const float w = 1.0f / sc.brick_mass[t] + l.x * l.x / I[0] + l.y * l.y / I[1] + l.z * l.z / I[2];
  return i == S.seg_brick ? w / sc.seg_mass_scale : w;
}

// ---------------------------------------------------------------- A: FK (level-parallel over the tree)
// world inertia times vector: R I R^T x with I = (xx yy zz xy xz yz) in the link frame
__device__ __forceinline__ f3 inertia_mul(f4 q, const float* I, f3 x) {
  const f3 l = qrot(qconj(q), x);
  return qrot(q, F3(I[0] * l.x + I[3] * l.y + I[4] * l.z, I[3] * l.x + I[1] * l.y + I[5] * l.z, I[4] * l.x + I[5] * l.y + I[2] * l.z));
}
__device__ void fk(const SdxConst* C, PhysLds& S, int tid) {
  const sdx_scene_desc& sc = C->sc;
  if (tid == 0) {
    st4(S.lq[0], ld4(sc.base_quat));
    st3(S.lp[0], ld3(sc.base_pos));
    st3(S.la[0], F3(0, 0, 1));
    st3(S.lv[0], F3(0, 0, 0));
    st3(S.lw[0], F3(0, 0, 0));
    st3(S.lal[0], F3(0, 0, 0)); st3(S.lao[0], F3(0, 0, 0)); st3(S.lF[0], F3(0, 0, 0)); st3(S.lN[0], F3(0, 0, 0));
    st3(S.lc[0], ld3(sc.base_pos) + qrot(ld4(sc.base_quat), ld3(sc.link_com[0])));
  }
  if (tid > 0 && tid < NL) sincosf(0.5f * S.q[tid - 1], &S.sincos[tid][0], &S.sincos[tid][1]);
  __syncthreads();
  for (int d = 1; d <= C->max_depth; ++d) {
    if (tid > 0 && tid < NL && C->depth[tid] == d) {
      const int k = tid, p = sc.parent[k];
      const f4 qp = ld4(S.lq[p]);
      const f3 pp = ld3(S.lp[p]);
      const f4 qj = qmul(qp, ld4(sc.joint_quat[k]));
      const f3 ax = ld3(sc.joint_axis[k]);
      f4 qa; qa.x = ax.x * S.sincos[k][0]; qa.y = ax.y * S.sincos[k][0]; qa.z = ax.z * S.sincos[k][0]; qa.w = S.sincos[k][1];
      const f4 qk = qnormalize(qmul(qj, qa));
      const f3 pk = pp + qrot(qp, ld3(sc.joint_pos[k]));
      const f3 ak = qrot(qj, ax);
      const f3 wp = ld3(S.lw[p]);
      st4(S.lq[k], qk);
      st3(S.lp[k], pk);
      st3(S.la[k], ak);
      st3(S.lc[k], pk + qrot(qk, ld3(sc.link_com[k])));
      const f3 wk = wp + ak * S.qd[k - 1];
      st3(S.lw[k], wk);
      st3(S.lv[k], ld3(S.lv[p]) + cross(wp, pk - pp));
      // velocity-product terms (recursive Newton-Euler at zero joint acceleration, fixed base, no gravity on the robot)
      const f3 alp = ld3(S.lal[p]), r = pk - pp;
      const f3 alk = alp + cross(wp, ak * S.qd[k - 1]);
      const f3 aok = ld3(S.lao[p]) + cross(alp, r) + cross(wp, cross(wp, r));
      st3(S.lal[k], alk);
      st3(S.lao[k], aok);
      const f3 dk = qrot(qk, ld3(sc.link_com[k]));
      const f3 acom = aok + cross(alk, dk) + cross(wk, cross(wk, dk));
      st3(S.lF[k], acom * sc.link_mass[k]);
      st3(S.lN[k], inertia_mul(qk, sc.link_inertia[k], alk) + cross(wk, inertia_mul(qk, sc.link_inertia[k], wk)));
    }
    __syncthreads();
  }
  if (tid < sc.n_rbox) {
    const int k = sc.rbox_link[tid];
    const f4 qk = ld4(S.lq[k]);
    st3(S.rc[tid], ld3(S.lp[k]) + qrot(qk, ld3(sc.rbox_center[tid])));
    st4(S.rq[tid], qmul(qk, ld4(sc.rbox_quat[tid])));
  }
  __syncthreads();
}

// link twists from qd: w_k = sum_j a_j qd_j, v_k = sum_j (a_j qd_j) x (p_k - p_j) over the dofs j on the path
__device__ void twists(const SdxConst* C, PhysLds& S, int tid) {
  if (tid > 0 && tid < NL) {
    f3 w = F3(0, 0, 0), v = F3(0, 0, 0);
    const f3 pk = ld3(S.lp[tid]);
    uint32_t m = C->anc[tid];
    while (m) {
      const int j = __ffs(m) - 1;
      m &= m - 1;
      const f3 aj = ld3(S.la[j + 1]) * S.qd[j];
      w = w + aj;
      v = v + cross(aj, pk - ld3(S.lp[j + 1]));
    }
    st3(S.lw[tid], w);
    st3(S.lv[tid], v);
  }
  __syncthreads();
}

// ---------------------------------------------------------------- B: H = M + implicit PD terms, Hinv
__device__ void mass_matrix(const SdxConst* C, PhysLds& S, int tid, float h) {
  const sdx_scene_desc& sc = C->sc;
  if (tid < NL) {  // world inertia R I R^T of each link (xx yy zz xy xz yz)
    const f4 q = ld4(S.lq[tid]);
    const float* I = sc.link_inertia[tid];
    const f3 ex = qrot(q, F3(1, 0, 0)), ey = qrot(q, F3(0, 1, 0)), ez = qrot(q, F3(0, 0, 1));  // columns of R
    // Iw = sum_ab I_ab e_a e_b^T
    const f3 c0 = ex * I[0] + ey * I[3] + ez * I[4];
    const f3 c1 = ex * I[3] + ey * I[1] + ez * I[5];
    const f3 c2 = ex * I[4] + ey * I[5] + ez * I[2];
    // Iw = [c0 c1 c2] R^T  -> Iw_rc = c0_r ex_c + c1_r ey_c + c2_r ez_c
    S.lI[tid][0] = c0.x * ex.x + c1.x * ey.x + c2.x * ez.x;
    S.lI[tid][1] = c0.y * ex.y + c1.y * ey.y + c2.y * ez.y;
    S.lI[tid][2] = c0.z * ex.z + c1.z * ey.z + c2.z * ez.z;
    S.lI[tid][3] = c0.x * ex.y + c1.x * ey.y + c2.x * ez.y;
    S.lI[tid][4] = c0.x * ex.z + c1.x * ey.z + c2.x * ez.z;
    S.lI[tid][5] = c0.y * ex.z + c1.y * ey.z + c2.y * ez.z;
  }
  __syncthreads();
  for (int idx = tid; idx < ND * (ND + 1) / 2; idx += NT) {
    int i = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
    while ((i + 1) * (i + 2) / 2 <= idx) ++i;
    while (i * (i + 1) / 2 > idx) --i;
    const int j = idx - i * (i + 1) / 2;
    float s = 0.0f;
    if ((C->anc[i + 1] >> j) & 1u) {
      const f3 ai = ld3(S.la[i + 1]), aj = ld3(S.la[j + 1]);
      const f3 pi = ld3(S.lp[i + 1]), pj = ld3(S.lp[j + 1]);
      for (int k = i + 1; k < NL; ++k) {
        if (!((C->anc[k] >> i) & 1u)) continue;
        const f3 ck = ld3(S.lc[k]);
        const f3 li = cross(ai, ck - pi), lj = cross(aj, ck - pj);
        const float* I = S.lI[k];
        const f3 Ia = F3(I[0] * aj.x + I[3] * aj.y + I[4] * aj.z, I[3] * aj.x + I[1] * aj.y + I[5] * aj.z,
                         I[4] * aj.x + I[5] * aj.y + I[2] * aj.z);
        s += sc.link_mass[k] * dot(li, lj) + dot(ai, Ia);
      }
    }
    if (i == j) s += sc.armature[i] + h * sc.kd[i] + h * h * sc.kp[i];
    S.A[i][j] = s;
    S.A[j][i] = s;
  }
  __syncthreads();
  // left-looking Cholesky, lane = row
  for (int j = 0; j < ND; ++j) {
    float s = 0.0f;
    if (tid >= j && tid < ND) {
      s = S.A[tid][j];
      for (int k = 0; k < j; ++k) s -= S.A[tid][k] * S.A[j][k];
    }
    const float d = sqrtf(__shfl(s, j, 64));   // rows 0..22 live in wave 0; other waves idle through this loop
    if (tid >= j && tid < ND) S.A[tid][j] = (tid == j) ? d : s / d;
    __syncthreads();
  }
  // T = L^-1, lane = column
  if (tid < ND) {
    const int c = tid;
    for (int i = 0; i < c; ++i) S_T(S)[i][c] = 0.0f;
    for (int i = c; i < ND; ++i) {
      float s = (i == c) ? 1.0f : 0.0f;
      for (int k = c; k < i; ++k) s -= S.A[i][k] * S_T(S)[k][c];
      S_T(S)[i][c] = s / S.A[i][i];
    }
  }
  __syncthreads();
  // Hinv = T^T T
  for (int idx = tid; idx < ND * (ND + 1) / 2; idx += NT) {
    int i = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
    while ((i + 1) * (i + 2) / 2 <= idx) ++i;
    while (i * (i + 1) / 2 > idx) --i;
    const int j = idx - i * (i + 1) / 2;
    float s = 0.0f;
    for (int k = i; k < ND; ++k) s += S_T(S)[k][i] * S_T(S)[k][j];
    S.A[i][j] = s;
    S.A[j][i] = s;
  }
  __syncthreads();
}

// ---------------------------------------------------------------- block-level exclusive scan helpers (NT = 4 waves)
// exclusive prefix of a 0/1 flag over the workgroup in thread order; *total = number of set flags.  Two barriers.
__device__ __forceinline__ int block_scan_flag(PhysLds& S, bool flag, int tid, int* total) {
  const int lane = tid & 63, wave = tid >> 6;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const uint64_t bal = __ballot(flag);
  if (lane == 0) S.wsum[wave] = __popcll(bal);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NWAVE; ++w) { const int c = S.wsum[w]; if (w < wave) off += c; tot += c; }
  __syncthreads();
  *total = tot;
  return off + __popcll(bal & lt);
}
// exclusive prefix of a small count k (0..4)
__device__ __forceinline__ int block_scan_small(PhysLds& S, int k, int tid, int* total) {
  const int lane = tid & 63, wave = tid >> 6;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int pre = 0, wt = 0;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const uint64_t bal = __ballot((k >> b) & 1);
    pre += __popcll(bal & lt) << b;
    wt += __popcll(bal) << b;
  }
  if (lane == 0) S.wsum[wave] = wt;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NWAVE; ++w) { const int c = S.wsum[w]; if (w < wave) off += c; tot += c; }
  __syncthreads();
  *total = tot;
  return off + pre;
}

// ---------------------------------------------------------------- D: contacts
__device__ void collide(const SdxConst* C, PhysLds& S, int tid) {
  float (*cs)[SDX_MAXC] = S.stage;
  const sdx_scene_desc& sc = C->sc;
  const float off = sc.contact_offset;
  const int ns = sc.n_static;
  // ---- broadphase: fixed enumeration order (brick/static, brick/brick, then per robot box: bricks, statics)
  const int n1 = NF * ns, n2 = NF * NF, per = NF + ns, n3 = sc.n_rbox * per;
  int np = 0;
  for (int base = 0; base < n1 + n2 + n3; base += NT) {
    const int idx = base + tid;
    bool hit = false;
    uint32_t pr = 0;
    if (idx < n1) {
      const int i = idx / ns, s = idx % ns;
      const float r = C->brick_radius[sc.brick_type[i]];
      f3 stc, sth;
      static_box(sc, s, &stc, &sth);
      hit = box_sdf_val(ld3(S.bp[i]) - stc, sth) <= r + off;
      pr = (uint32_t)i | ((uint32_t)(128 + s) << 8);
    } else if (idx < n1 + n2) {
      const int t = idx - n1, i = t / NF, j = t % NF;
      if (j > i) {
        const f3 d = ld3(S.bp[i]) - ld3(S.bp[j]);
        const float rr = C->brick_radius[sc.brick_type[i]] + C->brick_radius[sc.brick_type[j]] + off;
        hit = dot(d, d) <= rr * rr;
        pr = (uint32_t)i | ((uint32_t)j << 8);
      }
    } else if (idx < n1 + n2 + n3) {
      const int t = idx - n1 - n2, r = t / per, u = t % per;
      if (sc.rbox_link[r] != 0) {
        const f3 rc = ld3(S.rc[r]);
        const float rr0 = C->rbox_radius[r];
        if (u < NF) {
          const f3 d = rc - ld3(S.bp[u]);
          const float rr = rr0 + C->brick_radius[sc.brick_type[u]] + off;
          hit = dot(d, d) <= rr * rr;
          pr = (uint32_t)(NF + r) | ((uint32_t)u << 8);
        } else {
          const int s = u - NF;
          f3 stc, sth;
          static_box(sc, s, &stc, &sth);
          hit = box_sdf_val(rc - stc, sth) <= rr0 + off;
          pr = (uint32_t)(NF + r) | ((uint32_t)(128 + s) << 8);
        }
      }
    }
    int tot;
    const int pos = np + block_scan_flag(S, hit, tid, &tot);
    if (hit && pos < SDX_MAXP) S.pairs[pos] = pr;
    np += tot;
  }
  if (np > SDX_MAXP) np = SDX_MAXP;
  __syncthreads();
  // ---- narrowphase: lane = candidate pair; contacts appended in pair order (block prefix sum of the counts)
  int nc = 0;
  for (int base = 0; base < np; base += NT) {
    const int pi = base + tid;
    int k1 = 0, k2 = 0, ida = 0, idb = 0;
    uint32_t p1 = 0, p2 = 0;
    Box A, B;
    if (pi < np) {
      const uint32_t pr = S.pairs[pi];
      const int ba = pr & 0xff, bb = (pr >> 8) & 0xff;
      A = load_box(C, S, ba);
      B = load_box(C, S, bb);
      ida = box_body(C, ba);
      idb = box_body(C, bb);
      const int c1 = sample_dir(A, B, off, &p1);
      const int c2 = (bb >= 128) ? 0 : sample_dir(B, A, off, &p2);
      const int m2 = c2 < 2 ? c2 : 2;
      k1 = c1 < 4 - m2 ? c1 : 4 - m2;
      k2 = c2 < 4 - k1 ? c2 : 4 - k1;
    }
    int tot;
    const int pre = block_scan_small(S, k1 + k2, tid, &tot);
    if (k1 > 0) emit_dir(A, B, ida, idb, p1, k1, cs, nc + pre);
    if (k2 > 0) emit_dir(B, A, idb, ida, p2, k2, cs, nc + pre + k1);
    nc += tot;
  }
  if (tid == 0) {
    S.overflow = nc > SDX_MAXC ? nc - SDX_MAXC : 0;
    S.nc = nc > SDX_MAXC ? SDX_MAXC : nc;
    S.np = np;
  }
  __syncthreads();
}

// ---------------------------------------------------------------- E: solver
// contact rows owned by this lane, held in registers for the whole solve
struct Rows {
  int a[CPT], b[CPT];
  f3 p[CPT], n[CPT];
  float sep[CPT], lam[CPT][3], wA[CPT][3], wB[CPT][3];
};

// row weight of the robot side: J Hinv J^T with J_j = (a_j x (p - p_j)) . d over the <= 11 dofs on the link's path
__device__ __attribute__((noinline)) float robot_w(const SdxConst* C, const PhysLds& S, int k, f3 p, f3 d) {
  float Jp[11];
  int idx[11];
  uint32_t m = C->anc[k];
#pragma unroll
  for (int q = 0; q < 11; ++q) {
    if (m) {
      const int j = __ffs(m) - 1;
      m &= m - 1;
      idx[q] = j;
      Jp[q] = dot(cross(ld3(S.la[j + 1]), p - ld3(S.lp[j + 1])), d);
    } else { idx[q] = 0; Jp[q] = 0.0f; }
  }
  float acc = 0.0f;
#pragma unroll
  for (int q = 0; q < 11; ++q) {
    float t = 0.0f;
#pragma unroll
    for (int r = 0; r < 11; ++r) t += S.A[idx[q]][idx[r]] * Jp[r];
    acc += Jp[q] * t;
  }
  return acc;
}

__device__ void solve(const SdxConst* C, PhysLds& S, int tid, float h, bool last_substep, long long* dbg) {
  const sdx_scene_desc& sc = C->sc;
  const int nc = S.nc;
  const float mu = sc.friction;
  Rows R;
  SSTAMP(16);
  // ---- this lane's rows from the LDS staging area into registers
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const int c = tid + q * NT;
    R.a[q] = SDX_BODY_STATIC; R.b[q] = SDX_BODY_STATIC;
    R.p[q] = F3(0, 0, 0); R.n[q] = F3(0, 0, 1); R.sep[q] = 1.0f;
    if (c < nc) {
      const int ab = __float_as_int(S.stage[0][c]);
      R.a[q] = ab & 0xff; R.b[q] = (ab >> 8) & 0xff;
      R.p[q] = F3(S.stage[1][c], S.stage[2][c], S.stage[3][c]);
      R.n[q] = F3(S.stage[4][c], S.stage[5][c], S.stage[6][c]);
      R.sep[q] = S.stage[7][c];
    }
  }
  // ---- CSR of contact sides per brick (ascending contact index inside a brick = the oracle's summation order)
  for (int i = tid; i < NF; i += NT) { S.bcount[i] = 0; S.efill[i] = 0; }
  if (tid == 0) { S.nrobot = 0; S.rfill = 0; }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < CPT; ++q)
    if (tid + q * NT < nc) {
      const int a = R.a[q], b = R.b[q];
      if (a < NF) atomicAdd(&S.bcount[a], 1); else if (a != SDX_BODY_STATIC) atomicAdd(&S.nrobot, 1);
      if (b < NF) atomicAdd(&S.bcount[b], 1); else if (b != SDX_BODY_STATIC) atomicAdd(&S.nrobot, 1);
    }
  __syncthreads();
  if (tid == 0) {
    int o = 0;
    for (int i = 0; i < NF; ++i) { S.eoff[i] = o; o += S.bcount[i]; }
    S.eoff[NF] = o;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const int c = tid + q * NT;
    if (c < nc) {
      const int a = R.a[q], b = R.b[q];
      if (a < NF) S_ENT2(S)[S.eoff[a] + atomicAdd(&S.efill[a], 1)] = (unsigned short)c;
      else if (a != SDX_BODY_STATIC) { const int i = atomicAdd(&S.rfill, 1); if (i < SDX_MAXC) S_RENT2(S)[i] = (unsigned short)c; }
      if (b < NF) S_ENT2(S)[S.eoff[b] + atomicAdd(&S.efill[b], 1)] = (unsigned short)(c | 0x8000);
      else if (b != SDX_BODY_STATIC) { const int i = atomicAdd(&S.rfill, 1); if (i < SDX_MAXC) S_RENT2(S)[i] = (unsigned short)(c | 0x8000); }
    }
  }
  __syncthreads();
  // rank pass: entry -> position = number of entries of the same brick with a smaller contact index (a contact touches
  // a brick at most once, so indices are distinct) => every brick's list is in ascending contact order, deterministically
  {
    const int total = S.eoff[NF];
    for (int i = tid; i < total; i += NT) {
      int lo = 0, hi = NF;                       // brick of entry i: largest b with eoff[b] <= i
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (S.eoff[mid] <= i) lo = mid; else hi = mid; }
      const int o = S.eoff[lo], n = S.eoff[lo + 1] - o;
      const unsigned short v = S_ENT2(S)[i];
      const int key = v & 0x7fff;
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += (S_ENT2(S)[o + j] & 0x7fff) < key;
      S.ent[o + rank] = v;
    }
    // the same for the robot-side list (one list for the whole robot; key = contact index, then side)
    const int nr = min(S.nrobot, SDX_MAXC);
    for (int i = tid; i < nr; i += NT) {
      const unsigned short v = S_RENT2(S)[i];
      const int key = ((v & 0x7fff) << 1) | (v >> 15);
      int rank = 0;
      for (int j = 0; j < nr; ++j) { const unsigned short u = S_RENT2(S)[j]; rank += (((u & 0x7fff) << 1) | (u >> 15)) < key; }
      S.rent[rank] = v;
      const int ab = __float_as_int(S.stage[0][v & 0x7fff]);
      S.rlink[rank] = (unsigned char)(((v & 0x8000) ? ((ab >> 8) & 0xff) : (ab & 0xff)) - NF);
    }
  }
  __syncthreads();
  const bool has_robot = S.nrobot > 0;   // block-uniform (read after the barrier above)
  const int nrob = min(S.nrobot, SDX_MAXC);
  // ---- un-split inverse effective masses per row and side; zero accumulated impulses
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const bool on = tid + q * NT < nc;
    f3 t1, t2;
    tangents(R.n[q], &t1, &t2);
    const f3 dir[3] = {R.n[q], t1, t2};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float wa = 0.0f, wb = 0.0f;
      if (on) {
        const int a = R.a[q], b = R.b[q];
        if (a < NF) wa = brick_w(C, S, a, R.p[q], dir[r]);
        else if (a != SDX_BODY_STATIC) wa = robot_w(C, S, a - NF, R.p[q], dir[r]);
        if (b < NF) wb = brick_w(C, S, b, R.p[q], dir[r]);
        else if (b != SDX_BODY_STATIC) wb = robot_w(C, S, b - NF, R.p[q], dir[r]);
      }
      R.wA[q][r] = wa; R.wB[q][r] = wb; R.lam[q][r] = 0.0f;
    }
  }
  if (tid < ND) { S.qds[tid] = S.qd[tid]; }
  // gather lanes: GL lanes per brick (tid = GL*brick + sub), each caches its slice of the brick's entry list (entries
  // sub, sub+4, ... up to GE of them) in registers for all iterations
  constexpr int GE = 8, GL = 4;   // GL lanes per brick, GE cached entries per lane
  const int gbrick = tid / GL, gsub = tid % GL;
  const bool glane = gbrick < NF;
  int gent[GE];
  int gn = 0, gbeg = 0, gend = 0;
  if (glane) {
    gbeg = S.eoff[gbrick]; gend = S.eoff[gbrick + 1];
#pragma unroll
    for (int k = 0; k < GE; ++k) {
      const int i = gbeg + gsub + GL * k;
      gent[k] = i < gend ? (int)S.ent[i] : -1;
      if (i < gend) gn = k + 1;
    }
  }
  __syncthreads();
  SSTAMP(17);
  float (*Pm)[SDX_MAXC] = S.stage;   // rows 0..2: impulse P of the contact (on A), rows 3..5: p x P

  for (int it = 0; it < sc.solver_iters; ++it) {
    if (it == 1) dbg = nullptr;
    SSTAMP(18);
    // pass 1 (lane = contact): relative velocity, active flag
    f3 vr[CPT];
    bool act[CPT];
    int ract = 0;
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
      act[q] = false;
      vr[q] = F3(0, 0, 0);
      const int c = tid + q * NT;
      if (c < nc) {
        const int a = R.a[q], b = R.b[q];
        vr[q] = point_vel(S, a, R.p[q]) - point_vel(S, b, R.p[q]);
        const float sep = R.sep[q];
        const float target = sep > 0 ? -sep / h : fminf(sc.baumgarte * (-sep) / h, sc.max_depenetration_vel);
        act[q] = R.lam[q][0] > 0.0f || dot(vr[q], R.n[q]) < target;
        S.act[c] = act[q] ? 1 : 0;
        if (act[q]) ract += (a >= NF && a != SDX_BODY_STATIC) + (b >= NF && b != SDX_BODY_STATIC);
      }
    }
    if (tid == 0) S.rcount = 0;
    __syncthreads();
    SSTAMP(19);
    // counts of ACTIVE contacts per brick (4 lanes per brick, quad reduction) and on the robot
    if (glane) {
      int n = 0;
#pragma unroll
      for (int k = 0; k < GE; ++k) if (k < gn) n += S.act[gent[k] & 0x7fff];
      for (int i = gbeg + gsub + GL * GE; i < gend; i += GL) n += S.act[S.ent[i] & 0x7fff];
      n += __shfl_xor(n, 1, 64);
      n += __shfl_xor(n, 2, 64);
      if (GL == 8) n += __shfl_xor(n, 4, 64);
      if (gsub == 0) S.bcount[gbrick] = n;
    }
    if (has_robot && ract) atomicAdd(&S.rcount, ract);
    __syncthreads();
    SSTAMP(20);
    // pass 2 (lane = contact): Jacobi update from the same velocity snapshot; P and p x P to LDS
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
      const int c = tid + q * NT;
      f3 P = F3(0, 0, 0);
      if (act[q]) {
        const int a = R.a[q], b = R.b[q];
        const f3 n = R.n[q];
        const float sep = R.sep[q];
        const float target = sep > 0 ? -sep / h : fminf(sc.baumgarte * (-sep) / h, sc.max_depenetration_vel);
        f3 t1, t2;
        tangents(n, &t1, &t2);
        const float na = a == SDX_BODY_STATIC ? 0.0f : (a < NF ? (float)S.bcount[a] : (float)S.rcount);
        const float nb = b == SDX_BODY_STATIC ? 0.0f : (b < NF ? (float)S.bcount[b] : (float)S.rcount);
        const float w0 = na * R.wA[q][0] + nb * R.wB[q][0];
        const float w1 = na * R.wA[q][1] + nb * R.wB[q][1];
        const float w2 = na * R.wA[q][2] + nb * R.wB[q][2];
        const float lam0 = R.lam[q][0], lam1 = R.lam[q][1], lam2 = R.lam[q][2];
        const float ln = fmaxf(0.0f, lam0 - sc.jacobi_relax * (dot(vr[q], n) - target) / w0);
        const float lim = mu * ln;
        float l1 = lam1 - sc.jacobi_relax * dot(vr[q], t1) / w1;
        l1 = fminf(lim, fmaxf(-lim, l1));
        float l2 = lam2 - sc.jacobi_relax * dot(vr[q], t2) / w2;
        l2 = fminf(lim, fmaxf(-lim, l2));
        R.lam[q][0] = ln; R.lam[q][1] = l1; R.lam[q][2] = l2;
        P = n * (ln - lam0) + t1 * (l1 - lam1) + t2 * (l2 - lam2);
      }
      if (c < nc) {
        const f3 M = cross(R.p[q], P);
        Pm[0][c] = P.x; Pm[1][c] = P.y; Pm[2][c] = P.z;
        Pm[3][c] = M.x; Pm[4][c] = M.y; Pm[5][c] = M.z;
      }
    }
    __syncthreads();
    SSTAMP(21);
    // gather (4 lanes per brick): dv = sum(+-P)/m, dw = Iw^-1 (sum(+-(p x P)) - x x sum(+-P)); each lane sums its
    // slice in ascending contact order, the four partial sums are combined in a fixed order (deterministic)
    if (glane) {
      float acc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < GE; ++k)
        if (k < gn) {
          const int e = gent[k], c = e & 0x7fff;
          if (S.act[c]) {
            const float sg = (e & 0x8000) ? -1.0f : 1.0f;
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[r] += Pm[r][c] * sg;
          }
        }
      for (int i = gbeg + gsub + GL * GE; i < gend; i += GL) {
        const int e = S.ent[i], c = e & 0x7fff;
        if (S.act[c]) {
          const float sg = (e & 0x8000) ? -1.0f : 1.0f;
#pragma unroll
          for (int r = 0; r < 6; ++r) acc[r] += Pm[r][c] * sg;
        }
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        acc[r] += __shfl_xor(acc[r], 1, 64);
        acc[r] += __shfl_xor(acc[r], 2, 64);
        if (GL == 8) acc[r] += __shfl_xor(acc[r], 4, 64);
      }
      if (gsub == 0) {
        const f3 sp = F3(acc[0], acc[1], acc[2]), sm = F3(acc[3], acc[4], acc[5]);
        const int t = sc.brick_type[gbrick];
        const f3 x = ld3(S.bp[gbrick]);
        const f4 qq = ld4(S.bq[gbrick]);
        const f3 tau = sm - cross(x, sp);
        const f3 l = qrot(qconj(qq), tau);
        const float* I = sc.brick_inertia[t];
        const float isc = gbrick == S.seg_brick ? 1.0f / sc.seg_mass_scale : 1.0f;
        const f3 dw = qrot(qq, F3(l.x / I[0], l.y / I[1], l.z / I[2])) * isc;
        const float im = isc / sc.brick_mass[t];
        st3(S.bv[gbrick], ld3(S.bv[gbrick]) + sp * im);
        st3(S.bw[gbrick], ld3(S.bw[gbrick]) + dw);
      }
    }
    if (has_robot) {
      // robot side: generalised impulse Q_j += sum over the robot's contact sides of a_j . ((p - o_j) x (+-P)) = a_j . (+-(M - o_j x P))
      // for the dofs j on the path to the touched link; 8 lanes per dof walk the (contact, side)-ordered list with a fixed
      // stride and combine in a fixed order (deterministic; LDS float atomics are not)
      constexpr int RL = 8;
      const int rt = tid - NF * GL;
      if (rt >= 0 && rt < ND * RL) {
        const int j = rt / RL, rsub = rt % RL;
        const f3 aj = ld3(S.la[j + 1]), oj = ld3(S.lp[j + 1]);
        float acc = 0.0f;
        for (int i = rsub; i < nrob; i += RL) {
          const int e = S.rent[i], c = e & 0x7fff;
          if (S.act[c] && ((C->anc[S.rlink[i]] >> j) & 1u)) {
            const f3 P = F3(Pm[0][c], Pm[1][c], Pm[2][c]), M = F3(Pm[3][c], Pm[4][c], Pm[5][c]);
            const float t = dot(aj, M - cross(oj, P));
            acc += (e & 0x8000) ? -t : t;
          }
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (rsub == 0) S.Q[j] += acc;
      }
      __syncthreads();
      if (tid < ND) {
        float sacc = S.qds[tid];
        for (int j = 0; j < ND; ++j) sacc += S.A[tid][j] * S.Q[j];
        S.qd[tid] = sacc;
      }
      __syncthreads();
      SSTAMP(22);
      twists(C, S, tid);
    } else {
      __syncthreads();
      SSTAMP(22);
    }
    SSTAMP(23);
  }
  // net contact force on the robot bodies from the last substep's accumulated impulses (GS:1094; bodies 1..6 are read)
  if (last_substep) {
    for (int i = tid; i < NL * 3; i += NT) (&S.cf[0][0])[i] = 0.0f;
    __syncthreads();
    if (has_robot) {
      const float ih = 1.0f / h;
#pragma unroll
      for (int q = 0; q < CPT; ++q) {
        const int c = tid + q * NT;
        if (c < nc) {
          f3 t1, t2;
          tangents(R.n[q], &t1, &t2);
          const f3 P = (R.n[q] * R.lam[q][0] + t1 * R.lam[q][1] + t2 * R.lam[q][2]) * ih;
          Pm[0][c] = P.x; Pm[1][c] = P.y; Pm[2][c] = P.z;
        }
      }
      __syncthreads();
      constexpr int RL = 8;
      if (tid < NL * RL) {   // 8 lanes per link over the ordered robot-side list, fixed combination order
        const int k = tid / RL, rsub = tid % RL;
        float ax = 0.0f, ay = 0.0f, az = 0.0f;
        for (int i = rsub; i < nrob; i += RL) {
          if (S.rlink[i] == k) {
            const int e = S.rent[i], c = e & 0x7fff;
            const float sg = (e & 0x8000) ? -1.0f : 1.0f;
            ax += sg * Pm[0][c]; ay += sg * Pm[1][c]; az += sg * Pm[2][c];
          }
        }
#pragma unroll
        for (int o = 1; o < RL; o <<= 1) { ax += __shfl_xor(ax, o, 64); ay += __shfl_xor(ay, o, 64); az += __shfl_xor(az, o, 64); }
        if (rsub == 0) { S.cf[k][0] = ax; S.cf[k][1] = ay; S.cf[k][2] = az; }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- outputs shared by step and refresh
__device__ void write_kinematics(const SdxConst* C, PhysLds& S, const SdxBuf& B, int e, int tid) {
  const sdx_scene_desc& sc = C->sc;
  float* rb_e = B.rb + (size_t)e * SDX_BODIES * 13;
  for (int i = tid; i < NL * 13; i += NT) {
    const int k = i / 13, c = i % 13;
    float v;
    if (c < 3) v = S.lp[k][c];
    else if (c < 7) v = S.lq[k][c - 3];
    else if (c < 10) v = S.lv[k][c - 7];
    else v = S.lw[k][c - 10];
    rb_e[i] = v;
  }
  if (tid < 42) {  // geometric Jacobian of the hand-base body origin wrt the 7 arm dofs (GS:1601)
    const int r = tid / 7, j = tid % 7, ee = sc.hand_base_body;
    const f3 a = ld3(S.la[j + 1]);
    const f3 lin = cross(a, ld3(S.lp[ee]) - ld3(S.lp[j + 1]));
    const float v = r == 0 ? lin.x : r == 1 ? lin.y : r == 2 ? lin.z : r == 3 ? a.x : r == 4 ? a.y : a.z;
    B.jac[(size_t)e * 42 + tid] = v;
  }
}

// ---------------------------------------------------------------- the step kernel
__global__ __launch_bounds__(NT, 2) void k_physics(const SdxConst* __restrict__ C, SdxBuf B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PhysLds& S = *reinterpret_cast<PhysLds*>(smem);
  const int e = blockIdx.x, tid = threadIdx.x;
  const sdx_scene_desc& sc = C->sc;
  float* root_e = B.root + (size_t)e * SDX_ACTORS * 13;
  const float h = sc.dt / (float)sc.substeps;

  // ---- load per-env state (coalesced rows) into LDS
  if (tid == 0) S.seg_brick = seg_actor(e) - SDX_ACTOR_BRICK0;
  if (tid < ND) {
    S.q[tid] = B.dof[((size_t)e * ND + tid) * 2];
    S.qd[tid] = B.dof[((size_t)e * ND + tid) * 2 + 1];
    S.tgt[tid] = B.targets[(size_t)e * ND + tid];
  }
  for (int i = tid; i < NF; i += NT) {
    const float* s = root_e + (SDX_ACTOR_BRICK0 + i) * 13;
    const f4 q = qnormalize(ld4(s + 3));
    st4(S.bq[i], q);
    st3(S.bp[i], ld3(s) + qrot(q, ld3(sc.brick_center[sc.brick_type[i]])));
    st3(S.bv[i], ld3(s + 7));
    st3(S.bw[i], ld3(s + 10));
  }
  __syncthreads();

  for (int sub = 0; sub < sc.substeps; ++sub) {
    PSTAMP(0);
    fk(C, S, tid);
    PSTAMP(1);
    if (sub == 0) mass_matrix(C, S, tid, h);   // M(q) is evaluated once per step (frozen over the substeps)
    PSTAMP(2);
    // C: implicit PD drive (P1) + gravity on the free bricks
    if (tid < ND) {
      const float t = sc.kp[tid] * (S.tgt[tid] - S.q[tid]) - (sc.kd[tid] + h * sc.kp[tid]) * S.qd[tid];
      // velocity-product bias torque of dof tid: inertial wrenches of the links below it, projected on its axis
      float tc = 0.0f;
      const f3 aj = ld3(S.la[tid + 1]), oj = ld3(S.lp[tid + 1]);
      for (int k = 1; k < NL; ++k)
        if ((C->anc[k] >> tid) & 1u) tc += dot(aj, cross(ld3(S.lc[k]) - oj, ld3(S.lF[k])) + ld3(S.lN[k]));
      S.tau[tid] = fminf(sc.effort[tid], fmaxf(-sc.effort[tid], t)) - tc;   // the effort limit applies to the drive only
      S.Q[tid] = 0.0f;
    }
    __syncthreads();
    if (tid < ND) {
      float s = 0.0f;
      for (int j = 0; j < ND; ++j) s += S.A[tid][j] * S.tau[j];
      S.qd[tid] += h * s;
    }
    for (int i = tid; i < NF; i += NT) {
      S.bv[i][0] += sc.gravity[0] * h; S.bv[i][1] += sc.gravity[1] * h; S.bv[i][2] += sc.gravity[2] * h;
    }
    __syncthreads();
    twists(C, S, tid);
    PSTAMP(3);
    collide(C, S, tid);
    PSTAMP(4);
    solve(C, S, tid, h, sub == sc.substeps - 1, sub == 0 ? B.dbg : nullptr);
    PSTAMP(5);
    // F: integrate
    if (tid < ND) {
      float v = fminf(sc.vel_limit[tid], fmaxf(-sc.vel_limit[tid], S.qd[tid]));
      float qn = S.q[tid] + h * v;
      if (qn < sc.lower[tid]) { qn = sc.lower[tid]; v = fmaxf(v, 0.0f); }
      if (qn > sc.upper[tid]) { qn = sc.upper[tid]; v = fminf(v, 0.0f); }
      S.q[tid] = qn;
      S.qd[tid] = v;
    }
    for (int i = tid; i < NF; i += NT) {
      const f3 v = ld3(S.bv[i]), w = ld3(S.bw[i]);
      st3(S.bp[i], ld3(S.bp[i]) + v * h);
      const f4 q = ld4(S.bq[i]);
      f4 wq; wq.x = w.x; wq.y = w.y; wq.z = w.z; wq.w = 0.0f;
      const f4 dq = qmul(wq, q);
      f4 nq; nq.x = q.x + 0.5f * h * dq.x; nq.y = q.y + 0.5f * h * dq.y; nq.z = q.z + 0.5f * h * dq.z;
      nq.w = q.w + 0.5f * h * dq.w;
      st4(S.bq[i], qnormalize(nq));
    }
    __syncthreads();
    PSTAMP(6);
  }

  // ---- outputs (refresh_* of GS:1091-1095)
  fk(C, S, tid);
  if (tid < ND) {
    B.dof[((size_t)e * ND + tid) * 2] = S.q[tid];
    B.dof[((size_t)e * ND + tid) * 2 + 1] = S.qd[tid];
  }
  write_kinematics(C, S, B, e, tid);
  float* rb_e = B.rb + (size_t)e * SDX_BODIES * 13;
  for (int i = tid; i < NF * 13; i += NT) {
    const int k = i / 13, c = i % 13;
    float v;
    if (c < 3) {
      const f3 o = ld3(S.bp[k]) - qrot(ld4(S.bq[k]), ld3(sc.brick_center[sc.brick_type[k]]));
      v = c == 0 ? o.x : c == 1 ? o.y : o.z;
    } else if (c < 7) v = S.bq[k][c - 3];
    else if (c < 10) v = S.bv[k][c - 7];
    else v = S.bw[k][c - 10];
    root_e[SDX_ACTOR_BRICK0 * 13 + i] = v;
    rb_e[SDX_BODY_BRICK0 * 13 + i] = v;
  }
  for (int i = tid; i < NL * 3; i += NT) B.contact[(size_t)e * SDX_BODIES * 3 + i] = (&S.cf[0][0])[i];
  if (tid == 0) B.ncontacts[e] = S.nc + S.overflow;
}

// kinematics only: rigid-body states of the 24 links + end-effector Jacobian from SDX_T_DOF
__global__ __launch_bounds__(NT) void k_kinematics(const SdxConst* __restrict__ C, SdxBuf B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PhysLds& S = *reinterpret_cast<PhysLds*>(smem);
  const int e = blockIdx.x, tid = threadIdx.x;
  if (tid < ND) {
    S.q[tid] = B.dof[((size_t)e * ND + tid) * 2];
    S.qd[tid] = B.dof[((size_t)e * ND + tid) * 2 + 1];
  }
  __syncthreads();
  fk(C, S, tid);
  write_kinematics(C, S, B, e, tid);
}

extern "C" size_t sdxk_physics_lds_bytes() { return sizeof(PhysLds); }
static void ensure_lds_attr() {   // PhysLds exceeds the default 64 KiB dynamic-LDS limit (gfx950 has 160 KiB per CU)
  static bool done = false;
  if (done) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_physics), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PhysLds));
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_kinematics), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(PhysLds));
  done = true;
}
extern "C" void sdxk_physics(const SdxConst* C, const SdxBuf* B, hipStream_t st) {
  ensure_lds_attr();
  hipLaunchKernelGGL(k_physics, dim3(B->N), dim3(NT), sizeof(PhysLds), st, C, *B);
}
extern "C" void sdxk_kinematics(const SdxConst* C, const SdxBuf* B, hipStream_t st) {
  ensure_lds_attr();
  hipLaunchKernelGGL(k_kinematics, dim3(B->N), dim3(NT), sizeof(PhysLds), st, C, *B);
}
