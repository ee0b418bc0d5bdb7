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
// Wave-level structure: "lane = link / dof / brick / matrix entry / candidate pair / contact" in turn; every
// phase is a lane-strided loop separated by workgroup barriers (the workgroup IS one wave, so a barrier is
// an s_barrier on a single wave plus the LDS fence).
#include "sdx_common.h"

#define NT SDX_WAVE
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
  float q[ND], qd[ND], tgt[ND], qds[ND], Q[ND], dQ[ND], tau[ND];
  float lq[NL][4], lp[NL][3], la[NL][3], lc[NL][3], lv[NL][3], lw[NL][3], lI[NL][6];
  float A[ND][HP];   // H -> L -> Hinv
  float T[ND][HP];   // L^-1
  float bp[NF][3], bq[NF][4], bv[NF][3], bw[NF][3], dv[NF][3], dw[NF][3];
  int bcount[NF];
  int rcount, nc, np, overflow;
  float rc[SDX_MAX_RBOX][3], rq[SDX_MAX_RBOX][4];
  uint32_t pairs[SDX_MAXP];
  float J[ND][NT];
  float cf[NL][3];
};

struct Box { f3 c; f4 q; f3 h; };

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
    b.c = ld3(sc.static_center[s]); b.q.x = 0; b.q.y = 0; b.q.z = 0; b.q.w = 1; b.h = ld3(sc.static_half[s]);
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

__device__ __forceinline__ void cwrite(float* cs, int c, int a, int b, f3 p, f3 n, float sep) {
  cs[0 * SDX_MAXC + c] = __int_as_float(a | (b << 8));
  cs[1 * SDX_MAXC + c] = p.x; cs[2 * SDX_MAXC + c] = p.y; cs[3 * SDX_MAXC + c] = p.z;
  cs[4 * SDX_MAXC + c] = n.x; cs[5 * SDX_MAXC + c] = n.y; cs[6 * SDX_MAXC + c] = n.z;
  cs[7 * SDX_MAXC + c] = sep;
}

__device__ __forceinline__ void emit_dir(const Box& A, const Box& B, int ida, int idb, uint32_t packed, int k, float* cs,
                                         int base) {
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
  return 1.0f / sc.brick_mass[t] + l.x * l.x / I[0] + l.y * l.y / I[1] + l.z * l.z / I[2];
}

// ---------------------------------------------------------------- A: FK (level-parallel over the tree)
__device__ void fk(const SdxConst* C, PhysLds& S, int tid) {
  const sdx_scene_desc& sc = C->sc;
  if (tid == 0) {
    st4(S.lq[0], ld4(sc.base_quat));
    st3(S.lp[0], ld3(sc.base_pos));
    st3(S.la[0], F3(0, 0, 1));
    st3(S.lv[0], F3(0, 0, 0));
    st3(S.lw[0], F3(0, 0, 0));
    st3(S.lc[0], ld3(sc.base_pos) + qrot(ld4(sc.base_quat), ld3(sc.link_com[0])));
  }
  __syncthreads();
  for (int d = 1; d <= C->max_depth; ++d) {
    if (tid > 0 && tid < NL && C->depth[tid] == d) {
      const int k = tid, p = sc.parent[k];
      const f4 qp = ld4(S.lq[p]);
      const f3 pp = ld3(S.lp[p]);
      const f4 qj = qmul(qp, ld4(sc.joint_quat[k]));
      const f3 ax = ld3(sc.joint_axis[k]);
      const f4 qk = qnormalize(qmul(qj, qaxis(ax, S.q[k - 1])));
      const f3 pk = pp + qrot(qp, ld3(sc.joint_pos[k]));
      const f3 ak = qrot(qj, ax);
      const f3 wp = ld3(S.lw[p]);
      st4(S.lq[k], qk);
      st3(S.lp[k], pk);
      st3(S.la[k], ak);
      st3(S.lc[k], pk + qrot(qk, ld3(sc.link_com[k])));
      st3(S.lw[k], wp + ak * S.qd[k - 1]);
      st3(S.lv[k], ld3(S.lv[p]) + cross(wp, pk - pp));
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
    const float d = sqrtf(__shfl(s, j, NT));
    if (tid >= j && tid < ND) S.A[tid][j] = (tid == j) ? d : s / d;
    __syncthreads();
  }
  // T = L^-1, lane = column
  if (tid < ND) {
    const int c = tid;
    for (int i = 0; i < c; ++i) S.T[i][c] = 0.0f;
    for (int i = c; i < ND; ++i) {
      float s = (i == c) ? 1.0f : 0.0f;
      for (int k = c; k < i; ++k) s -= S.A[i][k] * S.T[k][c];
      S.T[i][c] = s / S.A[i][i];
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
    for (int k = i; k < ND; ++k) s += S.T[k][i] * S.T[k][j];
    S.A[i][j] = s;
    S.A[j][i] = s;
  }
  __syncthreads();
}

// ---------------------------------------------------------------- D: contacts
__device__ void collide(const SdxConst* C, PhysLds& S, int tid, float* cs) {
  const sdx_scene_desc& sc = C->sc;
  const float off = sc.contact_offset;
  const int ns = sc.n_static;
  const uint64_t lt_mask = (tid == 0) ? 0ull : (~0ull >> (64 - tid));
  if (tid == 0) { S.np = 0; S.nc = 0; S.overflow = 0; }
  __syncthreads();
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
      hit = box_sdf_val(ld3(S.bp[i]) - ld3(sc.static_center[s]), ld3(sc.static_half[s])) <= r + off;
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
          hit = box_sdf_val(rc - ld3(sc.static_center[s]), ld3(sc.static_half[s])) <= rr0 + off;
          pr = (uint32_t)(NF + r) | ((uint32_t)(128 + s) << 8);
        }
      }
    }
    const uint64_t bal = __ballot(hit);
    if (hit) {
      const int pos = np + __popcll(bal & lt_mask);
      if (pos < SDX_MAXP) S.pairs[pos] = pr;
    }
    np += __popcll(bal);
  }
  if (np > SDX_MAXP) np = SDX_MAXP;
  __syncthreads();
  // ---- narrowphase: lane = candidate pair; contacts appended in pair order (wave prefix sum of the counts)
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
    const int k = k1 + k2;
    // exclusive prefix sum of k (0..4) over the wave via three ballots
    int pre = 0, tot = 0;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const uint64_t bal = __ballot((k >> b) & 1);
      pre += __popcll(bal & lt_mask) << b;
      tot += __popcll(bal) << b;
    }
    if (k1 > 0) emit_dir(A, B, ida, idb, p1, k1, cs, nc + pre);
    if (k2 > 0) emit_dir(B, A, idb, ida, p2, k2, cs, nc + pre + k1);
    nc += tot;
  }
  if (tid == 0) {
    S.overflow = nc > SDX_MAXC ? nc - SDX_MAXC : 0;
    S.nc = nc > SDX_MAXC ? SDX_MAXC : nc;
  }
  __syncthreads();
}

// ---------------------------------------------------------------- E: solver
__device__ void solve(const SdxConst* C, PhysLds& S, int tid, float* cs, float h) {
  const sdx_scene_desc& sc = C->sc;
  const int nc = S.nc;
  const float mu = sc.friction;
  // un-split inverse effective masses per row and side; zero the accumulated impulses
  for (int base = 0; base < nc; base += NT) {
    const int c = base + tid;
    const bool on = c < nc;
    int a = SDX_BODY_STATIC, b = SDX_BODY_STATIC;
    f3 p = F3(0, 0, 0), n = F3(0, 0, 1), t1, t2;
    if (on) {
      const int ab = __float_as_int(cs[c]);
      a = ab & 0xff; b = (ab >> 8) & 0xff;
      p = F3(cs[1 * SDX_MAXC + c], cs[2 * SDX_MAXC + c], cs[3 * SDX_MAXC + c]);
      n = F3(cs[4 * SDX_MAXC + c], cs[5 * SDX_MAXC + c], cs[6 * SDX_MAXC + c]);
    }
    tangents(n, &t1, &t2);
    const f3 dir[3] = {n, t1, t2};
    const int ids[2] = {a, b};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int id = ids[s];
      const bool robot = on && id >= NF && id != SDX_BODY_STATIC;
      const bool any_robot = __any(robot);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float w = 0.0f;
        if (on && id < NF) w = brick_w(C, S, id, p, dir[r]);
        if (any_robot) {  // J row staged in LDS (dof-major, lane-minor), then w = J Hinv J^T
          const int k = robot ? id - NF : 0;
          const uint32_t m = robot ? C->anc[k] : 0u;
          for (int j = 0; j < ND; ++j)
            S.J[j][tid] = ((m >> j) & 1u) ? dot(cross(ld3(S.la[j + 1]), p - ld3(S.lp[j + 1])), dir[r]) : 0.0f;
          if (robot) {
            float acc = 0.0f;
            for (int i = 0; i < ND; ++i) {
              const float Ji = S.J[i][tid];
              if (Ji == 0.0f) continue;
              float t = 0.0f;
              for (int j = 0; j < ND; ++j) t += S.A[i][j] * S.J[j][tid];
              acc += Ji * t;
            }
            w = acc;
          }
        }
        if (on) cs[(11 + 3 * s + r) * SDX_MAXC + c] = w;
      }
    }
    if (on) {
      cs[8 * SDX_MAXC + c] = 0.0f; cs[9 * SDX_MAXC + c] = 0.0f; cs[10 * SDX_MAXC + c] = 0.0f;
    }
  }
  if (tid < ND) { S.qds[tid] = S.qd[tid]; }
  __syncthreads();

  for (int it = 0; it < sc.solver_iters; ++it) {
    for (int i = tid; i < NF; i += NT) {
      S.bcount[i] = 0;
      S.dv[i][0] = S.dv[i][1] = S.dv[i][2] = 0.0f;
      S.dw[i][0] = S.dw[i][1] = S.dw[i][2] = 0.0f;
    }
    if (tid < ND) S.dQ[tid] = 0.0f;
    if (tid == 0) S.rcount = 0;
    __syncthreads();
    // pass 1: active set, per-body counts of ACTIVE contacts
    for (int base = 0; base < nc; base += NT) {
      const int c = base + tid;
      if (c < nc) {
        const int ab = __float_as_int(cs[c]);
        const int a = ab & 0xff, b = (ab >> 8) & 0xff;
        const f3 p = F3(cs[1 * SDX_MAXC + c], cs[2 * SDX_MAXC + c], cs[3 * SDX_MAXC + c]);
        const f3 n = F3(cs[4 * SDX_MAXC + c], cs[5 * SDX_MAXC + c], cs[6 * SDX_MAXC + c]);
        const float sep = cs[7 * SDX_MAXC + c];
        const float lam0 = cs[8 * SDX_MAXC + c];
        const f3 vr = point_vel(S, a, p) - point_vel(S, b, p);
        const float target = sep > 0 ? -sep / h : fminf(sc.baumgarte * (-sep) / h, sc.max_depenetration_vel);
        if (lam0 > 0.0f || dot(vr, n) < target) {
          if (a < NF) atomicAdd(&S.bcount[a], 1); else if (a != SDX_BODY_STATIC) atomicAdd(&S.rcount, 1);
          if (b < NF) atomicAdd(&S.bcount[b], 1); else if (b != SDX_BODY_STATIC) atomicAdd(&S.rcount, 1);
        }
      }
    }
    __syncthreads();
    // pass 2: Jacobi update from the same velocity snapshot
    for (int base = 0; base < nc; base += NT) {
      const int c = base + tid;
      if (c < nc) {
        const int ab = __float_as_int(cs[c]);
        const int a = ab & 0xff, b = (ab >> 8) & 0xff;
        const f3 p = F3(cs[1 * SDX_MAXC + c], cs[2 * SDX_MAXC + c], cs[3 * SDX_MAXC + c]);
        const f3 n = F3(cs[4 * SDX_MAXC + c], cs[5 * SDX_MAXC + c], cs[6 * SDX_MAXC + c]);
        const float sep = cs[7 * SDX_MAXC + c];
        float lam0 = cs[8 * SDX_MAXC + c], lam1 = cs[9 * SDX_MAXC + c], lam2 = cs[10 * SDX_MAXC + c];
        const f3 vr = point_vel(S, a, p) - point_vel(S, b, p);
        const float target = sep > 0 ? -sep / h : fminf(sc.baumgarte * (-sep) / h, sc.max_depenetration_vel);
        const float vn = dot(vr, n);
        if (lam0 > 0.0f || vn < target) {
          f3 t1, t2;
          tangents(n, &t1, &t2);
          const float na = a == SDX_BODY_STATIC ? 0.0f : (a < NF ? (float)S.bcount[a] : (float)S.rcount);
          const float nb = b == SDX_BODY_STATIC ? 0.0f : (b < NF ? (float)S.bcount[b] : (float)S.rcount);
          const float w0 = na * cs[11 * SDX_MAXC + c] + nb * cs[14 * SDX_MAXC + c];
          const float w1 = na * cs[12 * SDX_MAXC + c] + nb * cs[15 * SDX_MAXC + c];
          const float w2 = na * cs[13 * SDX_MAXC + c] + nb * cs[16 * SDX_MAXC + c];
          const float ln = fmaxf(0.0f, lam0 - sc.jacobi_relax * (vn - target) / w0);
          const float d0 = ln - lam0;
          const float lim = mu * ln;
          float l1 = lam1 - sc.jacobi_relax * dot(vr, t1) / w1;
          l1 = fminf(lim, fmaxf(-lim, l1));
          const float d1 = l1 - lam1;
          float l2 = lam2 - sc.jacobi_relax * dot(vr, t2) / w2;
          l2 = fminf(lim, fmaxf(-lim, l2));
          const float d2 = l2 - lam2;
          cs[8 * SDX_MAXC + c] = ln; cs[9 * SDX_MAXC + c] = l1; cs[10 * SDX_MAXC + c] = l2;
          const f3 P = n * d0 + t1 * d1 + t2 * d2;
          const int ids[2] = {a, b};
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const int id = ids[s];
            const f3 Ps = s == 0 ? P : P * -1.0f;
            if (id == SDX_BODY_STATIC) continue;
            if (id < NF) {
              const int t = sc.brick_type[id];
              const float im = 1.0f / sc.brick_mass[t];
              atomicAdd(&S.dv[id][0], Ps.x * im); atomicAdd(&S.dv[id][1], Ps.y * im); atomicAdd(&S.dv[id][2], Ps.z * im);
              const f4 q = ld4(S.bq[id]);
              const f3 l = qrot(qconj(q), cross(p - ld3(S.bp[id]), Ps));
              const float* I = sc.brick_inertia[t];
              const f3 dw = qrot(q, F3(l.x / I[0], l.y / I[1], l.z / I[2]));
              atomicAdd(&S.dw[id][0], dw.x); atomicAdd(&S.dw[id][1], dw.y); atomicAdd(&S.dw[id][2], dw.z);
            } else {
              uint32_t m = C->anc[id - NF];
              while (m) {
                const int j = __ffs(m) - 1;
                m &= m - 1;
                atomicAdd(&S.dQ[j], dot(cross(ld3(S.la[j + 1]), p - ld3(S.lp[j + 1])), Ps));
              }
            }
          }
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < NF; i += NT) {
      S.bv[i][0] += S.dv[i][0]; S.bv[i][1] += S.dv[i][1]; S.bv[i][2] += S.dv[i][2];
      S.bw[i][0] += S.dw[i][0]; S.bw[i][1] += S.dw[i][1]; S.bw[i][2] += S.dw[i][2];
    }
    if (tid < ND) S.Q[tid] += S.dQ[tid];
    __syncthreads();
    if (tid < ND) {
      float s = S.qds[tid];
      for (int j = 0; j < ND; ++j) s += S.A[tid][j] * S.Q[j];
      S.qd[tid] = s;
    }
    __syncthreads();
    twists(C, S, tid);
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
__global__ __launch_bounds__(NT) void k_physics(const SdxConst* __restrict__ C, SdxBuf B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  PhysLds& S = *reinterpret_cast<PhysLds*>(smem);
  const int e = blockIdx.x, tid = threadIdx.x;
  const sdx_scene_desc& sc = C->sc;
  float* root_e = B.root + (size_t)e * SDX_ACTORS * 13;
  float* cs = B.cscratch + (size_t)e * SDX_CFIELDS * SDX_MAXC;
  const float h = sc.dt / (float)sc.substeps;

  // ---- load per-env state (coalesced rows) into LDS
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
    fk(C, S, tid);
    mass_matrix(C, S, tid, h);
    // C: implicit PD drive (P1) + gravity on the free bricks
    if (tid < ND) {
      const float t = sc.kp[tid] * (S.tgt[tid] - S.q[tid]) - (sc.kd[tid] + h * sc.kp[tid]) * S.qd[tid];
      S.tau[tid] = fminf(sc.effort[tid], fmaxf(-sc.effort[tid], t));
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
    collide(C, S, tid, cs);
    solve(C, S, tid, cs, h);
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
  // net contact force on the robot bodies from the last substep's impulses (GS:1094; bodies 1..6 are read)
  for (int i = tid; i < NL * 3; i += NT) (&S.cf[0][0])[i] = 0.0f;
  __syncthreads();
  const int nc = S.nc;
  const float ih = 1.0f / h;
  for (int c = tid; c < nc; c += NT) {
    const int ab = __float_as_int(cs[c]);
    const int a = ab & 0xff, b = (ab >> 8) & 0xff;
    const bool ra = a >= NF && a != SDX_BODY_STATIC, rbb = b >= NF && b != SDX_BODY_STATIC;
    if (ra || rbb) {
      const f3 n = F3(cs[4 * SDX_MAXC + c], cs[5 * SDX_MAXC + c], cs[6 * SDX_MAXC + c]);
      f3 t1, t2;
      tangents(n, &t1, &t2);
      const f3 P = (n * cs[8 * SDX_MAXC + c] + t1 * cs[9 * SDX_MAXC + c] + t2 * cs[10 * SDX_MAXC + c]) * ih;
      if (ra) { atomicAdd(&S.cf[a - NF][0], P.x); atomicAdd(&S.cf[a - NF][1], P.y); atomicAdd(&S.cf[a - NF][2], P.z); }
      if (rbb) { atomicAdd(&S.cf[b - NF][0], -P.x); atomicAdd(&S.cf[b - NF][1], -P.y); atomicAdd(&S.cf[b - NF][2], -P.z); }
    }
  }
  __syncthreads();
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
extern "C" void sdxk_physics(const SdxConst* C, const SdxBuf* B, hipStream_t st) {
  hipLaunchKernelGGL(k_physics, dim3(B->N), dim3(NT), sizeof(PhysLds), st, C, *B);
}
extern "C" void sdxk_kinematics(const SdxConst* C, const SdxBuf* B, hipStream_t st) {
  hipLaunchKernelGGL(k_kinematics, dim3(B->N), dim3(NT), sizeof(PhysLds), st, C, *B);
}
