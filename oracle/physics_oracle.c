/* TEST INFRASTRUCTURE ONLY (oracle/): plain-C, scalar, sequential restatement of the per-env physics step
 * that seqdex_amd/csrc/sdx_physics.hip runs one wavefront per env (DESIGN.md §3 "SDX-1 physics").
 *
 * PARITY UNPINNED: the reference's physics lives in Isaac Gym / PhysX (closed binary, absent from
 * /root/reference; call sites BT:138-144 gym.simulate/fetch_results, GS:1091-1095 refresh_*, parameters
 * CF:185-217 + cfg/allegro_hand_block_assembly_grasp_sim.yaml:155-167, scene GS:523-1058).  There is no
 * reference-side test or golden vector for it, so this file DEFINES the step (from the scene constants
 * A0/A1 of SURVEY.md §8(a)) and is pinned only by the known-answer tests we author in
 * tests/test_physics_oracle.py (free fall, resting contact, PD response, FK/Jacobian finite differences, stacks,
 * the compound-shape cases: hull profile, seated hollow brick on studs).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this code.
 *
 * Algorithm per env and per substep h = dt/substeps  (P1..P5 of SURVEY.md §8(a)):
 *   A  forward kinematics of the 24-body tree, link twists
 *   B  joint-space inertia M(q) (composite sums, once per step), H = M + diag(armature + h kd + h^2 kp), Hinv
 *   C  implicit PD drive (P1):  qd += Hinv h clamp(kp(q*-q) - (kd + h kp) qd, +-effort);  bricks: v += h g
 *   D  contacts (P3): every body is a compound of boxes (hull slabs of a brick, the hollow target brick of InsertSim,
 *      studded base plates, robot boxes).  Body pairs by bounding box; a pair of convex bodies
 *      contributes the box pair with the smallest separation bound, a pair with a compound side all of its box pairs;
 *      sample points of one box against the analytic SDF of the other, both directions, <= 4 contacts per box pair
 *      chosen to span the patch, kept when separation < contact_offset
 *   E  solve (P4): `solver_iters` mass-split Jacobi iterations on accumulated impulses, normal +
 *      2 friction rows, robot side in reduced coordinates (qd = qd* + Hinv J^T lambda)
 *   F  integrate (P5): semi-implicit Euler, joint limit / velocity clamps, quaternion renormalisation
 * then once per step: FK for outputs, end-effector Jacobian, net contact force of the last substep.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/seqdex.h"

#define NL SDX_NLINK
#define ND SDX_NDOF
#define NF SDX_NFREE
#ifndef SDXO_MAXC
#define SDXO_MAXC 1536
#endif
#define NSAMP 28
#define BODY_STATIC 255

typedef float real;

/* ---------------------------------------------------------------- small math */
typedef struct { real x, y, z; } v3;
typedef struct { real x, y, z, w; } q4;

static v3 V(real x, real y, real z) { v3 r = {x, y, z}; return r; }
static v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 vscale(v3 a, real s) { return V(a.x * s, a.y * s, a.z * s); }
static real vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 vcross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static q4 qmul(q4 a, q4 b) {
  q4 r;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  return r;
}
static q4 qconj(q4 a) { q4 r = {-a.x, -a.y, -a.z, a.w}; return r; }
static v3 qrot(q4 q, v3 v) { /* v + w t + u x t, t = 2 u x v */
  v3 u = V(q.x, q.y, q.z);
  v3 t = vscale(vcross(u, v), 2.0f);
  return vadd(vadd(v, vscale(t, q.w)), vcross(u, t));
}
static q4 qnormalize(q4 a) {
  real n = 1.0f / sqrtf(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
  q4 r = {a.x * n, a.y * n, a.z * n, a.w * n};
  return r;
}
static q4 qaxis(v3 ax, real ang) {
  real s = sinf(0.5f * ang), c = cosf(0.5f * ang);
  q4 r = {ax.x * s, ax.y * s, ax.z * s, c};
  return r;
}
static v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
static q4 ld4(const float* p) { q4 r = {p[0], p[1], p[2], p[3]}; return r; }
static void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static void st4(float* p, q4 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; p[3] = a.w; }

/* sample points of a box in units of its half extents: 8 corners, 12 edge midpoints, 8 quarter points on the
 * x-parallel edges.  Corners come first: selection is "first k in table order" (DESIGN.md §3.D). */
static const real SAMP[NSAMP][3] = {
    {1, 1, 1}, {1, -1, -1}, {-1, 1, -1}, {-1, -1, 1}, {-1, -1, -1}, {-1, 1, 1}, {1, -1, 1}, {1, 1, -1},
    {0, -1, -1}, {0, 1, -1}, {0, -1, 1}, {0, 1, 1}, {-1, 0, -1}, {1, 0, -1}, {-1, 0, 1}, {1, 0, 1},
    {-1, -1, 0}, {1, -1, 0}, {-1, 1, 0}, {1, 1, 0},
    {-0.5f, -1, -1}, {0.5f, -1, -1}, {-0.5f, 1, -1}, {0.5f, 1, -1}, {-0.5f, -1, 1}, {0.5f, -1, 1}, {-0.5f, 1, 1}, {0.5f, 1, 1}};

/* ---------------------------------------------------------------- per-env working state */
typedef struct {
  /* robot */
  real q[ND], qd[ND], tgt[ND];
  q4 lq[NL];           /* link orientation */
  v3 lp[NL];           /* link frame origin (= joint anchor) */
  v3 la[NL];           /* joint axis, world */
  v3 lc[NL];           /* link COM, world */
  v3 lv[NL], lw[NL];   /* link origin linear velocity, angular velocity */
  unsigned anc[NL];    /* bit j set: dof j (link j+1) is on the path base -> this link */
  real H[ND][ND], Hinv[ND][ND];
  real qd_star[ND], Q[ND];
  /* bricks (COM frame) */
  v3 bp[NF], bv[NF], bw[NF];
  q4 bq[NF];
  v3 dv[NF], dw[NF];
  int bcount[NF];
  int rcount;          /* contacts touching the robot */
  /* collision boxes of the robot in world */
  v3 rc[SDX_MAX_RBOX];
  q4 rq[SDX_MAX_RBOX];
  /* contacts */
  int nc;
  unsigned char ca[SDXO_MAXC], cb[SDXO_MAXC]; /* body ids: 0..71 brick, 72+k robot link k, 255 static */
  v3 cn[SDXO_MAXC], cp[SDXO_MAXC];
  real csep[SDXO_MAXC];
  unsigned ckey[SDXO_MAXC], cur_key; /* identity of every contact: (box a, box b, direction, sample) - the warm-start key */
  int* wcount; unsigned* wkey; float* wlam; /* this env's impulse cache: count, keys [MAXC], impulses [3][MAXC] of the previous solve */
  real lam[SDXO_MAXC][3], w[SDXO_MAXC][2][3]; /* w[c][side][row]: un-split inverse effective mass */
  unsigned char active[SDXO_MAXC];
  unsigned char cage[SDXO_MAXC]; /* warm start: consecutive solves each contact has existed before this one */
  int overflow;
  real incl;           /* inclusion threshold of the current collide pass */
  int rebuilt;         /* the last collide() had to drop the speculative contacts */
  int env_index;
  int seg_brick; /* this env's target brick: mass and inertia scaled by sc->seg_mass_scale */
} env_t;

typedef struct { v3 c; q4 q; v3 h; } box_t;

/* ---------------------------------------------------------------- A: forward kinematics */
static void fk(const sdx_scene_desc* sc, env_t* e) {
  e->lq[0] = ld4(sc->base_quat);
  e->lp[0] = ld3(sc->base_pos);
  e->la[0] = V(0, 0, 1);
  e->lv[0] = e->lw[0] = V(0, 0, 0);
  e->anc[0] = 0;
  e->lc[0] = vadd(e->lp[0], qrot(e->lq[0], ld3(sc->link_com[0])));
  for (int k = 1; k < NL; ++k) { /* parents precede children (depth-first order) */
    int p = sc->parent[k];
    q4 qj = qmul(e->lq[p], ld4(sc->joint_quat[k]));
    v3 ax = ld3(sc->joint_axis[k]);
    e->lq[k] = qnormalize(qmul(qj, qaxis(ax, e->q[k - 1])));
    e->lp[k] = vadd(e->lp[p], qrot(e->lq[p], ld3(sc->joint_pos[k])));
    e->la[k] = qrot(qj, ax);
    e->lc[k] = vadd(e->lp[k], qrot(e->lq[k], ld3(sc->link_com[k])));
    e->lw[k] = vadd(e->lw[p], vscale(e->la[k], e->qd[k - 1]));
    e->lv[k] = vadd(e->lv[p], vcross(e->lw[p], vsub(e->lp[k], e->lp[p])));
    e->anc[k] = e->anc[p] | (1u << (k - 1));
  }
  for (int r = 0; r < sc->n_rbox; ++r) {
    int k = sc->rbox_link[r];
    e->rc[r] = vadd(e->lp[k], qrot(e->lq[k], ld3(sc->rbox_center[r])));
    e->rq[r] = qmul(e->lq[k], ld4(sc->rbox_quat[r]));
  }
}

/* world inertia times vector: R diag-full(I) R^T x, I given as xx yy zz xy xz yz in body frame */
static v3 inertia_mul(q4 q, const float* I, v3 x) {
  v3 l = qrot(qconj(q), x);
  v3 m = V(I[0] * l.x + I[3] * l.y + I[4] * l.z, I[3] * l.x + I[1] * l.y + I[5] * l.z, I[4] * l.x + I[5] * l.y + I[2] * l.z);
  return qrot(q, m);
}

/* ---------------------------------------------------------------- A2: velocity-product bias torques */
/* C(q,qd) qd of the fixed-base tree = recursive Newton-Euler with zero joint accelerations (no gravity on the robot, GS:544):
 * link angular accelerations al, linear accelerations of the link origins ao, then the inertial wrench of every link about
 * its centre of mass, projected on the joint axes of its ancestors. */
static void coriolis(const sdx_scene_desc* sc, const env_t* e, real tauc[ND]) {
  v3 al[NL], ao[NL], F[NL], Nn[NL];
  al[0] = ao[0] = F[0] = Nn[0] = V(0, 0, 0);
  for (int k = 1; k < NL; ++k) {
    int p = sc->parent[k];
    v3 r = vsub(e->lp[k], e->lp[p]);
    al[k] = vadd(al[p], vcross(e->lw[p], vscale(e->la[k], e->qd[k - 1])));
    ao[k] = vadd(ao[p], vadd(vcross(al[p], r), vcross(e->lw[p], vcross(e->lw[p], r))));
    v3 d = vsub(e->lc[k], e->lp[k]);
    v3 acom = vadd(ao[k], vadd(vcross(al[k], d), vcross(e->lw[k], vcross(e->lw[k], d))));
    F[k] = vscale(acom, sc->link_mass[k]);
    Nn[k] = vadd(inertia_mul(e->lq[k], sc->link_inertia[k], al[k]),
                 vcross(e->lw[k], inertia_mul(e->lq[k], sc->link_inertia[k], e->lw[k])));
  }
  for (int j = 0; j < ND; ++j) {
    real s = 0;
    for (int k = 1; k < NL; ++k)
      if ((e->anc[k] >> j) & 1u) s += vdot(e->la[j + 1], vadd(vcross(vsub(e->lc[k], e->lp[j + 1]), F[k]), Nn[k]));
    tauc[j] = s;
  }
}

/* ---------------------------------------------------------------- B: joint-space inertia and its inverse */
static void mass_matrix(const sdx_scene_desc* sc, env_t* e, real h) {
  for (int i = 0; i < ND; ++i)
    for (int j = 0; j <= i; ++j) {
      real s = 0;
      for (int k = 1; k < NL; ++k) {
        if (!((e->anc[k] >> i) & 1u) || !((e->anc[k] >> j) & 1u)) continue;
        v3 ai = e->la[i + 1], aj = e->la[j + 1];
        v3 li = vcross(ai, vsub(e->lc[k], e->lp[i + 1]));
        v3 lj = vcross(aj, vsub(e->lc[k], e->lp[j + 1]));
        s += sc->link_mass[k] * vdot(li, lj) + vdot(ai, inertia_mul(e->lq[k], sc->link_inertia[k], aj));
      }
      e->H[i][j] = e->H[j][i] = s;
    }
  for (int i = 0; i < ND; ++i) e->H[i][i] += sc->armature[i] + h * sc->kd[i] + h * h * sc->kp[i];
  /* Cholesky H = L L^T (in a scratch copy), then Hinv = L^-T L^-1 */
  real L[ND][ND], Li[ND][ND];
  memset(L, 0, sizeof(L));
  memset(Li, 0, sizeof(Li));
  for (int j = 0; j < ND; ++j) {
    real d = e->H[j][j];
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    d = sqrtf(d);
    L[j][j] = d;
    for (int i = j + 1; i < ND; ++i) {
      real s = e->H[i][j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      L[i][j] = s / d;
    }
  }
  for (int c = 0; c < ND; ++c) /* Li = L^-1, column by column (forward substitution) */
    for (int i = c; i < ND; ++i) {
      real s = (i == c) ? 1.0f : 0.0f;
      for (int k = c; k < i; ++k) s -= L[i][k] * Li[k][c];
      Li[i][c] = s / L[i][i];
    }
  for (int i = 0; i < ND; ++i)
    for (int j = 0; j <= i; ++j) {
      real s = 0;
      for (int k = i; k < ND; ++k) s += Li[k][i] * Li[k][j];
      e->Hinv[i][j] = e->Hinv[j][i] = s;
    }
}

/* ---------------------------------------------------------------- D: box / box sampled-SDF contacts */
/* signed distance of point p (box frame) to box with half extents h; *g = outward unit gradient (box frame) */
static real box_sdf(v3 p, v3 h, v3* g) {
  v3 a = V(fabsf(p.x), fabsf(p.y), fabsf(p.z));
  v3 d = vsub(a, h);
  v3 s = V(p.x < 0 ? -1.0f : 1.0f, p.y < 0 ? -1.0f : 1.0f, p.z < 0 ? -1.0f : 1.0f);
  real mx = fmaxf(d.x, fmaxf(d.y, d.z));
  if (mx <= 0) { /* inside: nearest face */
    if (d.x >= d.y && d.x >= d.z) *g = V(s.x, 0, 0);
    else if (d.y >= d.z) *g = V(0, s.y, 0);
    else *g = V(0, 0, s.z);
    return mx;
  }
  v3 o = V(fmaxf(d.x, 0), fmaxf(d.y, 0), fmaxf(d.z, 0));
  real len = sqrtf(vdot(o, o));
  *g = V(s.x * o.x / len, s.y * o.y / len, s.z * o.z / len);
  return len;
}

static void add_contact(env_t* e, int a, int b, v3 p, v3 n, real sep) {
  if (e->nc >= SDXO_MAXC) { e->overflow++; return; }
  int c = e->nc++;
  e->ca[c] = (unsigned char)a;
  e->cb[c] = (unsigned char)b;
  e->cp[c] = p;
  e->cn[c] = n;
  e->csep[c] = sep;
  e->ckey[c] = e->cur_key;
}

/* ---- contact manifold of one direction (samples of A against box B), DESIGN.md section 3.D
 * Reference face: the face axis k of B with the smallest overlap of the two boxes' extents (separating-axis test over B's three face
 * normals: s_k = |t_k| - hB_k - sum_j |R_kj| hA_j, largest s wins, ties x > y > z), on the side of B's centre where A's centre lies.
 * A sample of A whose projection falls on that face (within FACE_TOL of its outline) is a FACE sample: normal = the face normal,
 * separation measured along it.  Every other sample uses the point's own signed distance to B (edge / corner regions).
 * Up to 4 samples per direction: face samples in table order first, then the others in table order.
 * Only for shallow overlaps (s_k >= -FACE_DEPTH x contact offset): a deeply interpenetrating pair (bricks spawned inside the floor slab)
 * has no meaningful meeting face and every sample keeps its own signed distance, as in round 1.
 * (Round 1 used the sample's nearest face and plain table order: two equal bricks stacked flush pushed each other sideways, the
 * speculative samples beside B used up the 4 slots, and the upper brick tipped over one edge or sank through.) */
#define FACE_TOL 1e-4f
#define FACE_DEPTH 4.0f
#define WARM_SPEED 0.25f /* m/s: relative speed at the contact point (before the solve) above which a contact starts cold */
#define WARM_DEPTH 1.5f /* contacts deeper than this many contact offsets are not warm-started (section 3.E) */
typedef struct { v3 t, ex, ey, ez; int kax; real sgn, smax; } dir_t; /* ex, ey, ez: A's half edges in B's frame: sample = t + sx ex + sy ey + sz ez */

static dir_t dir_setup(const box_t* A, const box_t* B, real offset) {
  dir_t D;
  q4 qbi = qconj(B->q);
  D.t = qrot(qbi, vsub(A->c, B->c));
  q4 qrel = qmul(qbi, A->q);
  D.ex = qrot(qrel, V(A->h.x, 0, 0));
  D.ey = qrot(qrel, V(0, A->h.y, 0));
  D.ez = qrot(qrel, V(0, 0, A->h.z));
  v3 ex = D.ex, ey = D.ey, ez = D.ez;
  real sx = fabsf(D.t.x) - B->h.x - (fabsf(ex.x) + fabsf(ey.x) + fabsf(ez.x));
  real sy = fabsf(D.t.y) - B->h.y - (fabsf(ex.y) + fabsf(ey.y) + fabsf(ez.y));
  real sz = fabsf(D.t.z) - B->h.z - (fabsf(ex.z) + fabsf(ey.z) + fabsf(ez.z));
  if (sx >= sy && sx >= sz) { D.kax = 0; D.sgn = D.t.x < 0 ? -1.0f : 1.0f; }
  else if (sy >= sz) { D.kax = 1; D.sgn = D.t.y < 0 ? -1.0f : 1.0f; }
  else { D.kax = 2; D.sgn = D.t.z < 0 ? -1.0f : 1.0f; }
  D.smax = fmaxf(sx, fmaxf(sy, sz)); /* >= offset: a face axis of B separates the boxes by the whole contact offset: no sample can be inside it */
  if (D.smax < -FACE_DEPTH * offset) D.kax = -1;
  return D;
}

static v3 sample_point(const dir_t* D, int s) {
  return vadd(vadd(vadd(D->t, vscale(D->ex, SAMP[s][0])), vscale(D->ey, SAMP[s][1])), vscale(D->ez, SAMP[s][2]));
}

/* one sample (B frame): returns 1 = face sample, 2 = other sample inside the contact offset, 0 = no contact; *g, *sd as box_sdf */
static int sample_contact(const dir_t* D, v3 pb, v3 h, real offset, v3* g, real* sd) {
  v3 d = V(fabsf(pb.x) - h.x, fabsf(pb.y) - h.y, fabsf(pb.z) - h.z);
  real lat = D->kax == 0 ? fmaxf(d.y, d.z) : D->kax == 1 ? fmaxf(d.x, d.z) : fmaxf(d.x, d.y);
  if (D->kax >= 0 && lat <= FACE_TOL) {
    real pk = D->kax == 0 ? pb.x : D->kax == 1 ? pb.y : pb.z, hk = D->kax == 0 ? h.x : D->kax == 1 ? h.y : h.z;
    *sd = D->sgn * pk - hk;
    *g = V(D->kax == 0 ? D->sgn : 0.0f, D->kax == 1 ? D->sgn : 0.0f, D->kax == 2 ? D->sgn : 0.0f);
    return *sd < offset ? 1 : 0;
  }
  *sd = box_sdf(pb, h, g);
  return *sd < offset ? 2 : 0;
}

/* class of one sample without the distance itself (the kernel's test: squared distance against the squared offset, no sqrt) */
static int sample_class(const dir_t* D, v3 pb, v3 h, real offset) {
  v3 d = V(fabsf(pb.x) - h.x, fabsf(pb.y) - h.y, fabsf(pb.z) - h.z);
  real lat = D->kax == 0 ? fmaxf(d.y, d.z) : D->kax == 1 ? fmaxf(d.x, d.z) : fmaxf(d.x, d.y);
  if (D->kax >= 0 && lat <= FACE_TOL) {
    real pk = D->kax == 0 ? pb.x : D->kax == 1 ? pb.y : pb.z, hk = D->kax == 0 ? h.x : D->kax == 1 ? h.y : h.z;
    return D->sgn * pk - hk < offset ? 1 : 0;
  }
  if (fmaxf(d.x, fmaxf(d.y, d.z)) <= 0) return 2;
  v3 o = V(fmaxf(d.x, 0), fmaxf(d.y, 0), fmaxf(d.z, 0));
  return vdot(o, o) < offset * offset ? 2 : 0;
}

static void emit_mask(env_t* e, const box_t* A, const box_t* B, int ida, int idb, unsigned mask, real offset, unsigned pkey) {
  dir_t D = dir_setup(A, B, offset);
  for (int s = 0; s < NSAMP; ++s) {
    if (!((mask >> s) & 1u)) continue;
    v3 pb = sample_point(&D, s);
    v3 g;
    real sd;
    sample_contact(&D, pb, B->h, 1e30f, &g, &sd);
    v3 n = qrot(B->q, g); /* out of B, towards A */
    v3 pw = vadd(B->c, qrot(B->q, pb));
    e->cur_key = pkey | (unsigned)s;
    add_contact(e, ida, idb, vsub(pw, vscale(n, 0.5f * sd)), n, sd);
  }
}

/* ---- compound shapes.  A brick collides as a compound of axis-aligned boxes in its own frame: the slabs of its
 * convex hull (sdx_scene_desc.brick_sub_*), or - the env's target brick when seg_hollow is set - the hollow compound of its mesh
 * (hollow_sub_*).  A static body is a compound too (static_sub_*: the studded base plate; every other static is one box).  A robot box is
 * one box.  The body's bounding box (brick_half about brick_center) serves the broadphase only. */
static int brick_is_hollow(const sdx_scene_desc* sc, const env_t* e, int i) { return sc->seg_hollow && i == e->seg_brick; }
static int brick_nsub(const sdx_scene_desc* sc, const env_t* e, int i) {
  int t = sc->brick_type[i];
  return brick_is_hollow(sc, e, i) ? sc->hollow_nsub[t] : sc->brick_nsub[t];
}
static box_t brick_sub(const sdx_scene_desc* sc, const env_t* e, int i, int k) {
  int t = sc->brick_type[i];
  const float* c = brick_is_hollow(sc, e, i) ? sc->hollow_sub_center[t][k] : sc->brick_sub_center[t][k];
  const float* h = brick_is_hollow(sc, e, i) ? sc->hollow_sub_half[t][k] : sc->brick_sub_half[t][k];
  box_t B = {vadd(e->bp[i], qrot(e->bq[i], vsub(ld3(c), ld3(sc->brick_com[t])))), e->bq[i], ld3(h)};
  return B;
}
static box_t brick_bound(const sdx_scene_desc* sc, const env_t* e, int i) {
  int t = sc->brick_type[i];
  box_t B = {vadd(e->bp[i], qrot(e->bq[i], vsub(ld3(sc->brick_center[t]), ld3(sc->brick_com[t])))), e->bq[i], ld3(sc->brick_half[t])};
  return B;
}
/* row of the static-body table that slot s shows to env `env_index`: InsertSim's base plate is one of three by env % 3 */
static int static_row(const sdx_scene_desc* sc, int s, int env_index) {
  return s == sc->static_var_slot ? sc->static_var_row[env_index % 3] : s;
}
static box_t static_bound(const sdx_scene_desc* sc, int s, int env_index) {
  int r = static_row(sc, s, env_index);
  box_t S = {ld3(sc->static_center[r]), {0, 0, 0, 1}, ld3(sc->static_half[r])};
  return S;
}
static box_t static_sub(const sdx_scene_desc* sc, int s, int env_index, int k) {
  int r = static_row(sc, s, env_index), b = sc->static_sub_first[r] + k;
  box_t S = {ld3(sc->static_sub_center[b]), {0, 0, 0, 1}, ld3(sc->static_sub_half[b])};
  return S;
}

/* pair of boxes -> <= 4 contacts.  Samples of A are classified against B (direction 1) and, when sample_b, samples
 * of B against A (direction 2; sample_b == 0: B is the body box of a static).  incl (e->incl): samples closer than this are contacts -
 * the contact offset, or 0 when the list is rebuilt after a capacity overflow.  The 4 slots go to
 *   1. FACE samples of both directions, chosen to SPAN the patch the boxes meet on.  Every face sample has lateral coordinates (a, b) in
 *      B's frame (B's axes without the axis of direction 1's reference face; a sample of B: its own table entry x hB).  p1 = the sample
 *      furthest along (1, 0.1) (the slight tilt decides between the two corners of an axis-aligned edge); p2 = the sample furthest from
 *      p1; p3 / p4 = the samples furthest to the left / to the right of the line p1 p2 (more than MANIFOLD_EPS off it).  Ties: the first
 *      in enumeration order (direction 1 in table order, then direction 2).  Remaining slots: the other face samples in that order;
 *   2. the other samples (edge / corner regions, speculative contacts) of direction 1, then of direction 2, in table order.
 * (Until round 4: per direction the first four in table order, face samples first, two slots reserved for direction 2 - on the narrower
 * boxes of a compound that picked four points at one end of the patch, or gave a slot to a speculative sample beside it.)
 * pkey: the pair's part of the contact identity (enumeration index of the body pair << 15 | index of the box pair inside it << 6) - with
 * the direction bit and the sample index it identifies a contact from one solve to the next */
#define MANIFOLD_EPS 1e-7f /* m^2: twice the area of the triangle (p1, p2, p) below which p counts as lying on the line p1 p2 */
/* ties go to the first candidate in enumeration order: a later one replaces the incumbent only when it is better by more than these margins
 * (p1's score in m; squared distance / line offset in m^2) - the samples of one box edge are equally far from a line parallel to it and
 * coincident samples of the two directions score the same, so that rounding would decide otherwise */
#define MANIFOLD_TIE_L 1e-6f
#define MANIFOLD_TIE_A 1e-8f
static void face_coords(const dir_t* D1, const box_t* B, int kref, int d, int s, real* a, real* b) {
  v3 p = d ? V(SAMP[s][0] * B->h.x, SAMP[s][1] * B->h.y, SAMP[s][2] * B->h.z) : sample_point(D1, s); /* the sample in B's frame */
  *a = kref == 0 ? p.y : p.x;
  *b = kref == 2 ? p.y : p.z;
}
static void collide_pair(env_t* e, const box_t* A, const box_t* B, int ida, int idb, int sample_b, real offset, unsigned pkey) {
  const real incl = e->incl;
  dir_t D1 = dir_setup(A, B, offset), D2;
  if (D1.smax >= incl) return; /* separated: no sample of either direction can be inside the threshold */
  if (sample_b) {
    D2 = dir_setup(B, A, offset);
    if (D2.smax >= incl) return;
  }
  const int kref = D1.kax >= 0 ? D1.kax : 2;
  unsigned face[2] = {0, 0}, other[2] = {0, 0};
  for (int d = 0; d < (sample_b ? 2 : 1); ++d) {
    const dir_t* D = d ? &D2 : &D1;
    const box_t* T = d ? A : B; /* the box the samples are tested against */
    for (int s = 0; s < NSAMP; ++s) {
      int cls = sample_class(D, sample_point(D, s), T->h, incl);
      if (cls == 1) face[d] |= 1u << s;
      if (cls == 2) other[d] |= 1u << s;
    }
  }
  unsigned sel[2] = {0, 0};
  int n = 0;
  if (face[0] | face[1]) {
    int p1 = -1, p2 = -1, p3 = -1, p4 = -1;
    real best = -1e30f, a1 = 0, b1 = 0, a2 = 0, b2 = 0, a, b;
    for (int id = 0; id < 64; ++id)
      if ((face[id >> 5] >> (id & 31)) & 1u) {
        face_coords(&D1, B, kref, id >> 5, id & 31, &a, &b);
        real k = a + 0.1f * b;
        if (k > best + MANIFOLD_TIE_L) { best = k; p1 = id; a1 = a; b1 = b; }
      }
    best = 0.0f;
    for (int id = 0; id < 64; ++id)
      if ((face[id >> 5] >> (id & 31)) & 1u) {
        face_coords(&D1, B, kref, id >> 5, id & 31, &a, &b);
        real k = (a - a1) * (a - a1) + (b - b1) * (b - b1);
        if (k > best + MANIFOLD_TIE_A) { best = k; p2 = id; a2 = a; b2 = b; }
      }
    if (p2 >= 0) {
      real hi = MANIFOLD_EPS, lo = -MANIFOLD_EPS;
      for (int id = 0; id < 64; ++id)
        if ((face[id >> 5] >> (id & 31)) & 1u) {
          face_coords(&D1, B, kref, id >> 5, id & 31, &a, &b);
          real k = (a2 - a1) * (b - b1) - (b2 - b1) * (a - a1);
          if (k > hi + MANIFOLD_TIE_A) { hi = k; p3 = id; }
          if (k < lo - MANIFOLD_TIE_A) { lo = k; p4 = id; }
        }
    }
    int pk[4] = {p1, p2, p3, p4};
    for (int k = 0; k < 4; ++k)
      if (pk[k] >= 0) { sel[pk[k] >> 5] |= 1u << (pk[k] & 31); ++n; }
  }
  for (int pass = 0; pass < 2; ++pass) /* remaining face samples, then the other samples */
    for (int d = 0; d < 2; ++d) {
      unsigned m = (pass ? other[d] : face[d]) & ~sel[d];
      for (int sidx = 0; sidx < NSAMP && n < 4; ++sidx)
        if ((m >> sidx) & 1u) { sel[d] |= 1u << sidx; ++n; }
    }
  emit_mask(e, A, B, ida, idb, sel[0], offset, pkey);
  if (sel[1]) emit_mask(e, B, A, idb, ida, sel[1], offset, pkey | 0x20u);
}

static real box_radius(v3 h) { return sqrtf(vdot(h, h)); }

/* largest face-axis separation of the box pair (both directions when the second one is sampled): < offset for every pair that can
 * produce a contact */
static real pair_sigma(const box_t* A, const box_t* B, int sample_b, real offset) {
  real sg = dir_setup(A, B, offset).smax;
  if (sample_b) sg = fmaxf(sg, dir_setup(B, A, offset).smax);
  return sg;
}

/* one body pair of the enumeration (index idx).  kind_a: 0 brick, 1 robot box; kind_b: 0 brick, 2 static slot.
 * CONVEX pairs - both sides stand for one convex shape: a brick as the slab compound of its hull, a robot box, a single-box static -
 * contribute the contacts of ONE box pair, the one with the smallest separation bound sigma (ties: the first in enumeration order; none
 * when every pair is separated by the contact offset): a pair of convex shapes has one contact patch, and as the configuration changes the
 * winning pair of boxes changes with it.  COMPOUND pairs - a hollow brick (its underside takes studs) or the studded base plate on either
 * side - contribute every box pair.  The studs of a static compound (boxes 1..) are sampled like a brick's boxes, a static BODY box (box 0)
 * only receives the other shape's samples. */
static void collide_bodies(const sdx_scene_desc* sc, env_t* e, int idx, int kind_a, int ia, int kind_b, int ib, const box_t* R, int rlink) {
  const real off = sc->contact_offset;
  int na = kind_a == 0 ? brick_nsub(sc, e, ia) : 1;
  int nb = kind_b == 0 ? brick_nsub(sc, e, ib) : sc->static_sub_n[static_row(sc, ib, e->env_index)];
  int convex = !(kind_a == 0 && brick_is_hollow(sc, e, ia)) && !(kind_b == 0 && brick_is_hollow(sc, e, ib)) && !(kind_b == 2 && nb > 1);
  int only = -1;
  if (convex) {
    real best = off;
    for (int s = 0; s < na * nb; ++s) {
      int a = s / nb, b = s % nb;
      box_t A = kind_a == 0 ? brick_sub(sc, e, ia, a) : *R;
      box_t B = kind_b == 0 ? brick_sub(sc, e, ib, b) : static_sub(sc, ib, e->env_index, b);
      real sg = pair_sigma(&A, &B, kind_b == 0 || b > 0, off);
      if (sg < best) { best = sg; only = s; }
    }
    if (only < 0) return;
  }
  for (int a = 0; a < na; ++a)
    for (int b = 0; b < nb; ++b) {
      if (convex && a * nb + b != only) continue;
      box_t A = kind_a == 0 ? brick_sub(sc, e, ia, a) : *R;
      box_t B = kind_b == 0 ? brick_sub(sc, e, ib, b) : static_sub(sc, ib, e->env_index, b);
      collide_pair(e, &A, &B, kind_a == 0 ? ia : NF + rlink, kind_b == 0 ? ib : BODY_STATIC, kind_b == 0 || b > 0, off,
                   ((unsigned)idx << 15) | ((unsigned)(a * nb + b) << 6));
    }
}

static void collide_pass(const sdx_scene_desc* sc, env_t* e, real incl) {
  const real off = sc->contact_offset;
  const int ns = sc->n_static, n1 = NF * ns, n2 = NF * (NF - 1) / 2, per = NF + ns;
  e->nc = 0;
  e->overflow = 0;
  e->incl = incl;
  box_t bb[NF];
  real br[NF];
  for (int i = 0; i < NF; ++i) {
    bb[i] = brick_bound(sc, e, i);
    br[i] = box_radius(bb[i].h);
  }
  /* (1) brick vs static */
  for (int i = 0; i < NF; ++i)
    for (int s = 0; s < ns; ++s) {
      box_t S = static_bound(sc, s, e->env_index);
      v3 g;
      if (box_sdf(vsub(bb[i].c, S.c), S.h, &g) > br[i] + off) continue;
      collide_bodies(sc, e, i * ns + s, 0, i, 2, s, NULL, 0);
    }
  /* (2) brick vs brick */
  for (int i = 0; i < NF; ++i)
    for (int j = i + 1; j < NF; ++j) {
      v3 d = vsub(bb[i].c, bb[j].c);
      real rr = br[i] + br[j] + off;
      if (vdot(d, d) > rr * rr) continue;
      collide_bodies(sc, e, n1 + (j - 1) * j / 2 + i, 0, i, 0, j, NULL, 0);
    }
  /* (3) robot box vs brick, (4) robot box vs static */
  for (int r = 0; r < sc->n_rbox; ++r) {
    int k = sc->rbox_link[r];
    if (k == 0) continue; /* the fixed base never generates contacts */
    box_t R = {e->rc[r], e->rq[r], ld3(sc->rbox_half[r])};
    real rr0 = box_radius(R.h);
    for (int i = 0; i < NF; ++i) {
      v3 d = vsub(R.c, bb[i].c);
      real rr = rr0 + br[i] + off;
      if (vdot(d, d) > rr * rr) continue;
      collide_bodies(sc, e, n1 + n2 + r * per + i, 1, r, 0, i, &R, k);
    }
    for (int s = 0; s < ns; ++s) {
      box_t S = static_bound(sc, s, e->env_index);
      v3 g;
      if (box_sdf(vsub(R.c, S.c), S.h, &g) > rr0 + off) continue;
      collide_bodies(sc, e, n1 + n2 + r * per + NF + s, 1, r, 2, s, &R, k);
    }
  }
}

/* D: the contact list of the substep.  Capacity rule: a list that would exceed SDXO_MAXC contacts is rebuilt
 * without its speculative part - only samples that touch or penetrate (inclusion threshold 0 instead of the contact offset); what
 * still does not fit is dropped in enumeration order and counted. */
static void collide(const sdx_scene_desc* sc, env_t* e) {
  collide_pass(sc, e, sc->contact_offset);
  e->rebuilt = 0;
  if (e->overflow > 0) {
    collide_pass(sc, e, 0.0f);
    e->rebuilt = 1;
  }
}

/* ---------------------------------------------------------------- E: solver */
/* orthonormal tangents of a unit normal (Duff et al. 2017: no square root, no branch on the direction); a vertical normal gives the
 * world's x and y axes */
static void tangents(v3 n, v3* t1, v3* t2) {
  real sg = n.z < 0.0f ? -1.0f : 1.0f;
  real a = -1.0f / (sg + n.z);
  real b = n.x * n.y * a;
  *t1 = V(1.0f + sg * n.x * n.x * a, sg * b, -sg * n.x);
  *t2 = V(b, sg + n.y * n.y * a, -n.y);
}

/* velocity of body `id` at world point p */
static v3 point_vel(const env_t* e, int id, v3 p) {
  if (id == BODY_STATIC) return V(0, 0, 0);
  if (id < NF) return vadd(e->bv[id], vcross(e->bw[id], vsub(p, e->bp[id])));
  int k = id - NF;
  return vadd(e->lv[k], vcross(e->lw[k], vsub(p, e->lp[k])));
}

/* robot Jacobian row for link k, point p, direction d: J[j] = (a_j x (p - p_j)) . d for dofs j on the path */
static void robot_jrow(const env_t* e, int k, v3 p, v3 d, real J[ND]) {
  for (int j = 0; j < ND; ++j)
    J[j] = ((e->anc[k] >> j) & 1u) ? vdot(vcross(e->la[j + 1], vsub(p, e->lp[j + 1])), d) : 0.0f;
}

static real brick_w(const sdx_scene_desc* sc, const env_t* e, int i, v3 p, v3 d) {
  int t = sc->brick_type[i];
  v3 rxd = vcross(vsub(p, e->bp[i]), d);
  v3 l = qrot(qconj(e->bq[i]), rxd);
  const float* I = sc->brick_inertia[t];
  real w = 1.0f / sc->brick_mass[t] + l.x * l.x / I[0] + l.y * l.y / I[1] + l.z * l.z / I[2];
  return i == e->seg_brick ? w / sc->seg_mass_scale : w;
}

static real robot_w(const env_t* e, int k, v3 p, v3 d) {
  real J[ND], s = 0;
  robot_jrow(e, k, p, d, J);
  for (int i = 0; i < ND; ++i) {
    if (J[i] == 0.0f) continue;
    real t = 0;
    for (int j = 0; j < ND; ++j) t += e->Hinv[i][j] * J[j];
    s += J[i] * t;
  }
  return s;
}

/* impulse P on side 0 (body a) and -P on side 1 (body b) of contact c: brick velocity deltas into dv / dw, robot side into dQ */
static void apply_impulse(const sdx_scene_desc* sc, env_t* e, int c, v3 P, real dQ[ND]) {
  int ids[2] = {e->ca[c], e->cb[c]};
  v3 p = e->cp[c];
  for (int s = 0; s < 2; ++s) {
    int id = ids[s];
    v3 Ps = s == 0 ? P : vscale(P, -1.0f);
    if (id == BODY_STATIC) continue;
    if (id < NF) {
      int t = sc->brick_type[id];
      real isc = id == e->seg_brick ? 1.0f / sc->seg_mass_scale : 1.0f;
      e->dv[id] = vadd(e->dv[id], vscale(Ps, isc / sc->brick_mass[t]));
      v3 l = qrot(qconj(e->bq[id]), vcross(vsub(p, e->bp[id]), Ps));
      const float* I = sc->brick_inertia[t];
      e->dw[id] = vadd(e->dw[id], vscale(qrot(e->bq[id], V(l.x / I[0], l.y / I[1], l.z / I[2])), isc));
    } else {
      int k = id - NF;
      for (int j = 0; j < ND; ++j)
        if ((e->anc[k] >> j) & 1u) dQ[j] += vdot(vcross(e->la[j + 1], vsub(p, e->lp[j + 1])), Ps);
    }
  }
}

/* the velocity update that closes an iteration (and the warm start): bricks += dv, dw; Q += dQ; qd = qd* + Hinv Q; link twists */
static void apply_deltas(const sdx_scene_desc* sc, env_t* e, const real dQ[ND]) {
  for (int i = 0; i < NF; ++i) {
    e->bv[i] = vadd(e->bv[i], e->dv[i]);
    e->bw[i] = vadd(e->bw[i], e->dw[i]);
  }
  for (int j = 0; j < ND; ++j) e->Q[j] += dQ[j];
  for (int i = 0; i < ND; ++i) {
    real s = e->qd_star[i];
    for (int j = 0; j < ND; ++j) s += e->Hinv[i][j] * e->Q[j];
    e->qd[i] = s;
  }
  for (int k = 1; k < NL; ++k) {
    int p = sc->parent[k];
    e->lw[k] = vadd(e->lw[p], vscale(e->la[k], e->qd[k - 1]));
    e->lv[k] = vadd(e->lv[p], vcross(e->lw[p], vsub(e->lp[k], e->lp[p])));
  }
}

static void solve(const sdx_scene_desc* sc, env_t* e, real h) {
  const real mu = sc->friction;
  /* base (un-split) inverse effective masses per row and side */
  for (int c = 0; c < e->nc; ++c) {
    v3 t1, t2, dir[3];
    tangents(e->cn[c], &t1, &t2);
    dir[0] = e->cn[c]; dir[1] = t1; dir[2] = t2;
    int ids[2] = {e->ca[c], e->cb[c]};
    for (int r = 0; r < 3; ++r) {
      for (int s = 0; s < 2; ++s) {
        int id = ids[s];
        real w = 0;
        if (id == BODY_STATIC) w = 0;
        else if (id < NF) w = brick_w(sc, e, id, e->cp[c], dir[r]);
        else w = robot_w(e, id - NF, e->cp[c], dir[r]);
        e->w[c][s][r] = w;
      }
      e->lam[c][r] = 0;
    }
  }
  for (int j = 0; j < ND; ++j) e->qd_star[j] = e->qd[j];
  for (int c = 0; c < e->nc; ++c) e->cage[c] = 0;
  /* warm start: a contact that existed in the previous solve - same boxes, direction and sample - starts from
   * warm_start x the impulses it ended with (normal impulse <= 0: from zero); their effect on the velocities is applied before the
   * first iteration.  The cache is in contact order, which changes little from solve to solve: the search resumes where the last
   * match was found. */
  if (sc->warm_start > 0 && e->wcount && *e->wcount > 0) {
    const int nold = *e->wcount;
    real dQ[ND];
    for (int j = 0; j < ND; ++j) dQ[j] = 0;
    for (int i = 0; i < NF; ++i) e->dv[i] = e->dw[i] = V(0, 0, 0);
    int pos = 0;
    const real inv_age = sc->warm_age > 0 ? 1.0f / sc->warm_age : 1e30f;
    for (int c = 0; c < e->nc; ++c) {
      int found = -1;
      for (int k = 0; k < nold; ++k) {
        int q = pos + k;
        if (q >= nold) q -= nold;
        if ((e->wkey[q] & 0x0fffffffu) == e->ckey[c]) { found = q; break; }
      }
      if (found < 0) continue;
      pos = found;
      /* bits 28..31 of a cached key: the number of consecutive solves the contact had existed before that solve (saturating at 15: the
       * ramp is over after warm_age <= 16 solves) */
      unsigned age = (e->wkey[found] >> 28) + 1u;
      e->cage[c] = (unsigned char)age;
      if (e->csep[c] < -WARM_DEPTH * sc->contact_offset) continue; /* deep penetration is recovery, not rest: cold */
      {
        v3 vr = vsub(point_vel(e, e->ca[c], e->cp[c]), point_vel(e, e->cb[c], e->cp[c]));
        if (vdot(vr, vr) > WARM_SPEED * WARM_SPEED) continue; /* an impact or a sliding contact: last solve's impulse says nothing */
      }
      /* the cached impulse is trusted in proportion to the age of the contact: the full fraction after warm_age solves, nothing for
       * a contact the previous solve saw for the first time (an impact) */
      const real bq = sc->warm_start * fminf(1.0f, (real)age * inv_age);
      real l0 = bq * e->wlam[0 * SDXO_MAXC + found];
      if (!(l0 > 0)) continue;
      e->lam[c][0] = l0;
      e->lam[c][1] = bq * e->wlam[1 * SDXO_MAXC + found];
      e->lam[c][2] = bq * e->wlam[2 * SDXO_MAXC + found];
      v3 t1, t2;
      tangents(e->cn[c], &t1, &t2);
      apply_impulse(sc, e, c, vadd(vadd(vscale(e->cn[c], e->lam[c][0]), vscale(t1, e->lam[c][1])), vscale(t2, e->lam[c][2])), dQ);
    }
    apply_deltas(sc, e, dQ);
  }
  /* Mass splitting: a row's inverse effective mass is n_A w_A + n_B w_B, n_X = the number of contacts that were
   * ACTIVE on body X in the PREVIOUS iteration (at least 1); the first iteration splits over every contact of the body.  (Rounds 1-2
   * counted the active set of the same iteration, which costs the kernel a workgroup barrier between counting and updating; stacks,
   * piles and the drop test behave the same with the lagged count.) */
  int nb_prev[NF], nr_prev = 0;
  for (int i = 0; i < NF; ++i) nb_prev[i] = 0;
  for (int c = 0; c < e->nc; ++c) {
    int a = e->ca[c], b = e->cb[c];
    if (a < NF) nb_prev[a]++; else if (a != BODY_STATIC) nr_prev++;
    if (b < NF) nb_prev[b]++; else if (b != BODY_STATIC) nr_prev++;
  }
  for (int it = 0; it < sc->solver_iters; ++it) {
    /* pass 1: active set and per-body active-contact counts (they split the masses of the NEXT iteration) */
    for (int i = 0; i < NF; ++i) e->bcount[i] = 0;
    e->rcount = 0;
    for (int c = 0; c < e->nc; ++c) {
      int a = e->ca[c], b = e->cb[c];
      v3 p = e->cp[c], n = e->cn[c];
      v3 vr = vsub(point_vel(e, a, p), point_vel(e, b, p));
      real sep = e->csep[c];
      real target = sep > 0 ? -sep / h : fminf(sc->baumgarte * (-sep) / h, sc->max_depenetration_vel);
      int act = (e->lam[c][0] > 0) || (vdot(vr, n) < target);
      e->active[c] = (unsigned char)act;
      if (!act) continue;
      if (a < NF) e->bcount[a]++; else if (a != BODY_STATIC) e->rcount++;
      if (b < NF) e->bcount[b]++; else if (b != BODY_STATIC) e->rcount++;
    }
    for (int i = 0; i < NF; ++i) e->dv[i] = e->dw[i] = V(0, 0, 0);
    real dQ[ND];
    for (int j = 0; j < ND; ++j) dQ[j] = 0;
    /* pass 2: Jacobi update of the accumulated impulses from the same velocity snapshot */
    for (int c = 0; c < e->nc; ++c) {
      if (!e->active[c]) continue;
      int a = e->ca[c], b = e->cb[c];
      v3 p = e->cp[c], n = e->cn[c], t1, t2;
      tangents(n, &t1, &t2);
      int ia = a == BODY_STATIC ? 0 : (a < NF ? nb_prev[a] : nr_prev), ib = b == BODY_STATIC ? 0 : (b < NF ? nb_prev[b] : nr_prev);
      real na = a == BODY_STATIC ? 0.0f : (real)(ia > 1 ? ia : 1);
      real nb = b == BODY_STATIC ? 0.0f : (real)(ib > 1 ? ib : 1);
      v3 vr = vsub(point_vel(e, a, p), point_vel(e, b, p));
      real sep = e->csep[c];
      real target = sep > 0 ? -sep / h : fminf(sc->baumgarte * (-sep) / h, sc->max_depenetration_vel);
      real dl[3];
      real w0 = na * e->w[c][0][0] + nb * e->w[c][1][0];
      real w1 = na * e->w[c][0][1] + nb * e->w[c][1][1];
      real w2 = na * e->w[c][0][2] + nb * e->w[c][1][2];
      real ln = fmaxf(0.0f, e->lam[c][0] - sc->jacobi_relax * (vdot(vr, n) - target) / w0);
      dl[0] = ln - e->lam[c][0];
      e->lam[c][0] = ln;
      real lim = mu * ln;
      real l1 = e->lam[c][1] - sc->jacobi_relax * vdot(vr, t1) / w1;
      l1 = fminf(lim, fmaxf(-lim, l1));
      dl[1] = l1 - e->lam[c][1];
      e->lam[c][1] = l1;
      real l2 = e->lam[c][2] - sc->jacobi_relax * vdot(vr, t2) / w2;
      l2 = fminf(lim, fmaxf(-lim, l2));
      dl[2] = l2 - e->lam[c][2];
      e->lam[c][2] = l2;
      apply_impulse(sc, e, c, vadd(vadd(vscale(n, dl[0]), vscale(t1, dl[1])), vscale(t2, dl[2])), dQ); /* impulse on A; -P on B */
    }
    apply_deltas(sc, e, dQ);
    for (int i = 0; i < NF; ++i) nb_prev[i] = e->bcount[i];
    nr_prev = e->rcount;
  }
  if (e->wcount && sc->warm_start > 0) { /* the cache for the next solve */
    *e->wcount = e->nc;
    for (int c = 0; c < e->nc; ++c) {
      e->wkey[c] = e->ckey[c] | ((unsigned)(e->cage[c] > 15 ? 15 : e->cage[c]) << 28);
      for (int r = 0; r < 3; ++r) e->wlam[r * SDXO_MAXC + c] = e->lam[c][r];
    }
  }
}

/* ---------------------------------------------------------------- one env, one step */
static void load_env(const sdx_scene_desc* sc, env_t* e, int env_index, const float* root, const float* dof, const float* targets) {
  e->env_index = env_index;
  { int b = env_index & 7; e->seg_brick = (b == 3 || b == 4 || b == 7) ? 0 : b; } /* GS:962-965,974-975 */
  for (int j = 0; j < ND; ++j) {
    e->q[j] = dof[2 * j];
    e->qd[j] = dof[2 * j + 1];
    e->tgt[j] = targets[j];
  }
  for (int i = 0; i < NF; ++i) {
    const float* s = root + (SDX_ACTOR_BRICK0 + i) * 13;
    e->bq[i] = qnormalize(ld4(s + 3));
    e->bp[i] = vadd(ld3(s), qrot(e->bq[i], ld3(sc->brick_com[sc->brick_type[i]])));
    e->bv[i] = ld3(s + 7); /* COM velocity = origin velocity + w x (R c); the stored root velocity is the COM's */
    e->bw[i] = ld3(s + 10);
  }
}

static void substep(const sdx_scene_desc* sc, env_t* e, real h, int first) {
  fk(sc, e);
  if (first) mass_matrix(sc, e, h); /* M(q) is evaluated once per step and frozen over the substeps */
  real tau[ND], tauc[ND];
  coriolis(sc, e, tauc);
  for (int j = 0; j < ND; ++j) {
    real t = sc->kp[j] * (e->tgt[j] - e->q[j]) - (sc->kd[j] + h * sc->kp[j]) * e->qd[j];
    tau[j] = fminf(sc->effort[j], fmaxf(-sc->effort[j], t)) - tauc[j];   /* the effort limit applies to the drive only */
  }
  for (int i = 0; i < ND; ++i) {
    real s = 0;
    for (int j = 0; j < ND; ++j) s += e->Hinv[i][j] * tau[j];
    e->qd[i] += h * s;
  }
  for (int j = 0; j < ND; ++j) e->Q[j] = 0;
  for (int k = 1; k < NL; ++k) { /* twists with the new qd */
    int p = sc->parent[k];
    e->lw[k] = vadd(e->lw[p], vscale(e->la[k], e->qd[k - 1]));
    e->lv[k] = vadd(e->lv[p], vcross(e->lw[p], vsub(e->lp[k], e->lp[p])));
  }
  v3 g = ld3(sc->gravity);
  for (int i = 0; i < NF; ++i) e->bv[i] = vadd(e->bv[i], vscale(g, h));
  collide(sc, e);
  solve(sc, e, h);
  for (int j = 0; j < ND; ++j) {
    real v = e->qd[j] * (1.0f - h * sc->robot_angular_damping); /* GS:546: asset_options.angular_damping = 0.01 */
    v = fminf(sc->vel_limit[j], fmaxf(-sc->vel_limit[j], v));
    real qn = e->q[j] + h * v;
    if (qn < sc->lower[j]) { qn = sc->lower[j]; v = fmaxf(v, 0.0f); }
    if (qn > sc->upper[j]) { qn = sc->upper[j]; v = fminf(v, 0.0f); }
    e->q[j] = qn;
    e->qd[j] = v;
  }
  for (int i = 0; i < NF; ++i) {
    e->bp[i] = vadd(e->bp[i], vscale(e->bv[i], h));
    q4 wq = {e->bw[i].x, e->bw[i].y, e->bw[i].z, 0};
    q4 dq = qmul(wq, e->bq[i]);
    q4 nq = {e->bq[i].x + 0.5f * h * dq.x, e->bq[i].y + 0.5f * h * dq.y, e->bq[i].z + 0.5f * h * dq.z,
             e->bq[i].w + 0.5f * h * dq.w};
    e->bq[i] = qnormalize(nq);
  }
}

static void store_env(const sdx_scene_desc* sc, env_t* e, real h, float* root, float* dof, float* rb, float* contact,
                      float* jac, int* ncontacts) {
  fk(sc, e);
  for (int j = 0; j < ND; ++j) {
    dof[2 * j] = e->q[j];
    dof[2 * j + 1] = e->qd[j];
  }
  for (int k = 0; k < NL; ++k) {
    float* s = rb + k * 13;
    st3(s, e->lp[k]);
    st4(s + 3, e->lq[k]);
    st3(s + 7, e->lv[k]);
    st3(s + 10, e->lw[k]);
  }
  for (int i = 0; i < NF; ++i) {
    float* s = root + (SDX_ACTOR_BRICK0 + i) * 13;
    st3(s, vsub(e->bp[i], qrot(e->bq[i], ld3(sc->brick_com[sc->brick_type[i]]))));
    st4(s + 3, e->bq[i]);
    st3(s + 7, e->bv[i]);
    st3(s + 10, e->bw[i]);
    memcpy(rb + (SDX_BODY_BRICK0 + i) * 13, s, 13 * sizeof(float));
  }
  /* end-effector Jacobian of body hand_base_body at its frame origin, arm dofs 0..6 (GS:1601) */
  int ee = sc->hand_base_body;
  for (int j = 0; j < 7; ++j) {
    v3 lin = vcross(e->la[j + 1], vsub(e->lp[ee], e->lp[j + 1]));
    v3 ang = e->la[j + 1];
    jac[0 * 7 + j] = lin.x; jac[1 * 7 + j] = lin.y; jac[2 * 7 + j] = lin.z;
    jac[3 * 7 + j] = ang.x; jac[4 * 7 + j] = ang.y; jac[5 * 7 + j] = ang.z;
  }
  /* net contact force on robot bodies from the last substep's impulses (GS:1094,1159-1162 read bodies 1..6) */
  for (int k = 0; k < NL; ++k) contact[k * 3] = contact[k * 3 + 1] = contact[k * 3 + 2] = 0;
  for (int c = 0; c < e->nc; ++c) {
    v3 t1, t2;
    tangents(e->cn[c], &t1, &t2);
    v3 P = vadd(vadd(vscale(e->cn[c], e->lam[c][0]), vscale(t1, e->lam[c][1])), vscale(t2, e->lam[c][2]));
    P = vscale(P, 1.0f / h);
    int a = e->ca[c], b = e->cb[c];
    if (a >= NF && a != BODY_STATIC) { float* f = contact + (a - NF) * 3; f[0] += P.x; f[1] += P.y; f[2] += P.z; }
    if (b >= NF && b != BODY_STATIC) { float* f = contact + (b - NF) * 3; f[0] -= P.x; f[1] -= P.y; f[2] -= P.z; }
  }
  if (ncontacts) *ncontacts = e->nc;
}

/* ---------------------------------------------------------------- exported entry points (ctypes) */

/* gym.simulate + refresh_* for N envs.  root [N,142,13], dof [N,23,2], targets [N,23], rb [N,165,13],
 * contact [N,165,3], jac [N,6,7], ncontacts [N] (may be NULL).
 * Warm-start cache of the envs (all three NULL: every call starts from an empty cache, as a freshly created simulator does; the second
 * substep still starts from the first one's impulses): wcount i32 [N], wkey u32 [N, MAXC], wlam f32 [N, 3, MAXC]; read and updated. */
void sdxo_simulate(const sdx_scene_desc* sc, int N, float* root, float* dof, const float* targets, float* rb,
                   float* contact, float* jac, int* ncontacts, int* wcount, unsigned* wkey, float* wlam) {
  real h = sc->dt / (real)sc->substeps;
  /* envs are independent (collision groups = env id, GS:907-968): OpenMP over envs; thread count = OMP_NUM_THREADS */
#pragma omp parallel
  {
    env_t* e = (env_t*)malloc(sizeof(env_t));
    int tcount = 0;
    unsigned* tkey = wcount ? NULL : (unsigned*)malloc(sizeof(unsigned) * SDXO_MAXC);
    float* tlam = wcount ? NULL : (float*)malloc(sizeof(float) * 3 * SDXO_MAXC);
#pragma omp for schedule(dynamic, 4)
    for (int n = 0; n < N; ++n) {
      float* r = root + (size_t)n * SDX_ACTORS * 13;
      float* d = dof + (size_t)n * ND * 2;
      e->overflow = 0;
      if (wcount) { e->wcount = wcount + n; e->wkey = wkey + (size_t)n * SDXO_MAXC; e->wlam = wlam + (size_t)n * 3 * SDXO_MAXC; }
      else { tcount = 0; e->wcount = &tcount; e->wkey = tkey; e->wlam = tlam; }
      load_env(sc, e, n, r, d, targets + (size_t)n * ND);
      for (int s = 0; s < sc->substeps; ++s) substep(sc, e, h, s == 0);
      store_env(sc, e, h, r, d, rb + (size_t)n * SDX_BODIES * 13, contact + (size_t)n * SDX_BODIES * 3,
                jac + (size_t)n * 42, ncontacts ? ncontacts + n : NULL);
    }
    free(e);
    free(tkey);
    free(tlam);
  }
}

/* kinematics only (sdx_refresh_kinematics): rb links + jac from dof */
void sdxo_kinematics(const sdx_scene_desc* sc, int N, const float* dof, float* rb, float* jac) {
  env_t* e = (env_t*)calloc(1, sizeof(env_t));
  for (int n = 0; n < N; ++n) {
    const float* d = dof + (size_t)n * ND * 2;
    for (int j = 0; j < ND; ++j) { e->q[j] = d[2 * j]; e->qd[j] = d[2 * j + 1]; }
    fk(sc, e);
    float* r = rb + (size_t)n * SDX_BODIES * 13;
    for (int k = 0; k < NL; ++k) {
      st3(r + k * 13, e->lp[k]); st4(r + k * 13 + 3, e->lq[k]); st3(r + k * 13 + 7, e->lv[k]); st3(r + k * 13 + 10, e->lw[k]);
    }
    float* J = jac + (size_t)n * 42;
    int ee = sc->hand_base_body;
    for (int j = 0; j < 7; ++j) {
      v3 lin = vcross(e->la[j + 1], vsub(e->lp[ee], e->lp[j + 1]));
      J[j] = lin.x; J[7 + j] = lin.y; J[14 + j] = lin.z;
      J[21 + j] = e->la[j + 1].x; J[28 + j] = e->la[j + 1].y; J[35 + j] = e->la[j + 1].z;
    }
  }
  free(e);
}

/* debug: joint-space inertia H (with implicit PD terms for substep h) of one configuration, [23,23] */
void sdxo_mass_matrix(const sdx_scene_desc* sc, const float* q, float h, float* H_out, float* Hinv_out) {
  env_t* e = (env_t*)calloc(1, sizeof(env_t));
  for (int j = 0; j < ND; ++j) e->q[j] = q[j];
  fk(sc, e);
  mass_matrix(sc, e, h);
  memcpy(H_out, e->H, sizeof(e->H));
  memcpy(Hinv_out, e->Hinv, sizeof(e->Hinv));
  free(e);
}

/* debug: contact list of one env after loading its state (no stepping): returns count; out [MAXC,9] =
 * (a, b, px, py, pz, nx, ny, nz, sep) */
int sdxo_contacts(const sdx_scene_desc* sc, const float* root_env, const float* dof_env, float* out, int cap) {
  env_t* e = (env_t*)calloc(1, sizeof(env_t));
  float tg[ND] = {0};
  load_env(sc, e, 0, root_env, dof_env, tg);
  fk(sc, e);
  collide(sc, e);
  int n = e->nc < cap ? e->nc : cap;
  for (int c = 0; c < n; ++c) {
    float* o = out + c * 9;
    o[0] = e->ca[c]; o[1] = e->cb[c];
    o[2] = e->cp[c].x; o[3] = e->cp[c].y; o[4] = e->cp[c].z;
    o[5] = e->cn[c].x; o[6] = e->cn[c].y; o[7] = e->cn[c].z;
    o[8] = e->csep[c];
  }
  int total = e->nc + e->overflow;
  free(e);
  return total;
}

/* debug: identity keys of the contact list sdxo_contacts returns for the same state (enumeration index of the body pair << 15 | box pair
 * inside it << 6 | direction << 5 | sample) */
int sdxo_contact_keys(const sdx_scene_desc* sc, const float* root_env, const float* dof_env, unsigned* out, int cap) {
  env_t* e = (env_t*)calloc(1, sizeof(env_t));
  float tg[ND] = {0};
  load_env(sc, e, 0, root_env, dof_env, tg);
  fk(sc, e);
  collide(sc, e);
  int n = e->nc < cap ? e->nc : cap;
  for (int c = 0; c < n; ++c) out[c] = e->ckey[c];
  free(e);
  return n;
}

int sdxo_max_contacts(void) { return SDXO_MAXC; }
