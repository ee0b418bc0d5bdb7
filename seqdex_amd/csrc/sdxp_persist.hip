// sdxp_persist.hip — the PPO update phase of one epoch (all mini-epochs x minibatches, three networks) as ONE persistent
// kernel on MI355X: 256 workgroups (one per CU) x 512 threads.  Every weight and its Adam moments stay in the VGPR file of the
// CU that owns them for the whole epoch (~132 resident registers per lane of 241), activations / gradient factors of the
// current minibatch live in LDS (147 KiB), and the only inter-CU traffic per optimiser step is the rank-MB factors, exchanged
// as (value, step tag) words - no grid barrier (replaces PS:294-326 / RC:1339-1365 of the reference loop):
//
//   A  dY0 -> Gram -> |g|^2 -> clip scale, Adam of layer 0, forward L0 of own rows                       -> x1  | shadow: Adam L1
//   B  gather x1, forward L1 of own rows                                                                 -> x2  | shadow: Gram x1, Adam L2/heads/biases
//   C  gather x2, forward L2 of own row                                                                  -> x3  | shadow: Gram x2
//   D  gather x3 + heads, PPO losses (replicated on every CU, bit-identical), backward heads and L2      -> dY1 | shadow: stats, LR rule, Gram x3, head norms
//   E  gather dY1, backward L1 (column copy)                                                             -> dY0 | shadow: Grams dY1, dY2
//
// Ownership (shipped network 396/564 -> 1024 -> 512 -> 256 -> 23/1, minibatch 4):
//   W0 rows      : CU g, wave w owns half (w&1) of row 4g+(w>>1) of actor, critic and central value
//   W1 rows      : wave w owns quarter (w&3) of row 2g+(w>>2) of the three nets     (forward)
//   W1 columns   : CU g owns columns 4g..4g+3 of the three nets, thread = row       (backward; duplicate copy, same Adam)
//   W2 rows      : wave w owns eighth w of row g of the three nets                  (forward)
//   W2 columns   : CU g owns columns 2g, 2g+1: thread = (column, row)               (backward; duplicate copy)
//   heads        : CU g runs Adam for 25 of the 6 400 head weights and publishes them; every CU reads the matrix back in phase D
// Row and column copies receive the same gradient g[n][k] = sum_s dY_s[n] X_s[k] (same operands, same order), so they
// stay bit-identical without communication.  The multi-kernel path (sdxp_kernels.hip) remains the fallback for other shapes
// and the explicit-gradient multi-rank path.
#include <cstddef>
#include <cstdlib>

#include "sdx_common.h"
#include "sdxp_types.h"

namespace {
// OBS is the LARGEST actor/critic input width (GraspSim, 396); narrower inputs (Orient: 188) run with the tail columns masked off
constexpr int MB = 4, OBS = 396, ST = 564, U0 = 1024, U1 = 512, U2 = 256, ACT = 23;
constexpr int NWG = 256, NTH = 512, NWV = NTH / 64;
constexpr int H0A = OBS / 2, H0V = ST / 2;   // a layer-0 row is split over two waves: 198 / 282 elements each
constexpr int I0A = (H0A + 63) / 64;   // 4 elements per lane of half an actor/critic layer-0 row
constexpr int I0V = (H0V + 63) / 64;   // 5 for half a central-value layer-0 row
constexpr int HR = 4;                  // head rows per wave (25 rows over 8 waves)
constexpr int HPC = (ACT + 2) * U2 / NWG;   // head weights whose Adam state one CU owns (25)
static_assert(HPC * NWG == (ACT + 2) * U2 && U2 == 256, "head ownership split");

__device__ __forceinline__ float elu(float x) { return x > 0.0f ? x : expm1f(x); }
__device__ __forceinline__ float elu_g(float h) { return h > 0.0f ? 1.0f : h + 1.0f; }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
  return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xB1, 0xF>(v); v = dpp_add<0x4E, 0xF>(v); v = dpp_add<0x141, 0xF>(v); v = dpp_add<0x140, 0xF>(v);
  v = dpp_add<0x142, 0xA>(v); v = dpp_add<0x143, 0xC>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Opaque copy of an index.  volatile asm statements keep their program order relative to each other and to sched_barrier,
// and an LDS load whose address depends on the result cannot be floated above it: this is what keeps the operand loads of
// Adam element i+1 below the arithmetic of element i.
// Passing the weight updated by element i as a (fake) input orders element i+1's loads after element i's arithmetic too.
__device__ __forceinline__ int opaque(int x, float after) { asm volatile("" : "+v"(x) : "v"(after)); return x; }
__device__ __forceinline__ void adam1(float& w, float g, float& m, float& v, float lr_bc1, float isq_bc2) {
  m = 0.9f * m + 0.1f * g;
  v = 0.999f * v + 0.001f * g * g;
  // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the IEEE sequences: the Adam arithmetic of ~45 elements per lane is VALU bound
  w -= lr_bc1 * m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) * isq_bc2 + 1e-8f);
  // one element at a time: without this the scheduler forms all ~45 gradients of the lane first and holds them (and the
  // half-updated moments) in registers while it interleaves the long sqrt/div chains
  __builtin_amdgcn_sched_barrier(0);
}

struct PLds {
  float obs[MB][OBS];        // layer-0 input of actor/critic (dataset rows)
  float cvx[MB][ST];         // layer-0 input of the central value (pre-normalised rows)
  float x1[3][MB][U0];
  float x2[3][MB][U1];
  float x3[3][MB][U2];
  float dy2[3][MB][U2];
  float dy1[3][MB][U1];
  // Gram matrices (MB x MB) of the factors of the minibatch currently held: inputs gx[net][layer 0..3], gradients gd[net][layer 0..2]
  // (round 6: the gradient factors' Grams never leave the reduction that forms them - layer 0's meets G_x0 on the producing CU, layers 1 / 2
  // meet theirs at the end of the dY0 shadow - so only the inputs' Grams and one partial squared norm per network are kept)
  float gx[3][4][16];
  float gd2[3][16];          // Gram of dY2 (formed in the dY1 shadow beside the Gram of x3, used at the end of the dY0 shadow)
  float n2rest[4];           // per network: sum over trunk layers 1, 2 of <G_dY, G_x> + their bias terms + the heads' terms (n2h): everything but layer 0
  float part[64 * 4];        // block_sum results
  float fpart[3 * MB * NWV];  // cross-wave partial dot products of the forward passes
  float red[4 * NWV][64];    // block_sum: one row per (wave, DPP row)
  float mu[MB][32], dmu[MB][32], z[MB][32], act[MB][32], omu[MB][32], osg[MB][32];
  float val[2][MB], dv[2][MB], gnlp[MB], ls[32], sgm[32], dls[32], stat[MB][8];   // sgm = exp(ls): formed with ls in the x3 shadow (round 6), not three times in the loss phase
  // biases of the owned rows and their Adam moments, one thread per slot: [0..11] L0 (net*4 + row), [12..17] L1 (12 + net*2 + row),
  // [18..20] L2 (18 + net), [21..45] heads (21 + row, replicated on every CU); logstd lives in the static bank s_b2
  float bias[64], bias_m[64], bias_v[64];
  float dyown[3][MB][8];     // dY of the owned rows: [net][s][0..3] L0 rows (per wave), [4..5] L1 rows, [6] L2 row
  float scal[16];            // broadcast scalars of the step: 0 ac_gscale 1 cv_gscale 2 ac_lr_bc1 3 ac_isq 4 cv_lr_bc1 5 cv_isq
  // control state machine, owned by thread 0 of every CU (every CU runs it on identical inputs, so the copies stay identical)
  struct Ctl {
    double ac_b1, ac_b2, cv_b1, cv_b2;
    float ac_lr, ac_lr_applied, cv_lr, sum_a, sum_c, sum_b, sum_kl, sum_cv, sum_ent, last_kl, ac_gn, cv_gn;
    int ac_t, cv_t;
  } ctl;
  int fail;
  long long tacc[32], tlast;   // phase clock accumulators of CU 0 (SDXP_PERSIST_STAMPS=1)
};

// ---- (value, tag) exchange words ("LL" protocol): one relaxed 8-byte store at agent scope carries the float and the tag of the
// optimiser step that produced it; a consumer re-reads until the tag it expects shows up.  No fences, no barriers, no reset:
// tags only grow, and the data dependencies of the step (x1 -> x2 -> x3 -> dY1 -> dY0 -> next x1) guarantee that a word is not
// overwritten before every CU has consumed it ("persistent update kernel").
typedef unsigned long long u64;
// Round 4: the five edges on the step's dependency chain (x1, x2, x3, dY1 and the dY0 Gram) travel as 16-BYTE words (v0, v1, v2, tag):
// the same (sample, unit) of the THREE networks in one word, written and read by one dwordx4 access per lane (device scope, sc1).  The
// all-gather alone, measured (tools/bench_exchange.py, profiles/r4_exchange_edge_floor.txt): x1 (12 288 floats) 6.0 us as 8-byte
// words, 3.6 us packed; x2 / dY1 3.3 -> 2.1; the Gram 4.7 -> 2.5; x3 (3 072 floats) 2.0 either way; no torn word in ~1e10 checked
// gathers (a 16-byte access of one lane is not architecturally single-copy atomic: the tag sits in the LAST dword, and the benchmark
// verifies every payload against its tag; RCCL's LL128 protocol relies on the same hardware behaviour).  Since round 5 the last dword
// carries tag ^ fold(payload) (lq_fold below): a torn word is rejected and polled again instead of being trusted to never occur.  The head matrix (LL_HW:
// published a phase ahead, gathered row- and column-wise by different consumers) keeps its 8-byte words.
// Layout of D.ll: 16-byte words LQ_* first, then the 8-byte head words at LL_HW (u64 index).
constexpr unsigned LQ_X1 = 0, LQ_X2 = LQ_X1 + MB * U0, LQ_X3 = LQ_X2 + MB * U1, LQ_DY1 = LQ_X3 + MB * U2, LQ_DY0 = LQ_DY1 + MB * U1,
                   LQ_G0 = LQ_DY0 + MB * U0, LQ_END = LQ_G0 + 16 * NWG;   // LQ_G0: [CU]: the CU's share of layer 0's squared gradient norm, the 3 nets per word (round 6; rounds 4-5: 10 Gram entries per CU)
constexpr size_t LL_HW = 2 * (size_t)LQ_END, LL_END = LL_HW + (ACT + 2) * U2;
static_assert(LL_END <= SDXP_LL_WORDS, "exchange buffer too small");
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lq_rsrc(void* ll) { return __builtin_amdgcn_make_buffer_rsrc(ll, 0, SDXP_LL_WORDS * 8, 0x00020000); }
// Tearing is DETECTABLE: the last dword is not the bare tag but tag ^ fold(payload); a consumer accepts a word only when
// last ^ fold(the payload it read) equals the tag it waits for.  A word whose dwords come from two different stores (new tag over a
// stale payload or the reverse) fails the test unless the mixed payload folds to the same value (2^-32, or the stale dwords equal the new
// ones - then nothing is lost) and is simply polled again, like a word that has not arrived.
// (fold = x ^ y ^ z: one v_xor3_b32.  With rotations of y and z in it - to decorrelate equal values of two networks - the checks cost the
// step 0.6 us on its dependent chain, 31.4 instead of 30.8; two networks' values that are equal before AND after a step are unchanged ones)
// Tags advance by 1 per step, so the XOR of two consecutive bare tags is 1, 3, 7, ...: a torn word (new payload over a stale last dword) whose
// payload fold moved by exactly that - one network's value changing by an ulp - would pass.  The tag is therefore spread over all
// 32 bits before it meets the fold (lq_tag: one scalar multiply per gather, off the dependent chain): consecutive tags now differ in ~16
// pseudo-random bit positions, which a slowly varying payload does not reproduce.
__device__ __forceinline__ unsigned lq_fold(unsigned x, unsigned y, unsigned z) { return x ^ y ^ z; }
__device__ __forceinline__ unsigned lq_tag(unsigned tag) { return tag * 0x9E3779B1u; }
__device__ __forceinline__ bool lq_ok(const u32x4& w, unsigned tag) { return (w.w ^ lq_fold(w.x, w.y, w.z)) == lq_tag(tag); }
__device__ __forceinline__ void lq_store(__amdgpu_buffer_rsrc_t q, unsigned idx, float a, float b, float c, unsigned tag) {
  u32x4 w;
  w.x = __float_as_uint(a); w.y = __float_as_uint(b); w.z = __float_as_uint(c); w.w = lq_tag(tag) ^ lq_fold(w.x, w.y, w.z);
  __builtin_amdgcn_raw_buffer_store_b128(w, q, idx * 16u, 0, 16);   // buffer_store_dwordx4 ... sc1
}
// gather N packed words idx0 + i * stride carrying `tag` -> the three networks' values; wave-uniform retry as ll_gather
template <int N>
__device__ __forceinline__ bool lq_gather(__amdgpu_buffer_rsrc_t q, unsigned idx0, unsigned stride, unsigned tag, float (&o0)[N], float (&o1)[N], float (&o2)[N],
                                          unsigned* failflag) {
  unsigned spins = 0;
  for (;;) {
    u32x4 w[N];
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = __builtin_amdgcn_raw_buffer_load_b128(q, (idx0 + i * stride) * 16u, 0, 16);   // buffer_load_dwordx4 ... sc1
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) ok = ok && lq_ok(w[i], tag);
#pragma unroll
    for (int i = 0; i < N; ++i) { o0[i] = __uint_as_float(w[i].x); o1[i] = __uint_as_float(w[i].y); o2[i] = __uint_as_float(w[i].z); }
    if (__builtin_amdgcn_ballot_w64(!ok) == 0) return true;
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 255u) == 0) {
      const unsigned f = __hip_atomic_load(failflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (spins > (1u << 20) || __builtin_amdgcn_readfirstlane(f) != 0) {
        __hip_atomic_store(failflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
}
// first look at N packed words, issued some work ahead of where they are needed (lq_peek), and the test whether all of them already
// carried `tag` (lq_take): when a shadow phase outlasts its exchange edge the words are there, and the round trip of the gather
// (about 0.6 us) runs under the shadow's last piece of work instead of after it.  A miss falls back to the polling lq_gather.
template <int N>
__device__ __forceinline__ void lq_peek(__amdgpu_buffer_rsrc_t q, unsigned idx0, unsigned stride, u32x4 (&w)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) w[i] = __builtin_amdgcn_raw_buffer_load_b128(q, (idx0 + i * stride) * 16u, 0, 16);
}
template <int N>
__device__ __forceinline__ bool lq_take(const u32x4 (&w)[N], unsigned tag, float (&o0)[N], float (&o1)[N], float (&o2)[N]) {
  bool ok = true;
#pragma unroll
  for (int i = 0; i < N; ++i) ok = ok && lq_ok(w[i], tag);
#pragma unroll
  for (int i = 0; i < N; ++i) { o0[i] = __uint_as_float(w[i].x); o1[i] = __uint_as_float(w[i].y); o2[i] = __uint_as_float(w[i].z); }
  return __builtin_amdgcn_ballot_w64(!ok) == 0;
}
__device__ __forceinline__ void ll_store(u64* p, float v, unsigned tag) {
  __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// gather N words p[i * stride] carrying `tag`; false (and *failflag set) if they do not arrive.  The retry loop is wave-uniform
// (all lanes reload until every lane has its words), which keeps the loaded values out of divergent-loop phis.
template <int N>
__device__ __forceinline__ bool ll_gather(const u64* p, size_t stride, unsigned tag, float (&out)[N], unsigned* failflag) {
  unsigned spins = 0;
  for (;;) {
    u64 w[N];
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = __hip_atomic_load(p + i * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < N; ++i) ok = ok && (unsigned)(w[i] >> 32) == tag;
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = __uint_as_float((unsigned)w[i]);
    if (__builtin_amdgcn_ballot_w64(!ok) == 0) return true;
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 255u) == 0) {
      const unsigned f = __hip_atomic_load(failflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (spins > (1u << 20) || __builtin_amdgcn_readfirstlane(f) != 0) {
        __hip_atomic_store(failflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
}
// the same without the scheduling fence (elements whose gradients already sit in registers: the sqrt / rcp chains of neighbours may interleave)
__device__ __forceinline__ void adam1f(float& w, float g, float& m, float& v, float lr_bc1, float isq_bc2) {
  m = 0.9f * m + 0.1f * g;
  v = 0.999f * v + 0.001f * g * g;
  w -= lr_bc1 * m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) * isq_bc2 + 1e-8f);
}
#define SDX_PIN4X(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define SDX_PIN3X(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
// layer 0 (the only Adam on the step's dependent chain): the moments arrive PRE-SCALED by 0.9 / 0.999 - done in the dY0 shadow, where the
// element's gradient is formed - so that behind the clip scale an element is 7 VALU + sqrt + rcp instead of 11 (round 6)
__device__ __forceinline__ void adam1p(float& w, float g, float& m9, float& v999, float lr_bc1, float isq_bc2) {
  m9 = __builtin_fmaf(0.1f, g, m9);
  v999 = __builtin_fmaf(0.001f * g, g, v999);
  w -= lr_bc1 * m9 * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v999) * isq_bc2 + 1e-8f);
  __builtin_amdgcn_sched_barrier(0);
}
#define ADAM0 adam1p
// accumulate the 4x4 Gram (lower triangle, 10 terms) and |sum|^2 of (a, b, c, d)
__device__ __forceinline__ void gram_acc(float* p, float a, float b, float c, float d) {
  p[0] += a * a; p[1] += b * a; p[2] += b * b; p[3] += c * a; p[4] += c * b; p[5] += c * c;
  p[6] += d * a; p[7] += d * b; p[8] += d * c; p[9] += d * d;
  const float sb = a + b + c + d;
  p[10] += sb * sb;
}
__device__ __forceinline__ void gram_acc10(float* p, float a, float b, float c, float d) {   // the 10 entries only (inputs: no bias term)
  p[0] += a * a; p[1] += b * a; p[2] += b * b; p[3] += c * a; p[4] += c * b; p[5] += c * c;
  p[6] += d * a; p[7] += d * b; p[8] += d * c; p[9] += d * d;
}
__device__ __forceinline__ int tri16(int i) { const int a = i >> 2, b = i & 3, hi = a > b ? a : b, lo = a > b ? b : a; return hi * (hi + 1) / 2 + lo; }
// sums over each 32-lane half of the wave; valid in lanes 16..31 and 48..63
__device__ __forceinline__ float half_sum(float v) {
  v = dpp_add<0xB1, 0xF>(v); v = dpp_add<0x4E, 0xF>(v); v = dpp_add<0x141, 0xF>(v); v = dpp_add<0x140, 0xF>(v);
  return dpp_add<0x142, 0xA>(v);
}

// block-wide sums of N per-thread values (N <= 64); result in out[0..N).  Two "halving" butterfly levels (lane pairs keep the even /
// odd half of the values, so level L costs N / 2^L exchanges instead of N), two more within the 16-lane DPP row, and the 4 rows x 8
// waves meet in LDS: ~3.5 N VALU ops + N/4 LDS writes per wave instead of 7 N + N masked writes for N independent wave sums.
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_mov(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xF, BANK, false));
}
// the two halving levels + the two levels inside the 16-lane DPP row: u2[j] = sum over the row's lanes of value 4 j + (lane & 3)
template <int N>
__device__ __forceinline__ void row_butterfly(const float (&v)[N], float (&u2)[(N + 3) / 4], int lane) {
  constexpr int N4 = (N + 3) / 4;
  const bool b0 = lane & 1, b1 = lane & 2;
  float u1[2 * N4];
#pragma unroll
  for (int j = 0; j < 2 * N4; ++j) {
    const float e = 2 * j < N ? v[2 * j] : 0.0f, o = 2 * j + 1 < N ? v[2 * j + 1] : 0.0f;
    const float keep = b0 ? o : e, send = b0 ? e : o;
    u1[j] = keep + dpp_mov<0xB1, 0xF>(0.0f, send);           // partner lane ^ 1
  }
#pragma unroll
  for (int j = 0; j < N4; ++j) {
    const float keep = b1 ? u1[2 * j + 1] : u1[2 * j], send = b1 ? u1[2 * j] : u1[2 * j + 1];
    float t = keep + dpp_mov<0x4E, 0xF>(0.0f, send);         // partner lane ^ 2: t = quad sum of value 4 j + (lane & 3)
    float x = dpp_mov<0x114, 0xA>(0.0f, t);                  // lane ^ 4: row_shr:4 into banks 1,3 ...
    x = dpp_mov<0x104, 0x5>(x, t);                           // ... row_shl:4 into banks 0,2
    t += x;
    t += dpp_mov<0x128, 0xF>(0.0f, t);                       // lane ^ 8 (row_ror:8): sum over the 16-lane row
    u2[j] = t;
  }
}
template <int N4>
__device__ __forceinline__ void rows_store(PLds& S, const float (&u2)[N4], int off, int wave, int lane) {
  if ((lane & 15) < 4) {
    float* r = S.red[wave * 4 + (lane >> 4)] + off;
#pragma unroll
    for (int j = 0; j < N4; ++j) r[4 * j + (lane & 3)] = u2[j];
  }
}
// LDS-only barriers (sdx_common.h): __syncthreads() also waits for the wave's outstanding global accesses - in the shadows that is the
// acknowledgement of the exchange words just published and the next minibatch's prefetched rows; nothing here is ordered through HBM
__device__ __forceinline__ void rows_reduce(PLds& S, int n, float* out, int tid) {
  SDX_LDS_BARRIER();
  if (tid < n) {
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < 4 * NWV; ++w) t += S.red[w][tid];
    out[tid] = t;
  }
  SDX_LDS_BARRIER();
}
template <int N>
__device__ __forceinline__ void block_sum(PLds& S, const float (&v)[N], float* out, int tid, int wave, int lane) {
  float u2[(N + 3) / 4];
  row_butterfly<N>(v, u2, lane);
  rows_store<(N + 3) / 4>(S, u2, 0, wave, lane);
  rows_reduce(S, N, out, tid);
}

static_assert(sizeof(PLds) + 512 <= 160 * 1024, "PLds (+ the static logstd bank) must fit the 160 KiB LDS of a gfx950 CU");
}  // namespace

// dataset-row helpers
__device__ __forceinline__ const float* obs_rows(const SdxpDev& D, int mb) { return D.mb_obs + (size_t)mb * MB * D.obs_dim; }
__device__ __forceinline__ const float* cvx_rows(const SdxpDev& D, int mb, int mini_epoch) {
  return (mini_epoch == 0 ? D.cvx0 : D.cvx1) + (size_t)mb * MB * ST;
}

// SINGLE = false: the whole update phase of an epoch (above).  SINGLE = true: forward + backward of ONE minibatch (the one the
// device cursor SdxpCtrl.mb_index points at) with the current parameters; no optimiser state is touched, the rank-MB factors
// are written to D.fact for the multi-rank exchange and the cursor / loss statistics advance as k_ctrl does in explicit mode.
template <bool SINGLE>
__global__ __launch_bounds__(NTH, 2) void k_update_persistent(SdxpDev D, int total_steps, unsigned* bar, unsigned* failflag, int stamps, int fault) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  PLds& S = *reinterpret_cast<PLds*>(smem_raw);
  // Thread coordinates are re-derived from an opaque copy of threadIdx/blockIdx at the start of every phase (refresh()):
  // otherwise every LDS address and ownership index of the step loop is hoisted into the loop preheader and the ~150 hoisted
  // values, on top of the resident weights and moments, push the kernel far over the 256-VGPR budget.
  int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = blockIdx.x;
  int r0w, h0, n0, r1w, q1, r1, c2c, c2n, he, hrow, hk;
  auto refresh = [&]() {
    int t = threadIdx.x, b = blockIdx.x;
    asm volatile("" : "+v"(t));
    asm volatile("" : "+s"(b));
    tid = t; lane = t & 63; wave = __builtin_amdgcn_readfirstlane(t >> 6); g = b;
    r0w = wave >> 1; h0 = wave & 1; n0 = 4 * g + r0w;
    r1w = wave >> 2; q1 = wave & 3; r1 = 2 * g + r1w;
    c2c = t >> 8; c2n = t & 255;
    he = HPC * g + t; hrow = he >> 8; hk = he & (U2 - 1);
  };
  refresh();
#define TS(i) if (stamps && g == (stamps >> 8) && tid == 0) { const long long t_ = __builtin_amdgcn_s_memtime(); S.tacc[i] += t_ - S.tlast; S.tlast = t_; }
  if (tid < 32) S.tacc[tid] = 0;
  if (tid == 0) S.fail = 0;
  if (tid == 0) S.tlast = __builtin_amdgcn_s_memtime();
  const int A = ACT;
  u64* const LL = D.ll;   // 8-byte exchange words of the head matrix (LL_HW)
  const __amdgpu_buffer_rsrc_t LQ = lq_rsrc(D.ll);   // 16-byte packed words of the chain edges, layout LQ_*: [s][unit], the three networks per word

  // ------------------------------------------------------------------ resident parameters
  // layer 0 rows: row n0 = 4g + (wave>>1), half h0 = wave&1 of K; net 0/1: K = OBS, net 2: K = ST
  float w0a[I0A], m0a[I0A], v0a[I0A], w0c[I0A], m0c[I0A], v0c[I0A], w0v[I0V], m0v[I0V], v0v[I0V];
  // layer 1 rows: row r1 = 2g + (wave>>2), quarter q1 = wave&3: k = q1*256 + lane + 64 i (i<4), three nets
  float w1[3][4], m1[3][4], v1[3][4];
  // layer 1 columns: cols 4g+c, row n = tid
  float c1[3][4], cm1[3][4], cv1[3][4];
  // layer 2 rows: row g, eighth = wave: k = wave*64 + lane
  float w2[3], m2[3], v2[3];
  // layer 2 columns: col 2g + (tid>>8), row n = tid & 255
  float c2[3], cm2[3], cv2[3];
  // heads ((A+2) x U2 = 6400 weights): CU g runs Adam for the HPC flat elements [HPC g, HPC g + HPC) on its first HPC threads and
  // publishes the new weight in the parameter array; phase D of every CU reads the whole head matrix back through L2.
  float hw = 0.0f, hm = 0.0f, hv = 0.0f;

  auto ldm = [&](const float* p, size_t o) -> float { return SINGLE ? 0.0f : p[o]; };   // Adam moments: not needed for one forward/backward
  auto P_of = [&](int net) -> float* { return net == 2 ? D.cv : D.ac; };
  auto M_of = [&](int net) -> float* { return net == 2 ? D.cv_m : D.ac_m; };
  auto V_of = [&](int net) -> float* { return net == 2 ? D.cv_v : D.ac_v; };
  auto woff = [&](int net, int l) -> size_t { return net == 0 ? D.off.a_w[l] : net == 1 ? D.off.c_w[l] : D.coff.w[l]; };
  auto boff = [&](int net, int l) -> size_t { return net == 0 ? D.off.a_b[l] : net == 1 ? D.off.c_b[l] : D.coff.b[l]; };
  auto head_woff = [&](int row) -> size_t { return row < A ? D.off.mu_w + (size_t)row * U2 : (row == A ? D.off.v_w : D.coff.v_w); };
  auto head_boff = [&](int row) -> size_t { return row < A ? D.off.mu_b + row : (row == A ? D.off.v_b : D.coff.v_b); };
  auto head_net = [&](int row) -> int { return row < A ? 0 : (row == A ? 1 : 2); };

#pragma unroll
  for (int i = 0; i < I0A; ++i) {
    const int kk = lane + 64 * i, k = h0 * H0A + kk;
    const bool ok = kk < H0A && k < D.obs_dim;
    const size_t oa = woff(0, 0) + (size_t)n0 * D.obs_dim + k, oc = woff(1, 0) + (size_t)n0 * D.obs_dim + k;
    w0a[i] = ok ? D.ac[oa] : 0.0f; m0a[i] = ok ? ldm(D.ac_m, oa) : 0.0f; v0a[i] = ok ? ldm(D.ac_v, oa) : 0.0f;
    w0c[i] = ok ? D.ac[oc] : 0.0f; m0c[i] = ok ? ldm(D.ac_m, oc) : 0.0f; v0c[i] = ok ? ldm(D.ac_v, oc) : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < I0V; ++i) {
    const int kk = lane + 64 * i, k = h0 * H0V + kk;
    const bool ok = kk < H0V;
    const size_t o = woff(2, 0) + (size_t)n0 * ST + k;
    w0v[i] = ok ? D.cv[o] : 0.0f; m0v[i] = ok ? ldm(D.cv_m, o) : 0.0f; v0v[i] = ok ? ldm(D.cv_v, o) : 0.0f;
  }
#pragma unroll
  for (int net = 0; net < 3; ++net) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t o = woff(net, 1) + (size_t)r1 * U0 + q1 * 256 + lane + 64 * i;
      w1[net][i] = P_of(net)[o]; m1[net][i] = ldm(M_of(net), o); v1[net][i] = ldm(V_of(net), o);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const size_t o = woff(net, 1) + (size_t)tid * U0 + 4 * g + c;
      c1[net][c] = P_of(net)[o]; cm1[net][c] = ldm(M_of(net), o); cv1[net][c] = ldm(V_of(net), o);
    }
    {
      const size_t o = woff(net, 2) + (size_t)g * U1 + wave * 64 + lane;
      w2[net] = P_of(net)[o]; m2[net] = ldm(M_of(net), o); v2[net] = ldm(V_of(net), o);
    }
    {
      const size_t o = woff(net, 2) + (size_t)c2n * U1 + 2 * g + c2c;
      c2[net] = P_of(net)[o]; cm2[net] = ldm(M_of(net), o); cv2[net] = ldm(V_of(net), o);
    }
  }
  if (tid < HPC) {
    const int net = head_net(hrow);
    const size_t o = head_woff(hrow) + hk;
    hw = P_of(net)[o]; hm = ldm(M_of(net), o); hv = ldm(V_of(net), o);
  }
  // biases + logstd (LDS, one thread each)
  __shared__ float s_b2[3][32];     // logstd value, m, v
  if (tid < 12) {   // bias slot map: L0 net*4 + row (12), L1 12 + net*2 + row (6), L2 18 + net (3), heads 21 + row (25)
    const int net = tid / 4, w = tid % 4;
    const size_t o = boff(net, 0) + 4 * g + w;
    S.bias[tid] = P_of(net)[o]; S.bias_m[tid] = ldm(M_of(net), o); S.bias_v[tid] = ldm(V_of(net), o);
  } else if (tid < 18) {
    const int net = (tid - 12) / 2, r = (tid - 12) % 2;
    const size_t o = boff(net, 1) + 2 * g + r;
    S.bias[tid] = P_of(net)[o]; S.bias_m[tid] = ldm(M_of(net), o); S.bias_v[tid] = ldm(V_of(net), o);
  } else if (tid < 21) {
    const int net = tid - 18;
    const size_t o = boff(net, 2) + g;
    S.bias[tid] = P_of(net)[o]; S.bias_m[tid] = ldm(M_of(net), o); S.bias_v[tid] = ldm(V_of(net), o);
  } else if (tid < 21 + A + 2) {
    const int row = tid - 21, net = head_net(row);
    const size_t o = head_boff(row);
    S.bias[tid] = P_of(net)[o]; S.bias_m[tid] = ldm(M_of(net), o); S.bias_v[tid] = ldm(V_of(net), o);
  } else if (tid >= 64 && tid < 64 + A) {
    const int a = tid - 64;
    const size_t o = D.off.logstd + a;
    s_b2[0][a] = D.ac[o]; s_b2[1][a] = ldm(D.ac_m, o); s_b2[2][a] = ldm(D.ac_v, o);
  }
  // replicated control state (every CU runs the same state machine on identical inputs)
  SdxpCtrl* gctl = D.ctrl;
  // first exchange tag of this launch minus one: read from the device control block (CU 0 advances it at the end of the launch, when
  // every CU has long read it: nobody finishes a step without everybody's words) - so launches replayed from a hipGraph keep growing tags
  const unsigned tag_base = gctl->ll_tag;
  if (tid == 0) {
    PLds::Ctl& C = S.ctl;
    C.ac_lr = gctl->ac_lr; C.ac_lr_applied = C.ac_lr; C.cv_lr = gctl->cv_lr; C.ac_t = gctl->ac_t; C.cv_t = gctl->cv_t;
    C.ac_b1 = gctl->ac_b1pow; C.ac_b2 = gctl->ac_b2pow; C.cv_b1 = gctl->cv_b1pow; C.cv_b2 = gctl->cv_b2pow;
    C.sum_a = C.sum_c = C.sum_b = C.sum_kl = C.sum_cv = C.sum_ent = C.last_kl = C.ac_gn = C.cv_gn = 0.0f;
  }
  bool pending = false;
  // layer-0 gradient elements of this lane (sum_s dY0_s[row] x0_s[k], before the clip scale), formed in the dY0 shadow of the step that
  // produced them - the LDS round trips of the rebuild are off the chain, Adam of layer 0 runs on registers (round 6)
  float g0a[I0A], g0c[I0A], g0v[I0V];
#pragma unroll
  for (int i = 0; i < I0A; ++i) { g0a[i] = 0.0f; g0c[i] = 0.0f; }
#pragma unroll
  for (int i = 0; i < I0V; ++i) g0v[i] = 0.0f;
  // columns of the layer-0 input image past obs_dim stay zero for the whole launch (the row loads below never touch them)
  for (int i = tid; i < MB * OBS; i += NTH) (&S.obs[0][0])[i] = 0.0f;
  __syncthreads();

    // This step's layer-0 inputs (dataset rows mb*MB .. +MB) go STRAIGHT into LDS (global_load_lds_dword: lane tid of a wave fetches element
  // wave 64 + lane of a row, the wave's 64 dwords land behind one wave-uniform LDS address; no VGPR round trip, no ds_write, no barrier in
  // front of the copy).  The old rows were last read by the dY0 shadow (the layer-0 gradient rebuild), which every wave finished before the
  // barrier at the top of the step; the loads fly under the norm gather and Adam of layer 0 and are waited for in front of forward L0.
  auto load_rows = [&](int mb_, int me_) {
    // row pointers are wave-uniform (scalar registers), the lane adds its 32-bit offset: ONE address VGPR for the twelve loads (twelve 64-bit
    // per-lane pointers formed up front were 24 VGPRs - the kernel has none to spare - and spilled the first looks of the x2 / dY1 words)
    // (the "+s" asm keeps each pointer in a scalar register pair and in place: hoisted out of the step loop as per-lane 64-bit values they
    // cost VGPR pairs for the whole step)
    auto uni = [](const float* q) -> const float* { unsigned long long a = (unsigned long long)q; asm volatile("" : "+s"(a)); return (const float*)a; };
    const float* o = uni(obs_rows(D, mb_));
    const float* c = uni(cvx_rows(D, mb_, me_));
    const unsigned odim = D.obs_dim;
    if (odim == OBS) {
      // full-width observations: the MB rows are contiguous in the dataset AND in LDS (S.obs, then S.cvx): 960 pieces of 16 bytes, two
      // global_load_lds_dwordx4 per lane (twelve dword loads with their exec masks and M0 writes cost phase A 0.4 us of issue)
      static_assert(offsetof(PLds, cvx) == sizeof(float) * MB * OBS && (MB * OBS) % 4 == 0 && (MB * ST) % 4 == 0, "obs and cvx images are adjacent 16-byte runs");
      constexpr int PO = MB * OBS / 4, PA = PO + MB * ST / 4;
      char* l0 = reinterpret_cast<char*>(&S.obs[0][0]);
#pragma unroll
      for (int j = 0; j < (PA + NTH - 1) / NTH; ++j) {
        const int pc = tid + NTH * j;
        if (pc < PA) {
          const float* src = pc < PO ? o + 4 * pc : c + 4 * (pc - PO);
          __builtin_amdgcn_global_load_lds(SDX_AS_GLOBAL(src), SDX_AS_LDS(l0 + (wave + NWV * j) * 1024), 16, 0, 0);
        }
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < MB; ++s) {
      const float* orow = uni(o + s * odim);
      const float* crow = uni(c + s * ST);
      if ((unsigned)tid < odim) __builtin_amdgcn_global_load_lds(SDX_AS_GLOBAL(orow + (unsigned)tid), SDX_AS_LDS(&S.obs[s][wave * 64]), 4, 0, 0);
      __builtin_amdgcn_global_load_lds(SDX_AS_GLOBAL(crow + (unsigned)tid), SDX_AS_LDS(&S.cvx[s][wave * 64]), 4, 0, 0);
      if (tid + NTH < ST) __builtin_amdgcn_global_load_lds(SDX_AS_GLOBAL(crow + NTH + (unsigned)tid), SDX_AS_LDS(&S.cvx[s][NTH]), 4, 0, 0);
    }
  };
  if (total_steps > 0) load_rows(SINGLE ? gctl->mb_index : 0, SINGLE ? gctl->mini_epoch : 0);   // rows of the first minibatch
  // first look at the gradient-norm words (round 6): every CU published its word at the end of phase E, 1.6 us before the top of the next
  // step, so the words are there when the dY0 shadow ends - what the gather at the top of the step paid for was its own round trip
  // (0.9 us of the 1.2: all 256 CUs show the same 1.2 us, there is no late producer).  The loads are requested behind the shadow's second
  // barrier and taken at the top of the next step; a miss falls back to the polling gather.
  u32x4 pg0[4];
  bool pg0_issued = false;
  for (int step = 0; step <= total_steps; ++step) {
    SDX_LDS_BARRIER();   // LDS only: the next step's row loads (requested in the dY0 shadow) stay in flight across it
    if (S.fail) return;   // an exchange word never arrived (not all CUs resident?): the host sees *failflag and reports it
    if (fault && g == NWG - 1 && step == 3) return;   // SDXP_PERSIST_FAULT=1 (tests): one CU goes silent, the others must time out
    refresh();
    const bool last = step == total_steps;   // the extra iteration only applies the optimiser step of the final minibatch
    const int mbi = SINGLE ? gctl->mb_index : step % D.num_minibatches, mini_epoch = SINGLE ? gctl->mini_epoch : step / D.num_minibatches;
    // tag of everything this step produces / of what the previous step produced.  tag_base advances from launch to launch (the
    // exchange buffer is never cleared), so a word left over from an earlier launch can never match
    const unsigned tag = tag_base + (unsigned)step + 1u, tag_prev = tag_base + (unsigned)step;
    // ================================================================== phase A: gradient norm, Adam of layer 0, forward L0
    if (pending) {
      // Squared gradient norm -> clip scale, WITHOUT a workgroup barrier or an LDS pass on the step's dependent chain (round 6).  Every CU
      // published ONE word at the end of the previous step: its four columns' share of <G_dY0, G_x0> + |sum_s dY0_s|^2 for the three
      // networks (rounds 4-5: the 10 entries of its Gram; 320 threads gathered 2 560 words, reduced them through LDS and wave 0 ran the
      // control block between two barriers: 2.2 us from the top of the step to the first Adam instruction).  Every WAVE now gathers the
      // 256 words itself - CUs lane, lane + 64, ... - and adds them in one fixed order (pair sums, then the DPP tree of wave_sum), so all
      // waves of all CUs hold bit-identical scalars and walk into Adam of layer 0 as their own words arrive.  What does not depend on dY0
      // - the other layers' and the heads' terms (S.n2rest), the bias corrections and the learning-rate scalars (S.scal[2..5]) - was
      // prepared in the shadows of the previous step.
      // (what the scalars below need from LDS is requested in front of the norm words' take: one LDS round trip less behind the DPP sums)
      float nr0 = S.n2rest[0], nr1 = S.n2rest[1], nr2 = S.n2rest[2];
      float ac_lr_bc1 = S.scal[2], ac_isq = S.scal[3], cv_lr_bc1 = S.scal[4], cv_isq = S.scal[5];
      float w0[4], w1[4], w2[4];
      if (!(pg0_issued && lq_take<4>(pg0, tag_prev, w0, w1, w2)) && !lq_gather<4>(LQ, LQ_G0 + lane, 64, tag_prev, w0, w1, w2, failflag)) S.fail = 1;
      float n2[3];
      n2[0] = (w0[0] + w0[1]) + (w0[2] + w0[3]); n2[1] = (w1[0] + w1[1]) + (w1[2] + w1[3]); n2[2] = (w2[0] + w2[1]) + (w2[2] + w2[3]);
#pragma unroll
      for (int net = 0; net < 3; ++net) n2[net] = wave_sum(n2[net]);     // three independent DPP chains
      SDX_PIN4X(nr0, nr1, nr2, ac_lr_bc1); SDX_PIN3X(ac_isq, cv_lr_bc1, cv_isq);   // (used from here on; requested above)
      n2[0] += nr0; n2[1] += nr1; n2[2] += nr2;
      // v_sqrt_f32 / v_rcp_f32 (1 ulp each, as in the Adam arithmetic) instead of the IEEE sequences: these four sit between the last word of the
      // norm and the first Adam instruction of every lane
      const float ac_gn = __builtin_amdgcn_sqrtf(n2[0] + n2[1]), cv_gn = __builtin_amdgcn_sqrtf(n2[2]);
      const float gs_ac = D.truncate_grads ? fminf(1.0f, D.grad_norm * __builtin_amdgcn_rcpf(ac_gn + 1e-6f)) : 1.0f;
      const float gs_cv = D.truncate_grads ? fminf(1.0f, D.grad_norm * __builtin_amdgcn_rcpf(cv_gn + 1e-6f)) : 1.0f;
      if (tid == 0) { S.ctl.ac_gn = ac_gn; S.ctl.cv_gn = cv_gn; S.scal[0] = gs_ac; S.scal[1] = gs_cv; }   // for the shadows' Adam phases (behind the next barrier)
      TS(0)
      // ---- layer 0 rows: gradient elements from registers (g0a / g0c / g0v, dY0 shadow of the previous iteration), scaled by the clip factor
#pragma unroll
      for (int i = 0; i < I0A; ++i) {
        ADAM0(w0a[i], g0a[i] * gs_ac, m0a[i], v0a[i], ac_lr_bc1, ac_isq);
        ADAM0(w0c[i], g0c[i] * gs_ac, m0c[i], v0c[i], ac_lr_bc1, ac_isq);
      }
#pragma unroll
      for (int i = 0; i < I0V; ++i) ADAM0(w0v[i], g0v[i] * gs_cv, m0v[i], v0v[i], cv_lr_bc1, cv_isq);
      if (tid < 12) {   // layer-0 biases (bias gradient = sum_s dY_s[row])
        const int net = tid / 4;
        float gg = 0.0f;
        for (int s = 0; s < MB; ++s) gg += S.dyown[net][s][tid % 4];
        float b = S.bias[tid], bm = S.bias_m[tid], bv = S.bias_v[tid];
        adam1(b, gg * (net == 2 ? gs_cv : gs_ac), bm, bv, net == 2 ? cv_lr_bc1 : ac_lr_bc1, net == 2 ? cv_isq : ac_isq);
        S.bias[tid] = b; S.bias_m[tid] = bm; S.bias_v[tid] = bv;
      }
      TS(1)
    }
    if (!last) {
      // ---- forward L0 of the owned rows (two half-row waves per row, combined through LDS) once this step's rows have landed
      SDX_WAIT_VMCNT0();
      __syncthreads();
      float pa[MB], pc[MB], pv[MB];
#pragma unroll
      for (int s = 0; s < MB; ++s) { pa[s] = 0.0f; pc[s] = 0.0f; pv[s] = 0.0f; }
#pragma unroll
      for (int i = 0; i < I0A; ++i) {
        const int kk = lane + 64 * i, k = h0 * H0A + kk;
        if (kk < H0A) {
#pragma unroll
          for (int s = 0; s < MB; ++s) { const float x = S.obs[s][k]; pa[s] += w0a[i] * x; pc[s] += w0c[i] * x; }
        }
      }
#pragma unroll
      for (int i = 0; i < I0V; ++i) {
        const int kk = lane + 64 * i, k = h0 * H0V + kk;
        if (kk < H0V) {
#pragma unroll
          for (int s = 0; s < MB; ++s) pv[s] += w0v[i] * S.cvx[s][k];
        }
      }
      // (all twelve wave reductions first - independent DPP chains the scheduler interleaves - then the stores: one reduction, one store
      // at a time serialises the chains behind each other's LDS traffic; same sums, same order, bit-identical results)
#pragma unroll
      for (int s = 0; s < MB; ++s) { pa[s] = wave_sum(pa[s]); pc[s] = wave_sum(pc[s]); pv[s] = wave_sum(pv[s]); }
      if (lane == 0) {
#pragma unroll
        for (int s = 0; s < MB; ++s) { S.fpart[(0 * MB + s) * NWV + wave] = pa[s]; S.fpart[(1 * MB + s) * NWV + wave] = pc[s]; S.fpart[(2 * MB + s) * NWV + wave] = pv[s]; }
      }
      __syncthreads();
      if (tid < MB * 4) {   // (sample s, owned row r): the three networks' outputs in one packed word
        const int s = tid / 4, r = tid % 4;
        float y[3];
#pragma unroll
        for (int net = 0; net < 3; ++net)
          y[net] = elu(S.fpart[(net * MB + s) * NWV + 2 * r] + S.fpart[(net * MB + s) * NWV + 2 * r + 1] + S.bias[net * 4 + r]);
        lq_store(LQ, LQ_X1 + (unsigned)s * U0 + 4 * g + r, y[0], y[1], y[2], tag);
      }
      TS(2)
      if constexpr (!SINGLE) {
      // ---- in the shadow of the x1 exchange: Gram of the layer-0 inputs
      float p[22];
#pragma unroll
      for (int i = 0; i < 22; ++i) p[i] = 0.0f;
      if (tid < OBS) gram_acc(&p[0], S.obs[0][tid], S.obs[1][tid], S.obs[2][tid], S.obs[3][tid]);
      gram_acc(&p[11], S.cvx[0][tid], S.cvx[1][tid], S.cvx[2][tid], S.cvx[3][tid]);
      if (tid + NTH < ST) gram_acc(&p[11], S.cvx[0][tid + NTH], S.cvx[1][tid + NTH], S.cvx[2][tid + NTH], S.cvx[3][tid + NTH]);
      block_sum<22>(S, p, S.part, tid, wave, lane);
      if (tid < 16) { const float v = S.part[tri16(tid)]; S.gx[0][0][tid] = v; S.gx[1][0][tid] = v; }
      else if (tid >= 64 && tid < 80) S.gx[2][0][tid - 64] = S.part[11 + tri16(tid - 64)];
      }
    }
    refresh();
    if (pending) {
      // ---- (shadow of x1) Adam of layer 1: row copy and column copy, operands are last step's S.x1 / S.dy1
      const float gs_ac = S.scal[0], gs_cv = S.scal[1];
      const float ac_lr_bc1 = S.scal[2], ac_isq = S.scal[3], cv_lr_bc1 = S.scal[4], cv_isq = S.scal[5];
      float dep = 0.0f;
#pragma unroll
      for (int net = 0; net < 3; ++net) {
        const float gs = net == 2 ? gs_cv : gs_ac, lrb = net == 2 ? cv_lr_bc1 : ac_lr_bc1, isq = net == 2 ? cv_isq : ac_isq;
        float d1r[MB], dn1[MB];
        const int o_r = opaque(r1w, dep), o_t = opaque(tid, dep);
#pragma unroll
        for (int s = 0; s < MB; ++s) { d1r[s] = S.dyown[net][s][4 + o_r] * gs; dn1[s] = S.dy1[net][s][o_t] * gs; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = opaque(q1 * 256 + lane + 64 * i, dep);
          float gg = 0.0f;
#pragma unroll
          for (int s = 0; s < MB; ++s) gg += d1r[s] * S.x1[net][s][k];
          adam1(w1[net][i], gg, m1[net][i], v1[net][i], lrb, isq); dep = w1[net][i];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int k = opaque(4 * g + c, dep);
          float gg = 0.0f;
#pragma unroll
          for (int s = 0; s < MB; ++s) gg += dn1[s] * S.x1[net][s][k];
          adam1(c1[net][c], gg, cm1[net][c], cv1[net][c], lrb, isq); dep = c1[net][c];
        }
      }
      if (tid >= 12 && tid < 18) {   // layer-1 biases
        const int net = (tid - 12) / 2;
        float gg = 0.0f;
        for (int s = 0; s < MB; ++s) gg += S.dyown[net][s][4 + (tid - 12) % 2];
        float b = S.bias[tid], bm = S.bias_m[tid], bv = S.bias_v[tid];
        adam1(b, gg * (net == 2 ? gs_cv : gs_ac), bm, bv, net == 2 ? cv_lr_bc1 : ac_lr_bc1, net == 2 ? cv_isq : ac_isq);
        S.bias[tid] = b; S.bias_m[tid] = bm; S.bias_v[tid] = bv;
      }
    }
    TS(3)
    if (!last) {
      __syncthreads();   // every wave is done with the old S.x1
      // ================================================================== phase B: gather x1, forward L1
      {
        float v0[8], v1[8], v2[8];   // word tid + NTH j = (sample, unit) s U0 + unit: S.x1[net] is [MB][U0], i.e. the same flat index
        if (!lq_gather<8>(LQ, LQ_X1 + tid, NTH, tag, v0, v1, v2, failflag)) S.fail = 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { (&S.x1[0][0][0])[tid + NTH * j] = v0[j]; (&S.x1[1][0][0])[tid + NTH * j] = v1[j]; (&S.x1[2][0][0])[tid + NTH * j] = v2[j]; }
      }
      TS(4)
      __syncthreads();
      {
        float q[3][MB];
#pragma unroll
        for (int net = 0; net < 3; ++net) {
#pragma unroll
          for (int s = 0; s < MB; ++s) q[net][s] = 0.0f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int k = q1 * 256 + lane + 64 * i;
#pragma unroll
            for (int s = 0; s < MB; ++s) q[net][s] += w1[net][i] * S.x1[net][s][k];
          }
        }
#pragma unroll
        for (int net = 0; net < 3; ++net)
#pragma unroll
          for (int s = 0; s < MB; ++s) q[net][s] = wave_sum(q[net][s]);
        if (lane == 0) {
#pragma unroll
          for (int net = 0; net < 3; ++net)
#pragma unroll
            for (int s = 0; s < MB; ++s) S.fpart[(net * MB + s) * NWV + wave] = q[net][s];
        }
      }
      __syncthreads();
      if (tid < MB * 2) {
        const int s = tid / 2, r = tid % 2;
        float y[3];
#pragma unroll
        for (int net = 0; net < 3; ++net) {
          const float* q = &S.fpart[(net * MB + s) * NWV + 4 * r];
          y[net] = elu(q[0] + q[1] + q[2] + q[3] + S.bias[12 + net * 2 + r]);
        }
        lq_store(LQ, LQ_X2 + (unsigned)s * U1 + 2 * g + r, y[0], y[1], y[2], tag);
      }
      TS(5)
      // (the Gram of x1 used to sit here: this shadow was 2.7 us against an edge of 2.1; it now shares ONE block reduction with the
      // Gram of x2 in the shadow of x3 - 60 values through the butterfly instead of 2 x 33 through two of them)
    }
    refresh();
    u32x4 px2[4];
    bool px2_issued = false;
    if (pending) {
      // ---- (shadow of x2) Adam of layer 2 (row and column copies), of this CU's head elements, of the remaining biases and logstd
      const float gs_ac = S.scal[0], gs_cv = S.scal[1];
      const float ac_lr_bc1 = S.scal[2], ac_isq = S.scal[3], cv_lr_bc1 = S.scal[4], cv_isq = S.scal[5];
      float dep = 0.0f;
#pragma unroll
      for (int net = 0; net < 3; ++net) {
        // first look at this lane's x2 words before the last network's update: the shadow has outlasted the edge by then (2.2 us of
        // its 2.8 against an edge of 2.1), what follows hides the round trip
        if (net == 2 && !last) { lq_peek<4>(LQ, LQ_X2 + tid, NTH, px2); px2_issued = true; }
        const float gs = net == 2 ? gs_cv : gs_ac, lrb = net == 2 ? cv_lr_bc1 : ac_lr_bc1, isq = net == 2 ? cv_isq : ac_isq;
        float d2r[MB], dn2[MB];
        const int o_n = opaque(c2n, dep);
#pragma unroll
        for (int s = 0; s < MB; ++s) { d2r[s] = S.dyown[net][s][6] * gs; dn2[s] = S.dy2[net][s][o_n] * gs; }
        {
          const int k = opaque(wave * 64 + lane, dep);
          float gg = 0.0f;
#pragma unroll
          for (int s = 0; s < MB; ++s) gg += d2r[s] * S.x2[net][s][k];
          adam1(w2[net], gg, m2[net], v2[net], lrb, isq); dep = w2[net];
        }
        {
          const int k = opaque(2 * g + c2c, dep);
          float gg = 0.0f;
#pragma unroll
          for (int s = 0; s < MB; ++s) gg += dn2[s] * S.x2[net][s][k];
          adam1(c2[net], gg, cm2[net], cv2[net], lrb, isq); dep = c2[net];
        }
      }
      // heads: this CU's HPC elements; the new weight is published for phase D of every CU (tag == step)
      if (tid < HPC) {
        const int net = head_net(hrow);
        const float gs = net == 2 ? gs_cv : gs_ac, lrb = net == 2 ? cv_lr_bc1 : ac_lr_bc1, isq = net == 2 ? cv_isq : ac_isq;
        float gg = 0.0f;
#pragma unroll
        for (int s = 0; s < MB; ++s) gg += (hrow < A ? S.dmu[s][hrow] : S.dv[hrow - A][s]) * gs * S.x3[net][s][hk];
        adam1(hw, gg, hm, hv, lrb, isq);
        ll_store(LL + LL_HW + he, hw, tag_prev);
      }
      if (tid >= 18 && tid < 21 + A + 2) {   // layer-2 and head biases
        int net; float gg = 0.0f;
        if (tid < 21) { net = tid - 18; for (int s = 0; s < MB; ++s) gg += S.dyown[net][s][6]; }
        else { const int row = tid - 21; net = head_net(row); for (int s = 0; s < MB; ++s) gg += row < A ? S.dmu[s][row] : S.dv[row - A][s]; }
        float b = S.bias[tid], bm = S.bias_m[tid], bv = S.bias_v[tid];
        adam1(b, gg * (net == 2 ? gs_cv : gs_ac), bm, bv, net == 2 ? cv_lr_bc1 : ac_lr_bc1, net == 2 ? cv_isq : ac_isq);
        S.bias[tid] = b; S.bias_m[tid] = bm; S.bias_v[tid] = bv;
      } else if (tid >= 64 && tid < 64 + A) {   // logstd
        const int a = tid - 64;
        float b = s_b2[0][a], bm = s_b2[1][a], bv = s_b2[2][a];
        adam1(b, S.dls[a] * gs_ac, bm, bv, ac_lr_bc1, ac_isq);
        s_b2[0][a] = b; s_b2[1][a] = bm; s_b2[2][a] = bv;
      }
    }
    TS(6)
    if (last) break;
    __syncthreads();   // every wave is done with the old S.x2 / S.x3 / S.dy2 / S.dmu / S.dv
    // ================================================================== phase C: gather x2, forward L2
    const size_t r0 = (size_t)mbi * MB;
    float pf_act = 0.0f, pf_omu = 0.0f, pf_osg = 1.0f;   // this minibatch's actions and old (mu, sigma): needed in phase D
    float pf_adv = 0.0f, pf_nlp = 0.0f, pf_ret = 0.0f, pf_val = 0.0f;   // per-sample scalars, held by the lane that forms the sample's losses
    if (tid < MB * 32 && (tid % 32) == 31) {
      const size_t i = r0 + tid / 32;
      pf_adv = D.adv[i]; pf_nlp = D.mb_neglogp[i]; pf_ret = D.returns[i]; pf_val = D.mb_values[i];
    }
    if (tid < MB * 32 && (tid % 32) < A) {
      const size_t i = (r0 + tid / 32) * A + tid % 32;
      pf_act = D.mb_actions[i];
      // (mu, sigma) were rewritten by CU 0 one mini-epoch ago (update_mu_sigma): agent-scope accesses keep them coherent across XCDs
      pf_omu = __hip_atomic_load(&D.mb_mus[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pf_osg = __hip_atomic_load(&D.mb_sigmas[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    {
      float v0[4], v1[4], v2[4];
      if (!(px2_issued && lq_take<4>(px2, tag, v0, v1, v2)) && !lq_gather<4>(LQ, LQ_X2 + tid, NTH, tag, v0, v1, v2, failflag)) S.fail = 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) { (&S.x2[0][0][0])[tid + NTH * j] = v0[j]; (&S.x2[1][0][0])[tid + NTH * j] = v1[j]; (&S.x2[2][0][0])[tid + NTH * j] = v2[j]; }
    }
    TS(7)
    __syncthreads();
    {
      float q[3][MB];
      const int k = wave * 64 + lane;
#pragma unroll
      for (int net = 0; net < 3; ++net)
#pragma unroll
        for (int s = 0; s < MB; ++s) q[net][s] = w2[net] * S.x2[net][s][k];
#pragma unroll
      for (int net = 0; net < 3; ++net)
#pragma unroll
        for (int s = 0; s < MB; ++s) q[net][s] = wave_sum(q[net][s]);
      if (lane == 0) {
#pragma unroll
        for (int net = 0; net < 3; ++net)
#pragma unroll
          for (int s = 0; s < MB; ++s) S.fpart[(net * MB + s) * NWV + wave] = q[net][s];
      }
    }
    __syncthreads();
    if (tid < MB) {
      const int s = tid;
      float y[3];
#pragma unroll
      for (int net = 0; net < 3; ++net) {
        const float* q = &S.fpart[(net * MB + s) * NWV];
        float t = S.bias[18 + net];
#pragma unroll
        for (int w = 0; w < NWV; ++w) t += q[w];
        y[net] = elu(t);
      }
      lq_store(LQ, LQ_X3 + (unsigned)s * U2 + g, y[0], y[1], y[2], tag);
    }
    TS(8)
    // (this minibatch's actions and old (mu, sigma) go to LDS before the Grams: three registers fewer across the 60-value reduction)
    if (tid < MB * 32) { S.act[tid / 32][tid % 32] = pf_act; S.omu[tid / 32][tid % 32] = pf_omu; S.osg[tid / 32][tid % 32] = pf_osg; }
    if (tid >= 128 && tid < 128 + 32) { const float l_ = (tid - 128 < A) ? s_b2[0][tid - 128] : 0.0f; S.ls[tid - 128] = l_; S.sgm[tid - 128] = expf(l_); }
    if constexpr (!SINGLE)
    {   // ---- (shadow of x3) Grams of x2 and x1 (S.x1 stays valid until the next step's gather).  Two butterflies, ONE pass through LDS
        // and its two barriers: out[0..29] = [x2: net][10], out[32..61] = [x1: net][10]
      float ua[8], ub[8];
      {
        float p[30];
#pragma unroll
        for (int i = 0; i < 30; ++i) p[i] = 0.0f;
#pragma unroll
        for (int net = 0; net < 3; ++net) gram_acc10(&p[net * 10], S.x2[net][0][tid], S.x2[net][1][tid], S.x2[net][2][tid], S.x2[net][3][tid]);
        row_butterfly<30>(p, ua, lane);
      }
      rows_store<8>(S, ua, 0, wave, lane);
      {
        float p[30];
#pragma unroll
        for (int i = 0; i < 30; ++i) p[i] = 0.0f;
#pragma unroll
        for (int h = 0; h < U0 / NTH; ++h)
#pragma unroll
          for (int net = 0; net < 3; ++net) { const int k = tid + NTH * h; gram_acc10(&p[net * 10], S.x1[net][0][k], S.x1[net][1][k], S.x1[net][2][k], S.x1[net][3][k]); }
        row_butterfly<30>(p, ub, lane);
      }
      rows_store<8>(S, ub, 32, wave, lane);
      rows_reduce(S, 64, S.part, tid);
      if (tid < 48) S.gx[tid >> 4][2][tid & 15] = S.part[(tid >> 4) * 10 + tri16(tid & 15)];
      else if (tid >= 64 && tid < 64 + 48) { const int t = tid - 64; S.gx[t >> 4][1][t & 15] = S.part[32 + (t >> 4) * 10 + tri16(t & 15)]; }
    }
    TS(9)
    refresh();
    // ================================================================== phase D: gather x3 and the heads, losses, backward to dY1
    float wh[HR][4];   // this wave's head rows for the forward (row = wave + 8 j, columns lane + 64 e)
    // The head words were published a phase ago (shadow of x2), so they are normally all there.  Row view (head forward): all 16 loads
    // of a lane are issued before the first tag is looked at, and they share their round trip with the first look at the lane's x3
    // words (round 3: five chained gathers, then the x3 gather: 2.4 us between the x3 publish and the head forward, 0.7 of them the
    // x3 edge itself).  If a word is missing the polling gathers below wait.  The column view is first used by the heads' backward:
    // it is requested after the head forward, under the loss phase (both views at once - 58 registers in flight - spilled).
    bool row_done = false;
    {
      // first look at this lane's two x3 words, issued BEFORE the head-word loads: both round trips overlap; if the x3 words of a slow
      // producer are not there yet, the polling gather below continues after the head words have been taken
      u32x4 w3[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) w3[i] = __builtin_amdgcn_raw_buffer_load_b128(LQ, (LQ_X3 + tid + i * NTH) * 16u, 0, 16);
      if (step != 0) {
        u64 ww[HR][4];
        const u64 absent = (u64)tag_prev << 32;        // rows past the 25 head rows: value 0 with the expected tag
#pragma unroll
        for (int j = 0; j < HR; ++j) {
          const int row = wave + 8 * j;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            ww[j][e] = row < A + 2 ? __hip_atomic_load(LL + LL_HW + (size_t)row * U2 + lane + 64 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : absent;
        }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < HR; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) ok = ok && (unsigned)(ww[j][e] >> 32) == tag_prev;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
#pragma unroll
          for (int j = 0; j < HR; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) wh[j][e] = __uint_as_float((unsigned)ww[j][e]);
          row_done = true;
        }
      }
      float v0[2], v1[2], v2[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { v0[i] = __uint_as_float(w3[i].x); v1[i] = __uint_as_float(w3[i].y); v2[i] = __uint_as_float(w3[i].z); }
      if (__builtin_amdgcn_ballot_w64(!(lq_ok(w3[0], tag) && lq_ok(w3[1], tag))) != 0) {
        if (!lq_gather<2>(LQ, LQ_X3 + tid, NTH, tag, v0, v1, v2, failflag)) S.fail = 1;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) { (&S.x3[0][0][0])[tid + NTH * j] = v0[j]; (&S.x3[1][0][0])[tid + NTH * j] = v1[j]; (&S.x3[2][0][0])[tid + NTH * j] = v2[j]; }
    }
    if (!row_done) {
#pragma unroll
    for (int j = 0; j < HR; ++j) {
      const int row = wave + 8 * j;
#pragma unroll
      for (int e = 0; e < 4; ++e) wh[j][e] = 0.0f;
      if (row < A + 2) {
        if (step == 0) {
          const float* hp = P_of(head_net(row)) + head_woff(row) + lane;
#pragma unroll
          for (int e = 0; e < 4; ++e) wh[j][e] = hp[64 * e];
        } else if (!ll_gather<4>(LL + LL_HW + (size_t)row * U2 + lane, 64, tag_prev, wh[j], failflag)) S.fail = 1;   // lane-contiguous words
      }
    }
    }
    TS(10)
    __syncthreads();
    // head forward: wave per row
    {
      // rows wave, wave + 8, wave + 16 (< 23: actor rows, except row 23 = the critic's value on wave 7) and wave + 24 (only wave 0: row 24,
      // the central value): the actor rows of a wave share ONE set of x3 loads
      float q[HR][MB], xa[MB][4];
#pragma unroll
      for (int s = 0; s < MB; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) xa[s][e] = S.x3[0][s][lane + 64 * e];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < MB; ++s) {
          float t = 0.0f;
#pragma unroll
          for (int e = 0; e < 4; ++e) t += wh[j][e] * xa[s][e];
          q[j][s] = t;
        }
      if (wave == NWV - 1) {          // row 23: critic value (net 1)
#pragma unroll
        for (int s = 0; s < MB; ++s) {
          float t = 0.0f;
#pragma unroll
          for (int e = 0; e < 4; ++e) t += wh[2][e] * S.x3[1][s][lane + 64 * e];
          q[2][s] = t;
        }
      } else {
#pragma unroll
        for (int s = 0; s < MB; ++s) {
          float t = 0.0f;
#pragma unroll
          for (int e = 0; e < 4; ++e) t += wh[2][e] * xa[s][e];
          q[2][s] = t;
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int s = 0; s < MB; ++s) q[j][s] = wave_sum(q[j][s]);                       // 12 independent chains, then the stores
      if (wave == 0) {                // row 24: central value (net 2)
#pragma unroll
        for (int s = 0; s < MB; ++s) {
          float t = 0.0f;
#pragma unroll
          for (int e = 0; e < 4; ++e) t += wh[3][e] * S.x3[2][s][lane + 64 * e];
          q[3][s] = wave_sum(t);
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < HR; ++j) {
          const int row = wave + 8 * j;
          if (row < A + 2) {
#pragma unroll
            for (int s = 0; s < MB; ++s) { const float y = q[j][s] + S.bias[21 + row]; if (row < A) S.mu[s][row] = y; else S.val[row - A][s] = y; }
          }
        }
      }
    }
    __syncthreads();
    TS(11)
    // Column view of the head words (this lane's column c2n, rows 13 c2c .. 13 c2c + 12; first used by the heads' backward): requested
    // here, under the loss phase.  No tags: between them the row views of the eight waves cover every head word, each wave has seen
    // its words' tags (fast path or polling gather) before the barrier in front of the head forward, and the words are not rewritten
    // before every CU has finished this step - so the low halves (the values) are read as plain agent-scope dwords, 13 registers.
    // (the per-sample scalars prefetched in phase C are taken NOW: the loads below sit behind lane-dependent conditions, and after such a
    // merge the compiler's counter model waits for everything outstanding at the next use of any loaded register - that use would be
    // pf_adv in the loss phase, i.e. the column loads' whole round trip: 0.6 us on the chain)
    SDX_OPAQUE(pf_adv); SDX_OPAQUE(pf_nlp); SDX_OPAQUE(pf_ret); SDX_OPAQUE(pf_val);
    float wc[13];      // (row 25 = 13 + 12 does not exist: its slot repeats row 24 and is never used)
    if (step == 0) {
#pragma unroll
      for (int j = 0; j < 13; ++j) { const int row = min(13 * c2c + j, A + 1); wc[j] = P_of(head_net(row))[head_woff(row) + c2n]; }
    } else {           // 13 unconditional loads, nothing between them that looks at a loaded value
      const unsigned* hwv = reinterpret_cast<const unsigned*>(LL + LL_HW) + 2 * c2n;
#pragma unroll
      for (int j = 0; j < 13; ++j) {
        const int row = min(13 * c2c + j, A + 1);
        wc[j] = __uint_as_float(__hip_atomic_load(hwv + (size_t)row * (2 * U2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
    }
    const float invM = 1.0f / (float)MB;
    {
      float r_nlp = 0.0f, r_kl = 0.0f, r_bl = 0.0f, r_ent = 0.0f;
      if (tid < MB * 32) {
        const int s = tid / 32, a = tid % 32;
        // z of (sample s, action a): formed by the thread that uses it here; S.z is only for the dmu / dlogstd lanes after the barrier below
        const float zz = (a < A) ? (S.act[s][a] - S.mu[s][a]) / S.sgm[a] : 0.0f;
        S.z[s][a] = zz;
        if (a < A) {
          const float ls = S.ls[a], sg = S.sgm[a], mu = S.mu[s][a];
          r_nlp = 0.5f * zz * zz + ls;
          const float omu = S.omu[s][a], osg = S.osg[s][a];
          r_kl = logf(osg / sg + 1e-5f) + (sg * sg + (omu - mu) * (omu - mu)) / (2.0f * (osg * osg + 1e-5f)) - 0.5f;
          const float hi = fmaxf(mu - 1.1f, 0.0f), lo = fminf(mu + 1.1f, 0.0f);
          r_bl = hi * hi + lo * lo;
          r_ent = 0.5f + 0.5f * 1.8378770664093453f + ls;
        }
      }
      r_nlp = half_sum(r_nlp); r_kl = half_sum(r_kl); r_bl = half_sum(r_bl); r_ent = half_sum(r_ent);
      if (tid < MB * 32 && (tid % 32) == 31) {
        const int s = tid / 32;
        const float nlp = r_nlp + 0.5f * 1.8378770664093453f * (float)A;
        const float adv = pf_adv;
        const float ratio = expf(pf_nlp - nlp);
        const float L1 = -adv * ratio, L2 = -adv * clampf(ratio, 1.0f - D.e_clip, 1.0f + D.e_clip);
        const bool inr = ratio >= 1.0f - D.e_clip && ratio <= 1.0f + D.e_clip;
        S.gnlp[s] = (L1 > L2 || inr) ? adv * ratio : 0.0f;
        const float R = pf_ret, vo = pf_val;
        float closs[2];
        for (int j = 0; j < 2; ++j) {
          const float v = S.val[j][s];
          const float vc = vo + clampf(v - vo, -D.e_clip, D.e_clip);
          const float c1v = (v - R) * (v - R), c2v = (vc - R) * (vc - R);
          float d;
          if (D.clip_value) {
            closs[j] = fmaxf(c1v, c2v);
            const bool inv = fabsf(v - vo) <= D.e_clip;
            d = (c1v > c2v || inv) ? 2.0f * (v - R) : 0.0f;
          } else { closs[j] = c1v; d = 2.0f * (v - R); }
          S.dv[j][s] = (j == 0 ? 0.5f * D.critic_coef : 1.0f) * d * invM;
        }
        S.stat[s][1] = fmaxf(L1, L2); S.stat[s][2] = closs[0]; S.stat[s][3] = r_bl; S.stat[s][4] = r_kl;
        S.stat[s][5] = closs[1]; S.stat[s][6] = r_ent;
      }
    }
    SDX_LDS_BARRIER();   // LDS traffic only: the column view's loads stay in flight (a __syncthreads() would wait for them)
    if (tid < MB * 32) {
      const int s = tid / 32, a = tid % 32;
      float dmu = 0.0f;
      if (a < A) {
        const float sg = S.sgm[a], mu = S.mu[s][a];
        const float hi = fmaxf(mu - 1.1f, 0.0f), lo = fminf(mu + 1.1f, 0.0f);
        dmu = S.gnlp[s] * (-(S.z[s][a] / sg)) * invM + D.bounds_coef * (2.0f * hi + 2.0f * lo) * invM;
        if (g == 0) { __hip_atomic_store(&D.mb_mus[(r0 + s) * A + a], mu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&D.mb_sigmas[(r0 + s) * A + a], sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // update_mu_sigma (RC:1358)
      }
      S.dmu[s][a] = dmu;
    }
    if (tid >= 128 && tid < 128 + 32) {
      const int a = tid - 128;
      float dls = 0.0f;
      if (a < A) for (int s = 0; s < MB; ++s) dls += S.gnlp[s] * (1.0f - S.z[s][a] * S.z[s][a]) * invM;
      S.dls[a] = dls;
    }
    SDX_LDS_BARRIER();   // LDS traffic only: the column view's loads stay in flight (a __syncthreads() would wait for them)
    TS(12)
    // backward through the heads: lane (column k, row half) forms its part of dX3[.][k], the halves meet in LDS; elu' applied here
    {
      const int k = c2n;
      float a0[MB], a1[MB], a2[MB];
#pragma unroll
      for (int s = 0; s < MB; ++s) { a0[s] = 0.0f; a1[s] = 0.0f; a2[s] = 0.0f; }
      if (c2c == 0) {
#pragma unroll
        for (int j = 0; j < 13; ++j)
#pragma unroll
          for (int s = 0; s < MB; ++s) a0[s] += S.dmu[s][j] * wc[j];
      } else {
#pragma unroll
        for (int j = 0; j < A - 13; ++j)
#pragma unroll
          for (int s = 0; s < MB; ++s) a0[s] += S.dmu[s][13 + j] * wc[j];
#pragma unroll
        for (int s = 0; s < MB; ++s) {
          a1[s] = S.dv[0][s] * wc[A - 13]; a2[s] = S.dv[1][s] * wc[A - 12];
          S.dy2[0][s][k] = a0[s];
          S.dy2[1][s][k] = a1[s] * elu_g(S.x3[1][s][k]);
          S.dy2[2][s][k] = a2[s] * elu_g(S.x3[2][s][k]);
        }
      }
      __syncthreads();
      if (c2c == 0) {
#pragma unroll
        for (int s = 0; s < MB; ++s) S.dy2[0][s][k] = (S.dy2[0][s][k] + a0[s]) * elu_g(S.x3[0][s][k]);
      }
      __syncthreads();
    }
    TS(13)
    // backward L2 with the column copy: dY1[net][s][2g+c] = (sum_n dY2[net][s][n] W2[n][2g+c]) * elu'(x2[net][s][2g+c])
    {
      float p[24];
#pragma unroll
      for (int net = 0; net < 3; ++net)
#pragma unroll
        for (int s = 0; s < MB; ++s) {
          const float v = S.dy2[net][s][c2n] * c2[net];
          p[(net * MB + s) * 2 + 0] = c2c == 0 ? v : 0.0f;
          p[(net * MB + s) * 2 + 1] = c2c == 1 ? v : 0.0f;
        }
      block_sum<24>(S, p, S.part, tid, wave, lane);
      if (tid < MB * 2) {   // S.part: [(net MB + s) 2 + c]
        const int s = tid / 2, c = tid % 2;
        float v[3];
#pragma unroll
        for (int net = 0; net < 3; ++net) {
          v[net] = S.part[(net * MB + s) * 2 + c] * elu_g(S.x2[net][s][2 * g + c]);
          S.dyown[net][s][4 + c] = v[net];
        }
        lq_store(LQ, LQ_DY1 + (unsigned)s * U1 + 2 * g + c, v[0], v[1], v[2], tag);
      }
    }
    TS(14)
    u32x4 pdy1[4];
    bool pdy1_issued = false;
    // ---- (shadow of dY1) loss statistics + LR rule, Gram of x3, head / logstd terms of the squared gradient norm.  Spread over the
    // waves: the x3 Gram's per-thread part only occupies threads 0..255, so the serial pieces - the statistics / LR rule, the products
    // dmu_a . dmu_b and the column sums of the head terms - run on waves 4..6 in front of the SAME block reduction instead of in two more
    // barrier-separated phases after it (3.1 -> about 2 us: this shadow was the longest overrun of its edge, section 4b of DESIGN.md).
    if (tid == 384) {
      PLds::Ctl& C = S.ctl;
      float t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
      for (int s = 0; s < MB; ++s) { t1 += S.stat[s][1]; t2 += S.stat[s][2]; t3 += S.stat[s][3]; t4 += S.stat[s][4]; t5 += S.stat[s][5]; t6 += S.stat[s][6]; }
      const float kl = t4 * invM;
      C.sum_a += t1 * invM; C.sum_c += t2 * invM; C.sum_b += t3 * invM; C.sum_kl += kl; C.sum_cv += t5 * invM; C.sum_ent += t6 * invM;
      C.last_kl = kl;
      C.ac_lr_applied = C.ac_lr;   // the optimiser step of THIS minibatch uses the current lr; the schedule moves it afterwards
      if (D.adaptive_lr) {         // legacy schedule: after every minibatch (PS:306-312)
        // (the two thresholds are formed HERE from an opaque copy: hoisted out of the step loop they cost two registers for the whole launch,
        // and with the first look at the norm words in flight one of them was spilled and reloaded - a scratch load and a vmcnt(0) - in this shadow)
        float klt = D.kl_threshold;
        SDX_OPAQUE(klt);
        if (kl > 2.0f * klt) C.ac_lr = fmaxf(C.ac_lr / 1.5f, 1e-6f);
        if (kl < 0.5f * klt) C.ac_lr = fminf(C.ac_lr * 1.5f, 1e-2f);
      }
      if constexpr (!SINGLE) {
        // Adam step counters / bias corrections of the optimiser step that applies THIS minibatch's gradient (phase A of the next
        // iteration): nothing here depends on the gradient, so it left the chain (round 6; the double-precision powers and the two
        // divisions sat between the norm and the first Adam instruction).  S.scal[2..5] were last read by the Adam shadows of this
        // iteration, which every wave finished before the barrier in front of phase C
        C.ac_t += 1; C.cv_t += 1;
        C.ac_b1 *= 0.9; C.ac_b2 *= 0.999; C.cv_b1 *= 0.9; C.cv_b2 *= 0.999;
        S.scal[2] = C.ac_lr_applied / (float)(1.0 - C.ac_b1); S.scal[3] = 1.0f / sqrtf((float)(1.0 - C.ac_b2));
        S.scal[4] = C.cv_lr / (float)(1.0 - C.cv_b1); S.scal[5] = 1.0f / sqrtf((float)(1.0 - C.cv_b2));
      }
    }
    if constexpr (!SINGLE) {
      // Round 6: the Gram of x3 (waves 0..3, unit = tid) and the Gram of dY2 (waves 4..7, unit = tid - 256; it used to wait for the dY0 shadow, whose
      // block reduction is on the chain since the norm edge shrank) go through ONE pass: columns 0..29 / 32..61, each summed over its 16 rows.
      // The heads' terms are only STORED here (S.part[128..], [192..]); the serial sums that closed this shadow (16 + 23 dependent LDS reads on
      // three threads, about 1 us after the reduction) are lanes of the DPP sum at the end of the dY0 shadow now.
      float p[30];
#pragma unroll
      for (int i = 0; i < 30; ++i) p[i] = 0.0f;
      if (tid >= 256 && tid < 256 + 48) {              // thread (net, a, b): the heads' gradient product that multiplies G_x3[a][b]
        const int t = tid - 256, net = t / 16, a = (t / 4) % 4, b = t % 4;
        float dd = 0.0f;
        if (net == 0) { for (int j = 0; j < A; ++j) dd += S.dmu[a][j] * S.dmu[b][j]; }
        else dd = S.dv[net - 1][a] * S.dv[net - 1][b];
        S.part[128 + t] = dd;
      } else if (tid >= 320 && tid < 320 + 32) {       // bias of the policy head and logstd: |sum_s dmu_s[j]|^2 + dlogstd[j]^2
        const int j = tid - 320;
        float t = 0.0f;
        if (j < A) { float sb = 0; for (int a = 0; a < MB; ++a) sb += S.dmu[a][j]; t = sb * sb + S.dls[j] * S.dls[j]; }
        S.part[192 + j] = t;
      }
      if (tid < U2) {
#pragma unroll
        for (int net = 0; net < 3; ++net) gram_acc10(&p[net * 10], S.x3[net][0][tid], S.x3[net][1][tid], S.x3[net][2][tid], S.x3[net][3][tid]);
      } else {
        const int k = tid - U2;
#pragma unroll
        for (int net = 0; net < 3; ++net) gram_acc10(&p[net * 10], S.dy2[net][0][k], S.dy2[net][1][k], S.dy2[net][2][k], S.dy2[net][3][k]);
      }
      {
        float u[8];
        row_butterfly<30>(p, u, lane);
        rows_store<8>(S, u, tid < U2 ? 0 : 32, wave, lane);
      }
      SDX_LDS_BARRIER();
      if (tid < 64) {                                  // column tid: rows of waves 0..3 (x3) or 4..7 (dY2)
        const int r0 = tid < 32 ? 0 : 2 * NWV;
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 2 * NWV; ++w) t += S.red[r0 + w][tid];
        const int c = tid & 31, net = c / 10, e = c - net * 10;
        // entry e of the lower triangle -> both (hi, lo) and (lo, hi) of the 4x4 matrix
        const int hi = e < 1 ? 0 : (e < 3 ? 1 : (e < 6 ? 2 : 3)), lo = e - hi * (hi + 1) / 2;
        if (c < 30) {
          float* dst = tid < 32 ? S.gx[net][3] : S.gd2[net];
          dst[hi * 4 + lo] = t; dst[lo * 4 + hi] = t;
        }
      }
      // first look at this lane's dY1 words (the polling gather of phase E takes over if a producer is late)
      lq_peek<4>(LQ, LQ_DY1 + tid, NTH, pdy1); pdy1_issued = true;
      SDX_LDS_BARRIER();
    }
    TS(15)
    refresh();
    // ================================================================== phase E: gather dY1, backward L1
    {
      float v0[4], v1[4], v2[4];
      if (!(pdy1_issued && lq_take<4>(pdy1, tag, v0, v1, v2)) && !lq_gather<4>(LQ, LQ_DY1 + tid, NTH, tag, v0, v1, v2, failflag)) S.fail = 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) { (&S.dy1[0][0][0])[tid + NTH * j] = v0[j]; (&S.dy1[1][0][0])[tid + NTH * j] = v1[j]; (&S.dy1[2][0][0])[tid + NTH * j] = v2[j]; }
    }
    TS(16)
    __syncthreads();
    {
      float p[48];
#pragma unroll
      for (int net = 0; net < 3; ++net)
#pragma unroll
        for (int s = 0; s < MB; ++s) {
          const float d = S.dy1[net][s][tid];
#pragma unroll
          for (int c = 0; c < 4; ++c) p[(net * MB + s) * 4 + c] = d * c1[net][c];
        }
      block_sum<48>(S, p, S.part, tid, wave, lane);
      if (tid < MB * 4) {   // S.part: [(net MB + s) 4 + c]
        const int s = tid / 4, c = tid % 4;
        float v[3];
#pragma unroll
        for (int net = 0; net < 3; ++net) {
          v[net] = S.part[(net * MB + s) * 4 + c] * elu_g(S.x1[net][s][4 * g + c]);
          S.dyown[net][s][c] = v[net];
        }
        if constexpr (SINGLE) lq_store(LQ, LQ_DY0 + (unsigned)s * U0 + 4 * g + c, v[0], v[1], v[2], tag);
      }
      if constexpr (!SINGLE) {
        // the other CUs need dY0 only for the gradient norm: publish this CU's share of the Gram dY0 dY0^T (its four columns).
        // Lanes 0..15 of wave 0 wrote dyown above; the same wave reads it back (LDS keeps program order within a wave)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (wave == 0) {
          // lane (a, b) < 16: entry (a, b) of the CU's 4x4 Gram of dY0 (its four columns) times (G_x0[a][b] + 1): the weight-gradient term
          // <G_dY0, G_x0> and the bias term |sum_s dY0_s|^2 = sum of all entries of the Gram; a DPP row sum, ONE word per CU
          float t[3] = {0.0f, 0.0f, 0.0f};
          if (lane < 16) {
            const int hi = lane >> 2, lo = lane & 3;
#pragma unroll
            for (int net = 0; net < 3; ++net) {
              float gg = 0.0f;
#pragma unroll
              for (int c = 0; c < 4; ++c) gg += S.dyown[net][hi][c] * S.dyown[net][lo][c];
              t[net] = gg * S.gx[net][0][lane] + gg;
            }
          }
#pragma unroll
          for (int net = 0; net < 3; ++net) {
            float x = t[net];
            x = dpp_add<0xB1, 0xF>(x); x = dpp_add<0x4E, 0xF>(x); x = dpp_add<0x141, 0xF>(x); x = dpp_add<0x140, 0xF>(x);   // 16-lane row sums
            t[net] = x;
          }
          if (lane == 0) lq_store(LQ, LQ_G0 + g, t[0], t[1], t[2], tag);
        }
      }
    }
    TS(17)
    if constexpr (SINGLE) {
      // ---- hand the minibatch over to the multi-rank exchange: rank-MB factors -> D.fact (layout SdxpFactOff), CU g writes slice g
      __syncthreads();
      float* F = D.fact;
      const int i = g * NTH + tid, stride = NWG * NTH;
      for (int j = i; j < MB * D.obs_dim; j += stride) {
        const float v = S.obs[j / D.obs_dim][j % D.obs_dim];
        F[D.foff.x[0][0] + j] = v; F[D.foff.x[1][0] + j] = v;
      }
      for (int j = i; j < MB * ST; j += stride) F[D.foff.x[2][0] + j] = (&S.cvx[0][0])[j];
#pragma unroll
      for (int net = 0; net < 3; ++net) {
        for (int j = i; j < MB * U0; j += stride) F[D.foff.x[net][1] + j] = (&S.x1[net][0][0])[j];
        for (int j = i; j < MB * U1; j += stride) { F[D.foff.x[net][2] + j] = (&S.x2[net][0][0])[j]; F[D.foff.dy[net][1] + j] = (&S.dy1[net][0][0])[j]; }
        for (int j = i; j < MB * U2; j += stride) { F[D.foff.h[net] + j] = (&S.x3[net][0][0])[j]; F[D.foff.dy[net][2] + j] = (&S.dy2[net][0][0])[j]; }
      }
      if (i < MB * U0) {   // dY0: every CU produced four columns of it; the first 8 CUs collect the packed words
        float v0[1], v1[1], v2[1];
        if (!lq_gather<1>(LQ, LQ_DY0 + i, 1, tag, v0, v1, v2, failflag)) S.fail = 1;
        F[D.foff.dy[0][0] + i] = v0[0]; F[D.foff.dy[1][0] + i] = v1[0]; F[D.foff.dy[2][0] + i] = v2[0];
      }
      if (g == 0) {
        if (tid < MB * 34) {
          const int s = tid / 34, c = tid % 34;
          F[D.foff.dh + tid] = c < A ? S.dmu[s][c] : (c == 32 ? S.dv[0][s] : (c == 33 ? S.dv[1][s] : 0.0f));
        }
        if (tid >= 192 && tid < 192 + 32) F[D.foff.dls + tid - 192] = tid - 192 < A ? S.dls[tid - 192] : 0.0f;
        if (tid == 0) {   // what k_ctrl does after a minibatch in explicit (multi-rank) mode: statistics, KL words, cursor
          const PLds::Ctl& C = S.ctl;
          gctl->sum_a_loss += C.sum_a; gctl->sum_c_loss += C.sum_c; gctl->sum_b_loss += C.sum_b; gctl->sum_kl += C.sum_kl;
          gctl->sum_cv_loss += C.sum_cv; gctl->sum_entropy += C.sum_ent; gctl->n_mb += 1; gctl->last_kl = C.last_kl;
          gctl->ac_pending = 0; gctl->cv_pending = 0;
          F[D.foff.kl] = C.last_kl; D.ac_g[D.g_tail] = C.last_kl;
          gctl->prev_mb = gctl->mb_index; gctl->prev_mini_epoch = gctl->mini_epoch;
          int mbn = gctl->mb_index + 1;
          if (mbn >= D.num_minibatches) { mbn = 0; gctl->mini_epoch += 1; }
          gctl->mb_index = mbn;
          gctl->step += 1;
          gctl->ll_tag = tag_base + 3u;
        }
      }
      return;
    }
    if constexpr (!SINGLE)
    {   // ---- (shadow of dY0; on the step's chain since the norm edge shrank to one word per CU) Gram of dY1 - the Gram of dY2 moved into the dY1
        // shadow (round 6) -, this lane's layer-0 gradient elements, the next step's rows, and the part of the squared norm that does not wait for dY0
      {
        float p[30], ua[8];
#pragma unroll
        for (int i = 0; i < 30; ++i) p[i] = 0.0f;
#pragma unroll
        for (int net = 0; net < 3; ++net) gram_acc10(&p[net * 10], S.dy1[net][0][tid], S.dy1[net][1][tid], S.dy1[net][2][tid], S.dy1[net][3][tid]);
        row_butterfly<30>(p, ua, lane);
        rows_store<8>(S, ua, 0, wave, lane);
      }
      SDX_LDS_BARRIER();          // (rows_reduce, opened up: the stage between its two barriers occupies 32 threads ...)
      lq_peek<4>(LQ, LQ_G0 + lane, 64, pg0); pg0_issued = true;   // (0.7 us into the shadow: the words left their CUs 0.7 + 2.0 (phase E) us ago; the round trip ends with the shadow)
      if (tid < 32) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 4 * NWV; ++w) t += S.red[w][tid];
        S.part[tid] = t;
      }
      // ... so this lane's layer-0 gradient elements are formed here, by every wave (dyown[.][.][0..3] were published by the barrier above;
      // S.obs / S.cvx still hold this minibatch): 4 + 5 batches of four independent LDS loads instead of 13 dependent round trips on the chain
      {
        float da[MB], dc[MB], dvv[MB];
#pragma unroll
        for (int s = 0; s < MB; ++s) { da[s] = S.dyown[0][s][r0w]; dc[s] = S.dyown[1][s][r0w]; dvv[s] = S.dyown[2][s][r0w]; }
#pragma unroll
        for (int i = 0; i < I0A; ++i) {
          const int kk = lane + 64 * i, k = h0 * H0A + kk;
          const bool ok = kk < H0A && k < D.obs_dim;
          const int kc = ok ? k : 0;
          float ga = 0.0f, gc = 0.0f;
#pragma unroll
          for (int s = 0; s < MB; ++s) { const float x = S.obs[s][kc]; ga += da[s] * x; gc += dc[s] * x; }
          g0a[i] = ok ? ga : 0.0f; g0c[i] = ok ? gc : 0.0f;
          m0a[i] *= 0.9f; v0a[i] *= 0.999f; m0c[i] *= 0.9f; v0c[i] *= 0.999f;   // (adam1p: off the chain)
        }
#pragma unroll
        for (int i = 0; i < I0V; ++i) {
          const int kk = lane + 64 * i, k = h0 * H0V + kk;
          const bool ok = kk < H0V;
          const int kc = ok ? k : 0;
          float gv = 0.0f;
#pragma unroll
          for (int s = 0; s < MB; ++s) gv += dvv[s] * S.cvx[s][kc];
          g0v[i] = ok ? gv : 0.0f;
          m0v[i] *= 0.9f; v0v[i] *= 0.999f;
        }
      }
      SDX_LDS_BARRIER();
      if (step + 1 < total_steps) { const bool wrap = mbi + 1 >= D.num_minibatches; load_rows(wrap ? 0 : mbi + 1, mini_epoch + (wrap ? 1 : 0)); }
      if (wave == 1) {
        // lane (net, i) < 48: entry i of the Grams of dY1 / dY2 against (G_x1 / G_x2 + 1) - weight and bias gradients of trunk layers 1, 2 -, of the
        // Gram of x3 against the heads' gradient products (S.part[128..], dY1 shadow), + the head-bias / logstd terms (32 entries over the 16 lanes
        // of net 0; |sum_s dV_s|^2 on lane 0 of nets 1, 2); a DPP row sum per network
        float t = 0.0f;
        if (lane < 48) {
          const int net = lane >> 4, i = lane & 15;
          t = S.part[net * 10 + tri16(i)] * (S.gx[net][1][i] + 1.0f) + S.gd2[net][i] * (S.gx[net][2][i] + 1.0f) + S.gx[net][3][i] * S.part[128 + net * 16 + i];
          if (net == 0) t += S.part[192 + i] + S.part[208 + i];
          else if (i == 0) { float sb = 0.0f; for (int a = 0; a < MB; ++a) sb += S.dv[net - 1][a]; t += sb * sb; }
        }
        t = dpp_add<0xB1, 0xF>(t); t = dpp_add<0x4E, 0xF>(t); t = dpp_add<0x141, 0xF>(t); t = dpp_add<0x140, 0xF>(t);   // 16-lane row sums
        if (lane < 48 && (lane & 15) == 0) S.n2rest[lane >> 4] = t;
      } else if (tid >= 128 && tid < 128 + 3 * MB) { const int net = (tid - 128) / MB, s = (tid - 128) % MB; S.dyown[net][s][6] = S.dy2[net][s][g]; }
    }
    TS(18)
    pending = true;
  }

  if (stamps && g == (stamps >> 8) && tid < 32) D.dbg[tid] = S.tacc[tid];
  // ------------------------------------------------------------------ epilogue: resident parameters back to HBM
  refresh();
#pragma unroll
  for (int i = 0; i < I0A; ++i) {
    const int kk = lane + 64 * i, k = h0 * H0A + kk;
    if (kk < H0A && k < D.obs_dim) {
      const size_t oa = woff(0, 0) + (size_t)n0 * D.obs_dim + k, oc = woff(1, 0) + (size_t)n0 * D.obs_dim + k;
      D.ac[oa] = w0a[i]; D.ac_m[oa] = m0a[i]; D.ac_v[oa] = v0a[i];
      D.ac[oc] = w0c[i]; D.ac_m[oc] = m0c[i]; D.ac_v[oc] = v0c[i];
    }
  }
#pragma unroll
  for (int i = 0; i < I0V; ++i) {
    const int kk = lane + 64 * i, k = h0 * H0V + kk;
    if (kk < H0V) { const size_t o = woff(2, 0) + (size_t)n0 * ST + k; D.cv[o] = w0v[i]; D.cv_m[o] = m0v[i]; D.cv_v[o] = v0v[i]; }
  }
#pragma unroll
  for (int net = 0; net < 3; ++net) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t o = woff(net, 1) + (size_t)r1 * U0 + q1 * 256 + lane + 64 * i;
      P_of(net)[o] = w1[net][i]; M_of(net)[o] = m1[net][i]; V_of(net)[o] = v1[net][i];
    }
    {
      const size_t o = woff(net, 2) + (size_t)g * U1 + wave * 64 + lane;
      P_of(net)[o] = w2[net]; M_of(net)[o] = m2[net]; V_of(net)[o] = v2[net];
    }
  }
  if (tid < HPC) {
    const int net = head_net(hrow);
    const size_t o = head_woff(hrow) + hk;
    P_of(net)[o] = hw; M_of(net)[o] = hm; V_of(net)[o] = hv;
  }
  if (g == 0) {
    if (tid >= 21 && tid < 21 + A + 2) {
      const int row = tid - 21, net = head_net(row);
      const size_t o = head_boff(row);
      P_of(net)[o] = S.bias[tid]; M_of(net)[o] = S.bias_m[tid]; V_of(net)[o] = S.bias_v[tid];
    }
    if (tid >= 64 && tid < 64 + A) {
      const int a = tid - 64;
      const size_t o = D.off.logstd + a;
      D.ac[o] = s_b2[0][a]; D.ac_m[o] = s_b2[1][a]; D.ac_v[o] = s_b2[2][a];
    }
    if (tid == 0) {
      const PLds::Ctl& C = S.ctl;
      gctl->ac_lr = C.ac_lr; gctl->ac_t = C.ac_t; gctl->cv_t = C.cv_t;
      gctl->ac_b1pow = C.ac_b1; gctl->ac_b2pow = C.ac_b2; gctl->cv_b1pow = C.cv_b1; gctl->cv_b2pow = C.cv_b2;
      gctl->sum_a_loss = C.sum_a; gctl->sum_c_loss = C.sum_c; gctl->sum_b_loss = C.sum_b; gctl->sum_kl = C.sum_kl;
      gctl->sum_cv_loss = C.sum_cv; gctl->sum_entropy = C.sum_ent; gctl->n_mb = total_steps; gctl->last_kl = C.last_kl;
      gctl->ac_gnorm = C.ac_gn; gctl->cv_gnorm = C.cv_gn; gctl->ac_pending = 0; gctl->cv_pending = 0;
      gctl->mb_index = 0; gctl->mini_epoch = total_steps / D.num_minibatches;
      gctl->ll_tag = tag_base + (unsigned)total_steps + 2u;
    }
  }
  if (tid < 12) {
    const int net = tid / 4, w = tid % 4;
    const size_t o = boff(net, 0) + 4 * g + w;
    P_of(net)[o] = S.bias[tid]; M_of(net)[o] = S.bias_m[tid]; V_of(net)[o] = S.bias_v[tid];
  } else if (tid < 18) {
    const int net = (tid - 12) / 2, r = (tid - 12) % 2;
    const size_t o = boff(net, 1) + 2 * g + r;
    P_of(net)[o] = S.bias[tid]; M_of(net)[o] = S.bias_m[tid]; V_of(net)[o] = S.bias_v[tid];
  } else if (tid < 21) {
    const int net = tid - 18;
    const size_t o = boff(net, 2) + g;
    P_of(net)[o] = S.bias[tid]; M_of(net)[o] = S.bias_m[tid]; V_of(net)[o] = S.bias_v[tid];
  }
}

extern "C" size_t sdxpk_persist_lds_bytes() { return sizeof(PLds); }
extern "C" int sdxpk_persist_supported(const SdxpDev* D, int minibatch, int n_cus) {
  return minibatch == MB && D->obs_dim <= OBS && D->obs_dim % 4 == 0 && D->state_dim == ST && D->units[0] == U0 && D->units[1] == U1 &&
         D->units[2] == U2 && D->act_dim == ACT && n_cus >= NWG;
}
extern "C" int sdxpk_update_persistent(const SdxpDev* D, int total_steps, unsigned* failflag, hipStream_t st) {
  static bool attr = false;
  // SDXP_PERSIST_STAMPS=1: phase clock of CU 0; SDXP_PERSIST_STAMP_CU=<g> picks another CU (bits 8.. of the kernel's flag)
  static const int stamps = (getenv("SDXP_PERSIST_STAMPS") && getenv("SDXP_PERSIST_STAMPS")[0] == '1')
                                ? (1 | ((getenv("SDXP_PERSIST_STAMP_CU") ? (atoi(getenv("SDXP_PERSIST_STAMP_CU")) & 255) : 0) << 8)) : 0;
  const char* fe = getenv("SDXP_PERSIST_FAULT");   // read per call: the failure-path test sets and clears it
  const int fault = (fe && fe[0] == '1') ? 1 : 0;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_update_persistent<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sizeof(PLds)) != hipSuccess) return -1;
    attr = true;
  }
  if (hipMemsetAsync(failflag, 0, sizeof(unsigned), st) != hipSuccess) return -1;
  hipLaunchKernelGGL(k_update_persistent<false>, dim3(NWG), dim3(NTH), sizeof(PLds), st, *D, total_steps, nullptr, failflag, stamps, fault);
  return 0;
}
// forward + backward of the minibatch under the device cursor -> D.fact (multi-rank path); the fail flag is sticky (not cleared here)
extern "C" int sdxpk_fwd_bwd_persistent(const SdxpDev* D, unsigned* failflag, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_update_persistent<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sizeof(PLds)) != hipSuccess) return -1;
    attr = true;
  }
  hipLaunchKernelGGL(k_update_persistent<true>, dim3(NWG), dim3(NTH), sizeof(PLds), st, *D, 1, nullptr, failflag, 0, 0);
  return 0;
}
