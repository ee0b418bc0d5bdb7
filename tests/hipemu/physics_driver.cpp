// TEST INFRASTRUCTURE ONLY: runs seqdex_amd/csrc/sdx_physics.hip (the product's kernel source, compiled by g++ against the SIMT
// emulator) for N envs on the CPU.  Same argument layout as oracle/physics_oracle.c::sdxo_simulate so that the two can be compared.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "sdx_common.h"
#include "sdx_const_build.h"

extern "C" void sdxk_physics(const SdxConst* C, const SdxBuf* B, hipStream_t st);
extern "C" void sdxk_kinematics(const SdxConst* C, const SdxBuf* B, hipStream_t st);
extern "C" size_t sdxk_physics_lds_bytes();

static int32_t g_cstats[4];   // SDX_T_CONTACT_STATS of the emulated launches since the last emu_cstats(reset = 1)
extern "C" void emu_cstats(int32_t* out, int reset) {
  memcpy(out, g_cstats, sizeof(g_cstats));
  if (reset) memset(g_cstats, 0, sizeof(g_cstats));
}
// wcount / wkey / wlam: the warm-start cache of the N envs ([N], [N, SDX_MAXC], [N, 3, SDX_MAXC]); NULL = an empty cache for this call
extern "C" int emu_simulate(const sdx_scene_desc* sc, int N, float* root, float* dof, const float* targets, float* rb, float* contact,
                            float* jac, int* ncontacts, long long* dbg, int* wcount, unsigned* wkey, float* wlam) {
  static SdxConst K;
  sdx_build_const(sc, &K);
  SdxBuf B;
  memset(&B, 0, sizeof(B));
  B.N = N; B.K = 1; B.task_kind = sc->task_kind; B.obs_w = SDX_NUM_OBS;
  B.root = root; B.dof = dof; B.targets = const_cast<float*>(targets); B.rb = rb; B.contact = contact; B.jac = jac;
  B.ncontacts = ncontacts;
  std::vector<long long> d(64, 0);
  B.dbg = dbg ? dbg : d.data();
  std::vector<int32_t> tc;
  std::vector<uint32_t> tk;
  std::vector<float> tl;
  if (!wcount) {
    tc.assign(N, 0); tk.assign((size_t)N * SDX_MAXC, 0u); tl.assign((size_t)N * 3 * SDX_MAXC, 0.0f);
    wcount = tc.data(); wkey = tk.data(); wlam = tl.data();
  }
  B.wcount = wcount; B.wkey = wkey; B.wlam = wlam;
  B.cstats = g_cstats;
  sdxk_physics(&K, &B, nullptr);
  return (int)sdxk_physics_lds_bytes();
}
extern "C" void emu_kinematics(const sdx_scene_desc* sc, int N, float* dof, float* rb, float* jac) {
  static SdxConst K;
  sdx_build_const(sc, &K);
  SdxBuf B;
  memset(&B, 0, sizeof(B));
  B.N = N; B.dof = dof; B.rb = rb; B.jac = jac;
  std::vector<long long> d(64, 0);
  B.dbg = d.data();
  sdxk_kinematics(&K, &B, nullptr);
}
