// sdx_const_build.h — host side: the derived tables of SdxConst from a scene description (used by sdx_create; the SIMT-emulator
// driver under tests/hipemu includes the same function so that emulated kernels see the tables the GPU sees).
#pragma once
#include <math.h>
#include <string.h>

#include "sdx_common.h"

static inline void sdx_build_const(const sdx_scene_desc* scene, SdxConst* out) {
  SdxConst& K = *out;
  memset(&K, 0, sizeof(K));
  K.sc = *scene;
  K.max_depth = 0;
  for (int k = 0; k < SDX_NLINK; ++k) {
    const int p = scene->parent[k];
    K.anc[k] = (k == 0) ? 0u : (K.anc[p] | (1u << (k - 1)));
    K.depth[k] = (k == 0) ? 0 : K.depth[p] + 1;
    if (K.depth[k] > K.max_depth) K.max_depth = K.depth[k];
  }
  for (int t = 0; t < SDX_NBRICK_TYPES; ++t) {
    const float* hh = scene->brick_half[t];
    K.brick_radius[t] = sqrtf(hh[0] * hh[0] + hh[1] * hh[1] + hh[2] * hh[2]);
  }
  for (int r = 0; r < scene->n_rbox; ++r) {
    const float* hh = scene->rbox_half[r];
    K.rbox_radius[r] = sqrtf(hh[0] * hh[0] + hh[1] * hh[1] + hh[2] * hh[2]);
  }
  for (int j = 0; j < SDX_NDOF; ++j) {
    if (j < 7) K.hand_reset_pose[j] = scene->arm_prepare_pose[j];
    else K.hand_reset_pose[j] = 0.5f * (scene->finger_reset_unscaled[j - 7] + 1.0f) * (scene->upper[j] - scene->lower[j]) + scene->lower[j];
  }
}
