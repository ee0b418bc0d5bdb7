// sdx_camera.hip — segmentation "camera" of BlockAssemblySearch (SE = tasks/block_assembly/allegro_hand_block_assembly_search.py).
// The reference renders IMAGE_SEGMENTATION with Isaac Gym's camera sensor (128 x 128, fixed pose, SE:755-757,873-878) and uses only two
// things of the image: how many pixels carry the target brick's segmentation id and where their centroid is (SE:1232-1241,1640-1646).
// Here the image is ray-cast against the scene's boxes (bricks = their bounding boxes with segmentation id brick index + 1 (SE:840),
// table / bin / robot = id 0): one thread per pixel, boxes of the env staged in LDS as (centre, rotation matrix, half extents, ray
// origin in the box frame), nearest hit wins.  Box geometry instead of the studded meshes is the approximation of this whole build
// (DESIGN.md section 3); the pixel counts are therefore NOT comparable digit by digit with Isaac Gym's renderer (parity unpinned) -
// the kernel is checked against oracle/camera_oracle.py, a numpy ray caster of the same boxes.
#include "sdx_common.h"

#define CAM_W 128
#define CAM_H 128
#define CAM_MAXBOX (SDX_NBRICK + SDX_MAX_STATIC + SDX_MAX_RBOX)

struct CamBox { float c[3]; float m[9]; float h[3]; float o[3]; int id; };   // m: rows = box axes in world coordinates; o = ray origin in the box frame

__device__ __forceinline__ void quat_rows(f4 q, float* m) {
  const f3 x = qrot(q, F3(1, 0, 0)), y = qrot(q, F3(0, 1, 0)), z = qrot(q, F3(0, 0, 1));
  m[0] = x.x; m[1] = x.y; m[2] = x.z; m[3] = y.x; m[4] = y.y; m[5] = y.z; m[6] = z.x; m[7] = z.y; m[8] = z.z;
}

// grid (CAM_H * CAM_W / 256, N); stats[e] = {count, sum of rows, sum of columns, 0} of the pixels showing the target brick
__global__ __launch_bounds__(256) void k_seg_camera(const SdxConst* __restrict__ C, SdxBuf B, int32_t* __restrict__ stats, int16_t* __restrict__ image) {
  __shared__ CamBox s_box[CAM_MAXBOX];
  __shared__ int s_acc[3];
  const sdx_scene_desc& sc = C->sc;
  const int e = blockIdx.y, tid = threadIdx.x;
  const float* root_e = B.root + (size_t)e * SDX_ACTORS * 13;
  const float* rb_e = B.rb + (size_t)e * SDX_BODIES * 13;
  const int ns = sc.n_static, nr = sc.n_rbox, nbox = SDX_NBRICK + ns + nr;
  const f3 cam = ld3(sc.seg_cam_pos);
  for (int i = tid; i < nbox; i += 256) {
    CamBox& b = s_box[i];
    f3 c, h;
    f4 q = {0.0f, 0.0f, 0.0f, 1.0f};
    int id = 0;
    if (i < SDX_NBRICK) {
      const float* r = root_e + (SDX_ACTOR_BRICK0 + i) * 13;
      const int t = sc.brick_type[i];
      q = ld4(r + 3);
      c = ld3(r) + qrot(q, ld3(sc.brick_center[t]));
      h = ld3(sc.brick_half[t]);
      id = i + 1;                                                               // segmentationId = lego_i + 1, SE:840
    } else if (i < SDX_NBRICK + ns) {
      c = ld3(sc.static_center[i - SDX_NBRICK]); h = ld3(sc.static_half[i - SDX_NBRICK]);
    } else {
      const int k = i - SDX_NBRICK - ns, l = sc.rbox_link[k];
      const f4 ql = ld4(rb_e + l * 13 + 3);
      c = ld3(rb_e + l * 13) + qrot(ql, ld3(sc.rbox_center[k]));
      q = qmul(ql, ld4(sc.rbox_quat[k]));
      h = ld3(sc.rbox_half[k]);
    }
    b.c[0] = c.x; b.c[1] = c.y; b.c[2] = c.z;
    quat_rows(q, b.m);
    b.h[0] = h.x; b.h[1] = h.y; b.h[2] = h.z;
    const f3 d = cam - c;
    b.o[0] = b.m[0] * d.x + b.m[1] * d.y + b.m[2] * d.z;
    b.o[1] = b.m[3] * d.x + b.m[4] * d.y + b.m[5] * d.z;
    b.o[2] = b.m[6] * d.x + b.m[7] * d.y + b.m[8] * d.z;
    b.id = id;
  }
  if (tid < 3) s_acc[tid] = 0;
  __syncthreads();
  // pinhole camera looking from seg_cam_pos at seg_cam_target, world z up; row 0 is the top of the image
  const f3 tgt = ld3(sc.seg_cam_target);
  f3 f = tgt - cam;
  f = f * (1.0f / sqrtf(dot(f, f)));
  f3 r = cross(f, F3(0.0f, 0.0f, 1.0f));
  r = r * (1.0f / sqrtf(dot(r, r)));
  const f3 u = cross(r, f);
  const float th = tanf(0.5f * sc.seg_cam_hfov_deg * 0.017453292519943295f);
  const int pix = blockIdx.x * 256 + tid, row = pix / CAM_W, col = pix % CAM_W;
  const float px = (2.0f * ((float)col + 0.5f) / (float)CAM_W - 1.0f) * th;
  const float py = (1.0f - 2.0f * ((float)row + 0.5f) / (float)CAM_H) * th;    // square image: the vertical extent equals the horizontal one
  const f3 d = f + r * px + u * py;
  float best = 3.0e38f;
  int best_id = 0;
  for (int i = 0; i < nbox; ++i) {
    const CamBox& b = s_box[i];
    float tmin = 0.0f, tmax = 3.0e38f;
    bool hit = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float da = b.m[3 * a] * d.x + b.m[3 * a + 1] * d.y + b.m[3 * a + 2] * d.z;
      const float oa = b.o[a], ha = b.h[a];
      if (fabsf(da) < 1e-12f) { if (fabsf(oa) > ha) hit = false; }
      else {
        const float inv = 1.0f / da;
        float t0 = (-ha - oa) * inv, t1 = (ha - oa) * inv;
        if (t0 > t1) { const float tt = t0; t0 = t1; t1 = tt; }
        tmin = fmaxf(tmin, t0); tmax = fminf(tmax, t1);
      }
    }
    if (hit && tmin <= tmax && tmin < best) { best = tmin; best_id = b.id; }
  }
  if (image) image[((size_t)e * CAM_H + row) * CAM_W + col] = (int16_t)best_id;
  const int target = seg_actor(e) - SDX_ACTOR_BRICK0 + 1;                       // segmentation_id_list[i], SE:846-847
  if (best_id == target) { atomicAdd(&s_acc[0], 1); atomicAdd(&s_acc[1], row); atomicAdd(&s_acc[2], col); }
  __syncthreads();
  if (tid < 3 && s_acc[tid]) atomicAdd(&stats[(size_t)e * 4 + tid], s_acc[tid]);
}

// pixel statistics (SE:1232-1241) and the emergence reward (SE:1640-1646) from the accumulated sums
__global__ void k_seg_finalize(SdxBuf B, const int32_t* __restrict__ stats) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B.N) return;
  const int n = stats[(size_t)e * 4];
  int cx = 0, cy = 0;
  if (n > 0) { cx = (int)((float)stats[(size_t)e * 4 + 1] / (float)n); cy = (int)((float)stats[(size_t)e * 4 + 2] / (float)n); }
  B.seg_pix[(size_t)e * 4 + 0] = (float)n;
  B.seg_pix[(size_t)e * 4 + 1] = (float)cx;
  B.seg_pix[(size_t)e * 4 + 2] = (float)cy;
  const float last = B.seg_pix[(size_t)e * 4 + 3];
  B.emergence[e] = ((float)n - last) * 5.0f;                                    // SE:1645
  B.seg_pix[(size_t)e * 4 + 3] = (float)n;                                      // last_emergence_pixel, SE:1646
}

extern "C" void sdxk_seg_camera(const SdxConst* C, const SdxBuf* B, hipStream_t st) {
  (void)hipMemsetAsync(B->seg_stats, 0, (size_t)B->N * 4 * sizeof(int32_t), st);
  hipLaunchKernelGGL(k_seg_camera, dim3(CAM_H * CAM_W / 256, B->N), dim3(256), 0, st, C, *B, B->seg_stats, B->seg_image);
  hipLaunchKernelGGL(k_seg_finalize, dim3((B->N + 255) / 256), dim3(256), 0, st, *B, B->seg_stats);
}
