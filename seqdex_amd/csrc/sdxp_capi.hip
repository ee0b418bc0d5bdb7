// sdxp_capi.hip — PPO side of the C ABI (placeholder while the kernels are being written)
#include "sdx_common.h"
struct sdxp_agent { int dummy; };
extern "C" {
int sdxp_create(const sdxp_config*, int32_t, uint64_t, sdxp_handle*) { return SDX_ERR_STATE; }
int sdxp_destroy(sdxp_handle) { return SDX_ERR_STATE; }
int sdxp_tensor(sdxp_handle, int32_t, void**, int64_t*, int32_t*, int32_t*) { return SDX_ERR_STATE; }
int64_t sdxp_param_count(sdxp_handle, int32_t) { return 0; }
int sdxp_act(sdxp_handle, int32_t, const float*, const float*, const float*, const float*, float*, void*) { return SDX_ERR_STATE; }
int sdxp_store_rewards(sdxp_handle, int32_t, const float*, void*) { return SDX_ERR_STATE; }
int sdxp_finish_rollout(sdxp_handle, const float*, const float*, void*) { return SDX_ERR_STATE; }
int sdxp_update(sdxp_handle, void*) { return SDX_ERR_STATE; }
int sdxp_backward(sdxp_handle, int32_t, int32_t, void*) { return SDX_ERR_STATE; }
int sdxp_apply(sdxp_handle, int32_t, float, void*) { return SDX_ERR_STATE; }
const char* sdxp_last_error(sdxp_handle) { return "PPO kernels not built yet"; }
}
