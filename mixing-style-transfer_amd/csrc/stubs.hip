// Entry points whose kernels are not built yet: exported so the ABI is complete, fail loudly.
#include "common.h"
extern "C" {
int mst_encoder_create(mst_encoder** out, const mst_encoder_config*, const mst_encoder_weights*) {
  if (out) *out = nullptr;
  return mst::fail(MST_EINVAL, "mst_encoder_create: not implemented in this build");
}
void mst_encoder_destroy(mst_encoder*) {}
size_t mst_encoder_workspace_bytes(const mst_encoder*, int, int) { return 0; }
int mst_encoder_forward(const mst_encoder*, const float*, int, const float*, int, float*, const mst_encoder_taps*,
                        void*, size_t, void*) {
  return mst::fail(MST_EINVAL, "mst_encoder_forward: not implemented in this build");
}
size_t mst_aug_workspace_bytes(int, int, int) { return 0; }
int mst_aug_apply(const mst_aug_clip*, int, int, float*, const float*, int, void*, size_t, void*) {
  return mst::fail(MST_EINVAL, "mst_aug_apply: not implemented in this build");
}
size_t mst_infonce_workspace_bytes(int, int) { return 0; }
int mst_infonce_forward(const float*, const int64_t*, int, int, int, int, float, float*, void*, size_t, void*) {
  return mst::fail(MST_EINVAL, "mst_infonce_forward: not implemented in this build");
}
}
