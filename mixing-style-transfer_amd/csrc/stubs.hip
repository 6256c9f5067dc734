// Entry points whose kernels are not built yet: exported so the ABI is complete, fail loudly.
#include "common.h"
extern "C" {
size_t mst_aug_workspace_bytes(int, int, int) { return 0; }
int mst_aug_apply(const mst_aug_clip*, int, int, float*, const float*, int, void*, size_t, void*) {
  return mst::fail(MST_EINVAL, "mst_aug_apply: not implemented in this build");
}
size_t mst_infonce_workspace_bytes(int, int) { return 0; }
int mst_infonce_forward(const float*, const int64_t*, int, int, int, int, float, float*, void*, size_t, void*) {
  return mst::fail(MST_EINVAL, "mst_infonce_forward: not implemented in this build");
}
}
