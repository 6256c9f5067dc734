// libmst.so: version / error plumbing shared by all entry points.
#include "common.h"

namespace mst {

char* err_buf() {
  static thread_local char buf[512] = "";
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace mst

extern "C" {
int mst_version(void) { return MST_ABI_VERSION; }
const char* mst_last_error(void) { return mst::err_buf(); }
}
