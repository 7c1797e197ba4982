// hx_core.hip — error string and version of libhx_mi355.so
#include "hx_common.h"

namespace hx {
char* error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace hx

extern "C" {
const char* hx_last_error(void) { return hx::error_buffer(); }
int hx_version(void) { return 100; }
}
