// miq C ABI: version and the thread-local error text.
#include "miq_common.h"

namespace miq {

char* error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace miq

extern "C" int miq_version(void) { return MIQ_VERSION; }

extern "C" const char* miq_last_error(void) { return miq::error_buffer(); }
