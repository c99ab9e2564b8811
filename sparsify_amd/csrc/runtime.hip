// Error reporting and ABI version for libsparsify_hip.
#include "common.hpp"

namespace sphip {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace sphip

extern "C" int sphip_abi_version(void) { return SPHIP_ABI_VERSION; }
extern "C" const char* sphip_last_error(void) { return sphip::g_err; }
