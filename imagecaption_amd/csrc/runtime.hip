// Error plumbing and ABI version for libcapdec.so.
#include <stdarg.h>

#include "common.h"

namespace capdec {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return CAPDEC_ERR_LAUNCH;
    }
    return CAPDEC_OK;
}

}  // namespace capdec

extern "C" int capdec_abi_version(void) { return 2; }  // 2: capdec_epilogue.ce_*, split-K tickets in the workspace
extern "C" const char* capdec_last_error(void) { return capdec::g_err; }
