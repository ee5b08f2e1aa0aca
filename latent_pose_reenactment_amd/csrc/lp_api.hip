// error plumbing of the C ABI
#include "lp_hip.h"
#include "lp_internal.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

int lp_set_error(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int lp_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", what, hipGetErrorString(e));
        return LP_ERR_HIP;
    }
    return LP_OK;
}

extern "C" const char* lp_last_error(void) { return g_err; }
extern "C" int lp_abi_version(void) { return 12; }
