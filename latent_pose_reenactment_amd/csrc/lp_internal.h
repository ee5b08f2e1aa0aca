// internal helpers shared by the translation units of liblp_hip.so (not part of the C ABI)
#pragma once
#include <hip/hip_runtime.h>
int lp_set_error(int code, const char* msg);
int lp_check_launch(const char* what);
