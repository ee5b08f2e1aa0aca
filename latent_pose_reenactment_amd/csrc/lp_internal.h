// internal helpers shared by the translation units of liblp_hip.so (not part of the C ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
int lp_set_error(int code, const char* msg);
int lp_check_launch(const char* what);

// conv_thin.hip: fp32 VALU weight gradient for convs with <= 4 channels on one side
bool lp_wgrad_thin_supported(int Cin, int Cout, int ksize, int upsample, int pro, int W);
int lp_wgrad_thin(const float* x, const float* dy, float* dw, float* workspace, const float* scale, const float* shift, int N, int H,
                  int W, int Cin, int Cout, int ksize, int pro, int splits, float* dbias, hipStream_t stream);
bool lp_conv_thin_fwd_supported(int Cin, int Cout, int ksize, int upsample, int pro, bool has_res, int W);
int lp_conv_thin_fwd(const float* x, const uint16_t* w_hi, const uint16_t* w_lo, float* y, const float* bias, const float* alpha, int N,
                     int H, int W, int Cin, int Cout, int CinP, int CoutP, int ksize, int f16, hipStream_t stream);
