// Internal interface of dpig_thin.hip (vector-ALU kernels for the 3-output-channel image conv and the
// 3-input-channel stem convs).
// Each *_try returns 1 if it handled the call, 0 if the layer is not eligible (caller falls through to the
// MFMA path), or a negative DPIG_* status.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "dpig_hip.h"

namespace dpig {
// wide_bf16: the WIDE tensor of the layer (x / dx for the 3-output-channel conv, y / dy for the 3-input-channel convs)
// is stored as bf16 ('bf16' storage mode); the 3-channel image side, the filter and every gradient of it stay fp32.
int thin_fwd_try(const DpigConvDesc* d, int pt, int pl, const void* x, const float* w, const float* bias,
                 const float* residual, float* y, float* y_act, hipStream_t st, bool wide_bf16 = false);
int thin_dgrad_try(const DpigConvDesc* d, int pt, int pl, const float* dy, const float* w, const float* accum,
                   const float* mask, void* dx, hipStream_t st, bool wide_bf16 = false);
size_t thin_wgrad_workspace_bytes(const DpigConvDesc* d, int pt, int pl);
int thin_wgrad_try(const DpigConvDesc* d, int pt, int pl, const void* x, const float* dy, float* dw, float beta,
                   float* db, float beta_b, void* ws, size_t ws_bytes, hipStream_t st, bool wide_bf16 = false);
// three-input-channel layers (encoder stem 3x3 s1, critic conv1 5x5 s2)
int fewc_fwd_try(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo, const float* x, const float* w,
                 const float* bias, const float* residual, void* y, float* y_act, hipStream_t st, bool wide_bf16 = false);
size_t fewc_wgrad_workspace_bytes(const DpigConvDesc* d);
int fewc_wgrad_try(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo, const float* x, const void* dy,
                   float* dw, float beta, float* db, float beta_b, void* ws, size_t ws_bytes, hipStream_t st,
                   bool wide_bf16 = false);
int fewc_dgrad_try(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo, const void* dy, const float* w,
                   const float* accum, const float* mask, float* dx, hipStream_t st, bool wide_bf16 = false);
// dpig_conv_bf16.hip: split-bf16 filter gradient from the split32 images of x and dy (both operands by LDS-DMA)
int wgrad_x3_try(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo, const uint16_t* x32, const uint16_t* dy32, float* dw,
                 float beta, float* db, float beta_b, void* ws, size_t ws_bytes, hipStream_t st, double pen);
}  // namespace dpig
