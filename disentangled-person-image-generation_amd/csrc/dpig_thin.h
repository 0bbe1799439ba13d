// Internal interface of dpig_thin.hip (vector-ALU kernels for the 3-output-channel image conv and the
// 3-input-channel stem convs).
// Each *_try returns 1 if it handled the call, 0 if the layer is not eligible (caller falls through to the
// MFMA path), or a negative DPIG_* status.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "dpig_hip.h"

namespace dpig {
int thin_fwd_try(const DpigConvDesc* d, int pt, int pl, const float* x, const float* w, const float* bias,
                 const float* residual, float* y, float* y_act, hipStream_t st);
int thin_dgrad_try(const DpigConvDesc* d, int pt, int pl, const float* dy, const float* w, const float* accum,
                   const float* mask, float* dx, hipStream_t st);
size_t thin_wgrad_workspace_bytes(const DpigConvDesc* d, int pt, int pl);
int thin_wgrad_try(const DpigConvDesc* d, int pt, int pl, const float* x, const float* dy, float* dw, float beta,
                   float* db, float beta_b, void* ws, size_t ws_bytes, hipStream_t st);
// three-input-channel layers (encoder stem 3x3 s1, critic conv1 5x5 s2)
int fewc_fwd_try(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo, const float* x, const float* w,
                 const float* bias, const float* residual, float* y, float* y_act, hipStream_t st);
size_t fewc_wgrad_workspace_bytes(const DpigConvDesc* d);
int fewc_wgrad_try(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo, const float* x, const float* dy,
                   float* dw, float beta, float* db, float beta_b, void* ws, size_t ws_bytes, hipStream_t st);
int fewc_dgrad_try(const DpigConvDesc* d, int pt, int pl, int Ho, int Wo, const float* dy, const float* w,
                   const float* accum, const float* mask, float* dx, hipStream_t st);
}  // namespace dpig
