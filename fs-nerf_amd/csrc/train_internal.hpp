// train_internal.hpp — entry points of the fused (MFMA) training path, called from the C-ABI in train.hip.
#pragma once
#include "common.hpp"

namespace fsn {

int64_t fused_train_workspace_floats(const fsn_mlp_desc& d, int prec, int64_t n);
int fused_train_fwd(const fsn_mlp_desc* d, int prec, const float* const* W, const float* const* b, const float* x,
                    const float* dirs, const float* pos_mask, const float* dir_mask, int64_t n, float* ws, float* out,
                    uint32_t* status, hipStream_t s);
int fused_train_bwd(const fsn_mlp_desc* d, int prec, const float* const* W, int64_t n, float* ws, const float* out,
                    const float* d_out, const float* grad_scale_dev, float* const* dW, float* const* db,
                    uint32_t* status, hipStream_t s);

}  // namespace fsn
