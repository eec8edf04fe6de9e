// train_internal.hpp — entry points of the fused (MFMA) training path, called from the C-ABI in train.hip.
#pragma once
#include "common.hpp"

namespace fsn {

struct TrainRays {  // ray form of the forward's inputs: sample s = midpoint of [t0[s], t1[s]) on ray ri[s]
  const float *rays_o, *rays_d, *t0, *t1;
  const int64_t* ri;
};

int64_t fused_train_workspace_floats(const fsn_mlp_desc& d, int prec, int64_t n);
int fused_train_fwd(const fsn_mlp_desc* d, int prec, const float* const* W, const float* const* b, const float* x,
                    const float* dirs, const float* pos_mask, const float* dir_mask, int64_t n, float* ws, float* out,
                    uint32_t* status, hipStream_t s, const TrainRays* rays = nullptr);
int fused_train_bwd(const fsn_mlp_desc* d, int prec, const float* const* W, int64_t n, float* ws, const float* out,
                    const float* d_out, const float* grad_scale_dev, float* const* dW, float* const* db,
                    bool accumulate, float* bscale, uint32_t* bamax, uint32_t* status, hipStream_t s);

}  // namespace fsn
