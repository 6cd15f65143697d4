// Interface of the linear-halo 3x3 kernel (conv3x3_lh.hip) towards the dispatch code in conv_igemm.hip.
#pragma once
#include "conv_common.h"

namespace primia {

// BatchNorm whose backward sums the data-gradient kernel emits from its write-back (primia_conv2d_dgrad_bnsums)
struct LhBnArgs {
    const void* y;          // the BatchNorm's input (bf16 [M][C])
    const uint8_t* mask;    // 1-bit ReLU mask bytes, or null (mask recomputed from y: needs beta)
    const float* gamma;
    const float* beta;      // may be null with `mask`
    const float* mean;
    const float* invstd;
    float* partials;        // [tiles_m][2][C]
};

// pixel tiles (= partial slots) if the shape is served by the kernel, else PRIMIA_ERR_UNSUPPORTED
int conv3x3_lh_tiles_m(int N, int H, int W, int Cs, int Nd);
int conv3x3_lh_dispatch(const bf16* src, const bf16* wt, bf16* dst, int N, int H, int W, int Cs, int Nd, int flip,
                        int accumulate, hipStream_t st, float* stat_partials = nullptr, const LhBnArgs* bn = nullptr,
                        const uint8_t* acc_mask = nullptr);

// second generation (conv3x3_lh2.hip): persistent 392- / 196-pixel tiles; no BatchNorm-backward sums
int conv3x3_lh2_tiles_m(int N, int H, int W, int Cs, int Nd);
int conv3x3_lh2_dispatch(const bf16* src, const bf16* wt, bf16* dst, int N, int H, int W, int Cs, int Nd, int flip,
                         int accumulate, hipStream_t st, float* stat_partials = nullptr,
                         const uint8_t* acc_mask = nullptr);

}  // namespace primia
