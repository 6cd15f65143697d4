// Interface of the linear-halo 3x3 kernel (conv3x3_lh2.hip) towards the dispatch code in conv_igemm.hip.
#pragma once
#include "conv_common.h"

namespace primia {

// Linear-halo 3x3 / stride-1 kernel (conv3x3_lh2.hip): persistent 392- / 196-pixel tiles.  (Its first generation,
// conv3x3_lh.hip — one tile per block, 74.8 us per launch against 55.6 — was superseded in round 3 and removed.)
// pixel tiles (= partial slots) if the shape is served by the kernel, else PRIMIA_ERR_UNSUPPORTED
int conv3x3_lh2_tiles_m(int N, int H, int W, int Cs, int Nd);
int conv3x3_lh2_dispatch(const bf16* src, const bf16* wt, bf16* dst, int N, int H, int W, int Cs, int Nd, int flip,
                         int accumulate, hipStream_t st, float* stat_partials = nullptr,
                         const uint8_t* acc_mask = nullptr);

}  // namespace primia
