// Interface of the linear-halo 3x3 kernel (conv3x3_lh2.hip) towards the dispatch code in conv_igemm.hip.
#pragma once
#include "conv_common.h"

namespace primia {

// Linear-halo 3x3 / stride-1 kernel (conv3x3_lh2.hip): persistent 392- / 196-pixel tiles.  (Its first generation,
// conv3x3_lh.hip — one tile per block, 74.8 us per launch against 55.6 — was superseded in round 3 and removed.)
// pixel tiles (= partial slots) if the shape is served by the kernel, else PRIMIA_ERR_UNSUPPORTED
int conv3x3_lh2_tiles_m(int N, int H, int W, int Cs, int Nd);
int conv3x3_lh_kernel_of(int N, int H, int W, int Cs, int Nd);   // 4 conv3x3_lh2_kernel | 6 conv3x3_lh4_kernel | 0 neither
int conv3x3_lh2_dispatch(const bf16* src, const bf16* wt, bf16* dst, int N, int H, int W, int Cs, int Nd, int flip,
                         int accumulate, hipStream_t st, float* stat_partials = nullptr,
                         const uint8_t* acc_mask = nullptr);

// Transition blocks on the parity planes (conv_s2lh.hip): 3x3 / stride 2 / pad 1 and 1x1 / stride 2 forward (one launch, either
// filter may be null) and data gradient (the downsample's optional).  PRIMIA_ERR_UNSUPPORTED where conv_s2lh_ok() is false.
bool conv_s2lh_ok(int N, int H, int W, int C, int K);
int conv_s2lh_tiles_m(int N, int H, int W);
int conv_s2lh_fwd(const bf16* x, const bf16* w, bf16* y, float* stat, const bf16* w_ds, bf16* y_ds, float* stat_ds, int N,
                  int H, int W, int C, int K, hipStream_t st);
int conv_s2lh_dgrad(const bf16* dy, const bf16* wd, const bf16* dy_ds, const bf16* wd_ds, bf16* dx, int N, int H, int W, int C,
                    int K, hipStream_t st);

}  // namespace primia
