// Launch parameters shared by the weight-gradient kernels (conv_wgrad.hip, conv_wgrad_patch.hip).
#pragma once
#include "conv_common.h"

namespace primia {

struct WgradParams {
    const void* x;
    const void* dy;
    float* dw;
    int N, H, W, C, K, R, S, stride, pad, Ho, Wo;
    int klen;
    long Md;           // N*Ho*Wo
    int ntaps;         // R*S, or R for the stem
    int nkt, nct;      // channel tiles
    int nsplit;        // pixel splits
    long pix_per_split;
    long split_stride;  // 0: every split accumulates into dw; else split i writes dw + i*split_stride
                        // (per-sample gradients for DP-SGD: one split per image, stride K*klen)
    int persample;
    double* sqnorm;    // per-sample mode: if set, split i adds the squared L2 norm of ITS gradient tile to
                       // sqnorm[i] instead of writing the tile (the DP-SGD norm pass needs nothing else)
    float* ws;         // optional workspace of the store-and-reduce path (conv_wgrad_patch.hip); null: atomics
    size_t ws_bytes;
    int xpad;          // stem only: x is the padded NHWC4p input [N][H+6][W+8][4] (stem_conv.hip)
};

// sum over the wave of the squares of `n` accumulator values per lane, added (fp64 atomic) to *dst
__device__ __forceinline__ void wave_sqnorm_add(double s, double* dst) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(dst, s);
}

int wgrad_dma_dispatch(const WgradParams& p, hipStream_t st);
size_t wgrad_dma_ws_bytes(const WgradParams& p);
// halo-patch kernel (3x3, stride 1, bf16); returns PRIMIA_ERR_UNSUPPORTED when the shape is not covered
int wgrad_patch_dispatch(const WgradParams& p, hipStream_t st);
int wgrad_patch_kernel_id(const WgradParams& p);
int wgrad_patch_persample_kernel_id(const WgradParams& p);
int wgrad_tap_persample_kernel_id(const WgradParams& p);
int dp_ghost_kernel_id(int H, int W, int C, int K, int R, int S, int stride, int pad);
size_t wgrad_patch_ws_bytes(const WgradParams& p);
// up to four layers of one shape in one launch (conv_wgrad_patch.hip): preferred group size for `count` such layers
// (0: shape not served, 1: no gain), workspace of a group of n, and the launch (p[0].ws / ws_bytes = the group's)
int wgrad_patch_group_size(const WgradParams& p, int count);
size_t wgrad_patch_group_ws_bytes(const WgradParams& p, int n);
int wgrad_patch_group_dispatch(const WgradParams* p, int n, hipStream_t st);
// per-tap kernel of the stride-2 / 1x1 layers, second generation (conv_wgrad_tap.hip); needs the workspace
int wgrad_tap_dispatch(const WgradParams& p, hipStream_t st);
int wgrad_tap_kernel_id(const WgradParams& p);
size_t wgrad_tap_ws_bytes(const WgradParams& p);
// DP-SGD norm pass of those layers: whole images per block, squared tile norms added to p.sqnorm (no workspace)
int wgrad_tap_persample_dispatch(const WgradParams& p, hipStream_t st);
// conv1 + downsample of a transition block in one launch (the downsample = a tenth tap with its own dy)
int wgrad_tap_pair_dispatch(const WgradParams& p, const WgradParams& p2, hipStream_t st);
size_t wgrad_tap_pair_ws_bytes(const WgradParams& p, const WgradParams& p2);
// stem (7x7/2) halo kernel on the padded input (stem_conv.hip)
int stem_wgrad_halo_dispatch(const bf16* xp, const bf16* dy, float* dw, int N, int H, int W, hipStream_t st,
                             float* ws = nullptr, size_t ws_bytes = 0, double* sqnorm = nullptr);
size_t stem_wgrad_halo_ws_bytes(int N, int H, int W);   // 0: shape not served by the halo kernel
bool stem_wgrad_halo_blocks(int N, int H, int W, int* total, int* per_block, int* grid);
// dw tile (kt, tap, ct) = sum over nsplit partial tiles of ws [combo][split][BMK*BNC], in split order
// (wgt: DP-SGD clipped sum — split s is sample s, weighted by wgt[s])
void wgrad_tile_reduce(const float* ws, float* dw, int nsplit, int combos, int BMK, int BNC, int nkt, int nct, int C,
                       int klen, int stem, hipStream_t st, const float* wgt = nullptr);
void wgrad_tile_reduce_pair(const float* ws, float* dw, float* dw2, int nsplit, int combos1, int combos2, int BMK, int BNC,
                            int nkt, int nct, int C, int klen, int klen2, hipStream_t st);
// DP-SGD: the patch kernel's norm pass keeping every sample's tiles ([combo][image][slab]); bytes (0: not served / too
// large to be worth it), the pass, and the clipped sum over the kept tiles
size_t wgrad_patch_keep_bytes(const WgradParams& p);
int wgrad_patch_keep_dispatch(const WgradParams& p, hipStream_t st);
int wgrad_patch_clipped_sum(const WgradParams& p, const float* slabs, const float* clip, hipStream_t st);

}  // namespace primia
