/*
 * primia_hip.h — C ABI of libprimia_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for PriMIA's hot path.  The reference is pure Python; the
 * seam an FFI would bind is (a) the top-level torch functions a PySyft worker executes for one
 * plaintext training step (syft/generic/pointers/object_pointer.py:196-217 ships them whole;
 * syft/workers/message_handler.py:59-142 executes them) and (b) the `@allow_command` per-share
 * functions of the secure path (syft/generic/utils.py:27-55).  Every entry point below names
 * the reference call it replaces.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host; the caller owns every
 *     buffer, workspaces are passed explicitly, there is no hidden allocation; the ONLY process-wide state is the
 *     table of tuning options below (primia_set_option): word-sized atomic entries, safe to read and write from any
 *     thread, but shared by every caller in the process — two engines cannot hold different options, and a change
 *     invalidates sizes asked earlier (primia_options_epoch tells);
 *   - `stream` is a hipStream_t (0 = default stream); all work is enqueued asynchronously;
 *   - return value: PRIMIA_OK (0) or a negative PRIMIA_ERR_* code; bad arguments are reported,
 *     never dereferenced;
 *   - element-wise entry points taking a count `n` accept n == 0 as a successful no-op (pointers
 *     may then be null), like the torch ops they replace do on empty tensors;
 *   - float tensors are NHWC ("channels last") in `dtype` PRIMIA_F32 or PRIMIA_BF16; weights
 *     handed over at the boundary keep the reference's OIHW / [out,in] fp32 layout;
 *   - ring tensors are int64 two's complement, arithmetic mod 2^64.
 */
#ifndef PRIMIA_HIP_H
#define PRIMIA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRIMIA_OK 0
#define PRIMIA_ERR_ARG (-1)         /* bad shape / null pointer / unsupported size */
#define PRIMIA_ERR_LAUNCH (-2)      /* HIP launch or runtime error */
#define PRIMIA_ERR_UNSUPPORTED (-3) /* valid request this build does not implement */
#define PRIMIA_ERR_WORKSPACE (-4)   /* workspace too small */

#define PRIMIA_F32 0
#define PRIMIA_BF16 1

typedef void* primia_stream_t; /* hipStream_t */

int primia_abi_version(void);

/* ---- options -------------------------------------------------------------------------------------------------------
 * The library never reads the environment.  Every switch that selects between kernels or sizes a launch (the A/B knobs
 * of the measurements under profiles/) is an entry of ONE process-wide table with the defaults the benchmarks run on:
 * primia_set_option("lh2", 0) keeps the wide 3x3 layers on the implicit GEMM, ("wgp_group", 0) launches every weight
 * gradient on its own, ("dp_ghost", 0) takes the DP-SGD norms from the weight-gradient kernels, ...  The names, defaults
 * and meanings are listed in primia_amd/csrc/options.h; unknown names return PRIMIA_ERR_ARG.  Options are read when a
 * call dispatches (and by the *_bytes / *_ok / *_kernel_id queries, which therefore must be asked again after a
 * change).  Each entry is read and written atomically (no torn values, callable from any thread), but a SEQUENCE of
 * changes is not a transaction: set options before the streams start.  primia_options_epoch() counts the changes made
 * so far (a call that leaves a value as it was does not count): a host object that sized workspaces from the queries
 * keeps the epoch it saw and re-plans — or refuses to run, as primia_amd.engine.ResNet18Engine does — when it has moved.
 * The two timing-experiment switches c64_dbg / s2lh_dbg (they SKIP parts of a kernel: wrong results) are honoured by
 * probe builds only (-DPRIMIA_PROBE=1, tools/micro); this library returns PRIMIA_ERR_UNSUPPORTED for a non-zero value.
 * The reference has no counterpart (torch picks its own kernels). */
int primia_set_option(const char* name, int value);
int64_t primia_options_epoch(void);
int primia_get_option(const char* name, int* value);
int primia_reset_options(void);
int primia_option_count(void);
int primia_option_name(int index, char* buf, int buf_len);

/* ------------------------------------------------------------------------------------------
 * Layout conversion at the boundary.  The reference hands NCHW fp32 batches to the model
 * (torchlib/utils.py:1169-1170); kernels run NHWC.  c_pad >= C pads channels with zeros
 * (the stem runs with 3 -> 4 channels).
 * ------------------------------------------------------------------------------------------ */
int primia_nchw_to_nhwc(const float* src_nchw, void* dst_nhwc, int N, int C, int H, int W,
                        int c_pad, int dtype, primia_stream_t stream);
/* The same conversion into a spatially padded destination [N][Hp][Wp][c_pad], interior at
 * (pad_top, pad_left).  The padding is NOT written: zero the buffer once after allocating it. */
int primia_nchw_to_nhwc_padded(const float* src_nchw, void* dst, int N, int C, int H, int W, int c_pad,
                               int pad_top, int pad_left, int Hp, int Wp, int dtype,
                               primia_stream_t stream);
int primia_nhwc_to_nchw(const void* src_nhwc, float* dst_nchw, int N, int C, int H, int W,
                        int c_pad, int dtype, primia_stream_t stream);
/* dst[i] = (dtype) src[i] and back; n elements. */
int primia_cast_from_f32(const float* src, void* dst, int64_t n, int dtype, primia_stream_t stream);
int primia_cast_to_f32(const void* src, float* dst, int64_t n, int dtype, primia_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Convolution — replaces torch.nn.functional.conv2d forward and its autograd backward for
 * nn.Conv2d(bias=False) (torchlib/models.py:219-235, 379-381).
 * ------------------------------------------------------------------------------------------ */
typedef struct primia_conv_desc {
    int32_t N, H, W, C;  /* input NHWC; C is the stored channel count (stem: 4)          */
    int32_t K, R, S;     /* output channels, kernel height, kernel width                  */
    int32_t stride, pad; /* same in both spatial dims (ResNet-18 only uses square convs)  */
    int32_t Ho, Wo;      /* output spatial size = (H + 2*pad - R)/stride + 1              */
} primia_conv_desc;

/* Element count of the compute copies of one conv weight (fwd layout [K][R][Sp][C'] and dgrad
 * layout [C][R][S][K]); Sp/C' include the stem's zero padding. */
int64_t primia_conv_wfwd_elems(const primia_conv_desc* d);
int64_t primia_conv_wdgrad_elems(const primia_conv_desc* d);

/* fp32 OIHW master weight [K][c_real][R][S] -> compute-dtype copies.  w_dgrad may be NULL. */
int primia_conv_weight_prepare(const primia_conv_desc* d, int c_real, const float* w_oihw,
                               void* w_fwd, void* w_dgrad, int dtype, primia_stream_t stream);

/* Batched forms of the two calls around them (one launch for all convolutions of a network, <= 24):
 * arrays of n descriptors / channel counts / pointers live on the HOST. */
int primia_conv_weight_prepare_many(const primia_conv_desc* descs_host, const int* c_real_host,
                                    const float* const* w_oihw_host, void* const* w_fwd_host,
                                    void* const* w_dgrad_host, int n, int dtype, primia_stream_t stream);
int primia_conv_wgrad_finalize_many(const primia_conv_desc* descs_host, const int* c_real_host,
                                    const float* const* dw_acc_host, float* const* dw_oihw_host, int n,
                                    primia_stream_t stream);

/* loss.backward() tail + optimizer.step() of the reference's batch loop (torchlib/utils.py:1016-1030, train.py:280-303
 * with optimizer = SGD) for the regular convolutions, as ONE pass per 32 x 64 x R*S weight tile: accumulator -> OIHW
 * gradient (still written: the caller finds every .grad), w <- w - lr * (g + weight_decay * w) on the fp32 master
 * (the expression of primia_sgd_step, bit-identical to it), master -> w_fwd / w_dgrad copies.  Replaces
 * primia_conv_wgrad_finalize_many + primia_sgd_step (on these ranges) + primia_conv_weight_prepare_many.
 * Only layers for which primia_conv_sgd_fusable() returns 1 (C % 64 == 0, c_real == C, 3x3 or 1x1) — the others, and
 * every other parameter, go through the unfused calls / primia_sgd_step_ranges.  Host arrays as above. */
int primia_conv_sgd_fusable(const primia_conv_desc* d, int c_real);
int primia_conv_sgd_step_many(const primia_conv_desc* descs_host, const int* c_real_host,
                              const float* const* dw_acc_host, float* const* dw_oihw_host,
                              float* const* w_oihw_host, void* const* w_fwd_host, void* const* w_dgrad_host,
                              int n, float lr, float weight_decay, int dtype, primia_stream_t stream);

/* A transition block's two forward convolutions of the same x — conv1 (3x3, stride 2) and the downsample (1x1, stride 2;
 * torchlib/models.py:236-247 `identity = self.downsample(x)` beside `out = self.conv1(x)`) — in ONE launch, each with its
 * per-tile BatchNorm partial sums (stat_sums may be NULL) exactly as primia_conv2d_fwd_stats writes them: same tiles,
 * same results bit for bit.  primia_conv_fwd_pair_ok() = 1 where the pair is served (bf16, K % 128 == 0). */
int primia_conv_fwd_pair_ok(const primia_conv_desc* d, const primia_conv_desc* d_ds, int dtype);
int primia_conv2d_fwd_stats_pair(const primia_conv_desc* d, const void* x, const void* w_fwd, void* y,
                                 float* stat_sums, const primia_conv_desc* d_ds, const void* w_fwd_ds, void* y_ds,
                                 float* stat_sums_ds, int dtype, primia_stream_t stream);

/* y[N,Ho,Wo,K] = conv(x[N,H,W,C], w). */
int primia_conv2d_fwd(const primia_conv_desc* d, const void* x, const void* w_fwd, void* y,
                      int dtype, primia_stream_t stream);
/* conv1 = Conv2d(3, 64, 7, 2, 3) (torchlib/models.py:371-372) on the padded channels-last input
 * [N][H + 6][W + 8][4] (primia_stem_pad_dims; interior at (3, 3), 4th channel and padding zero): the
 * filter lives in registers, the input patch is staged once per 8 x 16 output patch and the MFMA
 * operands are read straight from it.  Same result as primia_conv2d_fwd on the unpadded input.
 * bf16 only, H and W multiples of 32 (PRIMIA_ERR_UNSUPPORTED otherwise: use primia_conv2d_fwd). */
int primia_stem_pad_dims(int H, int W, int* Hp, int* Wp);
int primia_stem_conv_fwd(const void* x_padded, const void* w_fwd, void* y, int N, int H, int W, int dtype,
                         primia_stream_t stream);
/* The same, also emitting bn1's per-block partial sums [slots][2][64] (slots = primia_stem_conv_stat_slots) for
 * primia_bn_relu_maxpool_fwd_from_sums / primia_bn_fwd_train_from_sums: the statistics pass over the 411 MB
 * stem output is dropped. */
int primia_stem_conv_stat_slots(int N, int H, int W);
int primia_stem_conv_fwd_stats(const void* x_padded, const void* w_fwd, void* y, float* stat_partials, int N, int H,
                               int W, int dtype, primia_stream_t stream);
/* The stem's forward tail WITHOUT conv1's output (411 MB at batch 256; torchlib/models.py:466-471 conv1 -> bn1 -> relu ->
 * maxpool), two passes over the padded input:
 *   primia_stem_conv_stats  conv1's tiles -> bn1's partial sums [primia_stem_conv_stat_slots][2][64]; nothing stored
 *   (finalize: primia_bn_finalize_stats)
 *   primia_stem_conv_pool   conv1 recomputed -> scale / shift -> ReLU -> 3x3 / 2 max pool: pooled [N,H/4,W/4,64] + first-
 *                           maximum codes, bit-identical to primia_bn_relu_maxpool_fwd_from_sums on a stored y; y != NULL
 *                           also stores conv1's output (tests, and callers whose backward pass reads it).
 * bf16, H and W multiples of 32, W <= 256 (primia_stem_conv_pool_ok). */
int primia_stem_conv_stats(const void* x_padded, const void* w_fwd, float* stat_partials, int N, int H, int W, int dtype,
                           primia_stream_t stream);
int primia_stem_conv_pool_ok(int N, int H, int W, int dtype);
int primia_stem_conv_pool(const void* x_padded, const void* w_fwd, void* y, void* pooled, uint8_t* argmax,
                          const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, int N,
                          int H, int W, int dtype, primia_stream_t stream);
/* nn.BatchNorm2d in eval mode: the whole stem head (conv1 -> bn1 -> relu -> maxpool, torchlib/models.py:466-471) in one
 * pass over the input; equals primia_stem_conv_fwd -> primia_bn_fwd_eval -> primia_maxpool3x3s2_fwd bit for bit.  One block
 * per image: pays from about 160 images per launch (below that the three-kernel chain fills the chip better). */
int primia_stem_conv_pool_eval(const void* x_padded, const void* w_fwd, void* pooled, uint8_t* argmax, const float* gamma,
                               const float* beta, const float* running_mean, const float* running_var, float eps, int N,
                               int H, int W, int dtype, primia_stream_t stream);
/* Batch statistics from partial sums [slots][2][C] alone (what the *_from_sums entry points run first): mean / invstd of
 * M elements per channel, running statistics updated as nn.BatchNorm2d does (unbiased variance, momentum). */
int primia_bn_finalize_stats(const float* sums, int slots, int64_t M, int C, float eps, float momentum,
                             float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                             primia_stream_t stream);
/* Weight gradient of conv1 from the same padded input (accumulates into dw_acc like primia_conv2d_wgrad). */
int primia_stem_conv_wgrad(const void* x_padded, const void* dy, float* dw_acc, int N, int H, int W, int dtype,
                           primia_stream_t stream);
/* ... without atomics (see primia_conv2d_wgrad_ws): workspace from primia_stem_conv_wgrad_ws_bytes (0: this shape
 * takes the accumulate path, dw_acc must then be zeroed by the caller as for primia_stem_conv_wgrad). */
int primia_stem_conv_wgrad_persample_sqnorm(const void* x_padded, const void* dy, double* sqnorm, int N, int H, int W,
                                            int dtype, primia_stream_t stream);   /* DP-SGD norm pass, padded input */
int64_t primia_stem_conv_wgrad_ws_bytes(int N, int H, int W);
int primia_stem_conv_wgrad_ws(const void* x_padded, const void* dy, float* dw_acc, void* ws, int64_t ws_bytes, int N,
                              int H, int W, int dtype, primia_stream_t stream);
/* Same, and the per-channel sum / sum of squares of y (values as stored) — the batch statistics of the
 * BatchNorm that follows — are accumulated into stat_sums, laid out [slots][2][K] with
 * slots = primia_conv_stat_slots() partial sums (spread to keep atomics uncontended); caller zeroes it. */
/* Which kernel the library's dispatch rules select for a convolution (measurement tooling: bench.py names its
 * roofline families with it, so a run under primia_set_option("lh2", 0) / ("wgrad_kernel", 2) / ... reports the kernel that
 * actually ran).  pass 0 = forward, 1 = data gradient:
 *   1 conv_igemm_kernel   2 conv3x3_c64_kernel   4 conv3x3_lh2_kernel   (3: conv3x3_lh_kernel, removed in round 4)
 *   5 conv_s2lh_kernel (the stride-2 3x3 / 1x1 layers of the transition blocks on parity planes, round 5)
 *   6 conv3x3_lh4_kernel (the linear-halo layers that take 196-pixel tiles: loader-wave form, round 4)
 * weight gradient:
 *   13 conv_wgrad_dma_kernel   14 conv_wgrad_kernel   (11 / 12: the first two patch kernels, removed in round 4)
 *   16 conv_wgrad_patch33_kernel (3 + 3 fragments per k-step, two alternating halves)   17 conv_wgrad_tap_kernel
 *   18 conv_wgrad_patch33lw_kernel (the same scheme on four matrix + four loader waves: the default since round 5)
 *   15 the stem's (stem_conv_wgrad_kernel on the padded bf16 input, conv_wgrad_kernel<STEM> otherwise) */
int primia_conv_kernel_id(const primia_conv_desc* d, int pass, int dtype);
int primia_conv_wgrad_kernel_id(const primia_conv_desc* d, int dtype);
/* ... and the kernel that serves primia_conv2d_wgrad_persample_sqnorm (the DP-SGD norm pass, train.py:325-334) for `d`:
 *   21 dp_ghost_sqnorm7_kernel (7x7 outputs, two Gram matrices per sample)   22 dp_ghost_sqnorm7s2_kernel (stride 2)
 *   23 dp_ghost_sqnorm14_kernel   24 conv_wgrad_patch33_kernel, one block per (image, slab)
 *   25 conv_wgrad_patch33_kernel, whole images per half-block   26 conv_wgrad_tap_kernel, whole images per block
 *   13 / 14 / 15 as above */
int primia_conv_wgrad_persample_kernel_id(const primia_conv_desc* d, int dtype);

int primia_conv_stat_slots(void);
/* Slots the kernel chosen for `d` writes: kernels that own whole output rows (layer1's 64->64 convolution)
 * emit one deterministic partial per block, written rather than accumulated (no zeroing needed); the
 * generic kernel accumulates into primia_conv_stat_slots() zeroed slots.  Size stat_sums for this count
 * and pass it as `slots` to primia_bn_fwd_train_from_sums. */
int primia_conv_stat_slots_for(const primia_conv_desc* d, int dtype);
/* 1: the forward kernel of this conv WRITES its BatchNorm partial sums per tile (deterministic, nothing to zero, no
 * statistics pass needed: primia_conv2d_fwd_stats + primia_bn_fwd_train_from_sums); 0: the generic atomic slots.  The
 * slot count alone does not tell the two apart. */
int primia_conv_stats_per_tile(const primia_conv_desc* d, int dtype);
int primia_conv2d_fwd_stats(const primia_conv_desc* d, const void* x, const void* w_fwd, void* y,
                            float* stat_sums, int dtype, primia_stream_t stream);
/* dx[N,H,W,C] = conv_transpose(dy[N,Ho,Wo,K], w).  If accumulate != 0, dx += (dx is read). */
int primia_conv2d_dgrad(const primia_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                        int accumulate, int dtype, primia_stream_t stream);
/* The plain data gradient of a wide 3x3 / stride-1 layer that also forms, in its write-back, the per-channel sums the BatchNorm
 * backward of the layer IN FRONT of it needs — dx of this call is that layer's dz: sum g and sum g * xhat, g = dz * [bn(y) > 0],
 * xhat = (y - mean) * invstd — as deterministic per-tile partials [slots][2][C]; primia_bn_relu_bwd_from_sums consumes them, the
 * separate reduction pass over (y, dz) is gone (torch's batch_norm backward reads both tensors twice; here once).
 * primia_conv_dgrad_bnsums_slots: rows of the partial table, 0 where the shape is not served (bf16; the wide 3x3 / stride-1 layers on
 * conv3x3_lh2 / lh4 and the 64 -> 64 layers on conv3x3_c64). */
int primia_conv_dgrad_bnsums_slots(const primia_conv_desc* d, int dtype);
/* ... and the paired data gradient of a transition block (primia_conv2d_dgrad_pair) forming the backward sums of the RESIDUAL
 * BatchNorm in front of the block (dx = gradient w.r.t. z = relu(bn(y) + identity); relu_mask = the bytes its forward pass wrote:
 * g = dx * mask bit): primia_bn_bwd_mask_from_sums = primia_bn_bwd_mask without its reduction pass.  bf16; conv_s2lh_kernel
 * for 64-channel dx (layer2.0), conv_igemm_kernel's parity-class walk for the wider ones (layer3.0 / layer4.0): one partial per
 * (128-pixel tile, class). */
int primia_conv_dgrad_pair_bnsums_slots(const primia_conv_desc* d, int dtype);
int primia_conv2d_dgrad_pair_bnsums(const primia_conv_desc* d, const void* dy, const void* w_dgrad,
                                    const primia_conv_desc* d_ds, const void* dy_ds, const void* w_dgrad_ds, void* dx,
                                    const void* bn_y, const uint8_t* relu_mask, const float* bn_mean, const float* bn_invstd,
                                    float* sums, int dtype, primia_stream_t stream);
int primia_bn_bwd_mask_from_sums(const void* y, const uint8_t* relu_mask, const void* dz, void* dy, void* g_out,
                                 const float* gamma, const float* save_mean, const float* save_invstd, float* dgamma,
                                 float* dbeta, const float* sums, int slots, int64_t M, int C, int dtype,
                                 primia_stream_t stream);
int primia_conv2d_dgrad_bnsums(const primia_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const void* bn_y,
                               const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                               float* sums, int dtype, primia_stream_t stream);
int primia_bn_relu_bwd_from_sums(const void* y, const void* dz, void* dy, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                                 const float* sums, int slots, int64_t M, int C, int dtype, primia_stream_t stream);
/* Identity block (torchlib/models.py:268-284, `out += identity; out = relu(out)`): dx = mask(dx) + dgrad(dy), where dx
 * holds the gradient of the block's OUTPUT and relu_mask the bits of that ReLU (primia_bn_fwd_train_mask).  The
 * masked residual gradient is then never written by the BatchNorm backward pass (g_out = NULL there): one tensor
 * write less per identity block.  primia_conv_dgrad_masked_acc_ok() says whether this conv's kernel has the form
 * (else: g_out + primia_conv2d_dgrad(accumulate = 1)). */
int primia_conv_dgrad_masked_acc_ok(const primia_conv_desc* d, int dtype);
int primia_conv2d_dgrad_masked_acc(const primia_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                                   const uint8_t* relu_mask, int dtype, primia_stream_t stream);
/* ... whose write-back also forms the backward sums of the BatchNorm whose OUTPUT gradient the call completes (dx after the call
 * is that gradient): sum g and sum g * xhat over g = dx AS STORED where the layer's ReLU passed, as per-block partials
 * [slots][2][64].  mode 2 — a residual BatchNorm in front of an identity block: aux = its input y, aux_mask = the ReLU-mask bytes
 * its forward pass wrote, c0 = saved mean, c1 = saved invstd; consumer primia_bn_bwd_mask_from_sums.  mode 3 — the stem's
 * BatchNorm seen through the 3x3 / 2 max-pool (reference: torchlib/models.py conv1 -> bn1 -> relu -> maxpool): aux = the pooled
 * activation p, ReLU = [p > 0], xhat = (p - beta) / gamma, c0 = beta, c1 = gamma, aux_mask unused; consumer
 * primia_bn_relu_maxpool_bwd_from_sums (which also serves the channels whose gamma is 0 or below 2^-6 |beta| — xhat is not
 * recoverable from the STORED p there, the producer writes a zero partial for them — from y at the argmax).  64 -> 64 bf16 layers;
 * primia_conv_dgrad_masked_acc_bnsums_slots = rows of the partial table, 0 where not served. */
int primia_conv_dgrad_masked_acc_bnsums_slots(const primia_conv_desc* d, int dtype);
int primia_conv2d_dgrad_masked_acc_bnsums(const primia_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                                          const uint8_t* relu_mask, int mode, const void* aux, const uint8_t* aux_mask,
                                          const float* c0, const float* c1, float* sums, int dtype, primia_stream_t stream);
/* Transition block (torchlib/models.py:268-284 with a downsample, :232-235): dx = dgrad(conv1 3x3/2, dy) +
 * dgrad(downsample 1x1/2, dy_ds) in ONE pass — both convolutions read the same x, the downsample's gradient lands
 * on the even/even pixels, where conv1's only tap is the centre one at the same dy pixel, so it is folded into
 * that pixel class's reduction axis.  Replaces primia_conv2d_dgrad(conv1, accumulate=0) followed by
 * primia_conv2d_dgrad(downsample, accumulate=1) (one rounding instead of two).  H, W even. */
int primia_conv2d_dgrad_pair(const primia_conv_desc* d, const void* dy, const void* w_dgrad,
                             const primia_conv_desc* d_ds, const void* dy_ds, const void* w_dgrad_ds, void* dx,
                             int dtype, primia_stream_t stream);
/* dw_acc (fp32, fwd layout [K][R][Sp][C']) += sum over pixels.  Caller zeroes dw_acc first. */
int primia_conv2d_wgrad(const primia_conv_desc* d, const void* x, const void* dy, float* dw_acc,
                        int dtype, primia_stream_t stream);
/* The same weight gradient without atomics: every kernel block stores its partial tile in `ws`, a second kernel adds
 * the partials in a fixed order, so the result is deterministic and OVERWRITES dw_acc (no zeroing needed).
 * primia_conv_wgrad_ws_bytes() gives the workspace the layer needs (< 0: bad descriptor); with ws == NULL or
 * ws_bytes too small the call is primia_conv2d_wgrad (accumulate into a zeroed dw_acc).  The reference has no
 * counterpart: torch's conv backward owns its workspace (torchlib/models.py:219-235 via autograd). */
int64_t primia_conv_wgrad_ws_bytes(const primia_conv_desc* d, int dtype);
int primia_conv2d_wgrad_ws(const primia_conv_desc* d, const void* x, const void* dy, float* dw_acc,
                           void* ws, int64_t ws_bytes, int dtype, primia_stream_t stream);
/* A transition block's two weight gradients in ONE launch: conv1 (3x3 / stride 2, desc d) and the downsample (1x1 /
 * stride 2, desc d2) read the same x — the downsample's input pixel is the 3x3's centre tap — so the downsample runs as a
 * tenth tap with its own dy (torchlib/models.py:272-281 BasicBlock.conv1 / downsample[0], backwards).  dw and dw2
 * differ from their single calls only in the grouping of the ordered fp32 sums (the pixel split is chosen so that all
 * ten taps fit one round of blocks).  Workspace from primia_conv_wgrad_pair_ws_bytes (0: this pair is not
 * served — PRIMIA_ERR_UNSUPPORTED from the call — use the two single calls). */
int64_t primia_conv_wgrad_pair_ws_bytes(const primia_conv_desc* d, const primia_conv_desc* d2, int dtype);
int primia_conv2d_wgrad_pair_ws(const primia_conv_desc* d, const void* x, const void* dy, float* dw,
                                const primia_conv_desc* d2, const void* dy2, float* dw2, void* ws, int64_t ws_bytes,
                                int dtype, primia_stream_t stream);

/* Several layers of ONE shape (3x3 / stride 1, bf16) in one launch: a weight-gradient call carries ~25-29 us that do not
 * shrink with the work next to a 45-55-us main loop (profiles/r03_wgp33_phase_profile.txt), and weight gradients are
 * leaves of the backward graph — a caller may hold a layer's (x, dy) back until the siblings of its ResNet stage are
 * ready.  _group_size: preferred group size for `count` layers of this shape (0: shape not served by the grouped
 * kernel, 1: no gain); _group_ws_bytes: workspace of a group of n (0: not served); _group_ws: n layers, 2 <= n <= 4,
 * unused triples null; results per layer as primia_conv2d_wgrad_ws, deterministic.  Same reference counterpart:
 * autograd of conv3x3 (torchlib/models.py:219-235). */
int primia_conv_wgrad_group_size(const primia_conv_desc* d, int count, int dtype);
int64_t primia_conv_wgrad_group_ws_bytes(const primia_conv_desc* d, int n, int dtype);
int primia_conv2d_wgrad_group_ws(const primia_conv_desc* d, int n, const void* x0, const void* dy0, float* dw_acc0,
                                 const void* x1, const void* dy1, float* dw_acc1, const void* x2, const void* dy2,
                                 float* dw_acc2, const void* x3, const void* dy3, float* dw_acc3, void* workspace,
                                 int64_t workspace_bytes, int dtype, primia_stream_t stream);

/* Per-sample weight gradients for DP-SGD: dw_ps [N][K][klen] (fwd layout, fp32, zeroed by the caller),
 * image n's gradient in slab n. */
int primia_conv2d_wgrad_persample(const primia_conv_desc* d, const void* x, const void* dy,
                                  float* dw_ps, int dtype, primia_stream_t stream);
/* The DP-SGD norm pass without the per-sample gradients themselves: sqnorm[n] += ||dW_n||_F^2 (fp64,
 * accumulated across calls, i.e. across layers).  Every kernel block of the per-sample form holds one
 * sample's complete gradient tile in registers, so the squares are summed there and the N x |W| slab
 * (11 GB per step for ResNet-18 at batch 256) is never written, zeroed or re-read. */
int primia_conv2d_wgrad_persample_sqnorm(const primia_conv_desc* d, const void* x, const void* dy,
                                         double* sqnorm, int dtype, primia_stream_t stream);

/* DP-SGD where a layer's per-sample gradients are small (the stem: 64 KiB per sample; layer1's 64 -> 64 convs: 144 KiB):
 * the norm pass KEEPS every sample's complete tile next to adding its squares, and the clipped sum  sum_n clip_n g_n  is
 * one ordered reduce over the kept tiles weighted by the clip factors — instead of scaling the rows of dy (a read + write
 * of the whole dy) and a second weight-gradient pass.  _slab_bytes: size of the keep buffer (0: layer not served, or its
 * tiles exceed the budget — use primia_conv2d_wgrad_persample_sqnorm + primia_scale_rows + primia_conv2d_wgrad_ws).
 * Replaces pytorch-dp's per-sample gradient + clip + sum for these layers (train.py:325-334). */
int64_t primia_conv_wgrad_persample_slab_bytes(const primia_conv_desc* d, int dtype);
int primia_conv2d_wgrad_persample_sqnorm_keep(const primia_conv_desc* d, const void* x, const void* dy, double* sqnorm,
                                              void* slabs, int64_t slab_bytes, int dtype, primia_stream_t stream);
int primia_conv_wgrad_clipped_sum(const primia_conv_desc* d, const void* slabs, const float* clip, float* dw_acc, int dtype,
                                  primia_stream_t stream);
int64_t primia_stem_conv_wgrad_persample_slab_bytes(int N, int H, int W);
int primia_stem_conv_wgrad_persample_sqnorm_keep(const void* x_padded, const void* dy, double* sqnorm, void* slabs,
                                                 int64_t slab_bytes, int N, int H, int W, int dtype,
                                                 primia_stream_t stream);
int primia_stem_conv_wgrad_clipped_sum(const void* slabs, const float* clip, float* dw_acc, int N, primia_stream_t stream);
/* dw_oihw[K][c_real][R][S] = transpose(dw_acc) (drops padding). */
int primia_conv_wgrad_finalize(const primia_conv_desc* d, int c_real, const float* dw_acc,
                               float* dw_oihw, primia_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Data path in front of the step: MixUp pair mixing (torchlib/utils.py:337-400), one-hot
 * targets (utils.py:449-466) and per-channel dataset statistics (torchlib/dataloader.py:220-247).
 * ------------------------------------------------------------------------------------------ */
/* The deterministic core of create_albu_transform (torchlib/dataloader.py:138-217) for ONE decoded image:
 * a.Resize(R, R) -> a.RandomCrop(S, S) at (oy, ox) [-> a.VerticalFlip] -> a.ToFloat(255) ->
 * a.Normalize(mean, std, max_pixel_value=1.0).  src: uint8 [Hin][Win][C] (C = 1 or 3, device memory);
 * out: fp32 [C][S][S].  Bilinear, half-pixel centres, clamped (cv2.INTER_LINEAR), rounded to a uint8 level before
 * ToFloat.  mean == std == NULL stops after ToFloat (the statistics pass of torchlib/utils.py:645-666). */
int primia_image_prepare(const uint8_t* src, int Hin, int Win, int C, int R, int oy, int ox, int S, int flip_v,
                         const float* mean, const float* std, float* out, primia_stream_t stream);

/* The augmentation chain of create_albu_transform (torchlib/dataloader.py:138-217) as separate steps on uint8 HWC
 * images in device memory (C = 1 or 3); the host draws the random parameters (primia_amd/augment.py), every call is a
 * deterministic function of image and parameters.  In the order the reference applies them:
 *   primia_image_affine_u8       transforms.RandomAffine: PIL AFFINE / NEAREST with the INVERSE matrix (a b c; d e f),
 *                                source index floor(a (x+.5) + b (y+.5) + c), fill 0.  out != src.
 *   primia_image_resize_crop_u8  a.Resize(R, R) -> a.RandomCrop(S, S) at (oy, ox) [-> a.VerticalFlip], uint8 out
 *                                (the sampling of primia_image_prepare).
 *   primia_clahe_u8              a.CLAHE(clip_limit, tile grid 8 x 8): OpenCV's algorithm on the plane (C = 1) or on L of
 *                                CIE L*a*b* (C = 3); workspace = primia_clahe_workspace_bytes.  in-place allowed.
 *   primia_image_lut_u8          cv2.LUT with the 256-entry table albumentations builds (RandomGamma, RandomBrightness).
 *   primia_image_box_blur_u8     a.Blur -> cv2.blur(k, k), BORDER_REFLECT_101, anchor k / 2 (any k <= 63).  out != in.
 *   primia_image_remap_u8        cv2.remap(INTER_LINEAR, BORDER_REFLECT_101) with float32 coordinate maps [H][W]: the
 *                                sampling pass of a.ElasticTransform / a.OpticalDistortion / a.GridDistortion
 *                                (torchlib/dataloader.py:167-172; albumentations 0.4.6 functional.py).  out != src.
 *   primia_warp_map_affine       maps of cv2.warpAffine from the INVERSE 2 x 3 matrix (ElasticTransform's first stage).
 *   primia_warp_map_optical      cv2.initUndistortRectifyMap(camera (fx, fy, cx, cy), distortion (k, k, 0, 0, 0), new
 *                                camera (fx, fy, new_cx, new_cy)) — F.optical_distortion.
 *   primia_warp_map_grid         meshgrid(xx[W], yy[H]) — F.grid_distortion's piecewise-linear axes (built by the host).
 *   primia_warp_map_elastic      (x + dx, y + dy) with d = gaussian_filter(2 field - 1, sigma) * alpha: F.elastic_transform's
 *                                second stage from the two uniform [0, 1) fields of numpy's RandomState (float64 [H][W]);
 *                                workspace >= 16 H W bytes.
 *   primia_image_fog_u8          F.add_fog's haze discs: per (x, y) of haze_xy a white disc of radius hw / 2 centred at
 *                                (x + hw / 2, y + hw / 2) blended with cv2.addWeighted(alpha); follow with
 *                                primia_image_box_blur_u8(hw / 10).  a.RandomFog, torchlib/dataloader.py:188-189.
 *   primia_image_add_noise_u8    a.GaussNoise: uint8(clip(img + noise, 0, 255)), noise fp32 per element.
 *   primia_image_finish          a.ToFloat(255) -> a.Normalize(mean, std, 1.0): uint8 HWC -> fp32 [C][S][S]. */
int primia_image_affine_u8(const uint8_t* src, int H, int W, int C, float a, float b, float c, float d, float e, float f,
                           uint8_t* out, primia_stream_t stream);
int primia_image_resize_crop_u8(const uint8_t* src, int Hin, int Win, int C, int R, int oy, int ox, int S, int flip_v,
                                uint8_t* out, primia_stream_t stream);
int64_t primia_clahe_workspace_bytes(int H, int W, int C);
int primia_clahe_u8(const uint8_t* img, int H, int W, int C, float clip_limit, void* workspace, int64_t workspace_bytes,
                    uint8_t* out, primia_stream_t stream);
int primia_image_lut_u8(const uint8_t* in, int64_t n, const uint8_t* table256, uint8_t* out, primia_stream_t stream);
int primia_image_box_blur_u8(const uint8_t* in, int H, int W, int C, int k, uint8_t* out, primia_stream_t stream);
int primia_image_remap_u8(const uint8_t* src, int H, int W, int C, const float* map_x, const float* map_y, uint8_t* out,
                          primia_stream_t stream);
int primia_warp_map_affine(int H, int W, double a, double b, double c, double d, double e, double f, float* map_x,
                           float* map_y, primia_stream_t stream);
int primia_warp_map_optical(int H, int W, double k, double fx, double fy, double cx, double cy, double new_cx,
                            double new_cy, float* map_x, float* map_y, primia_stream_t stream);
int primia_warp_map_grid(int H, int W, const float* xx, const float* yy, float* map_x, float* map_y,
                         primia_stream_t stream);
int primia_warp_map_elastic(int H, int W, const double* field_x, const double* field_y, double sigma, double alpha,
                            void* workspace, int64_t workspace_bytes, float* map_x, float* map_y,
                            primia_stream_t stream);
int primia_image_fog_u8(const uint8_t* in, int H, int W, int C, const int32_t* haze_xy, int n_haze, int hw, float alpha,
                        uint8_t* out, primia_stream_t stream);
/* The rest of create_albu_transform's members (torchlib/dataloader.py:173-201; `yes` in configs/torch/
 * pneumonia-resnet-pretrained-fast.ini).  InvertImg and Solarize are tables for primia_image_lut_u8.
 *   primia_image_equalize_u8     a.Equalize(mode "cv", by_channels): cv2.equalizeHist per channel; workspace >= 3840 bytes.
 *   primia_image_fill_rects_u8   F.cutout: rects[n][4] = (x1, y1, x2, y2) filled in place — a.Cutout's five holes,
 *                                a.GridDropout's grid (both lists are built by the host from the reference's draws).
 *   primia_image_swap_tiles_u8   F.swap_tiles_on_image (a.RandomGridShuffle): tiles[n][6] = (y, x, old_y, old_x, h, w).
 *   primia_image_hsv_shift_u8    F._shift_hsv_uint8 (a.HueSaturationValue): cv2 RGB2HSV -> luts[3][256] (hue, saturation,
 *                                value tables) -> cv2 HSV2RGB, 3 channels.
 *   primia_image_shadow_u8       F.add_shadow (a.RandomShadow): lightness halved in HLS under the union of the polygons
 *                                vertices[n_polygons][n_vertices][2] = (x, y), 3 channels.
 *   primia_image_sun_flare_u8    F.add_sun_flare (a.RandomSunFlare): steps[n][6] = (x, y, radius, r, g, b) filled circles
 *                                drawn onto the overlay, blended with alpha[k] / beta[k] after each; the overlay restarts
 *                                from the output at step n_first (the main flare's 40 growing circles), 3 channels. */
int primia_image_equalize_u8(const uint8_t* in, int H, int W, int C, void* workspace, int64_t workspace_bytes, uint8_t* out,
                             primia_stream_t stream);
int primia_image_fill_rects_u8(uint8_t* img, int H, int W, int C, const int32_t* rects, int n, int fill,
                               primia_stream_t stream);
int primia_image_swap_tiles_u8(const uint8_t* src, int H, int W, int C, const int32_t* tiles, int n, uint8_t* dst,
                               primia_stream_t stream);
int primia_image_hsv_shift_u8(const uint8_t* in, int H, int W, const uint8_t* luts3x256, uint8_t* out,
                              primia_stream_t stream);
int primia_image_shadow_u8(const uint8_t* in, int H, int W, const int32_t* vertices, int n_polygons, int n_vertices,
                           uint8_t* out, primia_stream_t stream);
int primia_image_sun_flare_u8(const uint8_t* in, int H, int W, const int32_t* steps, const float* alpha, const float* beta,
                              int n_steps, int n_first, uint8_t* out, primia_stream_t stream);
int primia_image_add_noise_u8(const uint8_t* in, const float* noise, int64_t n, uint8_t* out, primia_stream_t stream);
int primia_image_finish(const uint8_t* in, int S, int C, const float* mean, const float* std, float* out,
                        primia_stream_t stream);

/* out[i] = lam * x[i] + one_minus_lam * x[L/2 + i] for i < L/2 over rows of `per_sample` floats (three
 * roundings, like the reference's expression); an odd trailing sample is copied to out[L/2].  The same call
 * mixes the one-hot targets (per_sample = classes).  out has ceil(L/2) rows. */
int primia_mixup(const float* x, float* out_x, int64_t L, int64_t per_sample, float lam, float one_minus_lam,
                 primia_stream_t stream);
int primia_to_one_hot(const int64_t* labels, float* out, int64_t n, int classes, primia_stream_t stream);
/* mean[c], std[c] (unbiased) of an NCHW fp32 tensor over (N, H, W) = torch.std_mean(dim=(0,2,3)). */
int64_t primia_channel_stats_workspace_bytes(int C);
int primia_channel_mean_std(const float* x_nchw, int64_t N, int C, int64_t HW, float* mean, float* stdv,
                            void* workspace, int64_t workspace_bytes, primia_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm2d (+ fused ReLU / residual add) — replaces F.batch_norm, F.relu and Tensor.__iadd__
 * (torchlib/models.py:261-264, 268-284, 382-383); torch defaults momentum 0.1, eps 1e-5.
 * Tensors are [M, C] with M = N*H*W.
 * ------------------------------------------------------------------------------------------ */
int64_t primia_bn_workspace_bytes(int64_t M, int C);
/* Training forward: batch statistics, running-stat EMA (unbiased variance), then
 * z = act(gamma * (y - mean) * invstd + beta [+ residual]).  save_mean / save_invstd [C] are
 * kept for backward. */
int primia_bn_fwd_train(const void* y, const void* residual, void* z, const float* gamma,
                        const float* beta, float* running_mean, float* running_var,
                        float* save_mean, float* save_invstd, int64_t M, int C, float eps,
                        float momentum, int relu, void* workspace, int64_t workspace_bytes,
                        int dtype, primia_stream_t stream);
/* As primia_bn_fwd_train, with the batch sums already known (primia_conv2d_fwd_stats):
 * sums = [slots][2][C] partial sums, combined in fp64. */
int primia_bn_fwd_train_from_sums(const void* y, const void* residual, void* z, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var,
                                  float* save_mean, float* save_invstd, const float* sums, int slots,
                                  int64_t M, int C, float eps, float momentum, int relu, int dtype,
                                  primia_stream_t stream);
/* Eval forward with running statistics. */
int primia_bn_fwd_eval(const void* y, const void* residual, void* z, const float* gamma,
                       const float* beta, const float* running_mean, const float* running_var,
                       int64_t M, int C, float eps, int relu, int dtype, primia_stream_t stream);
/* Backward.  dz is the gradient w.r.t. z; if relu != 0 it is masked with (z > 0) first.
 * Outputs: dy (gradient w.r.t. the conv output y), dgamma, dbeta; if g_out != NULL the masked
 * gradient is also written there (the residual branch's gradient; may alias dz). */
int primia_bn_bwd(const void* y, const void* z, const void* dz, void* dy, void* g_out,
                  const float* gamma, const float* save_mean, const float* save_invstd,
                  float* dgamma, float* dbeta, int64_t M, int C, int relu, void* workspace,
                  int64_t workspace_bytes, int dtype, primia_stream_t stream);
/* Residual layers (bn2 of every BasicBlock, models.py:277-282): the forward pass also writes one mask byte per
 * 16-byte chunk (bit i = stored z_i > 0; M*C*sizeof(elem)/16 bytes) and both backward passes read it instead
 * of z — 1/16 of the bytes.  `sums` (with `slots`) as in primia_bn_fwd_train_from_sums, or NULL to compute the
 * statistics here (then `workspace` is required).  Same results as primia_bn_fwd_train(relu = 1) /
 * primia_bn_bwd(relu = 1). */
int primia_bn_fwd_train_mask(const void* y, const void* residual, void* z, uint8_t* relu_mask, const float* gamma,
                             const float* beta, float* running_mean, float* running_var, float* save_mean,
                             float* save_invstd, const float* sums, int slots, int64_t M, int C, float eps,
                             float momentum, void* workspace, int64_t workspace_bytes, int dtype,
                             primia_stream_t stream);
/* primia_bn_fwd_train_from_sums (relu_mask = NULL) / primia_bn_fwd_train_mask (relu_mask given, relu implied) with the
 * one-block finalize launch folded into the apply kernel (round 5): its first C / 4 blocks combine the partial sums — same
 * slicing and order as the finalize kernel, bit-identical mean / invstd / running statistics — and publish them behind one flag
 * word each; every block waits for the flags of its channels.  `flags`: C / 4 uint32 words that are ZERO when the call starts
 * (the caller zeroes them; primia_amd.engine keeps every layer's words in one buffer and clears it once per step). */
int primia_bn_fwd_train_apply_inline(const void* y, const void* residual, void* z, uint8_t* relu_mask, const float* gamma,
                                     const float* beta, float* running_mean, float* running_var, float* save_mean,
                                     float* save_invstd, const float* sums, int slots, int64_t M, int C, float eps,
                                     float momentum, int relu, uint32_t* flags, int dtype, primia_stream_t stream);
int primia_bn_bwd_mask(const void* y, const uint8_t* relu_mask, const void* dz, void* dy, void* g_out,
                       const float* gamma, const float* save_mean, const float* save_invstd, float* dgamma,
                       float* dbeta, int64_t M, int C, void* workspace, int64_t workspace_bytes, int dtype,
                       primia_stream_t stream);
/* Backward of z = relu(bn(y)) WITHOUT a residual (bn1 of every BasicBlock and the stem,
 * torchlib/models.py:261-263, 469-470): the mask (z > 0) is recomputed from y with the forward pass's
 * own fma, so z is not read — one tensor less in each of the two passes.  Same outputs as
 * primia_bn_bwd(relu = 1, g_out = NULL). */
int primia_bn_relu_bwd(const void* y, const void* dz, void* dy, const float* gamma, const float* beta,
                       const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                       int64_t M, int C, void* workspace, int64_t workspace_bytes, int dtype,
                       primia_stream_t stream);
/* Transition block forward (torchlib/models.py:277-283): z = relu(bn2(y2) + bn_d(yd)) and its 1-bit mask, batch
 * statistics / running estimates of BOTH BatchNorms updated; the downsample BatchNorm's output is applied on the fly
 * (rounded to the storage type as if stored) and never written.  Bit-identical to primia_bn_fwd_train(yd -> idn,
 * relu = 0) followed by primia_bn_fwd_train_mask(y2, residual = idn).  sums2 / slots2, sums_d / slots_d: the partial
 * sums the convolutions emitted (primia_conv2d_fwd_stats) or NULL (a statistics pass over that tensor is run). */
int primia_bn_fwd_train_pair(const void* y2, const void* yd, void* z, uint8_t* relu_mask, const float* gamma2,
                             const float* beta2, float* running_mean2, float* running_var2, float* save_mean2,
                             float* save_invstd2, const float* sums2, int slots2, const float* gamma_d,
                             const float* beta_d, float* running_mean_d, float* running_var_d, float* save_mean_d,
                             float* save_invstd_d, const float* sums_d, int slots_d, int64_t M, int C, float eps,
                             float momentum, void* workspace, int64_t workspace_bytes, int dtype,
                             primia_stream_t stream);
/* Transition block (torchlib/models.py:277-283 with a downsample): out = relu(bn2(y2) + bn_d(yd)).  Both BatchNorm
 * backward passes in ONE reduction + ONE apply pass over (y2, yd, dz): dy2, dyd and the four parameter gradients; the
 * masked gradient dz * relu_mask is never written (the block's input gradient is primia_conv2d_dgrad_pair(dy1, dyd)).
 * Bit-identical to primia_bn_bwd_mask(y2, .., g_out = g) followed by primia_bn_bwd(yd, dz = g, relu = 0).
 * workspace >= 1024 * 3 * C floats. */
int primia_bn_bwd_pair(const void* y2, const void* yd, const void* dz, const uint8_t* relu_mask, void* dy2, void* dyd,
                       const float* gamma2, const float* save_mean2, const float* save_invstd2, const float* gamma_d,
                       const float* save_mean_d, const float* save_invstd_d, float* dgamma2, float* dbeta2,
                       float* dgamma_d, float* dbeta_d, int64_t M, int C, void* workspace, int64_t workspace_bytes,
                       int dtype, primia_stream_t stream);
/* The stem's tail fused: pooled, argmax = MaxPool2d(3, 2, 1)(relu(bn1(y))) straight from the conv
 * output (torchlib/models.py:468-471), batch statistics included — z = relu(bn(y)), the largest
 * activation of the network, is never written.  Same results as primia_bn_fwd_train(relu = 1)
 * followed by primia_maxpool3x3s2_fwd (values are rounded to `dtype` before the window comparison). */
int primia_bn_relu_maxpool_fwd(const void* y, void* pooled, uint8_t* argmax, const float* gamma,
                               const float* beta, float* running_mean, float* running_var,
                               float* save_mean, float* save_invstd, int N, int H, int W, int C, float eps,
                               float momentum, void* workspace, int64_t workspace_bytes, int dtype,
                               primia_stream_t stream);
/* The same with the batch sums already available as [slots][2][C] partials (from the conv kernel). */
int primia_bn_relu_maxpool_fwd_from_sums(const void* y, void* pooled, uint8_t* argmax, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var,
                                         float* save_mean, float* save_invstd, const float* sums, int slots, int N,
                                         int H, int W, int C, float eps, float momentum, int dtype,
                                         primia_stream_t stream);
/* Its backward: from the pooled output, dpooled and the argmax codes to dy / dgamma / dbeta.  dgamma / dbeta
 * are summed at pooled resolution (a window's gradient reaches exactly its argmax, whose activation is the
 * pooled value), the per-element pool gradient is gathered only in the apply pass.  Equal to
 * primia_maxpool3x3s2_bwd followed by primia_bn_bwd(relu = 1) up to rounding (that chain rounds the pool
 * gradient to `dtype` in between and normalises with y rather than with the stored activation). */
int primia_bn_relu_maxpool_bwd(const void* y, const void* pooled, const void* dpooled, const uint8_t* argmax, void* dy,
                               const float* gamma, const float* beta, const float* save_mean,
                               const float* save_invstd, float* dgamma, float* dbeta, int N, int H, int W,
                               int C, void* workspace, int64_t workspace_bytes, int dtype,
                               primia_stream_t stream);
/* primia_bn_relu_maxpool_bwd(dy = NULL) — dgamma / dbeta only — when the reduction over (pooled, dpooled) already happened in the
 * write-back of the data gradient that produced dpooled (primia_conv2d_dgrad_masked_acc_bnsums, mode 3): `sums` = its partials
 * [slots][2][C].  A channel with |gamma| < 2^-6 * |beta| or gamma == 0 (the rounding error of the stored pooled value divided by
 * gamma would swamp xhat; pretrained bn1 layers have such channels) is served from y at the argmax positions, as in the
 * unfused call. */
int primia_bn_relu_maxpool_bwd_from_sums(const void* y, const void* pooled, const void* dpooled, const uint8_t* argmax,
                                         const float* gamma, const float* beta, const float* save_mean,
                                         const float* save_invstd, float* dgamma, float* dbeta, const float* sums, int slots,
                                         int N, int H, int W, int C, int dtype, primia_stream_t stream);
/* The stem's backward tail without the 411 MB dy (batch 256, 224 x 224): primia_bn_relu_maxpool_bwd with dy = NULL
 * forms dgamma / dbeta only, and this call is conv1's weight gradient (primia_stem_conv_wgrad_ws) whose dy tiles are
 * produced on the fly, per 8 x 16 output patch, from y, dpooled, the argmax codes and bn1's statistics — the apply
 * pass of primia_bn_relu_maxpool_bwd moved into the operand staging of the weight-gradient kernel (the reference's
 * loss.backward() through conv1 <- bn1 <- relu <- maxpool, torchlib/models.py:466-471).  dw_acc is bit-identical to
 * the two-call chain.  bf16, H and W multiples of 32; PRIMIA_ERR_UNSUPPORTED otherwise (use the chain). */
int primia_stem_bwd_fused_ok(int N, int H, int W, int dtype);
int primia_stem_bwd_fused(const void* x_padded, const void* y, const void* dpooled, const uint8_t* argmax,
                          const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                          const float* dgamma, const float* dbeta, float* dw_acc, void* ws, int64_t ws_bytes, int N,
                          int H, int W, int dtype, primia_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm (+ fused ReLU / residual) and the per-sample pieces of DP-SGD — BASELINE.json
 * configs[3].  The reference attaches pytorch-dp's PrivacyEngine (alphas [1,10,100], noise_multiplier
 * 1.3, max_grad_norm 1.0; train.py:325-334), which rejects BatchNorm (train.py:308); ResNet's
 * `norm_layer` hook (torchlib/models.py:355,362-364) takes GroupNorm instead.  x is [N, HW, C],
 * G groups of C/G consecutive channels, statistics per (sample, group), eps inside the sqrt.
 * ------------------------------------------------------------------------------------------ */
int64_t primia_gn_workspace_bytes(int N, int C, int G);
int primia_gn_fwd(const void* y, const void* residual, void* z, const float* gamma, const float* beta,
                  float* save_mean, float* save_invstd, int N, int HW, int C, int G, float eps, int relu,
                  void* workspace, int64_t workspace_bytes, int dtype, primia_stream_t stream);
/* Backward: dy, optional masked gradient g_out, and PER-SAMPLE affine gradients ps_dgamma / ps_dbeta
 * [N][C] (summed over samples they are the usual dgamma / dbeta). */
int primia_gn_bwd(const void* y, const void* z, const void* dz, void* dy, void* g_out, const float* gamma,
                  const float* save_mean, const float* save_invstd, float* ps_dgamma, float* ps_dbeta,
                  int N, int HW, int C, int G, int relu, void* workspace, int64_t workspace_bytes,
                  int dtype, primia_stream_t stream);
/* Residual layers, z = relu(gn(y) + residual) (BasicBlock.forward, torchlib/models.py:238-247 with the GroupNorm
 * norm_layer of train.py:308): the forward pass also writes one mask byte per 16-byte chunk (bit i = stored z_i > 0)
 * and the backward passes read that byte instead of z — 1/16 of the bytes; with g_out = NULL the masked gradient is not
 * written either (primia_conv2d_dgrad_masked_acc applies the mask to the residual gradient itself).  Same results as
 * primia_gn_fwd(relu = 1) / primia_gn_bwd(relu = 1), bit for bit. */
int primia_gn_fwd_mask(const void* y, const void* residual, void* z, uint8_t* relu_mask, const float* gamma,
                       const float* beta, float* save_mean, float* save_invstd, int N, int HW, int C, int G,
                       float eps, void* workspace, int64_t workspace_bytes, int dtype, primia_stream_t stream);
int primia_gn_bwd_mask(const void* y, const uint8_t* relu_mask, const void* dz, void* dy, void* g_out,
                       const float* gamma, const float* save_mean, const float* save_invstd, float* ps_dgamma,
                       float* ps_dbeta, int N, int HW, int C, int G, void* workspace, int64_t workspace_bytes,
                       int dtype, primia_stream_t stream);
/* The same for z = relu(gn(y)) WITHOUT a residual: the ReLU mask is recomputed from y (the value the forward pass
 * rounded, one shared expression), so z is not read — two tensor reads instead of three in both passes.  Replaces the
 * autograd of F.relu(GroupNorm(y)) (torchlib/models.py:362-364 norm_layer hook, :238-240 forward). */
int primia_gn_relu_bwd(const void* y, const void* dz, void* dy, const float* gamma, const float* beta,
                       const float* save_mean, const float* save_invstd, float* ps_dgamma, float* ps_dbeta, int N, int HW,
                       int C, int G, void* workspace, int64_t workspace_bytes, int dtype, primia_stream_t stream);

/* Stem tail under GroupNorm, gn1 -> relu -> maxpool(3, 2, 1), as ONE op each way (the full-resolution z and dz
 * tensors are never written): statistics + pooled output and argmax codes (0..8, first maximum); backward = per-sample
 * reductions at pooled resolution + dy and the per-sample affine gradients.  _bwd needs even H and W
 * (PRIMIA_ERR_UNSUPPORTED otherwise: use primia_maxpool3x3s2_bwd + primia_gn_relu_bwd).  Replaces
 * self.maxpool(self.relu(self.bn1(x))) with norm_layer = GroupNorm (torchlib/models.py:362-364, :414-417) and its
 * autograd. */
int primia_gn_relu_maxpool_fwd(const void* y, void* pooled, uint8_t* argmax, const float* gamma, const float* beta,
                               float* save_mean, float* save_invstd, int N, int H, int W, int C, int G, float eps,
                               void* workspace, int64_t workspace_bytes, int dtype, primia_stream_t stream);
int primia_gn_relu_maxpool_bwd(const void* y, const void* pooled, const void* dpooled, const uint8_t* argmax, void* dy,
                               const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                               float* ps_dgamma, float* ps_dbeta, int N, int H, int W, int C, int G, void* workspace,
                               int64_t workspace_bytes, int dtype, primia_stream_t stream);
/* sq_acc[n] += sum_j x[n][j]^2 (fp64): per-sample squared gradient norm, accumulated layer by layer. */
int primia_persample_sqnorm(const float* x, int N, int64_t per_sample, double* sq_acc,
                            primia_stream_t stream);
/* ... for `count` small tensors x_k [N][width_k] in one launch (DEVICE arrays of pointers / widths) */
int primia_persample_sqnorm_many(const void* xs_dev, const int* widths_dev, int count, int N, double* sq_acc,
                                 primia_stream_t stream);
/* clip[n] = min(1, max_grad_norm / (sqrt(sq[n]) + 1e-6)). */
int primia_dp_clip_factors(const double* sq, float* clip, int N, float max_grad_norm,
                           primia_stream_t stream);
/* x[n][:] *= s[n] for N consecutive blocks of elems_per_sample elements (dtype tensor, in place). */
int primia_scale_rows(void* x, const float* s, int N, int64_t elems_per_sample, int dtype,
                      primia_stream_t stream);
/* The same for up to 16 tensors in one launch (host arrays of pointers / per-sample element counts). */
int primia_scale_rows_many(void* const* xs_host, const int64_t* elems_per_sample_host, int count, const float* s,
                           int N, int dtype, primia_stream_t stream);
/* out[c] = sum_n w[n] * x[n][c]. */
int primia_weighted_colsum(const float* x, const float* w, float* out, int N, int C,
                           primia_stream_t stream);
/* `count` such sums in one launch: xs_dev / outs_dev are DEVICE arrays of `count` pointers, widths_dev of `count` ints */
int primia_weighted_colsum_many(const void* xs_dev, const float* w, const void* outs_dev, const int* widths_dev, int count,
                                int max_width, int N, primia_stream_t stream);
/* ps[n] = [dy[n]^T x[n] flattened (out_f*in_f) | dy[n] (out_f)]: per-sample gradients of nn.Linear. */
int primia_fc_persample_grads(const float* x, const float* dy, float* ps, int N, int in_f, int out_f,
                              primia_stream_t stream);
/* g = (g + noise * sigma) * inv_batch — the Gaussian mechanism on the summed clipped gradient. */
int primia_dp_add_noise(float* g, const float* noise, int64_t n, float sigma, float inv_batch,
                        primia_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Pooling — replaces F.max_pool2d(3,2,1) / F.avg_pool2d(3,2,1) (torchlib/models.py:384-389)
 * and nn.AvgPool2d(7) + flatten (models.py:400-404, 477-478).
 * ------------------------------------------------------------------------------------------ */
int primia_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* argmax, int N, int H, int W, int C,
                            int dtype, primia_stream_t stream);
int primia_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int N, int H, int W,
                            int C, int dtype, primia_stream_t stream);
int primia_avgpool3x3s2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype,
                            primia_stream_t stream);
int primia_avgpool3x3s2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype,
                            primia_stream_t stream);
/* feat[N,C] (fp32) = mean over HW of x[N,HW,C]; backward spreads dfeat/HW. */
int primia_global_avgpool_fwd(const void* x, float* feat, int N, int HW, int C, int dtype,
                              primia_stream_t stream);
int primia_global_avgpool_bwd(const float* dfeat, void* dx, int N, int HW, int C, int dtype,
                              primia_stream_t stream);

/* The head in two launches (torchlib/models.py:400-404, 478-481, 495: AvgPool2d(7) -> flatten -> Linear):
 * primia_head_fwd = primia_global_avgpool_fwd + primia_linear_fwd (feat [N][C] is kept for the weight gradient),
 * primia_head_bwd = dx of primia_linear_bwd + primia_global_avgpool_bwd (C a multiple of the 16-byte chunk). */
int primia_head_fwd(const void* x, const float* w, const float* b, float* feat, float* logits, int N, int HW, int C,
                    int out_f, int dtype, primia_stream_t stream);
int primia_head_bwd(const float* w, const float* dlogits, void* dx, int N, int HW, int C, int out_f, int dtype,
                    primia_stream_t stream);
/* primia_head_bwd that also forms the backward sums of the residual BatchNorm whose output gradient it writes (the last block's
 * bn2; dx = the gradient w.r.t. z = relu(bn(y) + identity), relu_mask = the bytes that layer's forward pass wrote): partials
 * [N][2][C] of sum g and sum g * xhat for primia_bn_bwd_mask_from_sums(slots = N).  PRIMIA_ERR_UNSUPPORTED unless 256 is a
 * multiple of C / (channels per 16-byte chunk). */
int primia_head_bwd_bnsums(const float* w, const float* dlogits, void* dx, const void* bn_y, const uint8_t* relu_mask,
                           const float* bn_mean, const float* bn_invstd, float* sums, int N, int HW, int C, int out_f, int dtype,
                           primia_stream_t stream);
/* ------------------------------------------------------------------------------------------
 * Classifier head and loss — replaces F.linear (models.py:479,495), nn.CrossEntropyLoss
 * (train.py:335-340) and Cross_entropy_one_hot (torchlib/utils.py:404-441).  All fp32.
 * ------------------------------------------------------------------------------------------ */
int primia_linear_fwd(const float* x, const float* w, const float* b, float* y, int N, int in_f,
                      int out_f, primia_stream_t stream);
int primia_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw,
                      float* db, int N, int in_f, int out_f, primia_stream_t stream);
/* Hard labels: loss = sum_n w[t_n] * nll_n / sum_n w[t_n]; class_weight may be NULL.  A label outside [0, C)
 * (torch raises IndexError there) turns the loss and that sample's dlogits row into NaN: nothing is read out of bounds. */
int primia_xent_hard(const float* logits, const int64_t* target, const float* class_weight,
                     float* loss, float* dlogits, int N, int C, primia_stream_t stream);
/* Soft labels: loss = mean_n[(sum_c w_c t_nc) * (-sum_c t_nc * logsoftmax(o)_nc)]. */
int primia_xent_soft(const float* logits, const float* target, const float* class_weight,
                     float* loss, float* dlogits, int N, int C, primia_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimizers over a flat fp32 arena — replaces torch.optim.SGD(lr, weight_decay) without
 * momentum and torch-1.4 torch.optim.Adam (train.py:280-303).
 * ------------------------------------------------------------------------------------------ */
int primia_sgd_step(float* p, const float* g, int64_t n, float lr, float weight_decay,
                    primia_stream_t stream);
/* The same SGD step over n <= 32 element ranges [begin, begin + len) of the arena in one launch (host arrays): what
 * primia_conv_sgd_step_many leaves over (BatchNorm / fc parameters, the stem filter). */
int primia_sgd_step_ranges(float* p, const float* g, const int64_t* begin_host, const int64_t* len_host, int n,
                           float lr, float weight_decay, primia_stream_t stream);
int primia_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n,
                     float lr, float beta1, float beta2, float eps, float weight_decay,
                     int64_t step, primia_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * FedAvg helpers — the arithmetic of aggregation() (torchlib/utils.py:1000-1092) on the flat
 * arena; the exchange itself is an RCCL all-reduce issued by the host.
 * ------------------------------------------------------------------------------------------ */
int primia_scale(float* x, int64_t n, float a, primia_stream_t stream);  /* x *= a   (theta_k * w_k) */
int primia_divide(float* x, int64_t n, float d, primia_stream_t stream); /* x /= d   (sum / K)       */
/* y += a * x — in-process clients: sum_k theta_k * w_k without a collective (utils.py:1087-1090). */
int primia_axpy(float* y, const float* x, int64_t n, float a, primia_stream_t stream);
/* q = int64(trunc(float32(x) * float32(scale)))  (FixedPrecisionTensor.fix_precision,
 * syft/frameworks/torch/tensors/interpreters/precision.py:117-132). */
int primia_fx_encode(const float* x, int64_t* q, int64_t n, float scale, primia_stream_t stream);
/* x = float32(q) / scale  (float_precision, precision.py:134-144). */
int primia_fx_decode(const int64_t* q, float* x, int64_t n, float scale, primia_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Exchange steps on an RCCL communicator of the CALLER (`comm` = its ncclComm_t; SURVEY.md §8b "comm").  RCCL is
 * looked up in the running process (ncclAllReduce), not linked: PRIMIA_ERR_UNSUPPORTED when the process has none.
 * ------------------------------------------------------------------------------------------ */
int primia_comm_available(void);   /* 1 when ncclAllReduce resolves */
/* aggregation() + send_new_models() (torchlib/utils.py:1000-1105) for one client per rank, in place on the client's
 * flat fp32 arena of n elements (parameters + BatchNorm running statistics): weight >= 0: theta *= weight, sum over
 * the communicator (weighted averaging, weights sum to 1); weight < 0: sum, then / nclients.  secure != 0 reproduces
 * the secure aggregation's numerics: fix_prec(10^precision_fractional) -> int64 ring sum -> float_prec
 * (utils.py:1046-1060,1079-1085); `scratch` holds n int64.  (Masking the encoded update before the sum, as
 * primia_amd.fed.PairwiseMasks does, is the caller's: primia_chacha20_fill + primia_ring_add / primia_ring_sub.) */
int primia_fedavg_allreduce(float* flat, int64_t n, float weight, int nclients, int secure, int precision_fractional,
                            int64_t* scratch, void* comm, primia_stream_t stream);
/* The two-party open of the SPDZ / FSS protocols (mpc/spdz.py:162-176: delta = sum(shares_delta); mpc/fss.py:158:
 * sum(shares)): buf += the peer's buf (mod 2^64), in place, over a 2-rank communicator. */
int primia_open2(int64_t* buf, int64_t n, void* comm, primia_stream_t stream);

/* ==========================================================================================
 * Encrypted-inference path: fixed-precision additive secret sharing over Z_2^64 (int64 with
 * wrap-around), Beaver triples and Function Secret Sharing — the per-share functions PySyft
 * registers with @allow_command and runs on each party (syft/generic/utils.py:27-55).
 * Every function below is PARTY-LOCAL: it sees one party's shares / keys plus opened (public)
 * values; the exchange ("open" = sum of the two parties' buffers) is done by the host.
 * All tensors int64 unless stated; n = element count.
 * ========================================================================================== */

/* The crypto provider's randomness: n int64 words of ChaCha20 keystream (key k0..k3 little-endian words, 64-bit
 * nonce, starting at 64-byte block `block0`; 8 words per block).  Replaces the process-seeded Mersenne Twisters the
 * reference's provider draws triples, masks and FSS seeds from (mpc/beaver.py:31-34,
 * tensors/interpreters/additive_shared.py:336-365, mpc/fss.py:344-358,495-501).  `out` must be 16-byte aligned. */
int primia_chacha20_fill(uint64_t k0, uint64_t k1, uint64_t k2, uint64_t k3, uint64_t nonce, uint64_t block0,
                         int64_t* out, int64_t n, primia_stream_t stream);
/* The same keystream with the provider's block counter kept ON THE DEVICE: block = *counter + block_offset + i.  A serving
 * deployment captures its whole refill (mpc/primitives.py:161-235: the provider re-provisions the crypto store between
 * requests) as ONE hipGraph; a host-side counter would be frozen into the graph and every replay would hand out the same
 * masks.  primia_u64_add(counter, blocks drawn) is the graph's last node: every replay draws fresh keystream. */
int primia_chacha20_fill_ctr(uint64_t k0, uint64_t k1, uint64_t k2, uint64_t k3, uint64_t nonce, const uint64_t* counter,
                             uint64_t block_offset, int64_t* out, int64_t n, primia_stream_t stream);
int primia_u64_add(uint64_t* word, uint64_t delta, primia_stream_t stream);
/* build_triple's second product share from input shares that were drawn directly (mpc/beaver.py:7-63: a, b uniform, c = a o b
 * split additively; a = a0 + a1 with both shares uniform IS a sharing of a uniform a): c1 = (x0 + x1) o (y0 + y1) - c0.
 * _mul_: element-wise, (y0, y1) of nb elements broadcast over the leading dims of the n-element x (nb divides n);
 * _matmul_: [M, K] @ [K, N], scratch = M * K + K * N int64. */
int primia_triple_mul_c1(const int64_t* x0, const int64_t* x1, const int64_t* y0, const int64_t* y1, const int64_t* c0,
                         int64_t* c1, int64_t n, int64_t nb, primia_stream_t stream);
int primia_triple_matmul_c1(const int64_t* a0, const int64_t* a1, const int64_t* b0, const int64_t* b1, const int64_t* c0,
                            int64_t* c1, int64_t* scratch, int M, int K, int N, primia_stream_t stream);

/* out[i] = a[i] (+|-|*) b[i % nb]  mod 2^64.  nb == n: plain element-wise; nb < n broadcasts b
 * over the leading dims ([HW, C] op [C], additive_shared.py:489-524, beaver.py:33-53). */
int primia_ring_add(const int64_t* a, const int64_t* b, int64_t* out, int64_t n, int64_t nb,
                    primia_stream_t stream);
int primia_ring_sub(const int64_t* a, const int64_t* b, int64_t* out, int64_t n, int64_t nb,
                    primia_stream_t stream);
int primia_ring_mul(const int64_t* a, const int64_t* b, int64_t* out, int64_t n, int64_t nb,
                    primia_stream_t stream);
/* out[i] = a[i] * k  (public integer scalar, AST._public_mul). */
int primia_ring_scale(const int64_t* a, int64_t k, int64_t* out, int64_t n, primia_stream_t stream);
/* out[i] = x[i] / d with C truncation toward zero — each party divides ITS OWN share
 * (FixedPrecisionTensor.truncate -> AST._public_div, precision.py:146-160,
 * additive_shared.py:672-678; torch-1.4 integer `/`). */
int primia_trunc_div(const int64_t* x, int64_t d, int64_t* out, int64_t n, primia_stream_t stream);
/* out[r] = sum_{k < w} x[r*w + k]  (per-share window sum of AST.mean, additive_shared.py:719-729). */
int primia_ring_rowsum(const int64_t* x, int64_t* out, int64_t rows, int64_t w, primia_stream_t stream);

/* out[r][j] = x[r][start + j], j < len: contiguous copy of a column range of x [rows, w]
 * (the tensor4d[:, :, :, a:b] slices of max_half_split, nn/functional.py:489-495). */
int primia_ring_slice_cols(const int64_t* x, int64_t* out, int64_t rows, int64_t w, int64_t start,
                           int64_t len, primia_stream_t stream);

/* C[M,N] = A[M,K] @ B[K,N]  (+ C if accumulate) mod 2^64 — torch.matmul on LongTensors
 * (mpc/spdz.py:54-59). */
int primia_ring_matmul(const int64_t* A, const int64_t* B, int64_t* C, int M, int K, int N,
                       int accumulate, primia_stream_t stream);

/* _pre_conv (nn/functional.py:78-166): x [B,C,H,W] -> im [B, Ho*Wo, C*R*S], rows ordered
 * (ho, wo), columns (c, r, s); zero padding. */
int primia_im2col_syft(const int64_t* x, int64_t* im, int B, int C, int H, int W, int R, int S,
                       int stride, int pad, primia_stream_t stream);
/* _post_conv (nn/functional.py:169-201): res [B, Ho*Wo, O] (+ bias[O]) -> [B, O, Ho, Wo]. */
int primia_col2out_syft(const int64_t* res, const int64_t* bias, int64_t* out, int B, int HoWo,
                        int O, primia_stream_t stream);
/* _pre_pool (nn/functional.py:311-393): x [B,C,H,W] -> [B, C, Ho*Wo, k*k], zero padding. */
int primia_pool_unroll_syft(const int64_t* x, int64_t* out, int B, int C, int H, int W, int k,
                            int stride, int pad, primia_stream_t stream);

/* FixedPrecisionTensor.reciprocal(method="newton") (precision.py:507-518: C = 20, 80 steps — what eval-mode
 * batch_norm calls on running_var, nn/functional.py:62) when BOTH parties' shares live in this process: the whole
 * iteration for n elements in one launch, bit-identical to the chain of primia_ring_sub / primia_beaver_combine_mul /
 * primia_trunc_div calls it replaces and consuming the provider's primitives in the same order.  `prim` is a DEVICE
 * array of 1 + 79 * 19 pointers to int64 buffers: the mask of the first re-shared constant [1], then per iteration
 * triple(x*x) as (a0, b0, c0, a1, b1, c1) [n each], triple(v*xx), the constant's mask [1], triple(y*x). */
int primia_newton_reciprocal_local(const int64_t* v0, const int64_t* v1, const int64_t* const* prim, int64_t scale,
                                   int64_t* x0, int64_t* x1, int64_t n, primia_stream_t stream);
/* ---- in-process deployment: both parties' shares on this GPU (what inference.py's VirtualWorker run is) ----------
 * With both shares side by side an "open" (mpc/spdz.py:162-176) is an addition, so a layer's chain
 * mask -> open -> combine -> truncate -> re-layout runs as ONE pass in which a thread does what each party does, in the
 * party's own arithmetic; the provider's primitives are requested by the host in the reference's order and handed in by
 * pointer.  Bit-identical to the step-by-step entry points above (tests/test_gpu_secure_local.py); the three-role
 * deployment keeps those.  Suffix _2p: the same per-share operation for both parties in one launch; _local: a protocol
 * step that includes its opens.  Triples are (a0, b0, c0, a1, b1, c1): a masks the first operand, b the second.
 *   primia_ring_ew_2p            o_j = a_j + b_j (op 0) | a_j - b_j (op 1), b broadcast when nb < n
 *   primia_fpt_mul_local         FPT * FPT (precision.py:309-316): Beaver product + each party's truncation by `div`
 *                                (0: none) + optional addend; y (nb elements) broadcasts
 *   primia_dif_eval_local        fss.le(x1, x2) (mpc/fss.py:97-185): mask_builder, the mod-2^32 open and both parties'
 *                                DIF evaluations; x1 / x2 may be column ranges [start, start + len) of a [rows][w]
 *                                matrix; x1 == NULL: shares of zero (AST.relu, additive_shared.py:922-925)
 *   primia_max_combine_local     left + (right >= left) * (right - left) (nn/functional.py:494) on column ranges
 *   primia_bn_eval_local         batch_norm in eval mode (nn/functional.py:44-75) of one image: NCHW in, NCHW out, both
 *                                FPT products and the row / column re-layouts inside; t1 / t2: HOST arrays of the two
 *                                triples' six pointers (t1: a ~ inv [C], b, c ~ rows [HW, C]; t2: a, c ~ rows, b ~ weight)
 *   primia_im2col_syft_2p, primia_pool_unroll_syft_2p   the PySyft layouts of both shares
 *   primia_beaver_matmul_local   spdz_mul "matmul": both opens, then z_j = c_j + delta @ (b_j [+ eps]) + a_j @ eps for
 *                                both parties in one grid; scratch = primia_beaver_matmul_local_scratch_elems int64
 *   primia_trunc_col2out_2p      each party's truncation of its product share + conv2d's output re-layout (+ bias) */
int primia_ring_ew_2p(int op, const int64_t* a0, const int64_t* a1, const int64_t* b0, const int64_t* b1, int64_t* o0,
                      int64_t* o1, int64_t n, int64_t nb, primia_stream_t stream);
int primia_fpt_mul_local(const int64_t* x0, const int64_t* x1, const int64_t* y0, const int64_t* y1, const int64_t* a0,
                         const int64_t* b0, const int64_t* c0, const int64_t* a1, const int64_t* b1, const int64_t* c1,
                         const int64_t* add0, const int64_t* add1, int64_t* z0, int64_t* z1, int64_t n, int64_t nb,
                         int64_t div, primia_stream_t stream);
int primia_dif_eval_local(const int64_t* x1_0, const int64_t* x1_1, int w1, int start1, const int64_t* x2_0,
                          const int64_t* x2_1, int w2, int start2, int len, const uint64_t* alpha0, const uint64_t* alpha1,
                          const uint64_t* s0_0, const uint64_t* s0_1, const uint8_t* cw_bits, const uint64_t* cw_sigma,
                          const uint64_t* cw_s, const int32_t* cw_leaf, int64_t* out0, int64_t* out1, int64_t n,
                          primia_stream_t stream);
int primia_max_combine_local(const int64_t* bit0, const int64_t* bit1, const int64_t* left0, const int64_t* left1, int wl,
                             int start_left, const int64_t* right0, const int64_t* right1, int wr, int start_right,
                             const int64_t* a0, const int64_t* b0, const int64_t* c0, const int64_t* a1, const int64_t* b1,
                             const int64_t* c1, int64_t* out0, int64_t* out1, int64_t rows, int len,
                             primia_stream_t stream);
int primia_bn_eval_local(const int64_t* x0, const int64_t* x1, const int64_t* mean0, const int64_t* mean1,
                         const int64_t* inv0, const int64_t* inv1, const int64_t* w0, const int64_t* w1,
                         const int64_t* bias0, const int64_t* bias1, const int64_t* const* t1, const int64_t* const* t2,
                         int64_t* out0, int64_t* out1, int C, int HW, int64_t div, primia_stream_t stream);
int primia_im2col_syft_2p(const int64_t* x0, const int64_t* x1, int64_t* im0, int64_t* im1, int B, int C, int H, int W, int R,
                          int S, int stride, int pad, primia_stream_t stream);
int64_t primia_beaver_matmul_local_scratch_elems(int M, int K, int N);
int primia_beaver_matmul_local(const int64_t* x0, const int64_t* x1, const int64_t* y0, const int64_t* y1, const int64_t* a0,
                               const int64_t* b0, const int64_t* c0, const int64_t* a1, const int64_t* b1, const int64_t* c1,
                               int64_t* z0, int64_t* z1, int64_t* scratch, int M, int K, int N, primia_stream_t stream);
int primia_trunc_col2out_2p(const int64_t* res0, const int64_t* res1, const int64_t* bias0, const int64_t* bias1,
                            int64_t* out0, int64_t* out1, int B, int HoWo, int O, int64_t div, primia_stream_t stream);
int primia_pool_unroll_syft_2p(const int64_t* x0, const int64_t* x1, int64_t* out0, int64_t* out1, int B, int C, int H, int W,
                               int k, int stride, int pad, primia_stream_t stream);
/* spdz_compute (mpc/spdz.py:63-122), party j in {0,1}:
 *   mul   : z = delta*b + a*eps + c (+ delta*eps if j == 0), element-wise; b / eps hold nb
 *           elements and broadcast over the leading dims when nb < n;
 *   matmul: z[M,N] = delta[M,K] @ b[K,N] + a[M,K] @ eps[K,N] + c (+ delta @ eps if j == 0);
 *           `scratch` holds K*N int64 (b + eps for party 0).
 * spdz_mask (spdz.py:21-45) is primia_ring_sub. */
/* spdz_mask of both operands of a product in one launch: d = x - a (n elements), e = y - b (nb elements) — two
 * primia_ring_sub calls; and spdz_compute (mul) followed by the party's own truncation of its share by `div`
 * (FixedPrecisionTensor.__mul__, precision.py:309-316, 356-358: toward zero, like primia_trunc_div) without writing the
 * product share in between.  Party-local: valid in every deployment; bit-identical to the separate calls. */
int primia_beaver_mask(const int64_t* x, const int64_t* a, int64_t* d, int64_t n, const int64_t* y, const int64_t* b,
                       int64_t* e, int64_t nb, primia_stream_t stream);
int primia_beaver_combine_mul_trunc(int j, const int64_t* delta, const int64_t* eps, const int64_t* a,
                                    const int64_t* b, const int64_t* c, int64_t* z, int64_t n, int64_t nb,
                                    int64_t div, primia_stream_t stream);
int primia_beaver_combine_mul(int j, const int64_t* delta, const int64_t* eps, const int64_t* a,
                              const int64_t* b, const int64_t* c, int64_t* z, int64_t n,
                              int64_t nb, primia_stream_t stream);
int primia_beaver_combine_matmul(int j, const int64_t* delta, const int64_t* eps, const int64_t* a,
                                 const int64_t* b, const int64_t* c, int64_t* z, int64_t* scratch,
                                 int M, int K, int N, primia_stream_t stream);

/* ---- Function Secret Sharing (mpc/fss.py), n comparisons, lambda = 127, 32-bit inputs --------
 * Key layout (struct of arrays, comparison index fastest):
 *   s0        uint64 [2][n]        party's initial seed (word 0 has 63 bits)
 *   cw_bits   uint8  [32][n]       DIF: bit0 tauL, bit1 tL, bit2 tauR, bit3 tR;  DPF: bit0 tL, bit1 tR
 *   cw_sigma  uint64 [32][2][n]    DIF only
 *   cw_s      uint64 [32][2][n]
 *   cw_leaf   int32  [33][n]       DIF;  DPF: int64 [n]
 * = the content of the reference key tuple (alpha, s0, *_CW, CW_leaf), 1,204 B per DIF key. */
/* mask_builder (fss.py:189-204): r = (x1 - x2) + alpha_share. */
int primia_fss_mask(const int64_t* x1, const int64_t* x2, const uint64_t* alpha_share, int64_t* r,
                    int64_t n, primia_stream_t stream);
/* opened mask: x = (r0 + r1) mod 2^32 (fss.py:158). */
int primia_fss_open(const int64_t* r0, const int64_t* r1, uint32_t* x, int64_t n,
                    primia_stream_t stream);
/* DIF.eval (fss.py:400-428): out = party b's int64 share of [x <= alpha]. */
int primia_dif_eval(int b, const uint32_t* x, const uint64_t* s0, const uint8_t* cw_bits,
                    const uint64_t* cw_sigma, const uint64_t* cw_s, const int32_t* cw_leaf,
                    int64_t* out, int64_t n, primia_stream_t stream);
/* DPF.eval (fss.py:320-338): share of [x == alpha]. */
int primia_dpf_eval(int b, const uint32_t* x, const uint64_t* s0, const uint8_t* cw_bits,
                    const uint64_t* cw_s, const int64_t* cw_n, int64_t* out, int64_t n,
                    primia_stream_t stream);
/* Dealer, build_fss_keys' arithmetic on raw keystream words, in place (mpc/primitives.py:237-253: alpha and its mask are
 * drawn below 2^32, party 0 receives (alpha - mask) mod 2^32 and party 1 the mask; mpc/fss.py:495-501: word 0 of a seed
 * carries 63 bits): alpha [n] &= 2^32 - 1, r [n] &= 2^32 - 1, s0_pair [2][2][n] word 0 of both parties &= 2^63 - 1,
 * alpha0 [n] = (alpha - r) mod 2^32. */
int primia_fss_alpha_split(uint64_t* alpha, uint64_t* s0_pair, uint64_t* r, uint64_t* alpha0, int64_t n,
                           primia_stream_t stream);
/* Dealer: DIF.keygen / DPF.keygen (fss.py:344-398, 286-318) from explicit randomness:
 * alpha uint64 [n] (< 2^32), s0_pair uint64 [2 parties][2][n].  Correction words are common to
 * both parties; party b's key = (s0_pair[b], cw_*). */
int primia_dif_keygen(const uint64_t* alpha, const uint64_t* s0_pair, uint8_t* cw_bits,
                      uint64_t* cw_sigma, uint64_t* cw_s, int32_t* cw_leaf, int64_t n,
                      primia_stream_t stream);
int primia_dpf_keygen(const uint64_t* alpha, const uint64_t* s0_pair, uint8_t* cw_bits,
                      uint64_t* cw_s, int64_t* cw_n, int64_t n, primia_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PRIMIA_HIP_H */
