// Process-wide tuning / dispatch options of the library (include/primia_hip.h: primia_set_option).
// The library never reads the environment: every switch that selects between kernels or sizes a launch is an entry of
// this table, changed only by an explicit primia_set_option(name, value) call of the host.
#pragma once

namespace primia {

#define PRIMIA_OPTIONS(X)                                                                                              \
    X(lh2, 1)               /* 0: the wide 3x3 / stride-1 layers stay on the implicit GEMM */                         \
    X(lh2_bm, 0)            /* 392 | 196: force the linear-halo tile height (0: by shape) */                          \
    X(lh4, 1)               /* 196-pixel tiles on conv3x3_lh4_kernel (8 matrix + 4 loader waves, one barrier per step); 0: conv3x3_lh2 */ \
    X(lh_fwd_maxw, 30)      /* widest image the linear-halo kernel takes in the forward pass */                        \
    X(c64, 1)               /* 0: layer1's 64 -> 64 convolutions stay on the implicit GEMM */                          \
    X(c64_blocks, 512)      /* persistent blocks of conv3x3_c64_kernel (2 per CU) */                                   \
    X(c64_bnsums, 1)        /* primia_conv2d_dgrad_bnsums also on the 64 -> 64 layers (layer1) */                                   \
    X(c64_dbg, 0)           /* timing experiments only (results are WRONG when set): conv3x3_c64_kernel's debug bits */           \
    X(c64_stages, 4)        /* ring depth of conv3x3_c64_kernel: 3 | 4 */                                              \
    X(conv_cfg, 4)          /* implicit-GEMM tile configuration 0..5 (conv_igemm.hip, a..f) */                         \
    X(fwd_pair, 1)          /* transition blocks: conv1 + downsample forward in one grid */                            \
    X(s2lh, 1)              /* transition blocks on conv_s2lh_kernel (parity planes, linear halo), bits: 1 dgrad of <= 128-channel dx, 2 forward, 4 dgrad at every width; 0: implicit GEMM */ \
    X(s2lh_dx_max, 64)      /* bit 1 of s2lh: widest dx (channels) whose data gradient takes conv_s2lh_kernel */                    \
    X(s2lh_blocks, 0)       /* its block count (0: one per CU) */                                                    \
    X(s2lh_dbg, 0)          /* measurement only (wrong results): 1 no halo DMA after the prologue, 2 no weight DMA, 4 no MFMA, 8 no stores, 16 no write-back, 32 no BN partials, 64 linear halo */ \
    X(dgrad_classes, 1)     /* stride-2 data gradient as four parity classes */                                        \
    X(dgrad_cls_inner, 1)   /* ... the four classes of a pixel tile dispatched back to back (they share dy rows in L2); 0: class-major */ \
    X(wgrad_kernel, 0)      /* 0: by shape; 1 register-staged per-tap; 2 LDS-DMA per-tap (skips patch / tap kernels) */ \
    X(wgt_blocks, 1024)     /* block target of the fp32 / per-sample per-tap weight gradient */                        \
    X(wg_blocks, 504)       /* block target of conv_wgrad_dma_kernel */                                                \
    X(wgtap, 1)             /* conv_wgrad_tap_kernel for the stride-2 / 1x1 layers */                                  \
    X(wgtap_blocks, 0)      /* its block target (0: one round of CUs) */                                               \
    X(wgtap_persample, 1)   /* ... and its DP-SGD norm pass with whole images per block */                             \
    X(wgp_pairimg, 1)       /* DP-SGD norm pass of the patch kernel: whole images per half-block */                    \
    X(wgp_shape, -1)        /* sub-patch shape 0..2 = 8x8 | 8x4 | 16x2 (-1: by image size) */                          \
    X(wgp_blocks, 0)        /* block target of conv_wgrad_patch33_kernel (0: one per CU) */                            \
    X(wgp_order, 0)         /* block id order: 0 slab fastest (the slabs of a pixel range share x, dy in one L2), 1 pixel range fastest */                                                          \
    X(wgp_stages88, 3)      /* ring depth for 8 x 8 sub-patches: 3 | 4 */                                              \
    X(wgp_group, 1)         /* same-shape layers of a stage in one launch */                                           \
    X(wgp_lw, 1)            /* 3x3 weight gradient: four matrix waves (one per SIMD) + four loader waves instead of the two halves */ \
    X(wgp_group_minfill, 90) /* % of the CUs a grouped launch must fill */                                             \
    X(dp_keep_mb, 160)      /* DP-SGD: per-sample tiles of a layer are kept up to this many MiB */                     \
    X(dp_ghost, 1)          /* DP-SGD: Gram-matrix norms for the 7 x 7 layers */                                       \
    X(gn_sample, 1)         /* GroupNorm reductions with one block per sample (batch >= 128) */                        \
    X(stem_blocks, 256)     /* block target of stem_conv_fwd_kernel */                                                 \
    X(stem_wgrad_halo, 1)   /* 0: conv1's weight gradient on the per-tap kernel */                                     \
    X(stem_wg_blocks, 512)  /* block target of stem_conv_wgrad_kernel */                                               \
    X(bn_minrows, 32)       /* least pixel rows per block of a BatchNorm reduction */                                  \
    X(bn_unroll, 2)         /* chunks per thread and trip of bn_bwd_apply_kernel: 1 | 2 */

enum OptId : int {
#define PRIMIA_OPT_ENUM(name, def) kOpt_##name,
    PRIMIA_OPTIONS(PRIMIA_OPT_ENUM)
#undef PRIMIA_OPT_ENUM
        kOptCount
};

extern int g_options[kOptCount];

inline int opt(OptId id) { return __atomic_load_n(&g_options[id], __ATOMIC_RELAXED); }

}  // namespace primia

#define PRIMIA_OPT(name) (::primia::opt(::primia::kOpt_##name))
