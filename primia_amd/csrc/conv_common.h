// Geometry shared by the convolution kernels.
//
// Reduction ("k") axis of the implicit GEMM, as stored in the fwd-layout weight copy [K][klen]:
//   regular conv : e = (r*S + s)*C + c,            klen = R*S*C            (C % 64 == 0)
//   stem (7x7/2, C stored as 4): e = r*32 + s*4 + c with s < 8, c < 4, klen = 256 — one 32-element
//                  group per kernel row = 8 consecutive input pixels x 4 channels, which are
//                  contiguous in NHWC memory; s == 7, c == 3 and r == 7 carry zero weights.
#pragma once
#include "common.h"

namespace primia {

struct ConvGeom {
    int N, H, W, C, K, R, S, stride, pad, Ho, Wo;
    int klen;
    int stem;

    __host__ bool init(const primia_conv_desc& d) {
        N = d.N; H = d.H; W = d.W; C = d.C; K = d.K; R = d.R; S = d.S;
        stride = d.stride; pad = d.pad; Ho = d.Ho; Wo = d.Wo;
        if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0)
            return false;
        if (Ho != (H + 2 * pad - R) / stride + 1 || Wo != (W + 2 * pad - S) / stride + 1) return false;
        if (Ho <= 0 || Wo <= 0) return false;
        stem = (C == 4 && R == 7 && S == 7 && stride == 2 && pad == 3);
        if (stem) {
            klen = 256;
        } else {
            if (C % 64 != 0) return false;
            klen = R * S * C;
        }
        if (K % 64 != 0) return false;
        return true;
    }
    __host__ __device__ void decode_k(int e, int& r, int& s, int& c) const {
        if (stem) {
            r = e >> 5;
            s = (e & 31) >> 2;
            c = e & 3;
        } else {
            c = e % C;
            int t = e / C;
            s = t % S;
            r = t / S;
        }
    }
    __host__ __device__ int encode_k(int r, int s, int c) const {
        return stem ? (r * 32 + s * 4 + c) : ((r * S + s) * C + c);
    }
};


}  // namespace primia
