// Optimizer steps and FedAvg arithmetic over flat fp32 arenas (HBM streaming, float4 accesses).
#include "common.h"

namespace primia {

template <typename F>
__global__ __launch_bounds__(256) void flat_kernel(F f, long n) {
    // vector body on groups of 4, scalar tail
    const long nv = n >> 2;
    const long stride = (long)gridDim.x * 256;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nv; q += stride) f.vec(q);
    const long t = (nv << 2) + (long)blockIdx.x * 256 + threadIdx.x;
    if (t < n) f.one(t);
}

struct SgdFn {
    float* p;
    const float* g;
    float lr, wd;
    // torch.optim.SGD without momentum: d_p = g + wd*p ; p = p - lr*d_p
    __device__ __forceinline__ float upd(float p_, float g_) const { return sgd_update(p_, g_, lr, wd); }
    __device__ __forceinline__ void vec(long q) const {
        f32x4 a = ((f32x4*)p)[q];
        const f32x4 b = ((const f32x4*)g)[q];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = upd(a[i], b[i]);
        ((f32x4*)p)[q] = a;
    }
    __device__ __forceinline__ void one(long i) const { p[i] = upd(p[i], g[i]); }
};

struct AdamFn {
    float* p;
    const float* g;
    float* m;
    float* v;
    float beta1, beta2, eps, wd, step_size, inv_sqrt_bc2;
    // torch-1.4 torch.optim.Adam (L2-coupled weight decay)
    __device__ __forceinline__ void upd(float& p_, float g_, float& m_, float& v_) const {
        g_ = g_ + wd * p_;
        m_ = m_ * beta1 + (1.f - beta1) * g_;
        v_ = v_ * beta2 + (1.f - beta2) * g_ * g_;
        const float denom = sqrtf(v_) * inv_sqrt_bc2 + eps;
        p_ = p_ - step_size * (m_ / denom);
    }
    __device__ __forceinline__ void vec(long q) const {
        f32x4 a = ((f32x4*)p)[q], mm = ((f32x4*)m)[q], vv = ((f32x4*)v)[q];
        const f32x4 b = ((const f32x4*)g)[q];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float pi = a[i], mi = mm[i], vi = vv[i];
            upd(pi, b[i], mi, vi);
            a[i] = pi; mm[i] = mi; vv[i] = vi;
        }
        ((f32x4*)p)[q] = a;
        ((f32x4*)m)[q] = mm;
        ((f32x4*)v)[q] = vv;
    }
    __device__ __forceinline__ void one(long i) const { upd(p[i], g[i], m[i], v[i]); }
};

struct ScaleFn {
    float* x;
    float a;
    __device__ __forceinline__ void vec(long q) const {
        f32x4 v = ((f32x4*)x)[q];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= a;
        ((f32x4*)x)[q] = v;
    }
    __device__ __forceinline__ void one(long i) const { x[i] *= a; }
};

struct AxpyFn {
    float* y;
    const float* x;
    float a;
    __device__ __forceinline__ void vec(long q) const {
        f32x4 v = ((f32x4*)y)[q];
        const f32x4 u = ((const f32x4*)x)[q];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(v[i], __fmul_rn(u[i], a));  // product rounded first, as torch
        ((f32x4*)y)[q] = v;
    }
    __device__ __forceinline__ void one(long i) const { y[i] = __fadd_rn(y[i], __fmul_rn(x[i], a)); }
};

struct DivFn {
    float* x;
    float d;
    __device__ __forceinline__ void vec(long q) const {
        f32x4 v = ((f32x4*)x)[q];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] / d;
        ((f32x4*)x)[q] = v;
    }
    __device__ __forceinline__ void one(long i) const { x[i] = x[i] / d; }
};

__global__ __launch_bounds__(256) void fx_encode_kernel(const float* __restrict__ x, int64_t* __restrict__ q,
                                                        long n, float scale) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        // float32 multiply, then truncation toward zero (.long()), precision.py:121
        const float up = x[i] * scale;
        q[i] = (int64_t)up;
    }
}
__global__ __launch_bounds__(256) void fx_decode_kernel(const int64_t* __restrict__ q, float* __restrict__ x,
                                                        long n, float scale) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] = (float)q[i] / scale;
}

static inline int flat_blocks(long n) {
    long b = ((n >> 2) + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b > 2048 ? 2048 : b);
}
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace primia

using namespace primia;

extern "C" {

int primia_sgd_step(float* p, const float* g, int64_t n, float lr, float weight_decay,
                    primia_stream_t stream) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(p && g && n >= 0 && aligned16(p) && aligned16(g));
    if (n == 0) return PRIMIA_OK;
    SgdFn f{p, g, lr, weight_decay};
    flat_kernel<<<flat_blocks(n), 256, 0, (hipStream_t)stream>>>(f, n);
    return launch_status();
}

int primia_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n,
                     float lr, float beta1, float beta2, float eps, float weight_decay,
                     int64_t step, primia_stream_t stream) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(p && g && exp_avg && exp_avg_sq && n >= 0 && step >= 1);
    PRIMIA_REQUIRE(aligned16(p) && aligned16(g) && aligned16(exp_avg) && aligned16(exp_avg_sq));
    if (n == 0) return PRIMIA_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamFn f{p, g, exp_avg, exp_avg_sq, beta1, beta2, eps, weight_decay, (float)((double)lr / bc1),
             (float)(1.0 / sqrt(bc2))};
    flat_kernel<<<flat_blocks(n), 256, 0, (hipStream_t)stream>>>(f, n);
    return launch_status();
}

int primia_scale(float* x, int64_t n, float a, primia_stream_t stream) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(x && n >= 0 && aligned16(x));
    if (n == 0) return PRIMIA_OK;
    ScaleFn f{x, a};
    flat_kernel<<<flat_blocks(n), 256, 0, (hipStream_t)stream>>>(f, n);
    return launch_status();
}

int primia_axpy(float* y, const float* x, int64_t n, float a, primia_stream_t stream) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(y && x && n >= 0 && aligned16(y) && aligned16(x));
    if (n == 0) return PRIMIA_OK;
    AxpyFn f{y, x, a};
    flat_kernel<<<flat_blocks(n), 256, 0, (hipStream_t)stream>>>(f, n);
    return launch_status();
}

int primia_divide(float* x, int64_t n, float d, primia_stream_t stream) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(x && n >= 0 && aligned16(x) && d != 0.f);
    if (n == 0) return PRIMIA_OK;
    DivFn f{x, d};
    flat_kernel<<<flat_blocks(n), 256, 0, (hipStream_t)stream>>>(f, n);
    return launch_status();
}

int primia_fx_encode(const float* x, int64_t* q, int64_t n, float scale, primia_stream_t stream) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(x && q && n >= 0);
    if (n == 0) return PRIMIA_OK;
    fx_encode_kernel<<<flat_blocks(n * 4), 256, 0, (hipStream_t)stream>>>(x, q, n, scale);
    return launch_status();
}
int primia_fx_decode(const int64_t* q, float* x, int64_t n, float scale, primia_stream_t stream) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(x && q && n >= 0);
    if (n == 0) return PRIMIA_OK;
    fx_decode_kernel<<<flat_blocks(n * 4), 256, 0, (hipStream_t)stream>>>(q, x, n, scale);
    return launch_status();
}

}  // extern "C"
