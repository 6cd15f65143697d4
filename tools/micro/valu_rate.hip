// Issue rate of the integer vector instructions the FSS kernels are made of (csrc/fss.hip: v_alignbit_b32, v_lshl_add_u64,
// v_bitop3_b32, v_add_u32, v_add_co/v_addc pairs), W waves per SIMD, CH independent chains per wave, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate
// Prints wave-instructions per SIMD and cycle at the nominal 2.4 GHz, and the shader clock derived from s_memtime against
// s_memrealtime (100 MHz) inside the same kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

enum Op { ADD32, ALIGNBIT, LSHLADD64, BITOP3, ADD64PAIR, XOR32, MIX, FMA, FMAC, MULF, MAXF, CNDMASK, LSHLOR, ANDOR, PERM, CVTPK, MAX3F, PKFMA, PKMUL, MAXU, LSHLREV, FMA_SGPR, DOT2C, PKADD, BFE, CMPCND, CMPCND_S, MULCLAMP };

template <int OP, int CH>
__global__ __launch_bounds__(512) void k(uint64_t* out, int iters, uint64_t* clk) {
    uint32_t a[CH], b[CH];
    uint64_t q[CH];
    for (int c = 0; c < CH; ++c) {
        a[c] = threadIdx.x * 2654435761u + c;
        b[c] = threadIdx.x * 40503u + 7 * c + 1;
        q[c] = ((uint64_t)a[c] << 32) | b[c];
    }
    const uint64_t t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (OP == ADD32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
                if (OP == XOR32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
                if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[c]) : "v"(b[c]));
                if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(a[c]) : "v"(b[c]));
                if (OP == LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[c]) : "v"(q[(c + 1) % CH]));
                if (OP == ADD64PAIR) {
                    asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc"
                                 : "+v"(a[c]), "+v"(b[c]) : "v"(b[(c + 1) % CH]), "v"(a[(c + 1) % CH]) : "vcc");
                }
                if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(b[(c + 1) % CH]));
                if (OP == FMA_SGPR) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "s"(iters), "v"(b[c]));
                if (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(b[(c + 1) % CH]));
                if (OP == MULF) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
                if (OP == MAXF) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
                if (OP == MAXU) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
                if (OP == LSHLREV) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[c]));
                if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(b[c]) : );
                if (OP == LSHLOR) asm volatile("v_lshl_or_b32 %0, %0, 4, %1" : "+v"(a[c]) : "v"(b[c]));
                if (OP == ANDOR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(b[(c + 1) % CH]));
                if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(b[(c + 1) % CH]));
                if (OP == CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
                if (OP == MAX3F) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(b[(c + 1) % CH]));
                if (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q[c]) : "v"(q[(c + 1) % CH]));
                if (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[c]) : "v"(q[(c + 1) % CH]));
                if (OP == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[c]) : "v"(q[(c + 1) % CH]));
                if (OP == DOT2C) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[c]) : "v"(b[c]), "v"(b[(c + 1) % CH]));
                if (OP == BFE) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(a[c]));
                if (OP == CMPCND) asm volatile("v_cmp_gt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(b[c]) : "vcc");
                if (OP == CMPCND_S) asm volatile("v_cmp_gt_f32 s[20:21], %1, %0\n\tv_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[c]) : "v"(b[c]) : "s20", "s21");
                if (OP == MULCLAMP) asm volatile("v_mul_f32 %0, %1, %0 clamp\n\tv_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b[c]));
                if (OP == MIX) {      // the SHA-512 round's proportions: 3 alignbit : 2 lshl_add_u64 : 2 bitop3
                    asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[c]) : "v"(b[c]));
                    asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[c]) : "v"(q[(c + 1) % CH]));
                    asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(b[c]) : "v"(a[c]));
                    asm volatile("v_alignbit_b32 %0, %0, %1, 9" : "+v"(a[c]) : "v"(b[c]));
                    asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[c]) : "v"(q[(c + 2) % CH]));
                    asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(b[c]) : "v"(a[c]));
                    asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(a[c]) : "v"(b[c]));
                }
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    uint64_t s = 0;
    for (int c = 0; c < CH; ++c) s += a[c] + b[c] + q[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int OP, int CH>
void run(const char* name, int threads, int per_iter) {
    const int iters = 20000, grid = 256;
    uint64_t *out, *clk;
    hipMalloc(&out, (size_t)grid * threads * 8);
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP, CH><<<grid, threads>>>(out, 200, clk);
    hipEventRecord(e0);
    k<OP, CH><<<grid, threads>>>(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double insts = (double)iters * 8 * CH * per_iter * (threads / 64) * grid;       // wave-instructions
    const double per_simd_cycle = insts / (1024.0 * 2.4e9 * ms * 1e-3);
    printf("%-28s waves/SIMD=%d chains=%d: %.4f wave-instr / SIMD / cycle@2.4GHz (%.1f G/s)  s_memtime/s_memrealtime = %.2f -> %.0f MHz\n",
           name, threads / 256, CH, per_simd_cycle, insts / ms / 1e6, (double)h[0] / h[1], 100.0 * h[0] / h[1]);
    hipFree(out); hipFree(clk);
}

int main() {
    run<ADD32, 8>("v_add_u32", 512, 1);
    run<ADD32, 8>("v_add_u32", 256, 1);
    run<XOR32, 8>("v_xor_b32", 512, 1);
    run<ALIGNBIT, 8>("v_alignbit_b32", 512, 1);
    run<BITOP3, 8>("v_bitop3_b32", 512, 1);
    run<LSHLADD64, 8>("v_lshl_add_u64", 512, 1);
    run<ADD64PAIR, 8>("v_add_co + v_addc", 512, 2);
    run<MIX, 4>("SHA-512 mix 3:2:2", 512, 7);
    // the floating-point / packing instructions of the stem kernels (stem_bwd_fused, stem_fwd_fused, bn.hip pooling)
    run<FMA, 8>("v_fma_f32 (3 VGPR sources)", 512, 1);
    run<FMA_SGPR, 8>("v_fma_f32 (1 SGPR source)", 512, 1);
    run<FMAC, 8>("v_fmac_f32 (VOP2)", 512, 1);
    run<MULF, 8>("v_mul_f32", 512, 1);
    run<MAXF, 8>("v_max_f32", 512, 1);
    run<MAXU, 8>("v_max_u32", 512, 1);
    run<LSHLREV, 8>("v_lshlrev_b32 (1 source)", 512, 1);
    run<CNDMASK, 8>("v_cndmask_b32 (vcc)", 512, 1);
    run<LSHLOR, 8>("v_lshl_or_b32", 512, 1);
    run<ANDOR, 8>("v_and_or_b32", 512, 1);
    run<PERM, 8>("v_perm_b32", 512, 1);
    run<CVTPK, 8>("v_cvt_pk_bf16_f32", 512, 1);
    run<MAX3F, 8>("v_max3_f32", 512, 1);
    run<PKFMA, 8>("v_pk_fma_f32 (2 lanes-ops)", 512, 1);
    run<PKMUL, 8>("v_pk_mul_f32 (2 lane-ops)", 512, 1);
    run<PKADD, 8>("v_pk_add_f32 (2 lane-ops)", 512, 1);
    run<DOT2C, 8>("v_dot2c_f32_bf16", 512, 1);
    run<BFE, 8>("v_bfe_i32", 512, 1);
    run<CMPCND, 8>("v_cmp_gt_f32 vcc + v_cndmask_b32 vcc", 512, 2);
    run<CMPCND_S, 8>("v_cmp_gt_f32 sgpr + v_cndmask_b32 sgpr", 512, 2);
    run<MULCLAMP, 8>("v_mul_f32 clamp + v_mul_f32", 512, 2);
    return 0;
}
