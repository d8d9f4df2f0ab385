// ubench_valu.hip -- issue-rate microbenchmark of the VALU instructions the eps draw is made of
// (diagnostic, not part of the product path).  Each kernel runs ITERS x 8 independent chains of one
// instruction per wave; 4 waves per SIMD on every CU.  Prints time relative to v_xor_b32.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)

#define DEFK(NAME, ASM)                                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters, uint32_t seed)         \
    {                                                                                            \
        uint32_t a[8];                                                                           \
        uint64_t w[8];                                                                           \
        for (int i = 0; i < 8; ++i) { a[i] = (threadIdx.x * 2654435761u + i * 40503u + seed) | 0x3f000001u; w[i] = a[i]; }         \
        for (int it = 0; it < iters; ++it) {                                                     \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) ASM;                                   \
        }                                                                                        \
        uint32_t s = 0;                                                                          \
        for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)w[i] + (uint32_t)(w[i] >> 32);         \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                 \
    }

DEFK(k_xor, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7])))
DEFK(k_bitop3, asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(seed)))
DEFK(k_mad64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(a[i]), "s"(seed) : "vcc"))
DEFK(k_mulhi, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "s"(seed)))
DEFK(k_mullo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(seed)))
DEFK(k_mad24, asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "s"(seed)))
DEFK(k_fma, asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7])))
DEFK(k_pkfma, asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w[i])))
DEFK(k_log, asm volatile("v_log_f32 %0, %0" : "+v"(a[i])))
DEFK(k_exp, asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])))
DEFK(k_sin, asm volatile("v_sin_f32 %0, %0" : "+v"(a[i])))
DEFK(k_sqrt, asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i])))
DEFK(k_rcp, asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i])))
DEFK(k_cvtbf, asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7])))
DEFK(k_cvtu, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i])))
// the same instructions with the uniform operand in a VGPR instead of an SGPR / literal
DEFK(k_xor_s, asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(seed)))
DEFK(k_bitop3_v, asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7])))
DEFK(k_mad64_v, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc"))
DEFK(k_mulhi_v, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7])))
DEFK(k_fma3, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7])))
DEFK(k_xor2, asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(seed)))
// literal / inline constants
DEFK(k_mul_lit, asm volatile("v_mul_f32 %0, 0x3f317217, %0" : "+v"(a[i])))
DEFK(k_mul_inl, asm volatile("v_mul_f32 %0, -2.0, %0" : "+v"(a[i])))
DEFK(k_mul_v, asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7])))
DEFK(k_fmac_lit, asm volatile("v_fmac_f32 %0, 0x3f317217, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7])))
DEFK(k_fmac_v, asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7])))
DEFK(k_fmamk, asm volatile("v_fmamk_f32 %0, %0, 0x33800000, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7])))
DEFK(k_add_inl, asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a[i])))
DEFK(k_lshr, asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(a[i])))
DEFK(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7])))
DEFK(k_cmp_s, asm volatile("v_cmp_lt_f32 vcc, %1, %0" : "+v"(a[i]) : "s"(seed) : "vcc"))
DEFK(k_cmp_v, asm volatile("v_cmp_lt_f32 vcc, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]) : "vcc"))
DEFK(k_pkmul, asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(w[i])))
// a transcendental next to plain VALU work: do they overlap (separate unit) or add up?
DEFK(k_log_xor, asm volatile("v_log_f32 %0, %0\n v_xor_b32 %1, %1, %0\n v_xor_b32 %1, %1, %0\n v_xor_b32 %1, %1, %0" : "+v"(a[i]), "+v"(a[(i + 4) & 7])))
DEFK(k_mad64_xor, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0\n v_xor_b32 %1, %1, %1\n v_xor_b32 %1, %1, %1" : "=v"(w[i]), "+v"(a[i]) : "s"(seed) : "vcc"))

// ---- do MFMA and VALU work of DIFFERENT waves on one SIMD overlap?  One 512-thread workgroup per CU = 2 waves per
// SIMD (wave w -> SIMD w % 4): waves 0-3 run `mf` x 8 independent v_mfma_f32_16x16x32_bf16 per iteration, waves 4-7
// run `va` x 8 v_fma_f32 per iteration.  mode 1: MFMA waves only, 2: VALU waves only, 3: both.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void k_overlap(uint32_t *out, int iters, int mode, int swap)
{
    const int wave = threadIdx.x >> 6;
    const bool mfma_wave = swap ? (wave & 1) == 0 : wave < 4;       // swap: roles alternate wave by wave (same SIMD pairs differ)
    if (mfma_wave) {
        if (!(mode & 1)) return;
        f32x4_t acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        bf16x8_t a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)1.0f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * 512 + threadIdx.x] = __float_as_uint(s);
    } else {
        if (!(mode & 2)) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += v[i];
        out[blockIdx.x * 512 + threadIdx.x] = __float_as_uint(s);
    }
}

typedef void (*kern_t)(uint32_t *, int, uint32_t);

int main()
{
    uint32_t *out;
    const int blocks = 256 * 4, iters = 4096;
    if (hipMalloc(&out, blocks * 256 * 4) != hipSuccess) { printf("no device\n"); return 1; }
    struct { const char *n; kern_t k; int per; } ks[] = {
        {"v_xor_b32", k_xor, 1}, {"v_bitop3_b32", k_bitop3, 1}, {"v_mad_u64_u32", k_mad64, 1}, {"v_mul_hi_u32", k_mulhi, 1},
        {"v_mul_lo_u32", k_mullo, 1}, {"v_mad_u32_u24", k_mad24, 1}, {"v_fma_f32", k_fma, 1}, {"v_pk_fma_f32", k_pkfma, 1},
        {"v_log_f32", k_log, 1}, {"v_exp_f32", k_exp, 1}, {"v_sin_f32", k_sin, 1}, {"v_sqrt_f32", k_sqrt, 1}, {"v_rcp_f32", k_rcp, 1},
        {"v_cvt_pk_bf16_f32", k_cvtbf, 1}, {"v_cvt_f32_u32", k_cvtu, 1}, {"v_xor_b32 sgpr", k_xor_s, 1}, {"v_bitop3 vvv", k_bitop3_v, 1}, {"v_mad_u64 vgpr", k_mad64_v, 1}, 
        {"v_mul_hi vgpr", k_mulhi_v, 1}, {"v_fma_f32 3 regs", k_fma3, 1}, {"2 x v_xor (v, s)", k_xor2, 1}, {"v_mul_f32 literal", k_mul_lit, 1}, {"v_mul_f32 inline", k_mul_inl, 1}, {"v_mul_f32 vgpr", k_mul_v, 1}, {"v_fmac literal", k_fmac_lit, 1},
        {"v_fmac vgpr", k_fmac_v, 1}, {"v_fmamk literal", k_fmamk, 1}, {"v_add_f32 inline", k_add_inl, 1}, {"v_lshrrev", k_lshr, 1}, {"v_cndmask", k_cndmask, 1},
        {"v_cmp sgpr", k_cmp_s, 1}, {"v_cmp vgpr", k_cmp_v, 1}, {"v_pk_mul_f32", k_pkmul, 1}, {"log+3xor", k_log_xor, 1}, {"mad64+2xor", k_mad64_xor, 1}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float base = 0;
    for (auto &k : ks) {
        hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 64, 1u);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        if (base == 0) base = best;
        // 4 waves per SIMD, iters*8 instruction groups each
        const double ns_per = best * 1e6 / ((double)iters * 8 * 4);
        printf("%-20s %8.3f ms  %6.2f ns per wave-group per SIMD  x%.2f of v_xor_b32 (assume 4 clk -> %.1f clk)\n", k.n, best, ns_per,
               best / base, 4.0 * best / base);
    }
    // MFMA / VALU overlap: per iteration 8 MFMAs (8 x 16 = 128 matrix-pipe cycles at peak rate) against 32 v_fma (128 issue cycles)
    for (int swap = 0; swap < 2; ++swap)
        for (int mode = 1; mode <= 3; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k_overlap, dim3(256), dim3(512), 0, 0, out, 8192, mode, swap);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("overlap swap=%d %-10s %8.3f ms\n", swap, mode == 1 ? "mfma only" : mode == 2 ? "valu only" : "both", best);
        }
    return 0;
}
