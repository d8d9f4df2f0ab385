// bnn_gemm_params.hpp -- kernel-argument block shared by the contraction kernels.
#pragma once
#include "bnn_device.hpp"
#include "bnn_kl_body.hpp"

namespace bnn {

enum { A_DENSE = 0, A_IM2COL = 1 };
// B_SAMPLED_T: the drawn matrix is used TRANSPOSED (input gradient: reduction over the weight's rows)
enum { B_PLAIN = 0, B_SAMPLED = 1, B_SAMPLED_T = 2 };

// internal epilogue flag (not part of the C-ABI flags): row m = (image, pixel) -> y[image][n][pixel], i.e. the
// conv output in NCHW; OH * OW pixels per image, O channels
constexpr int kFlagStoreNCHW = 1 << 16;
// bits of the sticky device error word (word 0 of the registered workspace)
constexpr unsigned kDevErrHandoffTimeout = 1u;   // a bounded LDS hand-off wait of the fused linear kernel gave up
// internal compute mode of the fast linear kernel (not a C-ABI value): fp32-accurate results on the bf16 MFMA
constexpr int kComputeBf16x3 = 2;

struct GemmParams {
    // A operand
    const float *A;
    int64_t a_sample_stride;
    int64_t lda;
    // im2col geometry (A_IM2COL)
    int32_t C, H, W, OH, OW, KH, KW, sh, sw, ph, pw, dh, dw, Cg;
    // B operand
    const float *Bw;            // plain weights (N_total, K)
    int64_t b_sample_stride;
    const float *mu;            // sampled weights
    const float *rho;
    // bias: plain (bias) or sampled (mu_b, rho_b)
    const float *bias;
    int64_t bias_sample_stride;
    const float *mu_b;
    const float *rho_b;
    // output
    float *Y;
    int64_t y_sample_stride;
    int64_t ldy;
    int32_t O;                  // conv: total output channels
    // extents: per group M x N x K
    int32_t M, N, K;
    int32_t S, G;
    int32_t ntn, ntm;           // tiles
    int32_t xcd_a;              // > 0: 2-D XCD map, xcd_a sample groups x (8 / xcd_a) panel groups (bnn_linear.hip)
    int32_t flags;
    int32_t vecA, vecB;         // 16-B loads legal
    RngDev rng_w, rng_b;
    // registered scratch (bnn_set_workspace): [device error word + reserved: 64 KiB][slabs: rest]
    unsigned *dev_err;          // sticky device error word (kDevErr* bits; bnn_check_device reads and clears it), or NULL
    float *ws_slabs;            // fixed-order partial sums of the backward kernels' sample / row splits
    int64_t ws_slab_bytes;
    // KL first pass carried by this launch (bnn_linear_forward_sampled_kl): workgroups >= gemm_grid run kl_piggy_block
    KlPiggy kl;
    int32_t gemm_grid;
    unsigned long long *dbg;    // diagnostic stamps (bnn_linear.hip, STAMPS build), normally NULL
    int32_t dbg_block;
};

int dispatch_linear_v2(GemmParams &p, bool sampled, int compute, hipStream_t st, const char *who);
int dispatch_linear_dgrad(GemmParams &p, int compute, hipStream_t st, const char *who);
void fill_workspace(GemmParams &p);

}  // namespace bnn
