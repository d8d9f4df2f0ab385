/*
 * bnn_hip.h -- C-ABI of libbnn_hip.so, the MI355X (gfx950) variational-layer engine.
 *
 * The reference (Mirko-Nava/BayesianNeuralNetworks, pytorch_bayesian 0.0.4) is
 * pure Python and has NO FFI: its hot path is a sequence of stock torch calls.
 * Each entry point below therefore replaces a reference *Python call site*; the
 * citation after "replaces" is that site, relative to /root/reference/.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless the
 *     parameter is documented "host"; no allocation and no synchronisation inside
 *     (every call is legal during hipGraph / torch.cuda.graph capture);
 *   - `stream` is a hipStream_t passed as void*;
 *   - return value: 0 = success, <0 = BNN_E_* argument error (nothing launched),
 *     >0 = the hipError_t of a failed launch; bnn_last_error() gives the text;
 *   - tensors are dense row-major fp32 unless a dtype argument says otherwise.
 *
 * RNG contract (production eps source; its CPU twin is oracle/bnn_oracle.c, orc_eps4)
 *   eps for element e of tensor-stream `stream`, MC sample `sample`:
 *     (x0..x3) = Philox4x32-10(counter = (e / 4, (stream << 16) | sample,
 *                                         epoch_host, epoch_dev),
 *                              key     = (seed & 0xffffffff, seed >> 32))
 *     u  = ((x >> 8) + 0.5) * 2^-24
 *     z0 = r(u0) cos(2 pi u1), z1 = r(u0) sin(2 pi u1), z2, z3 likewise from x2, x3,
 *     r(u) = sqrt(-2 ln u);   eps[e] = z[e % 4].
 *   epoch_dev is read from device memory (*rng->epoch_dev + rng->epoch_dev_delta) so
 *   that a replayed graph draws fresh noise: bnn_rng_advance bumps it in-stream.
 *   The draw depends only on (seed, stream, sample, epochs, e): not on tiling,
 *   grid size, or the number of GPUs the MC samples are sharded over.
 *
 *   rng->generator selects the eps source (part of the key: every kernel that re-creates a draw -- the draw launch, the
 *   fused GEMM, the standalone sampler, the backward -- reads it from the same struct, so a draw is the same everywhere):
 *     BNN_GEN_PHILOX10_U24 (0, default): the stream above -- 24-bit uniforms, four eps per Philox4x32-10 block.
 *     BNN_GEN_PHILOX7_U16  (1): EIGHT eps per Philox4x32-7 block, from 16-bit uniforms -- for weights that are rounded to
 *       bf16 (8 significand bits) anyway; 0.4 x the integer work per eps (the draw launch of the BASELINE net: 16.5 -> 13.3 us):
 *         (x0..x3) = Philox4x32-7(counter = (e / 8, (stream << 16) | sample, epoch_host, epoch_dev), key as above)
 *         word x_k feeds elements 8 (e / 8) + 2 k and + 2 k + 1:  ua = ((x_k & 0xffff) + 0.5) 2^-16,  ub = ((x_k >> 16) + 0.5) 2^-16,
 *         z_even = r(ua) cos(2 pi ub), z_odd = r(ua) sin(2 pi ub);  |eps| <= r(2^-17) = 4.86.
 *       Philox4x32-7 is the 7-round member of the same family (Random123's kat_vectors hold its known answers, checked in
 *       tests/test_oracle_golden.py); CPU twin: orc_eps_fill_gen.
 */
#ifndef BNN_HIP_H
#define BNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BNN_ABI_VERSION 2

enum {
    BNN_OK = 0,
    BNN_E_NULL = -1,      /* required pointer is NULL */
    BNN_E_SHAPE = -2,     /* negative / zero / inconsistent extent */
    BNN_E_DTYPE = -3,     /* unknown dtype code */
    BNN_E_ALIGN = -4,     /* pointer not aligned to the element size */
    BNN_E_RANGE = -5,     /* value outside the supported range (e.g. >65535 samples) */
    BNN_E_UNSUPPORTED = -6,
    BNN_E_DEVICE = -7     /* a kernel reported an internal error through the device error word (bnn_check_device) */
};

enum { BNN_F32 = 0, BNN_BF16 = 1,
       BNN_BF16X3 = 2 /* an fp32 value as THREE bf16 planes h, m, l: h = bf16(v), m = bf16(v - h), l = bf16(v - h - m), v = h + m + l
                       * to 2^-24 |v| -- the operand format of the fp32 parity mode's dense contraction (bnn_dense_forward_x3) */ };

/* Compute mode of the contraction kernels. */
enum {
    BNN_COMPUTE_F32 = 0,  /* fp32 operands, fp32 accumulate; the 1e-5 parity mode.  Wide forward layers on the fast path run
                           * it as three-way bf16 splits on v_mfma_f32_16x16x32_bf16 (a = ah + am + al exactly; the six
                           * largest partial products, dropped terms <= 2^-25 |a b|, i.e. below one fp32 rounding) --
                           * 1.4 x faster than v_mfma_f32_16x16x4_f32, which everything else in this mode uses and which
                           * BNN_F32_MFMA=native (environment) selects everywhere */
    BNN_COMPUTE_BF16 = 1  /* operands rounded to bf16 (RNE), v_mfma_f32_16x16x32_bf16, fp32 accumulate */
};

enum { BNN_GEN_PHILOX10_U24 = 0, BNN_GEN_PHILOX7_U16 = 1 };

/* One eps stream (host struct, read at call time). */
typedef struct bnn_rng {
    uint64_t seed;
    uint32_t stream;            /* tensor-stream id, < 65536 */
    uint32_t sample0;           /* id of the first MC sample of this call; sample s uses sample0 + s */
    uint32_t epoch_host;        /* host-side draw counter */
    int32_t epoch_dev_delta;    /* added to *epoch_dev (e.g. -1 to re-create the previous replay's draw) */
    const uint32_t *epoch_dev;  /* device word bumped by bnn_rng_advance; NULL = 0 */
    uint32_t generator;         /* BNN_GEN_* (RNG contract above) */
    uint32_t reserved;          /* 0 */
} bnn_rng_t;

/* One Gaussian posterior tensor with its Gaussian prior (host struct). */
typedef struct bnn_kl_tensor {
    const float *mu;
    const float *rho;
    int64_t n;
    float prior_mu;
    float prior_sigma;
} bnn_kl_tensor_t;

/* ---- library / device queries ------------------------------------------- */
int bnn_abi_version(void);
const char *bnn_arch(void);          /* "gfx950" */
const char *bnn_last_error(void);    /* text of the last non-zero return on this thread */
/* Number of kernel launches issued through this library so far (tests use it to
 * prove the HIP path ran). */
uint64_t bnn_launch_count(void);

/* Optional per-device scratch: `bytes` >= 128 KiB of ZEROED device memory that stays valid until replaced
 * (ptr = NULL unregisters).  Layout: word 0 = the sticky DEVICE ERROR WORD, 64 KiB reserved, then slabs for
 * the fixed-order partial sums of the backward kernels that split samples / rows over workgroups (without the
 * workspace those entry points return BNN_E_UNSUPPORTED).  Host call, not stream-ordered: register before
 * launching.  Launches use the workspace of the CURRENT device: make the operands' device current. */
int bnn_set_workspace(int device, void *ptr, int64_t bytes);
/* Reads (and clears) the device error word of `device` after synchronising `stream`: BNN_OK, or BNN_E_DEVICE
 * when a kernel gave up on an internal protocol since the last check -- today only the bounded LDS hand-off
 * waits of the fused linear kernel, which then skip their tile's store instead of storing numbers computed on
 * an undrawn buffer.  A SYNCHRONISING host call (not graph-capturable): call it at a sync / check point. */
int bnn_check_device(int device, void *stream);

/* ---- K1: posterior draw --------------------------------------------------
 * replaces  WeightNormal.stddev / WeightNormal.sample
 *           pytorch_bayesian/nn/core.py:25-27, 44-45
 *   out = mu + (1e-10 + softplus(rho)) * eps        (softplus: beta 1, threshold 20)
 */
/* eps supplied by the caller (parity mode: eps from torch's CPU generator). */
int bnn_sample_affine_eps(const float *mu, const float *rho, const float *eps,
                          void *out, int64_t n, int out_dtype, void *stream);
/* eps drawn in-kernel; writes `nsamples` draws, draw s at out + s * out_sample_stride
 * elements. */
int bnn_sample_affine_philox(const float *mu, const float *rho, void *out, int64_t n,
                             int nsamples, int64_t out_sample_stride, int out_dtype,
                             const bnn_rng_t *rng, void *stream);
/* The raw eps stream (fp32), same layout as above. */
int bnn_eps_philox(float *out, int64_t n, int nsamples, int64_t out_sample_stride,
                   const bnn_rng_t *rng, void *stream);
/* sigma = 1e-10 + softplus(rho)   (WeightNormal.stddev, core.py:25-27) */
int bnn_sigma(const float *rho, float *out, int64_t n, void *stream);

/* Backward of K1 (what autograd derives from core.py:44-45), summed over samples:
 *   g_mu[e]  (+)= sum_s g_w[s][e]
 *   g_rho[e] (+)= sum_s g_w[s][e] * eps_s[e] * sigmoid(rho[e])
 * eps: external (eps != NULL, sample stride eps_sample_stride) or Philox (rng != NULL).
 * accumulate != 0 adds into g_mu / g_rho instead of overwriting. */
int bnn_sample_affine_bwd(const float *g_w, int64_t g_w_sample_stride, const float *rho,
                          const float *eps, int64_t eps_sample_stride, const bnn_rng_t *rng,
                          int64_t n, int nsamples, float *g_mu, float *g_rho, int accumulate,
                          void *stream);

/* epoch_dev[0] += inc, in stream order (a 1-thread kernel; graph-capturable). */
int bnn_rng_advance(uint32_t *epoch_dev, uint32_t inc, void *stream);

/* ---- K3: closed-form Gaussian KL ------------------------------------------
 * replaces  KLDivergence.compute_kl / KLDivergence.forward
 *           pytorch_bayesian/nn/loss.py:16-28, 30-38
 *           (torch.distributions.kl._kl_normal_normal)
 *   kl_e = 0.5 * (r + t - 1 - ln r),  r = (sigma/sigma_p)^2,  t = ((mu - mu_p)/sigma_p)^2
 * out[0..ntensors-1] = per-tensor SUMS (the reference's .mean() = sum / n);
 * out[ntensors]      = mean_t(sum_t / n_t) / n_batches   (loss.py:38).
 * `tensors` is a host array; `workspace` needs bnn_kl_workspace_bytes(ntensors) bytes; calls
 * sharing a workspace must be stream-ordered.
 * Deterministic: fixed-order two-pass reduction, no float atomics.  (Folding the second pass
 * into the first behind a last-workgroup ticket was measured SLOWER on MI355X: 23 us against
 * 6 + 7 us -- 1190 same-address atomics across 8 XCDs serialise.) */
int64_t bnn_kl_workspace_bytes(int ntensors);
int bnn_kl_forward(const bnn_kl_tensor_t *tensors, int ntensors, float n_batches,
                   float *out, void *workspace, void *stream);
/* The same result in two calls, for a step that ends in the MC reduction (examples/MNIST/uncertainty.py:50 after
 * nn/loss.py:30-38): bnn_kl_forward_partial launches only the first pass; bnn_mc_sum_kl (below) runs the second pass
 * as one extra workgroup of the reduction's launch.  Every launch costs >= 4 us on MI355X; this removes one from
 * each forward.  Same tensors / workspace in both calls, stream-ordered; values bit-identical to bnn_kl_forward. */
int bnn_kl_forward_partial(const bnn_kl_tensor_t *tensors, int ntensors, void *workspace, void *stream);
/* Backward: for tensor t,  g_mu (+)= scale_t * (mu - mu_p)/sigma_p^2,
 *   g_rho (+)= scale_t * (sigma/sigma_p^2 - 1/sigma) * sigmoid(rho),
 * scale_t = *upstream (device scalar, may be NULL = 1) / (n_t * ntensors * n_batches). */
int bnn_kl_backward(const bnn_kl_tensor_t *tensors, int ntensors, float n_batches,
                    const float *upstream, float *const *g_mu, float *const *g_rho,
                    int accumulate, void *stream);

/* ---- K2: sampled linear ----------------------------------------------------
 * replaces  NormalLinear.forward   pytorch_bayesian/nn/dense.py:56-60
 *           (sample(): dense.py:46-54 -> core.py:44-45, then F.linear)
 * and, with nsamples > 1, the MC loop of BayesianNetworkModule.forward
 *           pytorch_bayesian/nn/container.py:32-37  for this layer.
 *
 *   for s in [0, nsamples):
 *     w_s = mu_w + sigma(rho_w) * eps(rng_w, s)      (N, K)   never written to memory
 *     b_s = mu_b + sigma(rho_b) * eps(rng_b, s)      (N)      (mu_b == NULL: no bias)
 *     y[s] = x[s] @ w_s^T + b_s                      (M, N)
 *   x[s] = x + s * x_sample_stride (0 = every sample reads the same input),
 *   y[s] = y + s * y_sample_stride;  ldx / ldy = row strides in elements.
 * flags: BNN_FLAG_RELU applies max(.,0) in the epilogue. */
enum {
    BNN_FLAG_RELU = 1,
    /* bf16 compute mode only: x is bf16 in memory (ldx, strides in elements; K % 8 == 0,
     * 16-B aligned) / y is written as bf16.  Lets a chain of layers keep its hidden
     * activations in bf16: half the activation stream of the next layer. */
    BNN_FLAG_X_BF16 = 2,
    BNN_FLAG_Y_BF16 = 4
};
int bnn_linear_forward_sampled(const void *x, int64_t x_sample_stride, int64_t ldx,
                               const float *mu_w, const float *rho_w,
                               const float *mu_b, const float *rho_b,
                               void *y, int64_t y_sample_stride, int64_t ldy,
                               int64_t M, int64_t N, int64_t K, int nsamples,
                               const bnn_rng_t *rng_w, const bnn_rng_t *rng_b,
                               int compute, int flags, void *stream);
/* The same layer, and in the same launch the FIRST pass of the model's KL (bnn_kl_forward_partial): a narrow layer's
 * launch (N <= 16: a classifier head, 32 workgroups at the BASELINE shape) leaves most of MI355X's 256 CUs idle and a
 * launch of its own costs >= 4 us, so the KL partial sums ride along as extra workgroups.  When the layer is not a
 * narrow one on the fast path, or the model is large (>= 8 Mi scalars, > 8 tensors), the first pass is launched
 * separately by this call -- the result is the same either way; finish with bnn_mc_sum_kl (or run bnn_kl_forward).
 * replaces  nn/dense.py:56-60 + the first half of nn/loss.py:16-28. */
int bnn_linear_forward_sampled_kl(const void *x, int64_t x_sample_stride, int64_t ldx,
                                  const float *mu_w, const float *rho_w, const float *mu_b,
                                  const float *rho_b, void *y, int64_t y_sample_stride, int64_t ldy,
                                  int64_t M, int64_t N, int64_t K, int nsamples, const bnn_rng_t *rng_w,
                                  const bnn_rng_t *rng_b, int compute, int flags,
                                  const bnn_kl_tensor_t *tensors, int ntensors, void *kl_workspace, void *stream);
/* ---- draw-once path of the sampled linear layer (bf16 compute mode) ---------------------------------------
 * The same layer in two launches instead of one fused one: every posterior tensor of a forward is drawn ONCE
 * (all S MC samples, sigma computed once per weight) by bnn_draw_multi, then bnn_dense_forward contracts on the
 * drawn weights -- a dense MFMA GEMM with both operands arriving by LDS-DMA.  Same DrawKey -> the same draws as
 * the fused kernel and as bnn_sample_affine_philox, bit for bit (one device function).  On MI355X this is the
 * faster form at the BASELINE shapes: inside the GEMM the draw's ~150 VALU issue slots per 4 weights share each
 * SIMD's issue port with the MFMAs and pace the kernel through an LDS hand-off.
 *
 * One posterior tensor to draw (host struct, read at call time).
 * replaces  WeightNormal.sample  pytorch_bayesian/nn/core.py:44-45, called weight-then-bias by
 *           NormalLinear.sample  pytorch_bayesian/nn/dense.py:46-54, once per MC sample by the loop at
 *           pytorch_bayesian/nn/container.py:36-37 */
typedef struct bnn_draw_tensor {
    const float *mu;
    const float *rho;
    int64_t rows, cols;         /* posterior shape (N, K); a bias is (1, N).  rows > 1 needs cols % 4 == 0 */
    void *out;                  /* draw s at out + s * out_sample_stride elements: `rows` rows of `ld` elements */
    int64_t ld;                 /* >= cols (% 8 == 0 when rows > 1); columns cols .. ld - 1 are written as ZEROS */
    int64_t out_sample_stride;  /* elements */
    int out_dtype;              /* BNN_F32, BNN_BF16, or BNN_BF16X3: three planes, plane p of draw s at
                                 * out + (p * nsamples + s) * out_sample_stride (kinds 0, 1, 2) */
    int kind;                   /* 0: draw mu + sigma(rho) eps (rng used); 1: mu itself; 2: sigma(rho) itself (no eps: Flipout's
                                 * two operands, nsamples = 1); 3: `mu` as it is, written ONCE whatever nsamples is (rho ignored,
                                 * pass mu) -- with BNN_BF16X3 (planes out_sample_stride apart) this is bnn_split_bf16x3 of an
                                 * activation riding in the draw launch: the fp32 parity mode's input planes */
    int taps;                   /* 0 / 1: rows are written as they are.  KH * KW of a conv weight (O, C, KH, KW) viewed as
                                 * (O, C * KH * KW): element (o, c, t) is written to column t * C + c (tap-major, what
                                 * bnn_conv2d_dense_forward reads); the eps stream keeps the original element order */
    bnn_rng_t rng;
} bnn_draw_tensor_t;
/* Draws <= 8 tensors x nsamples MC samples in ONE launch.  kl_tensors != NULL: the launch also carries the first
 * pass of that model's KL (as bnn_linear_forward_sampled_kl does; same eligibility, same values) -- finish it with
 * bnn_mc_sum_kl; BNN_E_UNSUPPORTED (nothing launched) when the KL is not eligible. */
int bnn_draw_multi(const bnn_draw_tensor_t *tensors, int ntensors, int nsamples,
                   const bnn_kl_tensor_t *kl_tensors, int kl_ntensors, void *kl_workspace, void *stream);
/* y[s] = act(x[s] . w[s]^T + b[s]) on drawn weights: x (S or shared: x_sample_stride = 0) x M x ldx bf16,
 * w S x N x ldw bf16 with every row ZERO beyond K up to ldw >= roundup(K, 64) (what bnn_draw_multi writes),
 * b S x N fp32 or NULL, y fp32 or (BNN_FLAG_Y_BF16) bf16; BNN_FLAG_RELU.  K % 8 == 0, 16-B aligned rows.
 * N <= 16 (a classifier head) runs a K-split kernel without LDS staging.  fp32 accumulate (bf16 MFMA).
 * replaces  F.linear(x, *self.sampled)  pytorch_bayesian/nn/dense.py:60 */
int bnn_dense_forward(const void *x, int64_t x_sample_stride, int64_t ldx,
                      const void *w, int64_t w_sample_stride, int64_t ldw,
                      const float *b, int64_t b_sample_stride,
                      void *y, int64_t y_sample_stride, int64_t ldy,
                      int64_t M, int64_t N, int64_t K, int nsamples, int flags, void *stream);
/* A hidden layer AND the classifier head behind it in ONE launch (bf16 compute mode, inference): the hidden layer's output
 * act(x[s] w[s]^T + b[s]) is never stored -- every wave of the GEMM rounds its tile to bf16 (exactly what a stored bf16 hidden
 * activation holds) and contracts it with the matching columns of the head's drawn weights w_head (S x n_head x ldwh bf16, rows
 * zero beyond N up to ldwh, n_head <= 16), leaving PARTIAL logits
 *     partials[part][s][m][j],  part < bnn_dense_head_parts(M, N, nsamples),  fp32, M x n_head per (part, s);
 * partial 0 also carries the head's bias b_head (S x n_head fp32 or NULL).  The logits of sample s are the sum over `part`
 * (bnn_mc_sum with nsamples = parts, y_sample_stride = S * M * n_head, n = S * M * n_head), the predictive mean the sum over
 * (part, s) scaled by 1 / S (bnn_mc_sum / bnn_mc_sum_kl with nsamples = parts * S, y_sample_stride = M * n_head): one launch
 * (the head's own, >= 4 us) and the hidden activation's S * M * N * 2 bytes of stores fewer per forward.  Same bf16 products as
 * bnn_dense_forward twice; fp32 sums in another (fixed) order.
 * replaces  two consecutive F.linear(x, *self.sampled)  pytorch_bayesian/nn/dense.py:60 (+ the ReLU between them) */
int bnn_dense_head_parts(int64_t M, int64_t N, int nsamples);
int bnn_dense_forward_head(const void *x, int64_t x_sample_stride, int64_t ldx,
                           const void *w, int64_t w_sample_stride, int64_t ldw,
                           const float *b, int64_t b_sample_stride,
                           const void *w_head, int64_t wh_sample_stride, int64_t ldwh,
                           const float *b_head, int64_t bh_sample_stride, int64_t n_head,
                           float *partials, int64_t M, int64_t N, int64_t K, int nsamples, int flags, void *stream);
/* bnn_dense_forward_head in the fp32 PARITY mode: x, w and w_head are BNN_BF16X3 operands (plane strides in elements); the fp32
 * tile of the hidden layer is split into its three bf16 planes in the epilogue and contracted with the head's planes on the six
 * plane pairs of bnn_dense_forward_x3 (small pairs summed apart from (h, h)).  Partial logits as above, fp32. */
int bnn_dense_forward_x3_head(const void *x, int64_t x_plane_stride, int64_t x_sample_stride, int64_t ldx,
                              const void *w, int64_t w_plane_stride, int64_t w_sample_stride, int64_t ldw,
                              const float *b, int64_t b_sample_stride,
                              const void *w_head, int64_t wh_plane_stride, int64_t wh_sample_stride, int64_t ldwh,
                              const float *b_head, int64_t bh_sample_stride, int64_t n_head,
                              float *partials, int64_t M, int64_t N, int64_t K, int nsamples, int flags, void *stream);
/* The same layer in the fp32 PARITY mode (1e-5 against the reference) on the same kernel: x and w are BNN_BF16X3 operands
 * (plane p at + p * plane_stride elements; x from bnn_split_bf16x3 or a previous layer's BNN_FLAG_Y_BF16 output, w from
 * bnn_draw_multi with out_dtype BNN_BF16X3) and the contraction runs the six largest partial products of
 * (xh + xm + xl)(wh + wm + wl) on the bf16 MFMA with fp32 accumulation -- dropped terms <= 2^-25 |x w|, below one fp32
 * rounding; what BNN_COMPUTE_F32 does inside bnn_linear_forward_sampled.  y: fp32, or (BNN_FLAG_Y_BF16) three bf16
 * planes of the fp32 result, y_plane_stride apart, for the next layer.  N <= 16 (a classifier head: K <= 2048, fp32
 * outputs) runs the K-split kernel, the six plane pairs as six passes.
 * replaces  F.linear(x, *self.sampled)  pytorch_bayesian/nn/dense.py:60 */
int bnn_dense_forward_x3(const void *x, int64_t x_plane_stride, int64_t x_sample_stride, int64_t ldx,
                         const void *w, int64_t w_plane_stride, int64_t w_sample_stride, int64_t ldw,
                         const float *b, int64_t b_sample_stride,
                         void *y, int64_t y_plane_stride, int64_t y_sample_stride, int64_t ldy,
                         int64_t M, int64_t N, int64_t K, int nsamples, int flags, void *stream);
/* fp32 (rows x cols, row pitch ldx) -> BNN_BF16X3 planes (row pitch ld_out, planes plane_stride elements apart): the
 * input of the first bnn_dense_forward_x3 of a network.  cols % 8 == 0, 16-B aligned rows. */
int bnn_split_bf16x3(const float *x, int64_t rows, int64_t cols, int64_t ldx, void *out, int64_t ld_out, int64_t plane_stride,
                     void *stream);
/* bf16 (batch x rows x cols, row pitch ld_in) -> the transposes (batch x cols x ld_out), columns rows .. ld_out - 1 written as ZEROS:
 * drawn weights (S x N x ldw, from bnn_draw_multi) as the operand of the input gradient of a training step,
 *   gx[s] = gy[s] . w_s  =  bnn_dense_forward(x = gy, w = w_s^T (K rows of ld_out >= roundup(N, 64)), M, N' = K, K' = N)
 * -- the backward of F.linear (pytorch_bayesian/nn/dense.py:60, examples/MNIST/train.py:63-65) on the weights the forward drew,
 * with no second draw.  cols % 8 == 0, 16-B aligned rows on both sides. */
int bnn_transpose_bf16(const void *in, int64_t in_batch_stride, int64_t ld_in, void *out, int64_t out_batch_stride, int64_t ld_out,
                       int64_t rows, int64_t cols, int batch, void *stream);

/* Same contraction with the weights given (F.linear(x, w, b), dense.py:60):
 * w[s] = w + s * w_sample_stride, b[s] = b + s * b_sample_stride (b may be NULL). */
int bnn_linear_forward(const float *x, int64_t x_sample_stride, int64_t ldx,
                       const float *w, int64_t w_sample_stride,
                       const float *b, int64_t b_sample_stride,
                       float *y, int64_t y_sample_stride, int64_t ldy,
                       int64_t M, int64_t N, int64_t K, int nsamples,
                       int compute, int flags, void *stream);

/* ---- backward of K2 linear -----------------------------------
 * replaces  what autograd derives from F.linear(x, w, b) (dense.py:60) and
 *           w = mu + sigma(rho) * eps (core.py:44-45) in loss.backward()
 *           (examples/MNIST/train.py:63-65).  W is (N, K) row-major, as in the forward.
 *
 * Input gradient, fused with the re-creation of the forward's draw (same rng_w key):
 *     gx[s][m][k] = sum_n gy[s][m][n] * W_s[n][k],   W_s = mu_w + sigma(rho_w) * eps_s
 * flags: BNN_FLAG_X_BF16 = gy is bf16, BNN_FLAG_Y_BF16 = gx is written as bf16 (bf16 compute only).
 * Needs K % 4 == 0, N % 4 == 0 (fp32 gy) / N % 8 == 0 (bf16 gy), 16-B aligned operands;
 * returns BNN_E_UNSUPPORTED otherwise (callers then draw W_s with bnn_sample_affine_philox and
 * use bnn_linear_backward_input). */
int bnn_linear_backward_input_sampled(const void *gy, int64_t gy_sample_stride, int64_t ldgy,
                                      const float *mu_w, const float *rho_w,
                                      void *gx, int64_t gx_sample_stride, int64_t ldgx,
                                      int64_t M, int64_t N, int64_t K, int nsamples,
                                      const bnn_rng_t *rng_w, int compute, int flags, void *stream);
/* Same with the weights given: w[s] = w + s * w_sample_stride, fp32 (any shape). */
int bnn_linear_backward_input(const void *gy, int64_t gy_sample_stride, int64_t ldgy,
                              const float *w, int64_t w_sample_stride,
                              void *gx, int64_t gx_sample_stride, int64_t ldgx,
                              int64_t M, int64_t N, int64_t K, int nsamples, int flags, void *stream);
/* Weight gradient fused with the backward of the draw (dW_s = gy_s^T x_s is never stored):
 *     g_mu [n][k] (+)= sum_s dW_s[n][k]
 *     g_rho[n][k] (+)= sum_s dW_s[n][k] * eps_s[n][k] * sigmoid(rho_w[n][k])
 * x: (S, M, K) or shared (x_sample_stride = 0).  flags: BNN_FLAG_X_BF16 = x is bf16,
 * BNN_FLAG_Y_BF16 = gy is bf16 (bf16 compute only).  With few output tiles
 * the MC samples are split over workgroups through the registered workspace and added in a fixed
 * order: bitwise reproducible.
 * Bias (rho_b, g_mu_b, g_rho_b, rng_b all given, or all NULL): g_mu_b[n] (+)= sum_s c_s[n],
 * g_rho_b[n] (+)= sum_s c_s[n] * eps_b,s[n] * sigmoid(rho_b[n]), c_s = column sums of gy[s] -- taken
 * inside the same launch by the workgroups of k-tile 0 (one extra MFMA against a fragment of ones per
 * step), or by bnn_colsum + bnn_sample_affine_bwd when the samples are split.
 * KL (kl != NULL): the gradient of the layer's share of KLDivergence (loss.py:16-38) is added in the
 * same final store -- g_mu += c (mu - mu_p) / sigma_p^2, g_rho += c (sigma / sigma_p^2 - 1 / sigma)
 * sigmoid(rho), c = *upstream * scale -- instead of a separate bnn_kl_backward pass plus autograd's
 * accumulation adds.  scale_x = 1 / (n_x * ntensors * n_batches), n_x = elements of that tensor. */
typedef struct bnn_kl_fuse {
    const float *upstream;      /* device scalar: d loss / d KL */
    const float *mu_w;          /* (N, K) */
    const float *mu_b;          /* (N) or NULL */
    float scale_w, prior_mu_w, prior_sigma_w;
    float scale_b, prior_mu_b, prior_sigma_b;
} bnn_kl_fuse_t;
int bnn_linear_backward_weight_sampled(const void *x, int64_t x_sample_stride, int64_t ldx,
                                       const void *gy, int64_t gy_sample_stride, int64_t ldgy,
                                       const float *rho_w, float *g_mu, float *g_rho,
                                       const float *rho_b, float *g_mu_b, float *g_rho_b,
                                       int64_t M, int64_t N, int64_t K, int nsamples,
                                       const bnn_rng_t *rng_w, const bnn_rng_t *rng_b,
                                       const bnn_kl_fuse_t *kl,
                                       int compute, int flags, int accumulate, void *stream);
/* Whole backward of a NARROW layer (N <= 16, K % 4 == 0: a classifier head) in one pass over the
 * activations: gx (may be NULL), g_mu / g_rho of the weight, and the bias gradients -- same definitions
 * as the three entry points above.  gy fp32; flags: BNN_FLAG_X_BF16 = x is bf16, BNN_FLAG_Y_BF16 = gx
 * is written as bf16.  Needs the registered workspace (S * (2 N K + N) floats); BNN_E_UNSUPPORTED
 * otherwise (callers then use the general entry points). */
int bnn_linear_backward_narrow_sampled(const void *x, int64_t x_sample_stride, int64_t ldx,
                                       const float *gy, int64_t gy_sample_stride, int64_t ldgy,
                                       const float *mu_w, const float *rho_w,
                                       void *gx, int64_t gx_sample_stride, int64_t ldgx,
                                       float *g_mu, float *g_rho,
                                       const float *rho_b, float *g_mu_b, float *g_rho_b,
                                       int64_t M, int64_t N, int64_t K, int nsamples,
                                       const bnn_rng_t *rng_w, const bnn_rng_t *rng_b,
                                       const bnn_kl_fuse_t *kl, int flags, int accumulate, void *stream);
/* F.linear's own weight gradient, per sample: gw[s][n][k] (+)= sum_m gy[s][m][n] * x[s][m][k],
 * gw[s] = gw + s * gw_sample_stride (parity mode, where the draw is a separate op). */
int bnn_linear_backward_weight(const void *x, int64_t x_sample_stride, int64_t ldx,
                               const void *gy, int64_t gy_sample_stride, int64_t ldgy,
                               float *gw, int64_t gw_sample_stride,
                               int64_t M, int64_t N, int64_t K, int nsamples,
                               int compute, int flags, int accumulate, void *stream);
/* Bias: out[s][n] = sum_m gy[s][m][n]  (feed out to bnn_sample_affine_bwd with the bias key).
 * flags: BNN_FLAG_X_BF16 = gy is bf16. */
int bnn_colsum(const void *gy, int64_t gy_sample_stride, int64_t ldgy, float *out,
               int64_t M, int64_t N, int nsamples, int flags, void *stream);
/* Fused-ReLU layers: out[i] = y[i] > 0 ? g[i] : 0.  flags: BNN_FLAG_X_BF16 = g and out are bf16,
 * BNN_FLAG_Y_BF16 = y is bf16. */
int bnn_relu_backward(const void *g, const void *y, void *out, int64_t n, int flags, void *stream);

/* ---- K2: sampled conv2d (implicit GEMM) --------------------------------------
 * replaces  NormalConv2d.forward   pytorch_bayesian/nn/conv.py:112-119
 *   y[s] = conv2d(x[s], w_s, b_s, stride, padding, dilation, groups), NCHW / OIHW.
 * Implicit GEMM: M = B*OH*OW, N = O/groups, K = (C/groups)*KH*KW. */
typedef struct bnn_conv2d_shape {
    int32_t B, C, H, W;          /* input  (B, C, H, W) */
    int32_t O, KH, KW;           /* weight (O, C/groups, KH, KW) */
    int32_t stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups;
} bnn_conv2d_shape_t;
/* Two kernels behind these entry points.  FAST (groups == 1, K % 8 == 0, O >= 16, 16-B aligned
 * weights, a workspace of bnn_conv2d_workspace_bytes(shape, x_samples, compute) bytes, 16-B aligned):
 * an explicit im2col panel (bf16 in bf16 compute) written to the workspace, then the draw-paced
 * linear kernel (LDS-DMA activation rings, weights drawn once per 512 rows) with an NCHW-storing
 * epilogue.  GENERIC (workspace NULL / too small, or any other shape): one implicit-GEMM kernel with
 * scalar im2col loaders.  x_samples = 1 when x_sample_stride == 0 (a shared input is expanded once),
 * else nsamples.  bnn_conv2d_workspace_bytes returns 0 when only the generic kernel applies. */
int64_t bnn_conv2d_workspace_bytes(const bnn_conv2d_shape_t *shape, int x_samples, int compute);
int bnn_conv2d_forward_sampled(const float *x, int64_t x_sample_stride,
                               const float *mu_w, const float *rho_w,
                               const float *mu_b, const float *rho_b,
                               float *y, int64_t y_sample_stride,
                               const bnn_conv2d_shape_t *shape, int nsamples,
                               const bnn_rng_t *rng_w, const bnn_rng_t *rng_b,
                               int compute, int flags,
                               void *workspace, int64_t workspace_bytes, void *stream);
int bnn_conv2d_forward(const float *x, int64_t x_sample_stride,
                       const float *w, int64_t w_sample_stride,
                       const float *b, int64_t b_sample_stride,
                       float *y, int64_t y_sample_stride,
                       const bnn_conv2d_shape_t *shape, int nsamples,
                       int compute, int flags,
                       void *workspace, int64_t workspace_bytes, void *stream);

/* The conv on DRAWN weights as a true implicit GEMM (bf16 compute mode; no im2col panel, no workspace): w is what
 * bnn_draw_multi writes for the (O, C * KH * KW) posterior with taps = KH * KW -- S x O x ldw bf16, tap-major columns
 * (t * C + c), rows zero-padded to ldw >= roundup(C * KH * KW, 64).  A workgroup keeps its images in LDS (bf16, padded,
 * channel-last) and gathers the im2col rows in the address of its MFMA fragment reads; the weight tile streams by
 * LDS-DMA.  x fp32 NCHW (x_sample_stride = 0: shared), b S x O fp32 or NULL, y fp32 NCHW.  Built for groups = 1,
 * C = 64 or a multiple of 128, O = 64 or 128, <= 128 output pixels per image, one padded image + weight ring within
 * LDS (both BASELINE conv shapes); BNN_E_UNSUPPORTED otherwise (use bnn_conv2d_forward_sampled).
 * replaces  F.conv2d(x, *self.sampled, ...)  pytorch_bayesian/nn/conv.py:116-119 */
int bnn_conv2d_dense_forward(const float *x, int64_t x_sample_stride,
                             const void *w, int64_t w_sample_stride, int64_t ldw,
                             const float *b, int64_t b_sample_stride,
                             float *y, int64_t y_sample_stride,
                             const bnn_conv2d_shape_t *shape, int nsamples, int flags, void *stream);

/* The same convolution in the fp32 PARITY mode (1e-5 against the reference): w is a BNN_BF16X3 operand (three bf16 planes,
 * w_plane_stride elements apart: bnn_draw_multi with out_dtype BNN_BF16X3 and taps = KH * KW), the images are split into
 * three bf16 planes as they become resident in LDS, and every 64-k block is contracted on the six largest plane pairs of
 * (xh + xm + xl)(wh + wm + wl) with fp32 accumulation -- the five small pairs of every block first, then (h, h) over all of K,
 * as bnn_dense_forward_x3 does.  No im2col panel, no workspace.  Same shapes as bnn_conv2d_dense_forward.
 * replaces  F.conv2d(x, *self.sampled, ...)  pytorch_bayesian/nn/conv.py:116 */
int bnn_conv2d_dense_forward_x3(const float *x, int64_t x_sample_stride,
                                const void *w, int64_t w_plane_stride, int64_t w_sample_stride, int64_t ldw,
                                const float *b, int64_t b_sample_stride,
                                float *y, int64_t y_sample_stride,
                                const bnn_conv2d_shape_t *sh, int nsamples, int flags, void *stream);
/* Flipout conv2d in ONE launch: y[b] = conv(x[b], mean) + R[b] * conv(x[b] * S[b], stddev) with per-example
 * sign tensors S (B x C) and R (B x O) of +-1 (conv.py:154-161).  w = [O rows of the mean | O rows of the stddev], bf16
 * tap-major (bnn_draw_multi with kind = 1 / 2 and taps = KH * KW), ldw >= roundup(C KH KW, 64).  Both contractions share
 * the A fragment of the implicit GEMM above: S is XOR-ed into its sign bits for the second one, R scales that
 * accumulator in the epilogue.  No bias (the reference's Flipout conv has none).  Built for 2 O = 64 or 128.
 * replaces  FlipOutNormalConv2d.forward  pytorch_bayesian/nn/conv.py:207-221 */
int bnn_conv2d_flipout_forward(const float *x, const void *w, int64_t ldw, const float *sign_in, const float *sign_out,
                               float *y, const bnn_conv2d_shape_t *shape, int flags, void *stream);
/* The same launch in the fp32 PARITY mode: w = [O mean rows | O stddev rows] as BNN_BF16X3 planes (w_plane_stride elements apart,
 * >= 2 O ldw; bnn_draw_multi with kind = 1 / 2, taps, out_dtype = BNN_BF16X3 and out_sample_stride = the plane stride), the images
 * split into three planes in LDS, six plane pairs per 64-k block; S flips the sign bits of every plane of the A fragment alike.
 * 1e-5 of the output scale against float64.  replaces  FlipOutNormalConv2d.forward  pytorch_bayesian/nn/conv.py:207-221 */
int bnn_conv2d_flipout_forward_x3(const float *x, const void *w, int64_t w_plane_stride, int64_t ldw, const float *sign_in,
                                  const float *sign_out, float *y, const bnn_conv2d_shape_t *sh, int flags, void *stream);

/* ---- backward of K2 conv2d through the panel ------------------
 * replaces  autograd through F.conv2d (conv.py:116) for groups == 1, C*KH*KW % 8 == 0.  With
 * M = B*OH*OW rows, K = C*KH*KW, N = O:
 *   rows  = bnn_nchw_to_rows(gy)                 gy (S*B, O, OH*OW) -> (S*B*OH*OW, O), fp32 or bf16
 *   panel = bnn_conv2d_im2col(x)                 (x_samples*M, K), fp32 or bf16 (the forward's panel)
 *   bnn_linear_backward_weight_sampled(panel, rows, ...)   -> g_mu, g_rho of the (O, K) weight
 *   bnn_linear_backward_input_sampled(rows, ...) -> gpanel (S*M, K) fp32
 *   bnn_conv2d_col2im(gpanel)                    -> gx (S | 1, B, C, H, W), gather (no atomics);
 *                                                   shared_x != 0 also sums over the samples. */
int bnn_conv2d_im2col(const float *x, int64_t x_sample_stride, const bnn_conv2d_shape_t *shape,
                      int x_samples, void *panel, int out_bf16, void *stream);
int bnn_conv2d_col2im(const float *gpanel, const bnn_conv2d_shape_t *shape, int nsamples,
                      int shared_x, float *gx, void *stream);
int bnn_nchw_to_rows(const float *y, int64_t images, int channels, int pixels, void *rows,
                     int out_bf16, void *stream);

/* ---- diagnostics ------------------------------------------------------------
 * VALU cost of the draw, no memory traffic: `blocks` workgroups of 256 threads each run
 * `iters` Philox blocks (4 draws) of stage 0 (Philox4x32-10 only), 1 (+ Box-Muller),
 * 2 (+ softplus sigma), 3 (+ fma = full draw); out needs blocks * 256 floats. */
int bnn_diag_sampler(float *out, int blocks, int iters, int stage, void *stream);
/* L2 -> CU activation stream in the access shapes of the fused linear kernel: S * (M / rows_per_wg)
 * * ntn workgroups of nwaves waves each read their (rows_per_wg x K) block of x (S, M, K) once.
 * pattern 0: MFMA-fragment shaped loads; 1: row-contiguous loads; 2: row-contiguous LDS-DMA.
 * out needs grid * nwaves * 64 floats. */
int bnn_diag_astream(const float *x, int S, int M, int K, int rows_per_wg, int ntn, int pattern,
                     int nwaves, float *out, void *stream);

/* ---- training-loop callers (either side of the backward path) ----------------
 * replaces  torch.optim.Adam(model.parameters(), lr).step()   examples/MNIST/train.py:41,65
 *   g += wd p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 *   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps),   t = *step + 1; then *step += 1.
 * All tensors of the model in ONE launch (+ the 1-thread counter bump); `step` is a device float so
 * that a captured graph advances it. */
typedef struct bnn_adam_tensor {
    float *p;
    const float *g;
    float *m;
    float *v;
    int64_t n;
} bnn_adam_tensor_t;
int bnn_adam_step(const bnn_adam_tensor_t *tensors, int ntensors, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float *step, void *stream);
/* The same step, and advance_epoch[0] += advance_inc (may be NULL) in its last launch: the optimizer step is the end of
 * a training step (every draw of the step, forward and backward, has been consumed by then), so the fresh-noise bump of a
 * captured step needs no launch of its own (bnn_rng_advance). */
int bnn_adam_step_advance(const bnn_adam_tensor_t *tensors, int ntensors, float lr, float beta1, float beta2,
                          float eps, float weight_decay, float *step, uint32_t *advance_epoch,
                          uint32_t advance_inc, void *stream);
/* replaces  torch.nn.CrossEntropyLoss()(pred, y)   examples/MNIST/train.py:39,59-61  (reduction 'mean'):
 *   loss[0] = mean_r (logsumexp(x_r) - x_r[y_r]);  g_logits (may be NULL) = (softmax(x_r) - onehot(y_r)) / rows.
 * logits (rows, classes) fp32 row-major, target int64; workspace: bnn_xent_workspace_bytes(rows). */
int64_t bnn_xent_workspace_bytes(int64_t rows);
int bnn_softmax_xent(const float *logits, const int64_t *target, int64_t rows, int classes,
                     float *loss, float *g_logits, void *workspace, void *stream);

/* ---- pruning score --------------------------------------------
 * replaces  param.dist.log_prob(0)   pytorch_bayesian/prune/prune.py:11
 *   out[i] = log N(0; mu[i], sigma(rho[i])) = -mu^2 / (2 sigma^2) - ln sigma - ln sqrt(2 pi)
 * (the top-k selection and the masked assignment of prune.py:12-17 stay torch ops on the device). */
int bnn_prune_score(const float *mu, const float *rho, float *out, int64_t n, void *stream);

/* ---- MC reduction ----------------------------------------------------------
 * replaces  torch.stack(preds).mean(0)   examples/MNIST/uncertainty.py:50
 *   out[i] (+)= scale * sum_s y[s * y_sample_stride + i],  i < n   (nsamples <= 256; more than 32 addends per output -- the
 *   partial logits of bnn_dense_forward_head -- are summed by four waves per 64 outputs, fixed order).
 * advance_epoch (may be NULL): advance_epoch[0] += advance_inc in the same launch -- the
 * reduction is the tail of an MC step (every draw of the step has been consumed by the
 * kernels stream-ordered before it), so this saves the separate bnn_rng_advance launch. */
int bnn_mc_sum(const float *y, int64_t y_sample_stride, int nsamples, int64_t n,
               float scale, float *out, int accumulate, uint32_t *advance_epoch,
               uint32_t advance_inc, void *stream);

/* bnn_mc_sum + the second pass of a KL begun by bnn_kl_forward_partial(tensors, ntensors, workspace): kl_out as `out`
 * of bnn_kl_forward (ntensors + 1 floats). */
int bnn_mc_sum_kl(const float *y, int64_t y_sample_stride, int nsamples, int64_t n,
                  float scale, float *out, int accumulate, uint32_t *advance_epoch,
                  uint32_t advance_inc, const bnn_kl_tensor_t *tensors, int ntensors,
                  float n_batches, float *kl_out, const void *workspace, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BNN_HIP_H */
