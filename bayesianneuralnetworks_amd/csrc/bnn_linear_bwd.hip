// bnn_linear_bwd.hip -- backward of the sampled linear layer (the reference gets it
// from autograd through F.linear, dense.py:60, and mu + sigma * eps, core.py:45):
//
//   weight gradient, fused with the backward of the draw -- the per-sample dW_s is never stored:
//       g_mu [n][k]  = sum_s dW_s[n][k]                         dW_s = gy_s^T x_s   (sum over the batch)
//       g_rho[n][k]  = sum_s dW_s[n][k] * eps_s[n][k] * sigmoid(rho[n][k])
//     eps_s re-created from the draw's key (the forward's bits), in the accumulator registers.
//   input gradient with explicit weights (small / unaligned layers; the aligned ones run the fused
//   draw kernel of bnn_linear.hip in B_SAMPLED_T mode), bias column sums, ReLU mask.
//
// k_wgrad: output tile 128 (k) x 64 (n) per 256-thread workgroup, computed TRANSPOSED (rows = k):
// an MFMA accumulator lane then holds 4 consecutive k of one n -- exactly one Philox block -- so the
// eps epilogue costs one block per 4 outputs, as in the forward.  Both operands are reduction-major
// in memory ([m][k] and [m][n]):
//   bf16: the tiles are staged in LDS as they lie in memory ([32 m][cols], 16-B chunks XOR-swizzled)
//         and read with ds_read_b64_tr_b16, gfx950's transposing LDS read, which hands every lane the
//         reduction-contiguous fragment v_mfma_f32_16x16x32_bf16 wants;
//   fp32: v_mfma_f32_16x16x4_f32 takes one element per lane, so the [m][cols] image is read with
//         plain ds_read_b32 (rows padded by 16 floats: the four m of a step sit on disjoint banks).
// Small outputs (N * K small, e.g. the 10-wide head) split the MC samples over gridDim.y; the
// partials go to the registered workspace and a second kernel adds them in a fixed order.
#include <cstdlib>
#include <type_traits>

#include "bnn_device.hpp"
#include "bnn_gemm_params.hpp"
#include "bnn_dma.hpp"

namespace bnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct WgradParams {
    const void *x;              // (S | 1, M, K) fp32 or bf16
    int64_t x_sample_stride, ldx;
    const void *gy;             // (S, M, N) fp32 or bf16
    int64_t gy_sample_stride, ldgy;
    const float *rho;           // (N, K)
    float *g_mu, *g_rho;        // (N, K), or the workspace slabs when nsplit > 1
    int64_t slab_stride;        // floats between the partials of two sample groups (nsplit > 1)
    int32_t M, N, K, S, accumulate, ntk, ntn, nsplit, xcd_map;
    int32_t vecX, vecG;         // 16-B loads legal
    int32_t diag;               // BNN_WGRAD_DIAG (timing only, wrong results): 1 = no fragment reads / MFMAs, 2 = no LDS-DMA
    int32_t plain;              // 1: no draw -- out[s] = dW_s per sample (gridDim.y = S), F.linear's own gradient
    RngDev rng;
    // optional fused bias gradient (nsplit == 1 only): the workgroups of k-tile 0 also take the column sums
    // of gy (one extra MFMA against a fragment of ones per n-subtile and step) and apply the bias draw's backward
    const float *rho_b;
    float *g_mu_b, *g_rho_b;
    RngDev rng_b;
    // optional fused KL gradient (the closed form of bnn_kl_backward, added in the final store):
    //   g_mu += c (mu - mu_p) / sigma_p^2,  g_rho += c (sigma / sigma_p^2 - 1 / sigma) sigmoid(rho),  c = *kl_up * kl_scale
    const float *kl_up;         // device scalar (the upstream gradient of the KL term); NULL = no KL
    const float *mu_w, *mu_b;   // posterior means (the KL gradient needs them; the likelihood gradient does not)
    float kl_scale_w, kl_pm_w, kl_ps_w, kl_scale_b, kl_pm_b, kl_ps_b;
};

constexpr int W_TK = 128, W_TN = 64, W_BM = 32, W_NT = 256;

// Byte offset of 16-B chunk `ch` of row `row` in a [32][cols] bf16 image with ROWB-byte rows.  The
// XOR keeps the 16-B staging writes and the transposed reads conflict-free (cdna guide T10 (b)).
template <int ROWB>
__device__ __forceinline__ int img_off(int row, int ch)
{
    constexpr int NCH = ROWB / 16;
    return ROWB * row + 16 * ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) & (NCH - 1));
}

__device__ __forceinline__ s16x4 lds_tr_read(const char *p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
}

// 8 consecutive elements of one row as packed bf16 (zeros outside the matrix).
template <bool BF>
__device__ __forceinline__ uint4 load8_bf16(const void *base, int64_t row_off, int col, int ncols, bool row_ok, bool vec)
{
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (!row_ok || col >= ncols) return o;
    if constexpr (BF) {
        const uint16_t *q = reinterpret_cast<const uint16_t *>(base) + row_off + col;
        if (vec && col + 8 <= ncols) return *reinterpret_cast<const uint4 *>(q);
        uint16_t h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (col + j < ncols) ? q[j] : (uint16_t)0;
        o.x = h[0] | ((uint32_t)h[1] << 16); o.y = h[2] | ((uint32_t)h[3] << 16);
        o.z = h[4] | ((uint32_t)h[5] << 16); o.w = h[6] | ((uint32_t)h[7] << 16);
        return o;
    } else {
        const float *q = reinterpret_cast<const float *>(base) + row_off + col;
        float f[8];
        if (vec && col + 8 <= ncols) {
            const float4 a = *reinterpret_cast<const float4 *>(q), b = *reinterpret_cast<const float4 *>(q + 4);
            f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (col + j < ncols) ? q[j] : 0.f;
        }
        o.x = pack_bf16x2(f[0], f[1]); o.y = pack_bf16x2(f[2], f[3]);
        o.z = pack_bf16x2(f[4], f[5]); o.w = pack_bf16x2(f[6], f[7]);
        return o;
    }
}

__device__ __forceinline__ float4 load4_f32(const float *base, int64_t row_off, int col, int ncols, bool row_ok, bool vec)
{
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!row_ok || col >= ncols) return o;
    const float *q = base + row_off + col;
    if (vec && col + 4 <= ncols) return *reinterpret_cast<const float4 *>(q);
    o.x = q[0];
    if (col + 1 < ncols) o.y = q[1];
    if (col + 2 < ncols) o.z = q[2];
    if (col + 3 < ncols) o.w = q[3];
    return o;
}

// Tile decode, XCD-aware: workgroup L runs on XCD L % 8 (round-robin dispatch); the 8 XCDs form a 2 (k) x 4 (n)
// grid over the tile grid, so one XCD's L2 holds half of x's column panels and a quarter of gy's instead of
// all of both.  gridDim.x = 8 * sub_k * sub_n; workgroups that fall outside the tile grid exit at once.
__device__ __forceinline__ bool wgrad_tile(const WgradParams &p, int &kt, int &nt)
{
    if (!p.xcd_map) {                            // small tile grids: plain order, every workgroup has a tile
        kt = blockIdx.x % p.ntk;
        nt = blockIdx.x / p.ntk;
        return true;
    }
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int sub_k = (p.ntk + 1) / 2, sub_n = (p.ntn + 3) / 4;
    kt = (xcd & 1) * sub_k + idx % sub_k;
    nt = (xcd >> 1) * sub_n + idx / sub_k;
    return idx < sub_k * sub_n && kt < p.ntk && nt < p.ntn;
}

// Shared tail of both kernels.  Accumulator element acc[a][b][r] is output (n, k):
//   n = n0 + wn * 32 + b * 16 + (lane & 15),  k = k0 + wk * 64 + a * 16 + 4 * (lane >> 4) + r.
struct WgradAcc {
    f32x4 acc[4][2], gmu[4][2], grho[4][2];
};

// Closed-form d KL(N(mu, sigma(rho)^2) || N(pm, ps^2)) / d (mu, rho) times c, as in bnn_kl.hip:k_kl_backward;
// the rho part is returned BEFORE the sigmoid(rho) factor the callers apply to the whole sum.
__device__ __forceinline__ void kl_grad_terms(float mu, float rho, float c, float pm, float ps, float &dmu, float &drho_pre)
{
    const float inv_ps2 = 1.0f / (ps * ps);
    const float sg = sigma_accurate(rho);
    dmu = c * (mu - pm) * inv_ps2;
    drho_pre = c * (sg * inv_ps2 - 1.0f / sg);
}

// Fused bias gradient.  Each of the 4 waves of a k-tile-0 workgroup takes ONE 16-column subtile (wave (wk, wn):
// columns nb + 16 wk ..), so the extra MFMA per step is spread evenly.  cs = running column sums of gy over the
// current sample (every row of the MFMA result is the same sum; row 0 = lanes 0..15, element 0, is used).
struct WgradBias {
    f32x4 cs;
    float gmu, grho;
};

__device__ __forceinline__ void wgrad_bias_init(WgradBias &Bz)
{
    Bz.cs = f32x4{0.f, 0.f, 0.f, 0.f};
    Bz.gmu = Bz.grho = 0.f;
}

__device__ __forceinline__ void wgrad_bias_sample_end(WgradBias &Bz, const WgradParams &p, uint32_t edev_b, int s, int nbias)
{
    const int lane = threadIdx.x & 63;
    const int n = nbias + lane;
    const float cs = Bz.cs[0];
    if (lane < 16 && n < p.N) {
        Bz.gmu += cs;
        Bz.grho = fmaf(cs, eps1(p.rng_b, edev_b, (uint64_t)n, p.rng_b.sample0 + (uint32_t)s), Bz.grho);
    }
    Bz.cs = f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ void wgrad_bias_store(const WgradBias &Bz, const WgradParams &p, int nbias)
{
    const int lane = threadIdx.x & 63;
    const int n = nbias + lane;
    if (lane >= 16 || n >= p.N) return;
    float gm = Bz.gmu, gr = Bz.grho;
    const float rb = p.rho_b[n];
    if (p.kl_up && p.mu_b) {
        float dm, dr;
        kl_grad_terms(p.mu_b[n], rb, p.kl_up[0] * p.kl_scale_b, p.kl_pm_b, p.kl_ps_b, dm, dr);
        gm += dm;
        gr += dr;
    }
    gr *= dsoftplus(rb);
    if (p.accumulate) { gm += p.g_mu_b[n]; gr += p.g_rho_b[n]; }
    p.g_mu_b[n] = gm;
    p.g_rho_b[n] = gr;
}

// UNCOND: draw eps for every accumulator tile, in range or not (out-of-range outputs are discarded by
// wgrad_store) -- one straight-line block, so the independent Philox chains of the 8 tiles interleave.
template <bool UNCOND = false>
__device__ __forceinline__ void wgrad_sample_end(WgradAcc &A, const WgradParams &p, uint32_t edev, int s, int nb, int kb)
{
    const int lane = threadIdx.x & 63;
    const bool k4 = (p.K & 3) == 0;              // rows start on a Philox block: one block per 4 outputs
    if constexpr (UNCOND) {                      // (needs K % 4 == 0)
        if (!p.plain) {
            const uint32_t sample = p.rng.sample0 + (uint32_t)s;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int n = nb + b * 16 + (lane & 15), k = kb + a * 16 + 4 * (lane >> 4);
                    const float4 z = eps4(p.rng, edev, (uint32_t)(((int64_t)n * p.K + k) >> 2), sample);
                    A.grho[a][b][0] = fmaf(A.acc[a][b][0], z.x, A.grho[a][b][0]);
                    A.grho[a][b][1] = fmaf(A.acc[a][b][1], z.y, A.grho[a][b][1]);
                    A.grho[a][b][2] = fmaf(A.acc[a][b][2], z.z, A.grho[a][b][2]);
                    A.grho[a][b][3] = fmaf(A.acc[a][b][3], z.w, A.grho[a][b][3]);
                }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                A.gmu[a][b] += A.acc[a][b];
                A.acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = nb + b * 16 + (lane & 15), k = kb + a * 16 + 4 * (lane >> 4);
            if (n < p.N && k < p.K) {
                A.gmu[a][b] += A.acc[a][b];
                if (!p.plain) {
                    const int64_t e0 = (int64_t)n * p.K + k;
                    const uint32_t sample = p.rng.sample0 + (uint32_t)s;
                    float4 z;
                    if (k4) {
                        z = eps4(p.rng, edev, (uint32_t)(e0 >> 2), sample);
                    } else {                     // odd K: element by element (columns past K are discarded)
                        z.x = eps1(p.rng, edev, (uint64_t)e0, sample);
                        z.y = eps1(p.rng, edev, (uint64_t)e0 + 1, sample);
                        z.z = eps1(p.rng, edev, (uint64_t)e0 + 2, sample);
                        z.w = eps1(p.rng, edev, (uint64_t)e0 + 3, sample);
                    }
                    A.grho[a][b][0] = fmaf(A.acc[a][b][0], z.x, A.grho[a][b][0]);
                    A.grho[a][b][1] = fmaf(A.acc[a][b][1], z.y, A.grho[a][b][1]);
                    A.grho[a][b][2] = fmaf(A.acc[a][b][2], z.z, A.grho[a][b][2]);
                    A.grho[a][b][3] = fmaf(A.acc[a][b][3], z.w, A.grho[a][b][3]);
                }
            }
            A.acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
}

__device__ __forceinline__ void wgrad_store(const WgradAcc &A, const WgradParams &p, int nb, int kb)
{
    const int lane = threadIdx.x & 63;
    const bool k4 = (p.K & 3) == 0;
    const bool final_pass = p.nsplit == 1 && !p.plain;
    float *om = p.g_mu, *orho = p.g_rho;
    if (!final_pass) {                           // raw partials / per-sample dW
        om = p.g_mu + (int64_t)blockIdx.y * p.slab_stride;
        orho = om + (int64_t)p.N * p.K;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = nb + b * 16 + (lane & 15), k = kb + a * 16 + 4 * (lane >> 4);
            if (n >= p.N || k >= p.K) continue;
            const int64_t e0 = (int64_t)n * p.K + k;
            float gm[4] = {A.gmu[a][b][0], A.gmu[a][b][1], A.gmu[a][b][2], A.gmu[a][b][3]};
            float gr[4] = {A.grho[a][b][0], A.grho[a][b][1], A.grho[a][b][2], A.grho[a][b][3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k + r >= p.K) continue;
                if (final_pass) {
                    const float rw = p.rho[e0 + r];
                    if (p.kl_up) {
                        float dm, dr;
                        kl_grad_terms(p.mu_w[e0 + r], rw, p.kl_up[0] * p.kl_scale_w, p.kl_pm_w, p.kl_ps_w, dm, dr);
                        gm[r] += dm;
                        gr[r] += dr;
                    }
                    gr[r] *= dsoftplus(rw);
                    if (p.accumulate) { gm[r] += om[e0 + r]; gr[r] += orho[e0 + r]; }
                } else if (p.plain && p.accumulate) {
                    gm[r] += om[e0 + r];
                }
            }
            if (k4) {                            // 4 consecutive k of one row, 16-B aligned
                *reinterpret_cast<float4 *>(om + e0) = make_float4(gm[0], gm[1], gm[2], gm[3]);
                if (!p.plain) *reinterpret_cast<float4 *>(orho + e0) = make_float4(gr[0], gr[1], gr[2], gr[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (k + r >= p.K) continue;
                    om[e0 + r] = gm[r];
                    if (!p.plain) orho[e0 + r] = gr[r];
                }
            }
        }
}

// ------------------------------------------------------------------ bf16 operands, fp32 accumulate
template <bool XBF, bool GBF>
__global__ __launch_bounds__(W_NT) void k_wgrad_bf16(const WgradParams p)
{
    constexpr int XB = W_BM * W_TK * 2, GB = W_BM * W_TN * 2;     // 8 KB + 4 KB per buffer
    __shared__ __attribute__((aligned(16))) char lds[2 * (XB + GB)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1;                       // 2 x 2 waves, 64 k x 32 n each
    int kt, nt;
    if (!wgrad_tile(p, kt, nt)) return;
    const int k0 = kt * W_TK, n0 = nt * W_TN;
    const int s_lo = (int)((int64_t)blockIdx.y * p.S / p.nsplit), s_hi = (int)((int64_t)(blockIdx.y + 1) * p.S / p.nsplit);
    const int msteps = (p.M + W_BM - 1) / W_BM;
    const int total = (s_hi - s_lo) * msteps;
    const uint32_t edev = rng_epoch_dev(p.rng);

    // staging: X tile 32 x 16 chunks = 512 (2 per thread), G tile 32 x 8 chunks = 256 (1 per thread)
    const int xr0 = tid >> 4, xch = tid & 15;                      // rows xr0 and xr0 + 16
    const int gr0 = tid >> 3, gch = tid & 7;
    uint4 sx[2], sg;
    auto fetch = [&](int t) {
        const int s = s_lo + t / msteps, m0 = (t % msteps) * W_BM;
        const int64_t xs = (int64_t)s * p.x_sample_stride, gs = (int64_t)s * p.gy_sample_stride;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + xr0 + 16 * i;
            sx[i] = load8_bf16<XBF>(p.x, xs + (int64_t)m * p.ldx, k0 + 8 * xch, p.K, m < p.M, p.vecX != 0);
        }
        const int m = m0 + gr0;
        sg = load8_bf16<GBF>(p.gy, gs + (int64_t)m * p.ldgy, n0 + 8 * gch, p.N, m < p.M, p.vecG != 0);
    };
    auto stage = [&](int buf) {
        char *X = lds + buf * (XB + GB), *G = X + XB;
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4 *>(X + img_off<256>(xr0 + 16 * i, xch)) = sx[i];
        *reinterpret_cast<uint4 *>(G + img_off<128>(gr0, gch)) = sg;
    };

    WgradAcc A;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) A.acc[a][b] = A.gmu[a][b] = A.grho[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool bias_wave = p.g_mu_b != nullptr && kt == 0;         // fused bias gradient: subtile wk of this wave's two
    const uint32_t edev_b = bias_wave ? rng_epoch_dev(p.rng_b) : 0u;
    WgradBias Bz;
    wgrad_bias_init(Bz);
    const s16x8 ones8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0

    // transposed-read addressing: lane = 16 Q + 4 q + pp supplies row (8Q [+4] + q), columns 4 pp .. 4 pp + 3
    // of a 16-column block and receives column (lane & 15), rows 8Q [+4] .. +3
    const int Q = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    auto compute = [&](int buf) {
        const char *X = lds + buf * (XB + GB), *G = X + XB;
        s16x8 af[4], bfr[2];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int c0 = (wk * 64 + a * 16) / 8 + (pp >> 1);
            const s16x4 lo = lds_tr_read(X + img_off<256>(8 * Q + q, c0) + 8 * (pp & 1));
            const s16x4 hi = lds_tr_read(X + img_off<256>(8 * Q + 4 + q, c0) + 8 * (pp & 1));
            af[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c0 = (wn * 32 + b * 16) / 8 + (pp >> 1);
            const s16x4 lo = lds_tr_read(G + img_off<128>(8 * Q + q, c0) + 8 * (pp & 1));
            const s16x4 hi = lds_tr_read(G + img_off<128>(8 * Q + 4 + q, c0) + 8 * (pp & 1));
            bfr[b] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                A.acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[a]),
                                                                      __builtin_bit_cast(bf16x8, bfr[b]), A.acc[a][b], 0, 0, 0);
        if (bias_wave)
            Bz.cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones8),
                                                            __builtin_bit_cast(bf16x8, wk ? bfr[1] : bfr[0]), Bz.cs, 0, 0, 0);
    };

    const int nb = n0 + wn * 32, kb = k0 + wk * 64;
    if (total > 0) {
        fetch(0);
        stage(0);
    }
    __syncthreads();
    for (int t = 0; t < total; ++t) {
        if (t + 1 < total) fetch(t + 1);
        compute(t & 1);
        if (t + 1 < total) stage((t + 1) & 1);
        __syncthreads();
        if ((t + 1) % msteps == 0) {
            wgrad_sample_end(A, p, edev, s_lo + t / msteps, nb, kb);
            if (bias_wave) wgrad_bias_sample_end(Bz, p, edev_b, s_lo + t / msteps, nb + 16 * wk);
        }
    }
    wgrad_store(A, p, nb, kb);
    if (bias_wave) wgrad_bias_store(Bz, p, nb + 16 * wk);
}

// Same contraction for the hot case -- x and gy both bf16 in memory, M % 256 == 0, K % 8 == 0, N % 8 == 0:
// the [32 m][cols] images ARE the memory layout, so LDS-DMA (global_load_lds_dwordx4) fills them
// without touching registers, the XOR swizzle applied on the source side (lane l of a 1-KiB piece
// fetches the chunk that belongs in LDS slot l).  Ring of 8 images (96 KB) handled in pairs: one barrier per
// 64 rows, three pairs in flight per wave; wave w owns X pieces 2w, 2w+1 and G piece w of every step (rows 8w .. 8w+7).
// The loop is instruction-bound (one wave per SIMD), so it carries no address arithmetic: DMA sources
// are a scalar base (advanced with scalar adds) + a fixed 32-bit lane offset, LDS read addresses are
// fixed registers + the buffer's immediate offset (loop unrolled over the ring).
// Columns past K / N are fetched from the last legal chunk: they only feed outputs that are discarded.
__global__ __launch_bounds__(W_NT) void k_wgrad_bf16_dma(const WgradParams p)
{
    constexpr int XB = W_BM * W_TK * 2, GB = W_BM * W_TN * 2, NBUF = 8, OPS = 3, BUFB = XB + GB;
    __shared__ __attribute__((aligned(1024))) char lds[NBUF * BUFB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, wn = wave & 1;
    int kt, nt;
    if (!wgrad_tile(p, kt, nt)) return;
    const int k0 = kt * W_TK, n0 = nt * W_TN;
    const int s_lo = (int)((int64_t)blockIdx.y * p.S / p.nsplit), s_hi = (int)((int64_t)(blockIdx.y + 1) * p.S / p.nsplit);
    const int msteps = p.M / W_BM;
    const int total = (s_hi - s_lo) * msteps;
    const uint32_t edev = rng_epoch_dev(p.rng);

    // fixed lane offsets (bytes) from the scalar base of a step
    uint32_t xoff[2], goff;
    {
        const int sc = lane & 15;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 8 * wave + 4 * j + (lane >> 4);
            int col = k0 + 8 * (sc ^ (((row & 3) << 2) | ((row >> 2) & 3)));
            col = col < p.K - 8 ? col : p.K - 8;
            xoff[j] = (uint32_t)(((int64_t)row * p.ldx + col) * 2);
        }
        const int grow = 8 * wave + (lane >> 3), gsc = lane & 7;
        int gcol = n0 + 8 * (gsc ^ ((((grow & 3) << 2) | ((grow >> 2) & 3)) & 7));
        gcol = gcol < p.N - 8 ? gcol : p.N - 8;
        goff = (uint32_t)(((int64_t)grow * p.ldgy + gcol) * 2);
    }
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    // issue cursor: running scalar bases of the next 32-row image; a barrier covers TWO images (64 rows)
    const char *xb = reinterpret_cast<const char *>(p.x) + (int64_t)s_lo * p.x_sample_stride * 2;
    const char *gb = reinterpret_cast<const char *>(p.gy) + (int64_t)s_lo * p.gy_sample_stride * 2;
    const int64_t x_img = (int64_t)W_BM * p.ldx * 2, g_img = (int64_t)W_BM * p.ldgy * 2;   // bytes per 32 rows
    // correction when the cursor passes the last image of a sample: to the next sample's first row
    const int64_t x_wrap = (p.x_sample_stride - (int64_t)p.M * p.ldx) * 2, g_wrap = (p.gy_sample_stride - (int64_t)p.M * p.ldgy) * 2;
    int im = 0, left = total;                                      // images issued in this sample / images left to issue
    auto issue_pair = [&](int pair) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t base = lds0 + (uint32_t)(2 * pair + h) * BUFB;
            dma16_s3(xb, xoff[0], base + (uint32_t)wave * 2048u, xb, xoff[1], base + (uint32_t)wave * 2048u + 1024u,
                     gb, goff, base + XB + (uint32_t)wave * 1024u);
            if (left > 1) {                                        // past the end: harmless re-fetch of the last image
                --left;
                xb += x_img;
                gb += g_img;
                if (++im == msteps) { im = 0; xb += x_wrap; gb += g_wrap; }
            }
        }
    };

    WgradAcc A;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) A.acc[a][b] = A.gmu[a][b] = A.grho[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool bias_wave = p.g_mu_b != nullptr && kt == 0;         // fused bias gradient: subtile wk of this wave's two
    const uint32_t edev_b = bias_wave ? rng_epoch_dev(p.rng_b) : 0u;
    WgradBias Bz;
    wgrad_bias_init(Bz);
    const s16x8 ones8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0

    const int Q = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    // tr-read addresses in buffer 0 and in buffer 4 (the ds_read immediate offset reaches 64 KB; the ring is 96 KB)
    const char *aptr[2][4][2], *bptr[2][2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int c0 = (wk * 64 + a * 16) / 8 + (pp >> 1);
            aptr[h][a][0] = lds + h * 4 * BUFB + img_off<256>(8 * Q + q, c0) + 8 * (pp & 1);
            aptr[h][a][1] = lds + h * 4 * BUFB + img_off<256>(8 * Q + 4 + q, c0) + 8 * (pp & 1);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c0 = (wn * 32 + b * 16) / 8 + (pp >> 1);
            bptr[h][b][0] = lds + h * 4 * BUFB + XB + img_off<128>(8 * Q + q, c0) + 8 * (pp & 1);
            bptr[h][b][1] = lds + h * 4 * BUFB + XB + img_off<128>(8 * Q + 4 + q, c0) + 8 * (pp & 1);
        }
    }
    auto compute = [&](auto BUF) {
        constexpr int h = decltype(BUF)::value / 4;
        constexpr int o = (decltype(BUF)::value % 4) * BUFB;       // immediate offset of the ds_read
        s16x8 af[4], bfr[2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
            af[a] = __builtin_shufflevector(lds_tr_read(aptr[h][a][0] + o), lds_tr_read(aptr[h][a][1] + o), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int b = 0; b < 2; ++b)
            bfr[b] = __builtin_shufflevector(lds_tr_read(bptr[h][b][0] + o), lds_tr_read(bptr[h][b][1] + o), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                A.acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[a]),
                                                                      __builtin_bit_cast(bf16x8, bfr[b]), A.acc[a][b], 0, 0, 0);
        if (bias_wave)
            Bz.cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones8),
                                                            __builtin_bit_cast(bf16x8, wk ? bfr[1] : bfr[0]), Bz.cs, 0, 0, 0);
    };

    const int nb = n0 + wn * 32, kb = k0 + wk * 64;
    if (total > 0) {
        constexpr int NPAIR = NBUF / 2;
#pragma unroll
        for (int t = 0; t < NPAIR - 1; ++t) issue_pair(t);
        auto step = [&](auto PAIR) {
            constexpr int pr = decltype(PAIR)::value;
            // this wave's pieces of the pair have landed (the next two pairs may still be in flight) ...
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NPAIR - 2) * 2 * OPS) : "memory");
            __syncthreads();                                       // ... and everyone else's; the previous pair's buffers are free
            issue_pair((pr + NPAIR - 1) % NPAIR);
            compute(std::integral_constant<int, 2 * pr>{});
            compute(std::integral_constant<int, 2 * pr + 1>{});
        };
        for (int s = s_lo; s < s_hi; ++s) {                        // msteps % NBUF == 0: every sample starts in buffer 0
            for (int mi = 0; mi < msteps; mi += NBUF) {
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
                step(std::integral_constant<int, 2>{});
                step(std::integral_constant<int, 3>{});
            }
            wgrad_sample_end<true>(A, p, edev, s, nb, kb);
            if (bias_wave) wgrad_bias_sample_end(Bz, p, edev_b, s, nb + 16 * wk);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // nothing may still be writing this LDS at exit
    }
    wgrad_store(A, p, nb, kb);
    if (bias_wave) wgrad_bias_store(Bz, p, nb + 16 * wk);
}

// Two-role version of the kernel above (512 threads): waves 0..3 run the MFMA loop and nothing else; waves
// 4..7 -- one per SIMD beside an MFMA wave -- issue ALL the LDS-DMA (with its scalar bookkeeping and the
// vmcnt waits) and draw the eps of the CURRENT sample, one Philox block between two barriers, into a
// 32-KB LDS image that the MFMA waves read once at the end of the sample.  Measured: 54.5 -> 48.6 us at the
// layer-2 shape -- the MFMA loop itself drops to 29 us (no issue overhead), but a VALU-heavy wave and an
// MFMA-heavy wave on one SIMD add up rather than overlap, so the 19 us of eps draws are not hidden (spreading
// them evenly over all pair steps, 8 interleaved chains, measured WORSE: 65 us).  Both roles execute the same barriers: one per pair of
// images and one (E) per sample, after the helper's last eps write; the helper's first write of the next
// sample comes after the next pair barrier, which the MFMA waves pass only after their eps reads.
// Round 3, what bounds the loop (BNN_WGRAD_DIAG, BNN_WGRAD_NOEPS; layer-2 shape, us): full 49.6 (24-bit stream; 48.1 on the 16-bit
// stream with whole blocks), no eps 30.2, no eps + no fragment reads / MFMAs 29.3, no eps + no DMA 24.9, neither 12.5: the
// 128 x 64 tile streams 288 MB through LDS per launch (x re-read 19 x, gy 10 x) and THAT is the loop; the MFMAs are 1 us of it.
// The eps draw adds its full time wherever it runs: moved into the MFMA waves (which "wait" most of a step) -- 4 rows at 2
// points per sample 52.8, one row at each of 4 points 55.1 (24-bit) / 49.7 (16-bit) -- it was no better than in the helpers.
// What would help is fewer bytes per FLOP (a 128 x 128 tile: 192 MB), not a better place for the draw.
__global__ __launch_bounds__(2 * W_NT) void k_wgrad_bf16_dma8(const WgradParams p)
{
    constexpr int XB = W_BM * W_TK * 2, GB = W_BM * W_TN * 2, NBUF = 8, OPS = 3, BUFB = XB + GB, NPAIR = NBUF / 2;
    constexpr int EPSB = 4 * 8 * 64 * 16;                          // [wave][tile][lane] float4
    __shared__ __attribute__((aligned(1024))) char lds[NBUF * BUFB + EPSB];
    char *eps_lds = lds + NBUF * BUFB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool helper = wave8 >= 4;
    const int wave = wave8 & 3;                                    // the MFMA wave this wave is, or serves
    const int wk = wave >> 1, wn = wave & 1;
    int kt, nt;
    if (!wgrad_tile(p, kt, nt)) return;
    const int k0 = kt * W_TK, n0 = nt * W_TN;
    const int s_lo = (int)((int64_t)blockIdx.y * p.S / p.nsplit), s_hi = (int)((int64_t)(blockIdx.y + 1) * p.S / p.nsplit);
    const int msteps = p.M / W_BM;                                 // images per sample, a multiple of NBUF
    const int total = (s_hi - s_lo) * msteps;
    const int nb = n0 + wn * 32, kb = k0 + wk * 64;
    if (total <= 0) {                                              // (cannot happen: nsplit <= S)
        if (!helper) { WgradAcc Z; for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) Z.acc[a][b] = Z.gmu[a][b] = Z.grho[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; wgrad_store(Z, p, nb, kb); }
        return;
    }

    if (helper) {
        // ---------------------------------------------------------------- DMA + eps role
        const uint32_t edev = rng_epoch_dev(p.rng);
        const PhiloxKeys keys = philox_keys(p.rng.key0, p.rng.key1);   // round keys in VGPRs (this role has registers to spare)
        uint32_t xoff[2], goff;
        {
            const int sc = lane & 15;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 8 * wave + 4 * j + (lane >> 4);
                int col = k0 + 8 * (sc ^ (((row & 3) << 2) | ((row >> 2) & 3)));
                col = col < p.K - 8 ? col : p.K - 8;
                xoff[j] = (uint32_t)(((int64_t)row * p.ldx + col) * 2);
            }
            const int grow = 8 * wave + (lane >> 3), gsc = lane & 7;
            int gcol = n0 + 8 * (gsc ^ ((((grow & 3) << 2) | ((grow >> 2) & 3)) & 7));
            gcol = gcol < p.N - 8 ? gcol : p.N - 8;
            goff = (uint32_t)(((int64_t)grow * p.ldgy + gcol) * 2);
        }
        const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
        const char *xb = reinterpret_cast<const char *>(p.x) + (int64_t)s_lo * p.x_sample_stride * 2;
        const char *gb = reinterpret_cast<const char *>(p.gy) + (int64_t)s_lo * p.gy_sample_stride * 2;
        const int64_t x_img = (int64_t)W_BM * p.ldx * 2, g_img = (int64_t)W_BM * p.ldgy * 2;
        const int64_t x_wrap = (p.x_sample_stride - (int64_t)p.M * p.ldx) * 2, g_wrap = (p.gy_sample_stride - (int64_t)p.M * p.ldgy) * 2;
        int im = 0, left = total;
        auto issue_pair = [&](int pair) {
            if (p.diag & 2) return;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t base = lds0 + (uint32_t)(2 * pair + h) * BUFB;
                dma16_s3(xb, xoff[0], base + (uint32_t)wave * 2048u, xb, xoff[1], base + (uint32_t)wave * 2048u + 1024u,
                         gb, goff, base + XB + (uint32_t)wave * 1024u);
                if (left > 1) {                                    // past the end: harmless re-fetch of the last image
                    --left;
                    xb += x_img;
                    gb += g_img;
                    if (++im == msteps) { im = 0; xb += x_wrap; gb += g_wrap; }
                }
            }
        };
        const int pairs = msteps / 2;                              // pair barriers per sample (a multiple of 4)
        const int gen_every = pairs / 4;                           // 4 draw points per sample, TWO Philox blocks each: two
                                                                   // independent chains interleave (a lone chain is latency-bound)
        float4 *my_eps = reinterpret_cast<float4 *>(eps_lds) + (wave * 8) * 64 + lane;
        const bool pair8 = p.rng.gen == BNN_GEN_PHILOX7_U16 && (p.K & 7) == 0;     // (wave-uniform)
#pragma unroll
        for (int t = 0; t < NPAIR - 1; ++t) issue_pair(t);
        for (int s = s_lo; s < s_hi; ++s) {
            const uint32_t sample = p.rng.sample0 + (uint32_t)s;
            int tile = 0;
            for (int pr = 0; pr < pairs; ++pr) {
                // this wave's pieces of the pair have landed (the next two pairs may still be in flight) ...
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NPAIR - 2) * 2 * OPS) : "memory");
                __syncthreads();                                   // ... and everyone else's; the previous pair's buffers are free
                issue_pair((pr + NPAIR - 1) % NPAIR);
                if (!p.plain && pair8) {
                    // 16-bit stream: the quads of lane groups Q and Q ^ 1 (k .. k + 3 and k + 4 .. k + 7 of one n) are the two halves
                    // of ONE 8-eps Philox block.  Even Q draws the block of (n, k .. k + 7), odd Q the block of (n + 16, k - 4 ..
                    // k + 3), and each writes its own quad and its neighbour's: one 7-round block per lane and tile pair instead of
                    // two half-used ones.  Two draw points per sample, two independent chains each.
                    if (pr % (2 * gen_every) == 0) {
                        const int Q = lane >> 4, odd = Q & 1;
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const int a = (tile >> 1) + c;
                            const int n = nb + (lane & 15) + 16 * odd, k = kb + a * 16 + 4 * (Q - odd);
                            float4 za, zb;
                            eps8_u16(p.rng, keys, edev, (uint32_t)(((int64_t)n * p.K + k) >> 3), sample, za, zb);
                            float4 *e = my_eps + (tile + 2 * c + odd) * 64;      // tile (a, odd), this lane's slot
                            e[odd ? -16 : 0] = za;                               // the quad of the even lane group
                            e[odd ? 0 : 16] = zb;                                // the quad of the odd lane group
                        }
                        tile += 4;
                    }
                } else if (!p.plain && pr % gen_every == 0) {
                    // tiles (a, 0) and (a, 1): same k, n and n + 16
                    const int a = tile >> 1;
                    const int n = nb + (lane & 15), k = kb + a * 16 + 4 * (lane >> 4);
                    const float4 z0 = eps4(p.rng, keys, edev, (uint32_t)(((int64_t)n * p.K + k) >> 2), sample);
                    const float4 z1 = eps4(p.rng, keys, edev, (uint32_t)(((int64_t)(n + 16) * p.K + k) >> 2), sample);
                    my_eps[tile * 64] = z0;
                    my_eps[(tile + 1) * 64] = z1;
                    tile += 2;
                }
            }
            __syncthreads();                                       // E: this sample's eps image is complete
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // nothing may still be writing this LDS at exit
        return;
    }

    // -------------------------------------------------------------------- MFMA role
    WgradAcc A;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) A.acc[a][b] = A.gmu[a][b] = A.grho[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool bias_wave = p.g_mu_b != nullptr && kt == 0;         // fused bias gradient: subtile wk of this wave's two
    const uint32_t edev_b = bias_wave ? rng_epoch_dev(p.rng_b) : 0u;
    WgradBias Bz;
    wgrad_bias_init(Bz);
    const s16x8 ones8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0

    const int Q = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const char *aptr[2][4][2], *bptr[2][2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int c0 = (wk * 64 + a * 16) / 8 + (pp >> 1);
            aptr[h][a][0] = lds + h * 4 * BUFB + img_off<256>(8 * Q + q, c0) + 8 * (pp & 1);
            aptr[h][a][1] = lds + h * 4 * BUFB + img_off<256>(8 * Q + 4 + q, c0) + 8 * (pp & 1);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c0 = (wn * 32 + b * 16) / 8 + (pp >> 1);
            bptr[h][b][0] = lds + h * 4 * BUFB + XB + img_off<128>(8 * Q + q, c0) + 8 * (pp & 1);
            bptr[h][b][1] = lds + h * 4 * BUFB + XB + img_off<128>(8 * Q + 4 + q, c0) + 8 * (pp & 1);
        }
    }
    auto compute = [&](auto BUF) {
        constexpr int h = decltype(BUF)::value / 4;
        constexpr int o = (decltype(BUF)::value % 4) * BUFB;       // immediate offset of the ds_read
        s16x8 af[4], bfr[2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
            af[a] = __builtin_shufflevector(lds_tr_read(aptr[h][a][0] + o), lds_tr_read(aptr[h][a][1] + o), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int b = 0; b < 2; ++b)
            bfr[b] = __builtin_shufflevector(lds_tr_read(bptr[h][b][0] + o), lds_tr_read(bptr[h][b][1] + o), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                A.acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[a]),
                                                                      __builtin_bit_cast(bf16x8, bfr[b]), A.acc[a][b], 0, 0, 0);
        if (bias_wave)
            Bz.cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones8),
                                                            __builtin_bit_cast(bf16x8, wk ? bfr[1] : bfr[0]), Bz.cs, 0, 0, 0);
    };
    auto step = [&](auto PAIR) {
        constexpr int pr = decltype(PAIR)::value;
        __syncthreads();                                           // the pair's images have landed (helper waves waited for them)
        if (p.diag & 1) return;
        compute(std::integral_constant<int, 2 * pr>{});
        compute(std::integral_constant<int, 2 * pr + 1>{});
    };
    const float4 *my_eps = reinterpret_cast<const float4 *>(eps_lds) + (wave * 8) * 64 + lane;
    for (int s = s_lo; s < s_hi; ++s) {                            // msteps % NBUF == 0: every sample starts in buffer 0
        for (int mi = 0; mi < msteps; mi += NBUF) {
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
        }
        __syncthreads();                                           // E: the eps image of this sample is complete
        if (!p.plain) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float4 z = my_eps[(a * 2 + b) * 64];
                    A.grho[a][b][0] = fmaf(A.acc[a][b][0], z.x, A.grho[a][b][0]);
                    A.grho[a][b][1] = fmaf(A.acc[a][b][1], z.y, A.grho[a][b][1]);
                    A.grho[a][b][2] = fmaf(A.acc[a][b][2], z.z, A.grho[a][b][2]);
                    A.grho[a][b][3] = fmaf(A.acc[a][b][3], z.w, A.grho[a][b][3]);
                }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                A.gmu[a][b] += A.acc[a][b];
                A.acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        if (bias_wave) wgrad_bias_sample_end(Bz, p, edev_b, s, nb + 16 * wk);
    }
    wgrad_store(A, p, nb, kb);
    if (bias_wave) wgrad_bias_store(Bz, p, nb + 16 * wk);
}

// ------------------------------------------------------------------ exact fp32
__global__ __launch_bounds__(W_NT) void k_wgrad_f32(const WgradParams p)
{
    constexpr int XLD = W_TK + 16, GLD = W_TN + 16;                // padded rows (floats)
    constexpr int XF = W_BM * XLD, GF = W_BM * GLD;
    __shared__ __attribute__((aligned(16))) float lds[2 * (XF + GF)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1;
    int kt, nt;
    if (!wgrad_tile(p, kt, nt)) return;
    const int k0 = kt * W_TK, n0 = nt * W_TN;
    const int s_lo = (int)((int64_t)blockIdx.y * p.S / p.nsplit), s_hi = (int)((int64_t)(blockIdx.y + 1) * p.S / p.nsplit);
    const int msteps = (p.M + W_BM - 1) / W_BM;
    const int total = (s_hi - s_lo) * msteps;
    const uint32_t edev = rng_epoch_dev(p.rng);
    const float *xp = reinterpret_cast<const float *>(p.x), *gp = reinterpret_cast<const float *>(p.gy);

    // staging: X tile 32 x 32 float4 = 1024 (4 per thread), G tile 32 x 16 float4 = 512 (2 per thread)
    const int xr0 = tid >> 5, xc4 = tid & 31;                      // rows xr0 + 8 i
    const int gr0 = tid >> 4, gc4 = tid & 15;                      // rows gr0 + 16 i
    float4 sx[4], sg[2];
    auto fetch = [&](int t) {
        const int s = s_lo + t / msteps, m0 = (t % msteps) * W_BM;
        const int64_t xs = (int64_t)s * p.x_sample_stride, gs = (int64_t)s * p.gy_sample_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + xr0 + 8 * i;
            sx[i] = load4_f32(xp, xs + (int64_t)m * p.ldx, k0 + 4 * xc4, p.K, m < p.M, p.vecX != 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + gr0 + 16 * i;
            sg[i] = load4_f32(gp, gs + (int64_t)m * p.ldgy, n0 + 4 * gc4, p.N, m < p.M, p.vecG != 0);
        }
    };
    auto stage = [&](int buf) {
        float *X = lds + buf * (XF + GF), *G = X + XF;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(X + (xr0 + 8 * i) * XLD + 4 * xc4) = sx[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<float4 *>(G + (gr0 + 16 * i) * GLD + 4 * gc4) = sg[i];
    };

    WgradAcc A;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) A.acc[a][b] = A.gmu[a][b] = A.grho[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool bias_wave = p.g_mu_b != nullptr && kt == 0;         // fused bias gradient: subtile wk of this wave's two
    const uint32_t edev_b = bias_wave ? rng_epoch_dev(p.rng_b) : 0u;
    WgradBias Bz;
    wgrad_bias_init(Bz);

    const int Q = lane >> 4, g = lane & 15;
    auto compute = [&](int buf) {
        const float *X = lds + buf * (XF + GF), *G = X + XF;
#pragma unroll
        for (int mm = 0; mm < W_BM / 4; ++mm) {
            float av[4], bv[2];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = X[(4 * mm + Q) * XLD + wk * 64 + a * 16 + g];
#pragma unroll
            for (int b = 0; b < 2; ++b) bv[b] = G[(4 * mm + Q) * GLD + wn * 32 + b * 16 + g];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    A.acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], A.acc[a][b], 0, 0, 0);
            if (bias_wave) Bz.cs = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, wk ? bv[1] : bv[0], Bz.cs, 0, 0, 0);
        }
    };

    const int nb = n0 + wn * 32, kb = k0 + wk * 64;
    if (total > 0) {
        fetch(0);
        stage(0);
    }
    __syncthreads();
    for (int t = 0; t < total; ++t) {
        if (t + 1 < total) fetch(t + 1);
        compute(t & 1);
        if (t + 1 < total) stage((t + 1) & 1);
        __syncthreads();
        if ((t + 1) % msteps == 0) {
            wgrad_sample_end(A, p, edev, s_lo + t / msteps, nb, kb);
            if (bias_wave) wgrad_bias_sample_end(Bz, p, edev_b, s_lo + t / msteps, nb + 16 * wk);
        }
    }
    wgrad_store(A, p, nb, kb);
    if (bias_wave) wgrad_bias_store(Bz, p, nb + 16 * wk);
}

// Fixed-order sum of the sample-group partials (nsplit > 1), then sigmoid(rho) on the rho part.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ slabs, int64_t slab_stride, int nsplit,
                                                      const float *__restrict__ rho, float *__restrict__ g_mu,
                                                      float *__restrict__ g_rho, int64_t n, int accumulate,
                                                      const float *__restrict__ kl_up, const float *__restrict__ mu,
                                                      float kl_scale, float kl_pm, float kl_ps)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float gm = 0.f, gr = 0.f;
    for (int g = 0; g < nsplit; ++g) {
        gm += slabs[(int64_t)g * slab_stride + e];
        gr += slabs[(int64_t)g * slab_stride + n + e];
    }
    if (kl_up) {
        float dm, dr;
        kl_grad_terms(mu[e], rho[e], kl_up[0] * kl_scale, kl_pm, kl_ps, dm, dr);
        gm += dm;
        gr += dr;
    }
    gr *= dsoftplus(rho[e]);
    if (accumulate) { gm += g_mu[e]; gr += g_rho[e]; }
    g_mu[e] = gm;
    g_rho[e] = gr;
}

// ------------------------------------------------------------------ backward of a narrow layer (N <= 16)
// The classifier head (N = 10): 0.4 % of the FLOPs, but as tiles for the kernels above it cost 65 us of launches
// (weight gradient + reduce, K1 draw + plain input gradient, column sums).  Here ONE pass over the activations
// does all of it: a workgroup owns 64 columns k and 128 rows of one MC sample, a thread 4 consecutive k and
// 1/16 of the rows; the workgroup's N x 64 weights are drawn once (the forward's bits), then per row it reads x
// once, accumulates dW[n][k..k+3] += gy[m][n] x[m][k..k+3] and writes gx[m][k..k+3] = sum_n gy[m][n] W_s[n][k..].
// gy of the sample (M x N fp32, 20 KB at the BASELINE shape) is staged in LDS once.  Per-sample partials
// (dW_s, dW_s * eps_s) go to the workspace slabs that k_wgrad_reduce sums in a fixed order (and where the
// sigmoid(rho) factor and the fused KL gradient are applied); the column sums of gy ride along for the bias.
struct HeadBwdParams {
    const void *x; int64_t x_sample_stride, ldx;
    const float *gy; int64_t gy_sample_stride, ldgy;
    const float *mu, *rho;
    void *gx; int64_t gx_sample_stride, ldgx;      // NULL: no input gradient
    float *slabs; int64_t slab_stride;             // [s][2][N][K]
    float *colsums;                                // [s][N] or NULL
    int32_t M, N, K, S;
    RngDev rng;
};

// rows per workgroup (gridDim.z row slices per sample).  256: the BASELINE head is 19 x 8 x 2 = 304 workgroups, all resident at
// once (66 KiB of LDS each: two per CU) -- with 128 rows the 608 workgroups ran as two rounds (23.7 us; now see DESIGN)
constexpr int H_NMAX = 16, H_ROWS = 256;

// NP: the output columns the register loops cover (N <= 12 -> 12: a 10-way head skips a quarter of the FMAs of the 16-wide code)
template <bool XBF, bool GXBF, int NP>
__global__ __launch_bounds__(256) void k_head_bwd(const HeadBwdParams p)
{
    // gy rows of this workgroup's slice ([m][n], pitch N) during the row loop; the 16 x 16 partial tiles after it
    __shared__ float smem[16 * 16 * (H_NMAX * 4 + 1)];
    static_assert(H_ROWS * H_NMAX <= 16 * 16 * (H_NMAX * 4 + 1), "gy slice must fit the partial buffer");
    float *gys = smem;
    float (*part)[16][H_NMAX * 4 + 1] = reinterpret_cast<float (*)[16][H_NMAX * 4 + 1]>(smem);
    const int tid = threadIdx.x, kg = tid & 15, mp = tid >> 4;
    const int s = blockIdx.y, N = p.N;
    const int k = blockIdx.x * 64 + 4 * kg;
    const bool kok = k < p.K;                                      // K % 4 == 0: a group is in or out as a whole
    const uint32_t edev = rng_epoch_dev(p.rng);
    const uint32_t sample = p.rng.sample0 + (uint32_t)s;
    const int m_lo = blockIdx.z * H_ROWS;
    const int rows = p.M - m_lo < H_ROWS ? p.M - m_lo : H_ROWS;    // >= 1 by the grid

    // stage this slice's gy rows (coalesced), and draw the thread's 4 x N weights meanwhile
    const float *gyb = p.gy + (int64_t)s * p.gy_sample_stride + (int64_t)m_lo * p.ldgy;
    for (int i = tid; i < rows * H_NMAX; i += 256) {                // pitch 16, columns >= N zero: the row loop needs no bounds
        const int r = i >> 4, n = i & 15;
        gys[i] = n < N ? gyb[(int64_t)r * p.ldgy + n] : 0.f;
    }
    // the workgroup's N x 64 weights are drawn ONCE (thread t < 16 N: row t / 16, column group t % 16 -- one
    // Philox block each) and shared through LDS; every thread then keeps its column group's N x 4 in registers
    __shared__ float4 wl[H_NMAX][16], zl[H_NMAX][16];
    {                                                              // 256 threads = 16 rows x 16 column groups; rows >= N stay zero
        const int n = tid >> 4, kk = blockIdx.x * 64 + 4 * (tid & 15);
        float4 zz = make_float4(0.f, 0.f, 0.f, 0.f), ww = zz;
        if (n < N && kk < p.K) {
            const int64_t e0 = (int64_t)n * p.K + kk;
            const float4 m4 = *reinterpret_cast<const float4 *>(p.mu + e0), r4 = *reinterpret_cast<const float4 *>(p.rho + e0);
            zz = eps4(p.rng, edev, (uint32_t)(e0 >> 2), sample);
            ww = make_float4(fmaf(sigma_draw(r4.x), zz.x, m4.x), fmaf(sigma_draw(r4.y), zz.y, m4.y),
                             fmaf(sigma_draw(r4.z), zz.z, m4.z), fmaf(sigma_draw(r4.w), zz.w, m4.w));
        }
        wl[n][tid & 15] = ww;
        zl[n][tid & 15] = zz;
    }
    float cs[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) cs[n] = 0.f;
    __syncthreads();
    float4 w[NP], dW[NP];                                  // (eps stays in LDS until the epilogue: 2 x 64 VGPRs, not 3 x)
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        dW[n] = make_float4(0.f, 0.f, 0.f, 0.f);
        w[n] = wl[n][kg];
    }

    auto load_x = [&](int r) -> float4 {
        const int64_t xo = (int64_t)s * p.x_sample_stride + (int64_t)(m_lo + r) * p.ldx + k;
        if constexpr (XBF) {
            const uint2 h = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(p.x) + xo);
            return make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xFFFF0000u),
                               __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xFFFF0000u));
        } else {
            return *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p.x) + xo);
        }
    };
    // rows mp, mp + 16, ...: four at a time, their x loads in flight together (the loop is latency-bound otherwise)
    for (int r0 = mp; r0 < rows; r0 += 64) {
        float4 xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + 16 * u;
            xv[u] = (kok && r < rows) ? load_x(r) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + 16 * u;
            if (r < rows) {
            float g[NP];
#pragma unroll
            for (int q4 = 0; q4 < NP / 4; ++q4) {
                const float4 t = *reinterpret_cast<const float4 *>(gys + r * H_NMAX + 4 * q4);
                g[4 * q4] = t.x; g[4 * q4 + 1] = t.y; g[4 * q4 + 2] = t.z; g[4 * q4 + 3] = t.w;
            }
            if (kg == 0 && blockIdx.x == 0) {
#pragma unroll
                for (int n = 0; n < NP; ++n) cs[n] += g[n];
            }
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < NP; ++n) {                         // columns >= N carry g = 0, w = 0: straight-line code
                dW[n].x = fmaf(g[n], xv[u].x, dW[n].x); dW[n].y = fmaf(g[n], xv[u].y, dW[n].y);
                dW[n].z = fmaf(g[n], xv[u].z, dW[n].z); dW[n].w = fmaf(g[n], xv[u].w, dW[n].w);
                o.x = fmaf(g[n], w[n].x, o.x); o.y = fmaf(g[n], w[n].y, o.y);
                o.z = fmaf(g[n], w[n].z, o.z); o.w = fmaf(g[n], w[n].w, o.w);
            }
            if (kok && p.gx) {
                const int64_t go = (int64_t)s * p.gx_sample_stride + (int64_t)(m_lo + r) * p.ldgx + k;
                if constexpr (GXBF) {
                    uint2 h;
                    h.x = pack_bf16x2(o.x, o.y);
                    h.y = pack_bf16x2(o.z, o.w);
                    *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(p.gx) + go) = h;
                } else {
                    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.gx) + go) = o;
                }
            }
            }
        }
    }
    __syncthreads();                                               // gys is dead: its storage becomes the partial tiles
    // fixed-order reduction of the 16 row slices, then (dW, dW * eps) of this (sample, row slice) to its slab
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        part[mp][kg][n * 4 + 0] = dW[n].x; part[mp][kg][n * 4 + 1] = dW[n].y;
        part[mp][kg][n * 4 + 2] = dW[n].z; part[mp][kg][n * 4 + 3] = dW[n].w;
    }
    __syncthreads();
    const int slab = s * gridDim.z + blockIdx.z;
    {
        // all 256 threads: output o = (n, c) with c = 4 kg + j the column inside the 64-wide slice (fastest: the
        // stores of a row n are 256 contiguous bytes); 16 independent LDS reads each, added in a fixed order
        float *sm = p.slabs + (int64_t)slab * p.slab_stride, *sr = sm + (int64_t)N * p.K;
        for (int o = tid; o < N * 64; o += 256) {
            const int n = o >> 6, c = o & 63;
            const int kk = blockIdx.x * 64 + c;
            if (kk >= p.K) continue;
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += part[q][c >> 2][n * 4 + (c & 3)];
            const float4 zz = zl[n][c >> 2];
            const float ze = (c & 3) == 0 ? zz.x : (c & 3) == 1 ? zz.y : (c & 3) == 2 ? zz.z : zz.w;
            sm[(int64_t)n * p.K + kk] = t;
            sr[(int64_t)n * p.K + kk] = t * ze;
        }
    }
    if (p.colsums && blockIdx.x == 0) {                            // column sums of gy (bias): the kg == 0 threads hold them
        __syncthreads();
        if (kg == 0)
#pragma unroll
            for (int n = 0; n < NP; ++n) part[mp][0][n] = cs[n];
        __syncthreads();
        if (tid < N) {
            float t = 0.f;
            for (int q = 0; q < 16; ++q) t += part[q][0][tid];
            p.colsums[(int64_t)slab * N + tid] = t;                // per (sample, row slice); summed over slices below
        }
    }
}

// colsums[(s, z)][n] -> out[s][n]: fixed-order sum over the row slices
__global__ void k_head_colsum_fold(const float *__restrict__ cs, float *__restrict__ out, int S, int Z, int N)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S * N) return;
    const int s = i / N, n = i % N;
    float t = 0.f;
    for (int zz = 0; zz < Z; ++zz) t += cs[((int64_t)s * Z + zz) * N + n];
    out[i] = t;
}

// The tail of the narrow layer's backward in ONE launch (it was four: k_wgrad_reduce, k_head_colsum_fold, k_sample_affine_bwd,
// k_kl_backward -- 26 us of 4-10-us launches for 12 k weights): blocks 0 .. nbw - 1 sum the (sample, row slice) slabs of the
// weight gradient in a fixed order (k_wgrad_reduce's arithmetic), the LAST block folds the column sums of gy over the row slices
// and runs the bias draw's backward (k_sample_affine_bwd's arithmetic, same eps) and the bias' KL gradient (k_kl_backward's).
__global__ __launch_bounds__(256) void k_head_tail(const float *__restrict__ slabs, int64_t slab_stride, int nslab,
                                                   const float *__restrict__ rho, float *__restrict__ g_mu, float *__restrict__ g_rho,
                                                   int64_t n, int accumulate, const float *__restrict__ kl_up,
                                                   const float *__restrict__ mu, float kl_scale, float kl_pm, float kl_ps,
                                                   const float *__restrict__ cs_parts, int S, int Z, int N,
                                                   const float *__restrict__ rho_b, RngDev rng_b, float *__restrict__ g_mu_b,
                                                   float *__restrict__ g_rho_b, const float *__restrict__ mu_b, float kl_scale_b,
                                                   float kl_pm_b, float kl_ps_b)
{
    const int nbw = (int)((n + 255) / 256);
    if ((int)blockIdx.x < nbw) {
        const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (e >= n) return;
        float gm = 0.f, gr = 0.f;
#pragma unroll 8
        for (int g = 0; g < nslab; ++g) {
            gm += slabs[(int64_t)g * slab_stride + e];
            gr += slabs[(int64_t)g * slab_stride + n + e];
        }
        if (kl_up) {
            float dm, dr;
            kl_grad_terms(mu[e], rho[e], kl_up[0] * kl_scale, kl_pm, kl_ps, dm, dr);
            gm += dm;
            gr += dr;
        }
        gr *= dsoftplus(rho[e]);
        if (accumulate) { gm += g_mu[e]; gr += g_rho[e]; }
        g_mu[e] = gm;
        g_rho[e] = gr;
        return;
    }
    const int t = threadIdx.x;
    if (!cs_parts || t >= N) return;
    const uint32_t edev = rng_epoch_dev(rng_b);
    float am = 0.f, ar = 0.f;
    for (int s = 0; s < S; ++s) {
        float g = 0.f;
        for (int z = 0; z < Z; ++z) g += cs_parts[((int64_t)s * Z + z) * N + t];
        const float4 zz = eps4(rng_b, edev, (uint32_t)(t >> 2), rng_b.sample0 + (uint32_t)s);
        const float e = (t & 3) == 0 ? zz.x : (t & 3) == 1 ? zz.y : (t & 3) == 2 ? zz.z : zz.w;
        am += g;
        ar = fmaf(g, e, ar);
    }
    const float ds = dsoftplus(rho_b[t]);
    float gm = am, gr = ar * ds;
    if (accumulate) { gm += g_mu_b[t]; gr += g_rho_b[t]; }
    if (mu_b && kl_up) {
        float dm, dr;
        kl_grad_terms(mu_b[t], rho_b[t], kl_up[0] * kl_scale_b, kl_pm_b, kl_ps_b, dm, dr);
        gm += dm;
        gr += dr * ds;
    }
    g_mu_b[t] = gm;
    g_rho_b[t] = gr;
}

// ------------------------------------------------------------------ input gradient, explicit weights
// gx[s][m][k0..k0+3] = sum_n gy[s][m][n] * w[s][n][k0..k0+3]; one thread per (s, m, 4 columns).  For
// layers with few outputs (the 10-wide head) or shapes the fused kernel does not take.
template <bool GBF, bool XBF>
__global__ __launch_bounds__(256) void k_dgrad_plain(const void *__restrict__ gy, int64_t gy_sample_stride, int64_t ldgy,
                                                     const float *__restrict__ w, int64_t w_sample_stride,
                                                     void *__restrict__ gx, int64_t gx_sample_stride, int64_t ldgx,
                                                     int M, int N, int K, int S)
{
    const int kq = (K + 3) / 4;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)S * M * kq) return;
    const int c = (int)(id % kq);
    const int m = (int)((id / kq) % M);
    const int s = (int)(id / ((int64_t)kq * M));
    const int k0 = 4 * c;
    const float *ws = w + (int64_t)s * w_sample_stride;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < N; ++n) {
        float g;
        if constexpr (GBF) g = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(gy)[(int64_t)s * gy_sample_stride + (int64_t)m * ldgy + n] << 16);
        else g = reinterpret_cast<const float *>(gy)[(int64_t)s * gy_sample_stride + (int64_t)m * ldgy + n];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < K) acc[j] = fmaf(g, ws[(int64_t)n * K + k0 + j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (k0 + j >= K) continue;
        const int64_t o = (int64_t)s * gx_sample_stride + (int64_t)m * ldgx + k0 + j;
        if constexpr (XBF) reinterpret_cast<uint16_t *>(gx)[o] = f2bf(acc[j]);
        else reinterpret_cast<float *>(gx)[o] = acc[j];
    }
}

// Vector form (K % 8 == 0, 16-B aligned w and gx): a thread owns 8 output columns of one row; the N
// weights rows it needs are read as float4 pairs (shared by all rows of the batch through L1 / L2).
template <bool GBF, bool XBF>
__global__ __launch_bounds__(256) void k_dgrad_plain_v8(const void *__restrict__ gy, int64_t gy_sample_stride, int64_t ldgy,
                                                        const float *__restrict__ w, int64_t w_sample_stride,
                                                        void *__restrict__ gx, int64_t gx_sample_stride, int64_t ldgx,
                                                        int M, int N, int K, int S)
{
    const int kq = K / 8;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)S * M * kq) return;
    const int c = (int)(id % kq);
    const int m = (int)((id / kq) % M);
    const int s = (int)(id / ((int64_t)kq * M));
    const float *ws = w + (int64_t)s * w_sample_stride + 8 * c;
    const int64_t go = (int64_t)s * gy_sample_stride + (int64_t)m * ldgy;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < N; ++n) {
        float g;
        if constexpr (GBF) g = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(gy)[go + n] << 16);
        else g = reinterpret_cast<const float *>(gy)[go + n];
        const float4 u = *reinterpret_cast<const float4 *>(ws + (int64_t)n * K);
        const float4 v = *reinterpret_cast<const float4 *>(ws + (int64_t)n * K + 4);
        acc[0] = fmaf(g, u.x, acc[0]); acc[1] = fmaf(g, u.y, acc[1]); acc[2] = fmaf(g, u.z, acc[2]); acc[3] = fmaf(g, u.w, acc[3]);
        acc[4] = fmaf(g, v.x, acc[4]); acc[5] = fmaf(g, v.y, acc[5]); acc[6] = fmaf(g, v.z, acc[6]); acc[7] = fmaf(g, v.w, acc[7]);
    }
    const int64_t o = (int64_t)s * gx_sample_stride + (int64_t)m * ldgx + 8 * c;
    if constexpr (XBF) {
        uint4 h;
        h.x = pack_bf16x2(acc[0], acc[1]); h.y = pack_bf16x2(acc[2], acc[3]);
        h.z = pack_bf16x2(acc[4], acc[5]); h.w = pack_bf16x2(acc[6], acc[7]);
        *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(gx) + o) = h;
    } else {
        float *q = reinterpret_cast<float *>(gx) + o;
        *reinterpret_cast<float4 *>(q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4 *>(q + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

// ------------------------------------------------------------------ bias: column sums of gy per sample
// out[s][n] = sum_m gy[s][m][n]; block = 64 columns x 4 row-slices, fixed-order LDS reduction.
template <bool GBF>
__global__ __launch_bounds__(256) void k_colsum(const void *__restrict__ gy, int64_t gy_sample_stride, int64_t ldgy,
                                                float *__restrict__ out, int M, int N)
{
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6, s = blockIdx.y;
    float a = 0.f;
    if (col < N) {
        for (int m = part; m < M; m += 4) {
            const int64_t o = (int64_t)s * gy_sample_stride + (int64_t)m * ldgy + col;
            if constexpr (GBF) a += __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(gy)[o] << 16);
            else a += reinterpret_cast<const float *>(gy)[o];
        }
    }
    red[part][threadIdx.x & 63] = a;
    __syncthreads();
    if (part == 0 && col < N) out[(int64_t)s * N + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Narrow matrices (N <= 16, e.g. the logits' gradient): 16 columns x 64 row slices per block.
template <bool GBF>
__global__ __launch_bounds__(1024) void k_colsum_narrow(const void *__restrict__ gy, int64_t gy_sample_stride, int64_t ldgy,
                                                        float *__restrict__ out, int M, int N)
{
    __shared__ float red[64][17];
    const int col = threadIdx.x & 15, part = threadIdx.x >> 4, s = blockIdx.x;
    float a = 0.f;
    if (col < N) {
        for (int m = part; m < M; m += 64) {
            const int64_t o = (int64_t)s * gy_sample_stride + (int64_t)m * ldgy + col;
            if constexpr (GBF) a += __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(gy)[o] << 16);
            else a += reinterpret_cast<const float *>(gy)[o];
        }
    }
    red[part][col] = a;
    __syncthreads();
    if (threadIdx.x < 16 && threadIdx.x < N) {
        float t = 0.f;
        for (int q = 0; q < 64; ++q) t += red[q][threadIdx.x];
        out[(int64_t)s * N + threadIdx.x] = t;
    }
}

// Vector form (N % 8 == 0, 16-B aligned rows): a thread owns 8 columns (one 16-B load per row for bf16,
// two for fp32); block = 16 column groups x 16 row slices, fixed-order LDS reduction over the slices.
template <bool GBF>
__global__ __launch_bounds__(256) void k_colsum_v8(const void *__restrict__ gy, int64_t gy_sample_stride, int64_t ldgy,
                                                   float *__restrict__ out, int M, int N)
{
    __shared__ float red[16][16][9];
    const int cg = threadIdx.x & 15, part = threadIdx.x >> 4, s = blockIdx.y;
    const int col = blockIdx.x * 128 + cg * 8;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        for (int m = part; m < M; m += 16) {
            const int64_t o = (int64_t)s * gy_sample_stride + (int64_t)m * ldgy + col;
            if constexpr (GBF) {
                const uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(gy) + o);
                a[0] += __uint_as_float(v.x << 16); a[1] += __uint_as_float(v.x & 0xFFFF0000u);
                a[2] += __uint_as_float(v.y << 16); a[3] += __uint_as_float(v.y & 0xFFFF0000u);
                a[4] += __uint_as_float(v.z << 16); a[5] += __uint_as_float(v.z & 0xFFFF0000u);
                a[6] += __uint_as_float(v.w << 16); a[7] += __uint_as_float(v.w & 0xFFFF0000u);
            } else {
                const float4 u = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(gy) + o);
                const float4 w = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(gy) + o + 4);
                a[0] += u.x; a[1] += u.y; a[2] += u.z; a[3] += u.w; a[4] += w.x; a[5] += w.y; a[6] += w.z; a[7] += w.w;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[part][cg][j] = a[j];
    __syncthreads();
    if (threadIdx.x < 128) {
        const int c = threadIdx.x >> 3, j = threadIdx.x & 7;
        const int n = blockIdx.x * 128 + c * 8 + j;
        if (n < N) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += red[q][c][j];
            out[(int64_t)s * N + n] = t;
        }
    }
}

// ------------------------------------------------------------------ ReLU mask: g * (y > 0)
template <bool GBF, bool YBF>
__global__ __launch_bounds__(256) void k_relu_bwd(const void *__restrict__ g, const void *__restrict__ y,
                                                  void *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float yv;
    if constexpr (YBF) yv = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(y)[i] << 16);
    else yv = reinterpret_cast<const float *>(y)[i];
    if constexpr (GBF) {
        const uint16_t gv = reinterpret_cast<const uint16_t *>(g)[i];
        reinterpret_cast<uint16_t *>(out)[i] = yv > 0.f ? gv : (uint16_t)0;
    } else {
        const float gv = reinterpret_cast<const float *>(g)[i];
        reinterpret_cast<float *>(out)[i] = yv > 0.f ? gv : 0.f;
    }
}

// 16-B form for the common case (both bf16, n % 8 == 0, aligned): 8 elements per thread
__global__ __launch_bounds__(256) void k_relu_bwd_bf16x8(const uint4 *__restrict__ g, const uint4 *__restrict__ y,
                                                         uint4 *__restrict__ out, int64_t n8)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const uint4 gv = g[i], yv = y[i];
    // bf16 > 0  <=>  sign bit clear and not (+)0: compare the 16-bit patterns as signed shorts
    auto m2 = [](uint32_t gg, uint32_t yy) -> uint32_t {
        const uint32_t lo = ((int16_t)(yy & 0xFFFFu) > 0) ? 0xFFFFu : 0u;
        const uint32_t hi = ((int16_t)(yy >> 16) > 0) ? 0xFFFF0000u : 0u;
        return gg & (lo | hi);
    };
    out[i] = make_uint4(m2(gv.x, yv.x), m2(gv.y, yv.y), m2(gv.z, yv.z), m2(gv.w, yv.w));
}

static inline bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

}  // namespace bnn

using namespace bnn;

extern "C" {

int bnn_linear_backward_weight_sampled(const void *x, int64_t x_sample_stride, int64_t ldx,
                                       const void *gy, int64_t gy_sample_stride, int64_t ldgy,
                                       const float *rho_w, float *g_mu, float *g_rho, const float *rho_b,
                                       float *g_mu_b, float *g_rho_b, int64_t M, int64_t N,
                                       int64_t K, int nsamples, const bnn_rng_t *rng_w, const bnn_rng_t *rng_b,
                                       const bnn_kl_fuse_t *kl, int compute, int flags, int accumulate, void *stream)
{
    const char *who = "bnn_linear_backward_weight_sampled";
    if (M == 0 && !x) x = rho_w;                                    // empty batch: the operands are never read,
    if (M == 0 && !gy) gy = rho_w;                                  // only the zero / accumulate path below runs
    if (kl && (!kl->upstream || !kl->mu_w || !(kl->prior_sigma_w > 0.f) || (kl->mu_b && !(kl->prior_sigma_b > 0.f)))) {
        set_error("%s: kl needs upstream, mu_w and positive prior sigmas", who);
        return BNN_E_NULL;
    }
    if (!x || !gy || !rho_w || !g_mu || !g_rho) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    const bool want_bias = rho_b || g_mu_b || g_rho_b;
    if (want_bias && (!rho_b || !g_mu_b || !g_rho_b || !rng_b)) { set_error("%s: rho_b, g_mu_b, g_rho_b, rng_b must be given together", who); return BNN_E_NULL; }
    if (M < 0 || N < 1 || K < 1 || nsamples < 1 || ldx < K || ldgy < N) { set_error("%s: bad extent", who); return BNN_E_SHAPE; }
    if (M > 0x7FFFFFFF || N > 0x7FFFFFFF || K > 0x7FFFFFFF || N * K > ((int64_t)1 << 34)) { set_error("%s: extent too large", who); return BNN_E_RANGE; }
    if (K % 4 == 0 && (!al16(g_mu) || !al16(g_rho))) { set_error("%s: gradients must be 16-B aligned", who); return BNN_E_ALIGN; }
    const bool xh = (flags & BNN_FLAG_X_BF16) != 0, gh = (flags & BNN_FLAG_Y_BF16) != 0;
    if ((xh || gh) && compute != BNN_COMPUTE_BF16) { set_error("%s: bf16 operands need bf16 compute", who); return BNN_E_UNSUPPORTED; }
    if (compute != BNN_COMPUTE_F32 && compute != BNN_COMPUTE_BF16) { set_error("%s: unknown compute mode %d", who, compute); return BNN_E_DTYPE; }
    int rc = check_rng(rng_w, nsamples);
    if (rc) { set_error("%s: bad rng_w", who); return rc; }
    if (want_bias) { rc = check_rng(rng_b, nsamples); if (rc) { set_error("%s: bad rng_b", who); return rc; } }
    hipStream_t st = (hipStream_t)stream;
    WgradParams p{};
    p.x = x; p.x_sample_stride = x_sample_stride; p.ldx = ldx;
    p.gy = gy; p.gy_sample_stride = gy_sample_stride; p.ldgy = ldgy;
    p.rho = rho_w; p.g_mu = g_mu; p.g_rho = g_rho;
    p.M = (int32_t)M; p.N = (int32_t)N; p.K = (int32_t)K; p.S = nsamples; p.accumulate = accumulate;
    p.rng = make_rng(rng_w);
    p.vecX = al16(x) && (xh ? (ldx % 8 == 0 && x_sample_stride % 8 == 0) : (ldx % 4 == 0 && x_sample_stride % 4 == 0));
    p.vecG = al16(gy) && (gh ? (ldgy % 8 == 0 && gy_sample_stride % 8 == 0) : (ldgy % 4 == 0 && gy_sample_stride % 4 == 0));
    p.ntk = (int32_t)((K + W_TK - 1) / W_TK);
    p.ntn = (int32_t)((N + W_TN - 1) / W_TN);
    const int64_t tiles = (int64_t)p.ntk * p.ntn;
    // few tiles (small N * K): split the MC samples over gridDim.y, partials through the workspace
    p.nsplit = 1;
    GemmParams ws{};
    fill_workspace(ws);
    if (M == 0) {
        if (!accumulate) {
            rc = (int)hipMemsetAsync(g_mu, 0, (size_t)(N * K) * 4, st);
            if (!rc) rc = (int)hipMemsetAsync(g_rho, 0, (size_t)(N * K) * 4, st);
            if (!rc && want_bias) rc = (int)hipMemsetAsync(g_mu_b, 0, (size_t)N * 4, st);
            if (!rc && want_bias) rc = (int)hipMemsetAsync(g_rho_b, 0, (size_t)N * 4, st);
            if (rc) { set_error("%s: hipMemsetAsync failed (%d)", who, rc); return rc; }
        }
        return BNN_OK;
    }
    if (tiles < 64 && nsamples > 1 && ws.ws_slabs) {
        int ns = nsamples < 8 ? nsamples : 8;
        while (ns > 1 && ((int64_t)ns * 2 * N * K + (int64_t)nsamples * N) * 4 > ws.ws_slab_bytes) --ns;
        p.nsplit = ns;
    }
    if (p.nsplit > 1) {
        p.slab_stride = 2 * N * K;
        p.g_mu = ws.ws_slabs;
        p.g_rho = nullptr;
        if (want_bias) {
            // the sample split leaves no workgroup that sees every sample: column sums and the bias draw's
            // backward as two small launches (scratch: S x N floats behind the partial slabs)
            float *tmp = ws.ws_slabs + (int64_t)p.nsplit * p.slab_stride;
            rc = bnn_colsum(gy, gy_sample_stride, ldgy, tmp, M, N, nsamples, gh ? BNN_FLAG_X_BF16 : 0, stream);
            if (!rc) rc = bnn_sample_affine_bwd(tmp, N, rho_b, nullptr, 0, rng_b, N, nsamples, g_mu_b, g_rho_b, accumulate, stream);
            if (!rc && kl && kl->mu_b) {
                // bias KL gradient on top (one tensor; scale_b already folds 1 / (n_b * T * n_batches))
                const bnn_kl_tensor_t kt = {kl->mu_b, rho_b, N, kl->prior_mu_b, kl->prior_sigma_b};
                float *gmp[1] = {g_mu_b}, *grp[1] = {g_rho_b};
                rc = bnn_kl_backward(&kt, 1, 1.0f / (kl->scale_b * (float)N), kl->upstream, gmp, grp, 1, stream);
            }
            if (rc) return rc;
        }
    } else if (want_bias) {
        p.rho_b = rho_b; p.g_mu_b = g_mu_b; p.g_rho_b = g_rho_b; p.rng_b = make_rng(rng_b);
    }
    if (kl) {
        p.kl_up = kl->upstream; p.mu_w = kl->mu_w; p.mu_b = want_bias ? kl->mu_b : nullptr;
        p.kl_scale_w = kl->scale_w; p.kl_pm_w = kl->prior_mu_w; p.kl_ps_w = kl->prior_sigma_w;
        p.kl_scale_b = kl->scale_b; p.kl_pm_b = kl->prior_mu_b; p.kl_ps_b = kl->prior_sigma_b;
    }
    // diagnostic only (timing split of the loop vs the eps epilogue; results are then NOT the gradient)
    static const bool diag_noeps = [] { const char *e = getenv("BNN_WGRAD_NOEPS"); return e && e[0] == '1'; }();
    if (diag_noeps && p.nsplit == 1) p.plain = 1;
    static const int wdiag = [] { const char *e = getenv("BNN_WGRAD_DIAG"); return e ? atoi(e) : 0; }();
    p.diag = wdiag;
    p.xcd_map = (p.ntk >= 4 && p.ntn >= 8);
    const dim3 grid(p.xcd_map ? (unsigned)(8 * ((p.ntk + 1) / 2) * ((p.ntn + 3) / 4)) : (unsigned)tiles, (unsigned)p.nsplit);
    const bool dma_ok = p.vecX && p.vecG && M % (8 * W_BM) == 0 && K % 8 == 0 && N % 8 == 0;
    if (compute == BNN_COMPUTE_F32) {
        hipLaunchKernelGGL(k_wgrad_f32, grid, dim3(W_NT), 0, st, p);
    } else if (xh && gh && dma_ok) {
        // two-role kernel (MFMA waves + DMA / eps waves) unless BNN_WGRAD_ROLES=1 asks for the one-role one
        static const bool one_role = [] { const char *e = getenv("BNN_WGRAD_ROLES"); return e && e[0] == '1'; }();
        if (one_role) hipLaunchKernelGGL(k_wgrad_bf16_dma, grid, dim3(W_NT), 0, st, p);
        else hipLaunchKernelGGL(k_wgrad_bf16_dma8, grid, dim3(2 * W_NT), 0, st, p);
    } else if (xh && gh) {
        hipLaunchKernelGGL((k_wgrad_bf16<true, true>), grid, dim3(W_NT), 0, st, p);
    } else if (xh) {
        hipLaunchKernelGGL((k_wgrad_bf16<true, false>), grid, dim3(W_NT), 0, st, p);
    } else if (gh) {
        hipLaunchKernelGGL((k_wgrad_bf16<false, true>), grid, dim3(W_NT), 0, st, p);
    } else {
        hipLaunchKernelGGL((k_wgrad_bf16<false, false>), grid, dim3(W_NT), 0, st, p);
    }
    rc = check_launch(who);
    if (rc || p.nsplit == 1) return rc;
    const int64_t n = N * K;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws.ws_slabs, p.slab_stride,
                       p.nsplit, rho_w, g_mu, g_rho, n, accumulate, kl ? kl->upstream : nullptr, kl ? kl->mu_w : nullptr,
                       kl ? kl->scale_w : 0.f, kl ? kl->prior_mu_w : 0.f, kl ? kl->prior_sigma_w : 1.f);
    return check_launch(who);
}

int bnn_linear_backward_narrow_sampled(const void *x, int64_t x_sample_stride, int64_t ldx, const float *gy,
                                       int64_t gy_sample_stride, int64_t ldgy, const float *mu_w, const float *rho_w,
                                       void *gx, int64_t gx_sample_stride, int64_t ldgx, float *g_mu, float *g_rho,
                                       const float *rho_b, float *g_mu_b, float *g_rho_b, int64_t M, int64_t N, int64_t K,
                                       int nsamples, const bnn_rng_t *rng_w, const bnn_rng_t *rng_b,
                                       const bnn_kl_fuse_t *kl, int flags, int accumulate, void *stream)
{
    const char *who = "bnn_linear_backward_narrow_sampled";
    if (!x || !gy || !mu_w || !rho_w || !g_mu || !g_rho) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    const bool want_bias = rho_b || g_mu_b || g_rho_b;
    if (want_bias && (!rho_b || !g_mu_b || !g_rho_b || !rng_b)) { set_error("%s: rho_b, g_mu_b, g_rho_b, rng_b must be given together", who); return BNN_E_NULL; }
    if (kl && (!kl->upstream || !kl->mu_w || !(kl->prior_sigma_w > 0.f) || (kl->mu_b && !(kl->prior_sigma_b > 0.f)))) { set_error("%s: kl needs upstream, mu_w and positive prior sigmas", who); return BNN_E_NULL; }
    if (M < 1 || N < 1 || N > H_NMAX || K < 4 || K % 4 != 0 || nsamples < 1 || nsamples > 65535 || ldx < K || ldgy < N || (gx && ldgx < K)) { set_error("%s: needs 1 <= N <= 16, K %% 4 == 0, M >= 1", who); return BNN_E_UNSUPPORTED; }
    const bool xh = (flags & BNN_FLAG_X_BF16) != 0, gxh = (flags & BNN_FLAG_Y_BF16) != 0;
    const int xe = xh ? 2 : 4, ge = gxh ? 2 : 4;
    if (!al16(mu_w) || !al16(rho_w) || !al16(g_mu) || !al16(g_rho) || (reinterpret_cast<uintptr_t>(x) & (4 * xe - 1)) || (ldx * xe) % (4 * xe) != 0 ||
        (x_sample_stride * xe) % (4 * xe) != 0 || (gx && ((reinterpret_cast<uintptr_t>(gx) & (4 * ge - 1)) || (gx_sample_stride * ge) % (4 * ge) != 0))) {
        set_error("%s: misaligned operand", who);
        return BNN_E_ALIGN;
    }
    int rc = check_rng(rng_w, nsamples);
    if (rc) { set_error("%s: bad rng_w", who); return rc; }
    GemmParams ws{};
    fill_workspace(ws);
    const int64_t Z = (M + H_ROWS - 1) / H_ROWS, nslab = (int64_t)nsamples * Z;
    const int64_t need = (nslab * 2 * N * K + nslab * N + (int64_t)nsamples * N) * 4;
    if (!ws.ws_slabs || need > ws.ws_slab_bytes || Z > 65535 || nslab > 0x7FFFFFFF) { set_error("%s: registered workspace too small (%lld bytes needed)", who, (long long)need); return BNN_E_UNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    HeadBwdParams p{};
    p.x = x; p.x_sample_stride = x_sample_stride; p.ldx = ldx;
    p.gy = gy; p.gy_sample_stride = gy_sample_stride; p.ldgy = ldgy;
    p.mu = mu_w; p.rho = rho_w; p.gx = gx; p.gx_sample_stride = gx_sample_stride; p.ldgx = ldgx;
    p.slabs = ws.ws_slabs; p.slab_stride = 2 * N * K;
    float *cs_parts = ws.ws_slabs + nslab * p.slab_stride, *cs_sum = cs_parts + nslab * N;
    p.colsums = want_bias ? cs_parts : nullptr;
    p.M = (int32_t)M; p.N = (int32_t)N; p.K = (int32_t)K; p.S = nsamples; p.rng = make_rng(rng_w);
    const dim3 grid((unsigned)((K + 63) / 64), (unsigned)nsamples, (unsigned)Z);
#define BNN_HEAD_BWD(NP_) \
    do { \
        if (xh && gxh) hipLaunchKernelGGL((k_head_bwd<true, true, NP_>), grid, dim3(256), 0, st, p); \
        else if (xh) hipLaunchKernelGGL((k_head_bwd<true, false, NP_>), grid, dim3(256), 0, st, p); \
        else if (gxh) hipLaunchKernelGGL((k_head_bwd<false, true, NP_>), grid, dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((k_head_bwd<false, false, NP_>), grid, dim3(256), 0, st, p); \
    } while (0)
    if (N <= 12) BNN_HEAD_BWD(12); else BNN_HEAD_BWD(16);
#undef BNN_HEAD_BWD
    rc = check_launch(who);
    if (rc) return rc;
    const int64_t n = N * K;
    (void)cs_sum;
    RngDev rb{};
    if (want_bias) {
        rc = check_rng(rng_b, nsamples);
        if (rc) { set_error("%s: bad rng_b", who); return rc; }
        rb = make_rng(rng_b);
    }
    const bool klb = want_bias && kl && kl->mu_b;
    hipLaunchKernelGGL(k_head_tail, dim3((unsigned)((n + 255) / 256 + (want_bias ? 1 : 0))), dim3(256), 0, st, ws.ws_slabs, p.slab_stride,
                       (int)nslab, rho_w, g_mu, g_rho, n, accumulate, kl ? kl->upstream : nullptr, kl ? kl->mu_w : nullptr,
                       kl ? kl->scale_w : 0.f, kl ? kl->prior_mu_w : 0.f, kl ? kl->prior_sigma_w : 1.f,
                       want_bias ? cs_parts : nullptr, nsamples, (int)Z, (int)N, rho_b, rb, g_mu_b, g_rho_b,
                       klb ? kl->mu_b : nullptr, klb ? kl->scale_b : 0.f, klb ? kl->prior_mu_b : 0.f, klb ? kl->prior_sigma_b : 1.f);
    return check_launch(who);
}

int bnn_linear_backward_weight(const void *x, int64_t x_sample_stride, int64_t ldx, const void *gy,
                               int64_t gy_sample_stride, int64_t ldgy, float *gw, int64_t gw_sample_stride,
                               int64_t M, int64_t N, int64_t K, int nsamples, int compute, int flags,
                               int accumulate, void *stream)
{
    const char *who = "bnn_linear_backward_weight";
    if (!x || !gy || !gw) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (M < 0 || N < 1 || K < 1 || nsamples < 1 || nsamples > 65535 || ldx < K || ldgy < N || gw_sample_stride < N * K) { set_error("%s: bad extent", who); return BNN_E_SHAPE; }
    if (M > 0x7FFFFFFF || N > 0x7FFFFFFF || K > 0x7FFFFFFF || N * K > ((int64_t)1 << 34)) { set_error("%s: extent too large", who); return BNN_E_RANGE; }
    if (K % 4 == 0 && (!al16(gw) || gw_sample_stride % 4 != 0)) { set_error("%s: gw must be 16-B aligned", who); return BNN_E_ALIGN; }
    const bool xh = (flags & BNN_FLAG_X_BF16) != 0, gh = (flags & BNN_FLAG_Y_BF16) != 0;
    if ((xh || gh) && compute != BNN_COMPUTE_BF16) { set_error("%s: bf16 operands need bf16 compute", who); return BNN_E_UNSUPPORTED; }
    if (compute != BNN_COMPUTE_F32 && compute != BNN_COMPUTE_BF16) { set_error("%s: unknown compute mode %d", who, compute); return BNN_E_DTYPE; }
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        if (!accumulate)
            for (int s = 0; s < nsamples; ++s) {
                const int rc = (int)hipMemsetAsync(gw + (int64_t)s * gw_sample_stride, 0, (size_t)(N * K) * 4, st);
                if (rc) { set_error("%s: hipMemsetAsync failed (%d)", who, rc); return rc; }
            }
        return BNN_OK;
    }
    WgradParams p{};
    p.x = x; p.x_sample_stride = x_sample_stride; p.ldx = ldx;
    p.gy = gy; p.gy_sample_stride = gy_sample_stride; p.ldgy = ldgy;
    p.g_mu = gw; p.slab_stride = gw_sample_stride; p.plain = 1; p.nsplit = nsamples;
    p.M = (int32_t)M; p.N = (int32_t)N; p.K = (int32_t)K; p.S = nsamples; p.accumulate = accumulate;
    p.vecX = al16(x) && (xh ? (ldx % 8 == 0 && x_sample_stride % 8 == 0) : (ldx % 4 == 0 && x_sample_stride % 4 == 0));
    p.vecG = al16(gy) && (gh ? (ldgy % 8 == 0 && gy_sample_stride % 8 == 0) : (ldgy % 4 == 0 && gy_sample_stride % 4 == 0));
    p.ntk = (int32_t)((K + W_TK - 1) / W_TK);
    p.ntn = (int32_t)((N + W_TN - 1) / W_TN);
    p.xcd_map = (p.ntk >= 4 && p.ntn >= 8);
    const dim3 grid(p.xcd_map ? (unsigned)(8 * ((p.ntk + 1) / 2) * ((p.ntn + 3) / 4)) : (unsigned)(p.ntk * p.ntn), (unsigned)nsamples);
    if (compute == BNN_COMPUTE_F32) hipLaunchKernelGGL(k_wgrad_f32, grid, dim3(W_NT), 0, st, p);
    else if (xh && gh) hipLaunchKernelGGL((k_wgrad_bf16<true, true>), grid, dim3(W_NT), 0, st, p);
    else if (xh) hipLaunchKernelGGL((k_wgrad_bf16<true, false>), grid, dim3(W_NT), 0, st, p);
    else if (gh) hipLaunchKernelGGL((k_wgrad_bf16<false, true>), grid, dim3(W_NT), 0, st, p);
    else hipLaunchKernelGGL((k_wgrad_bf16<false, false>), grid, dim3(W_NT), 0, st, p);
    return check_launch(who);
}

int bnn_linear_backward_input(const void *gy, int64_t gy_sample_stride, int64_t ldgy, const float *w,
                              int64_t w_sample_stride, void *gx, int64_t gx_sample_stride, int64_t ldgx,
                              int64_t M, int64_t N, int64_t K, int nsamples, int flags, void *stream)
{
    const char *who = "bnn_linear_backward_input";
    if (M == 0 && N >= 1 && K >= 1 && nsamples >= 1) return BNN_OK;     // empty batch
    if (!gy || !w || !gx) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (M < 0 || N < 1 || K < 1 || nsamples < 1 || ldgy < N || ldgx < K) { set_error("%s: bad extent", who); return BNN_E_SHAPE; }
    const int64_t threads = (int64_t)nsamples * M * ((K + 3) / 4);
    if (M > 0x7FFFFFFF || N > 0x7FFFFFFF || K > 0x7FFFFFFF || threads > ((int64_t)1 << 38)) { set_error("%s: extent too large", who); return BNN_E_RANGE; }
    if (M == 0) return BNN_OK;
    const bool gh = (flags & BNN_FLAG_X_BF16) != 0, xh = (flags & BNN_FLAG_Y_BF16) != 0;
    hipStream_t st = (hipStream_t)stream;
    if (K % 8 == 0 && al16(w) && w_sample_stride % 4 == 0 && al16(gx) && ldgx % 8 == 0 && gx_sample_stride % 8 == 0) {
        const int64_t th8 = (int64_t)nsamples * M * (K / 8);
        const dim3 g8((unsigned)((th8 + 255) / 256));
#define BNN_DGRAD8(G, X) hipLaunchKernelGGL((k_dgrad_plain_v8<G, X>), g8, dim3(256), 0, st, gy, gy_sample_stride, ldgy, w, \
                                            w_sample_stride, gx, gx_sample_stride, ldgx, (int)M, (int)N, (int)K, nsamples)
        if (gh && xh) BNN_DGRAD8(true, true);
        else if (gh) BNN_DGRAD8(true, false);
        else if (xh) BNN_DGRAD8(false, true);
        else BNN_DGRAD8(false, false);
#undef BNN_DGRAD8
        return check_launch(who);
    }
    const dim3 grid((unsigned)((threads + 255) / 256));
#define BNN_DGRAD(G, X) hipLaunchKernelGGL((k_dgrad_plain<G, X>), grid, dim3(256), 0, st, gy, gy_sample_stride, ldgy, w, \
                                           w_sample_stride, gx, gx_sample_stride, ldgx, (int)M, (int)N, (int)K, nsamples)
    if (gh && xh) BNN_DGRAD(true, true);
    else if (gh) BNN_DGRAD(true, false);
    else if (xh) BNN_DGRAD(false, true);
    else BNN_DGRAD(false, false);
#undef BNN_DGRAD
    return check_launch(who);
}

int bnn_colsum(const void *gy, int64_t gy_sample_stride, int64_t ldgy, float *out, int64_t M, int64_t N,
               int nsamples, int flags, void *stream)
{
    const char *who = "bnn_colsum";
    if (M == 0 && out && N >= 1 && nsamples >= 1) {                  // empty batch: sums are zero
        const int rc0 = (int)hipMemsetAsync(out, 0, (size_t)nsamples * N * 4, (hipStream_t)stream);
        if (rc0) { set_error("%s: hipMemsetAsync failed (%d)", who, rc0); return rc0; }
        return BNN_OK;
    }
    if (!gy || !out) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (M < 0 || N < 1 || nsamples < 1 || ldgy < N) { set_error("%s: bad extent", who); return BNN_E_SHAPE; }
    if (M > 0x7FFFFFFF || N > 0x7FFFFFFF || nsamples > 65535) { set_error("%s: extent too large", who); return BNN_E_RANGE; }
    const bool gh = (flags & BNN_FLAG_X_BF16) != 0;
    if (N % 8 == 0 && ldgy % 8 == 0 && gy_sample_stride % 8 == 0 && al16(gy)) {
        const dim3 g8((unsigned)((N + 127) / 128), (unsigned)nsamples);
        if (gh) hipLaunchKernelGGL((k_colsum_v8<true>), g8, dim3(256), 0, (hipStream_t)stream, gy, gy_sample_stride, ldgy, out, (int)M, (int)N);
        else hipLaunchKernelGGL((k_colsum_v8<false>), g8, dim3(256), 0, (hipStream_t)stream, gy, gy_sample_stride, ldgy, out, (int)M, (int)N);
        return check_launch(who);
    }
    if (N <= 16) {
        if (gh) hipLaunchKernelGGL((k_colsum_narrow<true>), dim3((unsigned)nsamples), dim3(1024), 0, (hipStream_t)stream, gy, gy_sample_stride, ldgy, out, (int)M, (int)N);
        else hipLaunchKernelGGL((k_colsum_narrow<false>), dim3((unsigned)nsamples), dim3(1024), 0, (hipStream_t)stream, gy, gy_sample_stride, ldgy, out, (int)M, (int)N);
        return check_launch(who);
    }
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)nsamples);
    if (gh) hipLaunchKernelGGL((k_colsum<true>), grid, dim3(256), 0, (hipStream_t)stream, gy, gy_sample_stride, ldgy, out, (int)M, (int)N);
    else hipLaunchKernelGGL((k_colsum<false>), grid, dim3(256), 0, (hipStream_t)stream, gy, gy_sample_stride, ldgy, out, (int)M, (int)N);
    return check_launch(who);
}

int bnn_relu_backward(const void *g, const void *y, void *out, int64_t n, int flags, void *stream)
{
    const char *who = "bnn_relu_backward";
    if (!g || !y || !out) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (n < 0 || n > ((int64_t)1 << 38)) { set_error("%s: bad extent", who); return BNN_E_SHAPE; }
    if (n == 0) return BNN_OK;
    const bool gh = (flags & BNN_FLAG_X_BF16) != 0, yh = (flags & BNN_FLAG_Y_BF16) != 0;
    hipStream_t st = (hipStream_t)stream;
    if (gh && yh && n % 8 == 0 && al16(g) && al16(y) && al16(out)) {
        const int64_t n8 = n / 8;
        hipLaunchKernelGGL(k_relu_bwd_bf16x8, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint4 *>(g),
                           reinterpret_cast<const uint4 *>(y), reinterpret_cast<uint4 *>(out), n8);
        return check_launch(who);
    }
    const dim3 grid((unsigned)((n + 255) / 256));
    if (gh && yh) hipLaunchKernelGGL((k_relu_bwd<true, true>), grid, dim3(256), 0, st, g, y, out, n);
    else if (gh) hipLaunchKernelGGL((k_relu_bwd<true, false>), grid, dim3(256), 0, st, g, y, out, n);
    else if (yh) hipLaunchKernelGGL((k_relu_bwd<false, true>), grid, dim3(256), 0, st, g, y, out, n);
    else hipLaunchKernelGGL((k_relu_bwd<false, false>), grid, dim3(256), 0, st, g, y, out, n);
    return check_launch(who);
}

}  // extern "C"
