// bnn_dense.hip -- the draw-once path of the sampled linear layer (bf16 compute mode):
//
//   (1) k_draw_multi : K1 for EVERY posterior tensor of a forward in one launch -- w_s = mu + sigma(rho) * eps_s for all
//       S MC samples, written as bf16 rows zero-padded to a multiple of 64 columns (biases: fp32) -- optionally
//       carrying the KL's first pass as extra workgroups (it reads the same mu / rho);
//   (2) k_dense_bf16 : y[s] = act(x[s] . w_s^T + b_s) on the drawn weights -- a dense MFMA GEMM whose operands both
//       arrive by LDS-DMA (bf16 operands, or three bf16 planes per operand for the fp32 parity mode);
//   (3) k_head_bf16  : the same contraction for N <= 16 (the classifier head), K split over the waves of a workgroup.
//
// Why not fused (bnn_linear.hip draws inside the GEMM): one Philox block + Box-Muller + softplus is ~150 VALU issue
// slots per 4 weights; inside the GEMM that stream shares each SIMD's issue port with the MFMAs and is paced by the
// LDS hand-off between drawing and consuming waves (round-1 PMC: 48 % of a wave's timeline was hand-off; 200 workgroups
// on 256 CUs).  Drawn once up front the draw runs on all 1024 SIMDs at full occupancy with sigma computed once for the
// S samples of a weight (softplus: 36 of the ~150 slots), and the contraction becomes a GEMM bound by the L2 -> LDS
// DMA rate and the MFMA pipe.  The 2 bytes per drawn weight this writes and re-reads stay in L2 / Infinity Cache.
//
// k_dense_bf16 structure (MI355X: 160 KB LDS, LDS-DMA, 4 SIMDs per CU) -- details at the kernel:
//   * 8 waves per workgroup, two roles: waves 0-3 consume (an NWM x NWN grid of 16 TM x 16 TN wave tiles: fragment reads,
//     MFMAs, epilogue), waves 4-7 load (LDS-DMA of both operands, 1-KiB pieces taken round-robin).  One raw s_barrier per
//     64-k step, met by all eight waves, is the only hand-off; stages are requested ST - 1 steps ahead and waited for with
//     counted s_waitcnt vmcnt.
//   * tiles: 128 x 160 with a 4-stage ring (the BASELINE layers: 256 workgroups), 64 / 32 x 160 for launches over few
//     samples, 256 x 128 with 3 stages (wide layers), 256 x 80.
//   * LDS image of both operands: [row][8 x 16 B] with 16-B chunk c of row r at position c ^ (r & 7); the DMA writes
//     linearly (lane l -> position l & 7 of row l >> 3 of its 8-row piece), so lane l FETCHES chunk (l & 7) ^ (l >> 3):
//     eight lanes read one whole 128-B line.  ds_read_b128 fragment reads of this image are conflict-free.
//   * v_mfma_f32_16x16x32_bf16; lane (i = l & 15, q = l >> 4) holds k = 8 q .. 8 q + 7 of a 32-k half-step = chunk
//     4 h + q of the row: the DMA'd 16 B ARE the fragment.  Fragment reads are interleaved into the MFMA stream.
//   * K tail: A's chunk address is clamped inside the row (finite data), the weights are ZERO there (padded rows).
//   * XCD map: blockIdx % 8 = MC sample (mod 8): a sample's activations and drawn weights (1.2 + 2.9 MB at the
//     BASELINE layer) live in one XCD's 4 MB L2.
//   * epilogue: bias, optional ReLU, then each wave's results leave through its quarter of the ring as whole 16-B row chunks
//     -- fp32, bf16, or the three bf16 planes of the fp32 result.
//   * fp32 parity mode: every operand as three bf16 planes, six plane-pair k-steps per k-block (bnn_dense_forward_x3).
#include <cstdlib>
#include <type_traits>

#include "bnn_device.hpp"
#include "bnn_dma.hpp"
#include "bnn_kl_body.hpp"

namespace bnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------ (1) multi-tensor draw
constexpr int kDrawMaxTensors = 8;
constexpr int kDrawSpreadBelow = 4096;      // tensors of fewer 8-column groups: one thread per (group, sample)

struct DrawTensorDev {
    const float *mu;
    const float *rho;
    void *out;
    int64_t out_sample_stride;  // elements
    int32_t rows, cols, ld;     // (rows, cols) posterior, output rows of ld >= cols elements (zeros beyond cols)
    int32_t bf16;               // output: 0 fp32, 1 bf16, 2 three bf16 planes h, m, l (plane stride = nsamples * out_sample_stride)
    int32_t kind;               // 0: draw mu + sigma eps; 1: mu as it is; 2: sigma = 1e-10 + softplus(rho) (Flipout's two operands)
    int32_t perm_taps;          // > 1: a conv weight (O, C, KH, KW) written tap-major: column c * taps + t -> t * (cols / taps) + c
    int32_t first_item;         // first work item (8-column group) of this tensor within the launch
    int32_t spread;             // 1: a small tensor -- one thread per (8-column group, MC sample) instead of per group
    // flat (a large regular weight matrix): items first_item .. are the VALID 8-column groups in flat element order (item i =
    // elements 8 i .. 8 i + 7: a workgroup = 2048 consecutive scalars = one workgroup of the KL's first pass), items pad_first ..
    // the groups of the zero padding (rows x (ld - cols) / 8 of them, written as zeros by workgroups of their own)
    int32_t flat, pad_first;
    // tapg (a conv weight written tap-major, C % 8 == 0): a workgroup takes whole GROUPS of 8 channels x taps (8 taps consecutive
    // elements = `taps` work items), transposes them through LDS and stores whole 16-B chunks (tap t, channels c0 .. c0 + 7)
    int32_t tapg;
    // kl_on: this tensor's KL partial sums are computed by its own draw items (the eight (mu, rho) are in registers): partial
    // kl_first + workgroup index within the tensor; prior as in KlTensorDev
    int32_t kl_on, kl_first;
    float kl_prior_mu, kl_prior_sigma;
    RngDev rng;
};
struct DrawLaunch {
    DrawTensorDev t[kDrawMaxTensors];
    int32_t ntensors;
    int32_t nsamples;
    int32_t total_items;
    int32_t draw_blocks;        // workgroups of the draw (the first L.kl.nblocks workgroups of the grid run the KL's first pass)
    KlPiggy kl;
};

// One work item = 8 consecutive columns of one row = two Philox blocks; sigma once, then the S samples.
// SMALL tensors (a bias, a classifier head: `spread`) take one thread per (item, sample) instead: a thread that draws its S
// samples one after the other is a serial chain of ~1.2 us per sample when its wave has a SIMD to itself, and a launch is as
// long as its longest wave -- the three 1-workgroup bias tensors of the BASELINE net were 8.7 of the launch's 29 us
// (tools/draw_exp2.py: 20.5 us without them), although they are 0.1 % of the draws.
// A tensor's items start at a multiple of 256 (first_item): a workgroup works on ONE tensor, whose descriptor is then
// wave-uniform -- read once with scalar loads and kept in SGPRs.  (Indexed per thread, L.t[ti] was re-read from the
// kernel-argument segment by vector loads inside the sample loop -- the stores may alias it as far as the compiler
// knows -- and every s_waitcnt vmcnt(0) on those loads also waited for the previous sample's store: PMC showed the
// waves parked 47 % of the time.)
template <int U>
__global__ __launch_bounds__(256) void k_draw_multi(const DrawLaunch L)
{
    // The KL's workgroups come FIRST in the grid: they are short (8 or 16 scalars per thread, no sample loop) and the dispatcher
    // hands workgroups out in index order -- at the end of the grid they waited for a draw workgroup to retire and became the
    // launch's tail (BASELINE launch, 16-bit stream: 20.0 us; in front: see DESIGN).
    if ((int)blockIdx.x < L.kl.nblocks) {
        if (blockIdx.y == 0) kl_piggy_block(L.kl, (int)blockIdx.x);
        return;
    }
    const int item0 = ((int)blockIdx.x - L.kl.nblocks) * 256;
    int ti = 0;
#pragma unroll
    for (int i = 1; i < kDrawMaxTensors; ++i)
        if (i < L.ntensors && item0 >= L.t[i].first_item) ti = i;
    ti = __builtin_amdgcn_readfirstlane(ti);
    // the descriptor, copied out of the argument block once
    const float *const t_mu = L.t[ti].mu;
    const float *const t_rho = L.t[ti].rho;
    char *const t_out = reinterpret_cast<char *>(L.t[ti].out);
    const int64_t t_stride = L.t[ti].out_sample_stride;
    const int t_rows = L.t[ti].rows, t_cols = L.t[ti].cols, t_ld = L.t[ti].ld;
    const bool t_bf16 = L.t[ti].bf16 != 0;
    const bool t_x3 = L.t[ti].bf16 == 2;
    const int t_taps = L.t[ti].perm_taps;
    const int t_kind = L.t[ti].kind;
    const RngDev rng = L.t[ti].rng;
    const int S = L.nsamples;
    int local = item0 + (int)threadIdx.x - L.t[ti].first_item;
    if (L.t[ti].flat) {
        // ================= a large regular weight matrix: flat items, KL partial sums from the items themselves =================
        const int esz2 = 2;
        const int64_t sbytes2 = t_stride * esz2, pbytes2 = sbytes2 * S;
        const int padg = (t_ld - t_cols) >> 3;                          // padding groups per row
        if (item0 >= L.t[ti].pad_first) {
            // the zero padding beyond column `cols` (the dense kernel's K tail multiplies it with clamped, finite activations)
            const int pi = item0 + (int)threadIdx.x - L.t[ti].pad_first;
            const int prow = pi / (padg > 0 ? padg : 1);
            if (padg == 0 || prow >= t_rows) return;
            char *d0 = t_out + ((int64_t)prow * t_ld + t_cols + ((pi - prow * padg) << 3)) * esz2;
            for (int s = (int)blockIdx.y; s < S; s += (int)gridDim.y)
                for (int pl = 0; pl < (t_x3 ? 3 : 1); ++pl) *reinterpret_cast<uint4 *>(d0 + s * sbytes2 + pl * pbytes2) = make_uint4(0u, 0u, 0u, 0u);
            return;
        }
        const int64_t e0 = (int64_t)local * 8;
        const bool live = e0 < (int64_t)t_rows * t_cols;
        float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0, r0 = m0, r1 = m0;
        if (live) {
            m0 = *reinterpret_cast<const float4 *>(t_mu + e0); m1 = *reinterpret_cast<const float4 *>(t_mu + e0 + 4);
            r0 = *reinterpret_cast<const float4 *>(t_rho + e0); r1 = *reinterpret_cast<const float4 *>(t_rho + e0 + 4);
        }
        if (L.t[ti].kl_on && blockIdx.y == 0) {
            // the first pass of this tensor's KL: kl_partial_block<8>'s thread sum (same eight scalars, same order) and reduction
            const float inv_ps = __uint_as_float(uniform_vgpr(__float_as_uint(1.0f / L.t[ti].kl_prior_sigma)));
            const float pmu = __uint_as_float(uniform_vgpr(__float_as_uint(L.t[ti].kl_prior_mu)));
            float acc = 0.f;
            if (live) {
                acc += kl_elem(m0.x, r0.x, pmu, inv_ps); acc += kl_elem(m0.y, r0.y, pmu, inv_ps);
                acc += kl_elem(m0.z, r0.z, pmu, inv_ps); acc += kl_elem(m0.w, r0.w, pmu, inv_ps);
                acc += kl_elem(m1.x, r1.x, pmu, inv_ps); acc += kl_elem(m1.y, r1.y, pmu, inv_ps);
                acc += kl_elem(m1.z, r1.z, pmu, inv_ps); acc += kl_elem(m1.w, r1.w, pmu, inv_ps);
            }
            kl_block_reduce(acc, L.t[ti].kl_first + ((item0 - L.t[ti].first_item) >> 8), L.kl.partials);
        }
        if (!live) return;
        const int gpc = t_cols >> 3;                                    // valid groups per row (32-bit division)
        const int row = local / gpc, c0 = (local - row * gpc) << 3;
        const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        const float sg[8] = {sigma_draw(r0.x), sigma_draw(r0.y), sigma_draw(r0.z), sigma_draw(r0.w),
                             sigma_draw(r1.x), sigma_draw(r1.y), sigma_draw(r1.z), sigma_draw(r1.w)};
        const uint32_t edev = rng_epoch_dev(rng);
        const PhiloxKeys keys = philox_keys(rng.key0, rng.key1);
        const uint32_t blk = (uint32_t)(e0 >> 2);
        const bool gen16 = rng.gen == BNN_GEN_PHILOX7_U16;      // (wave-uniform)
        char *dst = t_out + ((int64_t)row * t_ld + c0) * esz2 + (int64_t)blockIdx.y * sbytes2;
        const int64_t step = sbytes2 * (int64_t)gridDim.y;
#pragma unroll U
        for (int s = (int)blockIdx.y; s < S; s += (int)gridDim.y, dst += step) {
            const uint32_t sample = rng.sample0 + (uint32_t)s;
            float4 za, zb;
            if (gen16) {
                eps8_u16(rng, keys, edev, blk >> 1, sample, za, zb);    // the item IS one 8-eps block
            } else {
                za = eps4(rng, keys, edev, blk, sample);
                zb = eps4(rng, keys, edev, blk + 1u, sample);
            }
            float w[8];
            w[0] = fmaf(sg[0], za.x, m[0]); w[1] = fmaf(sg[1], za.y, m[1]);
            w[2] = fmaf(sg[2], za.z, m[2]); w[3] = fmaf(sg[3], za.w, m[3]);
            w[4] = fmaf(sg[4], zb.x, m[4]); w[5] = fmaf(sg[5], zb.y, m[5]);
            w[6] = fmaf(sg[6], zb.z, m[6]); w[7] = fmaf(sg[7], zb.w, m[7]);
            if (t_x3) {
                uint4 h, mm, l;
                split_bf16x3(w[0], w[1], h.x, mm.x, l.x); split_bf16x3(w[2], w[3], h.y, mm.y, l.y);
                split_bf16x3(w[4], w[5], h.z, mm.z, l.z); split_bf16x3(w[6], w[7], h.w, mm.w, l.w);
                *reinterpret_cast<uint4 *>(dst) = h;
                *reinterpret_cast<uint4 *>(dst + pbytes2) = mm;
                *reinterpret_cast<uint4 *>(dst + 2 * pbytes2) = l;
            } else {
                uint4 o;
                o.x = pack_bf16x2(w[0], w[1]); o.y = pack_bf16x2(w[2], w[3]);
                o.z = pack_bf16x2(w[4], w[5]); o.w = pack_bf16x2(w[6], w[7]);
                *reinterpret_cast<uint4 *>(dst) = o;
            }
        }
        return;
    }
    if (L.t[ti].tapg) {
        // ================= a conv weight (O, C, KH, KW), tap-major output =================
        // Element (o, c, t) sits at flat index (o C + c) taps + t and goes to column t C + c of row o.  A GROUP = the 8 taps
        // consecutive elements of 8 channels of one row = `taps` items of 8 consecutive elements (one 8-eps block each); its
        // output is `taps` 16-B chunks.  A workgroup takes ng = 2048 / (8 taps) groups: the items scatter their eight bf16 values
        // into an LDS image [group][tap][8 channels], then thread i stores chunk i -- instead of eight 2-B stores per item (the
        // general body: 8.3 us for the 1.18 M draws of the configs[3] layer).  Samples over blockIdx.y.
        __shared__ __attribute__((aligned(16))) uint16_t timg[3][256 * 8];
        const int taps = t_taps;
        const int ng = 2048 / (8 * taps);
        const int Cn = t_cols / taps, C8 = Cn >> 3;
        const int NG = t_rows * C8;
        const int g0 = ((item0 - L.t[ti].first_item) >> 8) * ng;
        const int i = (int)threadIdx.x;
        const int gl = i / taps, b = i - gl * taps;                   // local group, item within the group (= tap of phase 2)
        const bool live = gl < ng && g0 + gl < NG;
        const int64_t e0 = ((int64_t)(g0 + gl) * taps + b) * 8;       // flat index of the item's first element
        float m[8], sg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { m[j] = 0.f; sg[j] = 0.f; }
        if (live) {
            const float4 m0 = *reinterpret_cast<const float4 *>(t_mu + e0), m1 = *reinterpret_cast<const float4 *>(t_mu + e0 + 4);
            const float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
            if (t_kind == 1 || t_kind == 3) {
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = mm[j];
            } else {
                const float4 r0 = *reinterpret_cast<const float4 *>(t_rho + e0), r1 = *reinterpret_cast<const float4 *>(t_rho + e0 + 4);
                const float rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (t_kind == 2) m[j] = sigma_accurate(rr[j]);
                    else { m[j] = mm[j]; sg[j] = sigma_draw(rr[j]); }
                }
            }
        }
        // where the item's element j goes in the image: channel (8 b + j) / taps of the group, tap (8 b + j) % taps
        int pos[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int jj = 8 * b + j, cl = jj / taps, tp = jj - cl * taps;
            pos[j] = (gl * taps + tp) * 8 + cl;
        }
        const int G = g0 + gl;
        const int orow = live ? G / C8 : 0, cg = live ? G - orow * C8 : 0;
        const int npl = t_x3 ? 3 : 1;
        const int64_t sbytes_t = t_stride * 2, pbytes_t = sbytes_t * (t_kind == 3 ? 1 : S);
        char *const drow = t_out + (int64_t)orow * t_ld * 2;
        const uint32_t edev = rng_epoch_dev(rng);
        const PhiloxKeys keys = philox_keys(rng.key0, rng.key1);
        const bool gen16 = rng.gen == BNN_GEN_PHILOX7_U16;      // (wave-uniform)
        const int s_end = t_kind == 3 ? 1 : S;
        for (int s = (int)blockIdx.y; s < s_end; s += (int)gridDim.y) {
            if (live) {
                float4 za = make_float4(0.f, 0.f, 0.f, 0.f), zb = za;
                if (t_kind == 0) {
                    const uint32_t sample = rng.sample0 + (uint32_t)s, blk = (uint32_t)(e0 >> 2);
                    if (gen16) eps8_u16(rng, keys, edev, blk >> 1, sample, za, zb);
                    else { za = eps4(rng, keys, edev, blk, sample); zb = eps4(rng, keys, edev, blk + 1u, sample); }
                }
                const float w[8] = {fmaf(sg[0], za.x, m[0]), fmaf(sg[1], za.y, m[1]), fmaf(sg[2], za.z, m[2]), fmaf(sg[3], za.w, m[3]),
                                    fmaf(sg[4], zb.x, m[4]), fmaf(sg[5], zb.y, m[5]), fmaf(sg[6], zb.z, m[6]), fmaf(sg[7], zb.w, m[7])};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (t_x3) {
                        uint32_t h, mm_, l;
                        split_bf16x3(w[j], 0.f, h, mm_, l);
                        timg[0][pos[j]] = (uint16_t)h; timg[1][pos[j]] = (uint16_t)mm_; timg[2][pos[j]] = (uint16_t)l;
                    } else timg[0][pos[j]] = f2bf(w[j]);
                }
            }
            __syncthreads();
            if (live) {
                char *d = drow + (int64_t)s * sbytes_t + ((int64_t)b * Cn + cg * 8) * 2;
                for (int pl = 0; pl < npl; ++pl)
                    *reinterpret_cast<uint4 *>(d + pl * pbytes_t) = *reinterpret_cast<const uint4 *>(&timg[pl][i * 8]);
                // the zero padding beyond column `cols`: the items of a row's last group write its (ld - cols) / 8 chunks
                if (cg == C8 - 1)
                    for (int pc = b; pc < ((t_ld - t_cols) >> 3); pc += taps)
                        for (int pl = 0; pl < npl; ++pl)
                            *reinterpret_cast<uint4 *>(drow + (int64_t)s * sbytes_t + pl * pbytes_t + ((int64_t)t_cols + 8 * pc) * 2) = make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();
        }
        return;
    }
    const int gpr = (t_ld + 7) >> 3;                 // 8-column groups per output row
    // the samples this thread draws: s_lo, s_lo + s_step, ... < s_hi
    int s_lo = (int)blockIdx.y, s_hi = S, s_step = (int)gridDim.y;
    if (L.t[ti].spread) {
        if (blockIdx.y != 0) return;
        const int nit = t_rows * gpr;
        s_lo = local / nit;
        local -= s_lo * nit;
        s_hi = s_lo < S ? s_lo + 1 : 0;
        s_step = 1;
    }
    if (t_kind == 3) {
        // kind 3: the tensor as it is, ONCE (not per sample): with a three-plane output this is bnn_split_bf16x3 of an
        // activation -- the fp32 parity mode's input planes -- riding in the draw launch instead of a launch of its own
        if (L.t[ti].spread ? s_lo != 0 : blockIdx.y != 0) return;
        s_lo = 0; s_hi = 1; s_step = 1;
    }
    const int row = local / gpr, c0 = (local - row * gpr) << 3;
    if (row >= t_rows) return;
    const int64_t orow = (int64_t)row * t_ld + c0;
    const int esz = t_bf16 ? 2 : 4;
    char *const dst0 = t_out + orow * esz;
    const int64_t sbytes = t_stride * esz;
    const int64_t pbytes = sbytes * (t_kind == 3 ? 1 : L.nsamples);     // (three-plane output; kind 3: one copy, planes out_sample_stride apart)
    // ---- the regular case, where the time goes: a weight matrix drawn as bf16 (or as three bf16 planes) whose rows are whole,
    // 16-B aligned 8-column groups.  What decides the code path is wave-uniform (read off the tensor's descriptor), the sample
    // loop is straight-line code, and the groups in the zero padding run the same loop with mean = sigma = 0 (a few % of wasted
    // draws instead of a divergent second loop in almost every wave: a row end falls into most 64-item spans).
    const bool regular = t_bf16 && t_kind == 0 && t_taps <= 1 && (t_cols & 7) == 0 &&
                         (((reinterpret_cast<uintptr_t>(t_mu) | reinterpret_cast<uintptr_t>(t_rho)) & 15u) == 0);
    if (regular) {
        const bool pad = c0 >= t_cols;
        const int64_t e0 = pad ? 0 : (int64_t)row * t_cols + c0;
        const float4 m0 = *reinterpret_cast<const float4 *>(t_mu + e0), m1 = *reinterpret_cast<const float4 *>(t_mu + e0 + 4);
        const float4 r0 = *reinterpret_cast<const float4 *>(t_rho + e0), r1 = *reinterpret_cast<const float4 *>(t_rho + e0 + 4);
        const float live = pad ? 0.f : 1.f;          // (0 * finite = 0; a NaN parameter would show in its own row anyway)
        const float m[8] = {pad ? 0.f : m0.x, pad ? 0.f : m0.y, pad ? 0.f : m0.z, pad ? 0.f : m0.w,
                            pad ? 0.f : m1.x, pad ? 0.f : m1.y, pad ? 0.f : m1.z, pad ? 0.f : m1.w};
        const float sg[8] = {live * sigma_draw(r0.x), live * sigma_draw(r0.y), live * sigma_draw(r0.z), live * sigma_draw(r0.w),
                             live * sigma_draw(r1.x), live * sigma_draw(r1.y), live * sigma_draw(r1.z), live * sigma_draw(r1.w)};
        const uint32_t edev = rng_epoch_dev(rng);
        const PhiloxKeys keys = philox_keys(rng.key0, rng.key1);
        const uint32_t blk = (uint32_t)(e0 >> 2);
        char *dst = dst0 + (int64_t)s_lo * sbytes;
        const int64_t step = sbytes * (int64_t)s_step;
        const bool gen16 = rng.gen == BNN_GEN_PHILOX7_U16;      // (wave-uniform)
#pragma unroll U
        for (int s = s_lo; s < s_hi; s += s_step, dst += step) {
            const uint32_t sample = rng.sample0 + (uint32_t)s;
            float4 za, zb;
            if (gen16) {
                eps8_u16(rng, keys, edev, blk >> 1, sample, za, zb);    // the item IS one 8-eps block (e0 % 8 == 0)
            } else {
                za = eps4(rng, keys, edev, blk, sample);
                zb = eps4(rng, keys, edev, blk + 1u, sample);
            }
            float w[8];
            w[0] = fmaf(sg[0], za.x, m[0]); w[1] = fmaf(sg[1], za.y, m[1]);
            w[2] = fmaf(sg[2], za.z, m[2]); w[3] = fmaf(sg[3], za.w, m[3]);
            w[4] = fmaf(sg[4], zb.x, m[4]); w[5] = fmaf(sg[5], zb.y, m[5]);
            w[6] = fmaf(sg[6], zb.z, m[6]); w[7] = fmaf(sg[7], zb.w, m[7]);
            if (t_x3) {
                uint4 h, mm, l;
                split_bf16x3(w[0], w[1], h.x, mm.x, l.x); split_bf16x3(w[2], w[3], h.y, mm.y, l.y);
                split_bf16x3(w[4], w[5], h.z, mm.z, l.z); split_bf16x3(w[6], w[7], h.w, mm.w, l.w);
                *reinterpret_cast<uint4 *>(dst) = h;
                *reinterpret_cast<uint4 *>(dst + pbytes) = mm;
                *reinterpret_cast<uint4 *>(dst + 2 * pbytes) = l;
            } else {
                uint4 o;
                o.x = pack_bf16x2(w[0], w[1]); o.y = pack_bf16x2(w[2], w[3]);
                o.z = pack_bf16x2(w[4], w[5]); o.w = pack_bf16x2(w[6], w[7]);
                *reinterpret_cast<uint4 *>(dst) = o;
            }
        }
        return;
    }
    if (c0 >= t_cols) {
        // padding columns: zeros (the dense kernel's K tail multiplies them with clamped, finite activations)
        for (int s = s_lo; s < s_hi; s += s_step) {
            if (t_bf16) {
                for (int pl = 0; pl < (t_x3 ? 3 : 1); ++pl) *reinterpret_cast<uint4 *>(dst0 + s * sbytes + pl * pbytes) = make_uint4(0u, 0u, 0u, 0u);
            } else
                for (int j = 0; j < 8; ++j)
                    if (c0 + j < t_ld) reinterpret_cast<float *>(dst0 + s * sbytes)[j] = 0.f;
        }
        return;
    }
    const int64_t e0 = (int64_t)row * t_cols + c0;   // flat element index: a multiple of 4 (cols % 4 == 0 or rows == 1)
    const int nval = t_cols - c0 < 8 ? t_cols - c0 : 8;
    const bool full = nval == 8 && (((reinterpret_cast<uintptr_t>(t_mu) | reinterpret_cast<uintptr_t>(t_rho)) & 15u) == 0) && (e0 & 3) == 0;
    float m[8], sg[8];
    if (t_kind != 0) {
        // no draw: the posterior's mean or its stddev themselves (Flipout contracts on both, dense.py:70-83 / conv.py:207-221)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            m[j] = j < nval ? (t_kind != 2 ? t_mu[e0 + j] : sigma_accurate(t_rho[e0 + j])) : 0.f;
            sg[j] = 0.f;
        }
    } else if (full) {
        const float4 m0 = *reinterpret_cast<const float4 *>(t_mu + e0), m1 = *reinterpret_cast<const float4 *>(t_mu + e0 + 4);
        const float4 r0 = *reinterpret_cast<const float4 *>(t_rho + e0), r1 = *reinterpret_cast<const float4 *>(t_rho + e0 + 4);
        m[0] = m0.x; m[1] = m0.y; m[2] = m0.z; m[3] = m0.w; m[4] = m1.x; m[5] = m1.y; m[6] = m1.z; m[7] = m1.w;
        sg[0] = sigma_draw(r0.x); sg[1] = sigma_draw(r0.y); sg[2] = sigma_draw(r0.z); sg[3] = sigma_draw(r0.w);
        sg[4] = sigma_draw(r1.x); sg[5] = sigma_draw(r1.y); sg[6] = sigma_draw(r1.z); sg[7] = sigma_draw(r1.w);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            m[j] = j < nval ? t_mu[e0 + j] : 0.f;
            sg[j] = j < nval ? sigma_draw(t_rho[e0 + j]) : 0.f;
        }
    }
    const uint32_t edev = rng_epoch_dev(rng);
    const PhiloxKeys keys = philox_keys(rng.key0, rng.key1);
    const uint32_t blk = (uint32_t)(e0 >> 2);
    const bool vec_store = t_bf16 && nval == 8 && t_taps <= 1;
    // gridDim.y > 1 (small launches: a conv weight is 18 workgroups): the samples are dealt over blockIdx.y as well -- sigma is
    // recomputed per slice, but a thread is no longer one long serial chain of S draws on a mostly idle chip
#pragma unroll U
    for (int s = s_lo; s < s_hi; s += s_step) {
        const uint32_t sample = rng.sample0 + (uint32_t)s;
        float4 za = make_float4(0.f, 0.f, 0.f, 0.f), zb = za;
        if (t_kind == 0) {
            za = eps4(rng, keys, edev, blk, sample);
            zb = eps4(rng, keys, edev, blk + 1u, sample);
        }
        float w[8];
        w[0] = fmaf(sg[0], za.x, m[0]); w[1] = fmaf(sg[1], za.y, m[1]);
        w[2] = fmaf(sg[2], za.z, m[2]); w[3] = fmaf(sg[3], za.w, m[3]);
        w[4] = fmaf(sg[4], zb.x, m[4]); w[5] = fmaf(sg[5], zb.y, m[5]);
        w[6] = fmaf(sg[6], zb.z, m[6]); w[7] = fmaf(sg[7], zb.w, m[7]);
        char *dst = dst0 + s * sbytes;
        if (vec_store && t_x3) {
            uint4 h, m, l;
            split_bf16x3(w[0], w[1], h.x, m.x, l.x); split_bf16x3(w[2], w[3], h.y, m.y, l.y);
            split_bf16x3(w[4], w[5], h.z, m.z, l.z); split_bf16x3(w[6], w[7], h.w, m.w, l.w);
            *reinterpret_cast<uint4 *>(dst) = h;
            *reinterpret_cast<uint4 *>(dst + pbytes) = m;
            *reinterpret_cast<uint4 *>(dst + 2 * pbytes) = l;
        } else if (vec_store) {
            uint4 o;
            o.x = pack_bf16x2(w[0], w[1]); o.y = pack_bf16x2(w[2], w[3]);
            o.z = pack_bf16x2(w[4], w[5]); o.w = pack_bf16x2(w[6], w[7]);
            *reinterpret_cast<uint4 *>(dst) = o;
        } else if (t_taps > 1) {
            // conv weight: element (o, c, tap) goes to column tap * C + c of row o (the implicit-GEMM kernel walks K
            // tap by tap, 64 channels at a time); the eps stream is addressed by the ORIGINAL flat index all the same
            const int Cn = t_cols / t_taps;
            uint16_t *rowp = reinterpret_cast<uint16_t *>(t_out + ((int64_t)row * t_ld) * 2 + s * sbytes);
            for (int j = 0; j < nval; ++j) {
                const int col = c0 + j, c = col / t_taps, tp = col - c * t_taps;
                if (t_x3) {
                    uint32_t h, m, l;
                    split_bf16x3(w[j], 0.f, h, m, l);
                    rowp[tp * Cn + c] = (uint16_t)h;
                    *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(rowp + tp * Cn + c) + pbytes) = (uint16_t)m;
                    *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(rowp + tp * Cn + c) + 2 * pbytes) = (uint16_t)l;
                } else
                    rowp[tp * Cn + c] = f2bf(w[j]);
            }
            // a group that straddles the end of the row (cols % 8 != 0): its padding columns are zeros like the rest of the padding
            for (int j = nval; j < 8 && c0 + j < t_ld; ++j)
                for (int pl = 0; pl < (t_x3 ? 3 : 1); ++pl)
                    *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(rowp + c0 + j) + pl * pbytes) = 0;
        } else {
            // a bias, or the ragged end of a row whose padding starts inside this group: values, then zeros
            for (int j = 0; j < 8; ++j) {
                if (c0 + j >= t_ld) break;
                const float v = j < nval ? w[j] : 0.f;
                if (t_x3) {
                    uint32_t h, m, l;
                    split_bf16x3(v, 0.f, h, m, l);
                    reinterpret_cast<uint16_t *>(dst)[j] = (uint16_t)h;
                    reinterpret_cast<uint16_t *>(dst + pbytes)[j] = (uint16_t)m;
                    reinterpret_cast<uint16_t *>(dst + 2 * pbytes)[j] = (uint16_t)l;
                } else if (t_bf16) reinterpret_cast<uint16_t *>(dst)[j] = f2bf(v);
                else reinterpret_cast<float *>(dst)[j] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ (2) dense GEMM
struct DenseParams {
    const uint16_t *A;          // (S or 1) x M x lda bf16
    int64_t a_sample_stride;    // elements; 0 = input shared by all samples
    int64_t lda;
    const uint16_t *W;          // S x N x ldw bf16, zero beyond K up to ldw >= roundup(K, 64)
    int64_t w_sample_stride;
    int64_t ldw;
    const float *bias;          // S x N fp32 or NULL
    int64_t bias_sample_stride;
    void *Y;                    // S x M x ldy, fp32 or bf16
    int64_t y_sample_stride;
    int64_t ldy;
    int32_t M, N, K, S;
    int32_t ntm, ntn;
    int32_t flags;              // BNN_FLAG_RELU, BNN_FLAG_Y_BF16
    // fp32 parity mode (bnn_dense_forward_x3): every operand is THREE bf16 planes h, m, l (v = h + m + l to 2^-24 |v|), plane
    // stride in elements; 0 = plain bf16 operands
    int32_t x3;
    int64_t a_plane_stride, w_plane_stride, y_plane_stride;
    // fused classifier head (bnn_dense_forward_head, YM = 3): the layer's output is NOT stored -- every consumer wave contracts
    // its (16 TM) x (16 TN) tile of act(x w^T + b), rounded to bf16 as a stored hidden activation would be, with the matching
    // columns of the head's drawn weights Wh (S x Nh x ldwh bf16, Nh <= 16) and stores the partial logits
    //   P[part][s][m][j],  part = column panel * NWN + wave column  (ntn * NWN partials; partial 0 also carries the head's bias)
    const uint16_t *Wh;
    int64_t wh_sample_stride, ldwh;
    const float *bias_h;
    int64_t bias_h_sample_stride;
    float *P;
    int32_t Nh;
    int64_t wh_plane_stride;    // x3: the head's weights are three bf16 planes too
};

constexpr int kDenseLoaderPrio = 1 << 21;   // internal flag (BNN_DENSE_LOADER_PRIO=1): loader waves at priority 2 (experiment)
constexpr int kDenseNoXcdMap = 1 << 20;     // internal flag (BNN_DENSE_XCD=0): plain sample-major block order, for A/B runs

// one LDS-DMA piece, scalar-base form: 64 lanes x 16 B from (base + voff) land at lds + 16 * lane
__device__ __forceinline__ void dma_piece(const void *base, uint32_t voff, uint32_t lds)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds) : "memory");
}

// ROLE split (8 waves): waves 0-3 consume (fragment reads + MFMA + epilogue), waves 4-7 load.  The consumers form an
// NWM x NWN grid of (16 TM) x (16 TN) wave tiles; a stage of the ring is the workgroup's A rows followed by its B rows
// (128-B rows, XOR-swizzled 16-B chunks), cut into 1-KiB pieces that loader i takes round-robin (piece q = i + 4 j).
// One s_barrier per 64-k step, met by all eight waves, is the only hand-off: at barrier t the loaders have waited for
// THEIR pieces of step t (counted vmcnt: the later steps' stay in flight) and the consumers have finished step t - 1,
// whose stage the loaders refill right after the barrier with step t + ST - 1.  The DMA issue stream (~5 instructions
// per 1-KiB piece) thus runs on the SIMD's second wave beside the MFMA stream instead of in front of it (first version,
// one role: DMA-only 16.6 us, MFMA-only 18.8 us, together 26.8 us).
//   <4, 5, 4, 1, 3>: 256 x 80 tile, 42 KiB per step, 3 stages (126 KiB)
//   <4, 5, 2, 2, 4>: 128 x 160 tile, 36 KiB per step for the same 40 MFMAs per wave, 4 stages (144 KiB) -- the BASELINE
//                    layers: 4 x 8 x 8 = 256 workgroups (N = 1200 rounds up to 1280: the clamped rows are never stored)
//   <4, 8, 4, 1, 3>: 256 x 128 tile for wide layers
// DIAG (BNN_DENSE_DIAG): timing-only builds whose outputs are wrong: 1 = consumers skip reads and MFMAs, 2 = loaders skip the
// DMA, 5 = no DMA and no per-step barriers either (2 vs 5 = what the barriers cost: layer 2 13.85 vs 11.9 us, ~200 cycles per
// step); correct builds for A/B runs of the read interleave: 3 = one MFMA per interleaved fragment read, 4 = two.
template <int TM, int TN, int NWM, int NWN, int ST, int YM, bool RELU, int DIAG = 0>
__global__ __launch_bounds__(512) void k_dense_bf16(const DenseParams p)
{
    constexpr int NWV = 4;                              // consumer waves = loader waves
    constexpr bool INTERLEAVE = true;
    constexpr int MPR = DIAG == 3 ? 1 : DIAG == 4 ? 2 : TN >= 8 ? 2 : 1;                // MFMAs per interleaved fragment read (measured: 64 x 80 wave tile 15.9 / 16.9 us with 1 / 2, 64 x 128 at 4096^3 171 / 140 us)
    static_assert(NWM * NWN == NWV, "four consumer waves");
    constexpr int WM = 16 * TM, WN = 16 * TN, BM = NWM * WM, BN = NWN * WN;
    constexpr int A_TOTAL = BM / 8;                     // 1-KiB pieces (8 rows x 128 B) of the A rows per stage
    constexpr int B_TOTAL = BN / 8;
    static_assert(A_TOTAL % NWV == 0, "A pieces divide evenly over the loaders");
    constexpr int A_PIECES = A_TOTAL / NWV;
    constexpr int B_BASE = B_TOTAL / NWV, B_EXTRA = B_TOTAL % NWV;   // loader i issues B_BASE (+1 if i < B_EXTRA)
    constexpr int A_STAGE = BM * 128;                   // bytes
    constexpr int B_STAGE = BN * 128;
    constexpr int STAGE = A_STAGE + B_STAGE;
    // epilogue staging (after the final barrier): EPI_A 16-row blocks of the wave's output tile at a time in its quarter of the ring
    constexpr bool YBF = YM != 0;                       // YM: 0 fp32, 1 bf16, 2 three bf16 planes (h, m, l) of the fp32 result, 3 fused head (no output)
    constexpr int ESZ = YBF ? 2 : 4;
    constexpr int EPI_BYTES = ST * STAGE / NWV;
    constexpr int EPI_A = (EPI_BYTES / (16 * (WN * ESZ + 16))) < TM ? (EPI_BYTES / (16 * (WN * ESZ + 16))) : TM;
    static_assert(EPI_A >= 1, "a quarter of the ring must hold at least 16 output rows (epilogue staging)");
    __shared__ __attribute__((aligned(16))) char lds[ST * STAGE];

    // ---- block decode: sample -> XCD when the samples fill the 8 XCDs
    // The hardware deals workgroups to the 8 XCDs round-robin (blockIdx % 8).  S % 8 == 0: XCD x runs the samples x, x + 8, ...
    // Otherwise (a wide layer at S = 1): XCD x takes a CONTIGUOUS eighth of the launch's tiles, so that the ~32 workgroups it runs
    // at a time are neighbours.  Within a sample the tiles are walked in groups of 4 tile rows, column panel by column panel
    // (32 consecutive tiles = 4 x 8 tiles: 4 A row blocks + 8 W panels per k-step through that XCD's L2 instead of 1 + 32 -- or,
    // dealt round-robin, nearly all of both operands through every L2: the 4096^2 layer in the fp32 mode moved 5.65 x its
    // algorithmic bytes over the fabric).
    const int per_s = p.ntm * p.ntn;
    int s, t;
    {
        const int L = (int)blockIdx.x;
        if (p.flags & kDenseNoXcdMap) {
            s = L / per_s;
            t = L % per_s;
        } else if (p.S % 8 == 0) {
            const int idx = L >> 3;
            s = (L & 7) + 8 * (idx / per_s);
            t = idx % per_s;
        } else {
            const int total = per_s * p.S, x = L & 7;
            const int u = x * (total >> 3) + (x < (total & 7) ? x : (total & 7)) + (L >> 3);    // XCD x: slots base_x .. base_x + count_x - 1
            s = u / per_s;
            t = u % per_s;
        }
    }
    int mt, panel;
    if (per_s > 32 && !(p.flags & kDenseNoXcdMap)) {
        constexpr int GM = 4;
        const int grp = t / (GM * p.ntn), first = grp * GM;
        const int gm = p.ntm - first < GM ? p.ntm - first : GM;
        const int v = t - grp * (GM * p.ntn);
        mt = first + v % gm;
        panel = v / gm;
    } else {
        mt = t / p.ntn;
        panel = t % p.ntn;
    }
    const int m0 = mt * BM, n0 = panel * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // x3: the six largest partial products of (xh + xm + xl)(wh + wm + wl) as 64-k steps on plane pairs (l, h) (h, l) (m, m)
    // (m, h) (h, m) -- those five for every k-block first -- and then (h, h) over all of K: the consumers see a plain GEMM of
    // depth 6 K (dropped terms <= 2^-25 |x w|: below one fp32 rounding; the products of bnn_linear.hip's kComputeBf16x3)
    const int nk = (p.K + 63) / 64 * (p.x3 ? 6 : 1);

    if (wave >= NWV) {
        // =============================== loader ===============================
        const int lw = wave - NWV;
        if (p.flags & kDenseLoaderPrio) __builtin_amdgcn_s_setprio(2);
        const uint32_t ring = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
        // Lane l writes position l & 7 of row l >> 3 of its piece and therefore fetches chunk (l & 7) ^ (l >> 3) of
        // that row (the image's XOR swizzle, applied on the source address).
        const int prow = lane >> 3;
        const int schunk = (lane & 7) ^ prow;
        const char *a_base = reinterpret_cast<const char *>(p.A + (int64_t)s * p.a_sample_stride);
        const char *w_base = reinterpret_cast<const char *>(p.W + (int64_t)s * p.w_sample_stride);
        uint32_t a_off[A_PIECES], b_off[B_BASE + 1];
        uint32_t a_dst[A_PIECES], b_dst[B_BASE + 1];    // LDS byte offsets of this wave's pieces within stage 0
#pragma unroll
        for (int j = 0; j < A_PIECES; ++j) {
            const int r0 = (lw + NWV * j) * 8;          // piece q = lw + 4 j of the A rows
            int m = m0 + r0 + prow;
            m = m < p.M ? m : p.M - 1;                  // rows >= M: clamped, results never stored
            a_off[j] = (uint32_t)((int64_t)m * p.lda * 2);
            a_dst[j] = (uint32_t)((r0 / WM) * (ST * WM * 128) + (r0 % WM) * 128);
        }
#pragma unroll
        for (int j = 0; j < B_BASE + 1; ++j) {
            const int r0 = (lw + NWV * j) * 8;          // piece q = lw + 4 j of the B rows
            int n = n0 + r0 + prow;
            n = n < p.N ? n : p.N - 1;                  // rows >= N: clamped, results never stored
            b_off[j] = (uint32_t)((int64_t)n * p.ldw * 2) + 16u * (uint32_t)schunk;
            b_dst[j] = (uint32_t)(ST * A_STAGE + (r0 / WN) * (ST * WN * 128) + (r0 % WN) * 128);
        }
        const uint32_t a_colmax = (uint32_t)(p.K * 2 - 16);    // last legal 16-B chunk of an activation row
        auto run = [&](auto nbp_c) {
            constexpr int NBP = decltype(nbp_c)::value;
            constexpr int P = A_PIECES + NBP;           // VMEM ops of this wave per stage
            auto issue = [&](int kt) {
                if constexpr (DIAG == 2 || DIAG == 5 || DIAG == 8 || DIAG == 9) return;
                const int stage = kt % ST;
                int kb = kt;
                const char *ab = a_base, *wb = w_base;
                if (p.x3) {
                    // TWO sweeps over K: first the five small plane pairs (l,h) (h,l) (m,m) (m,h) (h,m) of every k-block, then
                    // (h,h) over all of K.  The accumulator stays ~2^-8 of its final size through the first sweep, so the
                    // 5 K / 32 fp32 roundings of the small terms are 2^-8 smaller than when they were interleaved with the
                    // large products block by block (round 2: six pairs per k-block): max |fp32 mode - float64| on the
                    // BASELINE net 1.36e-4 -> see DESIGN (the reference's own fp32 sgemm: 6.6e-5).  Same steps, same traffic.
                    const int nkb = (p.K + 63) >> 6;
                    int pr = 5;
                    if (kt < 5 * nkb) { kb = kt / 5; pr = kt - 5 * kb; } else kb = kt - 5 * nkb;
                    ab += ((0x001102 >> (4 * pr)) & 3) * (p.a_plane_stride * 2);
                    wb += ((0x010120 >> (4 * pr)) & 3) * (p.w_plane_stride * 2);
                }
                const uint32_t colA0 = (uint32_t)(kb * 128 + 16 * schunk);
                const uint32_t colA = colA0 < a_colmax ? colA0 : a_colmax;   // k >= K: any finite chunk (the weights are 0 there)
                const uint32_t colB = (uint32_t)(kb * 128);
                const uint32_t sa = ring + (uint32_t)(stage * (WM * 128)), sb = ring + (uint32_t)(stage * (WN * 128));
#pragma unroll
                for (int j = 0; j < A_PIECES; ++j)
                    dma_piece(ab, a_off[j] + colA, sa + a_dst[j]);
#pragma unroll
                for (int j = 0; j < NBP; ++j)
                    dma_piece(wb, b_off[j] + colB, sb + b_dst[j]);
            };
#pragma unroll
            for (int i = 0; i < ST - 1; ++i)
                if (i < nk) issue(i);
            for (int kt = 0; kt < nk; ++kt) {
                // this wave's pieces of step kt have landed (the later steps' pieces may still be in flight)
                const int ahead = nk - 1 - kt < ST - 2 ? nk - 1 - kt : ST - 2;
                if (ST >= 4 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * P) : "memory");
                else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if constexpr (DIAG != 5 && DIAG != 9) __builtin_amdgcn_s_barrier();           // barrier kt: stage kt complete; stage (kt - 1) % ST is free
                if (kt + ST - 1 < nk) issue(kt + ST - 1);
            }
            __builtin_amdgcn_s_barrier();               // final barrier: the consumers reuse the ring for the epilogue
        };
        if (lw < B_EXTRA) run(std::integral_constant<int, B_BASE + 1>{});
        else run(std::integral_constant<int, B_BASE>{});
        return;
    }

    // =============================== consumer ===============================
    static_assert(ST == 3 || ST == 4, "ring depth");
    // DIAG 6 (BNN_DENSE_STAMPS=<device pointer>, a diagnostic build whose outputs stay correct): consumer wave 0 of every
    // workgroup leaves s_memrealtime stamps (100 MHz) -- entry, first stage landed, loop done, stores issued, stores retired --
    // in a buffer of its own (5 x uint64 per workgroup); no output value depends on them
    uint64_t stamp[5] = {0, 0, 0, 0, 0};
    if constexpr (DIAG >= 6) stamp[0] = __builtin_amdgcn_s_memrealtime();
    const int fi = lane & 15, fq = lane >> 4;
    const int wm = wave / NWN, wn = wave % NWN;
    // LDS: [wm][stage][WM rows] for A, then [wn][stage][WN rows] for B -- every fragment of a wave within 64 KiB of its two
    // bases (the reach of the ds_read immediate); with whole stages laid out one after the other the compiler carries an
    // extra lane-varying base per far stage and the 256 x 128 tile's accumulators spill
    const char *a_rows = lds + wm * (ST * WM * 128);        // + stage * WM * 128
    const char *b_rows = lds + ST * A_STAGE + wn * (ST * WN * 128);
    __builtin_amdgcn_s_setprio(1);                      // the MFMA stream goes first on its SIMD
    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment (row 16 x + fi, 32-k half h) sits at  row * 128 + (((4 h + fq) ^ (row & 7)) << 4), and row & 7 = fi & 7:
    // two lane-constant bases (h = 0, 1) + compile-time offsets (16-row block, stage)
    const int fbase0 = fi * 128 + (((0 + fq) ^ (fi & 7)) << 4);
    const int fbase1 = fi * 128 + (((4 + fq) ^ (fi & 7)) << 4);
    // Fragment registers are double-buffered over the two 32-k halves of a step, and barrier t + 1 sits in the MIDDLE of
    // step t: [reads (t, h1)] [MFMAs (t, h0)] [reads done -> barrier t + 1] [reads (t + 1, h0)] [MFMAs (t, h1)].  Every group
    // of fragment reads is issued one MFMA group (20 MFMAs, 320 cycles) before its first use, and a wave reaches the
    // barrier with its reads of stage t complete -- the invariant the loaders refill on.
    uint4 fa[2][TM], fb[2][TN];
    // stage: 0 .. ST - 1, a RUNTIME wave-uniform value (one v_add per group of reads): the loop below is not unrolled over the
    // ring, and its body has ONE chain of MFMAs (the last step is peeled) -- with the MFMAs of a phase in both arms of an
    // `if (more steps)` the accumulators were allocated twice (239 VGPRs for the 64 x 80 wave tile instead of 162; the
    // 64 x 128 tile spilled)
    auto rd = [&](auto buf_c, uint32_t stage, auto h_c) {
        constexpr int buf = decltype(buf_c)::value, h = decltype(h_c)::value;
        if constexpr (DIAG == 1 || DIAG == 7) return;
        const char *As = a_rows + (stage * (uint32_t)(WM * 128) + (uint32_t)(h ? fbase1 : fbase0));
        const char *Bs = b_rows + (stage * (uint32_t)(WN * 128) + (uint32_t)(h ? fbase1 : fbase0));
#pragma unroll
        for (int b = 0; b < TN; ++b) fb[buf][b] = *reinterpret_cast<const uint4 *>(Bs + b * 2048);
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[buf][a] = *reinterpret_cast<const uint4 *>(As + a * 2048);
    };
    auto mm = [&](auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        if constexpr (DIAG == 1 || DIAG == 7) return;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                // operands SWAPPED (w rows as the MFMA's A, x rows as its B): the accumulator block is the TRANSPOSE of the output
                // block -- lane (i, q), register r holds output row i, column 4 q + r -- so a lane's four values are four
                // CONSECUTIVE columns of one output row and leave as one 8-B (bf16) / 16-B (fp32) LDS write in the epilogue,
                // instead of four 2-B writes to four rows (320 ds_write_b16 per wave: 2.0 us of a 12.6-us workgroup)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[buf][b]),
                                                                    __builtin_bit_cast(bf16x8, fa[buf][a]), acc[a][b], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // Within a phase the fragment reads are interleaved into the MFMA stream, one ds_read_b128 behind each of the first
    // TM + TN MFMAs: a read issued in an MFMA's shadow costs no issue time, and none is needed before the next phase
    // (block scheduling: the reads-then-MFMAs order cost ~250 cycles of a 900-cycle step with the MFMA pipe idle).
    auto interleave = [&]() {
        if constexpr ((DIAG == 0 || (DIAG >= 3 && DIAG != 7)) && INTERLEAVE) {
            constexpr int NI = (TM * TN) / MPR < TM + TN ? (TM * TN) / MPR : TM + TN;   // reads that get MFMAs in front of them
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);    // MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one DS read
            }
            if constexpr (TM + TN > NI) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN - NI, 0);
            if constexpr (TM * TN > MPR * NI) __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - MPR * NI, 0);
        }
    };
    auto first_half = [&](uint32_t stage) {
        rd(I1{}, stage, I1{});                          // (t, h1) -> buffer 1
        if constexpr (!INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
        mm(I0{});                                       // (t, h0)
        interleave();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto second_half = [&](uint32_t next) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // this wave's reads of stage t have returned
        if constexpr (DIAG != 5 && DIAG != 9) __builtin_amdgcn_s_barrier();      // barrier t + 1
        asm volatile("" ::: "memory");
        rd(I0{}, next, I0{});                           // (t + 1, h0) -> buffer 0
        if constexpr (!INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
        mm(I1{});                                       // (t, h1)
        interleave();
        __builtin_amdgcn_sched_barrier(0);
    };
    // the bias of this wave's columns, requested BEFORE the loop (4 TN registers): loaded in the epilogue it was a full memory
    // round trip in front of the first store (stamps: 2.0 us from the last MFMA to the stores' issue)
    const float *bias = p.bias ? p.bias + (int64_t)s * p.bias_sample_stride : nullptr;
    const int mw = m0 + wm * WM, nw = n0 + wn * WN;
    // (the 64 x 128 wave tile has no registers for it -- 128 accumulators + 96 fragment registers: the prefetch spilled up to 102
    // registers there -- and loads the bias in the epilogue, where 2 us are 0.3 % of its launch)
    constexpr bool BIAS_PRE = TM * TN <= 20;
    float bv[TN][4];
    auto load_bias = [&]() {
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nw + b * 16 + fq * 4 + r;
                bv[b][r] = (bias && n < p.N) ? bias[n] : 0.f;
            }
    };
    if constexpr (BIAS_PRE) load_bias();
    if (DIAG == 0 && (nw >= p.N || mw >= p.M)) {
        // This wave's whole 16 TM x 16 TN tile lies in the padding of the grid (the BASELINE layers: N = 1200 is 7.5 column
        // panels of 160 -- the second wave column of the last panel, 6.25 % of the launch's MFMAs, LDS reads and energy): it keeps
        // the workgroup's barriers company and computes nothing.  (A fused head's partial slot of this wave is written as zeros:
        // the reduction adds every slot.)
        for (int kt = 0; kt < nk + 1; ++kt) __builtin_amdgcn_s_barrier();        // barrier 0, the nk - 1 mid-step ones, the final one
        if constexpr (YM == 3) {
            float *P = p.P + ((int64_t)(panel * NWN + wn) * p.S + s) * (int64_t)p.M * p.Nh;
            for (int i = lane; i < WM * p.Nh; i += 64) {
                const int m = mw + i / p.Nh;
                if (m < p.M) P[(int64_t)m * p.Nh + (i % p.Nh)] = 0.f;
            }
        }
        return;
    }
    if constexpr (DIAG != 5 && DIAG != 9) __builtin_amdgcn_s_barrier();         // barrier 0
    asm volatile("" ::: "memory");
    if constexpr (DIAG >= 6) stamp[1] = __builtin_amdgcn_s_memrealtime();
    rd(I0{}, 0u, I0{});
    uint32_t stage = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const uint32_t next = stage + 1 == (uint32_t)ST ? 0u : stage + 1;
        first_half(stage);
        second_half(next);
        stage = next;
    }
    first_half(stage);                                  // the last step: nothing left to read ahead
    mm(I1{});
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // final barrier: every wave's last fragment read is behind it
    asm volatile("" ::: "memory");
    if constexpr (DIAG >= 6) stamp[2] = __builtin_amdgcn_s_memrealtime();
    auto leave_stamps = [&]() {
        if constexpr (DIAG >= 6) {
            stamp[3] = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp[4] = __builtin_amdgcn_s_memrealtime();
            if (wave == 0 && lane == 0 && p.P) {
                uint64_t *q = reinterpret_cast<uint64_t *>(p.P) + (int64_t)blockIdx.x * 5;
                for (int i = 0; i < 5; ++i) q[i] = stamp[i];
            }
        }
    };

    // ---- epilogue: bias, activation, store (accumulator lane (i, q), register r = output row i, column 4 q + r of a 16 x 16 block)
    if constexpr (!BIAS_PRE) load_bias();               // (the fragment registers are dead here: one exposed round trip)
    const int64_t ybase = (int64_t)s * p.y_sample_stride * ESZ;
    // the lane's four values of block (a, b) after bias and activation
    auto vals = [&](int a, int b, float (&v)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = acc[a][b][r] + bv[b][r];
            if (RELU) v[r] = fmaxf(v[r], 0.f);
        }
    };
    if constexpr (YM == 3) {
        constexpr bool X3_HEAD_FITS = 3 * WM * (WN * 2 + 16) <= EPI_BYTES;        // (the 128 / 64 / 32 x 160 tiles; the launcher refuses the others)
        if constexpr (X3_HEAD_FITS) if (p.x3) {
            // ---- fused head, fp32 parity mode: the fp32 tile is split into its three bf16 planes (exact to 2^-24), staged
            // plane by plane, and contracted with the three planes of the head's weights on the six plane pairs of the main loop
            // -- the five small ones into an accumulator of their own, (h, h) into another, added once
            constexpr int pitch = WN * 2 + 16;
            char *T = lds + wave * EPI_BYTES;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    float v[4];
                    vals(a, b, v);
                    uint32_t h0, m0, l0, h1, m1, l1;
                    split_bf16x3(v[0], v[1], h0, m0, l0);
                    split_bf16x3(v[2], v[3], h1, m1, l1);
                    char *q = T + (a * 16 + fi) * pitch + (b * 16 + fq * 4) * 2;
                    *reinterpret_cast<uint2 *>(q) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2 *>(q + WM * pitch) = make_uint2(m0, m1);
                    *reinterpret_cast<uint2 *>(q + 2 * WM * pitch) = make_uint2(l0, l1);
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int Nh = p.Nh;
            const int hrow = fi < Nh ? fi : Nh - 1;
            const uint16_t *wh = p.Wh + (int64_t)s * p.wh_sample_stride + (int64_t)hrow * p.ldwh;
            f32x4 hacc[TM], hsm[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) { hacc[a] = f32x4{0.f, 0.f, 0.f, 0.f}; hsm[a] = hacc[a]; }
            constexpr int NC = (WN + 31) / 32;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int kcol = 32 * c + 8 * fq;
                const int n = nw + kcol;
                const bool okb = kcol < WN && fi < Nh && n + 8 <= (int)p.ldwh;
                uint4 bf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    bf[pl] = okb ? *reinterpret_cast<const uint4 *>(wh + pl * p.wh_plane_stride + n) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    uint4 af[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        af[pl] = kcol < WN ? *reinterpret_cast<const uint4 *>(T + pl * WM * pitch + (a * 16 + fi) * pitch + kcol * 2) : make_uint4(0u, 0u, 0u, 0u);
                    auto mf = [&](f32x4 c4, int pa, int pw) {
                        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[pa]), __builtin_bit_cast(bf16x8, bf[pw]), c4, 0, 0, 0);
                    };
                    hsm[a] = mf(mf(mf(mf(mf(hsm[a], 2, 0), 0, 2), 1, 1), 1, 0), 0, 1);      // (l,h) (h,l) (m,m) (m,h) (h,m)
                    hacc[a] = mf(hacc[a], 0, 0);                                             // (h,h)
                }
            }
            const int part = panel * NWN + wn;
            float hb = 0.f;
            if (part == 0 && p.bias_h && fi < Nh) hb = p.bias_h[(int64_t)s * p.bias_h_sample_stride + fi];
            float *P = p.P + ((int64_t)part * p.S + s) * (int64_t)p.M * Nh;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mw + a * 16 + fq * 4 + r;
                    if (m < p.M && fi < Nh) P[(int64_t)m * Nh + fi] = (hacc[a][r] + hsm[a][r]) + hb;
                }
            return;
        }
        // ---- fused head: stage the wave's tile as bf16 rows (pitch 16 TN * 2 + 16 B: the 16 rows of a fragment read -- and
        // of a staging write -- hit 16 different bank groups), read it back as A fragments, one MFMA per (16-row block, 32
        // columns) against the head's weights for those hidden units (fragments straight from memory: Nh rows x 16 TN
        // columns, read once per wave)
        constexpr int pitch = WN * 2 + 16;
        static_assert(WM * pitch <= EPI_BYTES, "the wave's quarter of the ring holds its bf16 tile");
        char *T = lds + wave * EPI_BYTES;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                float v[4];
                vals(a, b, v);
                *reinterpret_cast<uint2 *>(T + (a * 16 + fi) * pitch + (b * 16 + fq * 4) * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int Nh = p.Nh;
        const int hrow = fi < Nh ? fi : Nh - 1;
        const uint16_t *wh = p.Wh + (int64_t)s * p.wh_sample_stride + (int64_t)hrow * p.ldwh;
        f32x4 hacc[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) hacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NC = (WN + 31) / 32;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int kcol = 32 * c + 8 * fq;              // column of the wave tile
            const int n = nw + kcol;                       // hidden unit
            // hidden units >= N: the weight rows of THIS layer were clamped there (duplicates of row N - 1), so the head's
            // weights must be zero: they are, up to their row pitch (zero padding of the draw); beyond it nothing is read
            uint4 bf = make_uint4(0u, 0u, 0u, 0u);
            if (kcol < WN && fi < Nh && n + 8 <= (int)p.ldwh) bf = *reinterpret_cast<const uint4 *>(wh + n);
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                uint4 af = make_uint4(0u, 0u, 0u, 0u);
                if (kcol < WN) af = *reinterpret_cast<const uint4 *>(T + (a * 16 + fi) * pitch + kcol * 2);
                // (plain operand order: lane (i, q) of the result holds rows 4 q + r of the batch, head output i)
                hacc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), hacc[a], 0, 0, 0);
            }
        }
        const int part = panel * NWN + wn;
        float hb = 0.f;
        if (part == 0 && p.bias_h && fi < Nh) hb = p.bias_h[(int64_t)s * p.bias_h_sample_stride + fi];
        float *P = p.P + ((int64_t)part * p.S + s) * (int64_t)p.M * Nh;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mw + a * 16 + fq * 4 + r;
                if (m < p.M && fi < Nh) P[(int64_t)m * Nh + fi] = hacc[a][r] + hb;
            }
        return;
    }
    // plane pl of the result: bf16(v), bf16(v - h), bf16(v - h - m) -- the split of split_bf16x3 (residuals exact in fp32)
    auto plane_of = [&](float v, int pl) -> uint16_t {
        const uint16_t h = f2bf(v);
        if (pl == 0) return h;
        const float r = v - __uint_as_float((uint32_t)h << 16);
        const uint16_t m = f2bf(r);
        if (pl == 1) return m;
        return f2bf(r - __uint_as_float((uint32_t)m << 16));
    };
    constexpr int NP = YM == 2 ? 3 : 1;
    const bool wide = nw + WN <= p.N && mw + WM <= p.M && (p.ldy * ESZ) % 16 == 0 && (nw * ESZ) % 16 == 0 &&
                      ((reinterpret_cast<uintptr_t>(p.Y) + ybase) & 15u) == 0 && (NP == 1 || (p.y_plane_stride * ESZ) % 16 == 0);
    if (wide) {
        char *T = lds + wave * EPI_BYTES;
        constexpr int row_bytes = WN * ESZ;
        constexpr int pitch = row_bytes + 16;           // + 16 B: the 16 rows of a staging write land in 16 different bank groups
        constexpr int cpr = row_bytes / 16;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            char *Y8 = reinterpret_cast<char *>(p.Y) + ybase + ((int64_t)pl * p.y_plane_stride + (int64_t)mw * p.ldy + nw) * ESZ;
#pragma unroll
            for (int a0 = 0; a0 < TM; a0 += EPI_A) {
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int a = a0; a < a0 + EPI_A && a < TM; ++a) {
                        float v[4];
                        vals(a, b, v);
                        char *q = T + ((a - a0) * 16 + fi) * pitch + (b * 16 + fq * 4) * ESZ;
                        if (YM == 2)
                            *reinterpret_cast<uint2 *>(q) = make_uint2((uint32_t)plane_of(v[0], pl) | ((uint32_t)plane_of(v[1], pl) << 16),
                                                                       (uint32_t)plane_of(v[2], pl) | ((uint32_t)plane_of(v[3], pl) << 16));
                        else if (YM == 1) *reinterpret_cast<uint2 *>(q) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                        else *reinterpret_cast<float4 *>(q) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's ds_writes before its ds_reads
                __builtin_amdgcn_wave_barrier();
                const int nrows = (TM - a0 < EPI_A ? TM - a0 : EPI_A) * 16;
                for (int c = lane; c < nrows * cpr; c += 64) {
                    const int row = c / cpr, cc = c - row * cpr;
                    *reinterpret_cast<uint4 *>(Y8 + (int64_t)(a0 * 16 + row) * p.ldy * ESZ + cc * 16) =
                        *reinterpret_cast<const uint4 *>(T + row * pitch + cc * 16);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // ... and its ds_reads before the next pass's ds_writes
                __builtin_amdgcn_wave_barrier();
            }
        }
        leave_stamps();
        return;
    }
    float *Yf = reinterpret_cast<float *>(p.Y) + (int64_t)s * p.y_sample_stride;
    uint16_t *Yh = reinterpret_cast<uint16_t *>(p.Y) + (int64_t)s * p.y_sample_stride;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int m = mw + a * 16 + fi;
        if (m >= p.M) continue;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            float v[4];
            vals(a, b, v);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nw + b * 16 + fq * 4 + r;
                if (n >= p.N) continue;
                if (YM == 2) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) Yh[(int64_t)pl * p.y_plane_stride + (int64_t)m * p.ldy + n] = plane_of(v[r], pl);
                } else if (YM == 1) Yh[(int64_t)m * p.ldy + n] = f2bf(v[r]);
                else Yf[(int64_t)m * p.ldy + n] = v[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ (3) narrow head
// N <= 16 (the 10-way classifier head): 16 rows x 16 columns per workgroup, K split over its 4 waves; fragments are
// loaded straight to registers (one MFMA per 32 k: nothing to reuse through LDS), every load of a wave's K range is
// requested before the first use.  The four partial tiles are added in wave order through LDS (fixed order: bitwise
// reproducible).  grid = S * ceil(M / 16): the BASELINE head is 256 workgroups, one per CU.
constexpr int kHeadMaxSteps = 16;     // 32-k steps per wave: K <= 4 * 16 * 32 = 2048

__global__ __launch_bounds__(256) void k_head_bf16(const DenseParams p)
{
    __shared__ float red[4][16][17];
    const int ntm = p.ntm;
    const int s = (int)blockIdx.x / ntm, mt = (int)blockIdx.x % ntm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fq = lane >> 4;
    const int nk = (p.K + 31) / 32;
    const int per = (nk + 3) / 4;
    const int k0 = wave * per, k1 = (k0 + per < nk) ? k0 + per : nk;
    int m = mt * 16 + fi;
    m = m < p.M ? m : p.M - 1;
    int n = fi < p.N ? fi : p.N - 1;
    const uint16_t *arow = p.A + (int64_t)s * p.a_sample_stride + (int64_t)m * p.lda;
    const uint16_t *wrow = p.W + (int64_t)s * p.w_sample_stride + (int64_t)n * p.ldw;
    uint4 af[kHeadMaxSteps], bfr[kHeadMaxSteps];
    const int kmax = p.K - 8;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    // three-plane operands (fp32 parity mode): the same pass six times, on plane pairs (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
    const int npass = p.x3 ? 6 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        const uint16_t *ar = arow, *wr = wrow;
        if (p.x3) {
            ar += ((0x001102 >> (4 * pass)) & 3) * p.a_plane_stride;
            wr += ((0x010120 >> (4 * pass)) & 3) * p.w_plane_stride;
        }
#pragma unroll
        for (int i = 0; i < kHeadMaxSteps; ++i) {
            const int kt = k0 + i;
            int ka = kt * 32 + 8 * fq;
            const int kb = ka;
            ka = ka < kmax ? ka : kmax;                 // k >= K: clamped (finite) activations x zero-padded weights
            if (kt < k1) {
                af[i] = *reinterpret_cast<const uint4 *>(ar + ka);
                bfr[i] = *reinterpret_cast<const uint4 *>(wr + kb);
            } else {
                af[i] = make_uint4(0u, 0u, 0u, 0u);
                bfr[i] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
#pragma unroll
        for (int i = 0; i < kHeadMaxSteps; ++i)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bfr[i]), acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][fq * 4 + r][fi] = acc[r];
    __syncthreads();
    // 256 threads = 16 x 16 outputs
    const int row = tid >> 4, col = tid & 15;
    const int mo = mt * 16 + row;
    if (mo < p.M && col < p.N) {
        float v = ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
        if (p.bias) v += p.bias[(int64_t)s * p.bias_sample_stride + col];
        if (p.flags & BNN_FLAG_RELU) v = fmaxf(v, 0.f);
        if (p.flags & BNN_FLAG_Y_BF16) reinterpret_cast<uint16_t *>(p.Y)[(int64_t)s * p.y_sample_stride + (int64_t)mo * p.ldy + col] = f2bf(v);
        else reinterpret_cast<float *>(p.Y)[(int64_t)s * p.y_sample_stride + (int64_t)mo * p.ldy + col] = v;
    }
}

// ------------------------------------------------------------------------------------------------ (4) conv2d, implicit GEMM
// y[s] = conv2d(x[s], w_s, b_s) on drawn weights, groups = 1: M = (image, output pixel), N = O, K = (tap, channel).
// No im2col panel anywhere: a workgroup keeps ITS images resident in LDS -- read from memory once (fp32 NCHW, coalesced
// 16-B loads), rounded to bf16 and laid out [image][pixel][channel], UNPADDED -- and the im2col happens in the ADDRESS
// of the A-fragment read: output pixel (oh, ow), tap (kh, kw), channels c0 .. c0 + 7 are 16 contiguous bytes of input
// pixel (oh sh - ph + kh dh, ow sw - pw + kw dw); a tap that falls into the padding reads any valid pixel and the
// fragment is replaced by zeros (four v_cndmask).  The drawn weights arrive tap-major (bnn_draw_multi, taps = KH KW:
// column t C + c), so a 64-k step is one tap x 64 channels and the B tile streams through an LDS-DMA ring (loader waves
// 4-7, one barrier per step; ST stages, ST - 1 in flight: the steps are short -- 16 or 32 MFMAs -- so the ring is deep).
// 16-B chunk j of a pixel's channel vector sits at position j ^ swz(pixel) (C = 64: (pixel >> 1) & 7, 128-B pixels;
// C % 128 == 0: pixel & 15).  The static LDS block is sized for TWO workgroups per CU: the phases of a workgroup (image
// fill: memory-bound; k loop; output copy: memory-bound) do not overlap each other, those of two workgroups do.
// Output: the tile's (images x O x pixels) block is contiguous in NCHW when the tile spans all O channels (it does:
// BN = O), so it is staged in LDS and leaves as whole 16-B chunks.
// Measured on the way (LeNet shape, 8 x 1024 images): padded images + whole-region zero fill, 3-stage ring, one
// workgroup per CU: 48 us, of which fill 20, k loop 14 (DMA-latency-bound), skeleton 11.
struct ConvParams {
    const float *X;
    int64_t x_sample_stride;    // elements; 0 = input shared by all samples
    const uint16_t *W;          // S x O x ldw bf16, tap-major, zero-padded rows
    int64_t w_sample_stride;
    int64_t ldw;
    const float *bias;          // S x O fp32 or NULL
    int64_t bias_sample_stride;
    float *Y;                   // S x B x O x (OH * OW) fp32
    int64_t y_sample_stride;
    int32_t B, C, H, Wd, O, KH, KW, sh, sw, ph, pw, dh, dw, OH, OW;
    int32_t S, IMG, ntiles;     // images per workgroup, workgroups per sample
    int32_t img_bytes;          // H * W * C * 2
    int32_t flags;
    // Flipout (FLIP instantiation): W holds 2 O rows -- the means, then the stddevs -- and
    //   y[b][o] = conv(x[b], mean)[o] + R[b][o] * conv(x[b] * S[b], stddev)[o]           (conv.py:207-221)
    const float *sgn_in;        // S: B x C of +-1
    const float *sgn_out;       // R: B x O of +-1
    // fp32 parity mode (X3 instantiation, bnn_conv2d_dense_forward_x3): W is THREE bf16 planes (plane stride in elements), the
    // resident images are split into three planes in LDS, and every 64-k block is walked on the six plane pairs of
    // k_dense_bf16's parity mode -- the five small pairs for every block first, then (h, h) over all of K
    int64_t w_plane_stride;
};

constexpr int kConvLds = 78 * 1024;      // two workgroups per CU (O = 64)
constexpr int kConvLdsBig = 136 * 1024;  // one workgroup per CU with a 6-stage ring (O = 128: 16-KB stages, 32 MFMAs per step)

// FLIP: the Flipout estimator in ONE launch -- the tile's columns are [O means | O stddevs] (TN = 2 O / 16), both
// contractions share the A fragment: the second one takes it with the sign bits of S flipped in (a 16-B mask per
// (image, 8-channel chunk), built in LDS next to the images), and R multiplies its accumulator in the epilogue.
template <int TN, int ST, int LDSB, bool FLIP = false, bool X3 = false>
__global__ __launch_bounds__(512, LDSB <= 80 * 1024 ? 2 : 1) void k_conv_bf16(const ConvParams p)
{
    // (FLIP && X3: the Flipout launch of the fp32 parity mode -- the sign mask flips every plane of the A fragment alike)
    constexpr int NWV = 4, TM = 2, WM = 32, BN = 16 * TN;
    constexpr int B_TOTAL = BN / 8, NBP = B_TOTAL / NWV;        // B pieces per loader per stage (TN = 4: 2, TN = 8: 4)
    constexpr int B_STAGE = BN * 128;
    static_assert(B_TOTAL % NWV == 0, "B pieces split evenly over the loaders");
    static_assert(ST >= 2 && ST * B_STAGE < LDSB, "ring");
    __shared__ __attribute__((aligned(16))) char lds[LDSB];
    const int P = p.OH * p.OW, HW = p.H * p.Wd;
    const int CPX = p.C >> 3;                                   // 16-B chunks per pixel
    const int mask_region = FLIP ? p.IMG * CPX * 16 : 0;        // sign masks [image][chunk]
    const int img_region = p.IMG * p.img_bytes;        // one plane of this tile's images
    constexpr int NPL = X3 ? 3 : 1;
    char *masks = lds + NPL * img_region;
    char *b_ring = lds + NPL * img_region + mask_region;
    constexpr int NOUT = FLIP ? BN / 2 : BN;                    // output channels of the tile

    int s, t;
    {
        const int L = (int)blockIdx.x;
        if (p.S % 8 == 0) { const int idx = L >> 3; s = (L & 7) + 8 * (idx / p.ntiles); t = idx % p.ntiles; }
        else { s = L / p.ntiles; t = L % p.ntiles; }
    }
    const int b0 = t * p.IMG;
    const int imgs = (p.B - b0 < p.IMG) ? p.B - b0 : p.IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool swz16 = (p.C >> 3) >= 16;
    auto swz = [&](int pl) { return swz16 ? (pl & 15) : ((pl >> 1) & 7); };
    const int nkb = p.KH * p.KW * (p.C >> 6);          // 64-k blocks: (tap, 64 channels)
    const int nk = nkb * (X3 ? 6 : 1);

    // ---- loaders put the first ST - 1 weight stages in flight before anything else
    const int lw = wave - NWV;
    uint32_t b_off[NBP];
    uint32_t b_lds = 0;
    const char *w_base = reinterpret_cast<const char *>(p.W + (int64_t)s * p.w_sample_stride);
    auto issue = [&](int kt) {
        const int stage = kt % ST;
        int kb = kt;
        const char *wb = w_base;
        if constexpr (X3) {
            int pr = 5;
            if (kt < 5 * nkb) { kb = kt / 5; pr = kt - 5 * kb; } else kb = kt - 5 * nkb;
            wb += ((0x010120 >> (4 * pr)) & 3) * (p.w_plane_stride * 2);      // the W plane of pair pr: h l m h m h
        }
#pragma unroll
        for (int j = 0; j < NBP; ++j)
            dma_piece(wb, b_off[j] + (uint32_t)(kb * 128), b_lds + (uint32_t)(stage * B_STAGE + (lw + NWV * j) * 1024));
    };
    if (wave >= NWV) {
        b_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(b_ring));
        const int prow = lane >> 3, schunk = (lane & 7) ^ prow;
#pragma unroll
        for (int j = 0; j < NBP; ++j) {
            int n = (lw + NWV * j) * 8 + prow;
            const int nrows = FLIP ? 2 * p.O : p.O;
            n = n < nrows ? n : nrows - 1;
            b_off[j] = (uint32_t)((int64_t)n * p.ldw * 2) + 16u * (uint32_t)schunk;
        }
#pragma unroll
        for (int k0 = 0; k0 < ST - 1; ++k0)
            if (k0 < nk) issue(k0);
    }

    // (p.flags, BNN_CONV_DIAG: timing-only runs with wrong outputs -- 1 skips the image fill, 2 the k loop, 4 the output copy)
    // ---- phase 0: this tile's images -> LDS (bf16, channel-last, swizzled on the pixel index inside the image).
    // A thread owns one (channel pair, pixel quad) of the image layout and walks the images: its decomposition -- the only
    // integer divisions of the phase -- is computed once, every image then costs two 16-B loads and four 4-B LDS writes.
    {
        const float *xs = p.X + (int64_t)s * p.x_sample_stride + (int64_t)b0 * p.C * HW;
        const int Q = (HW + 3) >> 2, CH2 = p.C >> 1;
        const int per_img = (p.flags & 1) ? 0 : CH2 * Q;
        if constexpr (FLIP) {
            // sign masks: bit 15 of element j set where S[b][8 chunk + j] < 0
            for (int i = tid; i < imgs * CPX; i += 512) {
                const int il = i / CPX, ch = i - il * CPX;
                const float *sp = p.sgn_in + (int64_t)(b0 + il) * p.C + ch * 8;
                uint32_t w4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    w4[j] = (sp[2 * j] < 0.f ? 0x8000u : 0u) | (sp[2 * j + 1] < 0.f ? 0x80000000u : 0u);
                *reinterpret_cast<uint4 *>(masks + i * 16) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
        }
        const bool vec = (HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(xs) & 15u) == 0);
        const int img_elems = p.C * HW;
        for (int rem = tid; rem < per_img; rem += 512) {
            const int cp = rem / Q, q = rem - cp * Q;
            int dst[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pl = 4 * q + j;                       // pixel index inside the image
                dst[j] = pl < HW ? pl * (p.C * 2) + (((cp >> 2) ^ swz(pl)) << 4) + (cp & 3) * 4 : -1;
            }
            const float *g0 = xs + (int64_t)(2 * cp) * HW + 4 * q;
            // eight images (16 x 16-B loads) in flight per thread
            constexpr int UB = 8;
            for (int il = 0; il < imgs; il += UB) {
                float va[UB][4], vb[UB][4];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int iu = il + u < imgs ? il + u : imgs - 1;       // past the tile: a duplicate load, not written
                    const float *g = g0 + (int64_t)iu * img_elems;
                    if (vec) {
                        const float4 a = *reinterpret_cast<const float4 *>(g), b = *reinterpret_cast<const float4 *>(g + HW);
                        va[u][0] = a.x; va[u][1] = a.y; va[u][2] = a.z; va[u][3] = a.w;
                        vb[u][0] = b.x; vb[u][1] = b.y; vb[u][2] = b.z; vb[u][3] = b.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { va[u][j] = dst[j] >= 0 ? g[j] : 0.f; vb[u][j] = dst[j] >= 0 ? g[HW + j] : 0.f; }
                    }
                }
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (il + u < imgs) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (dst[j] >= 0) {
                                char *q = lds + (il + u) * p.img_bytes + dst[j];
                                if constexpr (X3) {
                                    uint32_t h, m, l;
                                    split_bf16x3(va[u][j], vb[u][j], h, m, l);
                                    *reinterpret_cast<uint32_t *>(q) = h;
                                    *reinterpret_cast<uint32_t *>(q + img_region) = m;
                                    *reinterpret_cast<uint32_t *>(q + 2 * img_region) = l;
                                } else
                                    *reinterpret_cast<uint32_t *>(q) = pack_bf16x2(va[u][j], vb[u][j]);
                            }
                    }
            }
        }
        __syncthreads();
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fi = lane & 15, fq = lane >> 4;

    const int nk_run = (p.flags & 2) ? 0 : nk;
    if (wave >= NWV) {
        // =============================== loader ===============================
        if (nk_run == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int kt = 0; kt < nk_run; ++kt) {
            // this wave's pieces of step kt have landed: at most min(ST - 2, nk - 1 - kt) younger stages may stay in flight
            const int younger = nk - 1 - kt < ST - 2 ? nk - 1 - kt : ST - 2;
            if (younger >= 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * NBP) : "memory");
            else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * NBP) : "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NBP) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NBP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // barrier kt: stage kt complete; stage (kt - 1) % ST is free
            if (kt + ST - 1 < nk) issue(kt + ST - 1);
        }
    } else {
        // =============================== consumer ===============================
        // this lane's two A rows (16-row blocks a = 0, 1): output pixel -> input row / column of tap (0, 0)
        int ih0[TM], iw0[TM], ib[TM], mb[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            int row = wave * WM + a * 16 + fi;
            row = row < imgs * P ? row : 0;             // slack rows: any valid pixel, results never staged
            const int il = row / P, pix = row - il * P;
            const int oh = pix / p.OW, ow = pix - oh * p.OW;
            ih0[a] = oh * p.sh - p.ph;
            iw0[a] = ow * p.sw - p.pw;
            ib[a] = il * p.img_bytes;
            mb[a] = il * CPX * 16;
        }
        const int fb0 = fi * 128 + (((0 + fq) ^ (fi & 7)) << 4), fb1 = fi * 128 + (((4 + fq) ^ (fi & 7)) << 4);
        const int CB = p.C >> 6;
        const int pixb = p.C * 2;
        // Fragment registers are double-buffered over the two 32-k halves of a step and barrier t + 1 sits in the MIDDLE of
        // step t, as in k_dense_bf16: [reads (t, h1) || MFMAs (t, h0)] [reads done -> barrier t + 1] [reads (t + 1, h0) || MFMAs
        // (t, h1)] -- every group of fragment reads is in flight under the previous group's MFMAs.  (Round 2 read the 2 + TN
        // fragments of a half and then multiplied: at O = 128 a step was 0.61 us for 0.24 us of MFMA issue.)
        uint4 fa[2][TM], fbr[2][TN], fmk[2][FLIP ? TM : 1];
        // (tap row, tap column, 64-channel block) of the step whose h0 half is read next: advanced by rd(.., h0), no divisions
        // in the loop; the step's fragment addresses are kept for its h1 half
        int s_kh = 0, s_kw = 0, s_cb = 0, s_stage = 0, cur_cb = 0;
        int s_pr = 0, s_kt = 0;                             // X3: plane pair of the step, step index
        int aoff[TM];
        bool okk[TM];
        const char *Bs = b_ring;
        auto rd = [&](auto buf_c, auto h_c) {
            constexpr int buf = decltype(buf_c)::value, h = decltype(h_c)::value;
            if constexpr (h == 0) {
                Bs = b_ring + s_stage * B_STAGE;
                cur_cb = s_cb;
                int aplane = 0;
                if constexpr (X3) aplane = ((0x001102 >> (4 * s_pr)) & 3) * img_region;        // the A plane of pair pr: l h m m h h
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const int ih = ih0[a] + s_kh * p.dh, iw = iw0[a] + s_kw * p.dw;
                    okk[a] = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.Wd;
                    const int pl = okk[a] ? ih * p.Wd + iw : 0;
                    aoff[a] = aplane + ib[a] + pl * pixb + (((s_cb * 8 + fq) ^ swz(pl)) << 4);
                }
                s_stage = s_stage + 1 == ST ? 0 : s_stage + 1;
                bool next_block = true;
                if constexpr (X3) {
                    // sweep 1 (steps < 5 nkb): pairs 0..4 of a block, then the next block; sweep 2: pair 5 of every block
                    ++s_kt;
                    if (s_kt < 5 * nkb) { next_block = s_pr == 4; s_pr = next_block ? 0 : s_pr + 1; }
                    else if (s_kt == 5 * nkb) { s_pr = 5; next_block = false; s_cb = s_kw = s_kh = 0; }
                }
                if (next_block) { if (++s_cb == CB) { s_cb = 0; if (++s_kw == p.KW) { s_kw = 0; ++s_kh; } } }
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) fbr[buf][b] = *reinterpret_cast<const uint4 *>(Bs + (h ? fb1 : fb0) + b * 2048);
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                // chunk (cb 8 + 4 h + fq) ^ swz: h flips bit 2 of the chunk index, i.e. 64 bytes of the address
                const uint4 v = *reinterpret_cast<const uint4 *>(lds + (h ? (aoff[a] ^ 64) : aoff[a]));
                fa[buf][a] = okk[a] ? v : make_uint4(0u, 0u, 0u, 0u);
                if constexpr (FLIP) fmk[buf][a] = *reinterpret_cast<const uint4 *>(masks + mb[a] + (cur_cb * 8 + 4 * h + fq) * 16);
            }
        };
        auto mm = [&](auto buf_c) {
            constexpr int buf = decltype(buf_c)::value;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                uint4 afs = fa[buf][a];
                if constexpr (FLIP) {
                    const uint4 m = fmk[buf][a];
                    afs = make_uint4(afs.x ^ m.x, afs.y ^ m.y, afs.z ^ m.z, afs.w ^ m.w);
                }
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const uint4 av = (FLIP && b >= TN / 2) ? afs : fa[buf][a];
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av),
                                                                        __builtin_bit_cast(bf16x8, fbr[buf][b]), acc[a][b], 0, 0, 0);
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        if (nk_run > 0) {
            __builtin_amdgcn_s_barrier();                                   // barrier 0: stage 0 has landed
            asm volatile("" ::: "memory");
            rd(I0{}, I0{});
            for (int kt = 0; kt + 1 < nk_run; ++kt) {
                rd(I1{}, I1{});
                mm(I0{});
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's reads of stage kt have returned
                __builtin_amdgcn_s_barrier();                               // barrier kt + 1
                asm volatile("" ::: "memory");
                rd(I0{}, I0{});
                mm(I1{});
                __builtin_amdgcn_sched_barrier(0);
            }
            rd(I1{}, I1{});
            mm(I0{});
            mm(I1{});
        }
    }

    // ---- epilogue: stage the tile as [image][channel][pixel] (its NCHW block is contiguous), then whole 16-B chunks
    __syncthreads();                                    // every wave is done with the images and the weight ring
    float *T = reinterpret_cast<float *>(lds);
    if (wave < NWV) {
        const float *bias = p.bias ? p.bias + (int64_t)s * p.bias_sample_stride : nullptr;
#pragma unroll
        for (int b = 0; b < NOUT / 16; ++b) {
            const int o = b * 16 + fi;
            const float bv = (bias && o < p.O) ? bias[o] : 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int row0 = wave * WM + a * 16 + fq * 4;
                int il = row0 / P, pix = row0 - il * P;         // the lane's 4 rows are 4 consecutive pixels
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (row0 + r < imgs * P && o < p.O) {
                        float v = acc[a][b][r] + bv;
                        if constexpr (FLIP) v = fmaf(p.sgn_out[(int64_t)(b0 + il) * p.O + o], acc[a][b + TN / 2][r], v);
                        T[(il * p.O + o) * P + pix] = v;
                    }
                    if (++pix == P) { pix = 0; ++il; }
                }
            }
        }
    }
    __syncthreads();
    {
        float *Yt = p.Y + (int64_t)s * p.y_sample_stride + (int64_t)b0 * p.O * P;
        const int n = (p.flags & 4) ? 0 : imgs * p.O * P;
        if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(Yt) & 15u) == 0) {
            for (int i = tid; i < (n >> 2); i += 512) reinterpret_cast<uint4 *>(Yt)[i] = reinterpret_cast<const uint4 *>(T)[i];
        } else {
            for (int i = tid; i < n; i += 512) Yt[i] = T[i];
        }
    }
}

static inline bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

}  // namespace bnn

using namespace bnn;

namespace bnn {
// fp32 -> three bf16 planes (h, m, l), 8 columns per thread
__global__ __launch_bounds__(256) void k_split_bf16x3(const float *__restrict__ x, int gpr, int64_t ldx, uint16_t *__restrict__ out,
                                                     int64_t ld_out, int64_t plane_stride, int items)
{
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= items) return;
    const int row = i / gpr, c0 = (i - row * gpr) * 8;
    const float4 a = *reinterpret_cast<const float4 *>(x + (int64_t)row * ldx + c0);
    const float4 b = *reinterpret_cast<const float4 *>(x + (int64_t)row * ldx + c0 + 4);
    uint4 h, m, l;
    split_bf16x3(a.x, a.y, h.x, m.x, l.x);
    split_bf16x3(a.z, a.w, h.y, m.y, l.y);
    split_bf16x3(b.x, b.y, h.z, m.z, l.z);
    split_bf16x3(b.z, b.w, h.w, m.w, l.w);
    uint16_t *o = out + (int64_t)row * ld_out + c0;
    *reinterpret_cast<uint4 *>(o) = h;
    *reinterpret_cast<uint4 *>(o + plane_stride) = m;
    *reinterpret_cast<uint4 *>(o + 2 * plane_stride) = l;
}

// bf16 (batch x rows x cols, row pitch ld_in) -> its transpose (batch x cols x ld_out), zeros in columns rows .. ld_out - 1: the
// drawn weights (S x N x ldw) turned into the operand of the INPUT gradient gx[s] = gy[s] . w_s, which the dense kernel
// contracts over n.  A workgroup moves a 64 x 64 tile through LDS ([c][r], pitch 72 elements: 16-B chunks on the way out).
__global__ __launch_bounds__(256) void k_transpose_bf16(const uint16_t *__restrict__ in, int64_t in_batch_stride, int64_t ld_in,
                                                       uint16_t *__restrict__ out, int64_t out_batch_stride, int64_t ld_out,
                                                       int rows, int cols)
{
    constexpr int PITCH = 72;
    __shared__ __attribute__((aligned(16))) uint16_t T[64 * PITCH];
    const int r0 = (int)blockIdx.x * 64, c0 = (int)blockIdx.y * 64;
    const uint16_t *src = in + (int64_t)blockIdx.z * in_batch_stride;
    uint16_t *dst = out + (int64_t)blockIdx.z * out_batch_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (int)threadIdx.x + 256 * j;
        const int r = q >> 3, cc = (q & 7) * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r0 + r < rows && c0 + cc < cols) v = *reinterpret_cast<const uint4 *>(src + (int64_t)(r0 + r) * ld_in + c0 + cc);
        const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T[(cc + 2 * i) * PITCH + r] = (uint16_t)(e[i] & 0xFFFFu);
            T[(cc + 2 * i + 1) * PITCH + r] = (uint16_t)(e[i] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (int)threadIdx.x + 256 * j;
        const int c = q >> 3, rc = (q & 7) * 8;
        if (c0 + c < cols && r0 + rc < ld_out)
            *reinterpret_cast<uint4 *>(dst + (int64_t)(c0 + c) * ld_out + r0 + rc) = *reinterpret_cast<const uint4 *>(&T[c * PITCH + rc]);
    }
}
}  // namespace bnn

// the fused head's operands (dense_launch's `head`, NULL = the plain layer)
struct HeadArgs {
    const void *wh;
    int64_t wh_plane_stride;    // 0: plain bf16 operands
    int64_t wh_sample_stride, ldwh;
    const float *bh;
    int64_t bh_sample_stride;
    float *partials;
    int64_t nh;
};

// tile of a dense launch: 0 (256 x 80), 1 (128 x 160), 2 (256 x 128), 3 (64 x 160), 4 (32 x 160) -- see dense_launch
static int dense_pick_tile(int64_t M, int64_t N, int nsamples)
{
    static const int force_tile = [] { const char *e = getenv("BNN_DENSE_TILE"); return e ? atoi(e) : -1; }();
    int tile = (N % 80 == 0 || N < 128) ? 1 : 2;
    if (N <= 80) tile = 0;
    if (force_tile >= 0 && force_tile <= 4) tile = force_tile;
    if (tile == 1 && force_tile < 0) {
        const int64_t cols = (N + 159) / 160;
        if (((M + 127) / 128) * cols * nsamples < 128 && M > 64) tile = 3;
        if (tile == 3 && ((M + 63) / 64) * cols * nsamples < 128 && M > 32) tile = 4;
    }
    return tile;
}

static int dense_launch(const char *who, const void *x, int64_t x_plane_stride, int64_t x_sample_stride, int64_t ldx,
                        const void *w, int64_t w_plane_stride, int64_t w_sample_stride, int64_t ldw,
                        const float *b, int64_t b_sample_stride,
                        void *y, int64_t y_plane_stride, int64_t y_sample_stride, int64_t ldy,
                        int64_t M, int64_t N, int64_t K, int nsamples, int flags, bool x3, void *stream, const HeadArgs *head = nullptr)
{
    if (M == 0 && N >= 1 && K >= 1 && nsamples >= 1) return BNN_OK;
    if (head) { y = head->partials; ldy = N; }           // (no output tensor: the checks below see the partials)
    if (!x || !w || !y) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (M < 0 || N < 1 || K < 1 || nsamples < 1 || ldx < K || ldy < N) { set_error("%s: bad extent", who); return BNN_E_SHAPE; }
    if (M > 0x7FFFFFFF || N > 0x7FFFFFFF || K > 0x7FFFFFFF) { set_error("%s: extent too large", who); return BNN_E_RANGE; }
    if (flags & ~(BNN_FLAG_RELU | BNN_FLAG_Y_BF16)) { set_error("%s: unknown flags", who); return BNN_E_UNSUPPORTED; }
    const int64_t kp = (K + 63) / 64 * 64;
    if (K % 8 != 0 || ldx % 8 != 0 || x_sample_stride % 8 != 0 || ldw % 8 != 0 || w_sample_stride % 8 != 0 || ldw < kp || !al16(x) || !al16(w)) {
        set_error("%s: needs K %% 8 == 0, 16-B aligned bf16 rows, and weight rows zero-padded to ldw >= roundup(K, 64)", who);
        return BNN_E_UNSUPPORTED;
    }
    if (M * ldx * 2 >= ((int64_t)1 << 32) || N * ldw * 2 >= ((int64_t)1 << 32)) { set_error("%s: one sample's operand exceeds 4 GiB", who); return BNN_E_RANGE; }
    const bool ybf = (flags & BNN_FLAG_Y_BF16) != 0;
    if (reinterpret_cast<uintptr_t>(y) & (ybf ? 1u : 3u)) { set_error("%s: misaligned output", who); return BNN_E_ALIGN; }
    DenseParams p{};
    p.A = reinterpret_cast<const uint16_t *>(x); p.a_sample_stride = x_sample_stride; p.lda = ldx;
    p.W = reinterpret_cast<const uint16_t *>(w); p.w_sample_stride = w_sample_stride; p.ldw = ldw;
    p.bias = b; p.bias_sample_stride = b_sample_stride;
    p.Y = y; p.y_sample_stride = y_sample_stride; p.ldy = ldy;
    p.M = (int32_t)M; p.N = (int32_t)N; p.K = (int32_t)K; p.S = nsamples; p.flags = flags;
    if (x3) {
        if (x_plane_stride % 8 != 0 || w_plane_stride % 8 != 0 || (ybf && y_plane_stride % 8 != 0) || x_plane_stride < M * ldx || w_plane_stride < N * ldw) {
            set_error("%s: bad plane stride", who);
            return BNN_E_SHAPE;
        }
        if (N <= 16 && (K > 4 * kHeadMaxSteps * 32 || ybf)) { set_error("%s: a narrow layer (N <= 16) on three-plane operands needs K <= %d and fp32 outputs", who, 4 * kHeadMaxSteps * 32); return BNN_E_UNSUPPORTED; }
        p.x3 = 1;
        p.a_plane_stride = x_plane_stride; p.w_plane_stride = w_plane_stride; p.y_plane_stride = y_plane_stride;
    }
    if (head) {
        if (N <= 16 || (flags & BNN_FLAG_Y_BF16) == 0) { set_error("%s: the fused head follows a hidden layer wider than 16", who); return BNN_E_UNSUPPORTED; }
        if (x3 && (head->wh_plane_stride % 8 != 0 || head->wh_plane_stride < head->nh * head->ldwh)) { set_error("%s: bad plane stride of the head's weights", who); return BNN_E_SHAPE; }
        if (x3) {
            const int t_ = dense_pick_tile(M, N, nsamples);
            if (t_ == 0 || t_ == 2) { set_error("%s: the three-plane fused head runs on the 160-column tiles (N %% 80 == 0 or N < 128, N > 80)", who); return BNN_E_UNSUPPORTED; }
        }
        if (!head->wh || head->nh < 1 || head->nh > 16) { set_error("%s: head of 1 .. 16 outputs", who); return BNN_E_SHAPE; }
        if (N % 8 != 0 || head->ldwh % 8 != 0 || head->ldwh < N || head->wh_sample_stride % 8 != 0 || !al16(head->wh) ||
            (reinterpret_cast<uintptr_t>(head->partials) & 3u)) {
            set_error("%s: the head's weights need N %% 8 == 0 and 16-B aligned rows zero-padded to ldwh >= N", who);
            return BNN_E_UNSUPPORTED;
        }
        p.Wh = reinterpret_cast<const uint16_t *>(head->wh); p.wh_sample_stride = head->wh_sample_stride; p.ldwh = head->ldwh;
        p.bias_h = head->bh; p.bias_h_sample_stride = head->bh_sample_stride;
        p.P = head->partials; p.Nh = (int32_t)head->nh;
        p.wh_plane_stride = head->wh_plane_stride;
    }
    hipStream_t st = (hipStream_t)stream;
    if (N <= 16 && K <= 4 * kHeadMaxSteps * 32) {
        p.ntm = (int32_t)((M + 15) / 16);
        p.ntn = 1;
        hipLaunchKernelGGL(k_head_bf16, dim3((unsigned)((int64_t)p.ntm * nsamples)), dim3(256), 0, st, p);
        return check_launch(who);
    }
    static const bool no_xcd = [] { const char *e = getenv("BNN_DENSE_XCD"); return e && e[0] == '0'; }();
    if (no_xcd) p.flags |= kDenseNoXcdMap;
    static const bool lprio = [] { const char *e = getenv("BNN_DENSE_LOADER_PRIO"); return e && e[0] == '1'; }();
    if (lprio) p.flags |= kDenseLoaderPrio;
    // tile: the BASELINE-shaped layers (N % 80 == 0: 1200 = 7.5 x 160) take 128 x 160 with a 4-stage ring -- 36 KiB per
    // 64-k step instead of 256 x 80's 42 for the same MFMAs, 4 x 8 x 8 = 256 workgroups; wide layers 256 x 128.
    // BNN_DENSE_TILE = 0 (256 x 80), 1 (128 x 160), 2 (256 x 128), 3 (64 x 160), 4 (32 x 160) forces one for A/B runs.
    const int tile = dense_pick_tile(M, N, nsamples);
    // few samples (a rank of a sharded MC job runs 8 / G of them): 64- and 32-row versions of the 128 x 160 tile, the largest
    // that still gives >= 128 workgroups -- a dense launch over ONE sample (32 workgroups) took as long as over eight.
    // Measured at the BASELINE layers, one stream / three steps in flight, us per step: 4 samples 54.2 / 28.3 (128 rows),
    // 51.0 / 30.9 (64), 58.7 / 37.4 (32); 2 samples 48.8 / 22.3, 44.3 / 20.8, 41.9 / 24.0; 1 sample 46.2 / 19.3, 41.3 / 17.8,
    // 39.0 / 17.9: with the chip already full of other steps' work the bigger tile wins (fewer re-reads of the weights).
    const int bm = tile == 1 ? 128 : tile == 3 ? 64 : tile == 4 ? 32 : 256;
    const int bn = tile == 0 ? 80 : tile == 2 ? 128 : 160;
    p.ntm = (int32_t)((M + bm - 1) / bm);
    p.ntn = (int32_t)((N + bn - 1) / bn);
    const int64_t grid = (int64_t)p.ntm * p.ntn * nsamples;
    if (grid > 0x7FFFFFFF) { set_error("%s: grid too large", who); return BNN_E_RANGE; }
    static const int diag = [] { const char *e = getenv("BNN_DENSE_DIAG"); return e ? atoi(e) : 0; }();
    if (diag >= 6 && diag <= 9 && !head) {
        static const uint64_t stamps = [] { const char *e = getenv("BNN_DENSE_STAMPS"); return e ? strtoull(e, nullptr, 0) : 0ull; }();
        p.P = reinterpret_cast<float *>(stamps);
    }
    const bool relu = (flags & BNN_FLAG_RELU) != 0;
    const dim3 g((unsigned)grid), blk(512);
#define BNN_DENSE_LAUNCH(TM_, TN_, NWM_, NWN_, ST_, RELU_) \
    do { \
        if (head) hipLaunchKernelGGL((k_dense_bf16<TM_, TN_, NWM_, NWN_, ST_, 3, RELU_>), g, blk, 0, st, p); \
        else if (x3 && ybf) hipLaunchKernelGGL((k_dense_bf16<TM_, TN_, NWM_, NWN_, ST_, 2, RELU_>), g, blk, 0, st, p); \
        else if (!ybf) hipLaunchKernelGGL((k_dense_bf16<TM_, TN_, NWM_, NWN_, ST_, 0, RELU_>), g, blk, 0, st, p); \
        else if (diag == 1) hipLaunchKernelGGL((k_dense_bf16<TM_, TN_, NWM_, NWN_, ST_, 1, RELU_, 1>), g, blk, 0, st, p); \
        else if (diag == 2) hipLaunchKernelGGL((k_dense_bf16<TM_, TN_, NWM_, NWN_, ST_, 1, RELU_, 2>), g, blk, 0, st, p); \
        else if (diag == 3) hipLaunchKernelGGL((k_dense_bf16<TM_, TN_, NWM_, NWN_, ST_, 1, RELU_, 3>), g, blk, 0, st, p); \
        else if (diag == 4) hipLaunchKernelGGL((k_dense_bf16<TM_, TN_, NWM_, NWN_, ST_, 1, RELU_, 4>), g, blk, 0, st, p); \
        else if (diag == 5) hipLaunchKernelGGL((k_dense_bf16<TM_, TN_, NWM_, NWN_, ST_, 1, RELU_, 5>), g, blk, 0, st, p); \
        else if (diag == 6) hipLaunchKernelGGL((k_dense_bf16<TM_, TN_, NWM_, NWN_, ST_, 1, RELU_, 6>), g, blk, 0, st, p); \
        else if (diag == 7) hipLaunchKernelGGL((k_dense_bf16<TM_, TN_, NWM_, NWN_, ST_, 1, RELU_, 7>), g, blk, 0, st, p); \
        else if (diag == 8) hipLaunchKernelGGL((k_dense_bf16<TM_, TN_, NWM_, NWN_, ST_, 1, RELU_, 8>), g, blk, 0, st, p); \
        else if (diag == 9) hipLaunchKernelGGL((k_dense_bf16<TM_, TN_, NWM_, NWN_, ST_, 1, RELU_, 9>), g, blk, 0, st, p); \
        else hipLaunchKernelGGL((k_dense_bf16<TM_, TN_, NWM_, NWN_, ST_, 1, RELU_>), g, blk, 0, st, p); \
    } while (0)
#define BNN_DENSE_PICK(TM_, TN_, NWM_, NWN_, ST_) \
    do { \
        if (relu) BNN_DENSE_LAUNCH(TM_, TN_, NWM_, NWN_, ST_, true); else BNN_DENSE_LAUNCH(TM_, TN_, NWM_, NWN_, ST_, false); \
    } while (0)
    if (tile == 0) BNN_DENSE_PICK(4, 5, 4, 1, 3);
    else if (tile == 1) BNN_DENSE_PICK(4, 5, 2, 2, 4);
    else if (tile == 3) BNN_DENSE_PICK(2, 5, 2, 2, 4);
    else if (tile == 4) BNN_DENSE_PICK(1, 5, 2, 2, 4);
    else BNN_DENSE_PICK(4, 8, 4, 1, 3);
#undef BNN_DENSE_PICK
#undef BNN_DENSE_LAUNCH
    return check_launch(who);
}


extern "C" {

int bnn_draw_multi(const bnn_draw_tensor_t *tensors, int ntensors, int nsamples,
                   const bnn_kl_tensor_t *kl_tensors, int kl_ntensors, void *kl_workspace, void *stream)
{
    const char *who = "bnn_draw_multi";
    if (!tensors) { set_error("%s: NULL tensors", who); return BNN_E_NULL; }
    if (ntensors < 1 || ntensors > kDrawMaxTensors) { set_error("%s: 1 .. %d tensors per call", who, kDrawMaxTensors); return BNN_E_RANGE; }
    if (nsamples < 1) { set_error("%s: nsamples < 1", who); return BNN_E_SHAPE; }
    DrawLaunch L{};
    L.ntensors = ntensors;
    L.nsamples = nsamples;
    int64_t items = 0;
    for (int i = 0; i < ntensors; ++i) {
        const bnn_draw_tensor_t &t = tensors[i];
        if (!t.mu || !t.rho || !t.out) { set_error("%s: tensor %d: NULL pointer", who, i); return BNN_E_NULL; }
        if (t.rows < 1 || t.cols < 1 || t.ld < t.cols || t.rows > 0x7FFFFFFF || t.ld > 0x7FFFFFFF) { set_error("%s: tensor %d: bad extent", who, i); return BNN_E_SHAPE; }
        if (t.out_dtype != BNN_F32 && t.out_dtype != BNN_BF16 && t.out_dtype != BNN_BF16X3) { set_error("%s: tensor %d: unknown dtype", who, i); return BNN_E_DTYPE; }
        // a Philox block is 4 consecutive elements of the flat tensor and a work item 8 columns of one row
        if (t.rows > 1 && t.cols % 4 != 0) { set_error("%s: tensor %d: cols %% 4 != 0 (use bnn_sample_affine_philox)", who, i); return BNN_E_UNSUPPORTED; }
        if (t.ld % 8 != 0 && t.rows > 1) { set_error("%s: tensor %d: ld %% 8 != 0", who, i); return BNN_E_UNSUPPORTED; }
        if (t.out_dtype != BNN_F32 && (!al16(t.out) || t.out_sample_stride % 8 != 0 || t.ld % 8 != 0)) { set_error("%s: tensor %d: bf16 output needs 16-B aligned rows", who, i); return BNN_E_ALIGN; }
        if ((reinterpret_cast<uintptr_t>(t.mu) | reinterpret_cast<uintptr_t>(t.rho) | reinterpret_cast<uintptr_t>(t.out)) & 3u) { set_error("%s: tensor %d: misaligned pointer", who, i); return BNN_E_ALIGN; }
        const int rc = t.kind == 0 ? check_rng(&t.rng, nsamples) : BNN_OK;
        if (rc) { set_error("%s: tensor %d: bad rng", who, i); return rc; }
        DrawTensorDev &d = L.t[i];
        d.mu = t.mu; d.rho = t.rho; d.out = t.out; d.out_sample_stride = t.out_sample_stride;
        d.rows = (int32_t)t.rows; d.cols = (int32_t)t.cols; d.ld = (int32_t)t.ld; d.bf16 = t.out_dtype == BNN_BF16 ? 1 : t.out_dtype == BNN_BF16X3 ? 2 : 0;
        d.perm_taps = t.taps > 1 ? t.taps : 1;
        d.kind = t.kind;
        if (t.kind < 0 || t.kind > 3) { set_error("%s: tensor %d: kind must be 0 (draw), 1 (mean), 2 (stddev) or 3 (as it is, once)", who, i); return BNN_E_RANGE; }
        if (t.taps > 1 && (t.cols % t.taps != 0 || t.out_dtype == BNN_F32)) { set_error("%s: tensor %d: taps must divide cols (bf16 or three-plane output)", who, i); return BNN_E_SHAPE; }
        d.first_item = (int32_t)items;
        d.rng = make_rng(&t.rng);
        const int64_t nit = t.rows * ((t.ld + 7) / 8);
        d.spread = (nit < kDrawSpreadBelow && nsamples > 1 && nit * nsamples <= 0x7FFFFFFF) ? 1 : 0;
        d.flat = (!d.spread && t.out_dtype != BNN_F32 && t.kind == 0 && t.taps <= 1 && t.cols % 8 == 0 && t.ld % 8 == 0 && al16(t.mu) && al16(t.rho)) ? 1 : 0;
        d.tapg = (t.taps > 1 && t.taps <= 256 && t.out_dtype != BNN_F32 && t.rows >= 1 && (t.cols / t.taps) % 8 == 0 && t.ld % 8 == 0 &&
                  al16(t.mu) && (t.kind == 1 || t.kind == 3 || al16(t.rho))) ? 1 : 0;
        static const bool no_tapg = [] { const char *e = getenv("BNN_DRAW_TAPG"); return e && e[0] == '0'; }();
        if (no_tapg) d.tapg = 0;
        if (d.tapg) {
            d.spread = 0;
            const int64_t ng = 2048 / (8 * t.taps), NG = t.rows * (t.cols / t.taps / 8);
            items += (NG + ng - 1) / ng * 256;                              // one workgroup per ng groups
        } else if (d.flat) {
            items += (t.rows * (t.cols / 8) + 255) / 256 * 256;             // the valid groups, flat ...
            d.pad_first = (int32_t)items;
            items += (t.rows * ((t.ld - t.cols) / 8) + 255) / 256 * 256;   // ... and the zero padding's
        } else
        items += ((d.spread ? nit * nsamples : nit) + 255) / 256 * 256;     // a workgroup works on one tensor
        if (items > 0x7FFFFF00) { set_error("%s: too many elements for one call", who); return BNN_E_RANGE; }
    }
    L.total_items = (int32_t)items;
    L.draw_blocks = (int32_t)((items + 255) / 256);
    int64_t grid = L.draw_blocks;
    if (kl_tensors && kl_ntensors > 0) {
        if (!kl_plan_piggy(kl_tensors, kl_ntensors, kl_workspace, L.kl)) {
            set_error("%s: the KL first pass cannot ride along (more than %d tensors, bad prior or no workspace): launch bnn_kl_forward_partial", who, kKlPiggyMax);
            return BNN_E_UNSUPPORTED;
        }
        // a KL tensor that IS one of this launch's flat draw tensors (same mu / rho, all of it): its partial sums come from the
        // draw items, which hold the posterior in registers -- no workgroups of its own, no second read of mu and rho (round 2:
        // FETCH_SIZE of the launch 36.4 MB for 19.2 MB of posterior)
        static const bool kl_items = [] { const char *e = getenv("BNN_DRAW_KL_ITEMS"); return !(e && e[0] == '0'); }();
        int32_t launched = 0;
        for (int j = 0; j < kl_ntensors; ++j) {
            const int32_t nb = L.kl.pg_first[j + 1] - L.kl.pg_first[j];     // (pg_first still = first_block here)
            bool in_items = false;
            for (int i = 0; i < ntensors && kl_items && !in_items; ++i) {
                DrawTensorDev &d = L.t[i];
                if (d.flat && !d.kl_on && d.mu == kl_tensors[j].mu && d.rho == kl_tensors[j].rho &&
                    kl_tensors[j].n == (int64_t)d.rows * d.cols) {
                    d.kl_on = 1; d.kl_first = L.kl.t[j].first_block;
                    d.kl_prior_mu = kl_tensors[j].prior_mu; d.kl_prior_sigma = kl_tensors[j].prior_sigma;
                    in_items = true;
                }
            }
            L.kl.pg_first[j] = launched;
            if (!in_items) launched += nb;
        }
        L.kl.pg_first[kl_ntensors] = launched;
        L.kl.nblocks = launched;
        grid += L.kl.nblocks;
    }
    // fewer than ~4 workgroups per CU: split the samples over gridDim.y too
    int split = 1;
    while (split < nsamples && (int64_t)L.draw_blocks * split < 1024) split *= 2;
    if (split > nsamples) split = nsamples;
    // BNN_DRAW_SPLIT forces the split for A/B runs (the BASELINE launch: 26.1 / 24.7 / 27.6 / 37.2 us with 1 / 2 / 4 / 8; in the
    // pipelined bench 156 / 153 / 143 k MC-samples/s with 1 / 2 / 4)
    static const int force_split = [] { const char *e = getenv("BNN_DRAW_SPLIT"); return e ? atoi(e) : 0; }();
    if (force_split >= 1 && force_split <= nsamples) split = force_split;
    static const int unroll = [] { const char *e = getenv("BNN_DRAW_UNROLL"); return e ? atoi(e) : 1; }();
    if (unroll == 2) hipLaunchKernelGGL(k_draw_multi<2>, dim3((unsigned)grid, (unsigned)split), dim3(256), 0, (hipStream_t)stream, L);
    else if (unroll == 4) hipLaunchKernelGGL(k_draw_multi<4>, dim3((unsigned)grid, (unsigned)split), dim3(256), 0, (hipStream_t)stream, L);
    else if (unroll == 8) hipLaunchKernelGGL(k_draw_multi<8>, dim3((unsigned)grid, (unsigned)split), dim3(256), 0, (hipStream_t)stream, L);
    else
    hipLaunchKernelGGL(k_draw_multi<1>, dim3((unsigned)grid, (unsigned)split), dim3(256), 0, (hipStream_t)stream, L);
    return check_launch(who);
}

int bnn_dense_forward(const void *x, int64_t x_sample_stride, int64_t ldx,
                      const void *w, int64_t w_sample_stride, int64_t ldw,
                      const float *b, int64_t b_sample_stride,
                      void *y, int64_t y_sample_stride, int64_t ldy,
                      int64_t M, int64_t N, int64_t K, int nsamples, int flags, void *stream)
{
    return dense_launch("bnn_dense_forward", x, 0, x_sample_stride, ldx, w, 0, w_sample_stride, ldw, b, b_sample_stride,
                        y, 0, y_sample_stride, ldy, M, N, K, nsamples, flags, false, stream);
}

int bnn_dense_head_parts(int64_t M, int64_t N, int nsamples)
{
    if (M < 1 || N < 17 || nsamples < 1) return 0;
    const int tile = dense_pick_tile(M, N, nsamples);
    const int bn = tile == 0 ? 80 : tile == 2 ? 128 : 160;
    const int nwn = (tile == 0 || tile == 2) ? 1 : 2;
    return (int)((N + bn - 1) / bn) * nwn;
}

int bnn_dense_forward_head(const void *x, int64_t x_sample_stride, int64_t ldx,
                           const void *w, int64_t w_sample_stride, int64_t ldw,
                           const float *b, int64_t b_sample_stride,
                           const void *w_head, int64_t wh_sample_stride, int64_t ldwh,
                           const float *b_head, int64_t bh_sample_stride, int64_t n_head,
                           float *partials, int64_t M, int64_t N, int64_t K, int nsamples, int flags, void *stream)
{
    HeadArgs h{w_head, 0, wh_sample_stride, ldwh, b_head, bh_sample_stride, partials, n_head};
    return dense_launch("bnn_dense_forward_head", x, 0, x_sample_stride, ldx, w, 0, w_sample_stride, ldw, b, b_sample_stride,
                        nullptr, 0, 0, N, M, N, K, nsamples, flags | BNN_FLAG_Y_BF16, false, stream, &h);
}

int bnn_dense_forward_x3_head(const void *x, int64_t x_plane_stride, int64_t x_sample_stride, int64_t ldx,
                              const void *w, int64_t w_plane_stride, int64_t w_sample_stride, int64_t ldw,
                              const float *b, int64_t b_sample_stride,
                              const void *w_head, int64_t wh_plane_stride, int64_t wh_sample_stride, int64_t ldwh,
                              const float *b_head, int64_t bh_sample_stride, int64_t n_head,
                              float *partials, int64_t M, int64_t N, int64_t K, int nsamples, int flags, void *stream)
{
    if (wh_plane_stride <= 0) { set_error("bnn_dense_forward_x3_head: plane stride of the head's weights"); return BNN_E_SHAPE; }
    HeadArgs h{w_head, wh_plane_stride, wh_sample_stride, ldwh, b_head, bh_sample_stride, partials, n_head};
    return dense_launch("bnn_dense_forward_x3_head", x, x_plane_stride, x_sample_stride, ldx, w, w_plane_stride, w_sample_stride, ldw,
                        b, b_sample_stride, nullptr, 0, 0, N, M, N, K, nsamples, flags | BNN_FLAG_Y_BF16, true, stream, &h);
}

int bnn_dense_forward_x3(const void *x, int64_t x_plane_stride, int64_t x_sample_stride, int64_t ldx,
                         const void *w, int64_t w_plane_stride, int64_t w_sample_stride, int64_t ldw,
                         const float *b, int64_t b_sample_stride,
                         void *y, int64_t y_plane_stride, int64_t y_sample_stride, int64_t ldy,
                         int64_t M, int64_t N, int64_t K, int nsamples, int flags, void *stream)
{
    return dense_launch("bnn_dense_forward_x3", x, x_plane_stride, x_sample_stride, ldx, w, w_plane_stride, w_sample_stride, ldw,
                        b, b_sample_stride, y, y_plane_stride, y_sample_stride, ldy, M, N, K, nsamples, flags, true, stream);
}

int bnn_split_bf16x3(const float *x, int64_t rows, int64_t cols, int64_t ldx, void *out, int64_t ld_out, int64_t plane_stride, void *stream)
{
    const char *who = "bnn_split_bf16x3";
    if (rows == 0 && cols >= 1) return BNN_OK;
    if (!x || !out) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (rows < 0 || cols < 1 || ldx < cols || ld_out < cols) { set_error("%s: bad extent", who); return BNN_E_SHAPE; }
    if (cols % 8 != 0 || ldx % 4 != 0 || ld_out % 8 != 0 || plane_stride % 8 != 0 || !al16(x) || !al16(out)) {
        set_error("%s: needs cols %% 8 == 0 and 16-B aligned rows on both sides", who);
        return BNN_E_UNSUPPORTED;
    }
    const int64_t items = rows * (cols / 8);
    if (items > 0x7FFFFFFF) { set_error("%s: too many elements", who); return BNN_E_RANGE; }
    hipLaunchKernelGGL(k_split_bf16x3, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, (int)(cols / 8), ldx, reinterpret_cast<uint16_t *>(out), ld_out, plane_stride, (int)items);
    return check_launch(who);
}

int bnn_transpose_bf16(const void *in, int64_t in_batch_stride, int64_t ld_in, void *out, int64_t out_batch_stride, int64_t ld_out,
                       int64_t rows, int64_t cols, int batch, void *stream)
{
    const char *who = "bnn_transpose_bf16";
    if ((rows == 0 || batch == 0) && cols >= 1) return BNN_OK;
    if (!in || !out) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (rows < 0 || cols < 1 || batch < 0 || ld_in < cols || ld_out < rows) { set_error("%s: bad extent", who); return BNN_E_SHAPE; }
    if (cols % 8 != 0 || ld_in % 8 != 0 || ld_out % 8 != 0 || in_batch_stride % 8 != 0 || out_batch_stride % 8 != 0 || !al16(in) || !al16(out)) {
        set_error("%s: needs cols %% 8 == 0 and 16-B aligned rows on both sides", who);
        return BNN_E_UNSUPPORTED;
    }
    if (rows > 0x7FFFFFFF || cols > 0x7FFFFFFF || (cols + 63) / 64 > 65535 || batch > 65535) { set_error("%s: extent too large", who); return BNN_E_RANGE; }
    hipLaunchKernelGGL(k_transpose_bf16, dim3((unsigned)((ld_out + 63) / 64), (unsigned)((cols + 63) / 64), (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const uint16_t *>(in), in_batch_stride, ld_in,
                       reinterpret_cast<uint16_t *>(out), out_batch_stride, ld_out, (int)rows, (int)cols);
    return check_launch(who);
}


}  // extern "C"

static int conv_dense_launch(const float *x, int64_t x_sample_stride, const void *w, int64_t w_sample_stride, int64_t ldw,
                             const float *b, int64_t b_sample_stride, const float *sgn_in, const float *sgn_out,
                             float *y, int64_t y_sample_stride, const bnn_conv2d_shape_t *sh, int nsamples, int flags,
                             void *stream, const char *who, int64_t w_plane_stride = 0)
{
    const bool flip = sgn_in != nullptr;
    const bool x3 = w_plane_stride != 0;
    if (!x || !w || !y || !sh || (flip && !sgn_out)) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (sh->B < 1 || sh->C < 1 || sh->H < 1 || sh->W < 1 || sh->O < 1 || sh->KH < 1 || sh->KW < 1 || sh->stride_h < 1 ||
        sh->stride_w < 1 || sh->pad_h < 0 || sh->pad_w < 0 || sh->dil_h < 1 || sh->dil_w < 1 || nsamples < 1) { set_error("%s: bad shape", who); return BNN_E_SHAPE; }
    if (flags != 0) { set_error("%s: unknown flags", who); return BNN_E_UNSUPPORTED; }
    const int OH = (sh->H + 2 * sh->pad_h - sh->dil_h * (sh->KH - 1) - 1) / sh->stride_h + 1;
    const int OW = (sh->W + 2 * sh->pad_w - sh->dil_w * (sh->KW - 1) - 1) / sh->stride_w + 1;
    if (OH < 1 || OW < 1) { set_error("%s: kernel larger than the padded input", who); return BNN_E_SHAPE; }
    // what the kernel is built for (everything else: bnn_conv2d_forward_sampled / bnn_conv2d_forward)
    const int K = sh->C * sh->KH * sh->KW;
    const int bn = flip ? 2 * sh->O : sh->O;            // weight rows of the tile
    const bool ok = sh->groups == 1 && (sh->C == 64 || sh->C % 128 == 0) && (bn == 64 || bn == 128) &&
                    ldw >= K && ldw % 8 == 0 && w_sample_stride % 8 == 0 && al16(w) &&
                    (reinterpret_cast<uintptr_t>(x) & 3u) == 0 && (reinterpret_cast<uintptr_t>(y) & 3u) == 0;
    if (!ok) { set_error("%s: needs groups = 1, C = 64 or a multiple of 128, O = 64 or 128 (Flipout: 32 or 64), tap-major bf16 weights with 16-B aligned rows", who); return BNN_E_UNSUPPORTED; }
    ConvParams p{};
    const int64_t img_bytes = (int64_t)sh->H * sh->W * sh->C * 2 + (flip ? (sh->C / 8) * 16 : 0);   // + the image's sign masks
    const int P = OH * OW;
    if (x3 && (w_plane_stride % 8 != 0 || w_plane_stride < (int64_t)bn * ldw)) { set_error("%s: bad plane stride", who); return BNN_E_SHAPE; }
    // ring stages (8 KB / 16 KB each); three-plane operands: the big block for both widths (three image planes), 4 / 3 stages
    const int st = bn == 64 ? 4 : (x3 ? 3 : flip ? 4 : 6);
    const int64_t lds_block = (bn == 64 && !x3) ? kConvLds : kConvLdsBig;
    const int64_t ring = (int64_t)st * bn * 128;
    int img = 128 / P;                                  // rows per workgroup <= 128
    while (img > 0 && (img * img_bytes * (x3 ? 3 : 1) + ring > lds_block || (int64_t)img * sh->O * P * 4 > lds_block)) --img;
    if (img > sh->B) img = sh->B;
    if (img < 1 || (int64_t)bn * ldw * 2 >= ((int64_t)1 << 32)) { set_error("%s: one image (%lld B bf16) + the weight ring do not fit the LDS block, or more than 128 output pixels per image", who, (long long)img_bytes); return BNN_E_UNSUPPORTED; }
    p.X = x; p.x_sample_stride = x_sample_stride;
    p.W = reinterpret_cast<const uint16_t *>(w); p.w_sample_stride = w_sample_stride; p.ldw = ldw;
    p.bias = b; p.bias_sample_stride = b_sample_stride;
    p.Y = y; p.y_sample_stride = y_sample_stride;
    p.sgn_in = sgn_in; p.sgn_out = sgn_out;
    p.w_plane_stride = w_plane_stride;
    p.B = sh->B; p.C = sh->C; p.H = sh->H; p.Wd = sh->W; p.O = sh->O; p.KH = sh->KH; p.KW = sh->KW;
    p.sh = sh->stride_h; p.sw = sh->stride_w; p.ph = sh->pad_h; p.pw = sh->pad_w; p.dh = sh->dil_h; p.dw = sh->dil_w;
    static const int cdiag = [] { const char *e = getenv("BNN_CONV_DIAG"); return e ? atoi(e) : 0; }();
    p.flags = cdiag;
    p.OH = OH; p.OW = OW; p.S = nsamples; p.IMG = img; p.ntiles = (sh->B + img - 1) / img;
    p.img_bytes = (int32_t)((int64_t)sh->H * sh->W * sh->C * 2);
    const int64_t grid = (int64_t)p.ntiles * nsamples;
    if (grid > 0x7FFFFFFF) { set_error("%s: grid too large", who); return BNN_E_RANGE; }
    const dim3 g((unsigned)grid), blk(512);
    hipStream_t stq = (hipStream_t)stream;
    if (x3 && flip) {
        if (bn == 64) hipLaunchKernelGGL((k_conv_bf16<4, 4, kConvLdsBig, true, true>), g, blk, 0, stq, p);
        else hipLaunchKernelGGL((k_conv_bf16<8, 3, kConvLdsBig, true, true>), g, blk, 0, stq, p);
    } else if (x3) {
        if (bn == 64) hipLaunchKernelGGL((k_conv_bf16<4, 4, kConvLdsBig, false, true>), g, blk, 0, stq, p);
        else hipLaunchKernelGGL((k_conv_bf16<8, 3, kConvLdsBig, false, true>), g, blk, 0, stq, p);
    } else if (flip) {
        if (bn == 64) hipLaunchKernelGGL((k_conv_bf16<4, 4, kConvLds, true>), g, blk, 0, stq, p);
        else hipLaunchKernelGGL((k_conv_bf16<8, 4, kConvLdsBig, true>), g, blk, 0, stq, p);
    } else {
        if (bn == 64) hipLaunchKernelGGL((k_conv_bf16<4, 4, kConvLds>), g, blk, 0, stq, p);
        else hipLaunchKernelGGL((k_conv_bf16<8, 6, kConvLdsBig>), g, blk, 0, stq, p);
    }
    return check_launch(who);
}

extern "C" {

int bnn_conv2d_dense_forward(const float *x, int64_t x_sample_stride,
                             const void *w, int64_t w_sample_stride, int64_t ldw,
                             const float *b, int64_t b_sample_stride,
                             float *y, int64_t y_sample_stride,
                             const bnn_conv2d_shape_t *sh, int nsamples, int flags, void *stream)
{
    return conv_dense_launch(x, x_sample_stride, w, w_sample_stride, ldw, b, b_sample_stride, nullptr, nullptr, y, y_sample_stride,
                             sh, nsamples, flags, stream, "bnn_conv2d_dense_forward");
}

int bnn_conv2d_dense_forward_x3(const float *x, int64_t x_sample_stride,
                                const void *w, int64_t w_plane_stride, int64_t w_sample_stride, int64_t ldw,
                                const float *b, int64_t b_sample_stride,
                                float *y, int64_t y_sample_stride,
                                const bnn_conv2d_shape_t *sh, int nsamples, int flags, void *stream)
{
    if (w_plane_stride <= 0) { set_error("bnn_conv2d_dense_forward_x3: plane stride"); return BNN_E_SHAPE; }
    return conv_dense_launch(x, x_sample_stride, w, w_sample_stride, ldw, b, b_sample_stride, nullptr, nullptr, y, y_sample_stride,
                             sh, nsamples, flags, stream, "bnn_conv2d_dense_forward_x3", w_plane_stride);
}

int bnn_conv2d_flipout_forward(const float *x, const void *w, int64_t ldw, const float *sign_in, const float *sign_out,
                               float *y, const bnn_conv2d_shape_t *sh, int flags, void *stream)
{
    if (!sign_in || !sign_out) { set_error("bnn_conv2d_flipout_forward: NULL sign tensor"); return BNN_E_NULL; }
    return conv_dense_launch(x, 0, w, 0, ldw, nullptr, 0, sign_in, sign_out, y, 0, sh, 1, flags, stream, "bnn_conv2d_flipout_forward");
}

int bnn_conv2d_flipout_forward_x3(const float *x, const void *w, int64_t w_plane_stride, int64_t ldw, const float *sign_in,
                                  const float *sign_out, float *y, const bnn_conv2d_shape_t *sh, int flags, void *stream)
{
    if (!sign_in || !sign_out) { set_error("bnn_conv2d_flipout_forward_x3: NULL sign tensor"); return BNN_E_NULL; }
    if (w_plane_stride <= 0) { set_error("bnn_conv2d_flipout_forward_x3: plane stride of the weights"); return BNN_E_SHAPE; }
    return conv_dense_launch(x, 0, w, 0, ldw, nullptr, 0, sign_in, sign_out, y, 0, sh, 1, flags, stream, "bnn_conv2d_flipout_forward_x3",
                             w_plane_stride);
}

}  // extern "C"
