// bnn_linear.hip -- K2-linear, the draw-paced fused kernel (dense A, 16-B aligned operands).
//
//   y[s] = x[s] . W_s^T + b_s,   W_s = mu + sigma(rho) * eps_s   drawn in the B-operand loader.
//
// Why a second kernel next to bnn_gemm.hip: at the BASELINE shapes the scarce resources are the
// eps draw (VALU: ~2.4 SIMD-cycles per draw, bnn_diag_sampler) and the activation stream through
// each CU's vector L1 -- not the MFMA (8 x 1.44 M draws but only 11.8 GFLOP of bf16 MFMA in the
// layer-2 launch).  Hence:
//   * one workgroup owns ALL batch rows of one MC sample for a column panel (BM = 512 in bf16
//     mode), so each weight of a sample is drawn exactly once;
//   * waves split the tile along M only: an activation row is consumed by exactly one wave,
//     so A never touches LDS -- every lane loads its own MFMA fragment (8 consecutive k) straight
//     from global memory / L2 into registers, one step ahead, with no barrier in its path;
//   * only the drawn weights go through LDS (12-20 KB, double-buffered, one barrier per step);
//     every wave draws an equal share of the Philox blocks of a step;
//   * block decode puts MC sample s on XCD s % 8: the sample's activations (2.4 MB at the
//     BASELINE shape) stay in that XCD's 4 MB L2 while mu / rho stream through.
//
// k order inside a 32-wide macro-step: MFMA lane (i = l & 15, q = l >> 4) holds k = 4q + t
// (t = 0..3) and k = 16 + 4q + (t - 4) (t = 4..7) -- two 64-B-contiguous global loads per row;
// the same permutation is applied to B when its 4-draw units are written to LDS, so the sum is
// unchanged.  bf16: one v_mfma_f32_16x16x32_bf16 per macro-step; fp32: eight
// v_mfma_f32_16x16x4_f32 (MFMA k-slot q, step t).
#include "bnn_device.hpp"
#include "bnn_gemm_params.hpp"

namespace bnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// LDS position of 16-B chunk `c` of B row `row`.
// bf16: [row][4 chunks], swizzle h((row >> 2) & 3), h = {0, 2, 3, 1};
// fp32: [row][8 chunks], swizzle row & 7.  Both conflict-free for ds_read_b128.
template <bool F32>
__device__ __forceinline__ int bpos(int row, int c)
{
    if constexpr (F32) return row * 8 + (c ^ (row & 7));
    else return row * 4 + (c ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3));
}

template <int BM, int BN, int B_MODE, int COMPUTE>
__global__ __launch_bounds__(512) void k_linear_v3(const GemmParams p)
{
    constexpr int NT = 512, NW = 8;
    constexpr bool F32 = (COMPUTE == BNN_COMPUTE_F32);
    constexpr int MS = F32 ? 1 : 2;             // 32-wide macro-steps per barrier
    constexpr int BK = 32 * MS;
    constexpr int WTM = BM / NW;
    constexpr int TM = WTM / 16, TN = BN / 16;
    static_assert(WTM % 16 == 0 && BN % 16 == 0, "tile");
    constexpr int CPR = F32 ? 8 : 4;            // 16-B chunks per B row per macro-step
    constexpr int B_TILE = MS * BN * CPR;       // uint4 per buffer
    constexpr int UPR = 8 * MS;                 // 4-draw units per B row per step
    constexpr int B_UNITS = BN * UPR;
    constexpr int B_PER = (B_UNITS + NT - 1) / NT;

    __shared__ __attribute__((aligned(16))) uint4 lds[2 * B_TILE];

    // ---- block decode
    const int L = blockIdx.x;
    int s, panel, mt;
    if (p.S % 8 == 0) {
        // MC sample -> XCD: blocks with equal blockIdx % 8 share one XCD's L2
        const int i_in = L >> 3;
        const int per_s = p.ntn * p.ntm;
        s = (L & 7) + 8 * (i_in / per_s);
        const int rem = i_in % per_s;
        panel = rem / p.ntm;
        mt = rem % p.ntm;
    } else {
        const int per_s = p.ntn * p.ntm;
        s = L / per_s;
        const int rem = L % per_s;
        panel = rem / p.ntm;
        mt = rem % p.ntm;
    }
    const int m0 = mt * BM, n0 = panel * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fq = lane >> 4;
    const uint32_t sample = p.rng_w.sample0 + (uint32_t)s;
    uint32_t edev_w = 0;
    if constexpr (B_MODE == B_SAMPLED) edev_w = rng_epoch_dev(p.rng_w);
    const float *Ab = p.A + (int64_t)s * p.a_sample_stride;
    const float *Bsrc = (B_MODE == B_SAMPLED) ? p.mu : p.Bw + (int64_t)s * p.b_sample_stride;

    // ---- A: per-lane fragment rows (fixed across the K loop).
    // Every load is UNCONDITIONAL (straight-line code lets hipcc count vmcnt instead of draining
    // it): rows >= M are clamped to row M-1 (their outputs are never stored) and k >= K is clamped
    // to K-4 (the matching B entries are exact zeros, and the duplicated values come from the
    // same row, so 0 * x cannot introduce a NaN the row would not have anyway).
    const float *arow[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        int m = m0 + wave * WTM + a * 16 + fi;
        m = m < p.M ? m : p.M - 1;
        arow[a] = Ab + (int64_t)m * p.lda;
    }
    const int kmax = p.K - 4;
    float4 ra[MS][TM][2];       // raw fragment of the NEXT step (in flight)
    auto load_A = [&](int k0) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int k = k0 + ms * 32 + 16 * h + 4 * fq;
                k = k < kmax ? k : kmax;
#pragma unroll
                for (int a = 0; a < TM; ++a)
                    ra[ms][a][h] = *reinterpret_cast<const float4 *>(arow[a] + k);
            }
    };

    // ---- B: raw (mu, rho) units of the NEXT step, drawn into LDS after the MFMAs
    float4 rm[B_PER], rr[B_PER];
    int64_t brow[B_PER];        // clamped row offsets (columns >= N are never stored)
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int u = tid + i * NT;
        int n = n0 + (u / UPR) % BN;
        n = n < p.N ? n : p.N - 1;
        brow[i] = (int64_t)n * p.K;
    }
    auto load_B = [&](int k0) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int u = tid + i * NT;
            int kb = k0 + 4 * (u % UPR);         // u % UPR = ms * 8 + c
            kb = kb < kmax ? kb : kmax;
            rm[i] = *reinterpret_cast<const float4 *>(Bsrc + brow[i] + kb);
            if constexpr (B_MODE == B_SAMPLED) rr[i] = *reinterpret_cast<const float4 *>(p.rho + brow[i] + kb);
        }
    };
    auto draw_B = [&](int buf, int k0) {
        char *Bs = reinterpret_cast<char *>(lds + buf * B_TILE);
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int u = tid + i * NT;
            if (u >= B_UNITS) continue;
            const int row = u / UPR, cc = u % UPR;
            const int ms = cc >> 3, c = cc & 7;
            const int n = n0 + row;
            const int kb = k0 + 4 * cc;
            float4 w = rm[i];
            if constexpr (B_MODE == B_SAMPLED) {
                // element index from the UNclamped (n, k): columns >= N draw garbage nobody reads
                const int64_t e0 = (int64_t)n * p.K + kb;
                const float4 z = eps4(p.rng_w, edev_w, (uint32_t)(e0 >> 2), sample);
                w.x = fmaf(sigma_draw(rr[i].x), z.x, rm[i].x);
                w.y = fmaf(sigma_draw(rr[i].y), z.y, rm[i].y);
                w.z = fmaf(sigma_draw(rr[i].z), z.z, rm[i].z);
                w.w = fmaf(sigma_draw(rr[i].w), z.w, rm[i].w);
            }
            if (kb >= p.K) w = make_float4(0.f, 0.f, 0.f, 0.f);   // K tail: exact zeros
            char *tile = Bs + ms * (BN * CPR * 16);
            if constexpr (F32) {
                uint4 o;
                o.x = __float_as_uint(w.x); o.y = __float_as_uint(w.y);
                o.z = __float_as_uint(w.z); o.w = __float_as_uint(w.w);
                *reinterpret_cast<uint4 *>(tile + bpos<true>(row, c) * 16) = o;
            } else {
                // unit c: k = 4c..4c+3 -> MFMA lane-q c & 3, elements 4 * (c >> 2) ..
                uint2 o;
                o.x = pack_bf16x2(w.x, w.y);
                o.y = pack_bf16x2(w.z, w.w);
                *reinterpret_cast<uint2 *>(tile + bpos<false>(row, c & 3) * 16 + (c >> 2) * 8) = o;
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    // Loads are issued unconditionally (clamped addresses make any k0 legal): with no load
    // under a branch hipcc counts vmcnt exactly, so the A fragment wait leaves the younger
    // (mu, rho) loads in flight and vice versa.
    load_A(0);
    load_B(0);
    draw_B(0, 0);
    load_B(BK);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const uint4 *Bs = lds + (kt & 1) * B_TILE;
        // take this step's A fragment out of the in-flight registers (bf16: packed at once),
        // then refill them with the next step's loads
        float fa[F32 ? MS : 1][F32 ? TM : 1][8];
        uint4 fp[F32 ? 1 : MS][F32 ? 1 : TM];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                if constexpr (F32) {
                    fa[ms][a][0] = ra[ms][a][0].x; fa[ms][a][1] = ra[ms][a][0].y;
                    fa[ms][a][2] = ra[ms][a][0].z; fa[ms][a][3] = ra[ms][a][0].w;
                    fa[ms][a][4] = ra[ms][a][1].x; fa[ms][a][5] = ra[ms][a][1].y;
                    fa[ms][a][6] = ra[ms][a][1].z; fa[ms][a][7] = ra[ms][a][1].w;
                } else {
                    fp[ms][a].x = pack_bf16x2(ra[ms][a][0].x, ra[ms][a][0].y);
                    fp[ms][a].y = pack_bf16x2(ra[ms][a][0].z, ra[ms][a][0].w);
                    fp[ms][a].z = pack_bf16x2(ra[ms][a][1].x, ra[ms][a][1].y);
                    fp[ms][a].w = pack_bf16x2(ra[ms][a][1].z, ra[ms][a][1].w);
                }
            }
        load_A((kt + 1) * BK);

#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            const uint4 *tile = Bs + ms * (BN * CPR);
            if constexpr (F32) {
                uint4 b0[TN], b1[TN];
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int row = b * 16 + fi;
                    b0[b] = tile[bpos<true>(row, fq)];
                    b1[b] = tile[bpos<true>(row, fq + 4)];
                }
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b) {
                            const uint4 bb = t < 4 ? b0[b] : b1[b];
                            const int tt = t & 3;
                            const uint32_t ub = tt == 0 ? bb.x : tt == 1 ? bb.y : tt == 2 ? bb.z : bb.w;
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ms][a][t], __uint_as_float(ub),
                                                                             acc[a][b], 0, 0, 0);
                        }
            } else {
                uint4 bfr[TN];
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int row = b * 16 + fi;
                    bfr[b] = tile[bpos<false>(row, fq)];
                }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fp[ms][a]),
                                                                            __builtin_bit_cast(bf16x8, bfr[b]),
                                                                            acc[a][b], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) draw_B((kt + 1) & 1, (kt + 1) * BK);
        load_B((kt + 2) * BK);
        __syncthreads();
    }

    // ---- epilogue: bias drawn per column, activation, store
    uint32_t edev_b = 0;
    const bool sampled_bias = (p.mu_b != nullptr);
    if (sampled_bias) edev_b = rng_epoch_dev(p.rng_b);
    float *Yb = p.Y + (int64_t)s * p.y_sample_stride;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + b * 16 + fi;
        if (n >= p.N) continue;
        float bias = 0.f;
        if (sampled_bias)
            bias = fmaf(sigma_draw(p.rho_b[n]), eps1(p.rng_b, edev_b, (uint64_t)n, p.rng_b.sample0 + (uint32_t)s), p.mu_b[n]);
        else if (p.bias)
            bias = p.bias[(int64_t)s * p.bias_sample_stride + n];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wave * WTM + a * 16 + fq * 4 + r;
                if (m >= p.M) continue;
                float v = acc[a][b][r] + bias;
                if (p.flags & BNN_FLAG_RELU) v = fmaxf(v, 0.f);
                Yb[(int64_t)m * p.ldy + n] = v;
            }
        }
    }
}

template <int BM, int BN, int BMODE, int CP>
static void launch_v3(GemmParams &p, hipStream_t st)
{
    p.ntm = (p.M + BM - 1) / BM;
    p.ntn = (p.N + BN - 1) / BN;
    const int64_t grid = (int64_t)p.ntn * p.ntm * p.S;
    hipLaunchKernelGGL((k_linear_v3<BM, BN, BMODE, CP>), dim3((unsigned)grid), dim3(512), 0, st, p);
}

template <int BMODE, int CP>
static void select_v3(GemmParams &p, hipStream_t st)
{
    // fp32 is MFMA-bound: fill the chip (256 x 80 -> 240 workgroups at the BASELINE shape; the
    // doubled draw hides under the MFMAs).  bf16 is draw-bound: draw once (512 rows per tile).
    if (p.N <= 16) {
        launch_v3<128, 16, BMODE, CP>(p, st);
    } else if (CP == BNN_COMPUTE_F32) {
        launch_v3<256, 80, BMODE, CP>(p, st);
    } else {
        launch_v3<512, 48, BMODE, CP>(p, st);
    }
}

// Called by linear_common (bnn_gemm.hip) when operands are 16-B aligned and K % 4 == 0.
int dispatch_linear_v2(GemmParams &p, bool sampled, int compute, hipStream_t st, const char *who)
{
    if (compute == BNN_COMPUTE_F32) {
        if (sampled) select_v3<B_SAMPLED, BNN_COMPUTE_F32>(p, st);
        else select_v3<B_PLAIN, BNN_COMPUTE_F32>(p, st);
    } else if (compute == BNN_COMPUTE_BF16) {
        if (sampled) select_v3<B_SAMPLED, BNN_COMPUTE_BF16>(p, st);
        else select_v3<B_PLAIN, BNN_COMPUTE_BF16>(p, st);
    } else {
        set_error("%s: unknown compute mode %d", who, compute);
        return BNN_E_DTYPE;
    }
    return check_launch(who);
}

}  // namespace bnn
