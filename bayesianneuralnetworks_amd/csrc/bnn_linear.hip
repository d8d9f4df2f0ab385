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
#include <cstdlib>

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

// One LDS-DMA piece: 64 lanes x 16 B land at lds_dst + 16 * lane (lds_dst wave-uniform);
// every lane supplies its own global source address.
__device__ __forceinline__ void dma16(const float *src, void *lds_dst)
{
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

template <int BM, int BN, int B_MODE, int COMPUTE>
__global__ __launch_bounds__(512) void k_linear_v4(const GemmParams p)
{
    constexpr int NT = 512, NW = 8;
    constexpr bool F32 = (COMPUTE == BNN_COMPUTE_F32);
    constexpr int BK = 32;
    constexpr int WTM = BM / NW;
    constexpr int TM = WTM / 16, TN = BN / 16;
    static_assert(WTM % 16 == 0 && BN % 16 == 0, "tile");
    constexpr int A_TILE = BM * 8;              // uint4 per A buffer: [BM rows][8 chunks of 4 fp32]
    constexpr int A_PIECES = BM / 8;            // 1-KiB LDS-DMA pieces (8 rows x 128 B) per tile
    constexpr int A_PPW = A_PIECES / NW;        // pieces per wave
    constexpr int CPR = F32 ? 8 : 4;            // 16-B chunks per B row
    constexpr int B_TILE = BN * CPR;
    constexpr int UPR = 8;                      // 4-draw units per B row per step
    constexpr int B_UNITS = BN * UPR;
    constexpr int B_PER = (B_UNITS + NT - 1) / NT;

    constexpr int NA_STAGES = 3;
    __shared__ __attribute__((aligned(16))) uint4 lds[NA_STAGES * A_TILE + 2 * B_TILE];
    uint4 *As0 = lds, *Bs0 = lds + NA_STAGES * A_TILE;

    // ---- block decode
    const int L = blockIdx.x;
    int s, panel, mt;
    {
        const int per_s = p.ntn * p.ntm;
        int rem;
        if (p.S % 8 == 0) {
            // MC sample -> XCD: blocks with equal blockIdx % 8 share one XCD's L2
            const int i_in = L >> 3;
            s = (L & 7) + 8 * (i_in / per_s);
            rem = i_in % per_s;
        } else {
            s = L / per_s;
            rem = L % per_s;
        }
        panel = rem / p.ntm;
        mt = rem % p.ntm;
    }
    const int m0 = mt * BM, n0 = panel * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fq = lane >> 4;
    const uint32_t sample = p.rng_w.sample0 + (uint32_t)s;
    uint32_t edev_w = 0;
    if constexpr (B_MODE == B_SAMPLED) edev_w = rng_epoch_dev(p.rng_w);
    const float *Ab = p.A + (int64_t)s * p.a_sample_stride;
    const float *Bsrc = (B_MODE == B_SAMPLED) ? p.mu : p.Bw + (int64_t)s * p.b_sample_stride;
    const int kmax = p.K - 4;

    // ---- A: LDS-DMA.  Piece pc = rows 8*pc .. 8*pc+7 of the tile; lane l lands at LDS chunk
    // position l & 7 of row l >> 3, so it must FETCH global chunk (l & 7) ^ (row & 7)
    // (the ds_read swizzle is applied on the source address; every 8 lanes read one full 128-B
    // line).  Rows >= M are clamped (outputs never stored), k >= K is clamped to K-4 (B is 0 there).
    const float *asrc[A_PPW];
#pragma unroll
    for (int j = 0; j < A_PPW; ++j) {
        const int row = (wave + j * NW) * 8 + (lane >> 3);
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;
        asrc[j] = Ab + (int64_t)m * p.lda;
    }
    const int a_chunk = (lane & 7) ^ ((lane >> 3) & 7);
    auto dma_A = [&](int buf, int k0) {
        int k = k0 + 4 * a_chunk;
        k = k < kmax ? k : kmax;
#pragma unroll
        for (int j = 0; j < A_PPW; ++j) {
            uint4 *dst = As0 + buf * A_TILE + (wave + j * NW) * 64;
            dma16(asrc[j] + k, dst);
        }
    };

    // ---- B: raw (mu, rho) units, two register sets (tile k+1 waiting to be drawn, tile k+2 in
    // flight).  The loads are inline asm: hipcc does not see them, so it cannot drain the LDS-DMA
    // pipeline with a vmcnt(0) when their results are used -- the counted waits below are ours.
    f32x4 rmA[B_PER], rrA[B_PER], rmB[B_PER], rrB[B_PER];
    int64_t brow[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int u = tid + i * NT;
        int n = n0 + (u / UPR) % BN;
        n = n < p.N ? n : p.N - 1;
        brow[i] = (int64_t)n * p.K;
    }
    constexpr int NB_OPS = B_PER * (B_MODE == B_SAMPLED ? 2 : 1);   // VMEM ops per load_B
    constexpr int NA_OPS = A_PPW;                                   // VMEM ops per dma_A
    auto load_B = [&](f32x4 (&rm)[B_PER], f32x4 (&rr)[B_PER], int k0) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int u = tid + i * NT;
            int kb = k0 + 4 * (u % UPR);
            kb = kb < kmax ? kb : kmax;
            const float *pm = Bsrc + brow[i] + kb;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rm[i]) : "v"(pm) : "memory");
            if constexpr (B_MODE == B_SAMPLED) {
                const float *pr = p.rho + brow[i] + kb;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rr[i]) : "v"(pr) : "memory");
            }
        }
    };
    // wait until all but the youngest N VMEM ops of this wave are done; the raw registers are
    // operands so that no consumer of them is scheduled above the wait
    auto wait_B = [&](f32x4 (&rm)[B_PER], f32x4 (&rr)[B_PER]) {
        if constexpr (B_PER == 1) {
            if constexpr (B_MODE == B_SAMPLED)
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rm[0]), "+v"(rr[0]) : "n"(NA_OPS + NB_OPS) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rm[0]) : "n"(NA_OPS + NB_OPS) : "memory");
        } else {
            static_assert(B_PER == 2, "B_PER");
            if constexpr (B_MODE == B_SAMPLED)
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(rm[0]), "+v"(rr[0]), "+v"(rm[1]), "+v"(rr[1]) : "n"(NA_OPS + NB_OPS) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rm[0]), "+v"(rm[1]) : "n"(NA_OPS + NB_OPS) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto draw_B = [&](const f32x4 (&rm)[B_PER], const f32x4 (&rr)[B_PER], int buf, int k0) {
        char *tile = reinterpret_cast<char *>(Bs0 + buf * B_TILE);
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int u = tid + i * NT;
            if (u >= B_UNITS) continue;
            const int row = u / UPR, c = u % UPR;
            const int n = n0 + row;
            const int kb = k0 + 4 * c;
            float4 w = make_float4(rm[i][0], rm[i][1], rm[i][2], rm[i][3]);
            if constexpr (B_MODE == B_SAMPLED) {
                // element index from the UNclamped (n, k): columns >= N draw garbage nobody reads
                const int64_t e0 = (int64_t)n * p.K + kb;
                const float4 z = eps4(p.rng_w, edev_w, (uint32_t)(e0 >> 2), sample);
                w.x = fmaf(sigma_draw(rr[i][0]), z.x, w.x);
                w.y = fmaf(sigma_draw(rr[i][1]), z.y, w.y);
                w.z = fmaf(sigma_draw(rr[i][2]), z.z, w.z);
                w.w = fmaf(sigma_draw(rr[i][3]), z.w, w.w);
            }
            if (kb >= p.K) w = make_float4(0.f, 0.f, 0.f, 0.f);   // K tail: exact zeros
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

    auto mfma_step = [&](int stage, int buf) {
        const uint4 *As = As0 + stage * A_TILE, *Bs = Bs0 + buf * B_TILE;
        if constexpr (F32) {
            uint4 b0[TN], b1[TN];
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int row = b * 16 + fi;
                b0[b] = Bs[bpos<true>(row, fq)];
                b1[b] = Bs[bpos<true>(row, fq + 4)];
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int row = wave * WTM + a * 16 + fi;
                const uint4 a0 = As[row * 8 + (fq ^ (row & 7))];
                const uint4 a1 = As[row * 8 + ((fq + 4) ^ (row & 7))];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const uint4 aa = t < 4 ? a0 : a1;
                    const int tt = t & 3;
                    const uint32_t ua = tt == 0 ? aa.x : tt == 1 ? aa.y : tt == 2 ? aa.z : aa.w;
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const uint4 bb = t < 4 ? b0[b] : b1[b];
                        const uint32_t ub = tt == 0 ? bb.x : tt == 1 ? bb.y : tt == 2 ? bb.z : bb.w;
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ua), __uint_as_float(ub),
                                                                         acc[a][b], 0, 0, 0);
                    }
                }
            }
        } else {
            uint4 bfr[TN];
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int row = b * 16 + fi;
                bfr[b] = Bs[bpos<false>(row, fq)];
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int row = wave * WTM + a * 16 + fi;
                const uint4 a0 = As[row * 8 + (fq ^ (row & 7))];
                const uint4 a1 = As[row * 8 + ((fq + 4) ^ (row & 7))];
                uint4 af;
                af.x = pack_bf16x2(__uint_as_float(a0.x), __uint_as_float(a0.y));
                af.y = pack_bf16x2(__uint_as_float(a0.z), __uint_as_float(a0.w));
                af.z = pack_bf16x2(__uint_as_float(a1.x), __uint_as_float(a1.y));
                af.w = pack_bf16x2(__uint_as_float(a1.z), __uint_as_float(a1.w));
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af),
                                                                        __builtin_bit_cast(bf16x8, bfr[b]),
                                                                        acc[a][b], 0, 0, 0);
            }
        }
    };

    // ---- pipeline.  Per wave and per step the VMEM queue gets, in this order, the NA_OPS DMA
    // pieces of A(k+2) and the NB_OPS raw loads of B(k+2); everything is waited for with ONE
    // counted s_waitcnt vmcnt(NA_OPS + NB_OPS) per step, one step later: it retires A(k+1) and
    // raw B(k+1) and leaves tile k+2 in flight across the barrier (raw s_barrier: a
    // __syncthreads() would drain the DMA queue).
    //   LDS:  A ring of 3 stages (k in use, k+1 landed/landing, k+2 in flight), drawn B x 2.
    const int nk = (p.K + BK - 1) / BK;
    dma_A(0, 0);
    load_B(rmA, rrA, 0);
    dma_A(1, BK);
    load_B(rmB, rrB, BK);
    wait_B(rmA, rrA);                                   // A(0), raw B(0) landed
    draw_B(rmA, rrA, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    auto step = [&](int kt, f32x4 (&rm_next)[B_PER], f32x4 (&rr_next)[B_PER],
                    f32x4 (&rm_free)[B_PER], f32x4 (&rr_free)[B_PER]) {
        // rm_next/rr_next: raw B(kt+1) (in flight or landed); rm_free/rr_free: drawn already
        int st2 = kt + 2;
        st2 = st2 % NA_STAGES;
        dma_A(st2, (kt + 2) * BK);
        load_B(rm_free, rr_free, (kt + 2) * BK);
        mfma_step(kt % NA_STAGES, kt & 1);
        wait_B(rm_next, rr_next);                       // A(kt+1) and raw B(kt+1) landed
        if (kt + 1 < nk) draw_B(rm_next, rr_next, (kt + 1) & 1, (kt + 1) * BK);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, rmB, rrB, rmA, rrA);
        if (kt + 1 < nk) step(kt + 1, rmA, rrA, rmB, rrB);
    }
    // nothing may still be writing this workgroup's LDS when it retires
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

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
    hipLaunchKernelGGL((k_linear_v4<BM, BN, BMODE, CP>), dim3((unsigned)grid), dim3(512), 0, st, p);
}

template <int BMODE, int CP>
static void select_v3(GemmParams &p, hipStream_t st)
{
    // 256 x 80 tiles: 240 workgroups at the BASELINE shape (N = 1200 = 15 x 80, 8 samples x 2
    // row tiles).  The activation stream (BM x K x 4 B per workgroup through one CU's L1) and the
    // draw (BN x K per workgroup) are the two costs a tile shape trades; see DESIGN.md.
    if (p.N <= 16) launch_v3<128, 16, BMODE, CP>(p, st);
    else launch_v3<256, 80, BMODE, CP>(p, st);
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
