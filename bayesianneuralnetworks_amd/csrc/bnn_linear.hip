// bnn_linear.hip -- K2-linear: the fused sampled GEMM as a producer / consumer pipeline
// (dense A, 16-B aligned operands, K % 4 == 0).
//
//   y[s] = x[s] . W_s^T + b_s,   W_s = mu + sigma(rho) * eps_s   drawn on the fly, never stored.
//
// What the measurements on MI355X said (tools/diag_*.py, profiles/):
//   * the eps draw costs ~2.4 SIMD-cycles per weight and is a ~600-instruction dependent chain
//     (Philox rounds -> Box-Muller -> softplus): put in series with the MFMAs of a k-step it
//     dominated every step;
//   * fragment-shaped or row-shaped global loads into VGPRs pull only 7-15 B/clk per CU out of
//     L2, LDS-DMA (global_load_lds_dwordx4) 40-50 B/clk;
//   * a workgroup barrier per k-step left every wave parked 60 % of the time (SQ_WAIT_ANY).
// Hence this structure, with NO workgroup barrier in the main loop:
//   consumer waves (NC): each owns 32 batch rows.  It LDS-DMAs ITS OWN rows of x into a private
//     3-stage LDS ring (8 lanes fetch one 128-B line, swizzled on the source address), waits with
//     a counted vmcnt for its own pieces only, reads A / B fragments with ds_read_b128 and issues
//     the MFMAs (v_mfma_f32_16x16x32_bf16, or 8 x v_mfma_f32_16x16x4_f32 for exact fp32).
//   producer waves (NP): fetch (mu, rho) one chunk ahead (inline-asm loads + counted vmcnt, so
//     hipcc cannot drain them), draw CH k-steps worth of weights at a time -- several independent
//     Philox blocks per lane in flight -- and write them to a double-buffered LDS chunk.
//   hand-off: per chunk buffer a FULL counter (producers add after their ds_writes completed)
//     and a FREE counter (consumers add after their last read), both plain LDS words polled with
//     ds_read + s_sleep.  All LDS lives in one __shared__ array.
//   block decode puts MC sample s on XCD s % 8: a sample's activations stay in that XCD's L2.
//
// k order inside a 32-wide step: MFMA lane (i = l & 15, q = l >> 4) holds k = 4q + t (t < 4) and
// k = 16 + 4q + (t - 4) (t >= 4); B's 4-draw unit c (k = 4c..4c+3) is written for lane-q c & 3,
// half c >> 2 -- the same permutation on both operands leaves the sum unchanged.
#include <cstdlib>
#include <type_traits>

#include "bnn_device.hpp"
#include "bnn_gemm_params.hpp"
#include "bnn_dma.hpp"

namespace bnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// LDS position (in 16-B chunks) of chunk `c` of B row `row` inside one 32-k sub-tile.
// bf16: [row][4 chunks], swizzle h((row >> 2) & 3), h = {0, 2, 3, 1};
// fp32: [row][8 chunks], swizzle row & 7.  Both conflict-free for ds_read_b128.
template <bool F32>
__device__ __forceinline__ int bpos(int row, int c)
{
    if constexpr (F32) return row * 8 + (c ^ (row & 7));
    else return row * 4 + (c ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3));
}

__device__ __forceinline__ int lds_load(int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Poll an LDS counter until it reaches `target`.  Bounded: a protocol bug must not hang the GPU -- and must not
// turn into wrong numbers either: on a timeout the wave sets the sticky device error word (word 0 of the registered
// workspace, read and cleared by bnn_check_device) and the caller skips its stores.  Returns false on a timeout.
__device__ __forceinline__ bool lds_wait_ge(int *p, int target, unsigned *err)
{
    bool ok = false;
    for (int spin = 0; spin < (1 << 22); ++spin) {
        if (lds_load(p) >= target) { ok = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (!ok && err && (threadIdx.x & 63) == 0)
        __hip_atomic_fetch_or(err, kDevErrHandoffTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ok;
}

// Symmetric pipeline: every wave draws its share of chunk ch+1, then consumes chunk ch for its own
// RW rows.  NW waves (16 -> 4 per SIMD: the draw saturates the VALU while other waves sit in DMA /
// MFMA), BM = NW * RW rows per workgroup, BN columns, CH k-steps (32 k each) per chunk, S-stage
// private A ring per wave.
// STAMPS: diagnostic build only (BNN_STAMPS=<device pointer>): wave 0 of block p.dbg_block writes
// s_memtime stamps to p.dbg; stamps never feed an output value.
// ABF: the activations are bf16 in memory (written so by the previous layer's epilogue): half the
// A stream, half the A ring, and the DMA'd 16-B chunk IS the MFMA fragment (no conversion).
// B_MODE == B_SAMPLED_T: the input-gradient contraction gx[m][c] = sum_r gy[m][r] * W_s[r][c] -- the
// reduction runs over the weight's ROWS, so a 4-draw unit (one Philox block = 4 consecutive columns of
// one row) lands in four LDS rows at one reduction position (4 scalar LDS writes instead of one
// vector write); p.N = columns of W (outputs here), p.K = rows of W (reduction), row length p.N.
// (A split-K variant for the N <= 16 head existed in round 1 behind BNN_SPLITK; it faulted once on the box at 64 tiles x 4
// splits, the cause was not found from the evidence in hand, and it bought 3 % of the step: removed in round 2.)
template <int NW, int RW, int BN, int CH, int S, int NB, int B_MODE, int COMPUTE, bool ABF = false, bool STAMPS = false>
__global__ __launch_bounds__(NW * 64) void k_linear_sym(const GemmParams p)
{
    constexpr bool F32 = (COMPUTE == BNN_COMPUTE_F32);
    // kComputeBf16x3 (internal): fp32-accurate contraction on the bf16 MFMA.  a = ah + am + al, three bf16 terms that
    // hold all 24 significand bits (each residual is exact in fp32); of the nine partial products the six largest are
    // kept -- the dropped ones are <= 2^-25 |a b|, below the rounding of one fp32 multiply -- and accumulated in fp32,
    // smallest first.  Six 16x16x32 bf16 MFMAs (96 cycles) stand for the eight 16x16x4 fp32 ones (256 cycles) of the
    // same block.  A stays fp32 in memory / LDS and is split at fragment load; the drawn weights are split once, at draw
    // time, into three bf16 LDS images.
    constexpr bool X3 = (COMPUTE == kComputeBf16x3);
    static_assert(!(X3 && (ABF || B_MODE == B_SAMPLED_T)), "bf16x3: fp32 activations, forward only");
    constexpr bool SAMPLED = (B_MODE != B_PLAIN);
    constexpr bool BT = (B_MODE == B_SAMPLED_T);
    static_assert(!(ABF && F32), "bf16 activations only in bf16 compute mode");
    // Narrow-layer (N <= 16) forward: its 32 workgroups leave most of the chip idle, so the launch can carry the
    // first pass of the model's KL as extra workgroups (a launch of its own costs >= 4 us).  They use the first 256
    // threads; the other waves retire at once (a retired wave no longer counts at the barrier).
    constexpr bool CARRIES_KL = (B_MODE == B_SAMPLED && BN == 16 && !STAMPS);
    if constexpr (CARRIES_KL) {
        if (p.kl.nblocks > 0 && (int)blockIdx.x >= p.gemm_grid) {
            if (threadIdx.x < 256) kl_piggy_block(p.kl, (int)blockIdx.x - p.gemm_grid);
            return;
        }
    }
    constexpr int NT = NW * 64;
    constexpr int BM = NW * RW;
    constexpr int TM = RW / 16, TN = BN / 16;
    static_assert(RW % 16 == 0 && BN % 16 == 0, "tile");
    constexpr int ACH = ABF ? 4 : 8;            // 16-B chunks per A row per 32-k step
    constexpr int RPP = 64 / ACH;               // rows per 1-KiB DMA piece
    constexpr int PW = RW / RPP;                // DMA pieces per wave per k-step
    constexpr int A_STAGE = RW * ACH;           // uint4 per wave per stage
    constexpr int CPR = F32 ? 8 : 4;            // 16-B chunks per B row per 32-k sub-tile
    constexpr int B_IMG = BN * CPR;             // uint4 per 32-k sub-tile image
    constexpr int B_SUB = B_IMG * (X3 ? 3 : 1); // uint4 per sub-tile (bf16x3: hi, mid, lo images one after the other)
    constexpr int B_CHUNK = CH * B_SUB;         // uint4 per chunk buffer
    constexpr int UPC = BN * 8 * CH;            // 4-draw units per chunk
    constexpr int UPL = (UPC + NT - 1) / NT;    // units per lane per chunk (lane t: units t, t + NT, ..)
    static_assert(UPL <= 2, "at most two units per lane per chunk");
    constexpr int LPU = (SAMPLED ? 2 : 1) * UPL;   // raw loads per lane per chunk fetch
    constexpr int A_WORDS = NW * S * A_STAGE;
    constexpr int WAIT_DRAW = (CH < S - 1 ? CH : S - 1) * PW + LPU;
    constexpr int LA = NB > 2 ? NB - 2 : 1;     // chunks drawn ahead of the one being consumed

    // NB chunk buffers: chunk c lives in buffer c % NB; with NB > 2 a wave may draw up to NB - 1
    // chunks ahead of the slowest consumer instead of meeting it at every chunk.
    __shared__ __attribute__((aligned(16))) uint4 lds[A_WORDS + NB * B_CHUNK + (2 * NB + 3) / 4 + (BN + 3) / 4];
    uint4 *Bs0 = lds + A_WORDS;
    int *full = reinterpret_cast<int *>(lds + A_WORDS + NB * B_CHUNK);   // full[NB], then free[NB]
    int *freec = full + NB;
    float *bias_lds = reinterpret_cast<float *>(lds + A_WORDS + NB * B_CHUNK + (2 * NB + 3) / 4);   // bias_lds[BN]

    unsigned long long *dbg = nullptr;
    int dbg_i = 0;
    auto stamp = [&](int tag) {
        if constexpr (STAMPS) {
            if (dbg && dbg_i < 2000) {
                unsigned long long t;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
                if ((threadIdx.x & 63) == 0) { dbg[2 * dbg_i] = (unsigned long long)tag; dbg[2 * dbg_i + 1] = t; }
                ++dbg_i;
            }
        }
    };

    // ---- block decode
    const int L = (int)blockIdx.x;
    int s, panel, mt;
    {
        const int per_s = p.ntn * p.ntm;
        int rem;
        if (p.xcd_a > 0) {
            // 2-D XCD map: XCD (blockIdx % 8) = (sample group, panel group).  One XCD's L2 then holds 1/xb of
            // mu / rho and 1/xa of the activations instead of ALL of mu / rho (sample -> XCD): fabric traffic
            // of layer 2 falls from 8 x 11.5 + 9.8 MB to 8 x (2.9 + 4.9) MB.  The grid is padded to the largest
            // XCD share; surplus workgroups leave at once (before any barrier).
            const int xa = p.xcd_a, xb = 8 / xa;
            const int xcd = L & 7, idx = L >> 3;
            const int sg = xcd % xa, pg = xcd / xa;
            const int spg = p.S / xa, ppg = (p.ntn + xb - 1) / xb;
            const int s_l = idx / (ppg * p.ntm), r2 = idx % (ppg * p.ntm);
            s = sg * spg + s_l;
            panel = pg * ppg + r2 / p.ntm;
            mt = r2 % p.ntm;
            if (s_l >= spg || panel >= p.ntn) return;
            rem = panel * p.ntm + mt;
        } else if (p.S % 8 == 0) {
            const int i_in = L >> 3;            // MC sample -> XCD (blockIdx % 8)
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
    const int kmax = p.K - 4;
    const int nk_all = (p.K + 31) / 32;                 // 32-wide k-steps
    const int nch_all = (nk_all + CH - 1) / CH;         // chunks (steps past K multiply exact zeros)
    const int nch = nch_all;
    constexpr int k_lo = 0;
    bool handoff_ok = true;                             // false after a bounded hand-off wait gave up

    if (tid < 2 * NB) full[tid] = 0;
    // the panel's BN bias values, drawn ONCE per workgroup (every wave needs all of them in its epilogue; per
    // wave it was TN serial Philox blocks on every lane) -- ordered before all reads by the barrier below
    if (p.mu_b != nullptr && tid < BN) {
        const int n = n0 + tid;
        float bv = 0.f;
        if (n < p.N)
            bv = fmaf(sigma_draw(p.rho_b[n]), eps1(p.rng_b, rng_epoch_dev(p.rng_b), (uint64_t)n, p.rng_b.sample0 + (uint32_t)s), p.mu_b[n]);
        bias_lds[tid] = bv;
    }
    __syncthreads();

    if constexpr (STAMPS) {
        if (p.dbg && (int)blockIdx.x == p.dbg_block && wave == 0) dbg = p.dbg;
    }
    stamp(10);                                                      // kernel entry (after the LDS flag init barrier)

    // ---- draw side: this lane's unit of every chunk
    const uint32_t sample = p.rng_w.sample0 + (uint32_t)s;
    uint32_t edev_w = 0;
    if constexpr (SAMPLED) edev_w = rng_epoch_dev(p.rng_w);
    // round keys in VGPRs where the register budget allows (20 VGPRs; the fp32 and bf16x3 instantiations have two units
    // per lane in flight and would spill)
    constexpr bool KEYS_V = SAMPLED && !X3 && !F32;
    PhiloxKeys keys_w{};
    if constexpr (KEYS_V) keys_w = philox_keys(p.rng_w.key0, p.rng_w.key1);
    const float *Bsrc = SAMPLED ? p.mu : p.Bw + (int64_t)s * p.b_sample_stride;
    // unit u of a chunk -> (row, cc): row = u / (8 CH), cc = u % (8 CH) = sub * 8 + c
    // (B_SAMPLED_T: u -> (r_local, cg) = (u / (BN/4), u % (BN/4)): reduction row, 4-column group)
    struct Raw { f32x4 m[UPL], r[UPL]; };
    constexpr int CG = BN / 4;
    int64_t brow[UPL];                          // fixed part of the unit's address
#pragma unroll
    for (int i = 0; i < UPL; ++i) {
        if constexpr (BT) {
            int c = n0 + 4 * ((tid + i * NT) % CG);
            c = c < p.N - 4 ? c : p.N - 4;      // columns >= N: clamped loads, results never stored
            brow[i] = c;
        } else {
            int n = n0 + ((tid + i * NT) / (8 * CH)) % BN;
            n = n < p.N ? n : p.N - 1;          // columns >= N: clamped loads, results never stored
            brow[i] = (int64_t)n * p.K;
        }
    }
    Raw rawA, rawB;                             // raw (mu, rho) of the chunk being drawn / in flight
#pragma unroll
    for (int i = 0; i < UPL; ++i) rawA.r[i] = rawB.r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto fetch_unit = [&](Raw &w, int ch) {
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            int64_t off;
            if constexpr (BT) {
                int r = k_lo + ch * (32 * CH) + ((tid + i * NT) / CG) % (32 * CH);
                r = r < p.K ? r : p.K - 1;      // past K: clamped, drawn as zeros
                off = (int64_t)r * p.N + brow[i];
            } else {
                const int u_cc = (tid + i * NT) % (8 * CH);
                int kb = k_lo + ch * (32 * CH) + 4 * u_cc;
                kb = kb < kmax ? kb : kmax;     // past K: clamped, drawn as zeros
                off = brow[i] + kb;
            }
            const float *pm = Bsrc + off;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w.m[i]) : "v"(pm) : "memory");
            if constexpr (SAMPLED) {
                const float *pr = p.rho + off;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(w.r[i]) : "v"(pr) : "memory");
            }
        }
    };
    // counted wait; the raw registers are operands so that no use of them is scheduled above it
    auto wait_raw = [&](Raw &w, auto N) {
        constexpr int n = decltype(N)::value;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory");
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            asm volatile("" : "+v"(w.m[i]));
            if constexpr (SAMPLED) asm volatile("" : "+v"(w.r[i]));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto draw_unit = [&](const Raw &w_, int ch) {
        char *buf = reinterpret_cast<char *>(Bs0 + (ch % NB) * B_CHUNK);
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            const int u = tid + i * NT;
            if (u >= UPC) continue;
            float4 w = make_float4(w_.m[i][0], w_.m[i][1], w_.m[i][2], w_.m[i][3]);
            if constexpr (BT) {
                const int r_local = u / CG, cg = u % CG;
                const int sub = r_local >> 5, rr = r_local & 31;
                const int r = k_lo + ch * (32 * CH) + r_local;            // weight row = reduction index
                const int c0 = n0 + 4 * cg;                               // weight column = output column
                // element index from the UNclamped (r, c0): columns >= N draw values nobody reads
                const int64_t e0 = (int64_t)r * p.N + c0;
                const float4 z = KEYS_V ? eps4(p.rng_w, keys_w, edev_w, (uint32_t)(e0 >> 2), sample) : eps4(p.rng_w, edev_w, (uint32_t)(e0 >> 2), sample);
                w.x = fmaf(sigma_draw(w_.r[i][0]), z.x, w.x);
                w.y = fmaf(sigma_draw(w_.r[i][1]), z.y, w.y);
                w.z = fmaf(sigma_draw(w_.r[i][2]), z.z, w.z);
                w.w = fmaf(sigma_draw(w_.r[i][3]), z.w, w.w);
                if (r >= p.K) w = make_float4(0.f, 0.f, 0.f, 0.f);       // reduction tail: exact zeros
                char *tile = buf + sub * (B_SUB * 16);
                const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = 4 * cg + j;
                    if constexpr (F32) {
                        *reinterpret_cast<float *>(tile + bpos<true>(row, rr >> 2) * 16 + (rr & 3) * 4) = wv[j];
                    } else {
                        // position of reduction element rr inside the lane-q fragment (see the fp32-A /
                        // bf16-A orders below)
                        const int bq = ABF ? (rr >> 3) : ((rr & 15) >> 2);
                        const int pos = ABF ? (rr & 7) : ((rr & 3) + 4 * (rr >> 4));
                        *reinterpret_cast<uint16_t *>(tile + bpos<false>(row, bq) * 16 + pos * 2) = f2bf(wv[j]);
                    }
                }
                continue;
            }
            const int u_row = u / (8 * CH), u_cc = u % (8 * CH);
            const int sub = u_cc >> 3, c = u_cc & 7;
            const int n = n0 + u_row;
            // columns >= N feed only outputs that are never stored: no draw, no LDS write (with 8 CH = 64 a wave owns
            // whole rows, so the head's waves of rows 10 .. 15 skip their unit altogether)
            if (n >= p.N) continue;
            const int kb = k_lo + ch * (32 * CH) + 4 * u_cc;
            if constexpr (B_MODE == B_SAMPLED) {
                // element index from the UNclamped (n, k): columns >= N draw values nobody reads
                const int64_t e0 = (int64_t)n * p.K + kb;
                const float4 z = KEYS_V ? eps4(p.rng_w, keys_w, edev_w, (uint32_t)(e0 >> 2), sample) : eps4(p.rng_w, edev_w, (uint32_t)(e0 >> 2), sample);
                w.x = fmaf(sigma_draw(w_.r[i][0]), z.x, w.x);
                w.y = fmaf(sigma_draw(w_.r[i][1]), z.y, w.y);
                w.z = fmaf(sigma_draw(w_.r[i][2]), z.z, w.z);
                w.w = fmaf(sigma_draw(w_.r[i][3]), z.w, w.w);
            }
            if (kb >= p.K) w = make_float4(0.f, 0.f, 0.f, 0.f);       // K tail: exact zeros
            char *tile = buf + sub * (B_SUB * 16);
            if constexpr (F32) {
                uint4 o;
                o.x = __float_as_uint(w.x); o.y = __float_as_uint(w.y);
                o.z = __float_as_uint(w.z); o.w = __float_as_uint(w.w);
                *reinterpret_cast<uint4 *>(tile + bpos<true>(u_row, c) * 16) = o;
            } else {
                // fp32 A: lane-q holds k = 4q+t, 16+4q+t  -> unit c goes to q = c & 3, half c >> 2
                // bf16 A: lane-q holds k = 8q .. 8q+7      -> unit c goes to q = c >> 1, half c & 1
                const int bq = ABF ? (c >> 1) : (c & 3), bh = ABF ? (c & 1) : (c >> 2);
                char *dst = tile + bpos<false>(u_row, bq) * 16 + bh * 8;
                if constexpr (X3) {
                    uint32_t h0, m0, l0, h1, m1, l1;
                    split_bf16x3(w.x, w.y, h0, m0, l0);
                    split_bf16x3(w.z, w.w, h1, m1, l1);
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2 *>(dst + B_IMG * 16) = make_uint2(m0, m1);
                    *reinterpret_cast<uint2 *>(dst + 2 * B_IMG * 16) = make_uint2(l0, l1);
                } else {
                    uint2 o;
                    o.x = pack_bf16x2(w.x, w.y);
                    o.y = pack_bf16x2(w.z, w.w);
                    *reinterpret_cast<uint2 *>(dst) = o;
                }
            }
        }
    };
    auto publish = [&](int ch) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // this wave's ds_writes are done
        if (lane == 0) __hip_atomic_fetch_add(&full[ch % NB], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    // ---- consume side: this wave's RW rows
    constexpr int ESZ = ABF ? 2 : 4;
    const char *Ab = reinterpret_cast<const char *>(p.A) + (int64_t)s * p.a_sample_stride * ESZ;
    uint4 *Aw = lds + wave * (S * A_STAGE);             // this wave's private ring
    const uint32_t aw_addr = __builtin_amdgcn_readfirstlane(lds_addr_of(Aw));
    // DMA piece j: rows RPP*j .. of this wave's RW; lane l lands at chunk position l % ACH of row
    // RPP*j + l / ACH, so it fetches global chunk (l % ACH) ^ swizzle(row) (swizzle on the source;
    // ACH lanes read one contiguous 128-B (fp32) / 64-B (bf16) row segment).
    const char *asrc[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        int m = m0 + wave * RW + j * RPP + lane / ACH;
        m = m < p.M ? m : p.M - 1;                      // rows >= M: clamped, outputs never stored
        asrc[j] = Ab + (int64_t)m * p.lda * ESZ;
    }
    const int a_row = lane / ACH;                       // row inside a piece (== row & (RPP-1))
    const int a_chunk = ABF ? ((lane & 3) ^ ((0x78 >> (((a_row >> 2) & 3) * 2)) & 3)) : ((lane & 7) ^ (a_row & 7));
    const int a_kmax_bytes = (p.K - 16 / ESZ) * ESZ;    // last legal 16-B chunk of a row
    auto dma_A = [&](int stage, int kt) {
        int kb = (k_lo + kt * 32) * ESZ + 16 * a_chunk;
        kb = kb < a_kmax_bytes ? kb : a_kmax_bytes;     // k >= K: clamped (B is exactly 0 there)
        dma16xN<PW>(asrc, kb, aw_addr + (uint32_t)(stage * A_STAGE) * 16u);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bf16x3: the five SMALL partial products (each <= 2^-8 of the (h,h) one) go to an accumulator of their own and are added
    // once, at the end -- interleaved into the large sum they cost it five fp32 roundings at full magnitude per 32 k instead
    // of one (the dense path's two sweeps over K, bnn_dense.hip, to the same end; 4 TM TN more VGPRs here, where the wave
    // tile is small)
    f32x4 acc_s[X3 ? TM : 1][X3 ? TN : 1];
    if constexpr (X3) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc_s[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    auto mfma_step = [&](int stage, const uint4 *Bs) {
        const uint4 *As = Aw + stage * A_STAGE;
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
                const int row = a * 16 + fi;
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
        } else if constexpr (X3) {
            uint4 ah[TM], am[TM], al[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                const int row = a * 16 + fi;
                const uint4 a0 = As[row * 8 + (fq ^ (row & 7))];
                const uint4 a1 = As[row * 8 + ((fq + 4) ^ (row & 7))];
                split_bf16x3(__uint_as_float(a0.x), __uint_as_float(a0.y), ah[a].x, am[a].x, al[a].x);
                split_bf16x3(__uint_as_float(a0.z), __uint_as_float(a0.w), ah[a].y, am[a].y, al[a].y);
                split_bf16x3(__uint_as_float(a1.x), __uint_as_float(a1.y), ah[a].z, am[a].z, al[a].z);
                split_bf16x3(__uint_as_float(a1.z), __uint_as_float(a1.w), ah[a].w, am[a].w, al[a].w);
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int pos = bpos<false>(b * 16 + fi, fq);
                const bf16x8 bh = __builtin_bit_cast(bf16x8, Bs[pos]);
                const bf16x8 bm = __builtin_bit_cast(bf16x8, Bs[B_IMG + pos]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, Bs[2 * B_IMG + pos]);
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[a]), xm = __builtin_bit_cast(bf16x8, am[a]),
                                 xl = __builtin_bit_cast(bf16x8, al[a]);
                    f32x4 c = acc_s[a][b];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, bm, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bm, c, 0, 0, 0);
                    acc_s[a][b] = c;
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bh, acc[a][b], 0, 0, 0);
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
                const int row = a * 16 + fi;
                uint4 af;
                if constexpr (ABF) {
                    af = As[bpos<false>(row, fq)];
                } else {
                    const uint4 a0 = As[row * 8 + (fq ^ (row & 7))];
                    const uint4 a1 = As[row * 8 + ((fq + 4) ^ (row & 7))];
                    af.x = pack_bf16x2(__uint_as_float(a0.x), __uint_as_float(a0.y));
                    af.y = pack_bf16x2(__uint_as_float(a0.z), __uint_as_float(a0.w));
                    af.z = pack_bf16x2(__uint_as_float(a1.x), __uint_as_float(a1.y));
                    af.w = pack_bf16x2(__uint_as_float(a1.z), __uint_as_float(a1.w));
                }
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af),
                                                                        __builtin_bit_cast(bf16x8, bfr[b]),
                                                                        acc[a][b], 0, 0, 0);
            }
        }
    };

    // ---- VMEM queue of one wave (everything inline asm, so every wait below is ours):
    //   iteration ch:  R(ch+2) [LPU ops]  then, per step j of the chunk, D(kt + S - 1) [PW ops].
    //   * raw R(ch+1) (issued at the top of iteration ch-1) is needed by the draw: younger ops are
    //     that iteration's CH * PW pieces (only (S-1) * PW prologue pieces in iteration 0) and
    //     R(ch+2)                                   -> vmcnt(min(CH, S-1) * PW + LPU) is safe for both;
    //   * pieces D(kt) are needed by step kt = ch * CH + j: younger are (S-1) * PW pieces and, when
    //     D(kt) was issued in the previous iteration (j < S-1), R(ch+2)
    //                                                            -> vmcnt((S-1) * PW + (j < S-1 ? LPU : 0)).
    auto iteration = [&](int ch, Raw &cur, Raw &nxt) {
        // cur: raw of chunk ch+LA (in flight); nxt: free
        stamp(0);
        fetch_unit(nxt, ch + LA + 1);
        wait_raw(cur, std::integral_constant<int, WAIT_DRAW>{});
        stamp(1);
        // chunk drawn in this iteration; its buffer held chunk cd - NB, which every wave left at
        // least one iteration ago when NB > LA + 1 (slack instead of a per-chunk rendezvous)
        const int cd = ch + LA;
        if (cd < nch) {
            if (cd >= NB) handoff_ok &= lds_wait_ge(&freec[cd % NB], NW * (cd / NB), p.dev_err);  // its buffer is free again
            stamp(2);
            draw_unit(cur, cd);
            publish(cd);
        }
        stamp(3);
        handoff_ok &= lds_wait_ge(&full[ch % NB], NW * (ch / NB + 1), p.dev_err);   // chunk ch drawn by every wave
        stamp(4);
#ifndef BNN_NO_PIPE
        if constexpr (!F32 && ABF && CH == 2 && S >= 3) {          // (CH = 8, the head: measured, no gain)
            // The k-steps of a chunk, software-pipelined one deep: the fragments of step j+1 are requested around the MFMAs
            // of step j (a step was wait -> ds_read_b128s -> MFMAs, the LDS latency exposed at every step; the 10-wide head
            // has ONE MFMA per step).  The stage refill D(kt+3) moves behind the MFMAs of step j: it overwrites the stage
            // step j's fragments were read from, so it must not be issued before those reads have returned (an MFMA that
            // has been issued has its operands).  VMEM order per iteration: R, D(kt0+2), D(kt0+3) .. D(kt0+CH+1) as before.
            //   wait before the reads of step 0:      queue D(kt0) D(kt0+1) R D(kt0+2)   -> vmcnt(2 PW + LPU)
            //   ... of step 1:                        queue D(kt0+1) R D(kt0+2)          -> vmcnt(PW + LPU)
            //   ... of step j+1 >= 2:                 queue D(kt0+j+1) D(kt0+j+2)        -> vmcnt(PW)
            const int kt0 = ch * CH;
            const uint4 *Bc = Bs0 + (ch % NB) * B_CHUNK;
            uint4 bfr[2][TN], afr[2][TM];
            auto load_frags = [&](int j) {
                const uint4 *As = Aw + ((kt0 + j) % S) * A_STAGE;
#pragma unroll
                for (int b = 0; b < TN; ++b) bfr[j & 1][b] = Bc[j * B_SUB + bpos<false>(b * 16 + fi, fq)];
#pragma unroll
                for (int a = 0; a < TM; ++a) afr[j & 1][a] = As[bpos<false>(a * 16 + fi, fq)];
            };
            dma_A((kt0 + S - 1) % S, kt0 + S - 1);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 1) * PW + LPU) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            load_frags(0);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (j + 1 < CH) {
                    // S-stage ring: D(kt0+j+1) is followed by D(kt0+j+2) .. D(kt0+j+S-1), and by R while j + 1 <= S - 2
                    if (j + 1 <= S - 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 2) * PW + LPU) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 2) * PW) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    load_frags(j + 1);
                    // (no scheduling barrier here: hipcc sinks part of these reads below step j's first MFMAs to reuse
                    // registers -- 110 VGPRs, step -2.3 %; forcing them all ahead costs 122 VGPRs and gains only 1 %)
                }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afr[j & 1][a]),
                                                                            __builtin_bit_cast(bf16x8, bfr[j & 1][b]), acc[a][b], 0, 0, 0);
                if (j + 1 < CH) {
                    // step j's MFMAs have been issued, i.e. its fragments have left LDS: the stage may be refilled
                    __builtin_amdgcn_sched_barrier(0);
                    dma_A((kt0 + j + S) % S, kt0 + j + S);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else
#endif
        {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int kt = ch * CH + j;
            dma_A((kt + S - 1) % S, kt + S - 1);
            if (j < S - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 1) * PW + LPU) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 1) * PW) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(kt % S, Bs0 + (ch % NB) * B_CHUNK + j * B_SUB);
        }
        }
        if constexpr (STAMPS) { asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[TM - 1][TN - 1])); }
        stamp(5);
        // last read of chunk ch by this wave: release the buffer once the reads have returned
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(&freec[ch % NB], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    // prologue: draw chunks 0 .. LA-1, then put the queue in the state the top of iteration 0
    // expects: [R(LA), D(0) .. D(S-2)]
    if constexpr (LA == 2) {
        // everything the first iterations need is put in flight at once -- the raw (mu, rho) of chunks 0, 1
        // and LA, then the first S-1 A stages -- so the prologue pays ONE memory round trip; the queue ends
        // in the state iteration 0 expects, [R(LA), D(0) .. D(S-2)]
        Raw rawC;
#pragma unroll
        for (int i = 0; i < UPL; ++i) rawC.r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        fetch_unit(rawB, 0);
        fetch_unit(rawC, 1);
        fetch_unit(rawA, LA);
#pragma unroll
        for (int j = 0; j < S - 1; ++j) dma_A(j, j);
        wait_raw(rawB, std::integral_constant<int, 2 * LPU + (S - 1) * PW>{});
        draw_unit(rawB, 0);                     // nch >= 1
        publish(0);
        wait_raw(rawC, std::integral_constant<int, LPU + (S - 1) * PW>{});
        if (1 < nch) {
            draw_unit(rawC, 1);
            publish(1);
        }
    } else {
#pragma unroll
        for (int c0 = 0; c0 < LA; ++c0) {
            fetch_unit(rawB, c0);
            wait_raw(rawB, std::integral_constant<int, 0>{});
            if (c0 < nch) {
                draw_unit(rawB, c0);
                publish(c0);
            }
        }
        fetch_unit(rawA, LA);
#pragma unroll
        for (int j = 0; j < S - 1; ++j) dma_A(j, j);
    }
    stamp(11);                                                      // prologue done
    for (int ch = 0; ch < nch; ch += 2) {
        iteration(ch, rawA, rawB);
        if (ch + 1 < nch) iteration(ch + 1, rawB, rawA);
    }
    // nothing may still be writing this workgroup's LDS when it retires
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(12);                                                      // loop done
    if constexpr (X3) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] += acc_s[a][b];
    }

    // a hand-off wait that gave up has consumed an undrawn chunk or drawn into a busy buffer: the device error word is
    // set (bnn_check_device reports it) and this wave's results are not stored
    if (!__builtin_amdgcn_readfirstlane((int)handoff_ok)) return;

    // ---- epilogue: bias (drawn at kernel entry), activation, store
    const bool sampled_bias = (p.mu_b != nullptr);
    float *Yb = p.Y + (int64_t)s * p.y_sample_stride;
    uint16_t *Yh = reinterpret_cast<uint16_t *>(p.Y) + (int64_t)s * p.y_sample_stride;
    const bool y_bf16 = (p.flags & BNN_FLAG_Y_BF16) != 0;
    // Full interior tiles with 16-B aligned rows: the wave's RW x BN results go through its own (now idle)
    // A ring in LDS and leave as 16-B row chunks -- 3 (bf16) / 6 (fp32) wide stores per lane instead of
    // 4 * TM * TN scattered 2- / 4-byte ones.  Everything else takes the element-wise path below.
    {
        const int esz = y_bf16 ? 2 : 4;
        const int64_t ybase = (int64_t)s * p.y_sample_stride * esz;
        const bool wide = !(p.flags & kFlagStoreNCHW) && n0 + BN <= p.N && m0 + wave * RW + RW <= p.M &&
                          (p.ldy * esz) % 16 == 0 && ((reinterpret_cast<uintptr_t>(p.Y) + ybase) & 15u) == 0 && (n0 * esz) % 16 == 0;
        constexpr bool RING_FITS = S * A_STAGE * 16 >= RW * BN * 4;   // the wave's A ring must hold its output tile
        if (RING_FITS && wide) {
            char *T = reinterpret_cast<char *>(Aw);
            const int pitch = BN * esz;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                float bias = 0.f;
                if (sampled_bias) bias = bias_lds[b * 16 + fi];
                else if (p.bias) bias = p.bias[(int64_t)s * p.bias_sample_stride + n0 + b * 16 + fi];
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[a][b][r] + bias;
                        if (p.flags & BNN_FLAG_RELU) v = fmaxf(v, 0.f);
                        char *q = T + (a * 16 + fq * 4 + r) * pitch + (b * 16 + fi) * esz;
                        if (y_bf16) *reinterpret_cast<uint16_t *>(q) = f2bf(v);
                        else *reinterpret_cast<float *>(q) = v;
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");    // this wave's ds_writes before its ds_reads
            __builtin_amdgcn_wave_barrier();
            const int cpr = pitch / 16;                              // 16-B chunks per row
            char *Y8 = reinterpret_cast<char *>(p.Y) + ybase + ((int64_t)(m0 + wave * RW) * p.ldy + n0) * esz;
            for (int c = lane; c < RW * cpr; c += 64) {
                const int row = c / cpr, cc = c % cpr;
                *reinterpret_cast<uint4 *>(Y8 + (int64_t)row * p.ldy * esz + cc * 16) = *reinterpret_cast<const uint4 *>(T + row * pitch + cc * 16);
            }
            if constexpr (STAMPS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            stamp(13);
            return;
        }
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + b * 16 + fi;
        if (n >= p.N) continue;
        float bias = 0.f;
        if (sampled_bias)
            bias = bias_lds[b * 16 + fi];
        else if (p.bias)
            bias = p.bias[(int64_t)s * p.bias_sample_stride + n];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            if (p.flags & kFlagStoreNCHW) {
                // conv: rows are (image, pixel); 4 consecutive rows of a lane = consecutive pixels
                const int P = p.OH * p.OW;
                const int mb = m0 + wave * RW + a * 16 + fq * 4;
                int img = mb / P, pix = mb % P;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (mb + r < p.M) {
                        float v = acc[a][b][r] + bias;
                        if (p.flags & BNN_FLAG_RELU) v = fmaxf(v, 0.f);
                        Yb[((int64_t)img * p.O + n) * P + pix] = v;
                    }
                    if (++pix == P) { pix = 0; ++img; }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wave * RW + a * 16 + fq * 4 + r;
                if (m >= p.M) continue;
                float v = acc[a][b][r] + bias;
                if (p.flags & BNN_FLAG_RELU) v = fmaxf(v, 0.f);
                if (y_bf16) Yh[(int64_t)m * p.ldy + n] = f2bf(v);
                else Yb[(int64_t)m * p.ldy + n] = v;
            }
        }
    }
    if constexpr (STAMPS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    stamp(13);                                                      // epilogue stores retired
}

template <int NW, int RW, int BN, int CH, int S, int NB, int BMODE, int CP, bool ABF = false>
static void launch_sym(GemmParams &p, hipStream_t st)
{
    constexpr int BM = NW * RW;
    p.ntm = (p.M + BM - 1) / BM;
    p.ntn = (p.N + BN - 1) / BN;
    int64_t grid = (int64_t)p.ntn * p.ntm * p.S;
    p.xcd_a = 0;
    if (p.S % 8 == 0 && p.ntn >= 8) {
        // choose the XCD grid (xa sample groups x 8/xa panel groups) with the least L2 fill per XCD:
        // weights / xb + activations / xa (a shared input counts once whatever xa is)
        static const int force = [] { const char *e = getenv("BNN_XCD_A"); return e ? atoi(e) : -1; }();
        const double wbytes = 8.0 * p.N * p.K * (BMODE == B_PLAIN ? 0.5 * p.S : 1.0);
        const double abytes = (double)p.M * p.K * (ABF ? 2 : 4) * (p.a_sample_stride == 0 ? 1 : p.S);
        int best = 8;
        double best_cost = 1e300;
        for (int xa = 1; xa <= 8; xa *= 2) {
            const int xb = 8 / xa, ppg = (p.ntn + xb - 1) / xb;
            const double pad = (double)(p.S / xa) * ppg * 8 / ((double)p.S * p.ntn);
            if (pad > 1.2) continue;                                // too many idle workgroup slots
            const double cost = wbytes / xb + (p.a_sample_stride == 0 ? abytes : abytes / xa);
            if (cost < best_cost) { best_cost = cost; best = xa; }
        }
        if (force >= 0) best = force;
        if (best >= 1 && best < 8 && 8 % best == 0 && p.S % best == 0) {
            p.xcd_a = best;
            const int xb = 8 / best, ppg = (p.ntn + xb - 1) / xb;
            grid = (int64_t)8 * (p.S / best) * ppg * p.ntm;
        }
    }
    if constexpr (BMODE == B_SAMPLED && CP == BNN_COMPUTE_BF16 && BN > 16) {
        // diagnostic build with in-kernel stamps (tools/stamps.py): BNN_STAMPS=<device pointer>
        static unsigned long long *dbg = [] { const char *e = getenv("BNN_STAMPS"); return e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; }();
        if (dbg) {
            p.dbg = dbg;
            p.dbg_block = 100;
            hipLaunchKernelGGL((k_linear_sym<NW, RW, BN, CH, S, NB, BMODE, CP, ABF, true>), dim3((unsigned)grid), dim3(NW * 64), 0, st, p);
            return;
        }
    }
    if constexpr (BMODE == B_SAMPLED && BN == 16 && NW >= 4) {
        if (p.kl.nblocks > 0 && !p.kl.taken && grid + p.kl.nblocks < 0x7FFFFFFF) {
            p.gemm_grid = (int32_t)grid;
            grid += p.kl.nblocks;
            p.kl.taken = 1;
        }
    }
    hipLaunchKernelGGL((k_linear_sym<NW, RW, BN, CH, S, NB, BMODE, CP, ABF>), dim3((unsigned)grid), dim3(NW * 64), 0, st, p);
}

// BNN_F32_MFMA=native: the fp32 mode on v_mfma_f32_16x16x4_f32 (exact fp32 products); default: bf16x3
static bool f32x3_enabled()
{
    static const bool on = [] { const char *e = getenv("BNN_F32_MFMA"); return !(e && e[0] == 'n'); }();
    return on;
}

template <int BMODE, int CP>
static void select_pc(GemmParams &p, hipStream_t st)
{
    static const int tile = [] { const char *e = getenv("BNN_TILE"); return e ? atoi(e) : 0; }();
    if constexpr (CP == BNN_COMPUTE_BF16) {
        if (p.flags & BNN_FLAG_X_BF16) {
            // bf16 activations: 2-KB stages -> a 3-stage ring and 4 chunk buffers fit easily
            if (p.N <= 16) {
                // head: 128 x 16 tiles, 256-k chunks (5 hand-offs instead of 19 at K = 1200; measured
                // against 64 x 16 / 64-k chunks: step 0.1203 -> 0.1168 ms)
                launch_sym<8, 16, 16, 8, 3, 4, BMODE, CP, true>(p, st);
            } else {
                // (a 4-stage A ring, one more k-step of DMA lead: measured equal, 0.0930 vs 0.0930 ms)
                launch_sym<16, 32, 48, 2, 3, 4, BMODE, CP, true>(p, st);
            }
            return;
        }
    }
    if (p.N <= 16) {
        // fp32 mode: the bf16 head's shape (128 x 16 tiles, 256-k chunks) on bf16x3 splits (fp32 step 0.1927 -> 0.1901 ms)
        if (CP == BNN_COMPUTE_F32 && f32x3_enabled()) launch_sym<8, 16, 16, 8, 3, 4, BMODE, kComputeBf16x3>(p, st);
        else launch_sym<4, 16, 16, 2, 3, 4, BMODE, CP>(p, st);
    } else if (CP == BNN_COMPUTE_F32 && f32x3_enabled() && tile == 0) {
        // fp32 results from the bf16 MFMA (kComputeBf16x3): 256 x 80 tiles, 64-k chunks, 2 chunk buffers (three B images)
        // (8 waves x 32 rows, which halves the B-fragment LDS reads, and 32-k chunks were measured: all within 4 %; 512 x 48
        // tiles with 32-k chunks, which draw every weight once: 6 % slower)
        launch_sym<16, 16, 80, 2, 2, 2, BMODE, kComputeBf16x3>(p, st);
    } else if ((CP == BNN_COMPUTE_F32 && tile != 512) || tile == 256) {
        // fp32 is MFMA-bound: 256 x 80 tiles fill the chip (240 workgroups at the BASELINE shape)
        // (64-k chunks, 2-stage A ring: 0.2876 -> 0.2749 ms per fp32 step against 32-k chunks / 3 stages; the kernel
        // runs at 80 TFLOP/s = 51 % of the fp32 MFMA peak, 70 % of its per-SIMD MFMA + draw-VALU cycle count)
        if constexpr (CP == BNN_COMPUTE_F32) launch_sym<16, 16, 80, 2, 2, 4, BMODE, CP>(p, st);
        else launch_sym<16, 16, 80, 1, 3, 4, BMODE, CP>(p, st);
    } else {
        // bf16 is draw-bound: 512 x 48 tiles draw every weight of a sample exactly once
        launch_sym<16, 32, 48, 2, 2, (CP == BNN_COMPUTE_F32 ? 2 : 4), BMODE, CP>(p, st);
    }
}

// Called by linear_common (bnn_gemm.hip) when operands are 16-B aligned and K % 4 == 0.
// Input gradient gx = gy . W_s (W_s re-drawn from the forward's key, used transposed): the same
// draw-paced pipeline with p.A = gy, p.N = columns of W, p.K = rows of W.
int dispatch_linear_dgrad(GemmParams &p, int compute, hipStream_t st, const char *who)
{
    if (compute == BNN_COMPUTE_F32) {
        if (p.N <= 16) launch_sym<4, 16, 16, 2, 3, 4, B_SAMPLED_T, BNN_COMPUTE_F32>(p, st);
        else launch_sym<16, 16, 80, 1, 3, 4, B_SAMPLED_T, BNN_COMPUTE_F32>(p, st);
    } else if (compute == BNN_COMPUTE_BF16) {
        if (p.flags & BNN_FLAG_X_BF16) {
            if (p.N <= 16) launch_sym<4, 16, 16, 2, 3, 4, B_SAMPLED_T, BNN_COMPUTE_BF16, true>(p, st);
            else launch_sym<16, 32, 48, 2, 3, 4, B_SAMPLED_T, BNN_COMPUTE_BF16, true>(p, st);
        } else {
            if (p.N <= 16) launch_sym<4, 16, 16, 2, 3, 4, B_SAMPLED_T, BNN_COMPUTE_BF16>(p, st);
            else launch_sym<16, 32, 48, 2, 2, 4, B_SAMPLED_T, BNN_COMPUTE_BF16>(p, st);
        }
    } else {
        set_error("%s: unknown compute mode %d", who, compute);
        return BNN_E_DTYPE;
    }
    return check_launch(who);
}

int dispatch_linear_v2(GemmParams &p, bool sampled, int compute, hipStream_t st, const char *who)
{
    if (compute == BNN_COMPUTE_F32) {
        if (sampled) select_pc<B_SAMPLED, BNN_COMPUTE_F32>(p, st);
        else select_pc<B_PLAIN, BNN_COMPUTE_F32>(p, st);
    } else if (compute == BNN_COMPUTE_BF16) {
        if (sampled) select_pc<B_SAMPLED, BNN_COMPUTE_BF16>(p, st);
        else select_pc<B_PLAIN, BNN_COMPUTE_BF16>(p, st);
    } else {
        set_error("%s: unknown compute mode %d", who, compute);
        return BNN_E_DTYPE;
    }
    return check_launch(who);
}

}  // namespace bnn
