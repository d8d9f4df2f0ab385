// bnn_gemm.hip -- K2: the contraction that consumes the posterior draw.
//
//   y[s] = A[s] . W_s^T + b_s          A: dense rows (linear) or im2col of an NCHW tensor
//   W_s  = mu + sigma(rho) * eps_s     drawn INSIDE the B-operand loader, never stored
//
// One "NT" MFMA GEMM (both operands K-contiguous) written for gfx950:
//   * v_mfma_f32_16x16x4_f32 (exact fp32) or v_mfma_f32_16x16x32_bf16 (fp32 accumulate);
//   * LDS tiles are [rows][4 chunks of 16 B]; MFMA lane (i = l & 15, q = l >> 4) reads
//     chunk q of row i with one ds_read_b128.  The k order inside a tile is therefore a
//     permutation of the natural one -- the same permutation for A and B, so the sum is
//     unchanged.  Chunks are XOR-swizzled by h((row >> 2) & 3), h = {0, 2, 3, 1}, which makes
//     every ds_read_b128 lane group hit 16 distinct 16-B slots (conflict-free, derived in
//     DESIGN.md);
//   * the B loader fetches mu / rho with 16-B loads, runs Philox4x32-10 + Box-Muller and
//     softplus in registers and writes the drawn weights to LDS (fp32 or bf16);
//   * global -> register prefetch of tile k+1 is issued before the MFMAs of tile k,
//     sampled and written to the other LDS buffer after them: one barrier per tile;
//   * all MC samples of a layer run in ONE grid (sample = part of the block index), and
//     the block index is decoded XCD-aware: workgroups that share a (mu, rho) column
//     panel have equal blockIdx % 8, i.e. share one XCD's L2.
#include <cstdlib>

#include "bnn_device.hpp"
#include "bnn_gemm_params.hpp"

namespace bnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int swz(int row, int chunk)
{
    // h = {0, 2, 3, 1} packed two bits per entry
    return chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
}

template <int BM, int BN, int WM, int WN, int A_MODE, int B_MODE, int COMPUTE>
__global__ __launch_bounds__(WM * WN * 64) void k_gemm_nt(const GemmParams p)
{
    constexpr int NT = WM * WN * 64;
    constexpr int KPC = (COMPUTE == BNN_COMPUTE_F32) ? 4 : 8;   // k per 16-B chunk
    constexpr int BK = 4 * KPC;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of 16x16");
    constexpr int A_CHUNKS = BM * 4, B_CHUNKS = BN * 4;
    constexpr int A_PER = (A_CHUNKS + NT - 1) / NT, B_PER = (B_CHUNKS + NT - 1) / NT;

    __shared__ __attribute__((aligned(16))) uint4 lds[2 * (A_CHUNKS + B_CHUNKS)];
    uint4 *As0 = lds, *Bs0 = lds + 2 * A_CHUNKS;

    // ---- block decode (XCD-aware) -------------------------------------------------
    const int L = blockIdx.x;
    const int xcd = L & 7, i_in = L >> 3;
    const int per_panel = p.ntm * p.S;
    const int panel = (i_in / per_panel) * 8 + xcd;          // panel = (group, n-tile)
    if (panel >= p.ntn * p.G) return;
    const int rem = i_in % per_panel;
    const int mt = rem % p.ntm, s = rem / p.ntm;
    const int g = panel / p.ntn, nt = panel % p.ntn;
    const int m0 = mt * BM, n0 = nt * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const uint32_t sample = p.rng_w.sample0 + (uint32_t)s;
    uint32_t edev_w = 0;
    if constexpr (B_MODE == B_SAMPLED) edev_w = rng_epoch_dev(p.rng_w);

    const float *Ab = p.A + (int64_t)s * p.a_sample_stride;
    const int64_t nrow0 = (int64_t)g * p.N;                  // first weight row of this group

    // ---- per-thread chunk ownership -------------------------------------------------
    // A chunk id -> (row, c); rows are fixed across the K loop.
    int64_t a_off[A_PER];       // dense: row offset; im2col: image base offset
    int a_ih0[A_PER], a_iw0[A_PER];
    bool a_ok[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int id = tid + i * NT;
        const int row = id >> 2;
        const int m = m0 + row;
        a_ok[i] = (id < A_CHUNKS) && (m < p.M);
        a_off[i] = 0; a_ih0[i] = 0; a_iw0[i] = 0;
        if (a_ok[i]) {
            if constexpr (A_MODE == A_DENSE) {
                a_off[i] = (int64_t)m * p.lda;
            } else {
                const int ow = m % p.OW, t = m / p.OW;
                const int oh = t % p.OH, b = t / p.OH;
                a_off[i] = ((int64_t)b * p.C + (int64_t)g * p.Cg) * p.H * p.W;
                a_ih0[i] = oh * p.sh - p.ph;
                a_iw0[i] = ow * p.sw - p.pw;
            }
        }
    }

    // staging registers
    float av[A_PER][KPC];
    float bm[B_PER][KPC], br[B_PER][KPC];

    auto load_A = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int id = tid + i * NT;
            const int c = id & 3;
            const int kb = k0 + c * KPC;
#pragma unroll
            for (int j = 0; j < KPC; ++j) av[i][j] = 0.f;
            if (!a_ok[i]) continue;
            if constexpr (A_MODE == A_DENSE) {
                if (p.vecA && kb + KPC <= p.K) {
#pragma unroll
                    for (int v = 0; v < KPC / 4; ++v) {
                        const float4 t = *reinterpret_cast<const float4 *>(Ab + a_off[i] + kb + 4 * v);
                        av[i][4 * v] = t.x; av[i][4 * v + 1] = t.y; av[i][4 * v + 2] = t.z; av[i][4 * v + 3] = t.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < KPC; ++j)
                        if (kb + j < p.K) av[i][j] = Ab[a_off[i] + kb + j];
                }
            } else {
                const int khw = p.KH * p.KW;
#pragma unroll
                for (int j = 0; j < KPC; ++j) {
                    const int k = kb + j;
                    if (k < p.K) {
                        const int ci = k / khw, r = k % khw;
                        const int kh = r / p.KW, kw = r % p.KW;
                        const int ih = a_ih0[i] + kh * p.dh, iw = a_iw0[i] + kw * p.dw;
                        if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                            av[i][j] = Ab[a_off[i] + ((int64_t)ci * p.H + ih) * p.W + iw];
                    }
                }
            }
        }
    };

    auto load_B = [&](int k0) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int id = tid + i * NT;
            const int row = id >> 2, c = id & 3;
            const int n = n0 + row;
            const int kb = k0 + c * KPC;
#pragma unroll
            for (int j = 0; j < KPC; ++j) { bm[i][j] = 0.f; br[i][j] = 0.f; }
            if (id >= B_CHUNKS || n >= p.N) continue;
            const int64_t e0 = (nrow0 + n) * (int64_t)p.K + kb;
            const float *src_m = (B_MODE == B_SAMPLED) ? p.mu : p.Bw + (int64_t)s * p.b_sample_stride;
            if (p.vecB && kb + KPC <= p.K) {
#pragma unroll
                for (int v = 0; v < KPC / 4; ++v) {
                    const float4 t = *reinterpret_cast<const float4 *>(src_m + e0 + 4 * v);
                    bm[i][4 * v] = t.x; bm[i][4 * v + 1] = t.y; bm[i][4 * v + 2] = t.z; bm[i][4 * v + 3] = t.w;
                    if constexpr (B_MODE == B_SAMPLED) {
                        const float4 u = *reinterpret_cast<const float4 *>(p.rho + e0 + 4 * v);
                        br[i][4 * v] = u.x; br[i][4 * v + 1] = u.y; br[i][4 * v + 2] = u.z; br[i][4 * v + 3] = u.w;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < KPC; ++j)
                    if (kb + j < p.K) {
                        bm[i][j] = src_m[e0 + j];
                        if constexpr (B_MODE == B_SAMPLED) br[i][j] = p.rho[e0 + j];
                    }
            }
        }
    };

    // draw the weights of the staged B chunks (in place: bm <- w)
    auto sample_B = [&](int k0) {
        if constexpr (B_MODE == B_SAMPLED) {
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int id = tid + i * NT;
                const int row = id >> 2, c = id & 3;
                const int n = n0 + row;
                const int kb = k0 + c * KPC;
                if (id >= B_CHUNKS || n >= p.N || kb >= p.K) continue;
                const int64_t e0 = (nrow0 + n) * (int64_t)p.K + kb;
                if (p.vecB && kb + KPC <= p.K) {
#pragma unroll
                    for (int v = 0; v < KPC / 4; ++v) {
                        const float4 z = eps4(p.rng_w, edev_w, (uint32_t)((e0 >> 2) + v), sample);
                        bm[i][4 * v] = fmaf(sigma_draw(br[i][4 * v]), z.x, bm[i][4 * v]);
                        bm[i][4 * v + 1] = fmaf(sigma_draw(br[i][4 * v + 1]), z.y, bm[i][4 * v + 1]);
                        bm[i][4 * v + 2] = fmaf(sigma_draw(br[i][4 * v + 2]), z.z, bm[i][4 * v + 2]);
                        bm[i][4 * v + 3] = fmaf(sigma_draw(br[i][4 * v + 3]), z.w, bm[i][4 * v + 3]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < KPC; ++j)
                        if (kb + j < p.K)
                            bm[i][j] = fmaf(sigma_draw(br[i][j]), eps1(p.rng_w, edev_w, (uint64_t)(e0 + j), sample), bm[i][j]);
                }
            }
        }
    };

    auto pack = [&](const float (&v)[KPC]) -> uint4 {
        uint4 o;
        if constexpr (COMPUTE == BNN_COMPUTE_F32) {
            o.x = __float_as_uint(v[0]); o.y = __float_as_uint(v[1]);
            o.z = __float_as_uint(v[2]); o.w = __float_as_uint(v[3]);
        } else {
            o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
            o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        }
        return o;
    };

    auto store_tiles = [&](int buf) {
        uint4 *As = As0 + buf * A_CHUNKS, *Bs = Bs0 + buf * B_CHUNKS;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int id = tid + i * NT;
            if (id < A_CHUNKS) {
                const int row = id >> 2, c = id & 3;
                As[row * 4 + swz(row, c)] = pack(av[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int id = tid + i * NT;
            if (id < B_CHUNKS) {
                const int row = id >> 2, c = id & 3;
                Bs[row * 4 + swz(row, c)] = pack(bm[i]);
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    const int fi = lane & 15, fq = lane >> 4;

    load_A(0);
    load_B(0);
    sample_B(0);
    store_tiles(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            load_A((kt + 1) * BK);
            load_B((kt + 1) * BK);
        }
        const uint4 *As = As0 + buf * A_CHUNKS, *Bs = Bs0 + buf * B_CHUNKS;
        uint4 af[TM], bf[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int row = wm * WTM + a * 16 + fi;
            af[a] = As[row * 4 + swz(row, fq)];
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int row = wn * WTN + b * 16 + fi;
            bf[b] = Bs[row * 4 + swz(row, fq)];
        }
        if constexpr (COMPUTE == BNN_COMPUTE_F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const uint32_t ua = j == 0 ? af[a].x : j == 1 ? af[a].y : j == 2 ? af[a].z : af[a].w;
                        const uint32_t ub = j == 0 ? bf[b].x : j == 1 ? bf[b].y : j == 2 ? bf[b].z : bf[b].w;
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(ua), __uint_as_float(ub),
                                                                         acc[a][b], 0, 0, 0);
                    }
        } else {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[a]),
                                                                        __builtin_bit_cast(bf16x8, bf[b]),
                                                                        acc[a][b], 0, 0, 0);
        }
        if (more) {
            sample_B((kt + 1) * BK);
            store_tiles(buf ^ 1);
        }
        __syncthreads();
    }

    // ---- epilogue: bias (drawn per column), activation, store -------------------------
    uint32_t edev_b = 0;
    const bool sampled_bias = (p.mu_b != nullptr);
    if (sampled_bias) edev_b = rng_epoch_dev(p.rng_b);
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + wn * WTN + b * 16 + fi;
        if (n >= p.N) continue;
        const int64_t ng = nrow0 + n;
        float bias = 0.f;
        if (sampled_bias)
            bias = fmaf(sigma_draw(p.rho_b[ng]), eps1(p.rng_b, edev_b, (uint64_t)ng, p.rng_b.sample0 + (uint32_t)s), p.mu_b[ng]);
        else if (p.bias)
            bias = p.bias[(int64_t)s * p.bias_sample_stride + ng];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * WTM + a * 16 + fq * 4 + r;
                if (m >= p.M) continue;
                float v = acc[a][b][r] + bias;
                if (p.flags & BNN_FLAG_RELU) v = fmaxf(v, 0.f);
                float *Yb = p.Y + (int64_t)s * p.y_sample_stride;
                if constexpr (A_MODE == A_DENSE) {
                    Yb[(int64_t)m * p.ldy + ng] = v;
                } else {
                    const int hw = p.OH * p.OW;
                    const int bimg = m / hw, pix = m % hw;
                    Yb[((int64_t)bimg * p.O + ng) * hw + pix] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ host side
struct TileCfg { int bm, bn; };

template <int BM, int BN, int WM, int WN, int AM, int BMODE, int CP>
static void launch_cfg(GemmParams &p, hipStream_t st)
{
    p.ntm = (p.M + BM - 1) / BM;
    p.ntn = (p.N + BN - 1) / BN;
    const int panels = p.ntn * p.G;
    const int64_t grid = (int64_t)8 * ((panels + 7) / 8) * p.ntm * p.S;
    hipLaunchKernelGGL((k_gemm_nt<BM, BN, WM, WN, AM, BMODE, CP>), dim3((unsigned)grid), dim3(WM * WN * 64), 0, st, p);
}

// Tile choice: big tiles amortise the weight draw over more rows; small ones fill the
// 256 CUs when the problem is small.  (Round 1: three shapes; tuning is tracked in DESIGN.md.)
template <int AM, int BMODE, int CP>
static void launch_select(GemmParams &p, hipStream_t st)
{
    const int64_t tiles_big = (int64_t)((p.M + 255) / 256) * ((p.N + 79) / 80) * p.S * p.G;
    if (p.N <= 16) {
        launch_cfg<64, 16, 4, 1, AM, BMODE, CP>(p, st);
    } else if (p.M >= 256 && p.N >= 80 && tiles_big >= 128) {
        launch_cfg<256, 80, 8, 1, AM, BMODE, CP>(p, st);
    } else {
        launch_cfg<64, 64, 2, 2, AM, BMODE, CP>(p, st);
    }
}

template <int AM>
static int dispatch(GemmParams &p, bool sampled, int compute, hipStream_t st, const char *who)
{
    if (compute == BNN_COMPUTE_F32) {
        if (sampled) launch_select<AM, B_SAMPLED, BNN_COMPUTE_F32>(p, st);
        else launch_select<AM, B_PLAIN, BNN_COMPUTE_F32>(p, st);
    } else if (compute == BNN_COMPUTE_BF16) {
        if (sampled) launch_select<AM, B_SAMPLED, BNN_COMPUTE_BF16>(p, st);
        else launch_select<AM, B_PLAIN, BNN_COMPUTE_BF16>(p, st);
    } else {
        set_error("%s: unknown compute mode %d", who, compute);
        return BNN_E_DTYPE;
    }
    return check_launch(who);
}

// Per-device scratch registered by the host (bnn_set_workspace): [error word + reserved: 64 KiB][slabs: rest].
static void *g_ws[64];
static int64_t g_ws_bytes[64];
constexpr int64_t kTicketBytes = 64 * 1024;

// The workspace of the device the launch goes to.  Every launch of this library goes to the CURRENT device (the host
// side makes the operands' device current around each call: _lib._StreamPtr), so that is the device whose scratch the
// kernel may touch.
void fill_workspace(GemmParams &p)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || !g_ws[dev] || g_ws_bytes[dev] <= kTicketBytes) return;
    p.dev_err = reinterpret_cast<unsigned *>(g_ws[dev]);
    p.ws_slabs = reinterpret_cast<float *>(reinterpret_cast<char *>(g_ws[dev]) + kTicketBytes);
    p.ws_slab_bytes = g_ws_bytes[dev] - kTicketBytes;
}

static inline bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }
static inline bool al4(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 3u) == 0; }

static int linear_common(const float *x, int64_t x_sample_stride, int64_t ldx,
                         const float *w, int64_t w_sample_stride, const float *b, int64_t b_sample_stride,
                         const float *mu_w, const float *rho_w, const float *mu_b, const float *rho_b,
                         float *y, int64_t y_sample_stride, int64_t ldy, int64_t M, int64_t N, int64_t K,
                         int nsamples, const bnn_rng_t *rng_w, const bnn_rng_t *rng_b, bool sampled,
                         int compute, int flags, void *stream, const char *who, KlPiggy *kl = nullptr)
{
    if (M == 0 && N >= 1 && K >= 1 && nsamples >= 1) return BNN_OK;     // empty batch: nothing to do (x / y may be NULL)
    if (!x || !y || (sampled ? (!mu_w || !rho_w) : !w)) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (sampled && ((mu_b == nullptr) != (rho_b == nullptr))) { set_error("%s: mu_b / rho_b must both be given or both NULL", who); return BNN_E_NULL; }
    if (M < 0 || N < 1 || K < 1 || nsamples < 1 || ldx < K || ldy < N) { set_error("%s: bad extent (M=%lld N=%lld K=%lld S=%d ldx=%lld ldy=%lld)", who, (long long)M, (long long)N, (long long)K, nsamples, (long long)ldx, (long long)ldy); return BNN_E_SHAPE; }
    if (M > 0x7FFFFFFF || N > 0x7FFFFFFF || K > 0x7FFFFFFF || N * K > ((int64_t)1 << 34)) { set_error("%s: extent too large", who); return BNN_E_RANGE; }
    const bool xh = (flags & BNN_FLAG_X_BF16) != 0, yh = (flags & BNN_FLAG_Y_BF16) != 0;
    if (!(xh ? (reinterpret_cast<uintptr_t>(x) & 1u) == 0 : al4(x)) || !(yh ? (reinterpret_cast<uintptr_t>(y) & 1u) == 0 : al4(y))) { set_error("%s: misaligned pointer", who); return BNN_E_ALIGN; }
    if (xh || yh) {
        // bf16 activations exist only on the fast bf16-compute path
        const bool ok = compute == BNN_COMPUTE_BF16 && (K % 4 == 0) &&
                        (!xh || (al16(x) && K % 8 == 0 && ldx % 8 == 0 && x_sample_stride % 8 == 0)) &&
                        (xh || (al16(x) && ldx % 4 == 0 && x_sample_stride % 4 == 0)) &&
                        (sampled ? (al16(mu_w) && al16(rho_w)) : (al16(w) && w_sample_stride % 4 == 0));
        if (!ok) { set_error("%s: BNN_FLAG_X_BF16 / Y_BF16 need bf16 compute, 16-B aligned operands, K %% 8 == 0", who); return BNN_E_UNSUPPORTED; }
    }
    if (sampled) {
        int rc = check_rng(rng_w, nsamples);
        if (rc) { set_error("%s: bad rng_w", who); return rc; }
        if (mu_b) { rc = check_rng(rng_b, nsamples); if (rc) { set_error("%s: bad rng_b", who); return rc; } }
    }
    if (M == 0) return BNN_OK;
    GemmParams p{};
    p.A = x; p.a_sample_stride = x_sample_stride; p.lda = ldx;
    p.Bw = w; p.b_sample_stride = w_sample_stride; p.mu = mu_w; p.rho = rho_w;
    p.bias = sampled ? nullptr : b; p.bias_sample_stride = b_sample_stride;
    p.mu_b = sampled ? mu_b : nullptr; p.rho_b = sampled ? rho_b : nullptr;
    p.Y = y; p.y_sample_stride = y_sample_stride; p.ldy = ldy; p.O = (int32_t)N;
    p.M = (int32_t)M; p.N = (int32_t)N; p.K = (int32_t)K; p.S = nsamples; p.G = 1; p.flags = flags;
    p.vecA = xh ? 1 : (al16(x) && (ldx % 4 == 0) && (x_sample_stride % 4 == 0));
    p.vecB = (K % 4 == 0) && (sampled ? (al16(mu_w) && al16(rho_w)) : (al16(w) && w_sample_stride % 4 == 0));
    if (sampled) { p.rng_w = make_rng(rng_w); p.rng_b = make_rng(mu_b ? rng_b : nullptr); }
    // sampler-paced kernel (bnn_linear.hip) whenever 16-B loads are legal; BNN_LINEAR_KERNEL=v1
    // forces the generic tile kernel (A/B comparisons).
    static const bool force_v1 = [] { const char *e = getenv("BNN_LINEAR_KERNEL"); return e && e[0] == 'v' && e[1] == '1'; }();
    if (p.vecA && p.vecB && !force_v1) {
        fill_workspace(p);
        if (kl) p.kl = *kl;                                     // carried by the narrow-layer launch if that is the one taken
        const int rc = dispatch_linear_v2(p, sampled, compute, (hipStream_t)stream, who);
        if (kl) kl->taken = p.kl.taken;
        return rc;
    }
    return dispatch<A_DENSE>(p, sampled, compute, (hipStream_t)stream, who);
}

// ---- explicit im2col panel for the fast conv path ---------------------------------------------
// out[(s, b, oh, ow)][(c, kh, kw)] = x[s][b][c][oh*sh - ph + kh*dh][ow*sw - pw + kw*dw] (0 outside), the
// column order of the weight's (O, C, KH, KW) rows, bf16 (bf16 compute) or fp32.  One thread per 8
// consecutive columns (K % 8 == 0): 16-B / 32-B stores, coalesced along the row.  The panel then feeds
// the draw-paced linear kernel (LDS-DMA A rings), whose epilogue stores NCHW.
template <bool BF>
__global__ __launch_bounds__(256) void k_im2col(const float *__restrict__ x, int64_t x_sample_stride, void *__restrict__ out,
                                                int B, int C, int H, int W, int OH, int OW, int KH, int KW,
                                                int sh_, int sw_, int ph, int pw, int dh, int dw, int K, int64_t groups8)
{
    const int KQ = K / 8, P = OH * OW, KHW = KH * KW;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < groups8; id += (int64_t)gridDim.x * 256) {
        const int kg = (int)(id % KQ);
        const int64_t mrow = id / KQ;
        const int s = (int)(mrow / ((int64_t)B * P));
        const int ml = (int)(mrow % ((int64_t)B * P));
        const int b = ml / P, pix = ml % P;
        const int oh = pix / OW, ow = pix % OW;
        int k = kg * 8;
        int c = k / KHW, rem = k % KHW;
        int kh = rem / KW, kw = rem % KW;
        const float *xb = x + (int64_t)s * x_sample_stride + (int64_t)b * C * H * W;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ih = oh * sh_ - ph + kh * dh, iw = ow * sw_ - pw + kw * dw;
            v[j] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? xb[((int64_t)c * H + ih) * W + iw] : 0.f;
            if (++kw == KW) { kw = 0; if (++kh == KH) { kh = 0; ++c; } }
        }
        if constexpr (BF) {
            uint4 o;
            o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
            o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
            *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(out) + mrow * K + k) = o;
        } else {
            float *q = reinterpret_cast<float *>(out) + mrow * K + k;
            *reinterpret_cast<float4 *>(q) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>(q + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// q = n / d, r = n % d for 0 <= n < 2^22 through the float unit (the reciprocal comes from the host) with an exact
// integer correction -- ~6 instructions instead of the ~25 of an integer division by a run-time value.
__device__ __forceinline__ void fdivmod(int n, int d, float inv_d, int &q, int &r)
{
    q = (int)((float)n * inv_d);
    r = n - q * d;
    if (r < 0) { r += d; --q; }
    if (r >= d) { r -= d; ++q; }
}

// Panel of ONE image per workgroup, the image staged in LDS: every input element is read from memory once
// (coalesced float4s) instead of once per tap that uses it (KH * KW times, as scattered 4-byte loads -- what
// bounded the flat kernel above: 57 us for the 75 MB panel of the CIFAR shape).  Needs C * H * W floats of LDS
// (<= 64 KB) and indices below 2^22; the flat kernel covers the rest.
template <bool BF>
__global__ __launch_bounds__(256) void k_im2col_img(const float *__restrict__ x, int64_t x_sample_stride, void *__restrict__ out,
                                                    int B, int C, int H, int W, int OH, int OW, int KH, int KW,
                                                    int sh_, int sw_, int ph, int pw, int dh, int dw, int K,
                                                    float inv_KQ, float inv_OW, float inv_KHW, float inv_KW)
{
    extern __shared__ float img[];
    const int chw = C * H * W, P = OH * OW, KQ = K / 8, KHW = KH * KW;
    const int s = blockIdx.x / B, b = blockIdx.x % B;
    const float *xb = x + (int64_t)s * x_sample_stride + (int64_t)b * chw;
    if ((chw & 3) == 0 && (reinterpret_cast<uintptr_t>(xb) & 15u) == 0) {
        for (int i = threadIdx.x; i < chw / 4; i += 256) reinterpret_cast<float4 *>(img)[i] = reinterpret_cast<const float4 *>(xb)[i];
    } else {
        for (int i = threadIdx.x; i < chw; i += 256) img[i] = xb[i];
    }
    __syncthreads();
    const int64_t row0 = (int64_t)blockIdx.x * P;
    for (int oct = threadIdx.x; oct < P * KQ; oct += 256) {
        int pix, kg, oh, ow, c, rem, kh, kw;
        fdivmod(oct, KQ, inv_KQ, pix, kg);
        fdivmod(pix, OW, inv_OW, oh, ow);
        const int k = kg * 8;
        fdivmod(k, KHW, inv_KHW, c, rem);
        fdivmod(rem, KW, inv_KW, kh, kw);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ih = oh * sh_ - ph + kh * dh, iw = ow * sw_ - pw + kw * dw;
            v[j] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? img[(c * H + ih) * W + iw] : 0.f;
            if (++kw == KW) { kw = 0; if (++kh == KH) { kh = 0; ++c; } }
        }
        const int64_t o = (row0 + pix) * K + k;
        if constexpr (BF) {
            uint4 q;
            q.x = pack_bf16x2(v[0], v[1]); q.y = pack_bf16x2(v[2], v[3]);
            q.z = pack_bf16x2(v[4], v[5]); q.w = pack_bf16x2(v[6], v[7]);
            *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(out) + o) = q;
        } else {
            float *q = reinterpret_cast<float *>(out) + o;
            *reinterpret_cast<float4 *>(q) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>(q + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// Launch of the panel kernel (returns after the launch; the caller checks it).
static void launch_im2col(const float *x, int64_t x_sample_stride, void *panel, const bnn_conv2d_shape_t *sh, int64_t OH, int64_t OW,
                          int64_t K, int nsx, int64_t M, bool bf, hipStream_t st)
{
    const int64_t rows = (int64_t)nsx * M;
    {
        const int64_t chw = (int64_t)sh->C * sh->H * sh->W, P = OH * OW, images = (int64_t)nsx * sh->B;
        if (chw * 4 <= 64 * 1024 && P * (K / 8) < (1 << 22) && K < (1 << 22) && images <= 0x7FFFFFFF) {
            const unsigned lds = (unsigned)(chw * 4);
            const float iKQ = 1.0f / (float)(K / 8), iOW = 1.0f / (float)OW, iKHW = 1.0f / (float)(sh->KH * sh->KW), iKW = 1.0f / (float)sh->KW;
            if (bf) hipLaunchKernelGGL((k_im2col_img<true>), dim3((unsigned)images), dim3(256), lds, st, x, x_sample_stride, panel, sh->B, sh->C, sh->H, sh->W, (int)OH, (int)OW, sh->KH, sh->KW, sh->stride_h, sh->stride_w, sh->pad_h, sh->pad_w, sh->dil_h, sh->dil_w, (int)K, iKQ, iOW, iKHW, iKW);
            else hipLaunchKernelGGL((k_im2col_img<false>), dim3((unsigned)images), dim3(256), lds, st, x, x_sample_stride, panel, sh->B, sh->C, sh->H, sh->W, (int)OH, (int)OW, sh->KH, sh->KW, sh->stride_h, sh->stride_w, sh->pad_h, sh->pad_w, sh->dil_h, sh->dil_w, (int)K, iKQ, iOW, iKHW, iKW);
            return;
        }
    }
    // (measured and rejected: one workgroup per row with a compile-time 3 x 3 window -- no gain at K = 1152, 1.6x
    // slower at K = 576; a 2-D launch with float-reciprocal index decode -- same at K = 1152, 1.3x slower at
    // K = 576: the kernel is bound by its 8 scattered loads per thread, not by the index arithmetic)
    const int64_t groups8 = rows * (K / 8);
    int64_t blocks = (groups8 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (bf) hipLaunchKernelGGL((k_im2col<true>), dim3((unsigned)blocks), dim3(256), 0, st, x, x_sample_stride, panel, sh->B, sh->C, sh->H, sh->W, (int)OH, (int)OW, sh->KH, sh->KW, sh->stride_h, sh->stride_w, sh->pad_h, sh->pad_w, sh->dil_h, sh->dil_w, (int)K, groups8);
    else hipLaunchKernelGGL((k_im2col<false>), dim3((unsigned)blocks), dim3(256), 0, st, x, x_sample_stride, panel, sh->B, sh->C, sh->H, sh->W, (int)OH, (int)OW, sh->KH, sh->KW, sh->stride_h, sh->stride_w, sh->pad_h, sh->pad_w, sh->dil_h, sh->dil_w, (int)K, groups8);
}

// Inverse of the panel for the input gradient: gx[s][b][c][ih][iw] = sum over the kernel taps (kh, kw) that
// reach (ih, iw) of gpanel[(s, b, oh, ow)][(c, kh, kw)] -- a gather (no atomics, one thread per input
// element, fixed summation order).  SHARED: the input was shared by all samples, so sum over s as well.
__global__ __launch_bounds__(256) void k_col2im(const float *__restrict__ gp, float *__restrict__ gx, int nsamples, int shared,
                                                int B, int C, int H, int W, int OH, int OW, int KH, int KW,
                                                int sh_, int sw_, int ph, int pw, int dh, int dw, int64_t total)
{
    const int K = C * KH * KW, P = OH * OW;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int iw = (int)(id % W);
        int64_t t = id / W;
        const int ih = (int)(t % H); t /= H;
        const int c = (int)(t % C); t /= C;
        const int b = (int)(t % B);
        const int s0 = (int)(t / B);                       // 0 when shared
        float acc = 0.f;
        const int s_lo = shared ? 0 : s0, s_hi = shared ? nsamples : s0 + 1;
        for (int s = s_lo; s < s_hi; ++s) {
            const float *base = gp + ((int64_t)s * B + b) * P * K + (int64_t)c * KH * KW;
            for (int kh = 0; kh < KH; ++kh) {
                const int th = ih + ph - kh * dh;
                if (th < 0 || th % sh_ != 0) continue;
                const int oh = th / sh_;
                if (oh >= OH) continue;
                for (int kw = 0; kw < KW; ++kw) {
                    const int tw = iw + pw - kw * dw;
                    if (tw < 0 || tw % sw_ != 0) continue;
                    const int ow = tw / sw_;
                    if (ow >= OW) continue;
                    acc += base[(int64_t)(oh * OW + ow) * K + kh * KW + kw];
                }
            }
        }
        gx[id] = acc;
    }
}

// (images, O, P) NCHW -> rows [(image, pixel)][O], fp32 or bf16: the gy operand of the linear backward kernels.
template <bool BF>
__global__ __launch_bounds__(256) void k_nchw_to_rows(const float *__restrict__ y, void *__restrict__ out, int O, int P, int64_t total)
{
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int o = (int)(id % O);
        const int64_t m = id / O;
        const int64_t img = m / P;
        const int pix = (int)(m % P);
        const float v = y[(img * O + o) * P + pix];
        if constexpr (BF) reinterpret_cast<uint16_t *>(out)[id] = f2bf(v);
        else reinterpret_cast<float *>(out)[id] = v;
    }
}

static int conv_geometry(const bnn_conv2d_shape_t *sh, int64_t &OH, int64_t &OW, const char *who)
{
    if (!sh) { set_error("%s: NULL shape", who); return BNN_E_NULL; }
    if (sh->B < 1 || sh->C < 1 || sh->H < 1 || sh->W < 1 || sh->KH < 1 || sh->KW < 1 || sh->stride_h < 1 || sh->stride_w < 1 ||
        sh->pad_h < 0 || sh->pad_w < 0 || sh->dil_h < 1 || sh->dil_w < 1 || sh->groups != 1) { set_error("%s: bad shape (groups must be 1)", who); return BNN_E_SHAPE; }
    OH = ((int64_t)sh->H + 2 * sh->pad_h - (int64_t)sh->dil_h * (sh->KH - 1) - 1) / sh->stride_h + 1;
    OW = ((int64_t)sh->W + 2 * sh->pad_w - (int64_t)sh->dil_w * (sh->KW - 1) - 1) / sh->stride_w + 1;
    if (OH < 1 || OW < 1) { set_error("%s: empty output", who); return BNN_E_SHAPE; }
    return BNN_OK;
}

static int conv_common(const float *x, int64_t x_sample_stride, const float *w, int64_t w_sample_stride,
                       const float *b, int64_t b_sample_stride, const float *mu_w, const float *rho_w,
                       const float *mu_b, const float *rho_b, float *y, int64_t y_sample_stride,
                       const bnn_conv2d_shape_t *sh, int nsamples, const bnn_rng_t *rng_w,
                       const bnn_rng_t *rng_b, bool sampled, int compute, int flags, void *workspace,
                       int64_t workspace_bytes, void *stream, const char *who)
{
    if (!x || !y || !sh || (sampled ? (!mu_w || !rho_w) : !w)) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (sampled && ((mu_b == nullptr) != (rho_b == nullptr))) { set_error("%s: mu_b / rho_b must both be given or both NULL", who); return BNN_E_NULL; }
    if (sh->B < 0 || sh->C < 1 || sh->H < 1 || sh->W < 1 || sh->O < 1 || sh->KH < 1 || sh->KW < 1 ||
        sh->stride_h < 1 || sh->stride_w < 1 || sh->pad_h < 0 || sh->pad_w < 0 || sh->dil_h < 1 || sh->dil_w < 1 ||
        sh->groups < 1 || nsamples < 1) { set_error("%s: bad shape", who); return BNN_E_SHAPE; }
    // conv.py:15-18: channels must be divisible by groups
    if (sh->C % sh->groups != 0 || sh->O % sh->groups != 0) { set_error("%s: channels must be divisible by groups", who); return BNN_E_SHAPE; }
    const int64_t OH = ((int64_t)sh->H + 2 * sh->pad_h - (int64_t)sh->dil_h * (sh->KH - 1) - 1) / sh->stride_h + 1;
    const int64_t OW = ((int64_t)sh->W + 2 * sh->pad_w - (int64_t)sh->dil_w * (sh->KW - 1) - 1) / sh->stride_w + 1;
    if (OH < 1 || OW < 1) { set_error("%s: empty output (kernel larger than padded input)", who); return BNN_E_SHAPE; }
    const int64_t M = (int64_t)sh->B * OH * OW;
    const int64_t Cg = sh->C / sh->groups, Ng = sh->O / sh->groups;
    const int64_t K = Cg * sh->KH * sh->KW;
    if (M > 0x7FFFFFFF || K > 0x7FFFFFFF || (int64_t)sh->O * K > ((int64_t)1 << 34)) { set_error("%s: extent too large", who); return BNN_E_RANGE; }
    if (sampled) {
        int rc = check_rng(rng_w, nsamples);
        if (rc) { set_error("%s: bad rng_w", who); return rc; }
        if (mu_b) { rc = check_rng(rng_b, nsamples); if (rc) { set_error("%s: bad rng_b", who); return rc; } }
    }
    if (M == 0) return BNN_OK;
    // Fast path: explicit im2col panel in the caller's workspace + the draw-paced linear kernel with an
    // NCHW-storing epilogue (groups == 1, K % 8 == 0, 16-B aligned weights).  Otherwise the generic
    // implicit-GEMM kernel below (any shape, no workspace).
    {
        const int64_t need = bnn_conv2d_workspace_bytes(sh, x_sample_stride == 0 ? 1 : nsamples, compute);
        static const bool force_v1 = [] { const char *e = getenv("BNN_LINEAR_KERNEL"); return e && e[0] == 'v' && e[1] == '1'; }();
        const bool fast = !force_v1 && need > 0 && workspace && workspace_bytes >= need && al16(workspace) && al4(y) &&
                          (compute == BNN_COMPUTE_F32 || compute == BNN_COMPUTE_BF16) && (flags & ~BNN_FLAG_RELU) == 0 &&
                          (sampled ? (al16(mu_w) && al16(rho_w)) : (al16(w) && w_sample_stride % 4 == 0));
        if (fast) {
            const bool bf = compute == BNN_COMPUTE_BF16;
            const int nsx = x_sample_stride == 0 ? 1 : nsamples;
            hipStream_t st = (hipStream_t)stream;
            launch_im2col(x, x_sample_stride, workspace, sh, OH, OW, K, nsx, M, bf, st);
            int rc = check_launch(who);
            if (rc) return rc;
            GemmParams q{};
            q.A = reinterpret_cast<const float *>(workspace); q.a_sample_stride = x_sample_stride == 0 ? 0 : M * K; q.lda = K;
            q.Bw = w; q.b_sample_stride = w_sample_stride; q.mu = mu_w; q.rho = rho_w;
            q.bias = sampled ? nullptr : b; q.bias_sample_stride = b_sample_stride;
            q.mu_b = sampled ? mu_b : nullptr; q.rho_b = sampled ? rho_b : nullptr;
            q.Y = y; q.y_sample_stride = y_sample_stride; q.ldy = sh->O; q.O = sh->O;
            q.OH = (int32_t)OH; q.OW = (int32_t)OW;
            q.M = (int32_t)M; q.N = sh->O; q.K = (int32_t)K; q.S = nsamples; q.G = 1;
            q.flags = flags | kFlagStoreNCHW | (bf ? BNN_FLAG_X_BF16 : 0);
            q.vecA = 1; q.vecB = 1;
            if (sampled) { q.rng_w = make_rng(rng_w); q.rng_b = make_rng(mu_b ? rng_b : nullptr); }
            fill_workspace(q);
            return dispatch_linear_v2(q, sampled, compute, st, who);
        }
    }
    GemmParams p{};
    p.A = x; p.a_sample_stride = x_sample_stride; p.lda = 0;
    p.C = sh->C; p.H = sh->H; p.W = sh->W; p.OH = (int32_t)OH; p.OW = (int32_t)OW; p.KH = sh->KH; p.KW = sh->KW;
    p.sh = sh->stride_h; p.sw = sh->stride_w; p.ph = sh->pad_h; p.pw = sh->pad_w; p.dh = sh->dil_h; p.dw = sh->dil_w;
    p.Cg = (int32_t)Cg;
    p.Bw = w; p.b_sample_stride = w_sample_stride; p.mu = mu_w; p.rho = rho_w;
    p.bias = sampled ? nullptr : b; p.bias_sample_stride = b_sample_stride;
    p.mu_b = sampled ? mu_b : nullptr; p.rho_b = sampled ? rho_b : nullptr;
    p.Y = y; p.y_sample_stride = y_sample_stride; p.ldy = 0; p.O = sh->O;
    p.M = (int32_t)M; p.N = (int32_t)Ng; p.K = (int32_t)K; p.S = nsamples; p.G = sh->groups; p.flags = flags;
    p.vecA = 0;
    p.vecB = (K % 4 == 0) && (sampled ? (al16(mu_w) && al16(rho_w)) : (al16(w) && w_sample_stride % 4 == 0));
    if (sampled) { p.rng_w = make_rng(rng_w); p.rng_b = make_rng(mu_b ? rng_b : nullptr); }
    return dispatch<A_IM2COL>(p, sampled, compute, (hipStream_t)stream, who);
}

}  // namespace bnn

using namespace bnn;

extern "C" {

int bnn_check_device(int device, void *stream)
{
    if (device < 0 || device >= 64) { set_error("bnn_check_device: device out of range"); return BNN_E_RANGE; }
    if (!g_ws[device]) return BNN_OK;                       // no workspace registered: kernels had no word to set
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != device && hipSetDevice(device) != hipSuccess) { set_error("bnn_check_device: cannot select device %d", device); return BNN_E_RANGE; }
    unsigned word = 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(&word, g_ws[device], sizeof(word), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && word) e = hipMemsetAsync(g_ws[device], 0, sizeof(word), st);
    if (cur != device && cur >= 0) (void)hipSetDevice(cur);
    if (e != hipSuccess) { set_error("bnn_check_device: %s", hipGetErrorString(e)); return (int)e; }
    if (word) {
        set_error("device error word 0x%x:%s", word,
                  (word & kDevErrHandoffTimeout) ? " a hand-off wait of the fused linear kernel timed out (its tile was not stored)" : "");
        return BNN_E_DEVICE;
    }
    return BNN_OK;
}

int bnn_set_workspace(int device, void *ptr, int64_t bytes)
{
    if (device < 0 || device >= 64) { set_error("bnn_set_workspace: device out of range"); return BNN_E_RANGE; }
    if (ptr && (bytes < 2 * kTicketBytes || (reinterpret_cast<uintptr_t>(ptr) & 15u))) { set_error("bnn_set_workspace: need >= 128 KiB, 16-B aligned"); return BNN_E_SHAPE; }
    g_ws[device] = ptr;
    g_ws_bytes[device] = ptr ? bytes : 0;
    return BNN_OK;
}

int bnn_linear_backward_input_sampled(const void *gy, int64_t gy_sample_stride, int64_t ldgy,
                                      const float *mu_w, const float *rho_w, void *gx,
                                      int64_t gx_sample_stride, int64_t ldgx, int64_t M, int64_t N, int64_t K,
                                      int nsamples, const bnn_rng_t *rng_w, int compute, int flags, void *stream)
{
    const char *who = "bnn_linear_backward_input_sampled";
    if (M == 0 && N >= 1 && K >= 1 && nsamples >= 1) return BNN_OK;     // empty batch
    if (!gy || !mu_w || !rho_w || !gx) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (M < 0 || N < 1 || K < 1 || nsamples < 1 || ldgy < N || ldgx < K) { set_error("%s: bad extent", who); return BNN_E_SHAPE; }
    if (M > 0x7FFFFFFF || N > 0x7FFFFFFF || K > 0x7FFFFFFF || N * K > ((int64_t)1 << 34)) { set_error("%s: extent too large", who); return BNN_E_RANGE; }
    const bool gh = (flags & BNN_FLAG_X_BF16) != 0, xh = (flags & BNN_FLAG_Y_BF16) != 0;
    // the fused kernel streams gy with 16-B LDS-DMA pieces and draws 4 columns of W per Philox block
    const bool ok = (compute == BNN_COMPUTE_BF16 || (compute == BNN_COMPUTE_F32 && !gh && !xh)) && K % 4 == 0 &&
                    al16(mu_w) && al16(rho_w) && al16(gy) &&
                    (gh ? (N % 8 == 0 && ldgy % 8 == 0 && gy_sample_stride % 8 == 0) : (N % 4 == 0 && ldgy % 4 == 0 && gy_sample_stride % 4 == 0)) &&
                    (xh ? (reinterpret_cast<uintptr_t>(gx) & 1u) == 0 : al4(gx));
    if (!ok) { set_error("%s: needs K %% 4 == 0, N %% 4 (fp32 gy) / N %% 8 (bf16 gy) == 0, 16-B aligned operands; bf16 only in bf16 compute", who); return BNN_E_UNSUPPORTED; }
    if (flags & ~(BNN_FLAG_X_BF16 | BNN_FLAG_Y_BF16)) { set_error("%s: unknown flags", who); return BNN_E_UNSUPPORTED; }
    int rc = check_rng(rng_w, nsamples);
    if (rc) { set_error("%s: bad rng_w", who); return rc; }
    if (M == 0) return BNN_OK;
    GemmParams p{};
    p.A = reinterpret_cast<const float *>(gy); p.a_sample_stride = gy_sample_stride; p.lda = ldgy;
    p.mu = mu_w; p.rho = rho_w;
    p.Y = reinterpret_cast<float *>(gx); p.y_sample_stride = gx_sample_stride; p.ldy = ldgx;
    p.M = (int32_t)M; p.N = (int32_t)K; p.K = (int32_t)N;           // outputs = columns of W, reduction = rows
    p.O = p.N; p.S = nsamples; p.G = 1; p.flags = flags; p.vecA = 1; p.vecB = 1;
    p.rng_w = make_rng(rng_w); p.rng_b = make_rng(nullptr);
    return dispatch_linear_dgrad(p, compute, (hipStream_t)stream, who);
}

int bnn_linear_forward_sampled(const void *x, int64_t x_sample_stride, int64_t ldx,
                               const float *mu_w, const float *rho_w, const float *mu_b,
                               const float *rho_b, void *y, int64_t y_sample_stride, int64_t ldy,
                               int64_t M, int64_t N, int64_t K, int nsamples, const bnn_rng_t *rng_w,
                               const bnn_rng_t *rng_b, int compute, int flags, void *stream)
{
    return linear_common((const float *)x, x_sample_stride, ldx, nullptr, 0, nullptr, 0, mu_w, rho_w, mu_b, rho_b, (float *)y,
                         y_sample_stride, ldy, M, N, K, nsamples, rng_w, rng_b, true, compute, flags,
                         stream, "bnn_linear_forward_sampled");
}

int bnn_linear_forward_sampled_kl(const void *x, int64_t x_sample_stride, int64_t ldx,
                                  const float *mu_w, const float *rho_w, const float *mu_b,
                                  const float *rho_b, void *y, int64_t y_sample_stride, int64_t ldy,
                                  int64_t M, int64_t N, int64_t K, int nsamples, const bnn_rng_t *rng_w,
                                  const bnn_rng_t *rng_b, int compute, int flags,
                                  const bnn_kl_tensor_t *tensors, int ntensors, void *kl_workspace, void *stream)
{
    const char *who = "bnn_linear_forward_sampled_kl";
    KlPiggy kl{};
    const bool planned = kl_plan_piggy(tensors, ntensors, kl_workspace, kl);
    int rc = linear_common((const float *)x, x_sample_stride, ldx, nullptr, 0, nullptr, 0, mu_w, rho_w, mu_b, rho_b, (float *)y,
                           y_sample_stride, ldy, M, N, K, nsamples, rng_w, rng_b, true, compute, flags, stream, who,
                           planned ? &kl : nullptr);
    if (rc) return rc;
    // not carried (a wide layer, an unaligned one, an empty batch, a big model): the first pass gets its own launch
    if (!planned || !kl.taken) rc = bnn_kl_forward_partial(tensors, ntensors, kl_workspace, stream);
    return rc;
}

int bnn_linear_forward(const float *x, int64_t x_sample_stride, int64_t ldx, const float *w,
                       int64_t w_sample_stride, const float *b, int64_t b_sample_stride, float *y,
                       int64_t y_sample_stride, int64_t ldy, int64_t M, int64_t N, int64_t K,
                       int nsamples, int compute, int flags, void *stream)
{
    return linear_common(x, x_sample_stride, ldx, w, w_sample_stride, b, b_sample_stride, nullptr, nullptr,
                         nullptr, nullptr, y, y_sample_stride, ldy, M, N, K, nsamples, nullptr, nullptr,
                         false, compute, flags, stream, "bnn_linear_forward");
}

int bnn_conv2d_im2col(const float *x, int64_t x_sample_stride, const bnn_conv2d_shape_t *sh, int x_samples,
                      void *panel, int out_bf16, void *stream)
{
    const char *who = "bnn_conv2d_im2col";
    int64_t OH, OW;
    int rc = conv_geometry(sh, OH, OW, who);
    if (rc) return rc;
    if (!x || !panel) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    const int64_t K = (int64_t)sh->C * sh->KH * sh->KW, M = (int64_t)sh->B * OH * OW;
    if (K % 8 != 0 || x_samples < 1 || M > 0x7FFFFFFF || !al16(panel)) { set_error("%s: needs C*KH*KW %% 8 == 0, a 16-B aligned panel", who); return BNN_E_UNSUPPORTED; }
    launch_im2col(x, x_sample_stride, panel, sh, OH, OW, K, x_samples, M, out_bf16 != 0, (hipStream_t)stream);
    return check_launch(who);
}

int bnn_conv2d_col2im(const float *gpanel, const bnn_conv2d_shape_t *sh, int nsamples, int shared_x, float *gx, void *stream)
{
    const char *who = "bnn_conv2d_col2im";
    int64_t OH, OW;
    int rc = conv_geometry(sh, OH, OW, who);
    if (rc) return rc;
    if (!gpanel || !gx || nsamples < 1) { set_error("%s: NULL pointer / nsamples < 1", who); return BNN_E_NULL; }
    const int64_t total = (int64_t)(shared_x ? 1 : nsamples) * sh->B * sh->C * sh->H * sh->W;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_col2im, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gpanel, gx, nsamples, shared_x ? 1 : 0,
                       sh->B, sh->C, sh->H, sh->W, (int)OH, (int)OW, sh->KH, sh->KW, sh->stride_h, sh->stride_w, sh->pad_h, sh->pad_w,
                       sh->dil_h, sh->dil_w, total);
    return check_launch(who);
}

int bnn_nchw_to_rows(const float *y, int64_t images, int channels, int pixels, void *rows, int out_bf16, void *stream)
{
    const char *who = "bnn_nchw_to_rows";
    if (!y || !rows) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (images < 1 || channels < 1 || pixels < 1) { set_error("%s: bad extent", who); return BNN_E_SHAPE; }
    const int64_t total = images * channels * pixels;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (out_bf16) hipLaunchKernelGGL((k_nchw_to_rows<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, rows, channels, pixels, total);
    else hipLaunchKernelGGL((k_nchw_to_rows<false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, rows, channels, pixels, total);
    return check_launch(who);
}

int64_t bnn_conv2d_workspace_bytes(const bnn_conv2d_shape_t *sh, int x_samples, int compute)
{
    if (!sh || x_samples < 1 || sh->groups != 1 || sh->B < 1 || sh->C < 1 || sh->KH < 1 || sh->KW < 1 ||
        sh->stride_h < 1 || sh->stride_w < 1 || sh->dil_h < 1 || sh->dil_w < 1) return 0;
    const int64_t OH = ((int64_t)sh->H + 2 * sh->pad_h - (int64_t)sh->dil_h * (sh->KH - 1) - 1) / sh->stride_h + 1;
    const int64_t OW = ((int64_t)sh->W + 2 * sh->pad_w - (int64_t)sh->dil_w * (sh->KW - 1) - 1) / sh->stride_w + 1;
    const int64_t K = (int64_t)sh->C * sh->KH * sh->KW;
    if (OH < 1 || OW < 1 || K % 8 != 0 || K < 32 || sh->O < 16) return 0;     // tiny layers: the generic kernel
    const int64_t M = (int64_t)sh->B * OH * OW;
    if (M > 0x7FFFFFFF) return 0;
    return (int64_t)x_samples * M * K * (compute == BNN_COMPUTE_BF16 ? 2 : 4);
}

int bnn_conv2d_forward_sampled(const float *x, int64_t x_sample_stride, const float *mu_w,
                               const float *rho_w, const float *mu_b, const float *rho_b, float *y,
                               int64_t y_sample_stride, const bnn_conv2d_shape_t *shape, int nsamples,
                               const bnn_rng_t *rng_w, const bnn_rng_t *rng_b, int compute, int flags,
                               void *workspace, int64_t workspace_bytes, void *stream)
{
    return conv_common(x, x_sample_stride, nullptr, 0, nullptr, 0, mu_w, rho_w, mu_b, rho_b, y,
                       y_sample_stride, shape, nsamples, rng_w, rng_b, true, compute, flags, workspace,
                       workspace_bytes, stream, "bnn_conv2d_forward_sampled");
}

int bnn_conv2d_forward(const float *x, int64_t x_sample_stride, const float *w, int64_t w_sample_stride,
                       const float *b, int64_t b_sample_stride, float *y, int64_t y_sample_stride,
                       const bnn_conv2d_shape_t *shape, int nsamples, int compute, int flags,
                       void *workspace, int64_t workspace_bytes, void *stream)
{
    return conv_common(x, x_sample_stride, w, w_sample_stride, b, b_sample_stride, nullptr, nullptr,
                       nullptr, nullptr, y, y_sample_stride, shape, nsamples, nullptr, nullptr, false,
                       compute, flags, workspace, workspace_bytes, stream, "bnn_conv2d_forward");
}

}  // extern "C"
