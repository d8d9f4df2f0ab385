// bnn_elementwise.hip -- K1 (posterior draw w = mu + sigma(rho) * eps), its backward,
// the raw eps stream, and the MC-sample reduction.  HBM-bound streaming kernels:
// 16-byte loads/stores per lane, grid-stride over <= 2048 workgroups of 256 threads.
//
// Algorithmic bytes per posterior scalar (fp32): 8 read (mu, rho) + 4 (fp32) or
// 2 (bf16) written per draw; eps-supplied mode reads 4 more.
#include <cstdarg>
#include <cstdio>
#include <atomic>

#include "bnn_device.hpp"
#include "bnn_mc.hpp"

namespace bnn {

static thread_local char g_err[256] = "";
static std::atomic<uint64_t> g_launches{0};

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
int check_launch(const char *what)
{
    count_launch();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return BNN_OK;
}

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;

static inline int grid_for(int64_t work_items)
{
    int64_t b = (work_items + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}

template <int DT>
__device__ __forceinline__ void store4(void *out, int64_t vec_idx, float4 v)
{
    if constexpr (DT == BNN_F32) {
        reinterpret_cast<float4 *>(out)[vec_idx] = v;
    } else {
        uint2 p;
        p.x = pack_bf16x2(v.x, v.y);
        p.y = pack_bf16x2(v.z, v.w);
        reinterpret_cast<uint2 *>(out)[vec_idx] = p;
    }
}
template <int DT>
__device__ __forceinline__ void store1(void *out, int64_t idx, float v)
{
    if constexpr (DT == BNN_F32) reinterpret_cast<float *>(out)[idx] = v;
    else reinterpret_cast<uint16_t *>(out)[idx] = f2bf(v);
}

// ---------------------------------------------------------------- K1, eps given
// (eps supplied = the parity mode: sigma to ~1e-6 RELATIVE accuracy, sigma_accurate -- the drawn weight then agrees with the
// reference's fp32 expression to its last bits; the Philox kernels keep sigma_draw, which every re-created draw shares)
template <int DT, bool VEC>
__global__ __launch_bounds__(kThreads) void k_sample_affine_eps(
    const float *__restrict__ mu, const float *__restrict__ rho, const float *__restrict__ eps,
    void *__restrict__ out, int64_t n)
{
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    if constexpr (VEC) {
        const int64_t nvec = n >> 2;
        for (int64_t v = tid; v < nvec; v += nthreads) {
            const float4 m = reinterpret_cast<const float4 *>(mu)[v];
            const float4 r = reinterpret_cast<const float4 *>(rho)[v];
            const float4 e = reinterpret_cast<const float4 *>(eps)[v];
            float4 w;
            w.x = fmaf(sigma_accurate(r.x), e.x, m.x);
            w.y = fmaf(sigma_accurate(r.y), e.y, m.y);
            w.z = fmaf(sigma_accurate(r.z), e.z, m.z);
            w.w = fmaf(sigma_accurate(r.w), e.w, m.w);
            store4<DT>(out, v, w);
        }
        for (int64_t i = (nvec << 2) + tid; i < n; i += nthreads)
            store1<DT>(out, i, fmaf(sigma_accurate(rho[i]), eps[i], mu[i]));
    } else {
        for (int64_t i = tid; i < n; i += nthreads)
            store1<DT>(out, i, fmaf(sigma_accurate(rho[i]), eps[i], mu[i]));
    }
}

// ---------------------------------------------------------------- K1, Philox eps
// MODE 0: out = mu + sigma * eps ; MODE 1: out = eps (mu / rho unused).
// (A branch-free 16-B loop with the next block's (mu, rho) prefetched, ragged tail separate, was measured on the 768-MiB
// stream: 4.98 TB/s against 5.42 for this plain grid-stride loop -- not kept.  Round keys in VGPRs: no difference here.
// Two blocks per thread and iteration with the four loads requested together: 5.35 against 5.61 on the same box.)
template <int DT, int MODE, bool VEC>
__global__ __launch_bounds__(kThreads) void k_sample_affine_philox(
    const float *__restrict__ mu, const float *__restrict__ rho, void *__restrict__ out,
    int64_t n, int nsamples, int64_t out_sample_stride, RngDev rng)
{
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    const uint32_t edev = rng_epoch_dev(rng);
    const int64_t nblk = (n + 3) >> 2;
    const int esz = (DT == BNN_F32) ? 4 : 2;
    for (int64_t v = tid; v < nblk; v += nthreads) {
        const int64_t base = v << 2;
        const bool full = base + 4 <= n;
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f), sg = make_float4(1.f, 1.f, 1.f, 1.f);
        if constexpr (MODE == 0) {
            float4 r;
            if (VEC && full) {
                m = reinterpret_cast<const float4 *>(mu)[v];
                r = reinterpret_cast<const float4 *>(rho)[v];
            } else {
                m.x = mu[base]; r.x = rho[base];
                m.y = base + 1 < n ? mu[base + 1] : 0.f; r.y = base + 1 < n ? rho[base + 1] : 0.f;
                m.z = base + 2 < n ? mu[base + 2] : 0.f; r.z = base + 2 < n ? rho[base + 2] : 0.f;
                m.w = base + 3 < n ? mu[base + 3] : 0.f; r.w = base + 3 < n ? rho[base + 3] : 0.f;
            }
            sg.x = sigma_draw(r.x); sg.y = sigma_draw(r.y);
            sg.z = sigma_draw(r.z); sg.w = sigma_draw(r.w);
        }
        for (int s = 0; s < nsamples; ++s) {
            const float4 z = eps4(rng, edev, (uint32_t)v, rng.sample0 + (uint32_t)s);
            float4 w;
            w.x = fmaf(sg.x, z.x, m.x); w.y = fmaf(sg.y, z.y, m.y);
            w.z = fmaf(sg.z, z.z, m.z); w.w = fmaf(sg.w, z.w, m.w);
            char *o = reinterpret_cast<char *>(out) + (int64_t)s * out_sample_stride * esz;
            if (VEC && full) {
                store4<DT>(o, v, w);
            } else {
                store1<DT>(o, base, w.x);
                if (base + 1 < n) store1<DT>(o, base + 1, w.y);
                if (base + 2 < n) store1<DT>(o, base + 2, w.z);
                if (base + 3 < n) store1<DT>(o, base + 3, w.w);
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_sigma(const float *__restrict__ rho,
                                                    float *__restrict__ out, int64_t n)
{
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    for (int64_t i = tid; i < n; i += nthreads) out[i] = sigma_accurate(rho[i]);
}

// ---------------------------------------------------------------- K1 backward
// g_mu = sum_s g_w[s]; g_rho = (sum_s g_w[s] * eps_s) * sigmoid(rho).
template <bool EXT_EPS>
__global__ __launch_bounds__(kThreads) void k_sample_affine_bwd(
    const float *__restrict__ g_w, int64_t g_w_sample_stride, const float *__restrict__ rho,
    const float *__restrict__ eps, int64_t eps_sample_stride, RngDev rng, int64_t n,
    int nsamples, float *__restrict__ g_mu, float *__restrict__ g_rho, int accumulate)
{
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    const uint32_t edev = EXT_EPS ? 0u : rng_epoch_dev(rng);
    const int64_t nblk = (n + 3) >> 2;
    for (int64_t v = tid; v < nblk; v += nthreads) {
        const int64_t base = v << 2;
        float am[4] = {0.f, 0.f, 0.f, 0.f}, ar[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < nsamples; ++s) {
            float z[4];
            if constexpr (!EXT_EPS) {
                const float4 zz = eps4(rng, edev, (uint32_t)v, rng.sample0 + (uint32_t)s);
                z[0] = zz.x; z[1] = zz.y; z[2] = zz.z; z[3] = zz.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (base + j < n) {
                    const float g = g_w[(int64_t)s * g_w_sample_stride + base + j];
                    const float e = EXT_EPS ? eps[(int64_t)s * eps_sample_stride + base + j] : z[j];
                    am[j] += g;
                    ar[j] = fmaf(g, e, ar[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (base + j < n) {
                const float gr = ar[j] * dsoftplus(rho[base + j]);
                if (accumulate) {
                    g_mu[base + j] += am[j];
                    g_rho[base + j] += gr;
                } else {
                    g_mu[base + j] = am[j];
                    g_rho[base + j] = gr;
                }
            }
        }
    }
}

// PruneNormal's score (prune/prune.py:11): log N(0; mu, sigma(rho)) = -mu^2 / (2 sigma^2) - ln sigma - ln sqrt(2 pi)
__global__ __launch_bounds__(kThreads) void k_prune_score(const float *__restrict__ mu, const float *__restrict__ rho,
                                                          float *__restrict__ out, int64_t n)
{
    const int64_t tid = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * kThreads;
    for (int64_t i = tid; i < n; i += nthreads) {
        const float sg = sigma_accurate(rho[i]);
        const float t = mu[i] / sg;
        out[i] = -0.5f * t * t - __builtin_amdgcn_logf(sg) * 0.693147180559945309f - 0.918938533204672742f;
    }
}

__global__ void k_rng_advance(uint32_t *epoch_dev, uint32_t inc) { epoch_dev[0] += inc; }

// ---------------------------------------------------------------- MC reduction
__global__ __launch_bounds__(kMcThreads) void k_mc_sum(const float *__restrict__ y,
                                                       int64_t y_sample_stride, int nsamples,
                                                       int64_t n, float scale,
                                                       float *__restrict__ out, int accumulate,
                                                       uint32_t *advance_epoch, uint32_t advance_inc)
{
    if (advance_epoch && blockIdx.x == 0 && threadIdx.x == 0) advance_epoch[0] += advance_inc;   // nothing in this kernel draws
    if (nsamples > kMcSplitAbove) mc_sum_split_body(y, y_sample_stride, nsamples, n, scale, out, accumulate, (int)blockIdx.x);
    else mc_sum_body(y, y_sample_stride, nsamples, n, scale, out, accumulate, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------- diagnostics
// VALU cost of the draw: every thread runs `iters` Philox blocks (4 draws each) of the chosen
// stage and keeps one live value; nothing but one store per thread touches memory.
// stage 0: Philox only; 1: + Box-Muller; 2: + sigma_fast per element; 3: + fma (full draw).
template <int STAGE>
__global__ __launch_bounds__(kThreads) void k_diag_sampler(float *__restrict__ out, int iters, RngDev rng)
{
    const uint32_t tid = blockIdx.x * kThreads + threadIdx.x;
    float acc = 0.f;
    float rho = -2.0f + 1e-6f * (float)(tid & 1023);
    for (int i = 0; i < iters; ++i) {
        const uint32_t blk = tid * (uint32_t)iters + (uint32_t)i;
        if constexpr (STAGE == 0) {
            const uint4 x = philox4x32_10(make_uint4(blk, rng.stream_hi, rng.epoch_host, 0u), rng.key0, rng.key1);
            acc += __uint_as_float((x.x ^ x.y ^ x.z ^ x.w) & 0x3F800000u);
        } else {
            const float4 z = eps4(rng, 0u, blk, 0u);
            if constexpr (STAGE == 1) {
                acc += z.x + z.y + z.z + z.w;
            } else {
                const float s0 = sigma_draw(rho), s1 = sigma_draw(rho + 0.25f), s2 = sigma_draw(rho + 0.5f),
                            s3 = sigma_draw(rho + 0.75f);
                rho += 1e-7f;
                if constexpr (STAGE == 2) acc += z.x + z.y + z.z + z.w + s0 + s1 + s2 + s3;
                else acc += fmaf(s0, z.x, 0.1f) + fmaf(s1, z.y, 0.2f) + fmaf(s2, z.z, 0.3f) + fmaf(s3, z.w, 0.4f);
            }
        }
    }
    out[tid] = acc;
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned_to(const void *p, unsigned a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

}  // namespace bnn

using namespace bnn;

extern "C" {

int bnn_abi_version(void) { return BNN_ABI_VERSION; }
const char *bnn_arch(void) { return "gfx950"; }
const char *bnn_last_error(void) { return g_err; }
uint64_t bnn_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

int bnn_sample_affine_eps(const float *mu, const float *rho, const float *eps, void *out,
                          int64_t n, int out_dtype, void *stream)
{
    if (!mu || !rho || !eps || !out) { set_error("bnn_sample_affine_eps: NULL pointer"); return BNN_E_NULL; }
    if (n < 0) { set_error("bnn_sample_affine_eps: n < 0"); return BNN_E_SHAPE; }
    if (out_dtype != BNN_F32 && out_dtype != BNN_BF16) { set_error("bnn_sample_affine_eps: dtype"); return BNN_E_DTYPE; }
    if (!aligned_to(mu, 4) || !aligned_to(rho, 4) || !aligned_to(eps, 4) ||
        !aligned_to(out, out_dtype == BNN_F32 ? 4 : 2)) { set_error("bnn_sample_affine_eps: misaligned"); return BNN_E_ALIGN; }
    if (n == 0) return BNN_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = aligned16(mu) && aligned16(rho) && aligned16(eps) &&
                     (out_dtype == BNN_F32 ? aligned16(out) : aligned_to(out, 8));
    const int grid = grid_for(vec ? (n + 3) / 4 : n);
#define LAUNCH(DT, V) hipLaunchKernelGGL((k_sample_affine_eps<DT, V>), dim3(grid), dim3(kThreads), 0, st, mu, rho, eps, out, n)
    if (out_dtype == BNN_F32) { if (vec) LAUNCH(BNN_F32, true); else LAUNCH(BNN_F32, false); }
    else { if (vec) LAUNCH(BNN_BF16, true); else LAUNCH(BNN_BF16, false); }
#undef LAUNCH
    return check_launch("bnn_sample_affine_eps");
}

static int launch_philox(const float *mu, const float *rho, void *out, int64_t n, int nsamples,
                         int64_t out_sample_stride, int out_dtype, int mode, const bnn_rng_t *rng,
                         void *stream, const char *who)
{
    if (!out || (mode == 0 && (!mu || !rho))) { set_error("%s: NULL pointer", who); return BNN_E_NULL; }
    if (n < 0 || nsamples < 1 || (nsamples > 1 && out_sample_stride < n)) { set_error("%s: bad extent", who); return BNN_E_SHAPE; }
    if (n > (int64_t)1 << 34) { set_error("%s: n > 2^34", who); return BNN_E_RANGE; }
    if (out_dtype != BNN_F32 && out_dtype != BNN_BF16) { set_error("%s: dtype", who); return BNN_E_DTYPE; }
    int rc = check_rng(rng, nsamples);
    if (rc) { set_error("%s: bad rng (stream < 65536, sample0 + nsamples <= 65536)", who); return rc; }
    if (n == 0) return BNN_OK;
    hipStream_t st = (hipStream_t)stream;
    const int esz = out_dtype == BNN_F32 ? 4 : 2;
    const bool vec = (mode == 1 || (aligned16(mu) && aligned16(rho))) && aligned_to(out, 4 * esz) &&
                     (nsamples == 1 || (out_sample_stride % 4) == 0);
    const int grid = grid_for((n + 3) / 4);
    const RngDev rd = make_rng(rng);
    // PF = true: measured equal to the plain loop on a 768-MiB stream (5.28 TB/s both); kept for small grids
#define LAUNCH(DT, M, V) hipLaunchKernelGGL((k_sample_affine_philox<DT, M, V>), dim3(grid), dim3(kThreads), 0, st, mu, rho, out, n, nsamples, out_sample_stride, rd)
    if (mode == 0) {
        if (out_dtype == BNN_F32) { if (vec) LAUNCH(BNN_F32, 0, true); else LAUNCH(BNN_F32, 0, false); }
        else { if (vec) LAUNCH(BNN_BF16, 0, true); else LAUNCH(BNN_BF16, 0, false); }
    } else {
        if (vec) LAUNCH(BNN_F32, 1, true); else LAUNCH(BNN_F32, 1, false);
    }
#undef LAUNCH
    return check_launch(who);
}

int bnn_sample_affine_philox(const float *mu, const float *rho, void *out, int64_t n, int nsamples,
                             int64_t out_sample_stride, int out_dtype, const bnn_rng_t *rng,
                             void *stream)
{
    return launch_philox(mu, rho, out, n, nsamples, out_sample_stride, out_dtype, 0, rng, stream,
                         "bnn_sample_affine_philox");
}

int bnn_eps_philox(float *out, int64_t n, int nsamples, int64_t out_sample_stride,
                   const bnn_rng_t *rng, void *stream)
{
    return launch_philox(nullptr, nullptr, out, n, nsamples, out_sample_stride, BNN_F32, 1, rng,
                         stream, "bnn_eps_philox");
}

int bnn_sigma(const float *rho, float *out, int64_t n, void *stream)
{
    if (!rho || !out) { set_error("bnn_sigma: NULL pointer"); return BNN_E_NULL; }
    if (n < 0) { set_error("bnn_sigma: n < 0"); return BNN_E_SHAPE; }
    if (n == 0) return BNN_OK;
    hipLaunchKernelGGL(k_sigma, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, rho, out, n);
    return check_launch("bnn_sigma");
}

int bnn_sample_affine_bwd(const float *g_w, int64_t g_w_sample_stride, const float *rho,
                          const float *eps, int64_t eps_sample_stride, const bnn_rng_t *rng,
                          int64_t n, int nsamples, float *g_mu, float *g_rho, int accumulate,
                          void *stream)
{
    if (!g_w || !rho || !g_mu || !g_rho) { set_error("bnn_sample_affine_bwd: NULL pointer"); return BNN_E_NULL; }
    if ((eps == nullptr) == (rng == nullptr)) { set_error("bnn_sample_affine_bwd: give exactly one of eps / rng"); return BNN_E_NULL; }
    if (n < 0 || nsamples < 1) { set_error("bnn_sample_affine_bwd: bad extent"); return BNN_E_SHAPE; }
    if (n > (int64_t)1 << 34) { set_error("bnn_sample_affine_bwd: n > 2^34"); return BNN_E_RANGE; }
    if (rng) { int rc = check_rng(rng, nsamples); if (rc) { set_error("bnn_sample_affine_bwd: bad rng"); return rc; } }
    if (n == 0) return BNN_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for((n + 3) / 4);
    const RngDev rd = make_rng(rng);
    if (eps)
        hipLaunchKernelGGL((k_sample_affine_bwd<true>), dim3(grid), dim3(kThreads), 0, st, g_w, g_w_sample_stride, rho, eps, eps_sample_stride, rd, n, nsamples, g_mu, g_rho, accumulate);
    else
        hipLaunchKernelGGL((k_sample_affine_bwd<false>), dim3(grid), dim3(kThreads), 0, st, g_w, g_w_sample_stride, rho, eps, eps_sample_stride, rd, n, nsamples, g_mu, g_rho, accumulate);
    return check_launch("bnn_sample_affine_bwd");
}

int bnn_prune_score(const float *mu, const float *rho, float *out, int64_t n, void *stream)
{
    if (!mu || !rho || !out) { set_error("bnn_prune_score: NULL pointer"); return BNN_E_NULL; }
    if (n < 0) { set_error("bnn_prune_score: n < 0"); return BNN_E_SHAPE; }
    if (n == 0) return BNN_OK;
    hipLaunchKernelGGL(k_prune_score, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, mu, rho, out, n);
    return check_launch("bnn_prune_score");
}

int bnn_rng_advance(uint32_t *epoch_dev, uint32_t inc, void *stream)
{
    if (!epoch_dev) { set_error("bnn_rng_advance: NULL pointer"); return BNN_E_NULL; }
    hipLaunchKernelGGL(k_rng_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, epoch_dev, inc);
    return check_launch("bnn_rng_advance");
}

int bnn_diag_sampler(float *out, int blocks, int iters, int stage, void *stream)
{
    if (!out) { set_error("bnn_diag_sampler: NULL pointer"); return BNN_E_NULL; }
    if (blocks < 1 || iters < 1 || stage < 0 || stage > 3) { set_error("bnn_diag_sampler: bad argument"); return BNN_E_SHAPE; }
    bnn_rng_t r{}; r.seed = 0x9E3779B97F4A7C15ull; r.stream = 1;
    const RngDev rd = make_rng(&r);
    hipStream_t st = (hipStream_t)stream;
    switch (stage) {
    case 0: hipLaunchKernelGGL(k_diag_sampler<0>, dim3(blocks), dim3(kThreads), 0, st, out, iters, rd); break;
    case 1: hipLaunchKernelGGL(k_diag_sampler<1>, dim3(blocks), dim3(kThreads), 0, st, out, iters, rd); break;
    case 2: hipLaunchKernelGGL(k_diag_sampler<2>, dim3(blocks), dim3(kThreads), 0, st, out, iters, rd); break;
    default: hipLaunchKernelGGL(k_diag_sampler<3>, dim3(blocks), dim3(kThreads), 0, st, out, iters, rd); break;
    }
    return check_launch("bnn_diag_sampler");
}

int bnn_mc_sum(const float *y, int64_t y_sample_stride, int nsamples, int64_t n, float scale,
               float *out, int accumulate, uint32_t *advance_epoch, uint32_t advance_inc, void *stream)
{
    if (!y || !out) { set_error("bnn_mc_sum: NULL pointer"); return BNN_E_NULL; }
    if (n < 0 || nsamples < 1) { set_error("bnn_mc_sum: bad extent"); return BNN_E_SHAPE; }
    if (n == 0) return advance_epoch ? bnn_rng_advance(advance_epoch, advance_inc, stream) : BNN_OK;
    unsigned grid = grid_for(n);
    if (nsamples > kMcSplitAbove) {
        if (nsamples > 4 * kMcSplitMax || (n + 63) / 64 > 0x7FFFFFF0) { set_error("bnn_mc_sum: more than %d addends per output (or too many outputs)", 4 * kMcSplitMax); return BNN_E_RANGE; }
        grid = (unsigned)((n + 63) / 64);
    }
    hipLaunchKernelGGL(k_mc_sum, dim3(grid), dim3(kMcThreads), 0, (hipStream_t)stream, y, y_sample_stride,
                       nsamples, n, scale, out, accumulate, advance_epoch, advance_inc);
    return check_launch("bnn_mc_sum");
}

}  // extern "C"
