// bnn_kl.hip -- K3: closed-form KL( N(mu, sigma(rho)^2) || N(mu_p, sigma_p^2) ) summed per
// tensor, all tensors of a model in ONE launch, plus its backward.
//
// HBM-bound reduction: 8 algorithmic bytes per posterior scalar (mu, rho), 16-byte loads,
// wave64 shuffle tree -> LDS -> one double partial per workgroup; a second, single-
// workgroup kernel adds the partials in a fixed order (bitwise reproducible, no float
// atomics) and forms the reference's mean-of-means / n_batches scalar.
#include "bnn_device.hpp"
#include "bnn_kl_body.hpp"
#include "bnn_mc.hpp"

namespace bnn {

template <int PT, int REP = 1>
__global__ __launch_bounds__(kKlThreads) void k_kl_partial(KlLaunch L, double *__restrict__ partials)
{
    const int t = find_tensor(L, blockIdx.x);
    kl_partial_block<PT, REP>(L.t[t], (int)blockIdx.x, L.partial_base + (int)blockIdx.x, partials);
}

// One workgroup: wave w adds the partials of tensors w, w + 4, ... in a fixed order (lane-strided,
// then the shuffle tree: no barrier per tensor); then the scalar of KLDivergence.forward
// (loss.py:38): mean over tensors of (sum_t / n_t), / n_batches, added in tensor order.
__device__ __forceinline__ void kl_final_body(const KlFinal &F, const double *__restrict__ partials, float *__restrict__ out)
{
    __shared__ double means[kKlMaxTensors];
    const int lane = threadIdx.x & 63;
    for (int t = threadIdx.x >> 6; t < F.ntensors; t += kKlThreads / 64) {
        // four independent lane-strided chains: the loop is a chain of dependent-latency loads otherwise (703 partials
        // of the MLP's largest tensor = 11 round trips; now 3); fixed order all the same
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int i = F.first[t] + lane;
        const int end = F.first[t + 1];
        for (; i + 192 < end; i += 256) {
            a0 += partials[i]; a1 += partials[i + 64]; a2 += partials[i + 128]; a3 += partials[i + 192];
        }
        for (; i < end; i += 64) a0 += partials[i];
        double a = wave_sum((a0 + a1) + (a2 + a3));
        if (lane == 0) {
            out[t] = (float)a;
            means[t] = (double)(float)(a / (double)F.n[t]);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int t = 0; t < F.ntensors; ++t) total += means[t];
        out[F.ntensors] = (float)((total / (double)F.ntensors) / (double)F.n_batches);
    }
}

__global__ __launch_bounds__(kKlThreads) void k_kl_final(KlFinal F, const double *__restrict__ partials,
                                                         float *__restrict__ out)
{
    kl_final_body(F, partials, out);
}

// Tail of an MC step in ONE launch: workgroups [0, gridDim.x - 1) run the MC reduction (and bump the device epoch), the
// last one runs KL's second pass over the partials an earlier bnn_kl_forward_partial left in the workspace.  A launch
// costs >= 4 us on MI355X whatever it does (k_rng_advance, 1 thread: 4.1 us); the BASELINE step had three such tails.
static_assert(kMcThreads == kKlThreads, "one block shape for both bodies");
__global__ __launch_bounds__(kKlThreads) void k_mc_sum_kl(const float *__restrict__ y, int64_t y_sample_stride, int nsamples,
                                                          int64_t n, float scale, float *__restrict__ out, int accumulate,
                                                          uint32_t *advance_epoch, uint32_t advance_inc,
                                                          KlFinal F, const double *__restrict__ partials, float *__restrict__ kl_out)
{
    const int nmc = (int)gridDim.x - 1;
    if ((int)blockIdx.x == nmc) {
        kl_final_body(F, partials, kl_out);
        return;
    }
    if (advance_epoch && blockIdx.x == 0 && threadIdx.x == 0) advance_epoch[0] += advance_inc;
    if (nsamples > kMcSplitAbove) mc_sum_split_body(y, y_sample_stride, nsamples, n, scale, out, accumulate, (int)blockIdx.x);   // (uniform)
    else mc_sum_body(y, y_sample_stride, nsamples, n, scale, out, accumulate, (int)blockIdx.x, nmc);
}

__global__ __launch_bounds__(kKlThreads) void k_kl_backward(KlLaunch L, const float *__restrict__ upstream,
                                                            int accumulate)
{
    const int t = find_tensor(L, blockIdx.x);
    const KlTensorDev T = L.t[t];
    const int64_t base = (int64_t)(blockIdx.x - T.first_block) * kKlChunk;
    const float up = upstream ? upstream[0] : 1.0f;
    const float sc = T.scale * up;
    const float inv_ps2 = 1.0f / (T.prior_sigma * T.prior_sigma);
#pragma unroll
    for (int it = 0; it < kKlPerThread; ++it) {
        const int64_t e = base + (int64_t)it * kKlThreads + threadIdx.x;
        if (e < T.n) {
            const float mu = T.mu[e], rho = T.rho[e];
            const float sg = sigma_accurate(rho);
            const float gm = sc * (mu - T.prior_mu) * inv_ps2;
            const float gr = sc * (sg * inv_ps2 - 1.0f / sg) * dsoftplus(rho);
            if (accumulate) {
                T.g_mu[e] += gm;
                T.g_rho[e] += gr;
            } else {
                T.g_mu[e] = gm;
                T.g_rho[e] = gr;
            }
        }
    }
}

static inline int64_t chunks_of(int64_t n) { return n == 0 ? 1 : (n + kKlChunk - 1) / kKlChunk; }

static int validate(const bnn_kl_tensor_t *tensors, int ntensors, const char *who)
{
    if (!tensors) { set_error("%s: NULL tensor list", who); return BNN_E_NULL; }
    if (ntensors < 1 || ntensors > kKlMaxTensors) { set_error("%s: ntensors must be in [1, %d]", who, kKlMaxTensors); return ntensors < 1 ? BNN_E_SHAPE : BNN_E_UNSUPPORTED; }
    int64_t blocks = 0;
    for (int t = 0; t < ntensors; ++t) {
        if (!tensors[t].mu || !tensors[t].rho) { set_error("%s: tensor %d NULL", who, t); return BNN_E_NULL; }
        if (tensors[t].n < 1) { set_error("%s: tensor %d empty", who, t); return BNN_E_SHAPE; }
        if (!(tensors[t].prior_sigma > 0.f)) { set_error("%s: tensor %d prior_sigma <= 0", who, t); return BNN_E_RANGE; }
        blocks += chunks_of(tensors[t].n);
    }
    if (blocks > 0x7FFFFFFF) { set_error("%s: too many elements", who); return BNN_E_RANGE; }
    return BNN_OK;
}

}  // namespace bnn

using namespace bnn;

extern "C" {

int64_t bnn_kl_workspace_bytes(int ntensors)
{
    (void)ntensors;
    // One double per 2048-scalar chunk; sized for 2^31 scalars in total plus one chunk
    // of slack per tensor.  (Callers that know their model may pass exactly
    // 8 * sum_t ceil(n_t / 2048) bytes.)
    return 8 * ((int64_t)(1ll << 31) / kKlChunk + kKlMaxTensors);
}

// First pass: the partial launches (launch = true), and / or the description of where each tensor's partials lie
// (F) -- a pure function of the tensor sizes, so bnn_mc_sum_kl can rebuild it for the second pass.
static int kl_first_pass(const bnn_kl_tensor_t *tensors, int ntensors, double *partials, hipStream_t st, KlFinal &F, bool launch,
                         const char *who)
{
    int32_t pbase = 0;
    int64_t total = 0;
    for (int t = 0; t < ntensors; ++t) total += tensors[t].n;
    const bool big = total >= ((int64_t)8 << 20);                  // 8192-scalar workgroups once the chip is full anyway
    const bool huge = total >= ((int64_t)32 << 20);                // 32768-scalar workgroups (>= 1024 of them)
    // below 8 Mi scalars: 4096-scalar workgroups from 1 Mi on (the MLP's 2.4 M: step 0.1028 -> 0.1011 ms against 2048; 8192
    // was slower again), 2048 for small models so that they still spread over the chip
    // below 8 Mi scalars: 2048-scalar workgroups (a thread = eight consecutive scalars): the partial-sum layout the draw
    // launch's items reproduce (bnn_draw_multi with kl_tensors).  (4096-scalar workgroups from 1 Mi scalars on were 1.7 %
    // faster on the stand-alone KL of the 2.4-M MLP; in the step that first pass no longer reads the posteriors at all.)
    const int small_pt = 8;
    const int64_t chunk = huge ? kKlThreads * 128 : big ? kKlThreads * 32 : kKlThreads * small_pt;
    for (int g0 = 0; g0 < ntensors; g0 += kKlMaxPerLaunch) {
        KlLaunch L{};
        const int cnt = ntensors - g0 < kKlMaxPerLaunch ? ntensors - g0 : kKlMaxPerLaunch;
        L.ntensors = cnt;
        L.partial_base = pbase;
        int32_t blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            const bnn_kl_tensor_t &s = tensors[g0 + i];
            L.t[i].mu = s.mu; L.t[i].rho = s.rho; L.t[i].g_mu = nullptr; L.t[i].g_rho = nullptr;
            L.t[i].n = s.n; L.t[i].prior_mu = s.prior_mu; L.t[i].prior_sigma = s.prior_sigma;
            L.t[i].first_block = blocks; L.t[i].scale = 0.f;
            F.first[g0 + i] = pbase + blocks;
            F.n[g0 + i] = s.n;
            blocks += (int32_t)((s.n + chunk - 1) / chunk);
        }
        if (launch) {
            if (huge) hipLaunchKernelGGL((k_kl_partial<32, 4>), dim3(blocks), dim3(kKlThreads), 0, st, L, partials);
            else if (big) hipLaunchKernelGGL(k_kl_partial<32>, dim3(blocks), dim3(kKlThreads), 0, st, L, partials);
            else hipLaunchKernelGGL(k_kl_partial<8>, dim3(blocks), dim3(kKlThreads), 0, st, L, partials);
            const int rc = check_launch(who);
            if (rc) return rc;
        }
        pbase += blocks;
    }
    F.first[ntensors] = pbase;
    return BNN_OK;
}

}  // extern "C"

namespace bnn {
bool kl_plan_piggy(const bnn_kl_tensor_t *tensors, int ntensors, void *workspace, KlPiggy &P)
{
    P.nblocks = 0;
    P.taken = 0;
    if (!tensors || !workspace || ntensors < 1 || ntensors > kKlPiggyMax) return false;
    if (validate(tensors, ntensors, "kl_plan_piggy")) return false;
    int64_t total = 0;
    for (int t = 0; t < ntensors; ++t) total += tensors[t].n;
    if (total >= ((int64_t)8 << 20)) return false;                 // big models: their own launch fills the chip
    const int pt = 8;                                              // == kl_first_pass's rule (the second pass rebuilds it)
    const int64_t chunk = (int64_t)kKlThreads * pt;
    int32_t blocks = 0;
    for (int i = 0; i < ntensors; ++i) {
        const bnn_kl_tensor_t &s = tensors[i];
        P.t[i].mu = s.mu; P.t[i].rho = s.rho; P.t[i].g_mu = nullptr; P.t[i].g_rho = nullptr;
        P.t[i].n = s.n; P.t[i].prior_mu = s.prior_mu; P.t[i].prior_sigma = s.prior_sigma;
        P.t[i].first_block = blocks; P.t[i].scale = 0.f;
        P.pg_first[i] = blocks;                                    // (every tensor launched; a carrier that computes some itself re-packs)
        blocks += (int32_t)((s.n + chunk - 1) / chunk);
    }
    P.pg_first[ntensors] = blocks;
    P.ntensors = ntensors;
    P.nblocks = blocks;
    P.pt = pt;
    P.partials = reinterpret_cast<double *>(workspace);
    return true;
}
}  // namespace bnn

extern "C" {

int bnn_kl_forward(const bnn_kl_tensor_t *tensors, int ntensors, float n_batches, float *out,
                   void *workspace, void *stream)
{
    int rc = validate(tensors, ntensors, "bnn_kl_forward");
    if (rc) return rc;
    if (!out || !workspace) { set_error("bnn_kl_forward: NULL out / workspace"); return BNN_E_NULL; }
    if (!(n_batches > 0.f)) { set_error("bnn_kl_forward: n_batches <= 0"); return BNN_E_RANGE; }
    hipStream_t st = (hipStream_t)stream;
    double *partials = reinterpret_cast<double *>(workspace);
    KlFinal F{};
    F.ntensors = ntensors;
    F.n_batches = n_batches;
    rc = kl_first_pass(tensors, ntensors, partials, st, F, true, "bnn_kl_forward(partial)");
    if (rc) return rc;
    hipLaunchKernelGGL(k_kl_final, dim3(1), dim3(kKlThreads), 0, st, F, partials, out);
    return check_launch("bnn_kl_forward(final)");
}

int bnn_kl_forward_partial(const bnn_kl_tensor_t *tensors, int ntensors, void *workspace, void *stream)
{
    int rc = validate(tensors, ntensors, "bnn_kl_forward_partial");
    if (rc) return rc;
    if (!workspace) { set_error("bnn_kl_forward_partial: NULL workspace"); return BNN_E_NULL; }
    KlFinal F{};
    return kl_first_pass(tensors, ntensors, reinterpret_cast<double *>(workspace), (hipStream_t)stream, F, true,
                         "bnn_kl_forward_partial");
}

int bnn_mc_sum_kl(const float *y, int64_t y_sample_stride, int nsamples, int64_t n, float scale, float *out, int accumulate,
                  uint32_t *advance_epoch, uint32_t advance_inc, const bnn_kl_tensor_t *tensors, int ntensors,
                  float n_batches, float *kl_out, const void *workspace, void *stream)
{
    if (!y || !out) { set_error("bnn_mc_sum_kl: NULL pointer"); return BNN_E_NULL; }
    if (n < 1 || nsamples < 1) { set_error("bnn_mc_sum_kl: bad extent"); return BNN_E_SHAPE; }
    int rc = validate(tensors, ntensors, "bnn_mc_sum_kl");
    if (rc) return rc;
    if (!kl_out || !workspace) { set_error("bnn_mc_sum_kl: NULL kl_out / workspace"); return BNN_E_NULL; }
    if (!(n_batches > 0.f)) { set_error("bnn_mc_sum_kl: n_batches <= 0"); return BNN_E_RANGE; }
    KlFinal F{};
    F.ntensors = ntensors;
    F.n_batches = n_batches;
    kl_first_pass(tensors, ntensors, nullptr, nullptr, F, false, "bnn_mc_sum_kl");
    int64_t b = (n + kMcThreads - 1) / kMcThreads;
    if (b > 2048) b = 2048;
    if (nsamples > kMcSplitAbove) {
        if (nsamples > 4 * kMcSplitMax) { set_error("bnn_mc_sum_kl: more than %d addends per output", 4 * kMcSplitMax); return BNN_E_RANGE; }
        b = (n + 63) / 64;
        if (b > 0x7FFFFFF0) { set_error("bnn_mc_sum_kl: too many outputs"); return BNN_E_RANGE; }
    }
    hipLaunchKernelGGL(k_mc_sum_kl, dim3((unsigned)b + 1), dim3(kKlThreads), 0, (hipStream_t)stream, y, y_sample_stride, nsamples, n,
                       scale, out, accumulate, advance_epoch, advance_inc, F, reinterpret_cast<const double *>(workspace), kl_out);
    return check_launch("bnn_mc_sum_kl");
}

int bnn_kl_backward(const bnn_kl_tensor_t *tensors, int ntensors, float n_batches,
                    const float *upstream, float *const *g_mu, float *const *g_rho, int accumulate,
                    void *stream)
{
    int rc = validate(tensors, ntensors, "bnn_kl_backward");
    if (rc) return rc;
    if (!g_mu || !g_rho) { set_error("bnn_kl_backward: NULL gradient lists"); return BNN_E_NULL; }
    if (!(n_batches > 0.f)) { set_error("bnn_kl_backward: n_batches <= 0"); return BNN_E_RANGE; }
    for (int t = 0; t < ntensors; ++t)
        if (!g_mu[t] || !g_rho[t]) { set_error("bnn_kl_backward: tensor %d NULL gradient", t); return BNN_E_NULL; }
    hipStream_t st = (hipStream_t)stream;
    for (int g0 = 0; g0 < ntensors; g0 += kKlMaxPerLaunch) {
        KlLaunch L{};
        const int cnt = ntensors - g0 < kKlMaxPerLaunch ? ntensors - g0 : kKlMaxPerLaunch;
        L.ntensors = cnt;
        L.partial_base = 0;
        int32_t blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            const bnn_kl_tensor_t &s = tensors[g0 + i];
            L.t[i].mu = s.mu; L.t[i].rho = s.rho; L.t[i].g_mu = g_mu[g0 + i]; L.t[i].g_rho = g_rho[g0 + i];
            L.t[i].n = s.n; L.t[i].prior_mu = s.prior_mu; L.t[i].prior_sigma = s.prior_sigma;
            L.t[i].first_block = blocks;
            L.t[i].scale = (float)(1.0 / ((double)s.n * (double)ntensors * (double)n_batches));
            blocks += (int32_t)chunks_of(s.n);
        }
        hipLaunchKernelGGL(k_kl_backward, dim3(blocks), dim3(kKlThreads), 0, st, L, upstream, accumulate);
        rc = check_launch("bnn_kl_backward");
        if (rc) return rc;
    }
    return BNN_OK;
}

}  // extern "C"
