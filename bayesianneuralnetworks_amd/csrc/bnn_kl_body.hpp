// bnn_kl_body.hpp -- first pass of K3 (per-workgroup partial KL sums) as a device function, shared by k_kl_partial
// (bnn_kl.hip) and the narrow-layer GEMM launch that carries it piggyback (bnn_linear.hip).
#pragma once
#include "bnn_device.hpp"

namespace bnn {

constexpr int kKlThreads = 256;
constexpr int kKlPerThread = 8;                        // 2 x float4
constexpr int kKlChunk = kKlThreads * kKlPerThread;    // 2048 scalars per workgroup
constexpr int kKlMaxPerLaunch = 64;
constexpr int kKlMaxTensors = 128;

struct KlTensorDev {
    const float *mu;
    const float *rho;
    float *g_mu;      // backward only
    float *g_rho;     // backward only
    int64_t n;
    float prior_mu, prior_sigma;
    int32_t first_block;   // first workgroup of this tensor within the launch
    float scale;           // backward: 1 / (n * ntensors * n_batches)
};
struct KlLaunch {
    KlTensorDev t[kKlMaxPerLaunch];
    int32_t ntensors;
    int32_t partial_base;  // index of this launch's first partial in the workspace
};
struct KlFinal {
    int32_t first[kKlMaxTensors + 1];  // partial ranges
    int64_t n[kKlMaxTensors];
    int32_t ntensors;
    float n_batches;
};

__device__ __forceinline__ float kl_elem(float mu, float rho, float pm, float inv_ps)
{
    const float sg = sigma_accurate(rho);
    const float r0 = sg * inv_ps;
    const float vr = r0 * r0;
    const float t0 = (mu - pm) * inv_ps;
    // ln(vr) = 2 ln(r0) on the native log2 unit (relative 1e-7; |ln| is O(1) or the term is tiny)
    const float lnvr = __builtin_amdgcn_logf(r0) * (2.0f * 0.693147180559945309f);
    return 0.5f * (vr + t0 * t0 - 1.0f - lnvr);
}

__device__ __forceinline__ int find_tensor(const KlLaunch &L, int block)
{
    int t = 0;
    for (int i = 1; i < L.ntensors; ++i)
        if (block >= L.t[i].first_block) t = i;
    return t;
}

// (first 256 threads of a workgroup; `block` = index within the KL launch, `partial_index` = slot in the workspace)
// PT scalars per thread: 8 (2048 per workgroup: small models stay spread over the chip) or 32 (8192: large
// tensors -- more loads in flight per thread, 4x fewer block reductions; 64 Mi scalars: 2.3 -> see DESIGN TB/s)
// REP > 1 (very large models): the workgroup walks REP consecutive PT-blocks and reduces ONCE -- the wave / LDS
// reduction and the workgroup's drain are a fixed cost per workgroup (2048 -> 8192 scalars per workgroup was 2.3 ->
// 4.1 TB/s on 64 Mi scalars); each block's fp32 thread sum is added to a double, so precision does not depend on REP.
template <int PT, int REP = 1>
__device__ __forceinline__ void kl_partial_block(const KlTensorDev &T, int block, int partial_index, double *__restrict__ partials)
{
    __shared__ double red[kKlThreads / 64];
    const int64_t base0 = (int64_t)(block - T.first_block) * (kKlThreads * PT * REP);
    // the two uniform operands of kl_elem as VGPR values: a VALU instruction that reads an SGPR issues ~1.4 x slower
    // on gfx950 (tools/ubench_valu.hip)
    const float inv_ps = __uint_as_float(uniform_vgpr(__float_as_uint(1.0f / T.prior_sigma)));
    const float pmu = __uint_as_float(uniform_vgpr(__float_as_uint(T.prior_mu)));
    const bool vec = ((reinterpret_cast<uintptr_t>(T.mu) | reinterpret_cast<uintptr_t>(T.rho)) & 15u) == 0;
    double dacc = 0.0;
    if constexpr (PT == 8 && REP == 1) {
        // 2048-scalar workgroups: a thread takes EIGHT CONSECUTIVE scalars (two adjacent 16-B loads of mu and of rho) and adds
        // their terms in index order -- the item of the draw launch (bnn_dense.hip, k_draw_multi), which holds the same eight
        // (mu, rho) in registers and leaves the same partial sum without reading them again (kl_block_reduce below)
        const int64_t e = base0 + (int64_t)threadIdx.x * 8;
        float acc = 0.f;
        if (vec && e + 8 <= T.n) {
            const float4 m0 = *reinterpret_cast<const float4 *>(T.mu + e), m1 = *reinterpret_cast<const float4 *>(T.mu + e + 4);
            const float4 r0 = *reinterpret_cast<const float4 *>(T.rho + e), r1 = *reinterpret_cast<const float4 *>(T.rho + e + 4);
            acc += kl_elem(m0.x, r0.x, pmu, inv_ps); acc += kl_elem(m0.y, r0.y, pmu, inv_ps);
            acc += kl_elem(m0.z, r0.z, pmu, inv_ps); acc += kl_elem(m0.w, r0.w, pmu, inv_ps);
            acc += kl_elem(m1.x, r1.x, pmu, inv_ps); acc += kl_elem(m1.y, r1.y, pmu, inv_ps);
            acc += kl_elem(m1.z, r1.z, pmu, inv_ps); acc += kl_elem(m1.w, r1.w, pmu, inv_ps);
        } else {
            for (int j = 0; j < 8; ++j)
                if (e + j < T.n) acc += kl_elem(T.mu[e + j], T.rho[e + j], pmu, inv_ps);
        }
        dacc = (double)acc;
    } else {
#pragma unroll 1
    for (int rep = 0; rep < REP; ++rep) {
    const int64_t base = base0 + (int64_t)rep * (kKlThreads * PT);
    if (REP > 1 && base >= T.n) break;
    float acc = 0.f;
    if (vec && base + (int64_t)kKlThreads * PT <= T.n) {
        // interior workgroup: 16-B loads in batches of LB (mu, rho) pairs, all requested before the first use -- with one
        // pair per wait a thread had 32 B in flight and the stream ran at half the rate it reaches with 4 pairs
        constexpr int LB = PT / 4 < 4 ? PT / 4 : 4;
#pragma unroll
        for (int it0 = 0; it0 < PT / 4; it0 += LB) {
            float4 m[LB], r[LB];
#pragma unroll
            for (int j = 0; j < LB; ++j) {
                const int64_t e = base + ((int64_t)(it0 + j) * kKlThreads + threadIdx.x) * 4;
                m[j] = *reinterpret_cast<const float4 *>(T.mu + e);
                r[j] = *reinterpret_cast<const float4 *>(T.rho + e);
            }
#pragma unroll
            for (int j = 0; j < LB; ++j) {
                acc += kl_elem(m[j].x, r[j].x, pmu, inv_ps);
                acc += kl_elem(m[j].y, r[j].y, pmu, inv_ps);
                acc += kl_elem(m[j].z, r[j].z, pmu, inv_ps);
                acc += kl_elem(m[j].w, r[j].w, pmu, inv_ps);
            }
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < PT / 4; ++it) {
            const int64_t e = base + ((int64_t)it * kKlThreads + threadIdx.x) * 4;
            if (vec && e + 4 <= T.n) {
                const float4 m = *reinterpret_cast<const float4 *>(T.mu + e);
                const float4 r = *reinterpret_cast<const float4 *>(T.rho + e);
                acc += kl_elem(m.x, r.x, pmu, inv_ps);
                acc += kl_elem(m.y, r.y, pmu, inv_ps);
                acc += kl_elem(m.z, r.z, pmu, inv_ps);
                acc += kl_elem(m.w, r.w, pmu, inv_ps);
            } else {
                for (int j = 0; j < 4; ++j)
                    if (e + j < T.n) acc += kl_elem(T.mu[e + j], T.rho[e + j], pmu, inv_ps);
            }
        }
    }
    dacc += (double)acc;
    }
    }
    double d = wave_sum(dacc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kKlThreads / 64; ++w) s += red[w];
        partials[partial_index] = s;
    }
}

// The workgroup reduction of kl_partial_block on a thread sum computed elsewhere (the draw launch's items): same order, same
// value.  All 256 threads of the workgroup call it.
__device__ __forceinline__ void kl_block_reduce(float thread_sum, int partial_index, double *__restrict__ partials)
{
    __shared__ double red[kKlThreads / 64];
    double d = wave_sum((double)thread_sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < kKlThreads / 64; ++w) s += red[w];
        partials[partial_index] = s;
    }
}

// The KL first pass carried by another kernel's launch (<= kKlPiggyMax tensors, 2048- or 4096-scalar workgroups).
constexpr int kKlPiggyMax = 8;
struct KlPiggy {
    KlTensorDev t[kKlPiggyMax];
    int32_t ntensors;
    int32_t nblocks;       // workgroups to launch: 0 = nothing to carry
    int32_t pt;            // 8 (2048-scalar workgroups)
    int32_t taken;         // host side: set by the launcher that carried it
    // pg_first[i] .. pg_first[i + 1]: the launched workgroups of tensor i (none for a tensor whose partial sums the carrying
    // launch computes itself: the draw launch's items); t[i].first_block stays the tensor's first PARTIAL index
    int32_t pg_first[kKlPiggyMax + 1];
    double *partials;
};

__device__ __forceinline__ void kl_piggy_block(const KlPiggy &P, int block)
{
    int t = 0;
    for (int i = 1; i < P.ntensors; ++i)
        if (block >= P.pg_first[i]) t = i;
    const int b = P.t[t].first_block + (block - P.pg_first[t]);
    kl_partial_block<8>(P.t[t], b, b, P.partials);
}

// host: plan of a piggyback first pass (bnn_kl.hip); false = not eligible (caller launches bnn_kl_forward_partial)
bool kl_plan_piggy(const bnn_kl_tensor_t *tensors, int ntensors, void *workspace, KlPiggy &P);

}  // namespace bnn
