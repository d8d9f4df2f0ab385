// bnn_mc.hpp -- body of the MC reduction (shared by k_mc_sum, bnn_elementwise.hip, and the step-tail kernel k_mc_sum_kl,
// bnn_kl.hip).
#pragma once
#include "bnn_device.hpp"

namespace bnn {

constexpr int kMcThreads = 256;

// out[i] (+)= scale * sum_s y[s * stride + i] for the outputs of workgroup `block` of `nblocks`.
__device__ __forceinline__ void mc_sum_body(const float *__restrict__ y, int64_t y_sample_stride, int nsamples, int64_t n,
                                            float scale, float *__restrict__ out, int accumulate, int block, int nblocks)
{
    const int64_t tid = (int64_t)block * kMcThreads + threadIdx.x;
    const int64_t nthreads = (int64_t)nblocks * kMcThreads;
    for (int64_t i = tid; i < n; i += nthreads) {
        // eight loads in flight, added in sample order (a runtime-bound loop of dependent adds paid one memory round
        // trip per sample: 4.3 us for the step's (8, 512, 10) reduction)
        float a = 0.f;
        int s = 0;
        for (; s + 8 <= nsamples; s += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = y[(int64_t)(s + j) * y_sample_stride + i];
#pragma unroll
            for (int j = 0; j < 8; ++j) a += v[j];
        }
        for (; s < nsamples; ++s) a += y[(int64_t)s * y_sample_stride + i];
        a *= scale;
        out[i] = accumulate ? out[i] + a : a;
    }
}

// The same reduction for MANY addends per output (a fused head's ntn * NWN * S partial logits: 128 at the BASELINE layer): a
// workgroup takes 64 outputs, its four waves a quarter of the addends each -- every load of a wave's quarter is requested before
// the first add (one memory round trip instead of nsamples / 8) -- and the four sums are added in wave order through LDS:
// fixed order, bitwise reproducible.  Launch with ceil(n / 64) workgroups of 256 threads.
constexpr int kMcSplitAbove = 32;       // addends per output from which the launchers take this body
constexpr int kMcSplitMax = 64;         // addends per wave held in registers: nsamples <= 4 * 64

__device__ __forceinline__ void mc_sum_split_body(const float *__restrict__ y, int64_t y_sample_stride, int nsamples, int64_t n,
                                                  float scale, float *__restrict__ out, int accumulate, int block)
{
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = (int64_t)block * 64 + lane;
    const int per = (nsamples + 3) >> 2;
    const int s0 = grp * per, s1 = s0 + per < nsamples ? s0 + per : nsamples;
    float a = 0.f;
    if (i < n) {
        float v[kMcSplitMax];
#pragma unroll
        for (int j = 0; j < kMcSplitMax; ++j) v[j] = (s0 + j < s1) ? y[(int64_t)(s0 + j) * y_sample_stride + i] : 0.f;
#pragma unroll
        for (int j = 0; j < kMcSplitMax; ++j) a += v[j];
    }
    part[grp][lane] = a;
    __syncthreads();
    if (grp == 0 && i < n) {
        const float t = (((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) * scale;
        out[i] = accumulate ? out[i] + t : t;
    }
}

}  // namespace bnn
