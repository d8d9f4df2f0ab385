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

}  // namespace bnn
