// ubench_store.hip -- what shape of kernel moves the draw launch's bytes (19.2 MB of fp32 mu / rho read, 38.3 MB of bf16 written
// as 8 planes) at the rate the chip reaches on a plain copy?  Diagnostic, not part of the product path.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_store.hip -o tools/ubench_store && tools/ubench_store
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int S = 8;

// (A) one item (8 scalars) per thread, S stores of 16 B to S planes: the draw kernel's shape.  VALU: `work` dependent fmas per sample.
template <int WORK, bool LOAD>
__global__ __launch_bounds__(256) void k_item_per_thread(const float *__restrict__ mu, const float *__restrict__ rho, uint4 *__restrict__ out,
                                                        int64_t items, int64_t plane)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= items) return;
    float4 a = make_float4(1.f, 2.f, 3.f, 4.f), b = a;
    if (LOAD) {
        a = *reinterpret_cast<const float4 *>(mu + i * 8);
        const float4 a2 = *reinterpret_cast<const float4 *>(mu + i * 8 + 4);
        b = *reinterpret_cast<const float4 *>(rho + i * 8);
        const float4 b2 = *reinterpret_cast<const float4 *>(rho + i * 8 + 4);
        a.x += a2.x; a.y += a2.y; b.x += b2.z; b.y += b2.w;
    }
    for (int s = 0; s < S; ++s) {
        float v = a.x + (float)s;
#pragma unroll
        for (int w = 0; w < WORK; ++w) v = fmaf(v, b.y, a.z);
        uint4 o = make_uint4(__float_as_uint(v), __float_as_uint(a.y), __float_as_uint(b.x), __float_as_uint(b.w));
        out[i + s * plane] = o;
    }
}

// (B) grid-stride: a fixed grid of `gridDim.x` workgroups, each thread walks items i, i + T, ...; loads of the next item are
// requested before the stores of the current one
template <int WORK>
__global__ __launch_bounds__(256) void k_grid_stride(const float *__restrict__ mu, const float *__restrict__ rho, uint4 *__restrict__ out,
                                                    int64_t items, int64_t plane)
{
    const int64_t T = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= items) return;
    float4 a = *reinterpret_cast<const float4 *>(mu + i * 8), a2 = *reinterpret_cast<const float4 *>(mu + i * 8 + 4);
    float4 b = *reinterpret_cast<const float4 *>(rho + i * 8), b2 = *reinterpret_cast<const float4 *>(rho + i * 8 + 4);
    while (true) {
        const int64_t n = i + T;
        float4 na = a, na2 = a2, nb = b, nb2 = b2;
        if (n < items) {
            na = *reinterpret_cast<const float4 *>(mu + n * 8); na2 = *reinterpret_cast<const float4 *>(mu + n * 8 + 4);
            nb = *reinterpret_cast<const float4 *>(rho + n * 8); nb2 = *reinterpret_cast<const float4 *>(rho + n * 8 + 4);
        }
        const float c0 = a.x + a2.x, c1 = b.y + b2.w;
        for (int s = 0; s < S; ++s) {
            float v = c0 + (float)s;
#pragma unroll
            for (int w = 0; w < WORK; ++w) v = fmaf(v, c1, a.z);
            out[i + s * plane] = make_uint4(__float_as_uint(v), __float_as_uint(a.y), __float_as_uint(b.x), __float_as_uint(b.w));
        }
        if (n >= items) break;
        i = n; a = na; a2 = na2; b = nb; b2 = nb2;
    }
}

// (C) plain fill: one 16-B store per thread over the whole output
__global__ __launch_bounds__(256) void k_fill(uint4 *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = make_uint4(1u, 2u, 3u, (uint32_t)i);
}

// (D) sample on blockIdx.y: one store per thread, loads repeated per sample
template <int WORK>
__global__ __launch_bounds__(256) void k_sample_y(const float *__restrict__ mu, const float *__restrict__ rho, uint4 *__restrict__ out,
                                                 int64_t items, int64_t plane)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= items) return;
    const int s = blockIdx.y;
    const float4 a = *reinterpret_cast<const float4 *>(mu + i * 8), a2 = *reinterpret_cast<const float4 *>(mu + i * 8 + 4);
    const float4 b = *reinterpret_cast<const float4 *>(rho + i * 8), b2 = *reinterpret_cast<const float4 *>(rho + i * 8 + 4);
    float v = a.x + a2.x + (float)s;
#pragma unroll
    for (int w = 0; w < WORK; ++w) v = fmaf(v, b.y + b2.w, a.z);
    out[i + s * plane] = make_uint4(__float_as_uint(v), __float_as_uint(a.y), __float_as_uint(b.x), __float_as_uint(b.w));
}

template <typename F>
static float time_us(F launch, int iters = 50)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

int main()
{
    const int64_t scalars = 2395210 / 8 * 8;
    const int64_t items = scalars / 8;
    const int64_t plane = items;             // uint4 per plane
    float *mu, *rho;
    uint4 *out;
    CK(hipMalloc(&mu, scalars * 4)); CK(hipMalloc(&rho, scalars * 4)); CK(hipMalloc(&out, items * 16 * S));
    CK(hipMemset(mu, 0, scalars * 4)); CK(hipMemset(rho, 0, scalars * 4));
    const unsigned nb = (unsigned)((items + 255) / 256);
    printf("items %lld, %u workgroups of 256; bytes read %.1f MB, written %.1f MB\n", (long long)items, nb, scalars * 8 / 1e6, items * 16.0 * S / 1e6);
    printf("(C) fill of the 8 planes, one 16-B store per thread          : %6.2f us\n", time_us([&] { hipLaunchKernelGGL(k_fill, dim3(nb * S), dim3(256), 0, 0, out, items * S); }));
    printf("(A) item per thread, no loads,   0 fma per sample            : %6.2f us\n", time_us([&] { hipLaunchKernelGGL((k_item_per_thread<0, false>), dim3(nb), dim3(256), 0, 0, mu, rho, out, items, plane); }));
    printf("(A) item per thread, loads,      0 fma per sample            : %6.2f us\n", time_us([&] { hipLaunchKernelGGL((k_item_per_thread<0, true>), dim3(nb), dim3(256), 0, 0, mu, rho, out, items, plane); }));
    printf("(A) item per thread, loads,     32 fma per sample            : %6.2f us\n", time_us([&] { hipLaunchKernelGGL((k_item_per_thread<32, true>), dim3(nb), dim3(256), 0, 0, mu, rho, out, items, plane); }));
    printf("(A) item per thread, loads,    128 fma per sample            : %6.2f us\n", time_us([&] { hipLaunchKernelGGL((k_item_per_thread<128, true>), dim3(nb), dim3(256), 0, 0, mu, rho, out, items, plane); }));
    for (int g : {256, 512, 1024, 2048, 4096}) {
        printf("(B) grid-stride, %4d workgroups, 0 fma                      : %6.2f us\n", g, time_us([&] { hipLaunchKernelGGL((k_grid_stride<0>), dim3(g), dim3(256), 0, 0, mu, rho, out, items, plane); }));
        printf("(B) grid-stride, %4d workgroups, 128 fma                    : %6.2f us\n", g, time_us([&] { hipLaunchKernelGGL((k_grid_stride<128>), dim3(g), dim3(256), 0, 0, mu, rho, out, items, plane); }));
    }
    printf("(D) sample on grid.y, one store per thread, 0 fma            : %6.2f us\n", time_us([&] { hipLaunchKernelGGL((k_sample_y<0>), dim3(nb, S), dim3(256), 0, 0, mu, rho, out, items, plane); }));
    printf("(D) sample on grid.y, one store per thread, 128 fma          : %6.2f us\n", time_us([&] { hipLaunchKernelGGL((k_sample_y<128>), dim3(nb, S), dim3(256), 0, 0, mu, rho, out, items, plane); }));
    return 0;
}
