// ubench_draw.hip -- shapes of the draw kernel on ONE regular tensor of the BASELINE net's size (2.4 M scalars, 8 samples, bf16
// out), with per-wave start / loaded / end stamps (s_memrealtime, 100 MHz).  Diagnostic, not part of the product path.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_draw.hip -o tools/ubench_draw && tools/ubench_draw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#include "../bayesianneuralnetworks_amd/csrc/bnn_device.hpp"
using namespace bnn;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int S = 8;

struct P {
    const float *mu, *rho;
    uint4 *out;
    int64_t items, plane;
    uint64_t *stamps;       // 3 per wave or NULL
    RngDev rng;
};

template <int GEN>
__device__ __forceinline__ void eps8(const RngDev &rng, const PhiloxKeys &keys, uint32_t blk, uint32_t c1, float4 &za, float4 &zb)
{
    if constexpr (GEN == 0) {
        const uint4 xa = philox4x32_10(make_uint4(blk, c1, rng.epoch_host, 0u), keys);
        const uint4 xb = philox4x32_10(make_uint4(blk + 1u, c1, rng.epoch_host, 0u), keys);
        box_muller(xa.x, xa.y, za.x, za.y); box_muller(xa.z, xa.w, za.z, za.w);
        box_muller(xb.x, xb.y, zb.x, zb.y); box_muller(xb.z, xb.w, zb.z, zb.w);
    } else if constexpr (GEN == 1) {
        const uint4 x = philox4x32_r<7>(make_uint4(blk >> 1, c1, rng.epoch_host, 0u), keys);
        box_muller16(x.x, za.x, za.y); box_muller16(x.y, za.z, za.w);
        box_muller16(x.z, zb.x, zb.y); box_muller16(x.w, zb.z, zb.w);
    } else {
        za = make_float4(__uint_as_float(0x3f800000u | (blk & 0xFFFFu)), 0.5f, 0.25f, (float)c1);
        zb = za;
    }
}

__device__ __forceinline__ uint4 affine_pack(const float *m, const float *sg, const float4 &za, const float4 &zb)
{
    uint4 o;
    o.x = pack_bf16x2(fmaf(sg[0], za.x, m[0]), fmaf(sg[1], za.y, m[1]));
    o.y = pack_bf16x2(fmaf(sg[2], za.z, m[2]), fmaf(sg[3], za.w, m[3]));
    o.z = pack_bf16x2(fmaf(sg[4], zb.x, m[4]), fmaf(sg[5], zb.y, m[5]));
    o.w = pack_bf16x2(fmaf(sg[6], zb.z, m[6]), fmaf(sg[7], zb.w, m[7]));
    return o;
}

// (A) the current shape: one item per thread, the S samples in a serial loop
template <int GEN, bool STORE>
__global__ __launch_bounds__(256) void k_A(const P p)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t t0 = 0, t1 = 0;
    if (p.stamps) t0 = __builtin_amdgcn_s_memrealtime();
    if (i >= p.items) return;
    const float4 m0 = *reinterpret_cast<const float4 *>(p.mu + i * 8), m1 = *reinterpret_cast<const float4 *>(p.mu + i * 8 + 4);
    const float4 r0 = *reinterpret_cast<const float4 *>(p.rho + i * 8), r1 = *reinterpret_cast<const float4 *>(p.rho + i * 8 + 4);
    const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
    const float sg[8] = {sigma_draw(r0.x), sigma_draw(r0.y), sigma_draw(r0.z), sigma_draw(r0.w),
                         sigma_draw(r1.x), sigma_draw(r1.y), sigma_draw(r1.z), sigma_draw(r1.w)};
    if (p.stamps) t1 = __builtin_amdgcn_s_memrealtime();
    const PhiloxKeys keys = philox_keys(p.rng.key0, p.rng.key1);
    const uint32_t blk = (uint32_t)(i * 2);
    float acc = 0.f;
    for (int s = 0; s < S; ++s) {
        float4 za, zb;
        eps8<GEN>(p.rng, keys, blk, p.rng.stream_hi | (uint32_t)s, za, zb);
        const uint4 o = affine_pack(m, sg, za, zb);
        if (STORE) p.out[i + s * p.plane] = o;
        else acc += __uint_as_float(o.x ^ o.y ^ o.z ^ o.w);
    }
    if (!STORE && acc == 1.2345e30f) p.out[i] = make_uint4(1u, 2u, 3u, 4u);
    if (p.stamps) {
        if (STORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t t2 = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0) {
            const int64_t w = i >> 6;
            p.stamps[3 * w] = t0; p.stamps[3 * w + 1] = t1; p.stamps[3 * w + 2] = t2;
        }
    }
}

// (B) persistent: gridDim.x workgroups, a wave walks 64-item groups w, w + W, ...; the next group's mu / rho are requested before
// the current group's samples are drawn
template <int GEN>
__global__ __launch_bounds__(256) void k_B(const P p)
{
    const int64_t T = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.items) return;
    const PhiloxKeys keys = philox_keys(p.rng.key0, p.rng.key1);
    float4 m0 = *reinterpret_cast<const float4 *>(p.mu + i * 8), m1 = *reinterpret_cast<const float4 *>(p.mu + i * 8 + 4);
    float4 r0 = *reinterpret_cast<const float4 *>(p.rho + i * 8), r1 = *reinterpret_cast<const float4 *>(p.rho + i * 8 + 4);
    while (true) {
        const int64_t n = i + T;
        const bool more = n < p.items;
        const int64_t nn = more ? n : i;
        const float4 nm0 = *reinterpret_cast<const float4 *>(p.mu + nn * 8), nm1 = *reinterpret_cast<const float4 *>(p.mu + nn * 8 + 4);
        const float4 nr0 = *reinterpret_cast<const float4 *>(p.rho + nn * 8), nr1 = *reinterpret_cast<const float4 *>(p.rho + nn * 8 + 4);
        const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        const float sg[8] = {sigma_draw(r0.x), sigma_draw(r0.y), sigma_draw(r0.z), sigma_draw(r0.w),
                             sigma_draw(r1.x), sigma_draw(r1.y), sigma_draw(r1.z), sigma_draw(r1.w)};
        const uint32_t blk = (uint32_t)(i * 2);
        for (int s = 0; s < S; ++s) {
            float4 za, zb;
            eps8<GEN>(p.rng, keys, blk, p.rng.stream_hi | (uint32_t)s, za, zb);
            p.out[i + s * p.plane] = affine_pack(m, sg, za, zb);
        }
        if (!more) break;
        i = n; m0 = nm0; m1 = nm1; r0 = nr0; r1 = nr1;
    }
}

// (C) half the samples per thread: 2 x the threads, each (item, sample half); sigma twice
template <int GEN>
__global__ __launch_bounds__(256) void k_C(const P p)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int half = (int)(blockIdx.x & 1);
    const int64_t i = (int64_t)(blockIdx.x >> 1) * 256 + threadIdx.x;
    (void)g;
    if (i >= p.items) return;
    const float4 m0 = *reinterpret_cast<const float4 *>(p.mu + i * 8), m1 = *reinterpret_cast<const float4 *>(p.mu + i * 8 + 4);
    const float4 r0 = *reinterpret_cast<const float4 *>(p.rho + i * 8), r1 = *reinterpret_cast<const float4 *>(p.rho + i * 8 + 4);
    const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
    const float sg[8] = {sigma_draw(r0.x), sigma_draw(r0.y), sigma_draw(r0.z), sigma_draw(r0.w),
                         sigma_draw(r1.x), sigma_draw(r1.y), sigma_draw(r1.z), sigma_draw(r1.w)};
    const PhiloxKeys keys = philox_keys(p.rng.key0, p.rng.key1);
    const uint32_t blk = (uint32_t)(i * 2);
    for (int s = half * (S / 2); s < (half + 1) * (S / 2); ++s) {
        float4 za, zb;
        eps8<GEN>(p.rng, keys, blk, p.rng.stream_hi | (uint32_t)s, za, zb);
        p.out[i + s * p.plane] = affine_pack(m, sg, za, zb);
    }
}

// (D) shape (A) on a PADDED matrix like the real launch: rows x cols posterior, output rows of ld >= cols columns (groups beyond
// cols are written as zeros), plane stride given.  PADMODE 0: pad threads leave through a zero-fill loop of their own (the
// current kernel); 1: pad threads run the same loop with mean = sigma = 0 (no divergence)
template <int GEN, int PADMODE>
__global__ __launch_bounds__(256) void k_D(const P p, int rows, int cols, int ld, int64_t plane_bytes)
{
    const int local = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int gpr = ld >> 3;
    const int row = local / gpr, c0 = (local - row * gpr) << 3;
    if (row >= rows) return;
    char *dst = reinterpret_cast<char *>(p.out) + ((int64_t)row * ld + c0) * 2;
    const bool pad = c0 >= cols;
    if (PADMODE == 0 && pad) {
        for (int s = 0; s < S; ++s) *reinterpret_cast<uint4 *>(dst + s * plane_bytes) = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const int64_t e0 = pad ? 0 : (int64_t)row * cols + c0;
    const float4 m0 = *reinterpret_cast<const float4 *>(p.mu + e0), m1 = *reinterpret_cast<const float4 *>(p.mu + e0 + 4);
    const float4 r0 = *reinterpret_cast<const float4 *>(p.rho + e0), r1 = *reinterpret_cast<const float4 *>(p.rho + e0 + 4);
    const float z = pad ? 0.f : 1.f;
    const float m[8] = {z * m0.x, z * m0.y, z * m0.z, z * m0.w, z * m1.x, z * m1.y, z * m1.z, z * m1.w};
    const float sg[8] = {z * sigma_draw(r0.x), z * sigma_draw(r0.y), z * sigma_draw(r0.z), z * sigma_draw(r0.w),
                         z * sigma_draw(r1.x), z * sigma_draw(r1.y), z * sigma_draw(r1.z), z * sigma_draw(r1.w)};
    const PhiloxKeys keys = philox_keys(p.rng.key0, p.rng.key1);
    const uint32_t blk = (uint32_t)(e0 >> 2);
    for (int s = 0; s < S; ++s, dst += plane_bytes) {
        float4 za, zb;
        eps8<GEN>(p.rng, keys, blk, p.rng.stream_hi | (uint32_t)s, za, zb);
        *reinterpret_cast<uint4 *>(dst) = affine_pack(m, sg, za, zb);
    }
}

template <typename F>
static float time_us(F launch, int iters = 50)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

static void stamp_report(const char *name, const std::vector<uint64_t> &st, int64_t waves)
{
    uint64_t tmin = ~0ull, tmax = 0;
    for (int64_t w = 0; w < waves; ++w) { tmin = std::min(tmin, st[3 * w]); tmax = std::max(tmax, st[3 * w + 2]); }
    std::vector<double> start, load, life;
    for (int64_t w = 0; w < waves; ++w) {
        start.push_back((st[3 * w] - tmin) * 0.01);
        load.push_back((st[3 * w + 1] - st[3 * w]) * 0.01);
        life.push_back((st[3 * w + 2] - st[3 * w]) * 0.01);
    }
    auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
    printf("   %s: span %.2f us | wave start (us after the first) p50 %.2f p90 %.2f max %.2f | loads + sigma p50 %.2f p90 %.2f max %.2f | "
           "lifetime p10 %.2f p50 %.2f p90 %.2f max %.2f\n", name, (tmax - tmin) * 0.01, q(start, .5), q(start, .9), q(start, 1.),
           q(load, .5), q(load, .9), q(load, 1.), q(life, .1), q(life, .5), q(life, .9), q(life, 1.));
}

int main()
{
    const int64_t scalars = 2395210 / 8 * 8, items = scalars / 8, plane = items;
    float *mu, *rho;
    uint4 *out;
    uint64_t *stamps;
    const int64_t waves = (items + 63) / 64;
    CK(hipMalloc(&mu, scalars * 4)); CK(hipMalloc(&rho, scalars * 4)); CK(hipMalloc(&out, items * 16 * S)); CK(hipMalloc(&stamps, (waves + 8) * 24));
    std::vector<float> h(scalars);
    for (int64_t i = 0; i < scalars; ++i) h[i] = 0.05f * (float)((i * 2654435761u) % 1000) / 1000.f;
    CK(hipMemcpy(mu, h.data(), scalars * 4, hipMemcpyHostToDevice));
    for (int64_t i = 0; i < scalars; ++i) h[i] = -2.0f - 0.3f * (float)((i * 40503u) % 1000) / 1000.f;
    CK(hipMemcpy(rho, h.data(), scalars * 4, hipMemcpyHostToDevice));
    P p{mu, rho, out, items, plane, nullptr, RngDev{1234u, 5678u, 7u << 16, 0u, 3u, 0, nullptr}};
    const unsigned nb = (unsigned)((items + 255) / 256);
    printf("items %lld = %lld waves = %u workgroups\n", (long long)items, (long long)waves, nb);
#define RUN_A(GEN, STORE, label) \
    do { \
        p.stamps = nullptr; \
        printf("(A) item per thread, %s: %6.2f us\n", label, time_us([&] { hipLaunchKernelGGL((k_A<GEN, STORE>), dim3(nb), dim3(256), 0, 0, p); })); \
        p.stamps = stamps; \
        (void)hipMemset(stamps, 0, (waves + 8) * 24); \
        hipLaunchKernelGGL((k_A<GEN, STORE>), dim3(nb), dim3(256), 0, 0, p); \
        (void)hipDeviceSynchronize(); \
        std::vector<uint64_t> st(3 * waves); \
        (void)hipMemcpy(st.data(), stamps, 3 * waves * 8, hipMemcpyDeviceToHost); \
        stamp_report(label, st, waves); \
        p.stamps = nullptr; \
    } while (0)
    RUN_A(2, true, "no RNG, stores            ");
    RUN_A(1, false, "Philox-7 / u16, no stores ");
    RUN_A(1, true, "Philox-7 / u16, stores    ");
    RUN_A(0, false, "Philox-10 / u24, no stores");
    RUN_A(0, true, "Philox-10 / u24, stores   ");
    for (int g : {256, 512, 768, 1024}) {
        printf("(B) persistent %4d workgroups, Philox-7 / u16 : %6.2f us\n", g, time_us([&] { hipLaunchKernelGGL((k_B<1>), dim3(g), dim3(256), 0, 0, p); }));
        printf("(B) persistent %4d workgroups, Philox-10 / u24: %6.2f us\n", g, time_us([&] { hipLaunchKernelGGL((k_B<0>), dim3(g), dim3(256), 0, 0, p); }));
        printf("(B) persistent %4d workgroups, no RNG         : %6.2f us\n", g, time_us([&] { hipLaunchKernelGGL((k_B<2>), dim3(g), dim3(256), 0, 0, p); }));
    }
    {
        // 2000 x 1200 posterior (2.4 M scalars)
        const int rows = 2000, cols = 1200;
        uint4 *big;
        CK(hipMalloc(&big, (int64_t)rows * 1280 * 2 * S + (64 << 20)));
        P q = p; q.out = big;
        for (int ld : {1200, 1216}) {
            const unsigned nbd = (unsigned)(((int64_t)rows * (ld / 8) + 255) / 256);
            for (int64_t extra : {(int64_t)0, (int64_t)16, (int64_t)4096 - ((int64_t)rows * ld * 2) % 4096, (int64_t)(1 << 20) - ((int64_t)rows * ld * 2) % (1 << 20)}) {
                const int64_t pb = (int64_t)rows * ld * 2 + extra;
                printf("(D) 2000 x 1200, ld %d, plane stride %lld B (= 2^%d x odd): ", ld, (long long)pb, __builtin_ctzll(pb));
                printf("pad by zero-fill loop: P10 %6.2f  P7 %6.2f  none %6.2f us | pad in the main loop: P10 %6.2f  P7 %6.2f  none %6.2f us\n",
                       time_us([&] { hipLaunchKernelGGL((k_D<0, 0>), dim3(nbd), dim3(256), 0, 0, q, rows, cols, ld, pb); }),
                       time_us([&] { hipLaunchKernelGGL((k_D<1, 0>), dim3(nbd), dim3(256), 0, 0, q, rows, cols, ld, pb); }),
                       time_us([&] { hipLaunchKernelGGL((k_D<2, 0>), dim3(nbd), dim3(256), 0, 0, q, rows, cols, ld, pb); }),
                       time_us([&] { hipLaunchKernelGGL((k_D<0, 1>), dim3(nbd), dim3(256), 0, 0, q, rows, cols, ld, pb); }),
                       time_us([&] { hipLaunchKernelGGL((k_D<1, 1>), dim3(nbd), dim3(256), 0, 0, q, rows, cols, ld, pb); }),
                       time_us([&] { hipLaunchKernelGGL((k_D<2, 1>), dim3(nbd), dim3(256), 0, 0, q, rows, cols, ld, pb); }));
            }
        }
    }
    {
        // the PRODUCT kernel (libbnn_hip.so, bnn_draw_multi) on the same 2000 x 1200 tensor, same buffers, timed the same way
        const int rows = 2000, cols = 1200, ld = 1216;
        uint4 *big;
        CK(hipMalloc(&big, (int64_t)rows * ld * 2 * S));
        uint32_t *edev;
        CK(hipMalloc(&edev, 16)); CK(hipMemset(edev, 0, 16));
        for (int with_edev = 0; with_edev < 2; ++with_edev) {
            bnn_draw_tensor_t t{};
            t.mu = mu; t.rho = rho; t.rows = rows; t.cols = cols; t.out = big; t.ld = ld; t.out_sample_stride = (int64_t)rows * ld;
            t.out_dtype = BNN_BF16; t.kind = 0; t.taps = 0;
            t.rng.seed = 1234; t.rng.stream = 7; t.rng.sample0 = 0; t.rng.epoch_host = 3; t.rng.epoch_dev_delta = 0; t.rng.epoch_dev = with_edev ? edev : nullptr;
            printf("(P) product kernel bnn_draw_multi, 2000 x 1200, ld 1216, epoch word %s: %6.2f us\n", with_edev ? "in memory" : "NULL     ",
                   time_us([&] { bnn_draw_multi(&t, 1, S, nullptr, 0, nullptr, nullptr); }));
        }
    }
    printf("(C) half the samples per thread, Philox-7 / u16 : %6.2f us\n", time_us([&] { hipLaunchKernelGGL((k_C<1>), dim3(2 * nb), dim3(256), 0, 0, p); }));
    printf("(C) half the samples per thread, Philox-10 / u24: %6.2f us\n", time_us([&] { hipLaunchKernelGGL((k_C<0>), dim3(2 * nb), dim3(256), 0, 0, p); }));
    printf("(C) half the samples per thread, no RNG         : %6.2f us\n", time_us([&] { hipLaunchKernelGGL((k_C<2>), dim3(2 * nb), dim3(256), 0, 0, p); }));
    return 0;
}
