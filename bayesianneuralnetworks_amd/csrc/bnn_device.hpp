// bnn_device.hpp -- device-side building blocks shared by every kernel of
// libbnn_hip.so: Philox4x32-10, the Box-Muller eps draw, sigma = 1e-10 + softplus(rho).
// gfx950 (CDNA4) only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bnn_hip.h"

namespace bnn {

// ----- host-side plumbing -------------------------------------------------------
void set_error(const char *fmt, ...);
void count_launch();
int check_launch(const char *what);

// Device view of a bnn_rng_t.
struct RngDev {
    uint32_t key0, key1;       // seed lo / hi
    uint32_t stream_hi;        // stream << 16
    uint32_t sample0;
    uint32_t epoch_host;
    int32_t epoch_dev_delta;
    const uint32_t *epoch_dev;
    uint32_t gen;              // BNN_GEN_*
};

static inline RngDev make_rng(const bnn_rng_t *r)
{
    RngDev d{};
    if (r) {
        d.key0 = (uint32_t)r->seed;
        d.key1 = (uint32_t)(r->seed >> 32);
        d.stream_hi = r->stream << 16;
        d.sample0 = r->sample0;
        d.epoch_host = r->epoch_host;
        d.epoch_dev_delta = r->epoch_dev_delta;
        d.epoch_dev = r->epoch_dev;
        d.gen = r->generator;
    }
    return d;
}

static inline int check_rng(const bnn_rng_t *r, int nsamples)
{
    if (!r) return BNN_E_NULL;
    if (r->stream > 0xFFFFu) return BNN_E_RANGE;
    if ((uint64_t)r->sample0 + (uint64_t)nsamples > 0x10000ull) return BNN_E_RANGE;
    if (r->generator > BNN_GEN_PHILOX7_U16) return BNN_E_RANGE;
    return BNN_OK;
}

// ----- Philox4x32-10 --------------------------------------------------------------
// A uniform value parked in a VGPR.  On gfx950 a VALU instruction that reads an SGPR issues ~1.35-1.5 x slower than
// the all-VGPR form (tools/ubench_valu.hip: v_xor_b32 v,s,v 1.35; v_bitop3_b32 v,v,s 1.49 against 0.97 for v,v,v), so
// the 20 round keys of a hot draw loop live in VGPRs (PhiloxKeys below).
__device__ __forceinline__ uint32_t uniform_vgpr(uint32_t s)
{
    uint32_t v;
    asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}

// The 20 round keys of one (seed) as VGPR values; built once per kernel (philox_keys) so the moves are not paid per draw.
struct PhiloxKeys { uint32_t k0[10], k1[10]; };

__device__ __forceinline__ PhiloxKeys philox_keys(uint32_t k0, uint32_t k1)
{
    PhiloxKeys k;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        k.k0[r] = uniform_vgpr(k0 + (uint32_t)r * 0x9E3779B9u);
        k.k1[r] = uniform_vgpr(k1 + (uint32_t)r * 0xBB67AE85u);
    }
    return k;
}

// three-input xor = ONE v_bitop3_b32 (truth table 0x96, gfx950) instead of the two v_xor_b32 hipcc emits
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, const PhiloxKeys &k)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        uint4 n;
        n.x = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c.y, k.k0[r], 0x96);
        n.y = (uint32_t)p1;
        n.z = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c.w, k.k1[r], 0x96);
        n.w = (uint32_t)p0;
        c = n;
    }
    return c;
}

// R rounds of the same function (R = 7: Philox4x32-7, the 7-round member Random123 ships known answers for)
template <int R>
__device__ __forceinline__ uint4 philox4x32_r(uint4 c, const PhiloxKeys &k)
{
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        uint4 n;
        n.x = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c.y, k.k0[r], 0x96);
        n.y = (uint32_t)p1;
        n.z = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c.w, k.k1[r], 0x96);
        n.w = (uint32_t)p0;
        c = n;
    }
    return c;
}

// Same function with the keys left in SGPRs (kernels that draw a handful of blocks: bias, tails).
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        uint4 n;
        n.x = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c.y, k0, 0x96);
        n.y = (uint32_t)p1;
        n.z = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c.w, k1, 0x96);
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

template <int R>
__device__ __forceinline__ uint4 philox4x32_r(uint4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        uint4 n;
        n.x = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c.y, k0, 0x96);
        n.y = (uint32_t)p1;
        n.z = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c.w, k1, 0x96);
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

__device__ __forceinline__ uint32_t rng_epoch_dev(const RngDev &r)
{
    uint32_t e = r.epoch_dev ? __builtin_nontemporal_load(r.epoch_dev) : 0u;
    return e + (uint32_t)r.epoch_dev_delta;
}

// u = ((x >> 8) + 0.5) * 2^-24 in (0, 1), as ONE fma: a 2^-24 + 2^-25 is the same real number rounded once
// (the two-instruction form rounds a + 0.5 and then scales by a power of two, which is exact) -- bit-identical.
__device__ __forceinline__ float u01(uint32_t x)
{
    return __builtin_fmaf((float)(x >> 8), 0x1p-24f, 0x1p-25f);
}

// One Box-Muller pair.  v_sin_f32 / v_cos_f32 take their argument in revolutions,
// so sin(2 pi u) is evaluated on the exact u (no 2*pi rounding).  ln via the
// accurate ocml logf: near u -> 1 the native log2 loses the relative accuracy
// r = sqrt(-2 ln u) needs.
// ln(u) for u in [2^-25, 1): ocml's logf without the branches this range never takes (denormal
// scaling, inf / nan pass-through) -- v_log_f32 and the same hi / lo multiplication by ln 2, so the
// result is bit-identical to logf(u) here, in 5 instructions instead of 13.
__device__ __forceinline__ float ln_unit(float u)
{
    const float l2 = __builtin_amdgcn_logf(u);
    const float c_hi = 0x1.62e42ep-1f, c_lo = 0x1.efa39ep-25f;      // 0x3f317217, 0x3377d1cf
    const float p = l2 * c_hi;
    float e = __builtin_fmaf(l2, c_hi, -p);
    e = __builtin_fmaf(l2, c_lo, e);
    return __builtin_fmaf(c_hi, l2, e);
}

__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb, float &z0, float &z1)
{
    const float ua = u01(xa), ub = u01(xb);
    const float r = __builtin_amdgcn_sqrtf(-2.0f * ln_unit(ua));
    z0 = r * __builtin_amdgcn_cosf(ub);
    z1 = r * __builtin_amdgcn_sinf(ub);
}

// One Box-Muller pair from the two 16-bit halves of ONE random word: u = (h + 0.5) 2^-16 in (0, 1).
__device__ __forceinline__ void box_muller16(uint32_t x, float &z0, float &z1)
{
    const float ua = __builtin_fmaf((float)(x & 0xFFFFu), 0x1p-16f, 0x1p-17f);
    const float ub = __builtin_fmaf((float)(x >> 16), 0x1p-16f, 0x1p-17f);
    const float r = __builtin_amdgcn_sqrtf(-2.0f * ln_unit(ua));
    z0 = r * __builtin_amdgcn_cosf(ub);
    z1 = r * __builtin_amdgcn_sinf(ub);
}

// BNN_GEN_PHILOX7_U16: the EIGHT eps values of block `block8` (elements 8 block8 .. 8 block8 + 7), keys in VGPRs
__device__ __forceinline__ void eps8_u16(const RngDev &r, const PhiloxKeys &keys, uint32_t epoch_dev, uint32_t block8,
                                         uint32_t sample, float4 &za, float4 &zb)
{
    const uint4 x = philox4x32_r<7>(make_uint4(block8, r.stream_hi | (sample & 0xFFFFu), r.epoch_host, epoch_dev), keys);
    box_muller16(x.x, za.x, za.y); box_muller16(x.y, za.z, za.w);
    box_muller16(x.z, zb.x, zb.y); box_muller16(x.w, zb.z, zb.w);
}

// The four eps values of quad `block` (elements 4*block .. 4*block+3) of the stream r.gen names (wave-uniform branch).
// BNN_GEN_PHILOX7_U16: the quad is one half of Philox block `block >> 1` (a caller that walks quads pays one 7-round block per
// quad; the draw launch, where the time goes, takes whole blocks: eps8_u16).
__device__ __forceinline__ float4 eps4(const RngDev &r, uint32_t epoch_dev, uint32_t block,
                                       uint32_t sample)
{
    float4 z;
    if (r.gen == BNN_GEN_PHILOX7_U16) {
        const uint4 x = philox4x32_r<7>(make_uint4(block >> 1, r.stream_hi | (sample & 0xFFFFu), r.epoch_host, epoch_dev), r.key0, r.key1);
        const bool hi = (block & 1u) != 0;
        box_muller16(hi ? x.z : x.x, z.x, z.y);
        box_muller16(hi ? x.w : x.y, z.z, z.w);
        return z;
    }
    const uint4 x = philox4x32_10(make_uint4(block, r.stream_hi | (sample & 0xFFFFu),
                                             r.epoch_host, epoch_dev), r.key0, r.key1);
    box_muller(x.x, x.y, z.x, z.y);
    box_muller(x.z, x.w, z.z, z.w);
    return z;
}

// Same values with the round keys in VGPRs (keys = philox_keys(r.key0, r.key1), once per kernel).
__device__ __forceinline__ float4 eps4(const RngDev &r, const PhiloxKeys &keys, uint32_t epoch_dev, uint32_t block,
                                       uint32_t sample)
{
    float4 z;
    if (r.gen == BNN_GEN_PHILOX7_U16) {
        const uint4 x = philox4x32_r<7>(make_uint4(block >> 1, r.stream_hi | (sample & 0xFFFFu), r.epoch_host, epoch_dev), keys);
        const bool hi = (block & 1u) != 0;
        box_muller16(hi ? x.z : x.x, z.x, z.y);
        box_muller16(hi ? x.w : x.y, z.z, z.w);
        return z;
    }
    const uint4 x = philox4x32_10(make_uint4(block, r.stream_hi | (sample & 0xFFFFu),
                                             r.epoch_host, epoch_dev), keys);
    box_muller(x.x, x.y, z.x, z.y);
    box_muller(x.z, x.w, z.z, z.w);
    return z;
}

__device__ __forceinline__ float eps1(const RngDev &r, uint32_t epoch_dev, uint64_t elem,
                                      uint32_t sample)
{
    const float4 z = eps4(r, epoch_dev, (uint32_t)(elem >> 2), sample);
    const uint32_t j = (uint32_t)elem & 3u;
    return j == 0 ? z.x : j == 1 ? z.y : j == 2 ? z.z : z.w;
}

// sigma = 1e-10 + softplus(rho), torch semantics (beta 1, threshold 20), for the DRAW:
// ln(1 + e^rho) on the raw exp2 / log2 units (v_exp_f32, v_log_f32; 8 instructions).
// Absolute error <= ~1e-7 (the rounding of 1 + e), i.e. <= 1e-7 * |eps| on a drawn weight --
// far inside the 1e-5 parity bar.  Every kernel that draws (K1, K2 loaders, backward)
// uses THIS function, so a draw re-created from its key is bit-identical everywhere.
// (KL takes ln(sigma) and therefore uses sigma_accurate below.)
__device__ __forceinline__ float sigma_draw(float rho)
{
    const float e = __builtin_amdgcn_exp2f(rho * 1.44269504088896341f);
    const float sp = __builtin_fmaf(__builtin_amdgcn_logf(1.0f + e), 0.693147180559945309f, 1e-10f);
    // (the threshold as v_min / v_max instead of v_cmp / v_cndmask was measured: fused GEMM +-0, K1 stream -8 %)
    return rho > 20.0f ? rho : sp;                 // (rho + 1e-10 == rho in fp32 above the threshold)
}

// Same value to ~1e-6 RELATIVE accuracy (used where ln(sigma) is taken: KL, and for
// WeightNormal.stddev): log1p(e) = log(u) * e / (u - 1), u = 1 + e -- the rounding of u cancels
// (and log1p(e) = e when u == 1) -- on the native exp2 / log2 / rcp units.
__device__ __forceinline__ float sigma_accurate(float rho)
{
    const float e = __builtin_amdgcn_exp2f(rho * 1.44269504088896341f);
    const float u = 1.0f + e;
    const float d = u - 1.0f;
    const float l = __builtin_amdgcn_logf(u) * 0.693147180559945309f;
    float sp = l * (e * __builtin_amdgcn_rcpf(d));
    sp = (d == 0.0f) ? e : sp;
    sp = rho > 20.0f ? rho : sp;
    return 1e-10f + sp;
}

// d softplus / d rho (torch: 1 above the threshold).
__device__ __forceinline__ float dsoftplus(float rho)
{
    return rho > 20.0f ? 1.0f : __fdividef(1.0f, 1.0f + __expf(-rho));
}

// fp32 -> bf16 bits, round-to-nearest-even, NaN kept NaN (v_cvt_pk_bf16_f32).
__device__ __forceinline__ uint16_t f2bf(float v)
{
    const __bf16 h = (__bf16)v;
    return __builtin_bit_cast(uint16_t, h);
}

// two fp32 -> packed bf16x2 (lo in bits 15:0): one v_cvt_pk_bf16_f32 (RNE, NaN kept NaN)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(uint32_t, r);
}

// (x, y) -> three packed bf16 pairs h, m, l with x = h.lo + m.lo + l.lo (+ <= 2^-27 |x|), same for y / .hi: every residual
// is exact in fp32 (a 24-bit significand minus its leading 8 bits fits), conversions round to nearest even.
__device__ __forceinline__ void split_bf16x3(float x, float y, uint32_t &h, uint32_t &m, uint32_t &l)
{
    h = pack_bf16x2(x, y);
    const float rx = x - __uint_as_float(h << 16), ry = y - __uint_as_float(h & 0xFFFF0000u);
    m = pack_bf16x2(rx, ry);
    const float qx = rx - __uint_as_float(m << 16), qy = ry - __uint_as_float(m & 0xFFFF0000u);
    l = pack_bf16x2(qx, qy);
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace bnn
