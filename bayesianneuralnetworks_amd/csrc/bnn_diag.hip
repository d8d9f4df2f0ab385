// bnn_diag.hip -- memory-path diagnostics (not part of the product path).
// k_diag_astream: every workgroup streams a (rows x K) fp32 activation block out of a
// (S, M, K) tensor exactly like the fused linear kernel would, with different per-lane
// access shapes, and does nothing else.  Tells how many bytes per clock one CU can pull
// from L2 in each shape.
#include "bnn_device.hpp"

namespace bnn {

// PATTERN 0: MFMA-fragment shape (lane (i, q): row i of a 16-row tile, 16 B at k = 16h + 4q)
// PATTERN 1: row-contiguous (a wave-instruction covers 4 rows x 256 B)
// PATTERN 2: LDS-DMA of the row-contiguous shape (global_load_lds_dwordx4, no VGPRs)
template <int PATTERN, int NWAVES, int RPW>
__global__ __launch_bounds__(NWAVES * 64) void k_diag_astream(const float *__restrict__ x, int M, int K,
                                                              int ntm, int ntn, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int rows_per_wave = RPW;
    constexpr int rows_per_wg = RPW * NWAVES;
    const int L = blockIdx.x;
    const int i_in = L >> 3;
    const int per_s = ntn * ntm;
    const int s = (L & 7) + 8 * (i_in / per_s);
    const int mt = (i_in % per_s) % ntm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *Ab = x + ((int64_t)s * M + (int64_t)mt * rows_per_wg) * K;
    float acc = 0.f;
    const int nk = K / 64;          // 64 k (256 B per row) per step
    if constexpr (PATTERN == 0) {
        const int fi = lane & 15, fq = lane >> 4;
        for (int kt = 0; kt < nk; ++kt) {
            float4 r[RPW / 16][4];
#pragma unroll
            for (int a = 0; a < rows_per_wave / 16; ++a) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float *p = Ab + (int64_t)(wave * rows_per_wave + a * 16 + fi) * K + kt * 64 + 16 * j + 4 * fq;
                    r[a][j] = *reinterpret_cast<const float4 *>(p);
                }
            }
#pragma unroll
            for (int a = 0; a < rows_per_wave / 16; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += r[a][j].x + r[a][j].w;
        }
    } else if constexpr (PATTERN == 1) {
        const int lr = lane >> 4, lc = lane & 15;
        for (int kt = 0; kt < nk; ++kt) {
            float4 r[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j * 4 < rows_per_wave) {
                    const float *p = Ab + (int64_t)(wave * rows_per_wave + j * 4 + lr) * K + kt * 64 + 4 * lc;
                    r[j] = *reinterpret_cast<const float4 *>(p);
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j * 4 < rows_per_wave) acc += r[j].x + r[j].w;
        }
    } else {
        const int lr = lane >> 4, lc = lane & 15;
        // each wave owns a private LDS slab of rows_per_wave x 256 B, refilled every step
        char *slab = smem + wave * rows_per_wave * 256;
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j * 4 < rows_per_wave) {
                    const float *p = Ab + (int64_t)(wave * rows_per_wave + j * 4 + lr) * K + kt * 64 + 4 * lc;
                    __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)(slab + j * 1024), 16, 0, 0);
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) (gfx9 encoding: vmcnt low bits 3:0 + 15:14)
            acc += *reinterpret_cast<float *>(slab + lane * 16);
        }
    }
    out[blockIdx.x * (NWAVES * 64) + tid] = acc;
}

}  // namespace bnn

using namespace bnn;

extern "C" int bnn_diag_astream(const float *x, int S, int M, int K, int rows_per_wg, int ntn, int pattern,
                                int nwaves, float *out, void *stream)
{
    if (!x || !out) { set_error("bnn_diag_astream: NULL"); return BNN_E_NULL; }
    const int rpw = rows_per_wg / (nwaves > 0 ? nwaves : 1);
    if (S % 8 || K % 64 || M % rows_per_wg || (rpw != 32 && rpw != 64) || (nwaves != 4 && nwaves != 8 && nwaves != 16)) {
        set_error("bnn_diag_astream: unsupported shape");
        return BNN_E_SHAPE;
    }
    const int ntm = M / rows_per_wg;
    const int grid = S * ntm * ntn;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)rows_per_wg * 256;
#define L3(P, W, R) hipLaunchKernelGGL((k_diag_astream<P, W, R>), dim3(grid), dim3(W * 64), (P == 2 ? lds : 0), st, x, M, K, ntm, ntn, out)
#define L2(P, W) do { if (rpw == 32) L3(P, W, 32); else L3(P, W, 64); } while (0)
#define L1(W) do { if (pattern == 0) L2(0, W); else if (pattern == 1) L2(1, W); else L2(2, W); } while (0)
    if (nwaves == 4) L1(4); else if (nwaves == 8) L1(8); else L1(16);
#undef L1
#undef L2
#undef L3
    return check_launch("bnn_diag_astream");
}
