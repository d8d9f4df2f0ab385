// bnn_dma.hpp -- LDS-DMA (global_load_lds_dwordx4) helpers shared by the contraction kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bnn {

// One LDS-DMA piece: 64 lanes x 16 B land at LDS byte address lds_addr + 16 * lane (lds_addr
// wave-uniform, in an SGPR -> M0); every lane supplies its own global source address.
// Inline asm ON PURPOSE: with __builtin_amdgcn_global_load_lds hipcc knows that LDS is written
// asynchronously and puts s_waitcnt vmcnt(0) in front of every ds_read of the kernel (measured:
// the DMA ring drained every k-step).  Here the counted s_waitcnt vmcnt(N) below is the only
// wait.  M0 is saved / restored inside the statement (it is compiler-reserved).
__device__ __forceinline__ void dma16(const float *src, uint32_t lds_addr)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_addr)
                 : "memory");
}
// PW pieces with ONE M0 write: the instruction offset is added to BOTH the LDS address and the
// global address, so piece j uses offset 1024 * j and a source pointer moved back by 1024 * j bytes.
template <int PW>
__device__ __forceinline__ void dma16xN(const char *const (&src)[PW], int byte_ofs, uint32_t lds_addr)
{
    uint32_t keep;
    if constexpr (PW == 1) {
        const char *p0 = src[0] + byte_ofs;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(p0), "s"(lds_addr) : "memory");
    } else if constexpr (PW == 2) {
        const char *p0 = src[0] + byte_ofs, *p1 = src[1] + byte_ofs - 1024;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(p0), "v"(p1), "s"(lds_addr) : "memory");
    } else {
        static_assert(PW == 4, "PW");
        const char *p0 = src[0] + byte_ofs, *p1 = src[1] + byte_ofs - 1024, *p2 = src[2] + byte_ofs - 2048,
                   *p3 = src[3] + byte_ofs - 3072;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %3, off offset:2048\n\tglobal_load_lds_dwordx4 %4, off offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(lds_addr) : "memory");
    }
}

// Three pieces, scalar-base form: piece i comes from (base_i + voff_i) -- a 64-bit wave-uniform base in
// SGPRs plus a 32-bit per-lane byte offset -- and lands at lds_i + 16 * lane.  No VALU address math per
// issue: the bases advance with scalar adds, the lane offsets are fixed for the life of the kernel.
__device__ __forceinline__ void dma16_s3(const void *base0, uint32_t voff0, uint32_t lds0,
                                         const void *base1, uint32_t voff1, uint32_t lds1,
                                         const void *base2, uint32_t voff2, uint32_t lds2)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
                 "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %8\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff0), "s"(base0), "s"(lds0), "v"(voff1), "s"(base1), "s"(lds1), "v"(voff2), "s"(base2), "s"(lds2)
                 : "memory");
}

__device__ __forceinline__ uint32_t lds_addr_of(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}

}  // namespace bnn
