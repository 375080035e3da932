// pg_pk16.h -- packed 16-bit arithmetic of the DP kernels (gfx950 VOP3P): two scores per VGPR, integer scores carried
// as half-precision numbers so that the three-input packed maximum can be used; DPP row shifts.  Shared by the gssw fill
// (pg_fill.hip) and the klib sweeps (pg_klib_packed.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pg_device.h"

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u16x2 asU(uint32_t x) { return __builtin_bit_cast(u16x2, x); }
__device__ __forceinline__ i16x2 asS(uint32_t x) { return __builtin_bit_cast(i16x2, x); }
__device__ __forceinline__ uint32_t asW(u16x2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t asW(i16x2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return asW(asU(a) + asU(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return asW(asU(a) - asU(b)); }
__device__ __forceinline__ uint32_t pk_maxi(uint32_t a, uint32_t b)
{
    return asW(__builtin_elementwise_max(asS(a), asS(b)));
}
__device__ __forceinline__ uint32_t pk_maxu(uint32_t a, uint32_t b)
{
    return asW(__builtin_elementwise_max(asU(a), asU(b)));
}
__device__ __forceinline__ uint32_t pk_minu(uint32_t a, uint32_t b)
{
    return asW(__builtin_elementwise_min(asU(a), asU(b)));
}
__device__ __forceinline__ uint32_t pk_subsat(uint32_t a, uint32_t b)
{
    return asW(__builtin_elementwise_sub_sat(asU(a), asU(b)));
}

// ---- the recurrence's arithmetic: integer scores carried as half-precision numbers -----------------------------------------
// A score n is held as the f16 number 1024 + n, two strands per VGPR.  Every integer below 2048 is exact in f16 and so are
// sums and differences of them, so this is integer arithmetic in disguise -- but gfx950 has a THREE-input packed maximum for
// f16 (v_pk_maximum3_f16) and none for integers: h = max(diag + s, E, F) is 2 instructions instead of 3, E' = max(E - ge,
// h - go, 0) is 2 with the clamp at zero folded in (gssw's saturating _mm_subs_epu8), and the 11-way column maximum is 5
// instead of 10.  7 instead of 8 instructions per cell pair.  Between 1024 and 2048 one ulp is 1, so the bit pattern of
// 1024 + n is 0x6400 | n: the byte / 16-bit value the H trace and the seeds store is simply the low part of the register,
// and comparing bit patterns as unsigned integers compares the scores (everything is positive), which is what the rare
// paths do with the integer v_pk_max_u16.
#define PG_F16_BIAS2 0x64006400u  // (1024.0, 1024.0)
#define PG_F16_NEG_GO2 0xC600C600u  // (-6.0, -6.0)
static_assert(PG_GAP_OPEN == 6 && PG_GAP_EXT == 1, "the f16 constants above encode gap open 6 / extend 1");
__device__ __forceinline__ uint32_t pk_addh(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_pk_add_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_addh_s(uint32_t a, uint32_t sconst)
{  // second operand wave-uniform (an SGPR)
    uint32_t d;
    asm("v_pk_add_f16 %0, %1, %2" : "=v"(d) : "v"(a), "s"(sconst));
    return d;
}
__device__ __forceinline__ uint32_t pk_dech(uint32_t a)
{  // a - gap extend
    uint32_t d;
    asm("v_pk_add_f16 %0, %1, -1.0 op_sel_hi:[1,0]" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ uint32_t pk_max3h(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t pk_maxh(uint32_t a, uint32_t b)
{  // signed f16 maximum (the integer v_pk_max_u16 on bit patterns only orders non-negative numbers)
    uint32_t d;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_max3h_s(uint32_t a, uint32_t b, uint32_t sconst)
{
    uint32_t d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(sconst));
    return d;
}
// ---- additions as 32-bit integer additions on the bit patterns ----------------------------------------------------------------
// gfx950 issues a wave64 v_pk_* (or any VOP3P / v_perm / v_bfi / integer max) instruction over FOUR cycles and v_add_u32,
// v_add_f32, v_fma_f32, v_mov_b32 over TWO (profiles/r04_valu_rate.json).  Between 1024 and 2048 the f16 pattern of 1024 + n is
// 0x6400 + n, so adding a small integer d to such a number IS adding d to its bit pattern -- and both halves of a register take
// their own d in ONE 32-bit addition of the word d_lo + (d_hi << 16) (mod 2^32): the whole-word sum is (hi + d_hi) * 65536 +
// (lo + d_lo), and as long as each half's result stays within 0 .. 65535 no carry or borrow crosses the boundary (a negative
// d_lo's "borrow" is already in the word's two's complement).  Every live value of the recurrence is a score >= -6 in a frame
// >= 8, i.e. a number >= 1024 (that is what PG_TAU0 = 8 is for), so the three additions of a cell pair are exact this way and
// the three maxima read the same patterns as before.  Padding rows (substitution score PG_PAD_SCORE) drop below 0x6400, where
// patterns are no longer linear in the value but still ordered like it: they stay below every real cell, which is all that
// is asked of them.
__device__ __forceinline__ constexpr uint32_t pk_delta(int lo, int hi) { return (uint32_t)lo + ((uint32_t)hi << 16); }
__device__ __forceinline__ constexpr uint32_t pk_delta2(int d) { return pk_delta(d, d); }

// in-place forms for the rare paths: the value stays in its register (a tied operand), so the merge after the branch needs no
// copy on the common path (without them the compiler copied all C registers of the previous column at the top of every step)
__device__ __forceinline__ void pk_maxu_into(uint32_t& acc, uint32_t x) { asm("v_pk_max_u16 %0, %0, %1" : "+v"(acc) : "v"(x)); }
__device__ __forceinline__ void mov_into(uint32_t& dst, uint32_t sconst) { asm("v_mov_b32 %0, %1" : "+v"(dst) : "s"(sconst)); }
__device__ __forceinline__ uint32_t f16_bits(int v) { return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)v); }
// maximum of N packed values with three-input instructions: ceil((N - 1) / 2) of them
template <int N> __device__ __forceinline__ uint32_t pk_max_all(const uint32_t (&v)[N])
{
    if constexpr (N == 1)
        return v[0];
    else if constexpr (N == 2)
        return asW(__builtin_elementwise_max(asU(v[0]), asU(v[1])));
    else if constexpr (N == 3)
        return pk_max3h(v[0], v[1], v[2]);
    else
    {
        // full triples are reduced, a remainder of one or two values is carried to the next level as it is
        constexpr int T = N / 3, R = N % 3, K = T + R;
        uint32_t w[K];
#pragma unroll
        for (int g = 0; g < T; ++g)
            w[g] = pk_max3h(v[3 * g], v[3 * g + 1], v[3 * g + 2]);
#pragma unroll
        for (int g = 0; g < R; ++g)
            w[T + g] = v[3 * T + g];
        return pk_max_all<K>(w);
    }
}

// row_shr:1 within each 16-lane DPP row. bound_ctrl=true: lane 0 of a row receives 0.
__device__ __forceinline__ uint32_t row_shr1_zero(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
}
// lane 0 of each row keeps `first`.
__device__ __forceinline__ uint32_t row_shr1_keep(uint32_t first, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x111, 0xf, 0xf, false);
}
// The same shift for groups of GL lanes (a read = GL lanes): GL = 16 is one DPP row; GL = 32 spans two rows, so lane 16 of a
// group first takes lane 15 of the row before it (row_bcast:15 into the odd rows), then row_shr:1 moves the rest -- lane 0 of
// every row keeps what it holds, which is `first` in the group's first lane and the broadcast value in its 17th.
template <int GL> __device__ __forceinline__ uint32_t group_shr1_keep(uint32_t first, uint32_t v)
{
    if constexpr (GL == 16)
        return row_shr1_keep(first, v);
    else
    {
        static_assert(GL == 32, "a read takes 16 or 32 lanes");
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15, rows 1 and 3
        return (uint32_t)__builtin_amdgcn_update_dpp((int)t, (int)v, 0x111, 0xf, 0xf, false);
    }
}
