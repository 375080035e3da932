// valu_rate.hip -- what a wave64 VALU instruction costs on gfx950, measured: issue rate and dependent-issue latency of the
// instructions the DP kernels are made of, at 1 / 2 / 4 / 8 wavefronts per SIMD, with 1 (every instruction waits for the one
// before it) to 8 independent chains per wavefront; the fill kernel's own 6-instruction recurrence (10 rows, its real
// dependency structure) three ways -- as the compiler emits it from one statement per instruction (it puts an `s_nop 0`
// between a packed instruction and a VALU instruction that reads its result: the gfx940 "dst_sel forwarding" hazard, which it
// assumes for every VOP3P instruction and every inline-asm statement), hand-scheduled in one block WITHOUT those nops, and in
// one block WITH them -- with the results of the three compared bit for bit (is the hazard real for full-dword packed
// writes?); and the clock the chip holds while it does so (s_memtime ticks against the 100 MHz s_memrealtime).
//
// Prints one JSON document on stdout (-> profiles/rNN_valu_rate.json); bench.py reads the cycles per instruction and the clock
// from it for roofline.valu instead of assuming them.
//
// Placement is forced, not hoped for: a workgroup asks for so much LDS that exactly ONE (or two, for 8 wavefronts per SIMD)
// fits a CU, and the grid is one workgroup per CU slot.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate tools/ubench/valu_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                     \
    do                                                                                            \
    {                                                                                             \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess)                                                                     \
        {                                                                                         \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));     \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

struct Times
{
    unsigned long long t0, t1, r0, r1;
};
__device__ __forceinline__ unsigned long long memtime()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned long long memrealtime()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// ---- single instructions: 64 of them in ONE asm block (the compiler cannot put anything between them) ---------------------------
// D = the chain's register (read and written), %[y] / %[z] loop-invariant VGPRs
#define R2(X) X X
#define R4(X) R2(R2(X))
#define R8(X) R2(R4(X))
#define R16(X) R2(R8(X))
#define R32(X) R2(R16(X))
#define R64(X) R2(R32(X))
#define BODY1(T) R64(T("%[x0]"))
#define BODY2(T) R32(T("%[x0]") T("%[x1]"))
#define BODY4(T) R16(T("%[x0]") T("%[x1]") T("%[x2]") T("%[x3]"))
#define BODY8(T) R8(T("%[x0]") T("%[x1]") T("%[x2]") T("%[x3]") T("%[x4]") T("%[x5]") T("%[x6]") T("%[x7]"))

#define T_pk_maximum3_f16(D) "v_pk_maximum3_f16 " D ", " D ", %[y], %[z]\n\t"
#define T_pk_add_f16(D) "v_pk_add_f16 " D ", " D ", %[y]\n\t"
#define T_pk_max_f16(D) "v_pk_max_f16 " D ", " D ", %[y]\n\t"
#define T_pk_max_u16(D) "v_pk_max_u16 " D ", " D ", %[y]\n\t"
#define T_pk_add_u16(D) "v_pk_add_u16 " D ", " D ", %[y]\n\t"
#define T_pk_fma_f16(D) "v_pk_fma_f16 " D ", " D ", %[y], %[z]\n\t"
#define T_perm_b32(D) "v_perm_b32 " D ", " D ", %[y], %[z]\n\t"
#define T_bfi_b32(D) "v_bfi_b32 " D ", %[y], " D ", %[z]\n\t"
#define T_add_u32(D) "v_add_u32 " D ", " D ", %[y]\n\t"
#define T_max_u32(D) "v_max_u32 " D ", " D ", %[y]\n\t"
#define T_max3_u32(D) "v_max3_u32 " D ", " D ", %[y], %[z]\n\t"
#define T_add_f32(D) "v_add_f32 " D ", " D ", %[y]\n\t"
#define T_fma_f32(D) "v_fma_f32 " D ", " D ", %[y], %[z]\n\t"
#define T_mov_b32(D) "v_mov_b32 " D ", %[y]\n\t"
// DPP: the source is a loop-invariant register (a DPP read of a VGPR a VALU instruction has just written needs two wait
// states -- a documented hazard -- so the dependent form cannot be timed without nops; this is the issue rate)
#define T_mov_dpp_row_shr1(D) "v_mov_b32_dpp " D ", %[y] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define T_add_u32_dpp_row_shr1(D) "v_add_u32_dpp " D ", %[y], " D " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
// the same dependent chain with the nop the compiler would put between the instructions
#define T_pk_add_f16_nop(D) "v_pk_add_f16 " D ", " D ", %[y]\n\ts_nop 0\n\t"
#define T_pk_maximum3_f16_nop(D) "v_pk_maximum3_f16 " D ", " D ", %[y], %[z]\n\ts_nop 0\n\t"

// ---- classification: which opcodes issue over two cycles, which over four (measured at 4 / 8 wavefronts per SIMD, 8 chains) --
#define T_sub_u32(D) "v_sub_u32 " D ", " D ", %[y]\n\t"
#define T_add_u32_literal(D) "v_add_u32 " D ", 0xfffafffb, " D "\n\t"
#define T_add_u32_sgpr(D) "v_add_u32 " D ", %[sy], " D "\n\t"
#define T_and_b32(D) "v_and_b32 " D ", " D ", %[y]\n\t"
#define T_or_b32(D) "v_or_b32 " D ", " D ", %[y]\n\t"
#define T_xor_b32(D) "v_xor_b32 " D ", " D ", %[y]\n\t"
#define T_lshlrev_b32(D) "v_lshlrev_b32 " D ", 1, " D "\n\t"
#define T_lshrrev_b32(D) "v_lshrrev_b32 " D ", 1, " D "\n\t"
#define T_cndmask_b32(D) "v_cndmask_b32 " D ", " D ", %[y], vcc\n\t"
#define T_cmp_lt_u32(D) "v_cmp_lt_u32 vcc, " D ", %[y]\n\t"
#define T_max_f32(D) "v_max_f32 " D ", " D ", %[y]\n\t"
#define T_max_i32(D) "v_max_i32 " D ", " D ", %[y]\n\t"
#define T_min_u32(D) "v_min_u32 " D ", " D ", %[y]\n\t"
#define T_mul_f32(D) "v_mul_f32 " D ", " D ", %[y]\n\t"
#define T_sub_f32(D) "v_sub_f32 " D ", " D ", %[y]\n\t"
#define T_add_f16(D) "v_add_f16 " D ", " D ", %[y]\n\t"
#define T_max_f16(D) "v_max_f16 " D ", " D ", %[y]\n\t"
#define T_add_u16(D) "v_add_u16 " D ", " D ", %[y]\n\t"
#define T_max_u16(D) "v_max_u16 " D ", " D ", %[y]\n\t"
#define T_add3_u32(D) "v_add3_u32 " D ", " D ", %[y], %[z]\n\t"
#define T_lshl_add_u32(D) "v_lshl_add_u32 " D ", " D ", 1, %[y]\n\t"
#define T_and_or_b32(D) "v_and_or_b32 " D ", " D ", %[y], %[z]\n\t"
#define T_lshl_or_b32(D) "v_lshl_or_b32 " D ", " D ", 1, %[y]\n\t"
#define T_alignbit_b32(D) "v_alignbit_b32 " D ", " D ", %[y], 8\n\t"
#define T_bfe_u32(D) "v_bfe_u32 " D ", " D ", 1, 30\n\t"
#define T_max3_f32(D) "v_max3_f32 " D ", " D ", %[y], %[z]\n\t"
#define T_maximum3_f32(D) "v_maximum3_f32 " D ", " D ", %[y], %[z]\n\t"
#define T_med3_i32(D) "v_med3_i32 " D ", " D ", %[y], %[z]\n\t"
#define T_mad_u32_u24(D) "v_mad_u32_u24 " D ", " D ", %[y], %[z]\n\t"
#define T_mul_lo_u32(D) "v_mul_lo_u32 " D ", " D ", %[y]\n\t"
#define T_pk_sub_u16(D) "v_pk_sub_u16 " D ", " D ", %[y]\n\t"
#define T_pk_ashrrev_i16(D) "v_pk_ashrrev_i16 " D ", 1, " D " op_sel_hi:[0,1]\n\t"
#define T_pk_mul_f16(D) "v_pk_mul_f16 " D ", " D ", %[y]\n\t"
#define T_mov_b32_sgpr(D) "v_mov_b32 " D ", %[sy]\n\t"
#define T_mov_b32_e64(D) "v_mov_b32_e64 " D ", %[y]\n\t"
#define T_add_u32_e64(D) "v_add_u32_e64 " D ", " D ", %[y]\n\t"
#define T_readfirstlane_mov(D) "v_readfirstlane_b32 s90, " D "\n\t"
#define T_sad_u8(D) "v_sad_u8 " D ", " D ", %[y], %[z]\n\t"
#define T_cvt_f32_u32(D) "v_cvt_f32_u32 " D ", " D "\n\t"
#define T_ds_bpermute(D) "ds_bpermute_b32 " D ", %[y], " D "\n\ts_waitcnt lgkmcnt(0)\n\t"

#define DEF_OP(NAME)                                                                                                            \
    template <int CH> struct Op_##NAME                                                                                          \
    {                                                                                                                           \
        static constexpr const char* name = #NAME;                                                                              \
        static __device__ __forceinline__ void f(uint32_t (&x)[8], uint32_t y, uint32_t z, uint32_t sy = 3u)                                \
        {                                                                                                                       \
            if constexpr (CH == 1)                                                                                              \
                asm volatile(BODY1(T_##NAME) : [x0] "+v"(x[0]) : [y] "v"(y), [z] "v"(z), [sy] "s"(sy) : "vcc", "s90");           \
            else if constexpr (CH == 2)                                                                                         \
                asm volatile(BODY2(T_##NAME) : [x0] "+v"(x[0]), [x1] "+v"(x[1]) : [y] "v"(y), [z] "v"(z), [sy] "s"(sy) : "vcc", "s90"); \
            else if constexpr (CH == 4)                                                                                         \
                asm volatile(BODY4(T_##NAME)                                                                                    \
                             : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3])                               \
                             : [y] "v"(y), [z] "v"(z), [sy] "s"(sy) : "vcc", "s90");                                             \
            else                                                                                                                \
                asm volatile(BODY8(T_##NAME)                                                                                    \
                             : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]),             \
                               [x5] "+v"(x[5]), [x6] "+v"(x[6]), [x7] "+v"(x[7])                                                \
                             : [y] "v"(y), [z] "v"(z), [sy] "s"(sy) : "vcc", "s90");                                             \
        }                                                                                                                       \
    };
DEF_OP(pk_maximum3_f16)
DEF_OP(pk_add_f16)
DEF_OP(pk_max_f16)
DEF_OP(pk_max_u16)
DEF_OP(pk_add_u16)
DEF_OP(pk_fma_f16)
DEF_OP(perm_b32)
DEF_OP(bfi_b32)
DEF_OP(add_u32)
DEF_OP(max_u32)
DEF_OP(max3_u32)
DEF_OP(add_f32)
DEF_OP(fma_f32)
DEF_OP(mov_b32)
DEF_OP(mov_dpp_row_shr1)
DEF_OP(add_u32_dpp_row_shr1)
DEF_OP(pk_add_f16_nop)
DEF_OP(pk_maximum3_f16_nop)
DEF_OP(sub_u32)
DEF_OP(add_u32_literal)
DEF_OP(add_u32_sgpr)
DEF_OP(and_b32)
DEF_OP(or_b32)
DEF_OP(xor_b32)
DEF_OP(lshlrev_b32)
DEF_OP(lshrrev_b32)
DEF_OP(cndmask_b32)
DEF_OP(cmp_lt_u32)
DEF_OP(max_f32)
DEF_OP(max_i32)
DEF_OP(min_u32)
DEF_OP(mul_f32)
DEF_OP(sub_f32)
DEF_OP(add_f16)
DEF_OP(max_f16)
DEF_OP(add_u16)
DEF_OP(max_u16)
DEF_OP(add3_u32)
DEF_OP(lshl_add_u32)
DEF_OP(and_or_b32)
DEF_OP(lshl_or_b32)
DEF_OP(alignbit_b32)
DEF_OP(bfe_u32)
DEF_OP(max3_f32)
DEF_OP(maximum3_f32)
DEF_OP(med3_i32)
DEF_OP(mad_u32_u24)
DEF_OP(mul_lo_u32)
DEF_OP(pk_sub_u16)
DEF_OP(pk_ashrrev_i16)
DEF_OP(pk_mul_f16)
DEF_OP(mov_b32_sgpr)
DEF_OP(mov_b32_e64)
DEF_OP(add_u32_e64)
DEF_OP(readfirstlane_mov)
DEF_OP(sad_u8)
DEF_OP(cvt_f32_u32)
DEF_OP(ds_bpermute)

template <typename O> __global__ void k_op(uint32_t* out, const uint32_t* in, int iters, Times* tm)
{
    extern __shared__ uint32_t lds[];
    uint32_t x[8];
    for (int i = 0; i < 8; ++i)
        x[i] = in[(threadIdx.x * 8 + i) & 2047];
    const uint32_t y = in[2048 + (threadIdx.x & 255)], z = in[2048 + ((threadIdx.x + 7) & 255)];
    if (threadIdx.x == 0xFFFFFF)
        lds[0] = 1;
    __syncthreads();
    const unsigned long long t0 = memtime(), r0 = memrealtime();
    for (int it = 0; it < iters; ++it)
        O::f(x, y, z);
    const unsigned long long t1 = memtime(), r1 = memrealtime();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i)
        s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0)
    {
        Times t = { t0, t1, r0, r1 };
        tm[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t;
    }
}

// ---- the fill kernel's column (pg_fill.hip column(), without trace / maximum / bookkeeping) ------------------------------------
// C = 10 rows x (f = F - 1; a = diag + s; h = max3(a, E, f); t = h - 5; E = max3(E, t, floor); F = max(f, t)): the F chain runs
// down the rows (4 dependent instructions per row), the other two hang off it.  Two steps per iteration (H ping-pong).
// MODE 0: one asm statement per instruction -- what pg_fill.hip is written as; the compiler schedules and inserts s_nop 0
// MODE 1: one block per step, program order, no nops            MODE 2: the same block with s_nop 0 where MODE 0 has them
// MODES 0-2 hold the additions as v_pk_add_f16 (round 3's form: every instruction of the packed class); MODES 3-4 as v_add_u32
#define ROW_TXT(HP, HO, E, S, NOP)                                \
    "v_pk_add_f16 %[F], %[F], -1.0 op_sel_hi:[1,0]\n\t"          \
    "v_pk_add_f16 %[a], " HP ", " S "\n\t" NOP                    \
    "v_pk_maximum3_f16 " HO ", %[a], " E ", %[F]\n\t" NOP         \
    "v_pk_add_f16 %[t], " HO ", %[n5]\n\t" NOP                    \
    "v_pk_maximum3_f16 " E ", " E ", %[t], %[fl]\n\t"             \
    "v_pk_max_u16 %[F], %[F], %[t]\n\t" NOP
#define STEP_TXT(NOP)                                                                                                   \
    ROW_TXT("%[d]", "%[o0]", "%[e0]", "%[s0]", NOP) ROW_TXT("%[i0]", "%[o1]", "%[e1]", "%[s1]", NOP)                     \
    ROW_TXT("%[i1]", "%[o2]", "%[e2]", "%[s2]", NOP) ROW_TXT("%[i2]", "%[o3]", "%[e3]", "%[s3]", NOP)                   \
    ROW_TXT("%[i3]", "%[o4]", "%[e4]", "%[s4]", NOP) ROW_TXT("%[i4]", "%[o5]", "%[e5]", "%[s5]", NOP)                   \
    ROW_TXT("%[i5]", "%[o6]", "%[e6]", "%[s6]", NOP) ROW_TXT("%[i6]", "%[o7]", "%[e7]", "%[s7]", NOP)                   \
    ROW_TXT("%[i7]", "%[o8]", "%[e8]", "%[s8]", NOP) ROW_TXT("%[i8]", "%[o9]", "%[e9]", "%[s9]", NOP)
#define STEP_OPERANDS(HI, HO)                                                                                                              \
    : [o0] "=&v"(HO[0]), [o1] "=&v"(HO[1]), [o2] "=&v"(HO[2]), [o3] "=&v"(HO[3]), [o4] "=&v"(HO[4]), [o5] "=&v"(HO[5]), [o6] "=&v"(HO[6]),  \
      [o7] "=&v"(HO[7]), [o8] "=&v"(HO[8]), [o9] "=&v"(HO[9]), [e0] "+v"(E[0]), [e1] "+v"(E[1]), [e2] "+v"(E[2]), [e3] "+v"(E[3]),          \
      [e4] "+v"(E[4]), [e5] "+v"(E[5]), [e6] "+v"(E[6]), [e7] "+v"(E[7]), [e8] "+v"(E[8]), [e9] "+v"(E[9]), [F] "+v"(F), [a] "=&v"(ta),     \
      [t] "=&v"(tt)                                                                                                                        \
    : [d] "v"(dH), [i0] "v"(HI[0]), [i1] "v"(HI[1]), [i2] "v"(HI[2]), [i3] "v"(HI[3]), [i4] "v"(HI[4]), [i5] "v"(HI[5]), [i6] "v"(HI[6]),   \
      [i7] "v"(HI[7]), [i8] "v"(HI[8]), [s0] "v"(S[0]), [s1] "v"(S[1]), [s2] "v"(S[2]), [s3] "v"(S[3]), [s4] "v"(S[4]), [s5] "v"(S[5]),     \
      [s6] "v"(S[6]), [s7] "v"(S[7]), [s8] "v"(S[8]), [s9] "v"(S[9]), [n5] "s"(NEG5), [fl] "s"(FLOOR)

template <int MODE> __global__ void k_recurrence(uint32_t* out, const uint32_t* in, int iters, Times* tm, uint32_t* state)
{
    constexpr int C = 10;
    extern __shared__ uint32_t lds[];
    uint32_t HA[C], HB[C], E[C], S[C];
    for (int r = 0; r < C; ++r)
    {
        HA[r] = 0x64006400u + (in[(threadIdx.x + r) & 2047] & 0x000F000Fu);
        HB[r] = HA[r];
        E[r] = 0x64006400u + (in[(threadIdx.x * 5 + r) & 2047] & 0x00030003u);
        S[r] = (in[(threadIdx.x * 3 + r + blockIdx.x) & 2047] & 1u) ? 0x3C003C00u : 0xC400C400u;  // +1 / -4
    }
    if (threadIdx.x == 0xFFFFFF)
        lds[0] = 1;
    __syncthreads();
    uint32_t F = 0x64006400u, dH = 0x64006400u, ta, tt;
    const uint32_t NEG5 = 0xC500C500u, FLOOR = 0x64006400u;
    const unsigned long long t0 = memtime(), r0 = memrealtime();
    for (int it = 0; it < iters; ++it)
    {
        if constexpr (MODE == 0)
        {
            auto col = [&](uint32_t (&HI)[C], uint32_t (&HO)[C]) __attribute__((always_inline)) {
                uint32_t diag = dH;
#pragma unroll
                for (int r = 0; r < C; ++r)
                {
                    uint32_t a, t;
                    asm volatile("v_pk_add_f16 %0, %0, -1.0 op_sel_hi:[1,0]" : "+v"(F));
                    asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(a) : "v"(diag), "v"(S[r]));
                    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(HO[r]) : "v"(a), "v"(E[r]), "v"(F));
                    diag = HI[r];
                    asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(t) : "v"(HO[r]), "s"(NEG5));
                    asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(E[r]) : "v"(t), "s"(FLOOR));
                    asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(F) : "v"(t));
                }
            };
            col(HA, HB);
            dH = HB[C - 1];
            col(HB, HA);
            dH = HA[C - 1];
        }
        else if constexpr (MODE == 3 || MODE == 4)
        {
            // the recurrence as pg_fill.hip has it since round 4: the three additions as 32-bit integer additions on the bit
            // patterns (two-cycle class).  MODE 3: program order, the compiler schedules; MODE 4: the ten diagonal additions
            // (independent of the F chain) first, as one run of two-cycle instructions
            auto col = [&](uint32_t (&HI)[C], uint32_t (&HO)[C]) __attribute__((always_inline)) {
                uint32_t diag = dH;
                uint32_t a[C];
                if constexpr (MODE == 4)
                {
#pragma unroll
                    for (int r = 0; r < C; ++r)
                    {
                        asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[r]) : "v"(diag), "v"(S[r]));
                        diag = HI[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < C; ++r)
                {
                    uint32_t t;
                    asm volatile("v_add_u32 %0, 0xfffeffff, %0" : "+v"(F));
                    if constexpr (MODE == 3)
                    {
                        asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[r]) : "v"(diag), "v"(S[r]));
                        diag = HI[r];
                    }
                    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(HO[r]) : "v"(a[r]), "v"(E[r]), "v"(F));
                    asm volatile("v_add_u32 %0, 0xfffafffb, %1" : "=v"(t) : "v"(HO[r]));
                    asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(E[r]) : "v"(t), "s"(FLOOR));
                    asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(F) : "v"(t));
                }
            };
            col(HA, HB);
            dH = HB[C - 1];
            col(HB, HA);
            dH = HA[C - 1];
        }
        else if constexpr (MODE == 5 || MODE == 6 || MODE == 7)
        {
            // Does wave priority steer the dual issue?  The SIMD issues two VALU instructions in one slot only when both are of the
            // two-cycle class and come from different wavefronts; the arbiter goes by priority, then age.  MODE 5: every addition
            // runs at priority 1 (s_setprio 1 before it, 0 after); MODE 6: the ten diagonal additions first as ONE run at
            // priority 1, the chain's additions each at priority 1; MODE 7: MODE 6's order without any s_setprio on the chain
            // (only the run of ten is raised)
            auto col = [&](uint32_t (&HI)[C], uint32_t (&HO)[C]) __attribute__((always_inline)) {
                uint32_t diag = dH;
                uint32_t a[C];
                if constexpr (MODE != 5)
                {
                    asm volatile("s_setprio 1");
#pragma unroll
                    for (int r = 0; r < C; ++r)
                    {
                        asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[r]) : "v"(diag), "v"(S[r]));
                        diag = HI[r];
                    }
                    asm volatile("s_setprio 0");
                }
#pragma unroll
                for (int r = 0; r < C; ++r)
                {
                    uint32_t t;
                    if constexpr (MODE == 7)
                        asm volatile("v_add_u32 %0, 0xfffeffff, %0" : "+v"(F));
                    else if constexpr (MODE == 6)
                        asm volatile("s_setprio 1\n\tv_add_u32 %0, 0xfffeffff, %0\n\ts_setprio 0" : "+v"(F));
                    else
                    {
                        asm volatile("s_setprio 1\n\tv_add_u32 %0, 0xfffeffff, %0" : "+v"(F));
                        asm volatile("v_add_u32 %0, %1, %2\n\ts_setprio 0" : "=v"(a[r]) : "v"(diag), "v"(S[r]));
                        diag = HI[r];
                    }
                    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(HO[r]) : "v"(a[r]), "v"(E[r]), "v"(F));
                    if constexpr (MODE == 7)
                        asm volatile("v_add_u32 %0, 0xfffafffb, %1" : "=v"(t) : "v"(HO[r]));
                    else
                        asm volatile("s_setprio 1\n\tv_add_u32 %0, 0xfffafffb, %1\n\ts_setprio 0" : "=v"(t) : "v"(HO[r]));
                    asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(E[r]) : "v"(t), "s"(FLOOR));
                    asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(F) : "v"(t));
                }
            };
            col(HA, HB);
            dH = HB[C - 1];
            col(HB, HA);
            dH = HA[C - 1];
        }
        else if constexpr (MODE == 1)
        {
            asm volatile(STEP_TXT("") STEP_OPERANDS(HA, HB));
            dH = HB[C - 1];
            asm volatile(STEP_TXT("") STEP_OPERANDS(HB, HA));
            dH = HA[C - 1];
        }
        else
        {
            asm volatile(STEP_TXT("s_nop 0\n\t") STEP_OPERANDS(HA, HB));
            dH = HB[C - 1];
            asm volatile(STEP_TXT("s_nop 0\n\t") STEP_OPERANDS(HB, HA));
            dH = HA[C - 1];
        }
        // keep the numbers in f16's exact range whatever the inputs: a renormalisation every iteration would be part of the
        // timed loop, so the scores are simply allowed to saturate the same way in all three modes (same instructions, same
        // order of evaluation per value => bit-identical results unless a hazard bites)
    }
    const unsigned long long t1 = memtime(), r1 = memrealtime();
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = F;
    for (int r = 0; r < C; ++r)
    {
        s ^= HA[r] * 3u ^ E[r] * 5u ^ HB[r] * 7u;
        if (state)
        {
            state[gid * 32 + r] = HA[r];
            state[gid * 32 + 10 + r] = E[r];
            state[gid * 32 + 20 + r] = HB[r];
        }
    }
    if (state)
        state[gid * 32 + 30] = F;
    out[gid] = s;
    if ((threadIdx.x & 63) == 0)
    {
        Times t = { t0, t1, r0, r1 };
        tm[gid / 64] = t;
    }
}

struct Place
{
    int waves_per_simd, threads, lds, blocks_per_cu;
};
static const Place PLACES[] = { { 1, 256, 96 * 1024, 1 }, { 2, 512, 96 * 1024, 1 }, { 4, 1024, 96 * 1024, 1 }, { 8, 1024, 64 * 1024, 2 } };

static int g_cus = 256;
static uint32_t *g_out, *g_in, *g_state;
static Times* g_tm;
static bool g_first = true;

template <typename L>
static void measure(L launch, const void* fn, const char* name, int chains, double inst_per_iter, int iters, const Place& p)
{
    const int blocks = g_cus * p.blocks_per_cu;
    CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, p.lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch(blocks, p);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep)
    {
        CK(hipEventRecord(a));
        launch(blocks, p);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    const int n_waves = blocks * p.threads / 64;
    std::vector<Times> tm(n_waves);
    CK(hipMemcpy(tm.data(), g_tm, sizeof(Times) * n_waves, hipMemcpyDeviceToHost));
    std::vector<double> dt(n_waves), ratio(n_waves);
    for (int w = 0; w < n_waves; ++w)
    {
        dt[w] = (double)(tm[w].t1 - tm[w].t0);
        const double dr = (double)(tm[w].r1 - tm[w].r0);
        ratio[w] = dr > 0 ? dt[w] / dr : 0;
    }
    std::sort(dt.begin(), dt.end());
    std::sort(ratio.begin(), ratio.end());
    const double inst_per_wave = inst_per_iter * iters;
    const double per_simd = inst_per_wave * p.waves_per_simd;  // wave-instructions one SIMD issues
    printf("%s\n  {\"op\": \"%s\", \"waves_per_simd\": %d, \"chains\": %d, \"ms\": %.4f, \"ns_per_wave_inst_per_simd\": %.5f, "
           "\"memtime_ticks_per_wave_inst_median\": %.4f, \"memtime_per_memrealtime_median\": %.4f}",
           g_first ? "" : ",", name, p.waves_per_simd, chains, best, best * 1e6 / per_simd, dt[n_waves / 2] / inst_per_wave,
           ratio[n_waves / 2]);
    g_first = false;
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
}

template <template <int> class O> static void run_op(int iters)
{
    for (const Place& p : PLACES)
    {
#define ONE(CH)                                                                                                                  \
    measure([&](int blocks, const Place& q) { hipLaunchKernelGGL((k_op<O<CH>>), dim3(blocks), dim3(q.threads), q.lds, 0, g_out, g_in, iters, g_tm); }, \
            (const void*)k_op<O<CH>>, O<CH>::name, CH, 64, iters, p)
        ONE(1);
        ONE(2);
        ONE(4);
        ONE(8);
#undef ONE
    }
}


template <template <int> class O> static void classify(int iters)
{
    for (const Place& p : PLACES)
    {
        if (p.waves_per_simd < 4)
            continue;
        measure([&](int blocks, const Place& q) { hipLaunchKernelGGL((k_op<O<8>>), dim3(blocks), dim3(q.threads), q.lds, 0, g_out, g_in, iters, g_tm); },
                (const void*)k_op<O<8>>, O<8>::name, 8, 64, iters, p);
    }
}

template <int MODE> static void run_rec(const char* name, int iters)
{
    for (const Place& p : PLACES)
        measure([&](int blocks, const Place& q) { hipLaunchKernelGGL((k_recurrence<MODE>), dim3(blocks), dim3(q.threads), q.lds, 0, g_out, g_in, iters, g_tm, (uint32_t*)nullptr); },
                (const void*)k_recurrence<MODE>, name, 0, 120, iters, p);
}

// the three forms of the recurrence on the same inputs: final H / E / F of every lane, bit for bit
template <int MODE> static std::vector<uint32_t> rec_state(int iters)
{
    const Place& p = PLACES[2];
    const int blocks = g_cus;
    CK(hipFuncSetAttribute((const void*)k_recurrence<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, p.lds));
    const size_t n = (size_t)blocks * p.threads * 32;
    CK(hipMemset(g_state, 0, n * 4));
    hipLaunchKernelGGL((k_recurrence<MODE>), dim3(blocks), dim3(p.threads), p.lds, 0, g_out, g_in, iters, g_tm, g_state);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(n);
    CK(hipMemcpy(h.data(), g_state, n * 4, hipMemcpyDeviceToHost));
    return h;
}

// `valu_rate occupancy [iters]`: the fill's recurrence (as pg_fill.hip has it: integer additions, program order) at EVERY
// occupancy from 1 to 8 wavefronts per SIMD.  A workgroup is 256 threads = one wavefront per SIMD, and asks for so much LDS that
// exactly N of them fit a CU (160 KB): N wavefronts per SIMD, evenly.  What a fifth / sixth wavefront would buy the fill's
// arithmetic if registers and LDS let it in (-> profiles/rNN_fill_recurrence_occupancy.json).
static void occupancy_sweep(int iters)
{
    printf("{\"what\": \"fill recurrence, C = 10 rows x 2 strands, 60 instructions per column (30 v_add_u32, 20 v_pk_maximum3_f16, 10 v_pk_max_u16), "
           "N workgroups of 256 threads per CU = N wavefronts per SIMD; ns_per_wave_inst_per_simd x N x 60 = ns a SIMD spends on one column of all its wavefronts\", "
           "\"iters\": %d,\n \"rows\": [", iters);
    for (int n = 1; n <= 8; ++n)
    {
        // n blocks fit, n + 1 do not: lds in (160 KB / (n + 1), 160 KB / n]
        const int lds = n == 1 ? 96 * 1024 : (160 * 1024 / n) & ~255;
        const Place p = { n, 256, lds, n };
        measure([&](int blocks, const Place& q) { hipLaunchKernelGGL((k_recurrence<3>), dim3(blocks), dim3(q.threads), q.lds, 0, g_out, g_in, iters, g_tm, (uint32_t*)nullptr); },
                (const void*)k_recurrence<3>, "fill_recurrence_C10_integer_adds", 0, 120, iters, p);
    }
    printf("\n ]}\n");
}

int main(int argc, char** argv)
{
    const bool occupancy = argc > 1 && !strcmp(argv[1], "occupancy");
    if (occupancy)
    {
        --argc;
        ++argv;
    }
    const int iters = argc > 1 ? atoi(argv[1]) : 3000;
    int dev = 0;
    CK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    g_cus = prop.multiProcessorCount;
    int wall_khz = 0;
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev);
    CK(hipMalloc(&g_out, (size_t)g_cus * 2 * 1024 * 4));
    CK(hipMalloc(&g_in, 4096 * 4));
    CK(hipMalloc(&g_tm, sizeof(Times) * g_cus * 2 * 16));
    CK(hipMalloc(&g_state, (size_t)g_cus * 1024 * 32 * 4));
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; ++i)
        h[i] = 0x3c003c00u + ((uint32_t)i * 2654435761u >> 20);
    CK(hipMemcpy(g_in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    if (occupancy)
    {
        occupancy_sweep(iters / 2);
        return 0;
    }
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"compute_units\": %d, \"simds\": %d, \"clock_rate_khz_reported\": %d, "
           "\"wall_clock_rate_khz\": %d, \"iters\": %d,\n \"what\": \"per wave64 instruction: ns of one SIMD's time (host events over the launch / "
           "wave-instructions per SIMD) and s_memtime ticks of the issuing wave; chains = independent dependency chains per wavefront "
           "(1 = every instruction waits for the one before it); 64 instructions per asm block, nothing between them; "
           "memtime_per_memrealtime x 100 MHz = the rate s_memtime counts at\",\n",
           prop.name, prop.gcnArchName, g_cus, g_cus * 4, prop.clockRate, wall_khz, iters);
    {
        // is the s_nop the compiler puts after packed instructions needed?  (short run: the scores stay in f16's exact range)
        const std::vector<uint32_t> s0 = rec_state<0>(40), s1 = rec_state<1>(40), s2 = rec_state<2>(40);
        size_t d01 = 0, d02 = 0, nz = 0;
        for (size_t i = 0; i < s0.size(); ++i)
        {
            d01 += s0[i] != s1[i];
            d02 += s0[i] != s2[i];
            nz += s0[i] != 0;
        }
        printf(" \"recurrence_without_nops_vs_compiler_emitted\": {\"words\": %zu, \"nonzero\": %zu, \"differ_no_nop\": %zu, \"differ_block_with_nop\": %zu},\n",
               s0.size(), nz, d01, d02);
    }
    printf(" \"rows\": [");
    run_op<Op_pk_maximum3_f16>(iters);
    run_op<Op_pk_add_f16>(iters);
    run_op<Op_pk_add_f16_nop>(iters);
    run_op<Op_pk_maximum3_f16_nop>(iters);
    run_op<Op_pk_max_u16>(iters);
    run_op<Op_pk_add_u16>(iters);
    run_op<Op_pk_max_f16>(iters);
    run_op<Op_pk_fma_f16>(iters);
    run_op<Op_perm_b32>(iters);
    run_op<Op_bfi_b32>(iters);
    run_op<Op_add_u32>(iters);
    run_op<Op_max_u32>(iters);
    run_op<Op_max3_u32>(iters);
    run_op<Op_add_f32>(iters);
    run_op<Op_fma_f32>(iters);
    run_op<Op_mov_b32>(iters);
    run_op<Op_mov_dpp_row_shr1>(iters);
    run_op<Op_add_u32_dpp_row_shr1>(iters);
    classify<Op_sub_u32>(iters);
    classify<Op_add_u32_literal>(iters);
    classify<Op_add_u32_sgpr>(iters);
    classify<Op_and_b32>(iters);
    classify<Op_or_b32>(iters);
    classify<Op_xor_b32>(iters);
    classify<Op_lshlrev_b32>(iters);
    classify<Op_lshrrev_b32>(iters);
    classify<Op_cndmask_b32>(iters);
    classify<Op_cmp_lt_u32>(iters);
    classify<Op_max_f32>(iters);
    classify<Op_max_i32>(iters);
    classify<Op_min_u32>(iters);
    classify<Op_mul_f32>(iters);
    classify<Op_sub_f32>(iters);
    classify<Op_add_f16>(iters);
    classify<Op_max_f16>(iters);
    classify<Op_add_u16>(iters);
    classify<Op_max_u16>(iters);
    classify<Op_add3_u32>(iters);
    classify<Op_lshl_add_u32>(iters);
    classify<Op_and_or_b32>(iters);
    classify<Op_lshl_or_b32>(iters);
    classify<Op_alignbit_b32>(iters);
    classify<Op_bfe_u32>(iters);
    classify<Op_max3_f32>(iters);
    classify<Op_maximum3_f32>(iters);
    classify<Op_med3_i32>(iters);
    classify<Op_mad_u32_u24>(iters);
    classify<Op_mul_lo_u32>(iters);
    classify<Op_pk_sub_u16>(iters);
    classify<Op_pk_ashrrev_i16>(iters);
    classify<Op_pk_mul_f16>(iters);
    classify<Op_mov_b32_sgpr>(iters);
    classify<Op_mov_b32_e64>(iters);
    classify<Op_add_u32_e64>(iters);
    classify<Op_readfirstlane_mov>(iters);
    classify<Op_sad_u8>(iters);
    classify<Op_cvt_f32_u32>(iters);
    classify<Op_ds_bpermute>(iters);
    run_rec<0>("fill_recurrence_C10_as_compiled", iters / 2);
    run_rec<1>("fill_recurrence_C10_one_block_no_nops", iters / 2);
    run_rec<2>("fill_recurrence_C10_one_block_with_nops", iters / 2);
    run_rec<3>("fill_recurrence_C10_integer_adds", iters / 2);
    run_rec<4>("fill_recurrence_C10_integer_adds_diagonals_first", iters / 2);
    run_rec<5>("fill_recurrence_C10_integer_adds_at_priority_1", iters / 2);
    run_rec<6>("fill_recurrence_C10_diagonals_first_additions_at_priority_1", iters / 2);
    run_rec<7>("fill_recurrence_C10_diagonals_first_run_at_priority_1", iters / 2);
    printf("\n ]}\n");
    return 0;
}
