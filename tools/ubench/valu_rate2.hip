// Issue rate of individual gfx950 VALU instructions (inline asm so that nothing is folded away): 8 waves per SIMD,
// 8 independent chains per wave.  Prints cycles per wave64 instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define OP2(NAME, TXT)                                                                   \
    struct NAME                                                                           \
    {                                                                                     \
        static __device__ __forceinline__ uint32_t f(uint32_t x, uint32_t y, uint32_t z)  \
        {                                                                                 \
            uint32_t r;                                                                   \
            asm volatile(TXT : "=v"(r) : "v"(x), "v"(y), "v"(z));                         \
            return r;                                                                     \
        }                                                                                 \
    };
OP2(PkMaxU16, "v_pk_max_u16 %0, %1, %2")
OP2(PkMaxI16, "v_pk_max_i16 %0, %1, %2")
OP2(PkAddU16, "v_pk_add_u16 %0, %1, %2")
OP2(PkSubU16Clamp, "v_pk_sub_u16 %0, %1, %2 clamp")
OP2(PkAddF16, "v_pk_add_f16 %0, %1, %2")
OP2(PkMulF16, "v_pk_mul_f16 %0, %1, %2")
OP2(PkFmaF16, "v_pk_fma_f16 %0, %1, %2, %3")
OP2(PkMaxF16, "v_pk_max_f16 %0, %1, %2")
OP2(PkMaximum3_3reg, "v_pk_maximum3_f16 %0, %1, %2, %3")
OP2(PkMaximum3_2reg, "v_pk_maximum3_f16 %0, %1, %2, %2")
OP2(PkMaximum3_const, "v_pk_maximum3_f16 %0, %1, %2, 1.0 op_sel_hi:[1,1,0]")
OP2(PkMinimum3_2reg, "v_pk_minimum3_f16 %0, %1, %2, %2")
OP2(PkMadU16, "v_pk_mad_u16 %0, %1, %2, %3")
OP2(Max3U32, "v_max3_u32 %0, %1, %2, %3")
OP2(MaxU32, "v_max_u32 %0, %1, %2")
OP2(AddU32, "v_add_u32 %0, %1, %2")
OP2(Perm, "v_perm_b32 %0, %1, %2, %3")
OP2(Bfi, "v_bfi_b32 %0, %1, %2, %3")
OP2(Maximum3F32, "v_maximum3_f32 %0, %1, %2, %3")
OP2(PkAddI16, "v_pk_add_i16 %0, %1, %2")
OP2(AddF16, "v_add_f16 %0, %1, %2")
OP2(DotF16, "v_dot2_f32_f16 %0, %1, %2, %3")
OP2(SubU32, "v_sub_u32 %0, %1, %2")
OP2(AndB32, "v_and_b32 %0, %1, %2")
OP2(XorB32, "v_xor_b32 %0, %1, %2")
OP2(LshlB32, "v_lshlrev_b32 %0, 3, %1")
OP2(MinU32, "v_min_u32 %0, %1, %2")
OP2(MaxI32, "v_max_i32 %0, %1, %2")
OP2(AddF32, "v_add_f32 %0, %1, %2")
OP2(MulF32, "v_mul_f32 %0, %1, %2")
OP2(FmaF32, "v_fma_f32 %0, %1, %2, %3")
OP2(MaxF32, "v_max_f32 %0, %1, %2")
OP2(MovB32, "v_mov_b32 %0, %1")
OP2(AddU16, "v_add_u16 %0, %1, %2")
OP2(MaxU16, "v_max_u16 %0, %1, %2")
OP2(SubU16, "v_sub_u16 %0, %1, %2")
OP2(MaxF16, "v_max_f16 %0, %1, %2")
OP2(Add3U32, "v_add3_u32 %0, %1, %2, %3")
OP2(LshlAdd, "v_lshl_add_u32 %0, %1, 2, %2")
OP2(AndOr, "v_and_or_b32 %0, %1, %2, %3")
OP2(Med3I32, "v_med3_i32 %0, %1, %2, %3")
OP2(SubrevU32, "v_subrev_u32 %0, %1, %2")
OP2(MulLoU32, "v_mul_lo_u32 %0, %1, %2")
OP2(MulU24, "v_mul_u32_u24 %0, %1, %2")
OP2(MadU24, "v_mad_u32_u24 %0, %1, %2, %3")
OP2(SadU8, "v_sad_u8 %0, %1, %2, %3")
OP2(MaxU32e64, "v_max_u32_e64 %0, %1, %2")
OP2(AddU32e64, "v_add_u32_e64 %0, %1, %2")

template <typename O> __global__ __launch_bounds__(256) void k(uint32_t* out, const uint32_t* in, int iters)
{
    uint32_t x[8], y[8];
    for (int i = 0; i < 8; ++i)
    {
        x[i] = in[threadIdx.x * 8 + i];
        y[i] = in[2048 + ((threadIdx.x + i) & 255)];
    }
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                x[i] = O::f(x[i], y[(i + u) & 7], y[(i + u + 3) & 7]);
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i)
        s ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename O> void run(const char* name, uint32_t* d, uint32_t* in)
{
    const int blocks = 256 * 8, iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<O>, dim3(blocks), dim3(256), 0, 0, d, in, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<O>, dim3(blocks), dim3(256), 0, 0, d, in, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double per_simd = 8.0 * iters * 64.0;
    const double ns = ms * 1e6 / per_simd;
    printf("%-24s %7.3f ms  %5.2f cycles @2.1GHz  %5.2f @2.4GHz\n", name, ms, ns * 2.1, ns * 2.4);
}

int main()
{
    uint32_t *d, *in;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    hipMalloc(&in, 4096 * 4);
    uint32_t h[4096];
    for (int i = 0; i < 4096; ++i)
        h[i] = 0x3c003c00u + (i * 2654435761u >> 20);
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
#define R(T) run<T>(#T, d, in)
    R(PkMaxU16); R(PkMaxI16); R(PkAddU16); R(PkAddI16); R(PkSubU16Clamp); R(PkMadU16);
    R(PkAddF16); R(PkMulF16); R(PkFmaF16); R(PkMaxF16); R(AddF16);
    R(PkMaximum3_3reg); R(PkMaximum3_2reg); R(PkMaximum3_const); R(PkMinimum3_2reg); R(Maximum3F32);
    R(Max3U32); R(MaxU32); R(AddU32); R(Perm); R(Bfi); R(DotF16);
    R(SubU32); R(SubrevU32); R(AndB32); R(XorB32); R(LshlB32); R(MinU32); R(MaxI32); R(MaxU32e64); R(AddU32e64);
    R(AddF32); R(MulF32); R(FmaF32); R(MaxF32); R(MovB32);
    R(AddU16); R(MaxU16); R(SubU16); R(MaxF16); R(Add3U32); R(LshlAdd); R(AndOr); R(Med3I32); R(MulLoU32); R(MulU24); R(MadU24); R(SadU8);
    return 0;
}
