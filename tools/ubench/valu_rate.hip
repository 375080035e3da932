// Micro-benchmark: issue rate (cycles per wave64 instruction per SIMD) of candidate VALU ops for the DP core
// on gfx950.  8 waves per SIMD, 8 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define BC(T, x) __builtin_bit_cast(T, x)

template <int MODE> __device__ __forceinline__ uint32_t op(uint32_t x, uint32_t y)
{
    if (MODE == 0) return BC(uint32_t, __builtin_elementwise_max(BC(u16x2, x), BC(u16x2, y)));      // v_pk_max_u16
    if (MODE == 1) return BC(uint32_t, __builtin_elementwise_sub_sat(BC(u16x2, x), BC(u16x2, y)));  // v_pk_sub_u16 clamp
    if (MODE == 2) return BC(uint32_t, BC(u16x2, x) + BC(u16x2, y));                                 // v_pk_add_u16
    if (MODE == 3) return BC(uint32_t, __builtin_elementwise_max(BC(h2, x), BC(h2, y)));            // v_pk_max_f16
    if (MODE == 4) return BC(uint32_t, BC(h2, x) + BC(h2, y));                                       // v_pk_add_f16
    if (MODE == 5) return max(x, y);                                                                  // v_max_u32
    if (MODE == 6) return BC(uint32_t, fmaxf(BC(float, x), BC(float, y)));                           // v_max_f32
    if (MODE == 7) return BC(uint32_t, BC(float, x) + BC(float, y));                                  // v_add_f32
    if (MODE == 8) return __builtin_amdgcn_perm(x, y, 0x06040200u);                                   // v_perm_b32
    if (MODE == 9) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true) ^ y; // dpp mov + xor
    if (MODE == 10) return x ^ y;                                                                      // v_xor_b32
    if (MODE == 11) return BC(uint32_t, __builtin_elementwise_min(BC(h2, x), BC(h2, y)));            // v_pk_min_f16
    if (MODE == 12)                                                                                    // v_pk_maximum3_f16
        return BC(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_maximum(BC(h2, x), BC(h2, y)), BC(h2, (x ^ 0x00010001u))));
    if (MODE == 13) return max(max(x, y), x ^ 1u) ;                                                    // v_max3_u32 (+ xor folded?)
    if (MODE == 14) return BC(uint32_t, __builtin_elementwise_maximum(BC(h2, x), BC(h2, y)));       // v_pk_maximum_f16 / maximum3 w/ dup
    return x;
}

template <int MODE> __global__ __launch_bounds__(256) void k(uint32_t* out, const uint32_t* in, int iters)
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
                x[i] = op<MODE>(x[i], y[(i + u) & 7]);
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> double run(uint32_t* d, uint32_t* in, int blocks, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, in, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, in, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    const int blocks = 256 * 8, iters = 4000;
    uint32_t *d, *in;
    hipMalloc(&d, blocks * 256 * 4);
    hipMalloc(&in, 4096 * 4);
    uint32_t h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = 0x3c003c00u + (i * 2654435761u >> 20);  // small f16-ish patterns
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    const char* names[] = { "v_pk_max_u16", "v_pk_sub_u16 clamp", "v_pk_add_u16", "v_pk_max_f16", "v_pk_add_f16", "v_max_u32",
                            "v_max_f32", "v_add_f32", "v_perm_b32", "v_mov_dpp+v_xor (2)", "v_xor_b32", "v_pk_min_f16", "v_pk_maximum3_f16+xor (2)",
                            "v_max3_u32+xor (2)", "v_pk_maximum(3)_f16" };
    double ms[15];
    ms[0] = run<0>(d, in, blocks, iters); ms[1] = run<1>(d, in, blocks, iters); ms[2] = run<2>(d, in, blocks, iters);
    ms[3] = run<3>(d, in, blocks, iters); ms[4] = run<4>(d, in, blocks, iters); ms[5] = run<5>(d, in, blocks, iters);
    ms[6] = run<6>(d, in, blocks, iters); ms[7] = run<7>(d, in, blocks, iters); ms[8] = run<8>(d, in, blocks, iters);
    ms[9] = run<9>(d, in, blocks, iters); ms[10] = run<10>(d, in, blocks, iters); ms[11] = run<11>(d, in, blocks, iters);
    ms[12] = run<12>(d, in, blocks, iters); ms[13] = run<13>(d, in, blocks, iters); ms[14] = run<14>(d, in, blocks, iters);
    // wave-instructions per SIMD: 8 waves/SIMD x iters x 64
    const double per_simd = 8.0 * iters * 64.0;
    for (int m = 0; m < 15; ++m)
    {
        const double mult = (m == 9 || m == 12 || m == 13) ? 2.0 : 1.0;
        const double ns_per_instr = ms[m] * 1e6 / (per_simd * mult);
        printf("%-22s %8.3f ms  %6.3f ns/wave-instr/SIMD  = %5.2f cycles @2.1GHz  %5.2f @2.4GHz\n", names[m], ms[m], ns_per_instr,
               ns_per_instr * 2.1, ns_per_instr * 2.4);
    }
    return 0;
}
