// store_bw.hip -- what does HBM take when every wavefront streams through a private region with the fill kernel's
// store pattern?  (hipcc --offload-arch=gfx950 -O3 -o store_bw store_bw.hip && ./store_bw)
//   pattern 0: per step 5 x global_store_dword (256 B per instruction, 1280 B contiguous per wave-step) -- the trace stores
//   pattern 1: per 2 steps 5 x global_store_dwordx2 (512 B per instruction)
//   pattern 2: per 4 steps 5 x global_store_dwordx4 (1 KiB per instruction)
// with `work` dependent VALU operations between the stores of a step (0 = pure store rate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int PAT> __global__ __launch_bounds__(64) void k(uint32_t* out, uint32_t steps, uint32_t work)
{
    uint32_t* base = out + (size_t)blockIdx.x * steps * 320;  // 1280 B per step
    const uint32_t lane = threadIdx.x;
    uint32_t v = lane * 2654435761u + blockIdx.x;
    for (uint32_t t = 0; t < steps; t += (PAT == 0 ? 1 : PAT == 1 ? 2 : 4))
    {
        for (uint32_t w = 0; w < work; ++w)
            v = v * 1664525u + 1013904223u;
        if (PAT == 0)
        {
#pragma unroll
            for (int d = 0; d < 5; ++d)
                base[(size_t)t * 320 + d * 64 + lane] = v + d;
        }
        else if (PAT == 1)
        {
#pragma unroll
            for (int d = 0; d < 5; ++d)
                ((uint2*)(base + (size_t)t * 320))[d * 64 + lane] = make_uint2(v + d, v ^ d);
        }
        else
        {
#pragma unroll
            for (int d = 0; d < 5; ++d)
                ((uint4*)(base + (size_t)t * 320))[d * 64 + lane] = make_uint4(v + d, v ^ d, v - d, v);
        }
    }
}

int main(int argc, char** argv)
{
    const uint32_t waves = argc > 1 ? atoi(argv[1]) : 50000, steps = 520;
    uint32_t* d;
    const size_t bytes = (size_t)waves * steps * 1280;
    if (hipMalloc(&d, bytes) != hipSuccess)
        return 1;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int pat = 0; pat < 3; ++pat)
        for (uint32_t work : { 0u, 16u, 64u, 128u })
        {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep)
            {
                hipEventRecord(a);
                if (pat == 0)
                    hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, d, steps, work);
                else if (pat == 1)
                    hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, d, steps, work * 2);
                else
                    hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, d, steps, work * 4);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
            }
            printf("pattern %d work/step %3u: %.2f ms  %.2f TB/s\n", pat, work, best, bytes / best / 1e9);
        }
    return 0;
}
