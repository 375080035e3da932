// sector_probe.hip -- what does one small read of an otherwise untouched 128-byte line cost in HBM traffic on gfx950?
// The traceback touches about one new line per path step and uses a few bytes of it; whether the memory side
// moves 128, 64 or 32 bytes for such a read decides what a different load flavour could save.  Each kernel below reads ONE dword
// per `stride` bytes of a 2 GiB buffer that was never read before it (a fresh region per kernel), with a different load:
// plain, non-temporal, system-scope (sc0 sc1), scalar (s_load_dword), and a byte load.  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE  (and TCC_EA0_RDREQ_sum / TCC_EA0_RDREQ_32B_sum in their own passes)
// and divide by the number of reads (tools/gpu/r03_n.sh).  Not a product program.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                         \
    do                                                                                   \
    {                                                                                    \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess)                                                            \
        {                                                                                \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

__global__ void k_plain(const uint32_t* __restrict__ p, size_t stride_words, uint32_t* out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = p[i * stride_words];
    if (v == 0x12345678u)
        out[0] = v;
}
__global__ void k_nt(const uint32_t* __restrict__ p, size_t stride_words, uint32_t* out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = __builtin_nontemporal_load(p + i * stride_words);
    if (v == 0x12345678u)
        out[0] = v;
}
__global__ void k_sc(const uint32_t* __restrict__ p, size_t stride_words, uint32_t* out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t* a = p + i * stride_words;
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    if (v == 0x12345678u)
        out[0] = v;
}
__global__ void k_byte(const uint32_t* __restrict__ p, size_t stride_words, uint32_t* out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint8_t v = ((const uint8_t*)(p + i * stride_words))[1];
    if (v == 0x5Au)
        out[0] = v;
}
// one scalar load per wavefront: the scalar cache has 64-byte lines
__global__ void k_scalar(const uint32_t* __restrict__ p, size_t stride_words, uint32_t* out)
{
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const uint32_t* a = p + (size_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w) * stride_words;
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(a) : "memory");
    if (v == 0x12345678u)
        out[0] = v;
}
// a 16-lane group reads 64 contiguous bytes (what one read's lanes own of a step record's dword row): half a line
__global__ void k_half(const uint32_t* __restrict__ p, size_t stride_words, uint32_t* out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = p[(i / 16) * stride_words + (i % 16)];
    if (v == 0x12345678u)
        out[0] = v;
}

int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1 << 22;  // reads per kernel
    const size_t strides[3] = { 128, 256, 1024 };
    // regions: 6 kernels x 3 strides, each n * stride bytes at most -> allocate per launch to keep every region cold
    uint32_t* out;
    CHECK(hipMalloc((void**)&out, 64));
    for (int s = 0; s < 3; ++s)
    {
        const size_t stride = strides[s];
        const size_t bytes = n * stride;
        for (int k = 0; k < 6; ++k)
        {
            uint32_t* buf;
            CHECK(hipMalloc((void**)&buf, bytes + 4096));
            CHECK(hipMemset(buf, 0, bytes + 4096));
            CHECK(hipDeviceSynchronize());
            // push the memset's lines out of the L2 / MALL: stream another buffer through
            {
                uint32_t* flush;
                CHECK(hipMalloc((void**)&flush, (size_t)1 << 30));
                CHECK(hipMemset(flush, 1, (size_t)1 << 30));
                CHECK(hipDeviceSynchronize());
                CHECK(hipFree(flush));
            }
            const dim3 block(256);
            const size_t sw = stride / 4;
            switch (k)
            {
            case 0: k_plain<<<dim3((unsigned)(n / 256)), block>>>(buf, sw, out); break;
            case 1: k_nt<<<dim3((unsigned)(n / 256)), block>>>(buf, sw, out); break;
            case 2: k_sc<<<dim3((unsigned)(n / 256)), block>>>(buf, sw, out); break;
            case 3: k_byte<<<dim3((unsigned)(n / 256)), block>>>(buf, sw, out); break;
            case 4: k_scalar<<<dim3((unsigned)(n * 64 / 256 / 16)), block>>>(buf, sw * 16, out); break;  // n/16 wavefronts, 16 x the stride
            case 5: k_half<<<dim3((unsigned)(n * 16 / 256 / 16)), block>>>(buf, sw * 16, out); break;  // n/16 groups of 16 lanes
            }
            CHECK(hipGetLastError());
            CHECK(hipDeviceSynchronize());
            CHECK(hipFree(buf));
        }
        printf("stride %zu: plain/nt/sc/byte read %zu dwords each; scalar %zu loads at stride %zu; half %zu groups of 64 B at stride %zu\n",
               stride, n, n / 16, stride * 16, n / 16, stride * 16);
    }
    return 0;
}
