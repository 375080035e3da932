#include <hip/hip_runtime.h>
__global__ void k(int* out)
{
    int v = threadIdx.x * 3 + 1;
    int s = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);  // wave_shr:1
    out[threadIdx.x] = s;
}
int main()
{
    int* d;
    hipMalloc(&d, 64 * 4);
    k<<<1, 64>>>(d);
    int h[64];
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i)
    {
        int want = i == 0 ? -1 : (i - 1) * 3 + 1;
        if (h[i] != want) { ++bad; printf("lane %d got %d want %d\n", i, h[i], want); }
    }
    printf("bad=%d\n", bad);
    return 0;
}
