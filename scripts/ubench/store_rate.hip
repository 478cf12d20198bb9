// Micro-benchmark: what store rate does ONE CU sustain with 16-B-per-lane stores, as a function of
// waves per CU and of the address pattern?  (Explains the body kernel's 25 us store cost)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// pattern 0: each wave-instruction writes 1 KiB contiguous; pattern 1: 16 pixels x 64 B at 128-B stride
template <int PATTERN>
__global__ void k_store(char* out, size_t bytes_per_wave, int iters)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    char* base = out + wave * bytes_per_wave;
    u32x4 v = {(unsigned)lane, 1u, 2u, (unsigned)wave};
    for (int i = 0; i < iters; ++i) {
        size_t off;
        if (PATTERN == 0) off = (size_t)i * 1024 + lane * 16;
        else off = (size_t)(i >> 1) * 2048 + (lane & 15) * 128 + (i & 1) * 64 + (lane >> 4) * 16;
        *(u32x4*)(base + off) = v;
        v.y += i;
    }
}

int main()
{
    const size_t total = 256ull << 20;   // 256 MiB per launch
    char* d;
    hipMalloc(&d, total);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pattern = 0; pattern < 2; ++pattern)
        for (int wpc : {4, 8, 16, 32}) {
            const int waves = 256 * wpc, blocks = waves / 4;
            const size_t bpw = total / waves;
            const int iters = (int)(bpw / 1024);
            float best = 1e9;
            for (int r = 0; r < 5; ++r) {
                hipEventRecord(e0);
                if (pattern == 0) hipLaunchKernelGGL(k_store<0>, dim3(blocks), dim3(256), 0, 0, d, bpw, iters);
                else hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(256), 0, 0, d, bpw, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("pattern %d, %2d waves/CU: %.1f us for 256 MiB = %.2f TB/s = %.1f B/clk/CU @2.4GHz\n", pattern, wpc, best * 1e3,
                   total / (best * 1e-3) / 1e12, total / (best * 1e-3) / 256 / 2.4e9);
        }
    return 0;
}
