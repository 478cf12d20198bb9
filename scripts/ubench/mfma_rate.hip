// Micro-benchmark: sustained fp16 MFMA rate of the whole chip on RANDOM register operands, one wave per
// SIMD (the body kernel's occupancy), for the two dense shapes.  Under the 1400 W package cap this is the
// practical MFMA ceiling the conv kernel can be compared with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// SLEEP > 0 inserts s_sleep(SLEEP) (64*SLEEP cycles) after every 16 MFMAs (256 issue cycles): a duty-cycle
// knob that tells a fixed clock (rate falls with duty) from a power-managed one (clock rises as duty falls).
template <int SHAPE, int SLEEP>
__global__ void __launch_bounds__(256, 1) k_mfma(const h8* in, float* out, int iters, long long* clk)
{
    const int lane = threadIdx.x & 63;
    const long long c0 = clock64(), r0 = wall_clock64();
    h8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = in[(i * 64 + lane) % 4096]; b[i] = in[(2048 + i * 64 + lane) % 4096]; }
    float sum = 0.f;
    if constexpr (SHAPE == 16) {
        f4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + k) & 7], b[k], acc[i], 0, 0, 0);
            if constexpr (SLEEP > 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) __builtin_amdgcn_s_sleep(SLEEP);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][3];
    } else {
        f16v acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + k) & 7], b[k], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][15];
    }
    if (sum == 123.456f) out[0] = sum;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
}

template <int SLEEP>
static void run16(const h8* d_in, float* d_out, long long* d_clk, hipEvent_t e0, hipEvent_t e1)
{
    const int iters = 20000;
    const double flop = 256.0 * 4 * iters * 8 * 16 * (2.0 * 16 * 16 * 32);
    for (int r = 0; r < 2; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_mfma<16, SLEEP>), dim3(256), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c[2]; hipMemcpy(c, d_clk, 16, hipMemcpyDeviceToHost);
        const double duty = 256.0 / (256.0 + 64.0 * SLEEP);
        if (r == 1) printf("mfma f16 16x16 sleep %d (issue duty %.0f %%): %.1f ms, %.0f TFLOP/s, clock64/wall_clock64 = %.3f\n",
                           SLEEP, duty * 100, ms, flop / (ms * 1e-3) / 1e12, (double)c[0] / (double)c[1]);
    }
}

int main()
{
    h8* d_in; float* d_out; long long* d_clk;
    hipMalloc(&d_in, 4096 * sizeof(h8)); hipMalloc(&d_out, 64); hipMalloc(&d_clk, 16);
    _Float16* h = (_Float16*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.5f);
    hipMemcpy(d_in, h, 4096 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    {
        const int iters = 20000;
        const double flop = 256.0 * 4 * iters * 8 * 4 * (2.0 * 32 * 32 * 16);
        for (int r = 0; r < 2; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k_mfma<32, 0>), dim3(256), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r == 1) printf("mfma f16 32x32: %.1f ms, %.0f TFLOP/s\n", ms, flop / (ms * 1e-3) / 1e12);
        }
    }
    run16<0>(d_in, d_out, d_clk, e0, e1);
    run16<1>(d_in, d_out, d_clk, e0, e1);
    run16<2>(d_in, d_out, d_clk, e0, e1);
    run16<4>(d_in, d_out, d_clk, e0, e1);
    run16<8>(d_in, d_out, d_clk, e0, e1);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock rate %d kHz\n", khz);
    return 0;
}
