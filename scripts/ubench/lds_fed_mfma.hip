// Micro-benchmark (round 5): what would a body kernel sustain that keeps the layer's WEIGHTS IN LDS instead of in registers?
// k_pair holds one layer's A fragments in 288 registers, which leaves room for ONE wave per SIMD: whatever is not an MFMA (operand
// reads, epilogue, addresses) is issued beside an idle matrix pipe.  With the weights in LDS a wave needs only its
// accumulators and a few operand fragments (< 256 registers): TWO waves per SIMD, each hiding the other's non-MFMA instructions —
// at the price of reading A from LDS as well.  Variants, whole chip, random data, no global traffic, no epilogue:
//   regA   1 wave / SIMD, A in registers (18 x 4 fragments), B from LDS, 24 reads per 144 MFMAs          (k_pair's operand diet)
//   ldsA   2 waves / SIMD, A and B from LDS: per k-step 4 A + PB B fragments for 4 x PB MFMAs (PB = 6: 10 reads per 24 MFMAs)
//   ldsA1  the same with ONE wave / SIMD (what the second wave buys)
// Prints TFLOP/s of each under the package power cap; mfma_rate.hip gives the bare-MFMA ceiling of the same box.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 scripts/ubench/lds_fed_mfma.hip -o scripts/ubench/lds_fed_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
constexpr int KSTEPS = 18, NCOB = 4;
constexpr int A_BYTES = KSTEPS * NCOB * 1024;          // 72 KiB: one layer's fragments
constexpr int B_ROWS = 8, B_BYTES = B_ROWS * 8192;     // an 8-row ring of 64 px x 128 B
constexpr int LDS = A_BYTES + B_BYTES + 1024;

// A in registers, B from LDS: two rows per step, the operand window of k_pair abstracted to its read count (24 per 144 MFMAs)
__global__ void __launch_bounds__(256, 1) k_regA(const h8* w, const h8* px, float* out, int steps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < B_BYTES / 16; i += 256) ((h8*)(smem + A_BYTES))[i] = px[i % 4096];
    h8 wf[KSTEPS][NCOB];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < NCOB; ++m) wf[s][m] = w[((s * NCOB + m) * 64 + lane) % 4096];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < NCOB; ++m) {
            if (s * NCOB + m < 64) asm volatile("" : "+a"(wf[s][m]));
            else asm volatile("" : "+v"(wf[s][m]));
        }
    float sum = 0.f;
    const char* B = smem + A_BYTES + wave * 2048 + lane * 16;
    for (int it = 0; it < steps; ++it) {
        f4 acc[NCOB][2];
#pragma unroll
        for (int m = 0; m < NCOB; ++m) { acc[m][0] = (f4){0, 0, 0, 0}; acc[m][1] = (f4){0, 0, 0, 0}; }
#pragma unroll
        for (int q = 0; q < 2; ++q) {                   // two px-blocks x 18 k-steps x 4 co-blocks x 2 rows = 288 MFMAs, 48 reads
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                h8 b0, b1;
                if (s % 3 == 0 || s < 6) b0 = *(const h8*)(B + ((it + s + 8 * q) & 7) * 8192 + (s & 3) * 1024);       // (2 of 3 slots reuse a fragment: the window)
                else b0 = *(const h8*)(B + ((it + 8 * q) & 7) * 8192);
                b1 = *(const h8*)(B + ((it + s + 1 + 8 * q) & 7) * 8192 + (s & 1) * 4096);
                if (s % 3 != 0 && s >= 6) asm volatile("" : "+v"(b0));
#pragma unroll
                for (int m = 0; m < NCOB; ++m) { acc[m][0] = MFMA16(wf[s][m], b0, acc[m][0]); acc[m][1] = MFMA16(wf[s][m], b1, acc[m][1]); }
            }
        }
#pragma unroll
        for (int m = 0; m < NCOB; ++m) sum += acc[m][0][0] + acc[m][1][3];
    }
    if (sum == 123.456f) out[0] = sum;
}

// A and B from LDS: PB px-blocks x 4 co-blocks of accumulators per wave; per k-step 4 A reads + PB B reads, 4 x PB MFMAs
template <int PB, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 1) k_ldsA(const h8* w, const h8* px, float* out, int steps)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < A_BYTES / 16; i += 64 * WAVES) ((h8*)smem)[i] = w[i % 4096];
    for (int i = threadIdx.x; i < B_BYTES / 16; i += 64 * WAVES) ((h8*)(smem + A_BYTES))[i] = px[i % 4096];
    __syncthreads();
    float sum = 0.f;
    const char* A = smem + lane * 16;
    const char* B = smem + A_BYTES + (wave & 7) * 1024 + lane * 16;
    for (int it = 0; it < steps; ++it) {
        f4 acc[NCOB][PB];
#pragma unroll
        for (int m = 0; m < NCOB; ++m)
#pragma unroll
            for (int p = 0; p < PB; ++p) acc[m][p] = (f4){0, 0, 0, 0};
#pragma unroll 2
        for (int s = 0; s < KSTEPS; ++s) {          // (unrolled by two, not fully: two waves per SIMD have 256 registers each)
            h8 a[NCOB], b[PB];
#pragma unroll
            for (int m = 0; m < NCOB; ++m) a[m] = *(const h8*)(A + (s * NCOB + m) * 1024);
#pragma unroll
            for (int p = 0; p < PB; ++p) b[p] = *(const h8*)(B + ((it + s + p) & 7) * 8192 + (p & 3) * 2048 - (p & 3) * 1024);
#pragma unroll
            for (int m = 0; m < NCOB; ++m)
#pragma unroll
                for (int p = 0; p < PB; ++p) acc[m][p] = MFMA16(a[m], b[p], acc[m][p]);
        }
#pragma unroll
        for (int m = 0; m < NCOB; ++m)
#pragma unroll
            for (int p = 0; p < PB; ++p) sum += acc[m][p][0] + acc[m][p][3];
    }
    if (sum == 123.456f) out[0] = sum;
}

template <typename K>
static void timeit(const char* name, K launch, double mfma_per_wg_step, int steps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r) best = ms < best ? ms : best;
    }
    const double flop = 256.0 * mfma_per_wg_step * steps * 16384.0;
    printf("%-44s %8.2f ms  %7.0f TFLOP/s\n", name, best, flop / (best * 1e-3) / 1e12);
}

int main()
{
    h8 *d_w, *d_px; float* d_out;
    hipMalloc(&d_w, 4096 * sizeof(h8)); hipMalloc(&d_px, 4096 * sizeof(h8)); hipMalloc(&d_out, 64);
    _Float16* h = (_Float16*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.5f);
    hipMemcpy(d_w, h, 4096 * 16, hipMemcpyHostToDevice);
    for (int i = 0; i < 4096 * 8; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    hipMemcpy(d_px, h, 4096 * 16, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_regA, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)k_ldsA<6, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)k_ldsA<6, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)k_ldsA<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipFuncSetAttribute((const void*)k_ldsA<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const int steps = 3000;
    timeit("regA: 1 wave/SIMD, A in registers, B from LDS", [&] { hipLaunchKernelGGL(k_regA, dim3(256), dim3(256), LDS, 0, d_w, d_px, d_out, steps); }, 4 * 288.0, steps);
    timeit("ldsA: 2 waves/SIMD, A + B from LDS, PB = 6", [&] { hipLaunchKernelGGL((k_ldsA<6, 8>), dim3(256), dim3(512), LDS, 0, d_w, d_px, d_out, steps); }, 8 * 18.0 * 24, steps);
    timeit("ldsA1: 1 wave/SIMD, A + B from LDS, PB = 6", [&] { hipLaunchKernelGGL((k_ldsA<6, 4>), dim3(256), dim3(256), LDS, 0, d_w, d_px, d_out, steps); }, 4 * 18.0 * 24, steps);
    timeit("ldsA: 2 waves/SIMD, A + B from LDS, PB = 8", [&] { hipLaunchKernelGGL((k_ldsA<8, 8>), dim3(256), dim3(512), LDS, 0, d_w, d_px, d_out, steps); }, 8 * 18.0 * 32, steps);
    timeit("ldsA: 2 waves/SIMD, A + B from LDS, PB = 4", [&] { hipLaunchKernelGGL((k_ldsA<4, 8>), dim3(256), dim3(512), LDS, 0, d_w, d_px, d_out, steps); }, 8 * 18.0 * 16, steps);
    return 0;
}
