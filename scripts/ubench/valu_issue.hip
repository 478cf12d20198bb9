// Micro-benchmark: what a VALU instruction costs ONE wave per SIMD (the occupancy of the 512-register kernels here) in issue
// cycles, alone and in the shadow of v_mfma_f32_16x16x32_f16: plain fp32 (v_add_f32), packed fp16 (v_pk_fma_f16, v_pk_add_f16),
// fp32 fma (v_fma_f32), and the same with K of them placed behind every MFMA.  The numbers behind DESIGN.md §4's
// "a packed fp16 instruction occupies a lone wave's issue for 8 cycles" (why Winograd's transforms cost what its MFMAs save).
// One workgroup of 256 threads per CU, cycles by s_memtime around the loop, median over workgroups printed per variant.
// Sections: (1) compiler-scheduled loops of 64 MFMAs with K VALU behind each; (2) hand-written groups of 8 MFMAs mixing VALU, SALU,
// s_nop and ds_read_b128 (their loop overhead — one taken branch per 8 MFMAs, ~30 cycles — is in every line: compare lines, not
// absolutes); (3) v_mfma_f32_32x32x16_f16 and the packed fp32 instructions; (4) register banks, v_fma_mix*, v_cvt_pk, v_pk_mul / max.
// What it shows (profiles/r04/ubench_valu_issue.txt): a lone wave issues ONE instruction of any class per 4 cycles; an MFMA
// 16x16x32 takes 8 of its 16 cycles of issue (32x32x16: 16 of 32), so two other instructions ride free behind it and every further
// one costs 4 cycles; v_fma_mixlo / mixhi_f16 take 8; v_pk_add / fma_f32 beside MFMAs cost ~9 each; source register banks do not
// matter; a taken branch costs ~50 cycles.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// KIND 0: v_add_f32, 1: v_pk_fma_f16, 2: v_pk_add_f16, 3: v_fma_f32 (build with -fno-slp-vectorize: no v_pk_add_f32); NV VALU instructions per group, NM MFMAs per group
template <int KIND, int NV, int NM>
__global__ void __launch_bounds__(256, 1) k(const h8* in, float* out, int iters, unsigned long long* cyc)
{
    const int lane = threadIdx.x & 63;
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(i * 64 + lane) % 2048]; b[i] = in[(1024 + i * 64 + lane) % 2048]; }
    f4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    float x[16];
    h2 p[16], q = {(_Float16)1.0009765625f, (_Float16)0.99951171875f};
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = (float)lane * 0.001f + i; p[i] = (h2){(_Float16)(lane * 0.01f + i), (_Float16)(i * 0.5f)}; }
    float c = 0.999f;
    asm volatile("" : "+v"(q), "+v"(c));
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[(grp * NM + m) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(grp + m) & 3], b[m & 3], acc[(grp * NM + m) & 7], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = (grp * NV + v) & 15;
                if constexpr (KIND == 0) x[i] = x[i] + c;
                else if constexpr (KIND == 1) p[i] = __builtin_elementwise_fma(p[i], q, q);
                else if constexpr (KIND == 2) p[i] = p[i] + q;
                else x[i] = __builtin_fmaf(x[i], c, c);
            }
            if constexpr (NM > 0) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, (NV + NM - 1) / NM, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += x[i] + (float)p[i][0] + (float)p[i][1];
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][3];
    if (sum == 123.456f) out[0] = sum;
    if (lane == 0 && (threadIdx.x >> 6) == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NV, int NM>
static void run(const char* what, const h8* d_in, float* d_out, unsigned long long* d_cyc)
{
    const int iters = 2000, grid = 256;
    hipLaunchKernelGGL((k<KIND, NV, NM>), dim3(grid), dim3(256), 0, 0, d_in, d_out, iters, d_cyc);
    hipLaunchKernelGGL((k<KIND, NV, NM>), dim3(grid), dim3(256), 0, 0, d_in, d_out, iters, d_cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(grid);
    hipMemcpy(c.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double per_group = (double)c[grid / 2] / (iters * 8.0);
    printf("%-46s %2d VALU + %2d MFMA per group: %7.1f cycles per group", what, NV, NM, per_group);
    if (NM == 0) printf("  = %.2f cycles per VALU instruction\n", per_group / NV);
    else printf("  (MFMAs alone would take %d; %.2f cycles beyond that per VALU instruction)\n", 16 * NM, (per_group - 16.0 * NM) / (NV ? NV : 1));
}


// Mixed instruction classes, written out as one asm block per group so that the order is exactly the one named:
// MIX 0: [MFMA, 2 v_add_f32, 2 s_add_u32] x 8; 1: [MFMA, 2 v_add_f32, 2 s_nop] x 8; 2: [MFMA, 4 s_add_u32] x 8; 3: 32 s_add_u32;
// 4: 32 s_nop 0; 5: [MFMA, 2 v_add_f32, 1 ds_read_b128] x 8 (+ one s_waitcnt lgkmcnt(0) per group); 6: [MFMA, 1 ds_read_b128] x 8;
// 7: [MFMA, 3 v_add_f32, 1 s_add_u32] x 8; 8: [MFMA, 2 v_add_f32] x 8 (the asm twin of the builtin variant above)
#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define MF "v_mfma_f32_16x16x32_f16 %[c0], %[a], %[b], %[c0]\n\t"
#define MF1 "v_mfma_f32_16x16x32_f16 %[c1], %[a], %[b], %[c1]\n\t"
#define VA "v_add_f32 %[x0], %[k], %[x0]\n\t"
#define VB "v_add_f32 %[x1], %[k], %[x1]\n\t"
#define SA "s_add_u32 %[s0], %[s0], 3\n\t"
#define SB "s_add_u32 %[s1], %[s1], 5\n\t"
#define SN "s_nop 0\n\t"
#define DS "ds_read_b128 %[d], %[addr]\n\t"
template <int MIX>
__global__ void __launch_bounds__(256, 1) kmix(const h8* in, float* out, int iters, unsigned long long* cyc)
{
    __shared__ h8 lds[512];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = in[threadIdx.x];
    lds[256 + threadIdx.x] = in[256 + threadIdx.x];
    __syncthreads();
    h8 a = in[lane], b = in[64 + lane];
    f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    float x0 = lane, x1 = lane * 2.f, k = 0.999f;
    unsigned s0 = 1, s1 = 2;
    u32x4 d = {0u, 0u, 0u, 0u};
    unsigned addr = (unsigned)(size_t)(lds) + 16 * ((lane * 5) & 255);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#define BODY(str) asm volatile(str : [c0] "+v"(c0), [c1] "+v"(c1), [x0] "+v"(x0), [x1] "+v"(x1), [s0] "+s"(s0), [s1] "+s"(s1), [d] "+v"(d) : [a] "v"(a), [b] "v"(b), [k] "v"(k), [addr] "v"(addr) : "memory", "scc")
        if constexpr (MIX == 0) BODY(R4(MF VA VB SA SB MF1 VA VB SA SB));
        else if constexpr (MIX == 1) BODY(R4(MF VA VB SN SN MF1 VA VB SN SN));
        else if constexpr (MIX == 2) BODY(R4(MF SA SB SA SB MF1 SA SB SA SB));
        else if constexpr (MIX == 3) BODY(R8(SA SB SA SB));
        else if constexpr (MIX == 4) BODY(R8(SN SN SN SN));
        else if constexpr (MIX == 5) BODY(R4(MF VA VB DS MF1 VA VB DS) "s_waitcnt lgkmcnt(0)\n\t");
        else if constexpr (MIX == 6) BODY(R4(MF DS MF1 DS) "s_waitcnt lgkmcnt(0)\n\t");
        else if constexpr (MIX == 7) BODY(R4(MF VA VB VA SA MF1 VA VB VA SB));
        else BODY(R4(MF VA VB MF1 VA VB));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    const float sum = c0[0] + c1[3] + x0 + x1 + (float)(s0 + s1) + (float)d[0];
    if (sum == 123.456f) out[0] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MIX>
static void run_mix(const char* what, int n_mfma, int n_other, const h8* d_in, float* d_out, unsigned long long* d_cyc)
{
    const int iters = 4000, grid = 256;
    hipLaunchKernelGGL((kmix<MIX>), dim3(grid), dim3(256), 0, 0, d_in, d_out, iters, d_cyc);
    hipLaunchKernelGGL((kmix<MIX>), dim3(grid), dim3(256), 0, 0, d_in, d_out, iters, d_cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(grid);
    hipMemcpy(c.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double per_group = (double)c[grid / 2] / iters;
    printf("%-58s %7.1f cycles per group of %d MFMA + %2d others", what, per_group, n_mfma, n_other);
    if (n_mfma) printf("  = %.2f per MFMA slot (16.6 alone)\n", per_group / n_mfma);
    else printf("  = %.2f cycles per instruction\n", per_group / n_other);
}

// The same question for v_mfma_f32_32x32x16_f16 (32 cycles of pipe: how many issue turns does it take?) and for the packed fp32
// instructions (two adds per instruction: one turn or two?).  PK 0: NV v_add_f32 behind each MFMA, 1: NV v_pk_add_f32, 2: v_pk_fma_f32
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK, int NV, int NM>
__global__ void __launch_bounds__(256, 1) k32(const h8* in, float* out, int iters, unsigned long long* cyc)
{
    const int lane = threadIdx.x & 63;
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = in[(i * 64 + lane) % 2048]; b[i] = in[(1024 + i * 64 + lane) % 2048]; }
    f16v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float x[32];
    float c = 0.999f;
    asm volatile("" : "+v"(c));
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = (float)lane * 0.001f + i;
    f2 xp[16], cc = {c, c};
#pragma unroll
    for (int i = 0; i < 16; ++i) xp[i] = (f2){x[i], x[i + 16]};
    asm volatile("" : "+v"(cc));
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[(grp * NM + m) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(grp + m) & 3], b[m & 3], acc[(grp * NM + m) & 3], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV * (NM ? NM : 1); ++v) {
                if constexpr (PK == 0) { const int i = (grp * NV + v) & 31; x[i] = x[i] + c; }
                else {
                    const int i = (grp * NV + v) & 15;
                    if constexpr (PK == 1) asm("v_pk_add_f32 %0, %0, %1" : "+v"(xp[i]) : "v"(cc));
                    else asm("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(xp[i]) : "v"(cc));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) sum += x[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += xp[i][0] + xp[i][1];
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][15];
    if (sum == 123.456f) out[0] = sum;
    if (lane == 0 && (threadIdx.x >> 6) == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int PK, int NV, int NM>
static void run32(const char* what, const h8* d_in, float* d_out, unsigned long long* d_cyc)
{
    const int iters = 2000, grid = 256;
    hipLaunchKernelGGL((k32<PK, NV, NM>), dim3(grid), dim3(256), 0, 0, d_in, d_out, iters, d_cyc);
    hipLaunchKernelGGL((k32<PK, NV, NM>), dim3(grid), dim3(256), 0, 0, d_in, d_out, iters, d_cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(grid);
    hipMemcpy(c.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double per_group = (double)c[grid / 2] / (iters * 4.0);
    if (NM) printf("%-50s %6.1f cycles per MFMA 32x32x16 with %d behind it (32 of pipe)\n", what, per_group / NM, NV);
    else printf("%-50s %6.2f cycles per instruction\n", what, per_group / NV);
}

// VGPR banks: a VALU instruction whose sources sit in the same register bank (register number mod 4) — what element-wise
// arithmetic on two 4-register vectors always does, element k of each in bank (base + k) mod 4 with both bases aligned alike.
// Hard-coded registers v100..v131: BANK 0: v_pk_add_f16 with both sources in one bank, 1: in different banks, 2: v_pk_fma_f16 with
// three sources in one bank, 3: in three banks, 4: v_add_f32 one bank, 5: two banks, 6: v_pk_fma_f16 two VGPR sources in one bank +
// an SGPR, 7: the same in two banks; each [MFMA, 3 of them] x 8 and alone x 32.
template <int BANK, bool WITH_MFMA>
__global__ void __launch_bounds__(256, 1) kbank(const h8* in, float* out, int iters, unsigned long long* cyc)
{
    const int lane = threadIdx.x & 63;
    h8 a = in[lane], b = in[64 + lane];
    f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    unsigned sc = 0x3c003c00u;
    asm volatile("" : "+s"(sc));
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#define BANKCLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131"
#define MFB0 "v_mfma_f32_16x16x32_f16 %[c0], %[a], %[b], %[c0]\n\t"
#define MFB1 "v_mfma_f32_16x16x32_f16 %[c1], %[a], %[b], %[c1]\n\t"
#define BODYB(m0, m1, i0, i1, i2, i3, i4, i5) asm volatile(R4(m0 i0 i1 i2 m1 i3 i4 i5) : [c0] "+v"(c0), [c1] "+v"(c1) : [a] "v"(a), [b] "v"(b), [s] "s"(sc) : "memory", BANKCLOB)
    for (int it = 0; it < iters; ++it) {
        if constexpr (WITH_MFMA) {
            if constexpr (BANK == 0) BODYB(MFB0, MFB1, "v_pk_add_f16 v120, v100, v104\n\t", "v_pk_add_f16 v121, v101, v105\n\t", "v_pk_add_f16 v122, v102, v106\n\t", "v_pk_add_f16 v123, v103, v107\n\t", "v_pk_add_f16 v124, v108, v112\n\t", "v_pk_add_f16 v125, v109, v113\n\t");
            else if constexpr (BANK == 1) BODYB(MFB0, MFB1, "v_pk_add_f16 v120, v100, v105\n\t", "v_pk_add_f16 v121, v101, v106\n\t", "v_pk_add_f16 v122, v102, v107\n\t", "v_pk_add_f16 v123, v103, v104\n\t", "v_pk_add_f16 v124, v108, v113\n\t", "v_pk_add_f16 v125, v109, v114\n\t");
            else if constexpr (BANK == 2) BODYB(MFB0, MFB1, "v_pk_fma_f16 v120, v100, v104, v108\n\t", "v_pk_fma_f16 v121, v101, v105, v109\n\t", "v_pk_fma_f16 v122, v102, v106, v110\n\t", "v_pk_fma_f16 v123, v103, v107, v111\n\t", "v_pk_fma_f16 v124, v112, v116, v100\n\t", "v_pk_fma_f16 v125, v113, v117, v101\n\t");
            else if constexpr (BANK == 3) BODYB(MFB0, MFB1, "v_pk_fma_f16 v120, v100, v105, v110\n\t", "v_pk_fma_f16 v121, v101, v106, v111\n\t", "v_pk_fma_f16 v122, v102, v107, v108\n\t", "v_pk_fma_f16 v123, v103, v104, v109\n\t", "v_pk_fma_f16 v124, v112, v117, v102\n\t", "v_pk_fma_f16 v125, v113, v118, v103\n\t");
            else if constexpr (BANK == 4) BODYB(MFB0, MFB1, "v_add_f32 v120, v100, v104\n\t", "v_add_f32 v121, v101, v105\n\t", "v_add_f32 v122, v102, v106\n\t", "v_add_f32 v123, v103, v107\n\t", "v_add_f32 v124, v108, v112\n\t", "v_add_f32 v125, v109, v113\n\t");
            else if constexpr (BANK == 5) BODYB(MFB0, MFB1, "v_add_f32 v120, v100, v105\n\t", "v_add_f32 v121, v101, v106\n\t", "v_add_f32 v122, v102, v107\n\t", "v_add_f32 v123, v103, v104\n\t", "v_add_f32 v124, v108, v113\n\t", "v_add_f32 v125, v109, v114\n\t");
            else if constexpr (BANK == 6) BODYB(MFB0, MFB1, "v_pk_fma_f16 v120, v100, %[s], v104\n\t", "v_pk_fma_f16 v121, v101, %[s], v105\n\t", "v_pk_fma_f16 v122, v102, %[s], v106\n\t", "v_pk_fma_f16 v123, v103, %[s], v107\n\t", "v_pk_fma_f16 v124, v108, %[s], v112\n\t", "v_pk_fma_f16 v125, v109, %[s], v113\n\t");
            else if constexpr (BANK == 8) BODYB(MFB0, MFB1, "v_fma_mixlo_f16 v120, v100, 1.0, v105\n\t", "v_fma_mixhi_f16 v120, v101, 1.0, v106\n\t", "v_fma_mixlo_f16 v122, v102, 1.0, v107\n\t", "v_fma_mixhi_f16 v122, v103, 1.0, v104\n\t", "v_fma_mixlo_f16 v124, v108, 1.0, v113\n\t", "v_fma_mixhi_f16 v124, v109, 1.0, v114\n\t");
            else if constexpr (BANK == 11) BODYB(MFB0, MFB1, "v_fma_mixlo_f16 v120, v100, 1.0, v105\n\t", "v_fma_mixlo_f16 v121, v101, 1.0, v106\n\t", "v_fma_mixlo_f16 v122, v102, 1.0, v107\n\t", "v_fma_mixlo_f16 v123, v103, 1.0, v104\n\t", "v_fma_mixlo_f16 v124, v108, 1.0, v113\n\t", "v_fma_mixlo_f16 v125, v109, 1.0, v114\n\t");
            else if constexpr (BANK == 12) BODYB(MFB0, MFB1, "v_fma_mixlo_f16 v120, v100, 1.0, v105\n\t", "v_fma_mixlo_f16 v121, v101, 1.0, v106\n\t", "v_fma_mixlo_f16 v122, v102, 1.0, v107\n\t", "v_fma_mixhi_f16 v120, v103, 1.0, v104\n\t", "v_fma_mixhi_f16 v121, v108, 1.0, v113\n\t", "v_fma_mixhi_f16 v122, v109, 1.0, v114\n\t");
            else if constexpr (BANK == 9) BODYB(MFB0, MFB1, "v_cvt_pk_f16_f32 v120, v100, v105\n\t", "v_cvt_pk_f16_f32 v121, v101, v106\n\t", "v_cvt_pk_f16_f32 v122, v102, v107\n\t", "v_cvt_pk_f16_f32 v123, v103, v104\n\t", "v_cvt_pk_f16_f32 v124, v108, v113\n\t", "v_cvt_pk_f16_f32 v125, v109, v114\n\t");
            else if constexpr (BANK == 10) BODYB(MFB0, MFB1, "v_pk_mul_f16 v120, v100, v105\n\t", "v_pk_max_f16 v121, v101, v106\n\t", "v_pk_mul_f16 v122, v102, v107\n\t", "v_pk_max_f16 v123, v103, v104\n\t", "v_pk_mul_f16 v124, v108, v113\n\t", "v_pk_max_f16 v125, v109, v114\n\t");
            else BODYB(MFB0, MFB1, "v_pk_fma_f16 v120, v100, %[s], v105\n\t", "v_pk_fma_f16 v121, v101, %[s], v106\n\t", "v_pk_fma_f16 v122, v102, %[s], v107\n\t", "v_pk_fma_f16 v123, v103, %[s], v104\n\t", "v_pk_fma_f16 v124, v108, %[s], v113\n\t", "v_pk_fma_f16 v125, v109, %[s], v114\n\t");
        } else {
            if constexpr (BANK == 0) BODYB("", "", "v_pk_add_f16 v120, v100, v104\n\t", "v_pk_add_f16 v121, v101, v105\n\t", "v_pk_add_f16 v122, v102, v106\n\t", "v_pk_add_f16 v123, v103, v107\n\t", "v_pk_add_f16 v124, v108, v112\n\t", "v_pk_add_f16 v125, v109, v113\n\t");
            else if constexpr (BANK == 1) BODYB("", "", "v_pk_add_f16 v120, v100, v105\n\t", "v_pk_add_f16 v121, v101, v106\n\t", "v_pk_add_f16 v122, v102, v107\n\t", "v_pk_add_f16 v123, v103, v104\n\t", "v_pk_add_f16 v124, v108, v113\n\t", "v_pk_add_f16 v125, v109, v114\n\t");
            else if constexpr (BANK == 2) BODYB("", "", "v_pk_fma_f16 v120, v100, v104, v108\n\t", "v_pk_fma_f16 v121, v101, v105, v109\n\t", "v_pk_fma_f16 v122, v102, v106, v110\n\t", "v_pk_fma_f16 v123, v103, v107, v111\n\t", "v_pk_fma_f16 v124, v112, v116, v100\n\t", "v_pk_fma_f16 v125, v113, v117, v101\n\t");
            else if constexpr (BANK == 3) BODYB("", "", "v_pk_fma_f16 v120, v100, v105, v110\n\t", "v_pk_fma_f16 v121, v101, v106, v111\n\t", "v_pk_fma_f16 v122, v102, v107, v108\n\t", "v_pk_fma_f16 v123, v103, v104, v109\n\t", "v_pk_fma_f16 v124, v112, v117, v102\n\t", "v_pk_fma_f16 v125, v113, v118, v103\n\t");
            else if constexpr (BANK == 4) BODYB("", "", "v_add_f32 v120, v100, v104\n\t", "v_add_f32 v121, v101, v105\n\t", "v_add_f32 v122, v102, v106\n\t", "v_add_f32 v123, v103, v107\n\t", "v_add_f32 v124, v108, v112\n\t", "v_add_f32 v125, v109, v113\n\t");
            else if constexpr (BANK == 5) BODYB("", "", "v_add_f32 v120, v100, v105\n\t", "v_add_f32 v121, v101, v106\n\t", "v_add_f32 v122, v102, v107\n\t", "v_add_f32 v123, v103, v104\n\t", "v_add_f32 v124, v108, v113\n\t", "v_add_f32 v125, v109, v114\n\t");
            else if constexpr (BANK == 6) BODYB("", "", "v_pk_fma_f16 v120, v100, %[s], v104\n\t", "v_pk_fma_f16 v121, v101, %[s], v105\n\t", "v_pk_fma_f16 v122, v102, %[s], v106\n\t", "v_pk_fma_f16 v123, v103, %[s], v107\n\t", "v_pk_fma_f16 v124, v108, %[s], v112\n\t", "v_pk_fma_f16 v125, v109, %[s], v113\n\t");
            else BODYB("", "", "v_pk_fma_f16 v120, v100, %[s], v105\n\t", "v_pk_fma_f16 v121, v101, %[s], v106\n\t", "v_pk_fma_f16 v122, v102, %[s], v107\n\t", "v_pk_fma_f16 v123, v103, %[s], v104\n\t", "v_pk_fma_f16 v124, v108, %[s], v113\n\t", "v_pk_fma_f16 v125, v109, %[s], v114\n\t");
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    const float sum = c0[0] + c1[3];
    if (sum == 123.456f) out[0] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int BANK, bool WITH_MFMA>
static void run_bank(const char* what, const h8* d_in, float* d_out, unsigned long long* d_cyc)
{
    const int iters = 4000, grid = 256;
    hipLaunchKernelGGL((kbank<BANK, WITH_MFMA>), dim3(grid), dim3(256), 0, 0, d_in, d_out, iters, d_cyc);
    hipLaunchKernelGGL((kbank<BANK, WITH_MFMA>), dim3(grid), dim3(256), 0, 0, d_in, d_out, iters, d_cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> c(grid);
    hipMemcpy(c.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double per = (double)c[grid / 2] / iters;
    if (WITH_MFMA) printf("%-64s %6.2f cycles per MFMA with 3 behind it\n", what, per / 8);
    else printf("%-64s %6.2f cycles per instruction\n", what, per / 24);
}

int main()
{
    h8* d_in; float* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_in, 2048 * sizeof(h8)); hipMalloc(&d_out, 64); hipMalloc(&d_cyc, 256 * 8);
    std::vector<_Float16> h(2048 * 8);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((s >> 9) & 0xffff) / 65536.0f - 0.5f); }
    hipMemcpy(d_in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<0, 16, 0>("v_add_f32 alone", d_in, d_out, d_cyc);
    run<1, 16, 0>("v_pk_fma_f16 alone", d_in, d_out, d_cyc);
    run<2, 16, 0>("v_pk_add_f16 alone", d_in, d_out, d_cyc);
    run<3, 16, 0>("v_fma_f32 alone", d_in, d_out, d_cyc);
    run<0, 0, 8>("MFMA 16x16x32 f16 alone", d_in, d_out, d_cyc);
    run<0, 8, 8>("v_add_f32, 1 behind each MFMA", d_in, d_out, d_cyc);
    run<0, 16, 8>("v_add_f32, 2 behind each MFMA", d_in, d_out, d_cyc);
    run<0, 24, 8>("v_add_f32, 3 behind each MFMA", d_in, d_out, d_cyc);
    run<0, 32, 8>("v_add_f32, 4 behind each MFMA", d_in, d_out, d_cyc);
    run<1, 8, 8>("v_pk_fma_f16, 1 behind each MFMA", d_in, d_out, d_cyc);
    run<1, 16, 8>("v_pk_fma_f16, 2 behind each MFMA", d_in, d_out, d_cyc);
    run<1, 24, 8>("v_pk_fma_f16, 3 behind each MFMA", d_in, d_out, d_cyc);
    run<2, 16, 8>("v_pk_add_f16, 2 behind each MFMA", d_in, d_out, d_cyc);
    run<3, 16, 8>("v_fma_f32, 2 behind each MFMA", d_in, d_out, d_cyc);
    run<3, 32, 8>("v_fma_f32, 4 behind each MFMA", d_in, d_out, d_cyc);
    run_mix<8>("asm: [MFMA, 2 v_add_f32] x 8", 8, 16, d_in, d_out, d_cyc);
    run_mix<0>("asm: [MFMA, 2 v_add_f32, 2 s_add_u32] x 8", 8, 32, d_in, d_out, d_cyc);
    run_mix<1>("asm: [MFMA, 2 v_add_f32, 2 s_nop 0] x 8", 8, 32, d_in, d_out, d_cyc);
    run_mix<7>("asm: [MFMA, 3 v_add_f32, 1 s_add_u32] x 8", 8, 32, d_in, d_out, d_cyc);
    run_mix<2>("asm: [MFMA, 4 s_add_u32] x 8", 8, 32, d_in, d_out, d_cyc);
    run_mix<3>("asm: 32 s_add_u32", 0, 32, d_in, d_out, d_cyc);
    run_mix<4>("asm: 32 s_nop 0", 0, 32, d_in, d_out, d_cyc);
    run_mix<5>("asm: [MFMA, 2 v_add_f32, 1 ds_read_b128] x 8 + waitcnt", 8, 25, d_in, d_out, d_cyc);
    run_mix<6>("asm: [MFMA, 1 ds_read_b128] x 8 + waitcnt", 8, 9, d_in, d_out, d_cyc);
    run32<0, 0, 2>("MFMA 32x32x16 f16 alone", d_in, d_out, d_cyc);
    run32<0, 2, 2>("v_add_f32 behind MFMA 32x32x16", d_in, d_out, d_cyc);
    run32<0, 4, 2>("v_add_f32 behind MFMA 32x32x16", d_in, d_out, d_cyc);
    run32<0, 5, 2>("v_add_f32 behind MFMA 32x32x16", d_in, d_out, d_cyc);
    run32<0, 6, 2>("v_add_f32 behind MFMA 32x32x16", d_in, d_out, d_cyc);
    run32<0, 7, 2>("v_add_f32 behind MFMA 32x32x16", d_in, d_out, d_cyc);
    run32<0, 8, 2>("v_add_f32 behind MFMA 32x32x16", d_in, d_out, d_cyc);
    run32<0, 12, 2>("v_add_f32 behind MFMA 32x32x16", d_in, d_out, d_cyc);
    run32<1, 16, 0>("v_pk_add_f32 alone", d_in, d_out, d_cyc);
    run32<2, 16, 0>("v_pk_fma_f32 alone", d_in, d_out, d_cyc);
    run32<1, 4, 2>("v_pk_add_f32 behind MFMA 32x32x16", d_in, d_out, d_cyc);
    run32<1, 8, 2>("v_pk_add_f32 behind MFMA 32x32x16", d_in, d_out, d_cyc);
    run32<2, 8, 2>("v_pk_fma_f32 behind MFMA 32x32x16", d_in, d_out, d_cyc);
    run_bank<0, false>("v_pk_add_f16, sources in ONE register bank, alone", d_in, d_out, d_cyc);
    run_bank<1, false>("v_pk_add_f16, sources in two banks, alone", d_in, d_out, d_cyc);
    run_bank<2, false>("v_pk_fma_f16, three sources in ONE bank, alone", d_in, d_out, d_cyc);
    run_bank<3, false>("v_pk_fma_f16, three banks, alone", d_in, d_out, d_cyc);
    run_bank<4, false>("v_add_f32, ONE bank, alone", d_in, d_out, d_cyc);
    run_bank<5, false>("v_add_f32, two banks, alone", d_in, d_out, d_cyc);
    run_bank<6, false>("v_pk_fma_f16 v, s, v: VGPRs in ONE bank, alone", d_in, d_out, d_cyc);
    run_bank<7, false>("v_pk_fma_f16 v, s, v: two banks, alone", d_in, d_out, d_cyc);
    run_bank<8, true>("v_fma_mixlo_f16 / v_fma_mixhi_f16 (fp32 sources)", d_in, d_out, d_cyc);
    run_bank<11, true>("v_fma_mixlo_f16 only, six destinations", d_in, d_out, d_cyc);
    run_bank<12, true>("3 mixlo then (behind the next MFMA) their 3 mixhi", d_in, d_out, d_cyc);
    run_bank<9, true>("v_cvt_pk_f16_f32", d_in, d_out, d_cyc);
    run_bank<10, true>("v_pk_mul_f16 / v_pk_max_f16", d_in, d_out, d_cyc);
    run_bank<0, true>("v_pk_add_f16, ONE bank", d_in, d_out, d_cyc);
    run_bank<1, true>("v_pk_add_f16, two banks", d_in, d_out, d_cyc);
    run_bank<2, true>("v_pk_fma_f16, three sources in ONE bank", d_in, d_out, d_cyc);
    run_bank<3, true>("v_pk_fma_f16, three banks", d_in, d_out, d_cyc);
    run_bank<4, true>("v_add_f32, ONE bank", d_in, d_out, d_cyc);
    run_bank<5, true>("v_add_f32, two banks", d_in, d_out, d_cyc);
    run_bank<6, true>("v_pk_fma_f16 v, s, v: VGPRs in ONE bank", d_in, d_out, d_cyc);
    run_bank<7, true>("v_pk_fma_f16 v, s, v: two banks", d_in, d_out, d_cyc);
    return 0;
}
