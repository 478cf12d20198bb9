// Micro-benchmark / probe: visibility of data between workgroups on DIFFERENT XCDs inside one kernel.
// 256 persistent workgroups (one per CU, workgroup b on XCD b % 8).  Per round: workgroup b fills its 64-KiB
// slab with a round-specific pattern (buffer stores with cache policy SAUX), waits for them (vmcnt(0)), raises
// flag[b]; then polls flag[b+1] (the next XCD), reads THAT slab with buffer loads of policy LAUX and counts
// words that are not the expected pattern; acks, and waits for its own reader's ack before overwriting.
// The same addresses are rewritten every round, so a reader XCD whose L2 kept the line from an earlier
// round shows up as a mismatch.  Flags use relaxed agent-scope atomics (no fences: the point is to see what
// the data accesses' own cache-policy bits guarantee).  aux bits: 1 = sc0, 2 = nt, 16 = sc1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SLAB = 64 * 1024;   // bytes per workgroup

template <int SAUX, int LAUX>
__global__ void __launch_bounds__(256, 1) k_probe(char* data, unsigned* flag, unsigned* ack, unsigned long long* errs, int rounds,
                                                  long long* cyc)
{
    const int b = blockIdx.x, G = gridDim.x, t = threadIdx.x;
    const int src = (b + 1) % G;        // the slab this workgroup reads (a workgroup on the next XCD wrote it)
    const int reader = (b + G - 1) % G; // the workgroup that reads mine
    auto mine = __builtin_amdgcn_make_buffer_rsrc((void*)(data + (size_t)b * SLAB), 0, SLAB, 0x00020000);
    auto theirs = __builtin_amdgcn_make_buffer_rsrc((void*)(data + (size_t)src * SLAB), 0, SLAB, 0x00020000);
    unsigned long long bad = 0;
    const long long t0 = clock64();
    for (int r = 1; r <= rounds; ++r) {
        const unsigned pat = (unsigned)r * 2654435761u + (unsigned)b;
        for (int o = t * 16; o < SLAB; o += 256 * 16)
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){pat, pat ^ (unsigned)o, pat + 1u, pat ^ 0x5a5a5a5au}, mine, o, 0, SAUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_store(&flag[b], (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == 0)
            while (__hip_atomic_load(&flag[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) __builtin_amdgcn_s_sleep(2);
        __syncthreads();
        const unsigned exp = (unsigned)r * 2654435761u + (unsigned)src;
        for (int o = t * 16; o < SLAB; o += 256 * 16) {
            const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(theirs, o, 0, LAUX));
            bad += (v[0] != exp) + (v[1] != (exp ^ (unsigned)o)) + (v[2] != exp + 1u) + (v[3] != (exp ^ 0x5a5a5a5au));
        }
        __syncthreads();
        if (t == 0) __hip_atomic_store(&ack[b], (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == 0)
            while (__hip_atomic_load(&ack[reader], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) __builtin_amdgcn_s_sleep(2);
        __syncthreads();
    }
    if (bad) atomicAdd(errs, bad);
    if (b == 0 && t == 0) cyc[0] = clock64() - t0;
}

template <int SAUX, int LAUX>
static void run(const char* name, char* data, unsigned* flag, unsigned* ack, unsigned long long* errs, long long* cyc, int G, int rounds)
{
    hipMemset(flag, 0, G * 4); hipMemset(ack, 0, G * 4); hipMemset(errs, 0, 8); hipMemset(data, 0, (size_t)G * SLAB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<SAUX, LAUX>), dim3(G), dim3(256), 0, 0, data, flag, ack, errs, rounds, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h = 0; hipMemcpy(&h, errs, 8, hipMemcpyDeviceToHost);
    printf("%-34s mismatching words %12llu of %llu   %.2f us per round\n", name, h, (unsigned long long)G * (SLAB / 4) * rounds,
           ms * 1e3 / rounds);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int G = p.multiProcessorCount, rounds = 300;
    char* data; unsigned *flag, *ack; unsigned long long* errs; long long* cyc;
    hipMalloc(&data, (size_t)G * SLAB); hipMalloc(&flag, G * 4); hipMalloc(&ack, G * 4); hipMalloc(&errs, 8); hipMalloc(&cyc, 8);
    printf("%d workgroups, %d rounds, 64 KiB slabs, reader on the next XCD\n", G, rounds);
    run<0, 0>("stores plain, loads plain", data, flag, ack, errs, cyc, G, rounds);
    run<16, 0>("stores sc1, loads plain", data, flag, ack, errs, cyc, G, rounds);
    run<0, 16>("stores plain, loads sc1", data, flag, ack, errs, cyc, G, rounds);
    run<16, 16>("stores sc1, loads sc1", data, flag, ack, errs, cyc, G, rounds);
    run<17, 17>("stores sc0 sc1, loads sc0 sc1", data, flag, ack, errs, cyc, G, rounds);
    run<16, 17>("stores sc1, loads sc0 sc1", data, flag, ack, errs, cyc, G, rounds);
    return 0;
}
