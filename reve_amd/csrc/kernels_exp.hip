// Experimental body-layer variants (none faster than k_body; selected by environment
// variables in engine.cpp, documented in DESIGN.md §4):
//   k_body3      REVE_BODY3=1    8x32 tiles, triple-buffered LDS image, DMA two tiles ahead
//   k_conv64_o2  REVE_BODY_O2=1  two single-buffered workgroups per CU (2 waves per SIMD)
#include "kernels_dev.h"

namespace reve {

// -------------------------------------------------------------------------------------------
// Body layer, 8x32 tiles, TRIPLE-buffered LDS image (k_body3).
// The layer is bound by HBM traffic: to keep the CU's share of HBM busy while the
// MFMAs run it needs ~60 KB in flight at all times.  With two 16x32 buffers the next tile's DMA
// cannot be issued before the current tile starts and is awaited when it ends; here the DMA runs TWO
// tiles ahead (3 x 44,032 B of LDS), issued in one burst right after the barrier and awaited a full
// tile later with a counted vmcnt.  Same arena, same weights, same math as k_body.
// -------------------------------------------------------------------------------------------
constexpr int T3_H = 8;
constexpr int T3_LDS_H = T3_H + 2;
constexpr int T3_PIX = T3_LDS_H * LDS_W;                    // 340
constexpr int T3_PIECES = (T3_PIX + 7) / 8;                 // 43
constexpr int T3_DMA_PER_WAVE = (T3_PIECES + NWAVES - 1) / NWAVES;   // 11
constexpr int T3_BUF_BYTES = T3_PIECES * 1024;              // 44,032
constexpr int T3_NBUF = 3;

__device__ __forceinline__ int t3_piece(int k, int wave)
{
    const int c = k * NWAVES + wave;
    return c < T3_PIECES ? c : T3_PIECES - 1;
}

__global__ void __launch_bounds__(256, 1) k_body3(const ConvArgs a, const PlaneDesc* __restrict__ planes, int tiles_y8)
{
    constexpr int CPW = 2, SPB = 4, NSUB = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = 4 * (wave & 1);
    const int wh = wave >> 1;
    const int pl = lane & 15, g = lane >> 4;
    const int cob0 = wh * CPW;

    h8 wf[KSTEPS][CPW];
    {
        const h8* wp = (const h8*)a.wpack;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int m = 0; m < CPW; ++m) wf[s][m] = wp[(s * 4 + cob0 + m) * 64 + lane];
    }
    float bias[CPW][4];
#pragma unroll
    for (int m = 0; m < CPW; ++m) {
        const h4 b = *(const h4*)(a.bias + 16 * (cob0 + m) + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[m][r] = (float)b[r];
    }
    const h4 s0 = *(const h4*)(a.slope + 32 * wh + 4 * g), s1 = *(const h4*)(a.slope + 32 * wh + 16 + 4 * g);
    const h8 slope8 = __builtin_shufflevector(s0, s1, 0, 1, 2, 3, 4, 5, 6, 7);

    int roff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            roff[dx][hf] = (row0 * LDS_W + pl + dx) * PIX_BYTES + 16 * ((4 * hf + g) ^ ((pl + dx) & 6));
    int voff[T3_DMA_PER_WAVE];
#pragma unroll
    for (int k = 0; k < T3_DMA_PER_WAVE; ++k) {
        int q = t3_piece(k, wave) * 8 + (lane >> 3);
        q = q < T3_PIX ? q : T3_PIX - 1;
        const int yy = q / LDS_W, xx = q - yy * LDS_W;
        voff[k] = (yy * a.Wp + xx) * PIX_BYTES + 16 * ((lane & 7) ^ (xx & 6));
    }

    const int n_items = a.n_planes * a.tiles_x * tiles_y8;
    auto decode = [&](int it, int& plane, int& ty, int& tx) {
        if (a.reverse) it = n_items - 1 - it;
        const int per = a.tiles_x * tiles_y8;
        plane = it / per;
        const int rem = it - plane * per;
        ty = rem / a.tiles_x;
        tx = rem - ty * a.tiles_x;
    };
    auto dma_tile3 = [&](int it, int buf) {   // all of this wave's pieces of tile `it` (clamped: a re-load is harmless)
        int plane, ty, tx;
        decode(it < n_items ? it : n_items - 1, plane, ty, tx);
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (unsigned long long)plane * a.plane_stride),
                                                      0, (int)a.plane_stride, 0x00020000);
        const int org = ((ty * T3_H) * a.Wp + tx * TILE_W) * PIX_BYTES;
#pragma unroll
        for (int k = 0; k < T3_DMA_PER_WAVE; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, to_lds(smem + buf * T3_BUF_BYTES + t3_piece(k, wave) * 1024), 16,
                                                     voff[k], org, 0, 0);
    };

    const int G = gridDim.x;
    const int b = blockIdx.x;
    int it = ((G & 7) == 0) ? (b & 7) * (G >> 3) + (b >> 3) : b;
    int cur = 0;
    if (it < n_items) {
        dma_tile3(it, 0);
        dma_tile3(it + G, 1);
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < CPW; ++m) asm volatile("" : "+v"(wf[s][m]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    while (it < n_items) {
        int plane, ty, tx;
        decode(it, plane, ty, tx);
        __builtin_amdgcn_s_barrier();      // tile `it` has landed for every wave; buffer (cur+2)%3 is free
        asm volatile("" ::: "memory");
        int nb = cur + 2;
        nb = nb >= T3_NBUF ? nb - T3_NBUF : nb;
        dma_tile3(it + 2 * G, nb);         // two tiles ahead, one burst
        __builtin_amdgcn_sched_barrier(0); // nothing may move across: the counted vmcnt below relies on the order
        const int bufoff = cur * T3_BUF_BYTES;
        const PlaneDesc pd = planes[plane];
        auto orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (unsigned long long)plane * a.plane_stride),
                                                       0, (int)a.plane_stride, 0x00020000);
        u32x4 pend_o[SPB];
        int pend_off[SPB];
#pragma unroll
        for (int si = 0; si < NSUB; ++si) {
            f4 acc[CPW][SPB];
#pragma unroll
            for (int m = 0; m < CPW; ++m)
#pragma unroll
                for (int q = 0; q < SPB; ++q) acc[m][q] = (f4){bias[m][0], bias[m][1], bias[m][2], bias[m][3]};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3, dx = t % 3;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int ks = t * 2 + hf;
                    h8 B[SPB];
#pragma unroll
                    for (int q = 0; q < SPB; ++q) {
                        const int rr = (si * SPB + q) >> 1, xb = (si * SPB + q) & 1;
                        B[q] = *(const h8*)(smem + bufoff + roff[dx][hf] + ((rr + dy) * LDS_W + 16 * xb) * PIX_BYTES);
                    }
                    if (si > 0) {
#pragma unroll
                        for (int q = 0; q < SPB; ++q)
                            if (ks == 2 + q * (KSTEPS - 2) / SPB)
                                __builtin_amdgcn_raw_buffer_store_b128(pend_o[q], orsrc, pend_off[q], 0, 0);
                    }
#pragma unroll
                    for (int m = 0; m < CPW; ++m)
#pragma unroll
                        for (int q = 0; q < SPB; ++q) acc[m][q] = MFMA16(wf[ks][m], B[q], acc[m][q]);
                }
            }
#pragma unroll
            for (int q = 0; q < SPB; ++q) {
                const int rr = (si * SPB + q) >> 1, xb = (si * SPB + q) & 1;
                const int oy = ty * T3_H + row0 + rr;
                const int ox = tx * TILE_W + 16 * xb + pl;
                h8 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] = (_Float16)acc[0][q][r];
                    o[4 + r] = (_Float16)acc[1][q][r];
                }
                o = prelu8(o, slope8);
                const bool ok = oy < pd.h && ox < pd.w;
                const int off = ((oy + 1) * a.Wp + (ox + 1)) * PIX_BYTES + 64 * wh + 16 * g;
                if (si + 1 < NSUB) {
                    pend_o[q] = __builtin_bit_cast(u32x4, o);
                    pend_off[q] = ok ? off : 0x7fffffff;
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc, ok ? off : 0x7fffffff, 0, 0);
                }
            }
        }
        // youngest in issue order: this iteration's 11 DMA pieces (tile it+2G) and its 8 stores;
        // everything older, the next tile's pieces included, must have landed before the barrier
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T3_DMA_PER_WAVE + NSUB * SPB) : "memory");
        cur = cur + 1 >= T3_NBUF ? 0 : cur + 1;
        it += G;
    }
}

// -------------------------------------------------------------------------------------------
// Body layer, occupancy-2 variant: TWO workgroups per CU (two waves per SIMD, 256 registers each),
// each with ONE single-buffered 16x32 tile image in LDS.  A workgroup loads its tile, waits,
// computes, stores; while it waits on its loads/stores/barriers the other workgroup's waves own the
// SIMD's MFMA pipe, so the overlap of memory and matrix work comes from occupancy instead of
// from software pipelining inside one wave.
// -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) k_conv64_o2(const ConvArgs a, const PlaneDesc* __restrict__ planes,
                                                       const uint32_t* __restrict__ items)
{
    constexpr int CPW = 2, SPB = 4, NSUB = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = 8 * (wave & 1);
    const int wh = wave >> 1;
    const int pl = lane & 15, g = lane >> 4;
    const int cob0 = wh * CPW;

    h8 wf[KSTEPS][CPW];
    {
        const h8* wp = (const h8*)a.wpack;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int m = 0; m < CPW; ++m) wf[s][m] = wp[(s * 4 + cob0 + m) * 64 + lane];
    }
    h4 bias_h[CPW];
#pragma unroll
    for (int m = 0; m < CPW; ++m) bias_h[m] = *(const h4*)(a.bias + 16 * (cob0 + m) + 4 * g);
    const h4 s0 = *(const h4*)(a.slope + 32 * wh + 4 * g), s1 = *(const h4*)(a.slope + 32 * wh + 16 + 4 * g);
    const h8 slope8 = __builtin_shufflevector(s0, s1, 0, 1, 2, 3, 4, 5, 6, 7);

    int roff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            roff[dx][hf] = (row0 * LDS_W + pl + dx) * PIX_BYTES + 16 * ((4 * hf + g) ^ ((pl + dx) & 6));
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < CPW; ++m) asm volatile("" : "+v"(wf[s][m]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int G = gridDim.x;
    for (int it = blockIdx.x; it < a.n_items; it += G) {
        const Item itm = decode_item(it, a, items);
        const PlaneDesc pd = planes[itm.plane];
        // ---- load this tile (all 77 pieces at once; the other workgroup on this CU computes meanwhile)
        {
            auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (unsigned long long)itm.plane * a.plane_stride),
                                                          0, (int)a.plane_stride, 0x00020000);
            const int org = ((itm.ty * TILE_H) * a.Wp + itm.tx * TILE_W) * PIX_BYTES;
#pragma unroll
            for (int k = 0; k < DMA_PER_WAVE; ++k) {
                int q = dma_piece(k, wave) * 8 + (lane >> 3);
                q = q < LDS_PIX ? q : LDS_PIX - 1;
                const int yy = q / LDS_W, xx = q - yy * LDS_W;
                const int vo = (yy * a.Wp + xx) * PIX_BYTES + 16 * ((lane & 7) ^ (xx & 6));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, to_lds(smem + dma_piece(k, wave) * 1024), 16, vo, org, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        auto orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (unsigned long long)itm.plane * a.plane_stride),
                                                       0, (int)a.plane_stride, 0x00020000);
#pragma unroll
        for (int si = 0; si < NSUB; ++si) {
            f4 acc[CPW][SPB];
#pragma unroll
            for (int m = 0; m < CPW; ++m)
#pragma unroll
                for (int q = 0; q < SPB; ++q)
                    acc[m][q] = (f4){(float)bias_h[m][0], (float)bias_h[m][1], (float)bias_h[m][2], (float)bias_h[m][3]};
            // B fragments one k-step ahead; scheduling fences keep hipcc from hoisting every LDS read of
            // the sub-iteration to its top (which would not fit the 256-register budget)
            h8 Bq[2][SPB];
#pragma unroll
            for (int q = 0; q < SPB; ++q)
                Bq[0][q] = *(const h8*)(smem + roff[0][0] + (((si * SPB + q) >> 1) * LDS_W + 16 * ((si * SPB + q) & 1)) * PIX_BYTES);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                if (ks + 1 < KSTEPS) {
                    const int t = (ks + 1) >> 1, hf = (ks + 1) & 1, dy = t / 3, dx = t % 3;
#pragma unroll
                    for (int q = 0; q < SPB; ++q)
                        Bq[(ks + 1) & 1][q] = *(const h8*)(smem + roff[dx][hf] + ((((si * SPB + q) >> 1) + dy) * LDS_W + 16 * ((si * SPB + q) & 1)) * PIX_BYTES);
                }
#pragma unroll
                for (int m = 0; m < CPW; ++m)
#pragma unroll
                    for (int q = 0; q < SPB; ++q) acc[m][q] = MFMA16(wf[ks][m], Bq[ks & 1][q], acc[m][q]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < SPB; ++q) {
                const int rr = (si * SPB + q) >> 1, xb = (si * SPB + q) & 1;
                const int oy = itm.ty * TILE_H + row0 + rr;
                const int ox = itm.tx * TILE_W + 16 * xb + pl;
                h8 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[r] = (_Float16)acc[0][q][r];
                    o[4 + r] = (_Float16)acc[1][q][r];
                }
                o = prelu8(o, slope8);
                const bool ok = oy < pd.h && ox < pd.w;
                const int off = ((oy + 1) * a.Wp + (ox + 1)) * PIX_BYTES + 64 * wh + 16 * g;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc, ok ? off : 0x7fffffff, 0, 0);
            }
        }
        __builtin_amdgcn_s_barrier();      // every wave is done reading the tile image
        asm volatile("" ::: "memory");
    }
}

int prepare_exp_kernels()
{
    return (int)hipFuncSetAttribute((const void*)k_body3, hipFuncAttributeMaxDynamicSharedMemorySize, T3_NBUF * T3_BUF_BYTES) |
           (int)hipFuncSetAttribute((const void*)k_conv64_o2, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUF_BYTES);
}

int launch_body3(const ConvArgs& a, int tiles_y8, int grid, void* stream)
{
    hipLaunchKernelGGL(k_body3, dim3(grid), dim3(256), T3_NBUF * T3_BUF_BYTES, (hipStream_t)stream, a, a.planes, tiles_y8);
    return (int)hipGetLastError();
}

int launch_body_o2(const ConvArgs& a, int grid, void* stream)
{
    hipLaunchKernelGGL(k_conv64_o2, dim3(grid), dim3(256), LDS_BUF_BYTES, (hipStream_t)stream, a, a.planes, a.items);
    return (int)hipGetLastError();
}

}  // namespace reve
