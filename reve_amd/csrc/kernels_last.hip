// conv_last for the x2 model on gfx950: 64 -> 12 channel 3x3 convolution fused with PixelShuffle(2),
// the nearest-upsampled residual and the fp16 -> u8 post-process (k_last2).
//
// Same tile image, LDS-DMA double buffer, persistent XCD-aware tile walk and operand layout as the body
// kernel (kernels.hip), but with one co-block there is only ONE MFMA per B fragment, so the body's
// "read a fragment per tap" loop would be LDS-bound (the tile image is read 9 times).  This kernel is
// input-row stationary instead: a wave owns 4 output rows x 32 pixels (8 accumulators of 4 registers),
// walks the 6 input rows they touch, reads each (row, column tap, channel half) fragment ONCE and feeds
// it to the up-to-three output rows that use it as their dy = 0/1/2 tap: 72 ds_read_b128 instead of
// 144 for the same 144 MFMAs.
#include "kernels_dev.h"

namespace reve {

#ifndef LAST2_WAVES
#define LAST2_WAVES 8
#endif

// one LDS-DMA piece: 64 lanes x 16 B from rsrc[voff + soff] to LDS base + lane*16 (a plain device function:
// used directly inside the kernel template the builtin breaks the host-side instantiation)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, lds_void_t* dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 0);
}

// NW waves per workgroup (one workgroup per CU): 4 = one per SIMD, 8 = two per SIMD, so that a wave
// stalled on the issue of an LDS-DMA instruction leaves its SIMD to the other one.
template <int NW>
__global__ void __launch_bounds__(64 * NW, 1) k_last2(const ConvArgs a, const PlaneDesc* __restrict__ planes,
                                                       const uint32_t* __restrict__ items)
{
    constexpr int SCALE = 2;
    constexpr int ROWS = TILE_H / NW;             // output rows per wave
    constexpr int PER_WAVE = (DMA_PIECES + NW - 1) / NW;   // DMA pieces per wave
    constexpr int NPB = ROWS * 2;                 // 16-pixel blocks per wave
    constexpr int NSTEP = (ROWS + 2) * 3 * 2 * 2; // fragment reads per tile: input rows x dx x half x column block
    constexpr int DMA_SPAN = NSTEP - NSTEP / 6;   // the next tile's DMA pieces are issued over the first 5/6 of the reads
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = ROWS * wave;
    const int pl = lane & 15, g = lane >> 4;

    // register-stationary weights: k-step ks = (dy*3 + dx)*2 + half, one co-block (12 channels + 4 zero)
    h8 wf[KSTEPS];
    {
        const h8* wp = (const h8*)a.wpack;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) wf[s] = wp[s * 64 + lane];
    }
    float bias[4];
    {
        const h4 b = *(const h4*)(a.bias + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[r] = (float)b[r];
    }

    int roff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            roff[dx][hf] = (row0 * LDS_W + pl + dx) * PIX_BYTES + 16 * ((4 * hf + g) ^ ((pl + dx) & 6));

    auto piece = [&](int k) { const int c = k * NW + wave; return c < DMA_PIECES ? c : DMA_PIECES - 1; };
    int voff[PER_WAVE];
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        int q = piece(k) * 8 + (lane >> 3);
        q = q < LDS_PIX ? q : LDS_PIX - 1;
        const int yy = q / LDS_W, xx = q - yy * LDS_W;
        voff[k] = (yy * a.Wp + xx) * PIX_BYTES + 16 * ((lane & 7) ^ (xx & 6));
    }

    const int G = gridDim.x;
    const int b = blockIdx.x;
    int it = ((G & 7) == 0) ? (b & 7) * (G >> 3) + (b >> 3) : b;
    int cur = 0;
    if (it < a.n_items) {
        const Item itm = decode_item(it, a, items);
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (unsigned long long)itm.plane * a.plane_stride),
                                                      0, (int)a.plane_stride, 0x00020000);
        const int org = ((itm.ty * TILE_H) * a.Wp + itm.tx * TILE_W) * PIX_BYTES;
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k)
            dma16(rsrc, to_lds(smem + piece(k) * 1024), voff[k], org);
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) asm volatile("" : "+v"(wf[s]));   // pin the wait for the weights here
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    auto srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, (int)(a.src_stride * a.frame_h), 0x00020000);
    auto drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, (int)(a.dst_stride * a.frame_h * SCALE), 0x00020000);

    // The post-process of tile i runs under the MFMAs of tile i+1 (its 32 byte stores and ~600 VALU
    // instructions would otherwise sit exposed between two k-loops): a tile's accumulators, residual
    // bytes and coordinates are carried into the next iteration as the "previous tile".
    // PixelShuffle + nearest residual + post-process, cropped to the un-padded part of the plane
    // (ncnn-compat tiles carry an apron of a.pad px).  Lane (pl, g) holds channels 4g..4g+3 of pixel
    // pl = colour g, sub-pixels (i, j) = (r>>1, r&1); group 3 is the zero padding of the co-block.
    auto post = [&](int pb, const f4& acc, unsigned rbyte, const Item& t, const PlaneDesc& pd, bool valid) {
        const int oy = t.ty * TILE_H + row0 + (pb >> 1), ox = t.tx * TILE_W + 16 * (pb & 1) + pl;
        const bool inside = valid && oy >= a.pad && oy < pd.h - a.pad && ox >= a.pad && ox < pd.w - a.pad && g < 3;
        const int fy = pd.y0 + oy, fx = pd.x0 + ox;   // frame coordinates (inside => in range)
        const float res = (float)(_Float16)((float)rbyte * (1.0f / 255.0f));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = (float)(_Float16)acc[r];
            const float o = (float)(_Float16)(v + res);
            float qv = o * 255.0f + 0.5f;
            qv = qv > 0.f ? qv : 0.f;     // also maps NaN to 0 like the oracle
            qv = qv > 255.f ? 255.f : qv;
            const int off = (fy * SCALE + (r >> 1)) * (int)a.dst_stride + (fx * SCALE + (r & 1)) * 3 + g;
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)qv, drsrc, inside ? off : 0x7fffffff, 0, 0);
        }
    };

    f4 pacc[NPB];
    unsigned presid[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) { pacc[pb] = (f4){0.f, 0.f, 0.f, 0.f}; presid[pb] = 0; }
    Item pitm{0, 0, 0};
    PlaneDesc ppd = planes[0];
    bool pvalid = false;

    while (it < a.n_items) {
        const Item itm = decode_item(it, a, items);
        __builtin_amdgcn_s_barrier();      // every wave's DMA share of this tile has landed and
        asm volatile("" ::: "memory");     // every wave is done reading the other buffer
        const int nxt = it + G;
        const Item nitm = decode_item(nxt < a.n_items ? nxt : it, a, items);
        auto nrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (unsigned long long)nitm.plane * a.plane_stride),
                                                       0, (int)a.plane_stride, 0x00020000);
        const int norg = ((nitm.ty * TILE_H) * a.Wp + nitm.tx * TILE_W) * PIX_BYTES;
        char* nbuf = smem + (cur ^ 1) * LDS_BUF_BYTES;
        const int bufoff = cur * LDS_BUF_BYTES;
        const PlaneDesc pd = planes[itm.plane];

        // residual (nearest-upsampled input) bytes of this wave's pixels: lane group g reads colour g
        unsigned resid[NPB];
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            const int oy = itm.ty * TILE_H + row0 + (pb >> 1), ox = itm.tx * TILE_W + 16 * (pb & 1) + pl;
            int fy = pd.y0 + oy, fx = pd.x0 + ox;
            fy = fy < 0 ? 0 : (fy >= a.frame_h ? a.frame_h - 1 : fy);
            fx = fx < 0 ? 0 : (fx >= a.frame_w ? a.frame_w - 1 : fx);
            resid[pb] = __builtin_amdgcn_raw_buffer_load_b8(srsrc, fy * (int)a.src_stride + fx * 3 + (g < 3 ? g : 0), 0, 0);
        }

        f4 acc[NPB];
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) acc[pb] = (f4){bias[0], bias[1], bias[2], bias[3]};

#pragma unroll
        for (int iy = 0; iy < ROWS + 2; ++iy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int xb = 0; xb < 2; ++xb) {
                        const int step = ((iy * 3 + dx) * 2 + hf) * 2 + xb;
                        const h8 B = *(const h8*)(smem + bufoff + roff[dx][hf] + (iy * LDS_W + 16 * xb) * PIX_BYTES);
#pragma unroll
                        for (int k = 0; k < PER_WAVE; ++k)
                            if (k * DMA_SPAN / PER_WAVE == step)
                                dma16(nrsrc, to_lds(nbuf + piece(k) * 1024), voff[k], norg);
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int r = iy - dy;   // output row (within the wave's four) that sees input row iy as tap dy
                            if (r >= 0 && r < ROWS) acc[r * 2 + xb] = MFMA16(wf[(dy * 3 + dx) * 2 + hf], B, acc[r * 2 + xb]);
                        }
                        // previous tile's post-process, one 16-pixel block every 8 fragment reads
                        if (step % 8 == 3 && step / 8 < NPB) post(step / 8, pacc[step / 8], presid[step / 8], pitm, ppd, pvalid);
                    }
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) { pacc[pb] = acc[pb]; presid[pb] = resid[pb]; }
        pitm = itm; ppd = pd; pvalid = true;
        // this wave's pieces of the next tile must have landed before the barrier; the previous tile's
        // stores were issued in the first 60 reads and have long retired
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cur ^= 1;
        it = nxt;
    }
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) post(pb, pacc[pb], presid[pb], pitm, ppd, pvalid);
}

template __global__ void k_last2<LAST2_WAVES>(const ConvArgs, const PlaneDesc* __restrict__, const uint32_t* __restrict__);

int launch_last2(const ConvArgs& a, int grid, void* stream)
{
    constexpr int NW = LAST2_WAVES;
    static int once = (int)hipFuncSetAttribute((const void*)k_last2<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF_BYTES);
    if (once != 0) return once;
    hipLaunchKernelGGL(k_last2<NW>, dim3(grid), dim3(64 * NW), 2 * LDS_BUF_BYTES, (hipStream_t)stream, a, a.planes, a.items);
    return (int)hipGetLastError();
}

}  // namespace reve
