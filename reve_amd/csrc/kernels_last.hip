// conv_last of the x3 graph on gfx950: 64 -> 27 channel 3x3 convolution fused with PixelShuffle(3), the
// nearest-upsampled residual and the fp16 -> u8 post-process (k_last<3, 4, 1>).  The x2 and x4 graphs run their conv_last on
// the body kernel's row pipeline (kernels.hip, k_body<ORDER, 2 / 4>) since round 2: x4 went from 146 to 119 us that way (its
// 12-wave instance of this kernel could not pipeline its post-process for lack of registers and stored 4 bytes per lane where
// the pipeline stores 12), x2 stayed at 72 us (bound by the un-hidden latency of one tile's LDS-DMA either way).  The template
// keeps its x2 / x4 code paths (byte stores for x2, pack_last()'s x4 store order) but only the x3 instance is built.
//
// Same tile image, LDS-DMA double buffer, persistent XCD-aware tile walk and operand layout as the body
// kernel (kernels.hip), but a wave owns ONE 16-channel co-block, so there is only one MFMA per B
// fragment and the body's "read a fragment per tap" loop would be LDS-bound (the tile image is read 9
// times).  This kernel is input-row stationary instead: a wave owns ROWS output rows x 32 pixels of its
// co-block, walks the ROWS+2 input rows they touch, reads each (row, column tap, channel half) fragment
// ONCE and feeds it to the up-to-three output rows that use it as their dy = 0/1/2 tap.
//
// Workgroup = NRG row groups x NXH column halves x NCOB co-blocks waves, one workgroup per CU:
//   x3: 27 channels -> 2 co-blocks, 4 row groups of 4 rows  =  8 waves (2 per SIMD)
// More than one wave per SIMD matters here: the issue of an LDS-DMA instruction stalls a wave for ~130
// cycles and this kernel has too few MFMAs per tile to hide that inside one wave.
#include "kernels_dev.h"

namespace reve {

#ifdef STAMPS
// Diagnostic build only (scripts/stamps.py --last): per-wave cycle totals of the tile loop's segments.
__device__ unsigned long long g_stamps_last[2048 * 8];
#define LSTAMP(i)                                                                           \
    do {                                                                                    \
        unsigned long long t_;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        seg_[i] += t_ - last_;                                                              \
        last_ = t_;                                                                         \
    } while (0)
#else
#define LSTAMP(i) (void)0
#endif

constexpr int last_cobs(int scale) { return (3 * scale * scale + 15) / 16; }          // co-blocks computed
constexpr int last_cobs_packed(int scale) { return scale == 2 ? 1 : (scale == 3 ? 2 : 4); }   // model.cpp last_ncob()

template <int SCALE, int NRG, int NXH>
__global__ void __launch_bounds__(64 * NRG * NXH * last_cobs(SCALE), 1)
    k_last(const ConvArgs a, const PlaneDesc* __restrict__ planes, const uint32_t* __restrict__ items)
{
    constexpr int NCOB = last_cobs(SCALE), NPACK = last_cobs_packed(SCALE);
    constexpr int NW = NRG * NXH * NCOB;          // waves per workgroup
    constexpr int ROWS = TILE_H / NRG;            // output rows per wave
    constexpr int XBW = 2 / NXH;                  // 16-pixel column blocks per wave (the tile has two)
    constexpr int PER_WAVE = (DMA_PIECES + NW - 1) / NW;   // DMA pieces per wave
    constexpr int NPB = ROWS * XBW;               // 16-pixel blocks per wave
    constexpr int NSTEP = (ROWS + 2) * 3 * 2 * XBW;   // fragment reads per tile: input rows x dx x half x column block
#ifndef LAST_DMA_SPAN
#define LAST_DMA_SPAN (NSTEP - NSTEP / 6)
#endif
    constexpr int DMA_SPAN = LAST_DMA_SPAN;   // the next tile's DMA pieces are issued over the first DMA_SPAN reads
    constexpr int NRES = 1;                       // residual registers per pixel: x2 one byte (colour g), x3/x4 the pixel's RGB in one dword
    // With up to two waves per SIMD a tile's post-process is carried into the next iteration and runs
    // under that tile's MFMAs; with three the register budget (170) does not allow the second set of
    // accumulators and the other two waves cover it anyway.
    constexpr bool PIPE = NW <= 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves w, w+4, ... share a SIMD (NRG = 4): same rows, other column half / co-block
    const int rg = wave % NRG, xb0 = (wave / NRG) % NXH * XBW, cob = wave / (NRG * NXH);
    const int row0 = ROWS * rg;
    const int pl = lane & 15, g = lane >> 4;

    // register-stationary weights of this wave's co-block: k-step ks = (dy*3 + dx)*2 + half
    h8 wf[KSTEPS];
    {
        const h8* wp = (const h8*)a.wpack;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) wf[s] = wp[(s * NPACK + cob) * 64 + lane];
    }
    float bias[4];
    {
        const h4 b = *(const h4*)(a.bias + 16 * cob + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[r] = (float)b[r];
    }

    int roff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            roff[dx][hf] = (row0 * LDS_W + 16 * xb0 + pl + dx) * PIX_BYTES + 16 * ((4 * hf + g) ^ ((pl + dx) & 6));

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

    // (x3/x4 read the residual pixel as a dword: the range is rounded up so that a 3-byte frame still loads)
    auto srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, ((int)(a.src_stride * a.frame_h) + 3) & ~3, 0x00020000);
    auto drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, (int)(a.dst_stride * a.frame_h * SCALE), 0x00020000);

    // Residual (nearest-upsampled input) bytes of one 16-pixel block.  Lane (pl, g) holds channels
    // co = 16*cob + 4g + r of pixel pl; channel co is colour co / s^2, sub-pixel (co % s^2) / s, (co % s^2) % s:
    //   x2: colour g (group 3 is the co-block's zero padding);  x3: up to two colours per lane.
    // x4 uses the store order of pack_last() instead: row 4g + r of co-block cob is byte 4*cob + r of the 12-byte
    // run (4 sub-pixels x RGB) of output sub-row g, so a lane's four values are ONE aligned 4-byte store.
    auto fetch_resid = [&](int pb, const Item& t, const PlaneDesc& pd, unsigned (&out)[NRES]) {
        const int oy = t.ty * TILE_H + row0 + pb / XBW, ox = t.tx * TILE_W + 16 * (xb0 + pb % XBW) + pl;
        int fy = pd.y0 + oy, fx = pd.x0 + ox;
        fy = fy < 0 ? 0 : (fy >= a.frame_h ? a.frame_h - 1 : fy);
        fx = fx < 0 ? 0 : (fx >= a.frame_w ? a.frame_w - 1 : fx);
        const int off = fy * (int)a.src_stride + fx * 3;
        if constexpr (SCALE == 2) out[0] = __builtin_amdgcn_raw_buffer_load_b8(srsrc, off + (g < 3 ? g : 0), 0, 0);
        else {
            // one (unaligned) 4-byte load instead of three byte loads; at the very end of the frame buffer
            // the load is moved back inside it and the pixel shifted down into place
            const int lim = (int)(a.src_stride * a.frame_h) - 4;
            const int o4 = off < lim ? off : (lim > 0 ? lim : 0);
            out[0] = __builtin_amdgcn_raw_buffer_load_b32(srsrc, o4, 0, 0) >> (8 * (off - o4));
        }
    };
    // PixelShuffle + nearest residual + post-process, cropped to the un-padded part of the plane
    // (ncnn-compat tiles carry an apron of a.pad px).  Branch-free: masked lanes get an offset the
    // descriptor's bounds check drops.
    auto post = [&](int pb, const f4& acc, const unsigned (&rb)[NRES], const Item& t, const PlaneDesc& pd, bool valid) {
        const int oy = t.ty * TILE_H + row0 + pb / XBW, ox = t.tx * TILE_W + 16 * (xb0 + pb % XBW) + pl;
        const bool inside = valid && oy >= a.pad && oy < pd.h - a.pad && ox >= a.pad && ox < pd.w - a.pad;
        const int fy = pd.y0 + oy, fx = pd.x0 + ox;   // frame coordinates (inside => in range)
        // clamp(trunc(o * 255 + 0.5), 0, 255) as floor + v_cvt_pk_u8_f32: the conversion saturates to 0..255 and maps NaN to 0
        // like the oracle's comparison chain, and on the already integral value its rounding mode is immaterial; the byte is
        // merged into `into` at position `byte` (x4 builds its 4-byte word this way: no shifts, no ors)
        auto quant = [&](float accv, unsigned rbyte, unsigned byte, unsigned into) {
            const float res = (float)(_Float16)((float)rbyte * (1.0f / 255.0f));
            const float v = (float)(_Float16)accv;
            const float o = (float)(_Float16)(v + res);
            return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(o * 255.0f + 0.5f), byte, into);
        };
        if constexpr (SCALE == 4) {
            unsigned word = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = (4 * cob + r) % 3;   // uniform per wave
                word = quant(acc[r], (rb[0] >> (8 * c)) & 0xffu, r, word);
            }
            const int off = (fy * SCALE + g) * (int)a.dst_stride + fx * (3 * SCALE) + 4 * cob;
            __builtin_amdgcn_raw_buffer_store_b32(word, drsrc, inside ? off : 0x7fffffff, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * cob + 4 * g + r;
                const int c = co / (SCALE * SCALE), ij = co % (SCALE * SCALE);
                const int i = ij / SCALE, j = ij % SCALE;
                const unsigned rbyte = SCALE == 2 ? rb[0] : (rb[0] >> (8 * c)) & 0xffu;
                const bool ok = inside && co < 3 * SCALE * SCALE;
                const int off = (fy * SCALE + i) * (int)a.dst_stride + (fx * SCALE + j) * 3 + c;
#ifdef ABL_LAST_NOSTORE
                asm volatile("" ::"v"(quant(acc[r], rbyte, 0, 0)), "v"(ok ? off : 0x7fffffff));
#else
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)quant(acc[r], rbyte, 0, 0), drsrc, ok ? off : 0x7fffffff, 0, 0);
#endif
            }
        }
    };

    // the "previous tile" carried into the next iteration (PIPE only)
    f4 pacc[PIPE ? NPB : 1];
    unsigned presid[PIPE ? NPB : 1][NRES];
    Item pitm{0, 0, 0};
    PlaneDesc ppd = planes[0];
    bool pvalid = false;
    if constexpr (PIPE) {
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
            pacc[pb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NRES; ++c) presid[pb][c] = 0;
        }
    }

#ifdef STAMPS
    unsigned long long seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
    // Work items and plane descriptors are decoded two tiles ahead (scalar loads whose latency would
    // otherwise sit exposed right after the barrier, where every wave of the workgroup waits for it).
    auto item_at = [&](int i) { return decode_item(i < a.n_items ? i : it, a, items); };
    Item itm = item_at(it), nitm = item_at(it + G);
    PlaneDesc pd = planes[itm.plane], npd = planes[nitm.plane];
    while (it < a.n_items) {
        LSTAMP(7);
        __builtin_amdgcn_s_barrier();      // every wave's DMA share of this tile has landed and
        asm volatile("" ::: "memory");     // every wave is done reading the other buffer
        LSTAMP(0);                         // barrier wait
        const int nxt = it + G;
        const Item nnitm = item_at(nxt + G);             // used by the next iteration
        const PlaneDesc nnpd = planes[nnitm.plane];
        auto nrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (unsigned long long)nitm.plane * a.plane_stride),
                                                       0, (int)a.plane_stride, 0x00020000);
        const int norg = ((nitm.ty * TILE_H) * a.Wp + nitm.tx * TILE_W) * PIX_BYTES;
        char* nbuf = smem + (cur ^ 1) * LDS_BUF_BYTES;
        const int bufoff = cur * LDS_BUF_BYTES;

        unsigned resid[NPB][NRES];         // requested now, used after the MFMAs
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) fetch_resid(pb, itm, pd, resid[pb]);

        f4 acc[NPB];
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) acc[pb] = (f4){bias[0], bias[1], bias[2], bias[3]};
        LSTAMP(1);                         // tile set-up

#pragma unroll
        for (int iy = 0; iy < ROWS + 2; ++iy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int xb = 0; xb < XBW; ++xb) {
                        const int step = ((iy * 3 + dx) * 2 + hf) * XBW + xb;
                        const h8 B = *(const h8*)(smem + bufoff + roff[dx][hf] + (iy * LDS_W + 16 * xb) * PIX_BYTES);
#pragma unroll
                        for (int k = 0; k < PER_WAVE; ++k)
                            if (k * DMA_SPAN / PER_WAVE == step)
                                dma16(nrsrc, to_lds(nbuf + piece(k) * 1024), voff[k], norg);
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int r = iy - dy;   // output row (within the wave's rows) that sees input row iy as tap dy
                            if (r >= 0 && r < ROWS) acc[r * XBW + xb] = MFMA16(wf[(dy * 3 + dx) * 2 + hf], B, acc[r * XBW + xb]);
                        }
                        // previous tile's post-process, one 16-pixel block every 8 fragment reads
                        if constexpr (PIPE)
                            if (step % 8 == 3 && step / 8 < NPB) post(step / 8, pacc[step / 8], presid[step / 8], pitm, ppd, pvalid);
                    }
        LSTAMP(2);                         // k-loop (+ the previous tile's post-process with PIPE)
        if constexpr (PIPE) {
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
                pacc[pb] = acc[pb];
#pragma unroll
                for (int c = 0; c < NRES; ++c) presid[pb][c] = resid[pb][c];
            }
            pitm = itm; ppd = pd; pvalid = true;
        } else {
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) post(pb, acc[pb], resid[pb], itm, pd, true);
        }
        // this wave's pieces of the next tile must have landed before the barrier (with PIPE the previous
        // tile's stores were issued early in the k-loop and have long retired)
        LSTAMP(3);                         // post-process (without PIPE) / hand-over
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LSTAMP(4);                         // wait for the next tile's DMA (and this tile's stores)
        cur ^= 1;
        it = nxt;
        itm = nitm; pd = npd;
        nitm = nnitm; npd = nnpd;
    }
#ifdef STAMPS
    if (lane == 0 && blockIdx.x < 2048 / NW) {
#pragma unroll
        for (int i = 0; i < 8; ++i) g_stamps_last[(blockIdx.x * NW + wave) * 8 + i] = seg_[i];
    }
#endif
    if constexpr (PIPE) {
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) post(pb, pacc[pb], presid[pb], pitm, ppd, pvalid);
    }
}

template __global__ void k_last<3, 4, 1>(const ConvArgs, const PlaneDesc* __restrict__, const uint32_t* __restrict__);

#ifdef STAMPS
extern "C" int reve_debug_read_stamps_last(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_last), sizeof(unsigned long long) * n);
}
#endif

template <int SCALE, int NRG, int NXH>
static int launch(const ConvArgs& a, int grid, void* stream)
{
    hipLaunchKernelGGL((k_last<SCALE, NRG, NXH>), dim3(grid), dim3(64 * NRG * NXH * last_cobs(SCALE)), 2 * LDS_BUF_BYTES, (hipStream_t)stream,
                       a, a.planes, a.items);
    return (int)hipGetLastError();
}

int prepare_last_kernels()
{
    return (int)hipFuncSetAttribute((const void*)k_last<3, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF_BYTES);
}

int launch_last(const ConvArgs& a, int scale, int grid, void* stream)
{
    switch (scale) {
    case 2: return launch_last_x2(a, grid, stream);
    case 3: return launch<3, 4, 1>(a, grid, stream);
    case 4: return launch_last_x4(a, grid, stream);
    default: return -1;
    }
}

}  // namespace reve
