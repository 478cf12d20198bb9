// Hand-written gfx950 (CDNA4 / MI355X) kernels for the realesr-animevideov3 SRVGGNetCompact graph: the body layers.
//
// Replaces the compute that ONdraid/reve reaches by spawning `realesrgan-ncnn-vulkan`
// (reve-shared/src/lib.rs:134-147): ncnn layers Convolution/PReLU x17, Convolution, PixelShuffle,
// Interp(nearest), BinaryOp(add) plus the binary's pre/post-processing.
// conv_first is in kernels_first.hip; conv_last (with PixelShuffle, the nearest residual and the post-process) runs on this
// kernel's pipeline as k_body<ORDER, 2 / 3 / 4>.
//
// Data layout in HBM ("activation arena"): planes of (tiles_y*16+2) x (tiles_x*32+2) pixels,
// 128 B per pixel (64 channels fp16, channel order permuted by chan_phys()), image pixel (0,0)
// at arena pixel (1,1).  Everything outside the image stays ZERO for the life of the arena, so
// the convolutions' zero padding and all halo loads need no bounds checks.
//
// k_body (64->64 + bias + fp16 round + PReLU): implicit GEMM on v_mfma_f32_16x16x32_f16 with
//   A = weights (16 output channels x 32 k), REGISTER-STATIONARY for the whole persistent launch: a workgroup is 4
//       waves (one per SIMD, 512-register budget); wave w owns tile rows w, w+4, w+8, w+12 and all 64 output channels
//       (18 k-steps x 4 co-blocks x 4 = 288 registers of weights, 256 of them in AGPRs: the MFMA reads its A operand
//       from there, -mllvm -amdgpu-mfma-vgpr-form=1).  All four waves need the same 72 KiB, so the fragments come in
//       through LDS once per workgroup (a quarter DMA'd by each wave into the second tile buffer);
//   B = pixels (32 k x 16 pixels), one ds_read_b128 per fragment (k = 8 consecutive physical channels of one tap) from
//       an LDS image of the (16+2)x(32+2) input tile, XOR-swizzled per column (kernels_dev.h), each fragment read once
//       per workgroup and fed to 4 MFMAs;
//   the tile image is filled by LDS-DMA (buffer_load_dwordx4 ... lds), double-buffered: the next tile's 77 pieces are
//   issued between the current tile's MFMAs and land under them; one s_barrier per tile.
//
// The instruction stream of a tile is ROW-PIPELINED (round 2; the round-1 kernel computed 2 rows x 2 px-blocks per
// sub-iteration and let hipcc place the epilogue — it came out as one ~150-instruction VALU block between two MFMA streams
// — and the B reads, which it sank to 3 MFMAs above their first use, inside the LDS latency: 9.2 k MFMA-issue cycles in
// 14.2 k per tile):
//   - a wave walks ONE row (2 px-blocks, 8 accumulators) at a time;
//   - B fragments are double-buffered in registers: the two ds_read_b128 of k-step F+1 are issued at the head of
//     k-step F, a full k-step (8 MFMAs, 128 cycles) ahead of their use;
//   - the epilogue of row r runs in four pieces (px-block x channel half, 16 VALU each) under the MFMAs of row r+1, each
//     piece's 16-byte store two k-steps after it; the last row of a tile is carried in registers across the barrier and
//     finished under the first row of the workgroup's NEXT tile;
//   - the 20 LDS-DMA pieces of the next tile sit on even k-steps of rows 0-2, epilogue pieces and stores on odd ones;
//   - a fence per k-step and `sched_group_barrier`s pin that interleave in the emitted stream (11.5 k cycles per tile);
//   - within a k-step the MFMAs run co-block outer, px-block inner (1 % faster than the other order under the power cap);
// Tried and dropped for tiles the plane only partly covers (the bottom tile row of a 1080-row frame has 8 valid rows, of a
// 220-row ncnn tile 12; rows interleaved over the waves so that every wave would save the same): rows below the plane without
// LDS reads and MFMAs, an empty right px-block skipped.  Chosen per row inside the one tile body it cost 3 % on every tile;
// as a second copy of the tile body behind one branch per tile it cost 0.9 % on whole frames and gained 1.0 % with the
// 200-pixel tiling (one tile in seven is partial there): not kept.
#include <algorithm>
#include <type_traits>

#include "kernels_dev.h"

namespace reve {

#ifndef STORE2_AUX
#define STORE2_AUX 0
#endif
// The timing-only instrumentation this kernel carried in rounds 2 and 3 (STAMPS, ABL2_*: in-kernel clocks, what a launch costs
// without its stores / epilogue / DMA / LDS reads / MFMAs, the best cases of L2-resident fusion and of gap-free launches) was
// removed in round 4 — the tables it produced are profiles/r02/ablation_table_body.txt and docs/LAB_NOTES.md, the code is in the
// history.  The guard stays: an old command line with such a switch stops instead of building a kernel without it.
#if defined(STAMPS) || defined(ABL2_ITEMS_MUL) || defined(ABL2_L2RES) || defined(ABL2_NO_LDS) || defined(ABL2_DOUBLE_LDS) || defined(ABL2_HALF_LDS) || \
    defined(ABL2_HALF_DMA) || defined(ABL2_NO_DMA) || defined(ABL2_NO_STORE) || defined(ABL2_NO_EPI) || defined(ABL2_HALF_STORES) || defined(ABL2_NO_MFMA)
#error "STAMPS / ABL2_* were timing-only diagnostic switches of k_body; they no longer exist (see profiles/r02, docs/LAB_NOTES.md)"
#endif
#ifndef VALU_PER_MFMA
#define VALU_PER_MFMA 3     // epilogue VALU slots behind each MFMA of a body row (sched_group_barrier)
#endif
#ifndef STORE_LAG
#define STORE_LAG 2         // k-steps between a body epilogue piece and its store
#endif
#ifndef B_AHEAD
#define B_AHEAD 1           // k-steps between a B fragment's ds_read and its MFMAs (register buffers: B_AHEAD + 1)
#endif
#ifndef KB_LAST_DMA_AUX
#define KB_LAST_DMA_AUX 0       // cache policy of conv_last's LDS-DMA loads (the activation's last use: 2 = nt)
#endif
#ifndef KB_LAST2_SINGLE
// x2 conv_last as two single-buffered workgroups per CU instead of the body kernel's one double-buffered
// workgroup: built, parity-green, and 10 % SLOWER in the same process (74.9 against 67.6 us, profiles/r03/ab_conv_last_occupancy_and_shared_rows.txt):
// two workgroups that each stop at two barriers per tile lose more than the second tile in flight gains.  Left behind this switch.
#define KB_LAST2_SINGLE 0
#endif
#ifndef MFMA_ORDER
#define MFMA_ORDER 1      // 1: co-block outer, px-block inner (shipped); 0: px-block outer (B constant over 4 MFMAs)
#endif


namespace {
constexpr int KB_NW = 4;                                   // waves per workgroup
constexpr int KB_ROWS = TILE_H / KB_NW;                    // tile rows per wave = rows per tile iteration
constexpr int KB_PER_WAVE = (DMA_PIECES + KB_NW - 1) / KB_NW;
constexpr int KB_STEPS = KB_ROWS * KSTEPS;                 // flat k-steps per tile (72)
// next tile's DMA pieces: seven per row at the even k-steps 0..12 of rows 0 and 1, six in row 2 (the epilogue pieces and
// their stores sit on odd k-steps, so a k-step never carries two vector-memory instructions)
#ifndef DMA_PER_ROW
#define DMA_PER_ROW 7
#endif
constexpr int dma_step(int k) { return (k / DMA_PER_ROW) * KSTEPS + 2 * (k % DMA_PER_ROW); }
constexpr int KB_DMA_LAST = dma_step(KB_PER_WAVE - 1);     // flat step 46: row 2, k-step 10
static_assert(KB_PER_WAVE <= 21 && dma_step(KB_PER_WAVE - 1) < 3 * KSTEPS, "the DMA schedule must end inside row 2");
// Epilogue pieces of the previous row.  Body layer: piece p = 2*q + hh (px-block x channel half), VALU at k-step 1 + 4p, its
// 16-byte store at k-step 3 + 4p.  conv_last x4 (LAST): piece p = 3*q + m (px-block x co-block), VALU at k-step 6 + 2p (the
// residual pixels of all four rows are loaded right after the tile's barrier, BEFORE its first DMA piece: vmcnt retires
// in order, so a load issued between DMA pieces would make its consumer wait for every older piece — 1.2 k cycles per row
// when it was done that way); the three co-blocks of a px-block are the 12 contiguous bytes (4 sub-pixels x
// RGB) of one output sub-row, stored as ONE 12-byte store per lane at k-step 7 + 2p of the px-block's last piece (-1: none).
// conv_last x2 (LAST == 2): one co-block, piece p = q (px-block) at k-step 5 + 8p, stored at 7 + 8p.
// conv_last x3 (LAST == 3): two co-blocks, piece p = 2*q + m at k-step 3 + 4p, a px-block stored at k-step 5 + 4p of its second piece.
constexpr int n_pieces(int last) { return last == 4 ? 6 : (last == 2 ? 2 : 4); }   // (x3: 4 = px-block x co-block)
constexpr int epi_ks(int last, int p) { return last == 4 ? 6 + 2 * p : (last == 2 ? 5 + 8 * p : (last == 3 ? 3 + 4 * p : 1 + 4 * p)); }
constexpr int store_ks(int last, int p) { return last == 4 ? (p % 3 == 2 ? 7 + 2 * p : -1) : (last == 2 ? 7 + 8 * p : (last == 3 ? (p % 2 == 1 ? 5 + 4 * p : -1) : 1 + STORE_LAG + 4 * p)); }
// vector-memory instructions issued after the last DMA piece of a tile (they stay in flight across the barrier: counted vmcnt)
// (x2's px-block store is two instructions: a dword for the even sub-row groups, a short for the odd ones; x3's is four:
// 8 bytes for lane groups 0..2 and three single bytes for group 3)
constexpr int vmem_after_last_dma(int last)
{
    int n = 0;
    for (int si = 0; si < KB_ROWS; ++si) {
        for (int p = 0; p < n_pieces(last); ++p)
            if (store_ks(last, p) >= 0 && si * KSTEPS + store_ks(last, p) > KB_DMA_LAST) n += last == 2 ? 2 : (last == 3 ? 4 : 1);
    }
    return n;
}
}  // namespace

// LAST = 0: a body layer (64 -> 64, PReLU, fp16 to the other arena).  LAST = 4 / 2: conv_last of the x4 / x2 graph on the same
// pipeline, PixelShuffle + nearest residual + post-process in the epilogue pieces, channels in pack_last()'s store order:
//   x4: 48 channels = 3 co-blocks, row 4g + r of co-block m is byte 4m + r of the 12-byte run (4 sub-pixels x RGB) an LR pixel
//       contributes to output sub-row g: one 12-byte store per lane and px-block;
//   x2: 12 channels = 1 co-block, an LR pixel contributes 6 bytes (2 sub-pixels x RGB) to each of its 2 output sub-rows:
//       lane group g = 2i holds bytes 0..3 of sub-row i (a dword store), g = 2i + 1 bytes 4, 5 (a short store);
//   x3: 27 channels = 2 co-blocks, 9 bytes per sub-row: lane groups 0..2 hold bytes 0..7 of sub-row g (one 8-byte store),
//       group 3 holds byte 8 of the three sub-rows in rows 0..2 of co-block 0 (three byte stores).
template <int ORDER, int LAST, bool UNIT_SLOPES>
__global__ void __launch_bounds__(64 * KB_NW, (KB_LAST2_SINGLE && LAST == 2 && !UNIT_SLOPES) ? 2 : 1) k_body(const ConvArgs a, const PlaneDesc* __restrict__ planes,
                                                         const uint32_t* __restrict__ items)
{
    constexpr int NCOB = LAST == 4 ? 3 : (LAST == 3 ? 2 : (LAST == 2 ? 1 : 4));   // co-blocks computed
    constexpr int NPACK = LAST == 2 ? 1 : (LAST == 3 ? 2 : 4);                    // co-blocks per k-step in a.wpack (conv_last x4: the fourth is all zero)
    constexpr int SC = LAST ? LAST : 1;                         // upscale factor of conv_last
    constexpr int NPIECE = n_pieces(LAST);
    // conv_last only (UNIT_SLOPES has no meaning there): the parity probe of reve_debug_run_layers — the fp16 conv_last output
    // BEFORE PixelShuffle / residual / quantisation goes to a.dst as [pixel][16 * NCOB channels in store order] fp16, nothing
    // else is written.  Its own instantiations: the product kernels carry none of it.
    constexpr bool PROBE = LAST != 0 && UNIT_SLOPES;
    // conv_last of the x2 graph (one co-block: a quarter of a body layer's MFMAs per tile, 72 weight registers) is bound by the
    // un-hidden latency of its tile's LDS-DMA: TWO workgroups per CU, each with ONE tile buffer (2 x 78,848 B of LDS, 256
    // registers per wave), so that one workgroup's DMA lands under the other's MFMAs.  A workgroup then has no DMA to hide
    // itself: the next tile's pieces are issued after a second barrier at the end of the tile, into the buffer just read.
    constexpr bool SINGLE = KB_LAST2_SINGLE && LAST == 2 && !UNIT_SLOPES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = wave;                     // tile rows of this wave: row0 + 4 * si (interleaved over the waves)
    auto piece = [&](int k) { const int c = k * KB_NW + wave; return c < DMA_PIECES ? c : DMA_PIECES - 1; };
    const int pl = lane & 15, g = lane >> 4;

    // ---- weights: each wave DMAs a quarter of the 72 fragments into the second tile buffer (idle until the first
    // iteration issues the second tile's DMA), every wave then reads all of them into its registers
    constexpr int NFRAG = KSTEPS * NPACK;
    constexpr bool W_VIA_LDS = NFRAG % KB_NW == 0;       // x2's 18 fragments come straight from global memory
    static_assert(NFRAG <= DMA_PIECES, "the packed weights must fit one tile buffer");
    if constexpr (W_VIA_LDS) {
        auto wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack, 0, NFRAG * 1024, 0x00020000);
#pragma unroll
        for (int f = 0; f < NFRAG / KB_NW; ++f)
            dma16(wrsrc, to_lds(smem + LDS_BUF_BYTES + (f * KB_NW + wave) * 1024), lane * 16, (f * KB_NW + wave) * 1024);
    }
    h8 wf[KSTEPS][NCOB];
    float bias[NCOB][4];
#pragma unroll
    for (int m = 0; m < NCOB; ++m) {
        const h4 b = *(const h4*)(a.bias + 16 * m + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[m][r] = (float)b[r];
    }
    h8 slope8[2];                  // slopes of this lane's 8 channels per 32-channel half, in store order [m][r]
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        if constexpr (LAST) {
            slope8[hh] = (h8)(_Float16)0;
        } else {
            const h4 s0 = *(const h4*)(a.slope + 32 * hh + 4 * g), s1 = *(const h4*)(a.slope + 32 * hh + 16 + 4 * g);
            slope8[hh] = __builtin_shufflevector(s0, s1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
    // conv_last: the u8 source frame (residual) and the u8 destination frame
    auto srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, LAST ? (((int)(a.src_stride * a.frame_h) + 3) & ~3) : 0, 0x00020000);
    auto drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, PROBE ? a.frame_w * a.frame_h * NCOB * 32 : (LAST ? (int)(a.dst_stride * a.frame_h * SC) : 0), 0x00020000);
    // (probe) co-block m of the px-block whose first pixel is plane pixel (oy, ox): this lane's four channels as fp16
    auto probe_store = [&](const f4& ac, int oy, int ox, int w, int h, int m) {
        const bool ok = oy < h && ox < w;
        const h4 v = {(_Float16)ac[0], (_Float16)ac[1], (_Float16)ac[2], (_Float16)ac[3]};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), drsrc, ok ? ((oy * a.frame_w + ox) * NCOB * 16 + 16 * m + 4 * g) * 2 : 0x7fffffff, 0, 0);
    };
    // lane-constant LDS read offsets [dx][half]; rows and the px-block are instruction immediates
    int roff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            roff[dx][hf] = (row0 * LDS_W + pl + dx) * PIX_BYTES + 16 * ((4 * hf + g) ^ ((pl + dx) & 6));
    // lane-constant DMA source offsets (relative to the tile's first input pixel)
    int voff[KB_PER_WAVE];
#pragma unroll
    for (int k = 0; k < KB_PER_WAVE; ++k) {
        int q = piece(k) * 8 + (lane >> 3);
        q = q < LDS_PIX ? q : LDS_PIX - 1;
        const int yy = q / LDS_W, xx = q - yy * LDS_W;
        voff[k] = (yy * a.Wp + xx) * PIX_BYTES + 16 * ((lane & 7) ^ (xx & 6));
    }
    // lane-constant part of a store offset: pixel (1 + row0, 1 + pl) of the arena, this lane's 16-byte chunk
    const int soff_lane = ((1 + row0) * a.Wp + 1 + pl) * PIX_BYTES + 16 * g;

    const int G = gridDim.x;
    const int b = blockIdx.x;
    const int first = ((G & 7) == 0) ? (b & 7) * (G >> 3) + (b >> 3) : b;
    int it = first;
    int cur = 0;
    const int n_items = a.n_items;
    auto item_at = [&](int i) {
        i = i < n_items ? i : it;
        if (a.reverse) i = a.n_items - 1 - i;
        if constexpr (ORDER == 0) {
            const uint32_t v = items[i];
            return Item{(int)(v >> 20), (int)((v >> 10) & 1023u), (int)(v & 1023u)};
        } else if constexpr (ORDER == 1) {
            return decode_blocked(i, a.tiles_x, a.tiles_y);
        } else {
            const int per = a.tiles_x * a.tiles_y;
            Item r;
            r.plane = i / per;
            const int rem = i - r.plane * per;
            r.ty = rem / a.tiles_x;
            r.tx = rem - r.ty * a.tiles_x;
            return r;
        }
    };
    Item itm = item_at(it), nitm = item_at(it + G);
    PlaneDesc pd = planes[itm.plane], npd = planes[nitm.plane];
    if (it < n_items) {
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + pd.base),
                                                      0, (int)pd.span, 0x00020000);
        const int org = ((itm.ty * TILE_H) * a.Wp + itm.tx * TILE_W) * PIX_BYTES;
#pragma unroll
        for (int k = 0; k < KB_PER_WAVE; ++k) dma16a<LAST ? KB_LAST_DMA_AUX : DMA_AUX>(rsrc, to_lds(smem + piece(k) * 1024), voff[k], org);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < NCOB; ++m) {
            if constexpr (W_VIA_LDS) wf[s][m] = *(const h8*)(smem + LDS_BUF_BYTES + (s * NPACK + m) * 1024 + lane * 16);
            else wf[s][m] = ((const h8*)a.wpack)[(s * NPACK + m) * 64 + lane];
        }
    // the weights' wait is pinned here (left alone hipcc waits at each fragment's first use inside the loop, where it
    // would drain the next tile's DMA); 256 of the 288 registers are parked in the accumulator file, the MFMA reads
    // its A operand from there (-mllvm -amdgpu-mfma-vgpr-form=1)
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < NCOB; ++m) {
            if (s * NCOB + m < 64) asm volatile("" : "+a"(wf[s][m]));
            else asm volatile("" : "+v"(wf[s][m]));
        }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // ---- the row carried over from the previous tile of this workgroup: accumulators + where they go
    f4 pacc[NCOB][2];
#pragma unroll
    for (int m = 0; m < NCOB; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q) pacc[m][q] = (f4){0.f, 0.f, 0.f, 0.f};
    int p_soff = 0, p_w = 0, p_h = 0, p_ox = 0, p_oy = 0;     // tile part of the store offset, plane size, first pixel
    int p_x0 = 0, p_y0 = 0;                                   // (conv_last) frame coordinates of the carried row's plane
    unsigned p_resid[2] = {0u, 0u};                           // (conv_last) its residual pixels
    __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0, 0x00020000);   // 0 bytes: every store dropped

    // one epilogue piece: px-block q, channel half hh of a row's accumulators -> 16 bytes per lane
    auto epi = [&](const f4 (&ac)[NCOB][2], int q, int hh) {
        h8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[r] = (_Float16)ac[2 * hh][q][r];
            o[4 + r] = (_Float16)ac[2 * hh + 1][q][r];
        }
        return __builtin_bit_cast(u32x4, UNIT_SLOPES ? prelu8_unit_slopes(o, slope8[hh]) : prelu8(o, slope8[hh]));
    };

    // ---- conv_last epilogue: PixelShuffle(s) + nearest-upsampled input + post-process clamp(v * 255 + 0.5) -> u8
    // residual pixel (RGB in one dword) of px-block q of the row at plane pixel (oy, ox): frame coordinates clamped (plane
    // pixels outside the frame replicate its border); at the very end of the frame buffer the load is moved back inside it
    auto fetch_resid = [&](int oy, int ox, int x0, int y0) -> unsigned {
        int fy = y0 + oy, fx = x0 + ox;
        fy = fy < 0 ? 0 : (fy >= a.frame_h ? a.frame_h - 1 : fy);
        fx = fx < 0 ? 0 : (fx >= a.frame_w ? a.frame_w - 1 : fx);
        const int off = fy * (int)a.src_stride + fx * 3;
        const int lim = (int)(a.src_stride * a.frame_h) - 4;
        const int o4 = off < lim ? off : (lim > 0 ? lim : 0);
        return __builtin_amdgcn_raw_buffer_load_b32(srsrc, o4, 0, 0) >> (8 * (off - o4));
    };
    // one piece: co-block m of px-block q -> the lane's 4 bytes (x4: sub-pixels x colours 4m .. 4m+3 of output sub-row g;
    // x2: bytes 4*(g & 1) + r of the 6-byte run of sub-row g >> 1, i.e. colour (r + (g & 1)) % 3)
    auto epi_last = [&](const f4& ac, unsigned rb, int m) -> unsigned {
        unsigned word = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // x3, lane group 3: rows 0..2 of co-block 0 are byte 8 (colour 2) of the three sub-rows
            const int c = LAST == 2 ? (r + (g & 1)) % 3 : ((LAST == 3 && g == 3) ? 2 : (4 * m + r) % 3);
            const float res = (float)(_Float16)((float)((rb >> (8 * c)) & 0xffu) * (1.0f / 255.0f));
            const float v = (float)(_Float16)ac[r];
            const float o = (float)(_Float16)(v + res);
            word = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(o * 255.0f + 0.5f), r, word);
        }
        return word;
    };

    // x3: a px-block's bytes of one lane: 8 bytes of sub-row g (lane groups 0..2), or the ninth byte of the three sub-rows
    auto store_x3 = [&](unsigned w0, unsigned w1, int off, int off3) {
        __builtin_amdgcn_raw_buffer_store_b64((u32x2){w0, w1}, drsrc, off, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(w0 >> (8 * i)), drsrc, off3 == 0x7fffffff ? off3 : off3 + i * (int)a.dst_stride, 0, 0);
    };

    while (it < n_items) {
        __builtin_amdgcn_s_barrier();      // every wave's DMA share of this tile has landed, every wave is done with the other buffer
        asm volatile("" ::: "memory");
        const int nxt = it + G;
        const Item nnitm = item_at(nxt + G);
        const PlaneDesc nnpd = planes[nnitm.plane];
        auto nrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + npd.base),
                                                       0, (int)npd.span, 0x00020000);
        const int norg = ((nitm.ty * TILE_H) * a.Wp + nitm.tx * TILE_W) * PIX_BYTES;
        char* nbuf = smem + (SINGLE ? 0 : (cur ^ 1) * LDS_BUF_BYTES);
        const char* tbuf = smem + (SINGLE ? 0 : cur * LDS_BUF_BYTES);
        auto orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + pd.base),
                                                       0, (int)pd.span, 0x00020000);
        const int t_soff = ((itm.ty * TILE_H) * a.Wp + itm.tx * TILE_W) * PIX_BYTES;
        const int t_oy = itm.ty * TILE_H + row0, t_ox = itm.tx * TILE_W + pl;   // first pixel of this lane: its rows are t_oy + 4 * si

        {
            // B fragments, double-buffered: Bb[F % (B_AHEAD + 1)][q] feeds flat step F
            h8 Bb[B_AHEAD + 1][2];
            auto load_b = [&](int F, int q) {
                const int si = F / KSTEPS, ks = F - si * KSTEPS, t = ks >> 1, hf = ks & 1, dy = t / 3, dx = t - 3 * dy;
                return *(const h8*)(tbuf + roff[dx][hf] + ((4 * si + dy) * LDS_W + 16 * q) * PIX_BYTES);
            };
#pragma unroll
            for (int f = 0; f < B_AHEAD; ++f) {
                Bb[f][0] = load_b(f, 0);
                Bb[f][1] = load_b(f, 1);
            }

            unsigned resid_all[KB_ROWS][2];      // conv_last: residual pixels (RGB in a dword) of this tile's rows, fetched ahead of its DMA
#pragma unroll
            for (int r = 0; r < KB_ROWS; ++r)
#pragma unroll
                for (int q = 0; q < 2; ++q) resid_all[r][q] = LAST ? fetch_resid(t_oy + 4 * r, t_ox + 16 * q, pd.x0, pd.y0) : 0u;
            f4 racc[NCOB][2];                    // the row whose epilogue is in progress
#pragma unroll
            for (int m = 0; m < NCOB; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q) racc[m][q] = pacc[m][q];
            u32x4 pend = (u32x4){0u, 0u, 0u, 0u};   // an epilogue piece between its VALU and its store
            int pend_off = 0x7fffffff;
            int pend_off3 = 0x7fffffff;             // x3: where lane group 3's byte of sub-row 0 goes

            // One row = 18 k-steps x 8 MFMAs, straight-line code
            auto row = [&](auto si_c) __attribute__((always_inline)) {
                constexpr int si = decltype(si_c)::value;
                f4 acc[NCOB][2];
#pragma unroll
                for (int m = 0; m < NCOB; ++m)
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc[m][q] = (f4){bias[m][0], bias[m][1], bias[m][2], bias[m][3]};
                // where the row in `racc` goes: the carried row (previous tile, last row of this wave) or row si-1 of this tile
                const int e_soff = si == 0 ? p_soff : t_soff + 4 * (si - 1) * a.Wp * PIX_BYTES;
                const int e_oy = si == 0 ? p_oy : t_oy + 4 * (si - 1), e_ox = si == 0 ? p_ox : t_ox;
                const int e_w = si == 0 ? p_w : pd.w, e_h = si == 0 ? p_h : pd.h;
                const int e_x0 = si == 0 ? p_x0 : pd.x0, e_y0 = si == 0 ? p_y0 : pd.y0;   // conv_last: plane origin in the frame
                // conv_last: residual pixels of the row in `racc`
                const unsigned resid[2] = {si == 0 ? p_resid[0] : resid_all[si > 0 ? si - 1 : 0][0], si == 0 ? p_resid[1] : resid_all[si > 0 ? si - 1 : 0][1]};
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    const int F = si * KSTEPS + ks;
                    if (F + B_AHEAD < KB_STEPS) {            // the reads of a later k-step
                        Bb[(F + B_AHEAD) % (B_AHEAD + 1)][0] = load_b(F + B_AHEAD, 0);
                        Bb[(F + B_AHEAD) % (B_AHEAD + 1)][1] = load_b(F + B_AHEAD, 1);
                    }
#pragma unroll
                    for (int k = 0; k < KB_PER_WAVE; ++k)
                        if (!SINGLE && dma_step(k) == F) {
                            dma16a<LAST ? KB_LAST_DMA_AUX : DMA_AUX>(nrsrc, to_lds(nbuf + piece(k) * 1024), voff[k], norg);
                        }
#pragma unroll
                    for (int p = 0; p < NPIECE; ++p)
                        if (ks == store_ks(LAST, p)) {
                            if constexpr (PROBE) {
                                // (stored with the epilogue piece itself)
                            } else if constexpr (LAST == 4) {
                                __builtin_amdgcn_raw_buffer_store_b96((u32x3){pend[0], pend[1], pend[2]}, drsrc, pend_off, 0, 0);
                            } else if constexpr (LAST == 2) {
                                __builtin_amdgcn_raw_buffer_store_b32(pend[0], drsrc, (g & 1) ? 0x7fffffff : pend_off, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)pend[0], drsrc, (g & 1) ? pend_off : 0x7fffffff, 0, 0);
                            } else if constexpr (LAST == 3) {
                                store_x3(pend[0], pend[1], pend_off, pend_off3);
                            }
                            else if (si == 0) __builtin_amdgcn_raw_buffer_store_b128(pend, p_rsrc, pend_off, 0, STORE2_AUX);
                            else __builtin_amdgcn_raw_buffer_store_b128(pend, orsrc, pend_off, 0, STORE2_AUX);
                        }
                    // The emitted order of a k-step: the LDS reads of the NEXT k-step and this one's vector-memory instruction
                    // above this fence, the MFMAs with the epilogue piece's VALU in their shadows below it.  (Left to itself
                    // hipcc sinks the reads to just above their first use, where their latency is exposed; scheduling groups
                    // for the reads did not hold them either.)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < NPIECE; ++p)
                        if (ks == epi_ks(LAST, p)) {
                            if constexpr (PROBE) {
                                const int q = LAST == 4 ? p / 3 : (LAST == 3 ? p >> 1 : p), m = LAST == 4 ? p % 3 : (LAST == 3 ? p & 1 : 0);
                                probe_store(racc[m][q], e_oy, e_ox + 16 * q, e_w, e_h, m);
                            } else if constexpr (LAST == 4) {
                                const int q = p / 3, m = p % 3;
                                pend[m] = epi_last(racc[m][q], resid[q], m);
                                if (m == 2) {
                                    // cropped to the un-padded part of the plane (ncnn-compat tiles carry an apron of a.pad px)
                                    const int oy = e_oy, ox = e_ox + 16 * q;
                                    const bool ok = oy >= a.pad && oy < e_h - a.pad && ox >= a.pad && ox < e_w - a.pad;
                                    pend_off = ok ? ((e_y0 + oy) * 4 + g) * (int)a.dst_stride + (e_x0 + ox) * 12 : 0x7fffffff;
                                }
                            } else if constexpr (LAST == 3) {
                                const int q = p >> 1, m = p & 1;
                                pend[m] = epi_last(racc[m][q], resid[q], m);
                                if (m == 1) {
                                    const int oy = e_oy, ox = e_ox + 16 * q;
                                    const bool ok = oy >= a.pad && oy < e_h - a.pad && ox >= a.pad && ox < e_w - a.pad;
                                    const int base = (e_y0 + oy) * 3 * (int)a.dst_stride + (e_x0 + ox) * 9;
                                    pend_off = ok && g < 3 ? base + g * (int)a.dst_stride : 0x7fffffff;     // bytes 0..7 of sub-row g
                                    pend_off3 = ok && g == 3 ? base + 8 : 0x7fffffff;                       // byte 8 of sub-rows 0..2
                                }
                            } else if constexpr (LAST == 2) {
                                const int q = p;
                                pend[0] = epi_last(racc[0][q], resid[q], 0);
                                const int oy = e_oy, ox = e_ox + 16 * q;
                                const bool ok = oy >= a.pad && oy < e_h - a.pad && ox >= a.pad && ox < e_w - a.pad;
                                pend_off = ok ? ((e_y0 + oy) * 2 + (g >> 1)) * (int)a.dst_stride + (e_x0 + ox) * 6 + 4 * (g & 1) : 0x7fffffff;
                            } else {
                                const int q = p >> 1, hh = p & 1;
                                pend = epi(racc, q, hh);
                                const bool ok = e_oy < e_h && e_ox + 16 * q < e_w;
                                pend_off = ok ? e_soff + soff_lane + (16 * q) * PIX_BYTES + 64 * hh : 0x7fffffff;
                            }
                        }
                    {
                        constexpr int NQ = 2;
                        if constexpr (MFMA_ORDER == 0) {
#pragma unroll
                            for (int q = 0; q < NQ; ++q)
#pragma unroll
                                for (int m = 0; m < NCOB; ++m) acc[m][q] = MFMA16(wf[ks][m], Bb[F % (B_AHEAD + 1)][q], acc[m][q]);
                        } else {
#pragma unroll
                            for (int m = 0; m < NCOB; ++m)
#pragma unroll
                                for (int q = 0; q < NQ; ++q) acc[m][q] = MFMA16(wf[ks][m], Bb[F % (B_AHEAD + 1)][q], acc[m][q]);
                        }
#pragma unroll
                        for (int j = 0; j < NCOB * NQ; ++j) {
                            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x2, LAST == 4 ? 5 : (LAST == 2 ? 12 : (LAST == 3 ? 6 : VALU_PER_MFMA)), 0);
                        }
                    }
                    // no store of a later k-step may move above the last DMA issue: the counted vmcnt at the end of the tile
                    // relies on at least stores_after_last_dma() vector-memory instructions being younger than every DMA
                    if (F == KB_DMA_LAST) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 0; m < NCOB; ++m)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        asm volatile("" : "+v"(acc[m][q]));   // accumulators in VGPRs: no v_accvgpr_read in the epilogue
                        racc[m][q] = acc[m][q];
                    }
            };
            // (written out row by row: hipcc peels the first iteration off a `#pragma unroll` loop here and leaves the rest rolled)
            static_assert(KB_ROWS == 4, "four rows per wave are written out");
            row(std::integral_constant<int, 0>{});
            row(std::integral_constant<int, 1>{});
            row(std::integral_constant<int, 2>{});
            row(std::integral_constant<int, 3>{});
            // carry the tile's last row into the next iteration
#pragma unroll
            for (int m = 0; m < NCOB; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q) pacc[m][q] = racc[m][q];
            p_resid[0] = resid_all[KB_ROWS - 1][0];
            p_resid[1] = resid_all[KB_ROWS - 1][1];
        }
        if constexpr (SINGLE) {
            // every wave is done reading the buffer: the next tile goes into it (the other workgroup of this CU computes meanwhile)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < KB_PER_WAVE; ++k) dma16a<LAST ? KB_LAST_DMA_AUX : DMA_AUX>(nrsrc, to_lds(nbuf + piece(k) * 1024), voff[k], norg);
        }
        // this wave's pieces of the next tile have landed; the stores issued after the last DMA stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PROBE || SINGLE) ? 0 : vmem_after_last_dma(LAST)) : "memory");
        p_soff = t_soff + 4 * (KB_ROWS - 1) * a.Wp * PIX_BYTES;
        p_oy = t_oy + 4 * (KB_ROWS - 1); p_ox = t_ox; p_w = pd.w; p_h = pd.h;
        p_x0 = pd.x0; p_y0 = pd.y0;
        p_rsrc = orsrc;
        cur ^= 1;
        it = nxt;
        itm = nitm; pd = npd;
        nitm = nnitm; npd = nnpd;
    }
    // the last tile's last row
    if constexpr (PROBE) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int m = 0; m < NCOB; ++m) probe_store(pacc[m][q], p_oy, p_ox + 16 * q, p_w, p_h, m);
    } else if constexpr (LAST == 3) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int oy = p_oy, ox = p_ox + 16 * q;
            const bool ok = oy >= a.pad && oy < p_h - a.pad && ox >= a.pad && ox < p_w - a.pad;
            const int base = (p_y0 + oy) * 3 * (int)a.dst_stride + (p_x0 + ox) * 9;
            store_x3(epi_last(pacc[0][q], p_resid[q], 0), epi_last(pacc[1][q], p_resid[q], 1), ok && g < 3 ? base + g * (int)a.dst_stride : 0x7fffffff,
                     ok && g == 3 ? base + 8 : 0x7fffffff);
        }
    } else if constexpr (LAST == 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int oy = p_oy, ox = p_ox + 16 * q;
            const bool ok = oy >= a.pad && oy < p_h - a.pad && ox >= a.pad && ox < p_w - a.pad;
            const int off = ok ? ((p_y0 + oy) * 2 + (g >> 1)) * (int)a.dst_stride + (p_x0 + ox) * 6 + 4 * (g & 1) : 0x7fffffff;
            const unsigned word = epi_last(pacc[0][q], p_resid[q], 0);
            __builtin_amdgcn_raw_buffer_store_b32(word, drsrc, (g & 1) ? 0x7fffffff : off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)word, drsrc, (g & 1) ? off : 0x7fffffff, 0, 0);
        }
    } else if constexpr (LAST == 4) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned rb = p_resid[q];
            const int oy = p_oy, ox = p_ox + 16 * q;
            const bool ok = oy >= a.pad && oy < p_h - a.pad && ox >= a.pad && ox < p_w - a.pad;
            __builtin_amdgcn_raw_buffer_store_b96((u32x3){epi_last(pacc[0][q], rb, 0), epi_last(pacc[1][q], rb, 1), epi_last(pacc[2][q], rb, 2)}, drsrc,
                                                  ok ? ((p_y0 + oy) * 4 + g) * (int)a.dst_stride + (p_x0 + ox) * 12 : 0x7fffffff, 0, 0);
        }
    } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int q = p >> 1, hh = p & 1;
            const bool ok = p_oy < p_h && p_ox + 16 * q < p_w;
            __builtin_amdgcn_raw_buffer_store_b128(epi(pacc, q, hh), p_rsrc, ok ? p_soff + soff_lane + (16 * q) * PIX_BYTES + 64 * hh : 0x7fffffff, 0, STORE2_AUX);
        }
    }
}

#define KB_INST(O, L, U) template __global__ void k_body<O, L, U>(const ConvArgs, const PlaneDesc* __restrict__, const uint32_t* __restrict__);
KB_INST(0, 0, false) KB_INST(1, 0, false) KB_INST(2, 0, false) KB_INST(0, 0, true) KB_INST(1, 0, true) KB_INST(2, 0, true)
KB_INST(0, 2, false) KB_INST(1, 2, false) KB_INST(2, 2, false) KB_INST(0, 3, false) KB_INST(1, 3, false) KB_INST(2, 3, false)
KB_INST(0, 4, false) KB_INST(1, 4, false) KB_INST(2, 4, false)
KB_INST(2, 2, true) KB_INST(2, 3, true) KB_INST(2, 4, true)      // conv_last parity probes (plain tile order)
#undef KB_INST


int conv_lds_bytes() { return 2 * LDS_BUF_BYTES; }

// host-side evaluation of the computed work order (tests: must equal Engine::configure()'s list order)
void debug_blocked_order(int tiles_x, int tiles_y, uint32_t* out)
{
    for (int it = 0; it < tiles_x * tiles_y; ++it) {
        const Item i = decode_blocked(it, tiles_x, tiles_y);
        out[it] = (uint32_t)i.tx | ((uint32_t)i.ty << 10);
    }
}

// Function attributes belong to the CURRENT device: Engine::init calls the prepare_* functions once per
// context after hipSetDevice (a process-wide "once" would leave the second GPU of a group without them).
int prepare_body_kernels()
{
    int rc = 0;
    for (const void* f : {(const void*)k_body<0, 0, false>, (const void*)k_body<1, 0, false>, (const void*)k_body<2, 0, false>,
                          (const void*)k_body<0, 0, true>, (const void*)k_body<1, 0, true>, (const void*)k_body<2, 0, true>,
                          (const void*)k_body<0, 2, false>, (const void*)k_body<1, 2, false>, (const void*)k_body<2, 2, false>,
                          (const void*)k_body<0, 3, false>, (const void*)k_body<1, 3, false>, (const void*)k_body<2, 3, false>,
                          (const void*)k_body<0, 4, false>, (const void*)k_body<1, 4, false>, (const void*)k_body<2, 4, false>,
                          (const void*)k_body<2, 2, true>, (const void*)k_body<2, 3, true>, (const void*)k_body<2, 4, true>})
        rc |= (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF_BYTES);
    return rc;      // (the single-buffered x2 conv_last asks for LDS_BUF_BYTES at launch: below every default limit)
}

template <int LAST, bool UNIT_SLOPES>
static int launch_k(const ConvArgs& a, int grid, void* stream)
{
    const size_t lds = (KB_LAST2_SINGLE && LAST == 2 && !UNIT_SLOPES) ? LDS_BUF_BYTES : 2 * LDS_BUF_BYTES;      // x2 conv_last: one tile buffer, two workgroups per CU
    launch_prepare();
    if (a.items) hipLaunchKernelGGL((k_body<0, LAST, UNIT_SLOPES>), dim3(grid), dim3(64 * KB_NW), lds, (hipStream_t)stream, a, a.planes, a.items);
    else if (a.blocked) hipLaunchKernelGGL((k_body<1, LAST, UNIT_SLOPES>), dim3(grid), dim3(64 * KB_NW), lds, (hipStream_t)stream, a, a.planes, a.items);
    else hipLaunchKernelGGL((k_body<2, LAST, UNIT_SLOPES>), dim3(grid), dim3(64 * KB_NW), lds, (hipStream_t)stream, a, a.planes, a.items);
    return launch_status();
}

int launch_body(const ConvArgs& a, int grid, void* stream)
{
    return a.unit_slopes ? launch_k<0, true>(a, grid, stream) : launch_k<0, false>(a, grid, stream);
}
// conv_last (+ PixelShuffle, residual, post-process) of the x2 / x3 / x4 graphs on the body kernel's pipeline
int launch_last(const ConvArgs& a, int scale, int grid, void* stream)
{
    switch (scale) {
    case 2: return launch_k<2, false>(a, KB_LAST2_SINGLE ? std::min(2 * grid, a.n_items) : grid, stream);      // two workgroups per CU
    case 3: return launch_k<3, false>(a, grid, stream);
    case 4: return launch_k<4, false>(a, grid, stream);
    default: return -1;
    }
}

// parity probe: fp16 conv_last output before PixelShuffle / residual / quantisation -> a.dst, [pixel][16 x co-blocks] in
// pack_last()'s store order (whole-frame geometry, one plane)
int launch_last_probe(const ConvArgs& a, int scale, int grid, void* stream)
{
    const size_t lds = 2 * LDS_BUF_BYTES;
    launch_prepare();
    switch (scale) {
    case 2: hipLaunchKernelGGL((k_body<2, 2, true>), dim3(grid), dim3(64 * KB_NW), lds, (hipStream_t)stream, a, a.planes, a.items); break;
    case 3: hipLaunchKernelGGL((k_body<2, 3, true>), dim3(grid), dim3(64 * KB_NW), lds, (hipStream_t)stream, a, a.planes, a.items); break;
    case 4: hipLaunchKernelGGL((k_body<2, 4, true>), dim3(grid), dim3(64 * KB_NW), lds, (hipStream_t)stream, a, a.planes, a.items); break;
    default: return -1;
    }
    return launch_status();
}

}  // namespace reve
