// Hand-written gfx950 (CDNA4 / MI355X) kernels for the realesr-animevideov3 SRVGGNetCompact graph: the body layers.
//
// Replaces the compute that ONdraid/reve reaches by spawning `realesrgan-ncnn-vulkan`
// (reve-shared/src/lib.rs:134-147): ncnn layers Convolution/PReLU x17, Convolution, PixelShuffle,
// Interp(nearest), BinaryOp(add) plus the binary's pre/post-processing.
// conv_first is in kernels_first.hip, conv_last in kernels_last.hip.
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
#include <type_traits>

#include "kernels_dev.h"

namespace reve {

#ifndef STORE2_AUX
#define STORE2_AUX 0
#endif
// ---- timing-only ablation switches (scripts/ablate.sh; outputs are wrong with any of them): what a launch costs without
// its stores / epilogue / next-tile DMA / LDS reads / MFMAs.  Values stay live through empty asm statements so that nothing
// upstream is dead-code-eliminated (cdna_hip_programming.md §5.4 rule 17).
#ifndef MFMA_ORDER
#define MFMA_ORDER 1      // 1: co-block outer, px-block inner (shipped); 0: px-block outer (B constant over 4 MFMAs)
#endif

#ifdef STAMPS
// Diagnostic build only (scripts/stamps.py, scripts/ab_libs.py): per wave {cycles waiting at the tile barrier, cycles in the tile loop,
// s_memrealtime at entry / exit (100 MHz), s_memtime at entry / exit (shader clock)} — the in-kernel clock is
// d(memtime) / d(memrealtime) x 100 MHz.  The values go to a buffer nothing else reads.
__device__ unsigned long long g_stamps2[1024 * 8];
#define ST2_NOW(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#endif

namespace {
constexpr int KB_NW = 4;                                   // waves per workgroup
constexpr int KB_ROWS = TILE_H / KB_NW;                    // tile rows per wave = rows per tile iteration
constexpr int KB_PER_WAVE = (DMA_PIECES + KB_NW - 1) / KB_NW;
constexpr int KB_STEPS = KB_ROWS * KSTEPS;                 // flat k-steps per tile (72)
// next tile's DMA pieces: seven per row at the even k-steps 0..12 of rows 0 and 1, six in row 2 (the epilogue pieces and
// their stores sit on odd k-steps, so a k-step never carries two vector-memory instructions)
constexpr int dma_step(int k) { return (k / 7) * KSTEPS + 2 * (k % 7); }
constexpr int KB_DMA_LAST = dma_step(KB_PER_WAVE - 1);     // flat step 46: row 2, k-step 10
static_assert(KB_PER_WAVE <= 21 && dma_step(KB_PER_WAVE - 1) < 3 * KSTEPS, "the DMA schedule must end inside row 2");
// epilogue piece p (= 2*q + hh) of the previous row: VALU at k-step 1 + 4p, its store at k-step 3 + 4p
constexpr int epi_ks(int p) { return 1 + 4 * p; }
constexpr int store_ks(int p) { return 3 + 4 * p; }
// stores issued after the last DMA piece of a tile (they stay in flight across the barrier: counted vmcnt)
constexpr int stores_after_last_dma()
{
    int n = 0;
    for (int si = 0; si < KB_ROWS; ++si)
        for (int p = 0; p < 4; ++p)
            if (si * KSTEPS + store_ks(p) > KB_DMA_LAST) ++n;
    return n;
}
}  // namespace

template <int ORDER>
__global__ void __launch_bounds__(64 * KB_NW, 1) k_body(const ConvArgs a, const PlaneDesc* __restrict__ planes,
                                                         const uint32_t* __restrict__ items)
{
    constexpr int NCOB = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef STAMPS
    unsigned long long st_t0, st_r0, st_bar = 0, st_loop0 = 0, st_a, st_b;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0), "=s"(st_r0)::"memory");
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = wave;                     // tile rows of this wave: row0 + 4 * si (interleaved over the waves)
    auto piece = [&](int k) { const int c = k * KB_NW + wave; return c < DMA_PIECES ? c : DMA_PIECES - 1; };
    const int pl = lane & 15, g = lane >> 4;

    // ---- weights: each wave DMAs a quarter of the 72 fragments into the second tile buffer (idle until the first
    // iteration issues the second tile's DMA), every wave then reads all of them into its registers
    constexpr int NFRAG = KSTEPS * NCOB;
    static_assert(NFRAG <= DMA_PIECES && NFRAG % KB_NW == 0, "the packed weights must fit one tile buffer");
    {
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
        const h4 s0 = *(const h4*)(a.slope + 32 * hh + 4 * g), s1 = *(const h4*)(a.slope + 32 * hh + 16 + 4 * g);
        slope8[hh] = __builtin_shufflevector(s0, s1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
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
    auto item_at = [&](int i) {
        i = i < a.n_items ? i : it;
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
    if (it < a.n_items) {
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (unsigned long long)itm.plane * a.plane_stride),
                                                      0, (int)a.plane_stride, 0x00020000);
        const int org = ((itm.ty * TILE_H) * a.Wp + itm.tx * TILE_W) * PIX_BYTES;
#pragma unroll
        for (int k = 0; k < KB_PER_WAVE; ++k) dma16(rsrc, to_lds(smem + piece(k) * 1024), voff[k], org);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < NCOB; ++m) wf[s][m] = *(const h8*)(smem + LDS_BUF_BYTES + (s * NCOB + m) * 1024 + lane * 16);
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
    __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0, 0x00020000);   // 0 bytes: every store dropped

    // one epilogue piece: px-block q, channel half hh of a row's accumulators -> 16 bytes per lane
    auto epi = [&](const f4 (&ac)[NCOB][2], int q, int hh) {
        h8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[r] = (_Float16)ac[2 * hh][q][r];
            o[4 + r] = (_Float16)ac[2 * hh + 1][q][r];
        }
        return __builtin_bit_cast(u32x4, prelu8(o, slope8[hh]));
    };

#ifdef STAMPS
    ST2_NOW(st_loop0);
#endif
    while (it < a.n_items) {
#ifdef STAMPS
        ST2_NOW(st_a);
#endif
        __builtin_amdgcn_s_barrier();      // every wave's DMA share of this tile has landed, every wave is done with the other buffer
        asm volatile("" ::: "memory");
#ifdef STAMPS
        ST2_NOW(st_b);
        st_bar += st_b - st_a;
#endif
        const int nxt = it + G;
        const Item nnitm = item_at(nxt + G);
        const PlaneDesc nnpd = planes[nnitm.plane];
        auto nrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (unsigned long long)nitm.plane * a.plane_stride),
                                                       0, (int)a.plane_stride, 0x00020000);
        const int norg = ((nitm.ty * TILE_H) * a.Wp + nitm.tx * TILE_W) * PIX_BYTES;
        char* nbuf = smem + (cur ^ 1) * LDS_BUF_BYTES;
        const char* tbuf = smem + cur * LDS_BUF_BYTES;
        auto orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (unsigned long long)itm.plane * a.plane_stride),
                                                       0, (int)a.plane_stride, 0x00020000);
        const int t_soff = ((itm.ty * TILE_H) * a.Wp + itm.tx * TILE_W) * PIX_BYTES;
        const int t_oy = itm.ty * TILE_H + row0, t_ox = itm.tx * TILE_W + pl;   // first pixel of this lane: its rows are t_oy + 4 * si

        {
#ifdef ABL2_NO_LDS
            h8 abl_bconst = __builtin_bit_cast(h8, (u32x4){(unsigned)lane * 2654435761u, (unsigned)lane ^ 0x3c003c00u, 0x3c003800u, 0xbc003c00u});
            asm volatile("" : "+v"(abl_bconst));
#endif
            // B fragments, double-buffered: Bb[F & 1][q] feeds flat step F
            h8 Bb[2][2];
            auto load_b = [&](int F, int q) {
                const int si = F / KSTEPS, ks = F - si * KSTEPS, t = ks >> 1, hf = ks & 1, dy = t / 3, dx = t - 3 * dy;
#ifdef ABL2_NO_LDS
                (void)si; (void)dy; (void)dx; (void)hf;
                return abl_bconst;
#else
                return *(const h8*)(tbuf + roff[dx][hf] + ((4 * si + dy) * LDS_W + 16 * q) * PIX_BYTES);
#endif
            };
            Bb[0][0] = load_b(0, 0);
            Bb[0][1] = load_b(0, 1);

            f4 racc[NCOB][2];                    // the row whose epilogue is in progress
#pragma unroll
            for (int m = 0; m < NCOB; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q) racc[m][q] = pacc[m][q];
            u32x4 pend = (u32x4){0u, 0u, 0u, 0u};   // an epilogue piece between its VALU and its store
            int pend_off = 0x7fffffff;

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
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    const int F = si * KSTEPS + ks;
                    if (F + 1 < KB_STEPS) {                  // the reads of the next k-step
                        Bb[(F + 1) & 1][0] = load_b(F + 1, 0);
                        Bb[(F + 1) & 1][1] = load_b(F + 1, 1);
                    }
#pragma unroll
                    for (int k = 0; k < KB_PER_WAVE; ++k)
                        if (dma_step(k) == F) {
#ifndef ABL2_NO_DMA
                            dma16(nrsrc, to_lds(nbuf + piece(k) * 1024), voff[k], norg);
#endif
                        }
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        if (ks == store_ks(p)) {
#if defined(ABL2_NO_STORE) || defined(ABL2_NO_EPI)
                            asm volatile("" ::"v"(pend), "v"(pend_off));
#else
                            if (si == 0) __builtin_amdgcn_raw_buffer_store_b128(pend, p_rsrc, pend_off, 0, STORE2_AUX);
                            else __builtin_amdgcn_raw_buffer_store_b128(pend, orsrc, pend_off, 0, STORE2_AUX);
#endif
                        }
                    // The emitted order of a k-step: the LDS reads of the NEXT k-step and this one's vector-memory instruction
                    // above this fence, the MFMAs with the epilogue piece's VALU in their shadows below it.  (Left to itself
                    // hipcc sinks the reads to just above their first use, where their latency is exposed; scheduling groups
                    // for the reads did not hold them either.)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        if (ks == epi_ks(p)) {
                            const int q = p >> 1, hh = p & 1;
#ifdef ABL2_NO_EPI
                            asm volatile("" ::"v"(racc[2 * hh][q]), "v"(racc[2 * hh + 1][q]));
                            (void)e_soff; (void)e_oy; (void)e_ox; (void)e_w; (void)e_h;
#else
                            pend = epi(racc, q, hh);
                            const bool ok = e_oy < e_h && e_ox + 16 * q < e_w;
                            pend_off = ok ? e_soff + soff_lane + (16 * q) * PIX_BYTES + 64 * hh : 0x7fffffff;
#endif
                        }
                    {
                        constexpr int NQ = 2;
#ifdef ABL2_NO_MFMA
                        asm volatile("" ::"v"(Bb[F & 1][0]), "v"(Bb[F & 1][1]));
                        if (ks == 0) {
#pragma unroll
                            for (int m = 0; m < NCOB; ++m)
#pragma unroll
                                for (int q = 0; q < NQ; ++q) acc[m][q] = MFMA16(wf[ks][m], Bb[F & 1][q], acc[m][q]);
                        }
#elif MFMA_ORDER == 0
#pragma unroll
                        for (int q = 0; q < NQ; ++q)
#pragma unroll
                            for (int m = 0; m < NCOB; ++m) acc[m][q] = MFMA16(wf[ks][m], Bb[F & 1][q], acc[m][q]);
#else
#pragma unroll
                        for (int m = 0; m < NCOB; ++m)
#pragma unroll
                            for (int q = 0; q < NQ; ++q) acc[m][q] = MFMA16(wf[ks][m], Bb[F & 1][q], acc[m][q]);
#endif
#pragma unroll
                        for (int j = 0; j < 4 * NQ; ++j) {
                            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
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
        }
        // this wave's pieces of the next tile have landed; the stores issued after the last DMA stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(stores_after_last_dma()) : "memory");
        p_soff = t_soff + 4 * (KB_ROWS - 1) * a.Wp * PIX_BYTES;
        p_oy = t_oy + 4 * (KB_ROWS - 1); p_ox = t_ox; p_w = pd.w; p_h = pd.h;
        p_rsrc = orsrc;
        cur ^= 1;
        it = nxt;
        itm = nitm; pd = npd;
        nitm = nnitm; npd = nnpd;
    }
#ifdef STAMPS
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
        unsigned long long* o = g_stamps2 + (blockIdx.x * KB_NW + wave) * 8;
        o[0] = st_bar; o[1] = t1 - st_loop0; o[2] = st_r0; o[3] = r1; o[4] = st_t0; o[5] = t1;
        o[6] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID[3:0]
    }
#endif
    // the last tile's last row
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int q = p >> 1, hh = p & 1;
        const bool ok = p_oy < p_h && p_ox + 16 * q < p_w;
        __builtin_amdgcn_raw_buffer_store_b128(epi(pacc, q, hh), p_rsrc, ok ? p_soff + soff_lane + (16 * q) * PIX_BYTES + 64 * hh : 0x7fffffff, 0, STORE2_AUX);
    }
}

template __global__ void k_body<0>(const ConvArgs, const PlaneDesc* __restrict__, const uint32_t* __restrict__);
template __global__ void k_body<1>(const ConvArgs, const PlaneDesc* __restrict__, const uint32_t* __restrict__);
template __global__ void k_body<2>(const ConvArgs, const PlaneDesc* __restrict__, const uint32_t* __restrict__);

#ifdef STAMPS
extern "C" int reve_debug_read_stamps2(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps2), sizeof(unsigned long long) * n);
}
#endif

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
    return (int)hipFuncSetAttribute((const void*)k_body<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF_BYTES) |
           (int)hipFuncSetAttribute((const void*)k_body<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF_BYTES) |
           (int)hipFuncSetAttribute((const void*)k_body<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF_BYTES);
}

int launch_body(const ConvArgs& a, int grid, void* stream)
{
    const size_t lds = 2 * LDS_BUF_BYTES;
    if (a.items) hipLaunchKernelGGL(k_body<0>, dim3(grid), dim3(64 * KB_NW), lds, (hipStream_t)stream, a, a.planes, a.items);
    else if (a.blocked) hipLaunchKernelGGL(k_body<1>, dim3(grid), dim3(64 * KB_NW), lds, (hipStream_t)stream, a, a.planes, a.items);
    else hipLaunchKernelGGL(k_body<2>, dim3(grid), dim3(64 * KB_NW), lds, (hipStream_t)stream, a, a.planes, a.items);
    return (int)hipGetLastError();
}

}  // namespace reve
