// Hand-written gfx950 (CDNA4 / MI355X) kernels for the realesr-animevideov3 SRVGGNetCompact graph.
//
// Replaces the compute that ONdraid/reve reaches by spawning `realesrgan-ncnn-vulkan`
// (reve-shared/src/lib.rs:134-147): ncnn layers Convolution/PReLU x17, Convolution, PixelShuffle,
// Interp(nearest), BinaryOp(add) plus the binary's pre/post-processing.
//
// Data layout in HBM ("activation arena"): planes of (tiles_y*16+2) x (tiles_x*32+2) pixels,
// 128 B per pixel (64 channels fp16, channel order permuted by chan_phys()), image pixel (0,0)
// at arena pixel (1,1).  Everything outside the image stays ZERO for the life of the arena, so
// the convolutions' zero padding and all halo loads need no bounds checks.
//
// k_body (64->64 + bias + PReLU): implicit GEMM on v_mfma_f32_16x16x32_f16 with
//   A = weights (16 output channels x 32 k), REGISTER-STATIONARY for the whole persistent launch:
//       a workgroup is 4 waves (one per SIMD, 512-register budget); wave w owns rows 4w..4w+3 of
//       the tile and all 64 output channels (18 k-steps x 4 co-blocks x 4 = 288 registers of
//       weights, 256 of them in AGPRs; 64 accumulator registers per sub-iteration);
//   B = pixels  (32 k x 16 pixels), read from an LDS image of the (16+2)x(32+2) input tile with
//       ds_read_b128 (k = 8 consecutive physical channels of one tap), XOR-swizzled per column;
//   the tile image is filled by LDS-DMA (buffer_load_dwordx4 ... lds), double-buffered: the next
//   tile's DMA pieces are issued between the current tile's MFMAs and land under them.
// conv_first is in kernels_first.hip, conv_last in kernels_last.hip.
#include "kernels_dev.h"

namespace reve {

#ifndef STAMPS
#define PSTAMP(i) (void)0
#endif
#ifdef STAMPS
// Diagnostic build only (scripts/stamps.py): per-wave cycle totals of the tile loop's segments.
__device__ unsigned long long g_stamps[2048 * 8];
__device__ unsigned long long g_pro[2048 * 4];
#define STAMP(i)                                                                            \
    do {                                                                                    \
        unsigned long long t_;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        seg_[i] += t_ - last_;                                                              \
        last_ = t_;                                                                         \
    } while (0)
#else
#define STAMP(i) (void)0
#endif

// -------------------------------------------------------------------------------------------
// 64 -> 64 channel 3x3 convolution + bias + fp16 round + PReLU, fp16 store to the other arena.
// 4 waves per workgroup, one per SIMD (512-register budget): wave w owns tile rows 4w..4w+3 (8 px-blocks
// of 16 px) and all 64 output channels (4 co-blocks).
// A wave walks its rows two at a time (sub-iteration = 2 rows x 2 px-blocks x 4 co-blocks = 16
// accumulators): the epilogue of one sub-iteration is scheduled under the MFMAs of the next.
// -------------------------------------------------------------------------------------------
// ORDER: how a work item index becomes a tile (launch-uniform, a template parameter so that the decode is
// straight-line scalar code the scheduler can sink under the MFMAs): 0 = work list (a.items), 1 = whole frame
// in 4x8 blocks of tiles (decode_blocked), 2 = every tile of every plane in plain order.
#ifndef BODY_WAVES
#define BODY_WAVES 4
#endif
template <int ORDER>
__global__ void __launch_bounds__(64 * BODY_WAVES, 1) k_body(const ConvArgs a, const PlaneDesc* __restrict__ planes,
                                                  const uint32_t* __restrict__ items)
{
#ifndef BODY_CPW
#define BODY_CPW 4
#endif
#if BODY_CPW == 4 && !defined(WF_IN_AGPR)
#define WF_IN_AGPR 1
#endif
    // BODY_CPW == 4 (shipped): wave w owns tile rows 4w..4w+3 and ALL four co-blocks, so every B fragment is
    // read from LDS once per workgroup; 288 weight registers, 256 of them parked in AGPRs (the MFMA reads
    // its A operand from there; needs -mllvm -amdgpu-mfma-vgpr-form=1, see the Makefile).
    // BODY_CPW == 2 (previous layout, kept for A/B): wave (rh, ch) owns rows 8rh..8rh+7 and co-blocks 2ch,
    // 2ch+1: 144 weight registers, but every B fragment is read by two waves (2.5 % slower).
    constexpr int NCOB = 4;                      // co-blocks of the layer
    constexpr int CPW = BODY_CPW;                // co-blocks per wave
    // BODY_WAVES == 8 with BODY_CPW == 2 (experiment): two waves per SIMD (4 row groups x 2 channel halves),
    // so that a wave blocked on the issue of a store or an LDS-DMA instruction leaves its SIMD to the other
    constexpr int NW = BODY_WAVES;               // waves per workgroup
    constexpr int NRG = NW / (NCOB / CPW);       // row groups
    constexpr int ROWS = TILE_H / NRG;           // tile rows per wave
    constexpr int PER_WAVE = (DMA_PIECES + NW - 1) / NW;   // DMA pieces per wave
    constexpr int NH = CPW / 2;                  // 32-channel halves (16-byte stores per pixel) per wave
#ifndef SUB_PB
#define SUB_PB 4
#endif
    constexpr int SPB = SUB_PB;                  // px-blocks (16 px) per sub-iteration
    constexpr int NSUB = ROWS * 2 / SPB;         // sub-iterations per tile
#ifndef DMA_SPAN_SUBS
#define DMA_SPAN_SUBS (NSUB - 1)
#endif
    constexpr int DMA_SPAN = (DMA_SPAN_SUBS) * KSTEPS;   // k-steps over which the next tile's DMA is issued
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef STAMPS
    unsigned long long pro_[4] = {0, 0, 0, 0};
#define PSTAMP(i)                                                                           \
    do {                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pro_[i])::"memory");    \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    } while (0)
    unsigned long long t_entry_, r_entry_;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry_), "=s"(r_entry_)::"memory");
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = ROWS * (wave % NRG);                          // first tile row of this wave
    const int wh = wave / NRG;                                     // first channel half of this wave
    auto piece = [&](int k) { const int c = k * NW + wave; return c < DMA_PIECES ? c : DMA_PIECES - 1; };
    const int pl = lane & 15, g = lane >> 4;
    const int cob0 = wh * CPW;                                     // first co-block of this wave
#ifndef WEIGHTS_VIA_LDS
#define WEIGHTS_VIA_LDS 1
#endif
    if constexpr (WEIGHTS_VIA_LDS) {
        // first thing in the kernel: each wave DMAs its quarter of the 72 weight fragments (1 KiB each,
        // contiguous in a.wpack) into the second tile buffer; collected further down
        constexpr int NFRAG = KSTEPS * NCOB;
        auto wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack, 0, NFRAG * 1024, 0x00020000);
#pragma unroll
        for (int f = 0; f < NFRAG / NW; ++f)
            dma16(wrsrc, to_lds(smem + LDS_BUF_BYTES + (f * NW + wave) * 1024), lane * 16, (f * NW + wave) * 1024);
    }

    // ---- register-stationary weights.  With all four co-blocks per wave every wave needs the SAME 72 KiB,
    // so they come in through LDS once per workgroup (each wave DMAs a quarter into the second tile buffer,
    // which is idle until the first iteration issues the second tile's DMA) instead of four times through
    // the CU's vector-memory path: see the prologue below.  BODY_CPW == 2: plain loads.
    h8 wf[KSTEPS][CPW];
    if constexpr (!(WEIGHTS_VIA_LDS)) {
        const h8* wp = (const h8*)a.wpack;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int m = 0; m < CPW; ++m) wf[s][m] = wp[(s * NCOB + cob0 + m) * 64 + lane];
    }
    float bias[CPW][4];
#pragma unroll
    for (int m = 0; m < CPW; ++m) {
        const h4 b = *(const h4*)(a.bias + 16 * (cob0 + m) + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[m][r] = (float)b[r];
    }
    h8 slope8[NH];                 // slopes of this lane's 8 channels per half in store order [m][r]
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
        const h4 s0 = *(const h4*)(a.slope + 32 * (wh + hh) + 4 * g), s1 = *(const h4*)(a.slope + 32 * (wh + hh) + 16 + 4 * g);
        slope8[hh] = __builtin_shufflevector(s0, s1, 0, 1, 2, 3, 4, 5, 6, 7);
    }

    // ---- lane-constant LDS read offsets: [dx][half]; rows/columns of a px-block are immediates
    int roff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            roff[dx][hf] = (row0 * LDS_W + pl + dx) * PIX_BYTES + 16 * ((4 * hf + g) ^ ((pl + dx) & 6));

    // ---- lane-constant DMA source offsets (relative to the tile's first input pixel)
    int voff[PER_WAVE];
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        int q = piece(k) * 8 + (lane >> 3);
        q = q < LDS_PIX ? q : LDS_PIX - 1;
        const int yy = q / LDS_W, xx = q - yy * LDS_W;
        voff[k] = (yy * a.Wp + xx) * PIX_BYTES + 16 * ((lane & 7) ^ (xx & 6));
    }

    PSTAMP(0);   // lane constants done
    // ---- persistent loop over work items; blocks that share an XCD (b % 8) take adjacent tiles
    const int G = gridDim.x;
    const int b = blockIdx.x;
    const int first = ((G & 7) == 0) ? (b & 7) * (G >> 3) + (b >> 3) : b;
    int it = first;
    int cur = 0;
    // item index -> tile; past the end of the list the CURRENT tile is returned (its unused re-load keeps
    // the tile body branch-free)
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
    // tiles and plane descriptors are decoded two iterations ahead and carried, so that no scalar load or
    // division sits between the barrier and the first MFMA of a tile
    Item itm = item_at(it), nitm = item_at(it + G);
    PlaneDesc pd = planes[itm.plane], npd = planes[nitm.plane];
    if (it < a.n_items) {
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (unsigned long long)itm.plane * a.plane_stride),
                                                      0, (int)a.plane_stride, 0x00020000);
        const int org = ((itm.ty * TILE_H) * a.Wp + itm.tx * TILE_W) * PIX_BYTES;
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k)
            dma16(rsrc, to_lds(smem + piece(k) * 1024), voff[k], org);
    }
    PSTAMP(1);   // first tile's DMA issued
    if constexpr (WEIGHTS_VIA_LDS) {
        static_assert(!(WEIGHTS_VIA_LDS) || (KSTEPS * NCOB <= DMA_PIECES && (KSTEPS * NCOB) % NW == 0), "the packed weights must fit one tile buffer");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        PSTAMP(2);   // weights (and the first tile) have landed in LDS
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int m = 0; m < CPW; ++m) wf[s][m] = *(const h8*)(smem + LDS_BUF_BYTES + (s * NCOB + cob0 + m) * 1024 + lane * 16);
    }
    // Pin the wait for the weight loads HERE: left to itself hipcc puts a counted wait at each fragment's
    // first use inside the loop, where it would drain the next tile's DMA every iteration.  (With the
    // weights read from LDS this is also what makes every wave finish reading the second tile buffer
    // before the first iteration's barrier lets anyone DMA into it.)
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < CPW; ++m) {
#ifdef WF_IN_AGPR
            // park the weights in the accumulator file: v_mfma reads them from there (256 AGPRs = 64 fragments)
            if (s * CPW + m < 64) asm volatile("" : "+a"(wf[s][m]));
            else asm volatile("" : "+v"(wf[s][m]));
#else
            asm volatile("" : "+v"(wf[s][m]));
#endif
        }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

#ifdef STAMPS
    unsigned long long seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
    seg_[7] = last_ - t_entry_;        // prologue: weights, lane constants, first tile's DMA issued and landed
#endif
#ifdef STAGGER_SLEEPS
    // de-phase the workgroups (they all start together and run identical work): delay by group
    for (int i = 0; i < (int)((blockIdx.x >> 3) & 7) * STAGGER_SLEEPS; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    while (it < a.n_items) {
        __builtin_amdgcn_s_barrier();      // every wave's DMA share of this tile has landed and
        asm volatile("" ::: "memory");     // every wave is done reading the other buffer
        STAMP(0);                          // barrier wait
#ifdef WAVE_SKEW
        // the four waves leave the barrier in lockstep and would hit the CU's store path with their
        // epilogue stores at the same instant; skew them by WAVE_SKEW*64 cycles each
        for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(WAVE_SKEW);
#endif
        const int nxt = it + G;
        const Item nnitm = item_at(nxt + G);          // used from the next iteration on
        const PlaneDesc nnpd = planes[nnitm.plane];
        // The next tile's DMA pieces are issued one per k-step under the first sub-iteration's
        // MFMAs.  On the last tile the (unused) re-load of the same tile keeps the body branch-free.
        auto nrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (unsigned long long)nitm.plane * a.plane_stride),
                                                       0, (int)a.plane_stride, 0x00020000);
#ifdef ABL_DMA_SAMEADDR
        const int norg = (nitm.tx & 1) * TILE_W * PIX_BYTES;
#else
        const int norg = ((nitm.ty * TILE_H) * a.Wp + nitm.tx * TILE_W) * PIX_BYTES;
#endif
        char* nbuf = smem + (cur ^ 1) * LDS_BUF_BYTES;
        const int bufoff = cur * LDS_BUF_BYTES;
        // stores go through a buffer descriptor so that masked pixels are dropped by the bounds
        // check instead of a branch (keeps the tile body one basic block)
        auto orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (unsigned long long)itm.plane * a.plane_stride),
                                                       0, (int)a.plane_stride, 0x00020000);

        STAMP(1);                          // tile set-up
#ifndef DEFER_STORES
#define DEFER_STORES 1
#endif
        // body layers: a sub-iteration's 16-byte stores are not issued in a burst behind its MFMAs
        // (the CU's store path takes ~200 cycles per 1-KiB store; a burst fills its FIFO and stalls
        // the wave, MFMAs included) but one at a time under the NEXT sub-iteration's MFMAs
        u32x4 pend_o[SPB * NH];
        int pend_off[SPB * NH];
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
#ifdef ABL_NO_LDS
                        B[q] = __builtin_bit_cast(h8, (u32x4){(unsigned)roff[dx][hf], (unsigned)rr, (unsigned)lane, 0x3c003c00u});
                        asm volatile("" : "+v"(B[q]));
#else
                        B[q] = *(const h8*)(smem + bufoff + roff[dx][hf] + ((rr + dy) * LDS_W + 16 * xb) * PIX_BYTES);
#endif
                    }
                    if constexpr (DEFER_STORES) {
                        if (si > 0) {
#pragma unroll
                            for (int q = 0; q < SPB * NH; ++q)
                                if (ks == 2 + q * (KSTEPS - 2) / (SPB * NH))
                                    __builtin_amdgcn_raw_buffer_store_b128(pend_o[q], orsrc, pend_off[q], 0, STORE_AUX);
                        }
                    }
#ifndef ABL_NO_DMA
                    {
                        // next tile's DMA pieces, spread evenly over the first DMA_SPAN k-steps of the
                        // tile: a burst of all 20 backs up the CU's in-order vector-memory path and the
                        // epilogue stores (and the MFMAs behind them) stall on it
                        const int gs = si * KSTEPS + ks;
#pragma unroll
                        for (int k = 0; k < PER_WAVE; ++k)
                            if (k * DMA_SPAN / PER_WAVE == gs)
                                dma16(nrsrc, to_lds(nbuf + piece(k) * 1024), voff[k], norg);
                        // hipcc is free to move stores and LDS-DMA loads past each other; the counted
                        // vmcnt at the end of the tile needs every DMA to be older than the stores it
                        // leaves in flight, so nothing may cross the point of the last DMA issue
                        constexpr int GS_LAST = (PER_WAVE - 1) * DMA_SPAN / PER_WAVE;
                        static_assert(!DEFER_STORES || (GS_LAST / KSTEPS == NSUB - 2 && GS_LAST % KSTEPS > 2 + (SPB - 1) * (KSTEPS - 2) / SPB) || NSUB < 3,
                                      "deferred stores of sub-iteration NSUB-3 must precede the last DMA issue");
                        if (gs == GS_LAST) __builtin_amdgcn_sched_barrier(0);
                    }
#endif
#ifdef ABL_NO_MFMA
#pragma unroll
                    for (int q = 0; q < SPB; ++q) asm volatile("" ::"v"(B[q]));
                    if (ks == 0) {
#pragma unroll
                        for (int m = 0; m < CPW; ++m)
#pragma unroll
                            for (int q = 0; q < SPB; ++q) acc[m][q] = MFMA16(wf[ks][m], B[q], acc[m][q]);
                    }
#else
#pragma unroll
                    for (int m = 0; m < CPW; ++m)
#pragma unroll
                        for (int q = 0; q < SPB; ++q) acc[m][q] = MFMA16(wf[ks][m], B[q], acc[m][q]);
#endif
                }
            }
#ifdef WF_IN_AGPR
#pragma unroll
            for (int m = 0; m < CPW; ++m)
#pragma unroll
                for (int q = 0; q < SPB; ++q) asm volatile("" : "+v"(acc[m][q]));   // accumulators in VGPRs: no v_accvgpr_read in the epilogue
#endif

#pragma unroll
            for (int q = 0; q < SPB; ++q) {
                const int rr = (si * SPB + q) >> 1, xb = (si * SPB + q) & 1;
                const int oy = itm.ty * TILE_H + row0 + rr;
                const int ox = itm.tx * TILE_W + 16 * xb + pl;
#ifdef ABL_NO_EPI
                {
#pragma unroll
                    for (int m = 0; m < CPW; ++m) asm volatile("" ::"v"(acc[m][q]));
                    (void)oy; (void)ox;
                }
#else
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) {
                    // lane holds channels 32ch+16m+4g+r of pixel (oy,ox) -> 16 contiguous bytes at 64ch+16g
                    h8 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        o[r] = (_Float16)acc[2 * hh][q][r];
                        o[4 + r] = (_Float16)acc[2 * hh + 1][q][r];
                    }
                    o = prelu8(o, slope8[hh]);
                    const bool ok = oy < pd.h && ox < pd.w;
#ifdef ABL_STORE_LINEAR
                    const int off = (((it * 16 + si * 4 + q) * 4 + wave) * 64 + lane) * 16;
#else
                    const int off = ((oy + 1) * a.Wp + (ox + 1)) * PIX_BYTES + 64 * (wh + hh) + 16 * g;
#endif
#ifdef ABL_EPI_NOSTORE
                    asm volatile("" ::"v"(o), "v"(ok ? off : 0x7fffffff));
#else
                    if (DEFER_STORES && si + 1 < NSUB) {
                        pend_o[q * NH + hh] = __builtin_bit_cast(u32x4, o);
                        pend_off[q * NH + hh] = ok ? off : 0x7fffffff;
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc,
                                                               ok ? off : 0x7fffffff, 0, STORE_AUX);
                    }
#endif
                }
#endif
            }
            STAMP(2 + (si < 4 ? si : 3));  // sub-iteration si: k-loop + the epilogue work scheduled in it
        }
        // Before the barrier every wave must know ITS pieces of the next tile have landed.  vmcnt
        // retires in issue order; behind the last DMA issue (pinned by the sched_barrier above) come
        // exactly the stores of the last two sub-iterations, so a counted wait leaves those in flight.
#if defined(ABL_NO_EPI) || defined(ABL_EPI_NOSTORE)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPB * NH * (NSUB - (DMA_SPAN_SUBS) + 1)) : "memory");
#endif
        STAMP(4);                          // counted vmcnt wait
        cur ^= 1;
        it = nxt;
        itm = nitm; pd = npd;
        nitm = nnitm; npd = nnpd;
    }
#ifdef STAMPS
    if (lane == 0 && blockIdx.x < 2048 / NW) {
        unsigned long long r_exit_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r_exit_)::"memory");
        seg_[5] = r_entry_;            // 100 MHz wall clock at entry and exit: start skew and tail of the launch
        seg_[6] = r_exit_;
#pragma unroll
        for (int i = 0; i < 8; ++i) g_stamps[(blockIdx.x * NW + wave) * 8 + i] = seg_[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) g_pro[(blockIdx.x * NW + wave) * 4 + i] = pro_[i] - t_entry_;
    }
#endif
}

// -------------------------------------------------------------------------------------------
int conv_lds_bytes() { return 2 * LDS_BUF_BYTES; }

#ifdef STAMPS
extern "C" int reve_debug_read_stamps(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
extern "C" int reve_debug_read_prologue(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pro), sizeof(unsigned long long) * n);
}
#endif

template __global__ void k_body<0>(const ConvArgs, const PlaneDesc* __restrict__, const uint32_t* __restrict__);
template __global__ void k_body<1>(const ConvArgs, const PlaneDesc* __restrict__, const uint32_t* __restrict__);
template __global__ void k_body<2>(const ConvArgs, const PlaneDesc* __restrict__, const uint32_t* __restrict__);

// Function attributes belong to the CURRENT device: Engine::init calls the prepare_* functions once per
// context after hipSetDevice (a process-wide "once" would leave the second GPU of a group without them).
int prepare_body_kernels()
{
    return (int)hipFuncSetAttribute((const void*)k_body<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF_BYTES) |
           (int)hipFuncSetAttribute((const void*)k_body<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF_BYTES) |
           (int)hipFuncSetAttribute((const void*)k_body<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BUF_BYTES);
}

// host-side evaluation of the computed work order (tests: must equal Engine::configure()'s list order)
void debug_blocked_order(int tiles_x, int tiles_y, uint32_t* out)
{
    for (int it = 0; it < tiles_x * tiles_y; ++it) {
        const Item i = decode_blocked(it, tiles_x, tiles_y);
        out[it] = (uint32_t)i.tx | ((uint32_t)i.ty << 10);
    }
}

int launch_body(const ConvArgs& a, int grid, void* stream)
{
    const size_t lds = 2 * LDS_BUF_BYTES;
    if (a.items) hipLaunchKernelGGL(k_body<0>, dim3(grid), dim3(64 * BODY_WAVES), lds, (hipStream_t)stream, a, a.planes, a.items);
    else if (a.blocked) hipLaunchKernelGGL(k_body<1>, dim3(grid), dim3(64 * BODY_WAVES), lds, (hipStream_t)stream, a, a.planes, a.items);
    else hipLaunchKernelGGL(k_body<2>, dim3(grid), dim3(64 * BODY_WAVES), lds, (hipStream_t)stream, a, a.planes, a.items);
    return (int)hipGetLastError();
}

}  // namespace reve
