// Two consecutive body layers (64 -> 64 3x3 conv + bias + fp16 round + PReLU, twice) in ONE launch for gfx950: the layer
// between them never leaves the CU.  Same arithmetic, same summation order and the same HBM layout as two k_body launches
// (kernels.hip), so the result is bit-identical to the layer-per-launch path; what changes is the traffic: one activation read
// and one activation write per PAIR (SURVEY.md §7.3 "fused multi-layer").  Replaces nothing in the reference
// beyond what k_body replaces (reve-shared/src/lib.rs:134-147: the realesrgan-ncnn-vulkan subprocess).
//
// Shape: a workgroup (4 waves, one per SIMD) owns a vertical STRIP of the frame — 62 output columns of the second layer — and
// rolls down a segment of its rows.  The waves are specialised by LAYER, because one wave can keep exactly one layer's weights
// in registers (18 k-steps x 4 co-blocks x 4 = 288 of its 512, 256 of them in AGPRs, as in k_body):
//   waves 0, 1 ("A"): first layer.  Input rows come from HBM by LDS-DMA into an 8-row ring (66 px x 128 B per row at a pitch of
//       72 px: nine pieces of 8 px per row); each wave computes 32 of the 64 columns of a row for all 64 output channels and
//       writes the fp16 result (pixels outside the frame forced to zero: they are the second layer's padding) into a second
//       8-row ring in LDS (64 px per row);
//   waves 2, 3 ("B"): second layer, three steps behind, reading that ring, storing to the output arena like k_body.
// Per strip both layers compute 64 columns (the first from 66 input columns: all 64 valid; the second's 62 valid): 31 x 64 =
// 1,984 columns per 1,920-px row, plus one extra A row above and below each segment of rows.
// A STEP is two rows per wave (288 MFMAs) and one s_barrier.  The two rows advance in lock-step through one px-block at a
// time, one input row apart: in slot (r, t) row 0 multiplies tap row r of input row r, row 1 tap row r of input row r + 1 — the
// SAME weights fragment — and the operand fragment row 1 has just used is the one row 0 needs six slots later, so it stays in a
// register window of eight fragments instead of being read from LDS again: 24 ds_read_b128 per 144 MFMAs (k_body: 36), every
// slot 8 independent MFMAs, each accumulator summing its taps in the same (dy, dx, hf) order as k_body: same bits.  The
// epilogue of a px-block's two rows runs in four pieces under the MFMAs of the next px-block, which is why B runs THREE steps
// behind A (rows computed in step s are complete in LDS in step s+1 and visible after that step's barrier).
// Ring slots: input row rho -> slot rho & 7, filled by DMA two steps before its first use; mid row r -> slot r & 7.
//   step s: A reads input rows 2s..2s+3, DMA fills rows 2s+6, 2s+7 (slots of rows 2s-2, 2s-1: free since step s-1);
//           A writes mid rows 2s-2..2s+1 (px-block by px-block); B reads mid rows 2s-6..2s-3 — disjoint slots.
// Work: units = strips x segments of rows; the host picks the segment height so that no workgroup has more than one unit where
// the frame allows it (1080p: 31 strips x 8 segments of 135 rows = 248 units on 256 CUs).  A unit restarts the pipeline (3
// steps of fill).  Each role runs its own loops over a unit's phases (idle steps, active steps, idle steps): a step is
// straight-line code with no branch and no register copy in it, 288 MFMAs + ~390 other instructions (tests/test_kernel_isa.py).
// Measurements, ablations and what was tried and dropped: DESIGN.md §4, docs/LAB_NOTES.md, profiles/r03/, profiles/r04/.
#include <type_traits>

// Cache policy of the input rows' LDS-DMA loads (buffer aux bits: 1 sc0, 2 nt, 16 sc1).  nt: a strip reads every row once
// (neighbouring strips share 2 of 66 columns), so the rows are loaded as streaming data and do not push what the launch WRITES
// out of L2 / the Infinity Cache — which is what the next launch reads.  Measured -3.0 % per layer against plain loads, the same
// with sc1 added (profiles/r03/ab_load_policy.txt); the tile kernel k_body, whose tiles re-read their halos (1.195x), is 4.7 %
// SLOWER with nt loads and keeps plain ones.
#ifndef KP_DMA_AUX
#define KP_DMA_AUX 2
#endif

#include "kernels_dev.h"

namespace reve {

// The timing-only instrumentation (in-kernel stamps, the ABLP_* ablations) lives in kernels_pair_diag.inc and exists in
// diagnostic builds only (scripts/ablate_pair.sh passes -DREVE_DIAGNOSTIC_BUILD); a product build sees the empty hooks below and
// compiles with that file absent (tests/test_kernel_isa.py).
#if (defined(STAMPS) || defined(ABLP_NO_LDS) || defined(ABLP_NO_EPI) || defined(ABLP_NO_EPI_ROLE) || defined(ABLP_NO_DMA) || defined(ABLP_UNUSED_LDS) || defined(ABLP_STORE_WRAP)) && !defined(REVE_DIAGNOSTIC_BUILD)
#error "STAMPS / ABLP_* are timing-only diagnostic switches: build them through scripts/ablate_pair.sh (-DREVE_DIAGNOSTIC_BUILD), never into libreve_hip.so"
#endif
#ifdef REVE_DIAGNOSTIC_BUILD
#include "kernels_pair_diag.inc"
#else
constexpr bool kpd_no_dma = false, kpd_counted_waits = true;
constexpr bool kpd_epi_off(int) { return false; }
#define KPD_ENTRY
#define KPD_LOOP_BEGIN
#define KPD_STEP_BEGIN
#define KPD_STEP_END
#define KPD_WAIT_BEGIN
#define KPD_WAIT_END(active)
#define KPD_EXIT
#define KPD_OPERANDS
#define KPD_LOAD_F(i)
#define KPD_BNEXT
#define KPD_OVERRIDE_OPERANDS(op0, op1, c, z)
#define KPD_KEEP(...)
#define KPD_STORE_FOLD(off) (off)
#endif

#ifndef KP_STORE_AUX
#define KP_STORE_AUX 0          // cache policy of the second layer's activation stores (buffer aux bits: 1 sc0, 2 nt, 16 sc1)
#endif
#ifndef KP_VALU_PER_MFMA
#define KP_VALU_PER_MFMA 3      // epilogue VALU instructions placed behind each MFMA of a slot (sched_group_barrier).  (2, the slot as ONE scheduling
#endif                          // region with its scalar work spread behind the MFMAs, the pending rows' addresses worked out in slot 3 instead
                                // of at the head of the px-block: all within 0.6 % of each other, profiles/r04/ab_pair_variants.txt — the clock
                                // gives back what the cycles save.)
#ifndef KP_MFMA_ORDER
#define KP_MFMA_ORDER 0     // order of a slot's eight MFMAs: 0 weights fragment constant over two (shipped), 1 snake, 2 pixels constant over four
#endif

namespace {
constexpr int KP_NW = 4;
constexpr int KP_COLS = PAIR_COLS;                       // columns computed per row and layer (4 px-blocks)
constexpr int KP_ROW_BYTES = KP_COLS * PIX_BYTES;        // a mid-ring row: 8,192 B
constexpr int KP_RING = 8;                               // rows per ring
constexpr int KP_RING_BYTES = KP_RING * KP_ROW_BYTES;    // the mid ring: 65,536 B
constexpr int KP_RPS = 2;                                // rows per step and wave
constexpr int KP_LAG = 3;                                // steps B runs behind A
constexpr int KP_NFRAG = KSTEPS * 4;                     // A fragments per layer (72 KiB)
// The INPUT ring holds KP_COLS + 2 columns per row, so that all 64 columns the first layer computes are valid and the second
// layer's strip is 62 columns wide (1080p: 31 strips x 64 = 1,984 columns computed per row instead of 32 x 64 = 2,048).
// A row is filled by nine LDS-DMA pieces of 8 px (the ninth carries columns 64, 65 and six slots nobody reads): row pitch
// 72 px, so that a piece never straddles rows and its row is a scalar offset.
constexpr int KP_IN_COLS = KP_COLS + 2;
constexpr int KP_PPR = (KP_IN_COLS + 7) / 8;             // DMA pieces per input row (9)
constexpr int KP_IN_ROW_BYTES = KP_PPR * 1024;           // 9,216
constexpr int KP_IN_RING_BYTES = KP_RING * KP_IN_ROW_BYTES;     // 73,728
constexpr int KP_MID_OFF = KP_IN_RING_BYTES;
constexpr int in_row_off(int rho) { return (rho & (KP_RING - 1)) * KP_IN_ROW_BYTES; }
// Start of a launch: the FIRST layer's weights come in through LDS (a quarter DMA'd by each wave, staged in the mid ring's
// space and 8 KiB beyond it, which nobody writes before step 0) together with the first input rows, one wait for both; the
// SECOND layer's weights are loaded by its two waves straight from global memory while the first layer's waves already
// compute: they land during the three fill steps in which B has nothing else to do.  (Both layers staged through LDS in
// front of everything: 144 KiB per CU before the first MFMA, ~13 us of a 250 us launch.)
constexpr int KP_STAGE_OFF = KP_MID_OFF;
constexpr int KP_LDS = KP_STAGE_OFF + KP_NFRAG * 1024 + 1024;
static_assert(KP_LDS >= KP_MID_OFF + KP_RING_BYTES + 1024 && KP_LDS <= 160 * 1024, "LDS budget of a CU");
static_assert(KP_NFRAG % KP_NW == 0, "the staged weights are dealt out evenly");
// a step's DMA pieces: two rows x nine = 18.  Wave w takes column groups 2w and 2w + 1 of both rows (four pieces); the ninth
// group (columns 64, 65) of row 0 goes to wave 0 and of row 1 to wave 1 — five pieces per step for the first layer's waves, four
// for the second's, which also have the stores: the end-of-step vmcnt is one compile-time number per role (5; 4 + 8)
constexpr int KP_DMA_PER_WAVE = 5;
constexpr int kp_dma_count(int role) { return role == 0 ? KP_DMA_PER_WAVE : KP_DMA_PER_WAVE - 1; }
static_assert(KP_PPR == 2 * KP_NW + 1 && KP_RPS == 2, "the piece assignment above");
}  // namespace

// GUT: the frame is a canvas of several planes (tiled frames, Engine::configure): the gutter columns (a.col_ok) and rows (a.gut_*)
// between planes are each plane's zero padding — the first layer writes zeros there, the second stores nothing.  Its own
// instantiations: whole frames carry none of it.
template <bool UNIT_SLOPES, bool GUT>
__global__ void __launch_bounds__(64 * KP_NW, 1) k_pair(const PairArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    KPD_ENTRY
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 1;                // 0: first layer (A), 1: second layer (B)
    const int half = wave & 1;                 // which 32 of the 64 columns
    const int pl = lane & 15, g = lane >> 4;

    // ---- first layer's weights -> LDS staging (every wave DMAs a quarter); waves 0, 1 read them into registers below
    {
        auto w0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack[0], 0, KP_NFRAG * 1024, 0x00020000);
#pragma unroll
        for (int f = 0; f < KP_NFRAG / KP_NW; ++f) {
            const int idx = f * KP_NW + wave;
            dma16(w0, to_lds(smem + KP_STAGE_OFF + idx * 1024), lane * 16, idx * 1024);
        }
    }
    const uint16_t* bias_p = role ? a.bias[1] : a.bias[0];
    const uint16_t* slope_p = role ? a.slope[1] : a.slope[0];
    float bias[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const h4 b = *(const h4*)(bias_p + 16 * m + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[m][r] = (float)b[r];
    }
    h8 slope8[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const h4 s0 = *(const h4*)(slope_p + 32 * hh + 4 * g), s1 = *(const h4*)(slope_p + 32 * hh + 16 + 4 * g);
        slope8[hh] = __builtin_shufflevector(s0, s1, 0, 1, 2, 3, 4, 5, 6, 7);
    }

    // ---- lane-constant address parts
    // operand reads: output column c = 32 * half + 16 * q + pl of a row reads ring columns c + dx of ring rows R + dy (the ring's
    // own offset and the row's are added per step)
    int roff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            roff[dx][hf] = (32 * half + pl + dx) * PIX_BYTES + 16 * ((4 * hf + g) ^ ((pl + dx) & 6));
    // (A's 16-byte piece of channel half hh for column 32 * half + pl goes to the same lane offset in a mid-ring row as the
    // operand read with dx = 0, hf = hh comes from: roff[0][hh])
    // B: arena pixel (1, 1 + 32 * half + pl), this lane's 16-byte chunk
    const int soff_lane = (a.Wp + 1 + 32 * half + pl) * PIX_BYTES + 16 * g;

    const int plane_bytes = a.Hp * a.Wp * PIX_BYTES;
    auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, plane_bytes, 0x00020000);
    auto no_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0, 0x00020000);       // zero bytes: loads fetch nothing
    auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, plane_bytes, 0x00020000);

    const int G = gridDim.x;
    const int bid = blockIdx.x;
    int u = ((G & 7) == 0) ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;      // blocks of one XCD take neighbouring units

    // ---- the unit in hand: a strip (62 output columns of B) x a segment of rows
    int x0 = 0, y0 = 0, y1 = 0, NA = 0, SA = 0, n_steps = 0;
    // column masks of this lane's pixels (all ones / zero): A zeroes what lies outside the frame, B stores only its 62 valid
    // columns inside the frame.  Applied with bit operations: written as `cond ? x : 0` hipcc turns them into exec-mask
    // branches, and a branch splits the scheduling region that pins the MFMA / VALU interleave.
    unsigned colmask[2] = {0u, 0u};
    // The k-th LDS-DMA piece of this wave for input rows rho0, rho0 + 1: k < 4: column group 2 * wave + (k >> 1) of row k & 1;
    // k == 4 (waves 0, 1): the ninth group of row `wave`.  Ring column j <-> arena column x0 - 1 + j, input row rho <-> arena row
    // y0 - 1 + rho (both clamped: the arena's border rows and columns are zero; columns past 65 repeat column 65 into slots
    // nobody reads).  vcol: the lane-constant source offset of each of the wave's three column groups.
    int vcol[3] = {0, 0, 0};
    auto dma_piece_k = [&](int rho0, int k, bool needed) {
        const int ci = k >> 1, c = ci < 2 ? 2 * wave + ci : KP_PPR - 1, row = k < 4 ? (k & 1) : wave;
        int ar = y0 - 1 + rho0 + row;
        ar = ar < 0 ? 0 : (ar > a.Hp - 1 ? a.Hp - 1 : ar);
        dma16a<KP_DMA_AUX>(needed ? in_rsrc : no_rsrc, to_lds(smem + in_row_off(rho0 + row) + c * 1024), vcol[ci], ar * a.Wp * PIX_BYTES);
    };
    // takes unit `uu` in hand and starts the DMA of input rows 0..5 (what steps 0 and 1 read)
    auto unit_setup = [&](int un) {
        const int uu = a.reverse ? a.n_units - 1 - un : un;
        const int sy = uu / a.n_strips, sx = uu - sy * a.n_strips;
        x0 = sx * PAIR_VALID;                        // image column of B's first output column
        y0 = sy * a.seg_h;
        y1 = y0 + a.seg_h < a.H ? y0 + a.seg_h : a.H;
        const int NB = y1 - y0;
        NA = NB + 2;
        const int SB = (NB + KP_RPS - 1) / KP_RPS;
        SA = (NA + KP_RPS - 1) / KP_RPS;
        n_steps = SB + KP_LAG;                       // = SA + 2
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = 32 * half + 16 * q + pl;
            bool ok = role ? (c < PAIR_VALID && x0 + c < a.W) : (x0 - 1 + c >= 0 && x0 - 1 + c < a.W);
            if constexpr (GUT) {
                int x = role ? x0 + c : x0 - 1 + c;
                x = x < 0 ? 0 : (x > a.W - 1 ? a.W - 1 : x);
                if (a.col_ok) ok = ok && a.col_ok[x] != 0;
            }
            colmask[q] = ok ? 0xffffffffu : 0u;
        }
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const int c = ci < 2 ? 2 * wave + ci : KP_PPR - 1;
            int j = 8 * c + (lane >> 3);
            j = j < KP_IN_COLS ? j : KP_IN_COLS - 1;
            int ac = x0 - 1 + j;
            ac = ac < 0 ? 0 : (ac > a.Wp - 1 ? a.Wp - 1 : ac);
            vcol[ci] = ac * PIX_BYTES + 16 * ((lane & 7) ^ (j & 6));
        }
#pragma unroll
        for (int blk = 0; blk < 3; ++blk)
#pragma unroll
            for (int k = 0; k < KP_DMA_PER_WAVE; ++k)
                if (k < KP_DMA_PER_WAVE - 1 || role == 0) dma_piece_k(KP_RPS * blk, k, true);
    };

    // ---- first unit: its rows travel with the staged weights, one wait for both
    unit_setup(u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // (bias and slopes are converted HERE, under the wait above: left to itself hipcc sinks the conversions below B's weight
    // loads, and their wait — for loads older than those — then drains them in front of the barrier)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bias[m][r]));
    asm volatile("" : "+v"(slope8[0]), "+v"(slope8[1]));
    h8 wf[KSTEPS][4];
    if (role == 0) {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int m = 0; m < 4; ++m) wf[s][m] = *(const h8*)(smem + KP_STAGE_OFF + (s * 4 + m) * 1024 + lane * 16);
    } else {
        // (in flight across the barrier below: B's first use of them is three steps away.  The eight fragments that stay in
        // VGPRs go first: hipcc merges them with the other branch's by register copies in front of the barrier, and a copy
        // waits for its load and every OLDER one)
#pragma unroll
        for (int i = 0; i < KP_NFRAG; ++i) {
            const int f = (i + 64) % KP_NFRAG;
            wf[f / 4][f % 4] = ((const h8*)a.wpack[1])[f * 64 + lane];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // the staging space goes back to the mid ring
    asm volatile("" ::: "memory");
    // 256 of the 288 registers are parked in the accumulator file, the MFMA reads its A operand from there
    // (-mllvm -amdgpu-mfma-vgpr-form=1); for B this is also where its loads are waited for
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (s * 4 + m < 64) asm volatile("" : "+a"(wf[s][m]));
            else asm volatile("" : "+v"(wf[s][m]));
        }

    auto epi = [&](const f4 (&ac)[4][2], int q, int hh) {
        h8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[r] = (_Float16)ac[2 * hh][q][r];
            o[4 + r] = (_Float16)ac[2 * hh + 1][q][r];
        }
        return __builtin_bit_cast(u32x4, UNIT_SLOPES ? prelu8_unit_slopes(o, slope8[hh]) : prelu8(o, slope8[hh]));
    };
    // hand one finished piece over: A -> mid ring (zero outside the frame), B -> arena (dropped outside its columns / rows)
    auto put = [&](auto role_c, u32x4 v, int q, int hh, int base, bool ok) {
        const unsigned m = colmask[q] & (ok ? 0xffffffffu : 0u);
        if constexpr (decltype(role_c)::value == 0) {
            v &= (u32x4){m, m, m, m};
            *(u32x4*)(smem + KP_MID_OFF + base + roff[0][hh] + 16 * q * PIX_BYTES) = v;
        } else {
            const unsigned off = (KPD_STORE_FOLD((unsigned)(base + soff_lane + 16 * q * PIX_BYTES + 64 * hh)) & m) | (0x7fffffffu & ~m);
            __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, (int)off, 0, KP_STORE_AUX);
        }
    };

    // A role's whole life in the launch.  The roles are separated OUTSIDE the loops and a unit is walked in phases (second layer:
    // KP_LAG idle steps, then its active steps; first layer: its active steps, then two idle ones).  A wave alone on its SIMD
    // issues one instruction of any class per 4 cycles (an MFMA takes two turns): what a loop body merges at its end — register
    // copies that bring two paths together, the branches themselves at ~50 cycles when taken — is paid in full, beside an MFMA
    // pipe that idles meanwhile (scripts/ubench/valu_issue.hip; the phases used to be branches inside one loop over the steps).
    auto life = [&](auto role_c) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(role_c)::value;
        // the end of a step: the DMA pieces of the PREVIOUS step (read in the next one) have landed; this wave's LDS writes are
        // done.  Younger than those pieces: everything of this step (A: 5 DMA pieces; B: 4 + its 8 stores when active).
        auto step_end = [&](auto active_c) {
            constexpr bool active = decltype(active_c)::value;
            if constexpr (!kpd_counted_waits) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (timing-only builds: the counts below do not apply)
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kp_dma_count(ROLE) + (ROLE && active ? 2 * KP_RPS * 2 : 0)) : "memory");
            KPD_WAIT_BEGIN
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            KPD_WAIT_END(active)
        };
        auto idle_step = [&](int s) {
            const bool dma_needed = KP_RPS * s + 6 <= NA + 1;
#pragma unroll
            for (int k = 0; k < kp_dma_count(ROLE); ++k) dma_piece_k(KP_RPS * s + 6, k, dma_needed);
            step_end(std::false_type{});
        };
        // LDS offset of ring row R as the role READS it: A the input ring (two-row blocks), B the mid ring
        auto ring_row = [&](int R) {
            if constexpr (ROLE == 0) return in_row_off(R);
            else return KP_MID_OFF + (R & (KP_RING - 1)) * KP_ROW_BYTES;
        };

        for (;;) {
            // the row whose epilogue is pending (computed last, not yet written): accumulators + where it goes
            f4 racc[4][2];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int q = 0; q < 2; ++q) racc[m][q] = (f4){0.f, 0.f, 0.f, 0.f};
            // the pending results: rows e_R, e_R + 1 of px-block e_q; e_live: there are any.  At the start of a unit A's first pieces go
            // to slots 6, 7 of the mid ring, which nobody reads yet
            int e_R = -2, e_q = 1;
            bool e_live = false;
            // Gutter rows are gut_first + k * gut_period.  A role asks about its rows pair by pair in ascending order, every pair twice
            // (once per px-block: ... 84, 85, 84, 85, 86, 87 ...): gut_next is the first gutter row >= the highest row asked about so
            // far minus one (scalar: one compare-and-add per query; nothing in whole-frame instantiations)
            int gut_next = 0;
            if constexpr (GUT) {
                const int y_first = (ROLE ? y0 : y0 - 1) - 3;        // (the unit's first queries are about the two rows above its first)
                const int k = (a.gut_period > 0 && y_first > a.gut_first) ? (y_first - a.gut_first + a.gut_period - 1) / a.gut_period : 0;      // (one row of planes: no gutter rows, period 0)
                gut_next = a.gut_period > 0 ? a.gut_first + k * a.gut_period : 0x7fffffff;
            }
            auto is_gutter = [&](int y) {
                if constexpr (GUT) {
                    gut_next += gut_next < y - 1 ? a.gut_period : 0;
                    return y == gut_next;
                }
                return false;
            };
            auto pend_base = [&](int row) {
                if constexpr (ROLE == 0) return ((e_R + row) & (KP_RING - 1)) * KP_ROW_BYTES;
                else return ((y0 + e_R + row) * a.Wp + x0) * PIX_BYTES;
            };
            auto pend_ok = [&](int row) {
                // (is_gutter is asked unconditionally and combined without short-circuit: it has a side effect, and a conditional call would
                // be a branch in the middle of a step)
                if constexpr (ROLE == 0) {
                    const int ya = y0 - 1 + e_R + row;
                    const bool gut = is_gutter(ya);
                    return (bool)((ya >= 0) & (ya < a.H) & !gut);
                } else {
                    const int yb = y0 + e_R + row;
                    const bool gut = is_gutter(yb);
                    return (bool)(e_live & (yb >= y0) & (yb < y1) & !gut);
                }
            };
            // the pending rows' four pieces with no MFMAs to ride under: the end of a role's work in this unit
            auto flush = [&]() {
#pragma unroll
                for (int p = 0; p < 4; ++p) put(role_c, epi(racc, p >> 1, p & 1), e_q, p & 1, pend_base(p >> 1), pend_ok(p >> 1));
            };
            // the first two operand fragments of a step are read at the END of the step before (their rows landed / were written at
            // least a step earlier), so their LDS latency passes under the barrier instead of in front of the step's first MFMA; those
            // of the first active step are read ahead of the loop
            h8 Bnext[2];

            // One active step: straight-line code (a branch would split the scheduling regions that pin the MFMA / VALU / memory
            // interleave)
            auto step = [&](int s) __attribute__((always_inline)) {
                const int R0 = ROLE ? KP_RPS * (s - KP_LAG) : KP_RPS * s;         // first row of this step (relative to the role's first row)
                const bool dma_needed = KP_RPS * s + 6 <= NA + 1;                   // input rows 2s+6, 2s+7 exist for this unit
                // ring rows this step reads: R0 .. R0 + 3
                int rb[KP_RPS + 2];
#pragma unroll
                for (int i = 0; i < KP_RPS + 2; ++i) rb[i] = ring_row(R0 + i);
                KPD_OPERANDS
                // operand fragment of input row i (0..3 of the step), column shift / channel half t = 2 * dx + hf, px-block q
                auto load_f = [&](int i, int t, int q) {
                    KPD_LOAD_F(i)
                    return *(const h8*)(smem + rb[i] + roff[t >> 1][t & 1] + 16 * q * PIX_BYTES);
                };
                constexpr int NS = 18;                  // slots per px-block: n = 6 * r + t
                h8 C[8];                                // window: the fragment row 1 uses in slot n sits in C[n % 8], row 0 takes it in slot n + 6
                // input row 0 (row 0's tap row 0 only), slots 0..5.  (Passing these through the window's two free entries instead
                // saved no register and cost 200 cycles per step: a fragment register was re-loaded one slot after its last MFMA.)
                h8 Z[2];
                C[0] = Bnext[0]; Z[0] = Bnext[1];
                u32x4 pend = (u32x4){0u, 0u, 0u, 0u};
                int pend_row = 0, pend_hh = 0;

                auto half = [&](auto q_c) __attribute__((always_inline)) {
                    constexpr int q = decltype(q_c)::value;
                    f4 acc[4][2];                       // [co-block][row]
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int r = 0; r < 2; ++r) acc[m][r] = (f4){bias[m][0], bias[m][1], bias[m][2], bias[m][3]};      // (no instruction: the first MFMA reads them as its C operand)
                    // the pending pair of rows (previous px-block): where they go and which of them are kept
                    const int p_q = e_q;
                    const int p_base[2] = {pend_base(0), pend_base(1)};
                    const bool p_ok[2] = {pend_ok(0), pend_ok(1)};
#pragma unroll
                    for (int n = 0; n < NS; ++n) {
                        const int r = n / 6;
                        // the reads of the NEXT slot (the first slot of the next px-block / of the next step at the end)
                        if (n + 1 < NS) {
                            C[(n + 1) & 7] = load_f((n + 1) / 6 + 1, (n + 1) % 6, q);
                            if (n + 1 < 6) Z[(n + 1) & 1] = load_f(0, n + 1, q);
                        } else if (q == 0) {
                            C[0] = load_f(1, 0, 1);
                            Z[0] = load_f(0, 0, 1);
                        } else KPD_BNEXT {
                            const int nb0 = ring_row(R0 + KP_RPS), nb1 = ring_row(R0 + KP_RPS + 1);
                            Bnext[0] = *(const h8*)(smem + nb1 + roff[0][0]);      // next step: input row 1, t = 0, px-block 0
                            Bnext[1] = *(const h8*)(smem + nb0 + roff[0][0]);      //            input row 0
                        }
                        // two of the step's four DMA pieces per px-block; the pending rows' four pieces under slots 2..16
                        if (n == 1 || n == 3 || (n == 5 && q == 0)) {
                            constexpr int kq = 3 * q;
                            const int k = kq + (n >> 1);              // three under the first px-block, two (B: one) under the second
                            if constexpr (!kpd_no_dma)
                                if (k < kp_dma_count(ROLE)) dma_piece_k(KP_RPS * s + 6, k, dma_needed);
                        }
                        if (n == 4 || n == 8 || n == 12 || n == 16) {
                            if constexpr (kpd_epi_off(ROLE)) { KPD_KEEP("v"(pend), "s"(p_base[0]), "s"(pend_row + pend_hh)) }
                            else put(role_c, pend, p_q, pend_hh, p_base[pend_row], p_ok[pend_row]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (n == 2 || n == 6 || n == 10 || n == 14) {
                            const int p = (n - 2) / 4;
                            pend_row = p >> 1; pend_hh = p & 1;
                            if constexpr (kpd_epi_off(ROLE)) { KPD_KEEP("v"(racc[2 * (p & 1)][p >> 1]), "v"(racc[2 * (p & 1) + 1][p >> 1])) }
                            else pend = epi(racc, p >> 1, p & 1);
                        }
                        h8 op0 = r == 0 ? Z[n & 1] : C[(n - 6) & 7], op1 = C[n & 7];
                        KPD_OVERRIDE_OPERANDS(op0, op1, C[n & 7], (r == 0 ? Z[n & 1] : C[(n - 6) & 7]))
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            if constexpr (KP_MFMA_ORDER == 1) {
                                // "snake": the operand that stays when the weights fragment changes is the pixels' (op0 op1 | op1 op0 | ...)
                                if (m & 1) { acc[m][1] = MFMA16(wf[n][m], op1, acc[m][1]); acc[m][0] = MFMA16(wf[n][m], op0, acc[m][0]); }
                                else { acc[m][0] = MFMA16(wf[n][m], op0, acc[m][0]); acc[m][1] = MFMA16(wf[n][m], op1, acc[m][1]); }
                            } else if constexpr (KP_MFMA_ORDER == 2) {
                                acc[m][0] = MFMA16(wf[n][m], op0, acc[m][0]);      // pixels constant over four MFMAs (second row below)
                            } else {
                                acc[m][0] = MFMA16(wf[n][m], op0, acc[m][0]);
                                acc[m][1] = MFMA16(wf[n][m], op1, acc[m][1]);
                            }
                        }
                        if constexpr (KP_MFMA_ORDER == 2) {
#pragma unroll
                            for (int m = 0; m < 4; ++m) acc[m][1] = MFMA16(wf[n][m], op1, acc[m][1]);
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x2, KP_VALU_PER_MFMA, 0);
                        }
                    }
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            asm volatile("" : "+v"(acc[m][r]));
                            racc[m][r] = acc[m][r];
                        }
                    e_R = R0; e_q = q; e_live = true;
                };
                half(std::integral_constant<int, 0>{});
                half(std::integral_constant<int, 1>{});
            };

            const int s_first = ROLE ? KP_LAG : 0, s_last = ROLE ? n_steps : SA;      // this role's active steps
            if constexpr (ROLE == 1)
                for (int s = 0; s < KP_LAG; ++s) idle_step(s);
            {
                KPD_OPERANDS
                KPD_BNEXT {
                const int nb0 = ring_row(0), nb1 = ring_row(1);
                Bnext[0] = *(const h8*)(smem + nb1 + roff[0][0]);
                Bnext[1] = *(const h8*)(smem + nb0 + roff[0][0]);
                }
            }
            for (int s = s_first; s < s_last; ++s) {
                KPD_STEP_BEGIN
                step(s);
                KPD_STEP_END
                step_end(std::true_type{});
            }
            flush();                                     // A: its last results still have to reach the mid ring; B: its last row of the unit
            if constexpr (ROLE == 0)
                for (int s = SA; s < n_steps; ++s) idle_step(s);
            u += G;
            if (u >= a.n_units) break;
            unit_setup(u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    };
    KPD_LOOP_BEGIN
    if (role == 0) life(std::integral_constant<int, 0>{});
    else life(std::integral_constant<int, 1>{});
    KPD_EXIT
}

template __global__ void k_pair<false, false>(const PairArgs);
template __global__ void k_pair<true, false>(const PairArgs);
template __global__ void k_pair<false, true>(const PairArgs);
template __global__ void k_pair<true, true>(const PairArgs);

int pair_lds_bytes() { return KP_LDS; }

int prepare_pair_kernels()
{
    int rc = 0;
    for (const void* f : {(const void*)k_pair<false, false>, (const void*)k_pair<true, false>, (const void*)k_pair<false, true>, (const void*)k_pair<true, true>})
        rc |= (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, KP_LDS);
    return rc;
}

int launch_pair(const PairArgs& a, int grid, void* stream)
{
    launch_prepare();
    const bool gut = a.col_ok != nullptr || a.gut_period > 0;
    if (a.unit_slopes && !gut) hipLaunchKernelGGL((k_pair<true, false>), dim3(grid), dim3(64 * KP_NW), KP_LDS, (hipStream_t)stream, a);
    else if (!gut) hipLaunchKernelGGL((k_pair<false, false>), dim3(grid), dim3(64 * KP_NW), KP_LDS, (hipStream_t)stream, a);
    else if (a.unit_slopes) hipLaunchKernelGGL((k_pair<true, true>), dim3(grid), dim3(64 * KP_NW), KP_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_pair<false, true>), dim3(grid), dim3(64 * KP_NW), KP_LDS, (hipStream_t)stream, a);
    return launch_status();
}

}  // namespace reve
