// Two consecutive body layers per launch, each evaluated by Winograd's minimal filtering F(2,3) ALONG THE ROW, nested with the
// direct sum over the three tap rows (gfx950).  Replaces what k_pair (kernels_pair.hip) replaces in the reference
// (reve-shared/src/lib.rs:134-147: the realesrgan-ncnn-vulkan subprocess, whose Vulkan backend may itself evaluate these
// 64 -> 64 layers in a Winograd domain); behind reve_set_option("winograd", 1).
//
// Arithmetic.  A tile = two neighbouring output pixels (x = 2t, 2t + 1) of one row; its four input pixels d0..d3 (x - 1 .. x + 2)
// of a tap row are transformed per channel to V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3, each ONE packed fp16
// add (v_pk_add_f16: the correctly rounded fp16 of the exact sum — a single-level transform needs no fp32 intermediate, which is
// why only the row direction is transformed: the 2-D F(2x2,3x3) needs two levels, i.e. fp32 VALU work of 24 adds per tile and
// channel in, 24 out, on a SIMD that one 512-register wave drives at 4 cycles per instruction).  The weights of tap row dy
// become U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2 (host, fp32, stored fp16: model.cpp
// pack_body_wino), the four products M_xi = sum over (dy, ci) of U_xi * V_xi are MFMA sums in fp32 (the bias rides in M1's
// accumulator), and the two outputs are y0 = M0 + (M1 + M2), y1 = (M1 - M2) - M3 (fp32), then fp16 round and PReLU as
// everywhere.  192 MFMAs per 64 pixels and layer instead of 288.  oracle/srvgg_ref.c mode 4 restates exactly this (everything but
// the MFMA's internal summation order).
//
// Shape.  k_pair's: a workgroup (4 waves, one per SIMD) owns a strip of 62 output columns of the second layer and rolls down a
// segment of rows, two rows per step; waves 0, 1 compute the first layer from an 8-row ring of input rows (LDS-DMA, 66 columns,
// filled two steps ahead) into a second 8-row ring, waves 2, 3 the second layer from that ring, three steps behind, storing to
// the arena.  What differs:
//   * the two waves of a layer split the OUTPUT CHANNELS (32 each: a layer's U is 96 KiB, 192 registers per wave, parked in
//     AGPRs), and each covers all 64 columns = 32 tiles = two MFMA column blocks q;
//   * a step walks the column blocks one after the other: for block q the four input rows of the step, channel half by channel
//     half, each read once (four ds_read_b128, issued a whole block ahead of their use), transformed (sixteen packed adds) and fed
//     to the one or two output rows it contributes to — 8 / 16 / 16 / 8 MFMAs; the sums of block q (2 rows x 4 xi x 2 co-blocks)
//     are output-transformed, rounded and written under the sixteen-MFMA blocks of the other column block (two accumulator sets
//     alternate by name);
//   * the rings keep every other pixel pair swapped and xor the chunk with column bits 2, 3 (kw_ring_off): the tile reads are free
//     of LDS bank conflicts.
// The kernel is bound by instruction ISSUE, not by the MFMA pipe: a wave alone on its SIMD issues one instruction of any class per
// 4 cycles, an MFMA takes two turns (scripts/ubench/valu_issue.hip).  Per step and wave: 192 MFMAs + ~720 other instructions (256
// packed transforms, 128 fp32 sums, 32 conversions, 64 PReLU, 64 reads, 32 read addresses, 8 stores or ring writes, 4-5 DMA pieces,
// ~130 scalar) = 8 x 192 + 4 x 720 = 4,416 cycles against 3,072 of MFMA pipe; measured 4,660-4,820 (docs/LAB_NOTES.md R4-6).
// Everything that is not an MFMA is therefore counted: roles and phases live OUTSIDE the step loop (no copies where paths merge),
// nothing is masked in the hot path, and the other instructions are spread three behind every MFMA.
#include <type_traits>

#ifndef KW_DMA_AUX
#define KW_DMA_AUX 2          // input rows are read once from HBM: streaming loads, as in k_pair
#endif
#ifndef KW_STORE_AUX
#define KW_STORE_AUX 0
#endif
#ifndef KW_RIDER0
#define KW_RIDER0 2             // the first of a column block's eight blocks that carries an epilogue piece of the other column block: the four
#endif                          // blocks of sixteen MFMAs (tap rows 1 and 2 x channel halves) carry the four pieces
#ifndef KW_VALU_PER_MFMA_16
#define KW_VALU_PER_MFMA_16 3   // VALU instructions of a rider piece placed behind each of the last eight MFMAs of its block (+ 1 behind every other)
#endif

#include "kernels_dev.h"

// Timing-only instrumentation lives in kernels_wino_diag.inc and exists in diagnostic builds only (scripts/ablate_pair.sh); a
// product build sees the empty hooks below and must compile with that file absent.
#if (defined(STAMPS) || defined(KWD_NO_DMA) || defined(KWD_NO_WAIT) || defined(KWD_NO_STORE) || defined(KWD_NO_EPI) || defined(KWD_NO_PRELU)) && !defined(REVE_DIAGNOSTIC_BUILD)
#error "STAMPS / KWD_* are timing-only diagnostic switches: build them through scripts/ablate_pair.sh (-DREVE_DIAGNOSTIC_BUILD), never into libreve_hip.so"
#endif
#ifdef REVE_DIAGNOSTIC_BUILD
#include "kernels_wino_diag.inc"
#else
constexpr bool kwd_no_dma = false, kwd_no_wait = false, kwd_no_store = false, kwd_no_epi = false, kwd_no_prelu = false;
#define KWD_ENTRY
#define KWD_LOOP_BEGIN
#define KWD_STEP_BEGIN
#define KWD_STEP_END
#define KWD_WAIT_BEGIN
#define KWD_WAIT_END(active)
#define KWD_EXIT
#endif

namespace reve {

namespace {
constexpr int KW_NW = 4;
constexpr int KW_COLS = PAIR_COLS;                        // columns computed per row and layer: 32 tiles
constexpr int KW_ROW_BYTES = KW_COLS * PIX_BYTES;         // a mid-ring row: 8,192 B
constexpr int KW_RING = 8;
constexpr int KW_RPS = 2;                                 // rows per step
constexpr int KW_LAG = 3;                                 // steps the second layer runs behind the first
constexpr int KW_IN_COLS = KW_COLS + 2;
constexpr int KW_PPR = (KW_IN_COLS + 7) / 8;              // DMA pieces per input row (9)
constexpr int KW_IN_ROW_BYTES = KW_PPR * 1024;            // 9,216
constexpr int KW_MID_OFF = KW_RING * KW_IN_ROW_BYTES;     // 73,728
constexpr int KW_LDS = KW_MID_OFF + KW_RING * KW_ROW_BYTES + 1024;       // + what tile 31 of the last ring row reads beyond column 63
constexpr int KW_NFRAG = 3 * 4 * 2 * 2;                   // U fragments per wave: [tap row][xi][channel half][co-block]
static_assert(KW_LDS <= 160 * 1024, "LDS budget of a CU");
constexpr int kw_in_row_off(int rho) { return (rho & (KW_RING - 1)) * KW_IN_ROW_BYTES; }
// Where a ring row keeps pixel j's 16-byte chunk c.  A tile's lane reads pixels 2t + i for ONE i, so all lanes of a ds_read_b128
// ask for pixels of one parity; with pixels in order they would all fall into one 128-byte half of the 256-byte bank row: 2-way
// conflicts whatever the chunk swizzle (measured: SQ_LDS_BANK_CONFLICT half of SQ_LDS_IDX_ACTIVE, the step LDS-bound).  So the
// two pixels of every other pair are swapped (pixel j sits in slot j ^ (j >> 1 & 1)): tiles of even and odd t use different halves;
// and the chunk is xor-ed with bits 2, 3 of j, which tells apart the four tiles of one parity and chunk in each 16-lane group of
// the read ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... : MI355X_MICROARCH.md, LDS): sixteen lanes, sixteen 16-byte slots.
__host__ __device__ constexpr int kw_pix_slot(int j) { return (j & ~1) | (((j >> 1) ^ j) & 1); }
__host__ __device__ constexpr int kw_chunk_pos(int j, int c) { return c ^ ((j >> 1) & 6); }
__host__ __device__ constexpr int kw_ring_off(int j, int c) { return kw_pix_slot(j) * PIX_BYTES + 16 * kw_chunk_pos(j, c); }
// a step's DMA pieces as in k_pair: two rows x nine = 18; wave w takes column groups 2w, 2w + 1 of both rows, the ninth group of
// row 0 goes to wave 0 and of row 1 to wave 1: five pieces per step for the first layer's waves, four (+ 8 stores) for the second's
constexpr int KW_DMA_PER_WAVE = 5;
constexpr int kw_dma_count(int role) { return role == 0 ? KW_DMA_PER_WAVE : KW_DMA_PER_WAVE - 1; }
}  // namespace

// GUT: the frame is a canvas of several planes (tiled frames, small frames that share their launches: Engine::configure): the gutter
// columns (a.col_ok) and rows (a.gut_*) between planes are each plane's zero padding — the first layer leaves zeros there, the
// second stores nothing (as in k_pair).  Its own instantiations: whole frames carry none of it.
template <bool UNIT_SLOPES, bool GUT>
__global__ void __launch_bounds__(64 * KW_NW, 1) k_wino(const PairArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    KWD_ENTRY
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 1;                // 0: first layer (A), 1: second layer (B)
    const int ch = wave & 1;                   // which 32 of the 64 output channels
    const int pl = lane & 15, g = lane >> 4;

    // ---- this wave's U fragments, straight from global memory (every CU reads the same 192 KiB: L2-resident)
    h8 U[3][4][2][2];
    {
        const h8* wp = (const h8*)(role ? a.wpack[1] : a.wpack[0]) + (size_t)ch * KW_NFRAG * 64 + lane;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int m = 0; m < 2; ++m) U[dy][xi][hf][m] = wp[(((dy * 4 + xi) * 2 + hf) * 2 + m) * 64];
    }
    const uint16_t* bias_p = role ? a.bias[1] : a.bias[0];
    const uint16_t* slope_p = role ? a.slope[1] : a.slope[0];
    f4 biasv[2];
    h8 slope8;
    {
        const h4 b0 = *(const h4*)(bias_p + 32 * ch + 4 * g), b1 = *(const h4*)(bias_p + 32 * ch + 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) { biasv[0][r] = (float)b0[r]; biasv[1][r] = (float)b1[r]; }
        const h4 s0 = *(const h4*)(slope_p + 32 * ch + 4 * g), s1 = *(const h4*)(slope_p + 32 * ch + 16 + 4 * g);
        slope8 = __builtin_shufflevector(s0, s1, 0, 1, 2, 3, 4, 5, 6, 7);
    }

    // ---- lane-constant address parts.  Tile (q, pl) = columns 32q + 2pl, + 1 of the layer's 64; it reads ring columns
    // 32q + 2pl + i, i = 0..3 (ring column j of a role's input <-> its output column j - 1): this lane the 16-byte chunk 4hf + g.
    int doff[2][4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int i = 0; i < 4; ++i) doff[hf][i] = kw_ring_off(2 * pl + i, 4 * hf + g);
    // first layer: its piece (channels 32ch + 8g ..) of column 2pl + jj goes where the second layer's reads expect chunk 4ch + g
    int woff[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) woff[jj] = kw_ring_off(2 * pl + jj, 4 * ch + g);
    // second layer: arena pixel (1, 1 + 2pl + jj), byte 64ch + 16g
    int slane[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) slane[jj] = (a.Wp + 1 + 2 * pl + jj) * PIX_BYTES + 64 * ch + 16 * g;

    const int plane_bytes = a.Hp * a.Wp * PIX_BYTES;
    auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, plane_bytes, 0x00020000);
    auto no_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0, 0x00020000);       // zero bytes: loads fetch nothing

    const int G = gridDim.x;
    const int bid = blockIdx.x;
    int u = ((G & 7) == 0) ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;      // blocks of one XCD take neighbouring units

    int x0 = 0, y0 = 0, y1 = 0, NA = 0, SA = 0, n_steps = 0;
    unsigned cm[2][2] = {{0u, 0u}, {0u, 0u}};      // column masks [q][jj]: A zeroes what lies outside the frame (a strip at the frame's edge only: `edge`)
    bool edge = false;
    // B stores its 62 valid columns: the lane's arena offset of piece (q, jj), or 2^31 — beyond any plane whatever row offset is
    // added (the row rides in the store's scalar offset)
    unsigned svoff[2][2] = {{0u, 0u}, {0u, 0u}};
    int vcol[3] = {0, 0, 0};
    // The k-th LDS-DMA piece of this wave for input rows rho0, rho0 + 1 (k_pair's assignment): k < 4: column group 2 * wave + (k >> 1)
    // of row k & 1; k == 4 (waves 0, 1): the ninth group of row `wave`
    auto dma_piece_k = [&](int rho0, int k, bool needed) {
        const int ci = k >> 1, c = ci < 2 ? 2 * wave + ci : KW_PPR - 1, row = k < 4 ? (k & 1) : wave;
        int ar = y0 - 1 + rho0 + row;
        ar = ar < 0 ? 0 : (ar > a.Hp - 1 ? a.Hp - 1 : ar);
        dma16a<KW_DMA_AUX>(needed ? in_rsrc : no_rsrc, to_lds(smem + kw_in_row_off(rho0 + row) + c * 1024), vcol[ci], ar * a.Wp * PIX_BYTES);
    };
    auto unit_setup = [&](int un) {
        const int uu = a.reverse ? a.n_units - 1 - un : un;
        const int sy = uu / a.n_strips, sx = uu - sy * a.n_strips;
        x0 = sx * PAIR_VALID;
        y0 = sy * a.seg_h;
        y1 = y0 + a.seg_h < a.H ? y0 + a.seg_h : a.H;
        const int NB = y1 - y0;
        NA = NB + 2;
        const int SB = (NB + KW_RPS - 1) / KW_RPS;
        SA = (NA + KW_RPS - 1) / KW_RPS;
        n_steps = SB + KW_LAG;                       // = SA + 2
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int c = 32 * q + 2 * pl + jj;
                bool ok = role ? (c < PAIR_VALID && x0 + c < a.W) : (x0 - 1 + c >= 0 && x0 - 1 + c < a.W);
                if constexpr (GUT) {
                    int x = role ? x0 + c : x0 - 1 + c;
                    x = x < 0 ? 0 : (x > a.W - 1 ? a.W - 1 : x);
                    if (a.col_ok) ok = ok && a.col_ok[x] != 0;
                }
                cm[q][jj] = ok ? 0xffffffffu : 0u;
                svoff[q][jj] = ok ? (unsigned)(slane[jj] + 32 * q * PIX_BYTES) : 0x80000000u;
            }
        edge = (x0 - 1 < 0) | (x0 - 1 + KW_COLS > a.W);
        if constexpr (GUT) edge |= __builtin_amdgcn_ballot_w64((cm[0][0] & cm[0][1] & cm[1][0] & cm[1][1]) == 0u) != 0ull;      // a gutter column in the strip
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const int c = ci < 2 ? 2 * wave + ci : KW_PPR - 1;
            // this lane fills bytes 16 * lane .. of the piece: pixel slot lane >> 3, chunk position lane & 7 (kw_ring_off read backwards:
            // the slot permutation is its own inverse, the chunk xor too)
            int j = kw_pix_slot(8 * c + (lane >> 3));
            j = j < KW_IN_COLS ? j : KW_IN_COLS - 1;
            int ac = x0 - 1 + j;
            ac = ac < 0 ? 0 : (ac > a.Wp - 1 ? a.Wp - 1 : ac);
            vcol[ci] = ac * PIX_BYTES + 16 * kw_chunk_pos(j, lane & 7);
        }
#pragma unroll
        for (int blk = 0; blk < 3; ++blk)
#pragma unroll
            for (int k = 0; k < KW_DMA_PER_WAVE; ++k)
                if (k < KW_DMA_PER_WAVE - 1 || role == 0) dma_piece_k(KW_RPS * blk, k, true);
    };

    unit_setup(u);
    // (the U loads, bias and slopes are waited for here, together with the first rows)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(biasv[m][r]));
    asm volatile("" : "+v"(slope8));
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int m = 0; m < 2; ++m) asm volatile("" : "+a"(U[dy][xi][hf][m]));      // parked in the accumulator file: the MFMA reads its A operand from there
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // the input transform of one tile's four pixels (this lane's eight channels): four packed fp16 instructions per xi.  a - b is
    // written fma(b, -1, a) with the -1 in a register the compiler cannot see through: `a - b` on fp16 vectors becomes eight
    // v_sub_f16 and four v_pack_b32_f16 (there is no v_pk_sub_f16 and hipcc does not use v_pk_add_f16's neg modifiers); the fma
    // rounds once, like the subtraction.
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 negone2 = (h2)(_Float16)-1.0f;
    asm volatile("" : "+v"(negone2));
    const h8 negone = __builtin_shufflevector(negone2, negone2, 0, 1, 0, 1, 0, 1, 0, 1);
    auto transform = [&](const h8 (&d)[4], h8 (&v)[4]) {
        v[3] = __builtin_elementwise_fma(d[3], negone, d[1]);      // (the last pixel read first: one wait for the four reads)
        v[0] = __builtin_elementwise_fma(d[2], negone, d[0]);
        v[1] = d[1] + d[2];
        v[2] = __builtin_elementwise_fma(d[1], negone, d[2]);
    };
    // y0 / y1 of one output row and column block: M0 + (M1 + M2), (M1 - M2) - M3 per co-block, rounded to fp16, PReLU.
    // (Element by element, and the file is built with -fno-slp-vectorize: written on f4 the sums become v_pk_add_f32, which beside
    // MFMAs costs several times the two v_add_f32 it replaces: scripts/ubench/valu_issue.hip.)  Both pixels' pieces are the same
    // 28 instructions (two sums, the rounding, PReLU), so that each rides under eight MFMAs at four and three apiece.
    // (v_fma_mixlo / mixhi_f16 would fuse the last sum with the rounding, but take TWO issue turns each, 16 cycles per channel pair
    // against 12 for two adds and a v_cvt_pk_f16_f32: scripts/ubench/valu_issue.hip.)
    auto finish = [&](f4 (&M)[4][2], int jj) {
        h8 o;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (jj == 0) o[4 * m + r] = (_Float16)(M[0][m][r] + (M[1][m][r] + M[2][m][r]));
                else o[4 * m + r] = (_Float16)((M[1][m][r] - M[2][m][r]) - M[3][m][r]);
            }
        if constexpr (kwd_no_prelu) return __builtin_bit_cast(u32x4, o);
        return __builtin_bit_cast(u32x4, UNIT_SLOPES ? prelu8_unit_slopes(o, slope8) : prelu8(o, slope8));
    };
    // (The first layer writes every piece as computed; what must be zero — the second layer's padding: columns and rows outside the
    // frame — is overwritten at the end of the step, `zero_outside`, in the rare steps that have any: four v_and per piece would be
    // 32 issue turns per step, a uniform branch per piece costs more than it saves.)
    auto put = [&](auto role_c, u32x4 v, int q, int jj, int base, bool ok) {
        if constexpr (decltype(role_c)::value == 0) {
            *(u32x4*)(smem + KW_MID_OFF + base + woff[jj] + 32 * q * PIX_BYTES) = v;
        } else {
            auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, ok ? plane_bytes : 0, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, kwd_no_store ? 0x80000000u : svoff[q][jj], base, KW_STORE_AUX);
        }
    };
    auto zero_outside = [&](int q, int jj, int base, bool ok) {
        if (!(cm[q][jj] && ok)) *(u32x4*)(smem + KW_MID_OFF + base + woff[jj] + 32 * q * PIX_BYTES) = (u32x4){0u, 0u, 0u, 0u};
    };

    // A role's whole life in the launch.  The roles are separated OUTSIDE the loops and a unit is walked in phases (second layer:
    // KW_LAG idle steps, then its active steps; first layer: its active steps, then two idle ones): a lone wave issues one
    // instruction of any class per 4 cycles, so what a loop body merges at its end (the copies that bring two paths' registers
    // together) is paid in full — 135 instructions per step when the phases were branches inside one loop.
    auto life = [&](auto role_c) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(role_c)::value;
        auto ring_row = [&](int R) {
            if constexpr (ROLE == 0) return kw_in_row_off(R);
            else return KW_MID_OFF + (R & (KW_RING - 1)) * KW_ROW_BYTES;
        };
        // where output row R of the role goes, and whether it is kept
        auto row_base = [&](int R) {
            if constexpr (ROLE == 0) return (R & (KW_RING - 1)) * KW_ROW_BYTES;
            else return ((y0 + R) * a.Wp + x0) * PIX_BYTES;
        };
        // Gutter rows are gut_first + k * gut_period.  A role asks about its rows in ascending order, every pair of rows twice (as the
        // pending pair of one step, as the new pair of the step before: ... 84, 85, 86, 87, 86, 87, 88, 89 ...): gut_next is the first
        // gutter row >= the highest row asked about so far minus one (k_pair's scheme: one compare-and-add per query)
        int gut_next = 0;
        auto gut_start = [&]() {
            if constexpr (GUT) {
                const int y_first = (ROLE ? y0 : y0 - 1) - 3;
                const int k = (a.gut_period > 0 && y_first > a.gut_first) ? (y_first - a.gut_first + a.gut_period - 1) / a.gut_period : 0;
                gut_next = a.gut_period > 0 ? a.gut_first + k * a.gut_period : 0x7fffffff;
            }
        };
        auto is_gutter = [&](int y) {
            if constexpr (GUT) {
                gut_next += gut_next < y - 1 ? a.gut_period : 0;
                return y == gut_next;
            }
            return false;
        };
        // (is_gutter is asked unconditionally and combined without short-circuit: it has a side effect)
        auto row_ok = [&](int R, bool live) {
            if constexpr (ROLE == 0) {
                const int ya = y0 - 1 + R;
                const bool gut = is_gutter(ya);
                return (bool)(live & (ya >= 0) & (ya < a.H) & !gut);
            } else {
                const int yb = y0 + R;
                const bool gut = is_gutter(yb);
                return (bool)(live & (R >= 0) & (yb < y1) & !gut);
            }
        };
        // the end of a step: the DMA pieces of the PREVIOUS step (read in the next one) have landed; this wave's LDS writes are
        // done.  Younger than those pieces: everything of this step (A: 5 DMA pieces; B: 4 + its 8 stores when active).
        auto step_end = [&](auto active_c) {
            constexpr bool active = decltype(active_c)::value;
            KWD_WAIT_BEGIN
            if (kwd_no_wait || kwd_no_dma || kwd_no_epi) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kw_dma_count(ROLE) + (ROLE && active ? 2 * KW_RPS * 2 : 0)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            KWD_WAIT_END(active)
        };
        auto idle_step = [&](int s) {
            const bool dma_needed = KW_RPS * s + 6 <= NA + 1;
#pragma unroll
            for (int k = 0; k < kw_dma_count(ROLE); ++k) dma_piece_k(KW_RPS * s + 6, k, dma_needed);
            step_end(std::false_type{});
        };

        for (;;) {
            // the sums of the two output rows of a step, per column block (the set of block q is finished under the MFMAs of the
            // other block): [q][row][xi][co-block]
            f4 acc[2][2][4][2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[q][r][xi][m] = (f4){0.f, 0.f, 0.f, 0.f};
            int e_R = -2;                 // first row of the pair of rows whose column block 1 is pending in acc[1]
            bool e_live = false;
            gut_start();
            // the transformed pixels of the block being multiplied / of the next one ([block & 1][xi]); V[0] of a step's first
            // block is built at the end of the step before (of the first active step: ahead of the loop)
            h8 V[2][4], D[4];

            // the pending column block 1 of rows e_R, e_R + 1 (no MFMAs to hide under: end of a role's work in this unit)
            auto flush = [&]() {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const bool ok = row_ok(e_R + r, e_live);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        put(role_c, finish(acc[1][r], jj), 1, jj, row_base(e_R + r), ok);
                        if constexpr (ROLE == 0) zero_outside(1, jj, row_base(e_R + r), ok);
                    }
                }
            };

            auto step = [&](int s) __attribute__((always_inline)) {
                const int R0 = ROLE ? KW_RPS * (s - KW_LAG) : KW_RPS * s;       // first row of this step (of the role's output rows = of its input ring rows)
                const bool dma_needed = KW_RPS * s + 6 <= NA + 1;               // input rows 2s+6, 2s+7 exist for this unit
                int rb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) rb[i] = ring_row(R0 + i);
                const int nrb = ring_row(R0 + KW_RPS), nrb1 = ring_row(R0 + KW_RPS + 1);      // the next step's first rows (this step's last)
                // the four pieces of each pending set: where they go
                const int pb1[2] = {row_base(e_R), row_base(e_R + 1)};
                const bool pk1[2] = {row_ok(e_R, e_live), row_ok(e_R + 1, e_live)};
                const int pb0[2] = {row_base(R0), row_base(R0 + 1)};
                const bool pk0[2] = {row_ok(R0, true), row_ok(R0 + 1, true)};
                auto block = [&](auto q_c, auto blk_c) __attribute__((always_inline)) {
                    {
                        constexpr int q = decltype(q_c)::value, blk = decltype(blk_c)::value;
                        constexpr int i = blk >> 1, hf = blk & 1, cur = blk & 1, nxt = cur ^ 1;
                        // The pixels of the block after next are read FIRST (into fresh registers: they have this whole block to arrive —
                        // read behind the transform, into its registers, the reads of an eight-MFMA block had two MFMAs' time and the
                        // next block began by waiting for them), then the next block's pixels (D, read during the block before) are
                        // transformed.  (Blocks are numbered through the step and into the next: block 16 = the next step's first.)
                        h8 Dn[4];
                        {
                            constexpr int nb = 8 * q + blk + 2;
                            constexpr int ns = nb >> 4, nq = (nb >> 3) & 1, ni = (nb >> 1) & 3, nhf = nb & 1;
                            const int nrow = ns ? (ni == 0 ? nrb : nrb1) : rb[ni];
#pragma unroll
                            for (int k = 0; k < 4; ++k) Dn[k] = *(const h8*)(smem + nrow + doff[nhf][k] + 32 * nq * PIX_BYTES);
                        }
                        transform(D, V[nxt]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) D[k] = Dn[k];
                        // this wave's DMA pieces: three under the first column block, two (second layer: one) under the second
                        if (blk == 1 || blk == 3 || (blk == 5 && q == 0)) {
                            constexpr int k = 3 * q + (blk >> 1);
                            if constexpr (!kwd_no_dma && k < kw_dma_count(ROLE)) dma_piece_k(KW_RPS * s + 6, k, dma_needed);
                        }
                        // riders: the other column block's four pieces (2 rows x 2 pixels), one per block from the second block on
                        if constexpr (blk >= KW_RIDER0 && blk < KW_RIDER0 + 4) {
                            constexpr int p = blk - KW_RIDER0, r = p >> 1, jj = p & 1;
                            if constexpr (!kwd_no_epi) {
                                if (q == 0) put(role_c, finish(acc[1][r], jj), 1, jj, pb1[r], pk1[r]);
                                else put(role_c, finish(acc[0][r], jj), 0, jj, pb0[r], pk0[r]);
                            } else {
                                asm volatile("" ::"v"(acc[q ^ 1][r][0][0]), "v"(acc[q ^ 1][r][1][0]), "v"(acc[q ^ 1][r][2][1]), "v"(acc[q ^ 1][r][3][1]));
                            }
                        }
                        constexpr int n_mfma = (i == 1 || i == 2) ? 16 : 8;
#pragma unroll
                        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                            for (int m = 0; m < 2; ++m)
#pragma unroll
                                for (int r = 0; r < 2; ++r) {
                                    const int dyr = i - r, dy = dyr < 0 ? 0 : (dyr > 2 ? 2 : dyr);
                                    if (dyr < 0 || dyr > 2) continue;       // (input row i is tap row i - r of output row r, if it is one at all)
                                    const f4 c0 = (dy == 0 && hf == 0) ? (xi == 1 ? biasv[m] : (f4){0.f, 0.f, 0.f, 0.f}) : acc[q][r][xi][m];
                                    acc[q][r][xi][m] = MFMA16(U[dy][xi][hf][m], V[cur][xi], c0);
                                }
                        // The interleave.  A step is bound by instruction issue (8 cycles an MFMA, 4 anything else; an MFMA with nothing
                        // behind it still takes its 16 cycles of pipe), so everything else is spread EVENLY behind the MFMAs, about three
                        // each: the read addresses and the four reads behind the first three, the transform (16 VALU) behind the next five, a rider
                        // piece (28 VALU) behind the other eight of a sixteen-MFMA block, its store or LDS write last.
                        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
                        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
#pragma unroll
                        for (int j = 3; j < 8; ++j) {
                            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
                        }
#pragma unroll
                        for (int j = 8; j < n_mfma; j += 2) {
                            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x2, KW_VALU_PER_MFMA_16 + 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x2, KW_VALU_PER_MFMA_16, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                auto column_block = [&](auto q_c) __attribute__((always_inline)) {
                    block(q_c, std::integral_constant<int, 0>{}); block(q_c, std::integral_constant<int, 1>{});
                    block(q_c, std::integral_constant<int, 2>{}); block(q_c, std::integral_constant<int, 3>{});
                    block(q_c, std::integral_constant<int, 4>{}); block(q_c, std::integral_constant<int, 5>{});
                    block(q_c, std::integral_constant<int, 6>{}); block(q_c, std::integral_constant<int, 7>{});
                };
                column_block(std::integral_constant<int, 0>{});
                column_block(std::integral_constant<int, 1>{});
                if constexpr (ROLE == 0 && !kwd_no_epi)
                    if (__builtin_expect(edge | !(pk1[0] & pk1[1] & pk0[0] & pk0[1]), 0)) {
#pragma unroll
                        for (int r = 0; r < 2; ++r)
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) { zero_outside(1, jj, pb1[r], pk1[r]); zero_outside(0, jj, pb0[r], pk0[r]); }
                    }
                e_R = R0; e_live = true;
            };

            const int s_first = ROLE ? KW_LAG : 0, s_last = ROLE ? n_steps : SA;      // this role's active steps
            if constexpr (ROLE == 1)
                for (int s = 0; s < KW_LAG; ++s) idle_step(s);
            {   // the first active step's first block: its pixels transformed, the second block's read
                const int r0 = ring_row(0);
#pragma unroll
                for (int k = 0; k < 4; ++k) D[k] = *(const h8*)(smem + r0 + doff[0][k]);
                transform(D, V[0]);
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) V[1][xi] = V[0][xi];
#pragma unroll
                for (int k = 0; k < 4; ++k) D[k] = *(const h8*)(smem + r0 + doff[1][k]);
            }
            for (int s = s_first; s < s_last; ++s) {
                KWD_STEP_BEGIN
                step(s);
                KWD_STEP_END
                step_end(std::true_type{});
            }
            if constexpr (ROLE == 0) {
                flush();                                  // A is done with this unit
                for (int s = SA; s < n_steps; ++s) idle_step(s);
            } else {
                flush();                                  // B's last column block of the unit
            }
            u += G;
            if (u >= a.n_units) break;
            unit_setup(u);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    };
    KWD_LOOP_BEGIN
    if (role == 0) life(std::integral_constant<int, 0>{});
    else life(std::integral_constant<int, 1>{});
    KWD_EXIT
}

template __global__ void k_wino<false, false>(const PairArgs);
template __global__ void k_wino<true, false>(const PairArgs);
template __global__ void k_wino<false, true>(const PairArgs);
template __global__ void k_wino<true, true>(const PairArgs);

int wino_lds_bytes() { return KW_LDS; }
int wino_ring_offset(int column, int chunk) { return kw_ring_off(column, chunk); }

int prepare_wino_kernels()
{
    int rc = 0;
    for (const void* f : {(const void*)k_wino<false, false>, (const void*)k_wino<true, false>, (const void*)k_wino<false, true>, (const void*)k_wino<true, true>})
        rc |= (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, KW_LDS);
    return rc;
}

int launch_wino(const PairArgs& a, int grid, void* stream)
{
    launch_prepare();
    const bool gut = a.col_ok != nullptr || a.gut_period > 0;
    if (a.unit_slopes && !gut) hipLaunchKernelGGL((k_wino<true, false>), dim3(grid), dim3(64 * KW_NW), KW_LDS, (hipStream_t)stream, a);
    else if (!gut) hipLaunchKernelGGL((k_wino<false, false>), dim3(grid), dim3(64 * KW_NW), KW_LDS, (hipStream_t)stream, a);
    else if (a.unit_slopes) hipLaunchKernelGGL((k_wino<true, true>), dim3(grid), dim3(64 * KW_NW), KW_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_wino<false, true>), dim3(grid), dim3(64 * KW_NW), KW_LDS, (hipStream_t)stream, a);
    return launch_status();
}

}  // namespace reve
