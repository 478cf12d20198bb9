// Two consecutive body layers per launch, each evaluated by Winograd's minimal filtering F(2,3) ALONG THE ROW, nested with the
// direct sum over the three tap rows (gfx950).  Replaces what k_pair (kernels_pair.hip) replaces in the reference
// (reve-shared/src/lib.rs:134-147: the realesrgan-ncnn-vulkan subprocess, whose Vulkan backend may itself evaluate these
// 64 -> 64 layers in a Winograd domain); behind reve_set_option("winograd", 1).
//
// Arithmetic.  A tile = two neighbouring output pixels (x = 2t, 2t + 1) of one row; its four input pixels d0..d3 (x - 1 .. x + 2)
// of a tap row are transformed per channel to V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3, each ONE packed fp16
// instruction (the correctly rounded fp16 of the exact sum — a single-level transform needs no fp32 intermediate, which is why
// only the row direction is transformed: the 2-D F(2x2,3x3) needs two levels, i.e. fp32 VALU work of 24 adds per tile and channel
// in, 24 out, on a SIMD that one 512-register wave drives at 4 cycles per instruction).  The weights of tap row dy become
// U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2 (host, fp32, stored fp16: model.cpp pack_body_wino), the four
// products M_xi = sum over (dy, ci) of U_xi * V_xi are MFMA sums in fp32 (the bias rides in M1's accumulator), and the two outputs
// are y0 = M0 + (M1 + M2), y1 = (M1 - M2) - M3 (fp32), then fp16 round and PReLU as everywhere.  192 MFMAs per 64 pixels and
// layer instead of 288.  oracle/srvgg_ref.c mode 4 restates exactly this (everything but the MFMA's internal summation order).
//
// Shape.  A workgroup (4 waves, one per SIMD) owns a strip of 30 output columns of the second layer and rolls down a segment of
// rows, ONE row per step; waves 0, 1 compute the first layer (32 columns = 16 tiles = one MFMA column block) from an 8-row ring of
// input rows (LDS-DMA, 34 columns, requested five steps ahead) into a 4-row ring in LDS, waves 2, 3 the second layer from that
// ring, five steps behind, storing to the arena.  The two waves of a layer split the OUTPUT CHANNELS (32 each: a layer's U is
// 96 KiB, 192 registers per wave, parked in AGPRs).
// A step STREAMS one input row: its pixels are read once (eight ds_read_b128 per lane), transformed once (32 packed
// instructions) and multiplied into the three output rows the row contributes to — tap row 0 of a new output row (the sums start
// here), tap row 1 of the row before, tap row 2 of the row before that, which is then finished.  The three accumulator sets in
// flight and the finished one — output-transformed, rounded and written under the next step's MFMAs — rotate by name over four
// copies of the step.  (The first version walked two rows per step over 64-column strips as k_pair does and transformed every
// input row twice; it was VALU-issue bound — a packed fp16 instruction occupies a lone wave's issue for 8 cycles — at 5,900-6,160
// cycles per 2 x 64 pixels against k_pair's 5,817: profiles/r04/ablation_wino_v2_two_rows_per_step.txt.)
// Per step and wave: 48 MFMAs, 8 ds_read_b128, 32 packed fp16 transforms, 32 fp32 adds, the epilogue of 32 x 32 values.
#include <type_traits>

#ifndef KW_DMA_AUX
#define KW_DMA_AUX 2          // input rows are read once from HBM: streaming loads, as in k_pair
#endif
#ifndef KW_STORE_AUX
#define KW_STORE_AUX 0
#endif
#ifndef KW_VALU_PER_MFMA
#define KW_VALU_PER_MFMA 3    // VALU instructions placed behind each MFMA (sched_group_barrier)
#endif

#include "kernels_dev.h"

// Timing-only instrumentation lives in kernels_wino_diag.inc and exists in diagnostic builds only (scripts/ablate_pair.sh); a
// product build sees the empty hooks below and must compile with that file absent.
#if (defined(STAMPS) || defined(KWD_NO_DMA) || defined(KWD_NO_WAIT) || defined(KWD_NO_STORE) || defined(KWD_NO_EPI)) && !defined(REVE_DIAGNOSTIC_BUILD)
#error "STAMPS / KWD_* are timing-only diagnostic switches: build them through scripts/ablate_pair.sh (-DREVE_DIAGNOSTIC_BUILD), never into libreve_hip.so"
#endif
#ifdef REVE_DIAGNOSTIC_BUILD
#include "kernels_wino_diag.inc"
#else
constexpr bool kwd_no_dma = false, kwd_no_wait = false, kwd_no_store = false, kwd_no_epi = false;
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
constexpr int KW_COLS = WINO_COLS;                        // columns computed per row and layer: 16 tiles
constexpr int KW_MID_ROW = KW_COLS * PIX_BYTES;           // 4,096
constexpr int KW_MID_RING = 4;
constexpr int KW_IN_COLS = KW_COLS + 2;
constexpr int KW_PPR = (KW_IN_COLS + 7) / 8;              // DMA pieces per input row (5: the fifth carries columns 32, 33)
constexpr int KW_IN_ROW = KW_PPR * 1024;                  // 5,120
constexpr int KW_IN_RING = 8;
constexpr int KW_LEAD = 5;                                // input row s + KW_LEAD is requested in step s
constexpr int KW_LAG = 5;                                 // the second layer takes first-layer row a in step a + KW_LAG
constexpr int KW_MID_OFF = KW_IN_RING * KW_IN_ROW;        // 40,960
constexpr int KW_LDS = KW_MID_OFF + KW_MID_RING * KW_MID_ROW + 1024;     // + what tile 15 of the last ring row reads beyond column 31
constexpr int KW_NFRAG = 3 * 4 * 2 * 2;                   // U fragments per wave: [tap row][xi][channel half][co-block]
static_assert(KW_LDS <= 160 * 1024, "LDS budget of a CU");
static_assert(KW_LEAD + 2 <= KW_IN_RING, "a requested row must not land on one that is still read");
static_assert(KW_PPR == KW_NW + 1, "one DMA piece per wave and row, the fifth to wave 0");
// Vector-memory operations a wave issues per step, in program order: its DMA piece(s) of row s + KW_LEAD (wave 0: two), second
// layer: + two stores.  At the end of step s rows <= s + 2 have to have landed (row s + 1 is read from under step s already, by
// the prefetch of its first pixels), i.e. everything issued up to step s + 2 - KW_LEAD: all but the operations of the last
// KW_LEAD - 2 steps.
constexpr int kw_vm_per_step(int wave) { return wave == 0 ? 2 : (wave == 1 ? 1 : 3); }
}  // namespace

template <bool UNIT_SLOPES>
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

    // ---- lane-constant address parts.  Tile pl = columns 2pl, 2pl + 1 of the layer's 32; it reads ring columns 2pl + i, i = 0..3
    // (ring column j of a role's input <-> its output column j - 1): this lane the 16-byte chunk 4hf + g.
    int doff[2][4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = 2 * pl + i;
            doff[hf][i] = j * PIX_BYTES + 16 * ((4 * hf + g) ^ (j & 6));
        }
    // first layer: its piece (channels 32ch + 8g ..) of column 2pl + jj goes where the second layer's reads expect chunk 4ch + g
    int woff[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) woff[jj] = (2 * pl + jj) * PIX_BYTES + 16 * ((4 * ch + g) ^ ((2 * pl + jj) & 6));
    // second layer: arena pixel (1, 1 + 2pl + jj), byte 64ch + 16g
    int slane[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) slane[jj] = (a.Wp + 1 + 2 * pl + jj) * PIX_BYTES + 64 * ch + 16 * g;

    const int plane_bytes = a.Hp * a.Wp * PIX_BYTES;
    auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, plane_bytes, 0x00020000);
    auto no_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0, 0x00020000);       // zero bytes: loads fetch nothing
    auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, plane_bytes, 0x00020000);

    const int G = gridDim.x;
    const int bid = blockIdx.x;
    int u = ((G & 7) == 0) ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;      // blocks of one XCD take neighbouring units

    int x0 = 0, y0 = 0, NB = 0, n_steps = 0;
    unsigned cm[2] = {0u, 0u};      // column masks [jj]: the first layer zeroes what lies outside the frame, the second stores its 30 valid columns
    int vcol[2] = {0, 0};
    // DMA piece k of this wave for input row rho: k = 0: column group `wave` (8 columns); k = 1 (wave 0 only): the fifth group
    auto dma_piece = [&](int rho, int k) {
        const int c = k == 0 ? wave : KW_PPR - 1;
        int ar = y0 - 1 + rho;
        ar = ar < 0 ? 0 : (ar > a.Hp - 1 ? a.Hp - 1 : ar);
        dma16a<KW_DMA_AUX>(rho <= NB + 3 ? in_rsrc : no_rsrc, to_lds(smem + (rho & (KW_IN_RING - 1)) * KW_IN_ROW + c * 1024), vcol[k], ar * a.Wp * PIX_BYTES);
    };
    auto unit_setup = [&](int un) {
        const int uu = a.reverse ? a.n_units - 1 - un : un;
        const int sy = uu / a.n_strips, sx = uu - sy * a.n_strips;
        x0 = sx * WINO_VALID;
        y0 = sy * a.seg_h;
        const int y1 = y0 + a.seg_h < a.H ? y0 + a.seg_h : a.H;
        NB = y1 - y0;
        n_steps = NB + 3 + KW_LAG;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int c = 2 * pl + jj;
            const bool ok = role ? (c < WINO_VALID && x0 + c < a.W) : (x0 - 1 + c >= 0 && x0 - 1 + c < a.W);
            cm[jj] = ok ? 0xffffffffu : 0u;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = k == 0 ? wave : KW_PPR - 1;
            int j = 8 * c + (lane >> 3);
            j = j < KW_IN_COLS ? j : KW_IN_COLS - 1;
            int ac = x0 - 1 + j;
            ac = ac < 0 ? 0 : (ac > a.Wp - 1 ? a.Wp - 1 : ac);
            vcol[k] = ac * PIX_BYTES + 16 * ((lane & 7) ^ (j & 6));
        }
        for (int rho = 0; rho < KW_LEAD; ++rho) {
            dma_piece(rho, 0);
            if (wave == 0) dma_piece(rho, 1);
        }
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
    auto transform1 = [&](const h8 (&d)[4], h8 (&v)[4], int xi) {
        v[xi] = xi == 0 ? __builtin_elementwise_fma(d[2], negone, d[0]) : (xi == 1 ? d[1] + d[2] : (xi == 2 ? __builtin_elementwise_fma(d[1], negone, d[2]) : __builtin_elementwise_fma(d[3], negone, d[1])));
    };
    auto transform = [&](const h8 (&d)[4], h8 (&v)[4]) {
        v[0] = __builtin_elementwise_fma(d[2], negone, d[0]);
        v[1] = d[1] + d[2];
        v[2] = __builtin_elementwise_fma(d[1], negone, d[2]);
        v[3] = __builtin_elementwise_fma(d[3], negone, d[1]);
    };
    // y0 / y1 of one finished output row: M0 + (M1 + M2), (M1 - M2) - M3 per co-block, rounded to fp16, PReLU.
    // (Element by element, and the file is built with -fno-slp-vectorize: written on f4 the sums become v_pk_add_f32, which beside
    // MFMAs costs more than the two v_add_f32 it replaces.)
    auto finish = [&](const f4 (&M)[4][2], int jj) {
        h8 o;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = jj == 0 ? M[0][m][r] + (M[1][m][r] + M[2][m][r]) : (M[1][m][r] - M[2][m][r]) - M[3][m][r];
                o[4 * m + r] = (_Float16)y;
            }
        return __builtin_bit_cast(u32x4, UNIT_SLOPES ? prelu8_unit_slopes(o, slope8) : prelu8(o, slope8));
    };
    // a finished piece: first layer -> its ring (zero outside the frame: the second layer's padding), second -> arena
    auto put = [&](auto role_c, u32x4 v, int jj, int base, bool ok) {
        const unsigned m = cm[jj] & (ok ? 0xffffffffu : 0u);
        if constexpr (decltype(role_c)::value == 0) {
            v &= (u32x4){m, m, m, m};
            *(u32x4*)(smem + KW_MID_OFF + base + woff[jj]) = v;
        } else {
            const unsigned off = kwd_no_store ? 0x7fffffffu : (((unsigned)(base + slane[jj]) & m) | (0x7fffffffu & ~m));
            __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, (int)off, 0, KW_STORE_AUX);
        }
    };
    // LDS offset of row t of the role's input: the input ring / the ring the first layer writes
    auto ring_row = [&](auto role_c, int t) {
        if constexpr (decltype(role_c)::value == 0) return (t & (KW_IN_RING - 1)) * KW_IN_ROW;
        else return KW_MID_OFF + (t & (KW_MID_RING - 1)) * KW_MID_ROW;
    };

    KWD_LOOP_BEGIN
    for (;;) {
        // The sums of the output rows in flight, [xi][co-block]: a1 has received tap row 0, a2 tap rows 0 and 1; a3 is the row that
        // received its last tap in the step before: it is output-transformed, rounded and written under this step's first MFMAs.
        // A step's first block (channel half 0) adds in place (and starts the new row in a0), its second block hands every row
        // on to the next age: a3 <- a2 + ..., a2 <- a1 + ..., a1 <- a0 + ... (an MFMA's result need not overwrite its addend: the
        // rotation costs no instruction and no renaming).
        f4 a0[4][2], a1[4][2], a2[4][2], a3[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int m = 0; m < 2; ++m) a0[xi][m] = a1[xi][m] = a2[xi][m] = a3[xi][m] = (f4){0.f, 0.f, 0.f, 0.f};
        // the transformed pixels of the block being multiplied / of the next one ([block & 1][xi]) and the pixels read for the
        // block after that; V[0] and D of a step's first blocks are prepared under the step before
        h8 V[2][4], D[4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) V[0][xi] = V[1][xi] = D[xi] = (h8)(_Float16)0;

        // One step of one role: the row read (t) gives tap row 0 to a new output row, tap row 1 to the row before, tap row 2 to
        // the row before that.  Outside its rows (the second layer's first KW_LAG steps, the first layer's last three) a role
        // multiplies what happens to lie in its ring and the results go nowhere: no branch in the loop but the role's.
        auto step = [&](auto role_c, int t, int s) __attribute__((always_inline)) {
            const int rbn = ring_row(role_c, t + 1);
            // where the row that finished in the step before (output row t - 3) goes, and whether it is kept
            int base;
            bool ok;
            if constexpr (decltype(role_c)::value == 0) {
                const int o = t - 3, ya = y0 - 1 + o;
                base = (o & (KW_MID_RING - 1)) * KW_MID_ROW;
                ok = (o >= 0) & (o <= NB + 1) & (ya >= 0) & (ya < a.H);
            } else {
                const int b = t - 3;
                base = ((y0 + b) * a.Wp + x0) * PIX_BYTES;
                ok = (b >= 0) & (b < NB);
            }
            // ---- first block: channel half 0.  The second half's pixels (D, read during the block before) are transformed under
            // its first MFMAs; then the next row's first half is read into the same registers, a block ahead of its transform.
            transform(D, V[1]);
#pragma unroll
            for (int k = 0; k < 4; ++k) D[k] = *(const h8*)(smem + rbn + doff[0][k]);
            if constexpr (!kwd_no_dma) dma_piece(s + KW_LEAD, 0);
            if constexpr (!kwd_no_epi) {
                put(role_c, finish(a3, 0), 0, base, ok);
                put(role_c, finish(a3, 1), 1, base, ok);
            } else {
                asm volatile("" ::"v"(a3[0][0]), "v"(a3[1][0]), "v"(a3[2][1]), "v"(a3[3][1]));
            }
#pragma unroll
            for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f4 c0 = xi == 1 ? biasv[m] : (f4){0.f, 0.f, 0.f, 0.f};
                    a0[xi][m] = MFMA16(U[0][xi][0][m], V[0][xi], c0);
                    a1[xi][m] = MFMA16(U[1][xi][0][m], V[0][xi], a1[xi][m]);
                    a2[xi][m] = MFMA16(U[2][xi][0][m], V[0][xi], a2[xi][m]);
                }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int j = 5; j < 24; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, KW_VALU_PER_MFMA, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- second block: channel half 1; every row moves on one age.  The three MFMAs of an accumulator (a3 <- a2, a2 <- a1,
            // a1 <- a0, in this order) stay together, so that each result can take the registers its addend's successor has just
            // left: scheduled freely the old and new sets overlap and the lane constants end up parked in AGPRs or spilled — and a
            // scratch reload waits for vmcnt(0), i.e. for every LDS-DMA piece in flight.
#pragma unroll
            for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    a3[xi][m] = MFMA16(U[2][xi][1][m], V[1][xi], a2[xi][m]);
                    a2[xi][m] = MFMA16(U[1][xi][1][m], V[1][xi], a1[xi][m]);
                    a1[xi][m] = MFMA16(U[0][xi][1][m], V[1][xi], a0[xi][m]);
                    // the next row's first half is transformed under the first four triples (one xi each), then its second half is read
                    if (2 * xi + m < 4) transform1(D, V[0], 2 * xi + m);
                    if (2 * xi + m == 3) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < 4; ++k) D[k] = *(const h8*)(smem + rbn + doff[1][k]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
        // the first row's pixels: first half transformed, second half read (what every step leaves behind for the next)
        {
            const int rb0 = role ? KW_MID_OFF + ((0 - KW_LAG) & (KW_MID_RING - 1)) * KW_MID_ROW : 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) D[k] = *(const h8*)(smem + rb0 + doff[0][k]);
            transform(D, V[0]);
#pragma unroll
            for (int k = 0; k < 4; ++k) D[k] = *(const h8*)(smem + rb0 + doff[1][k]);
        }
        for (int s = 0; s < n_steps; ++s) {
            const int t = role ? s - KW_LAG : s;
            KWD_STEP_BEGIN
            if (role == 0) step(std::integral_constant<int, 0>{}, t, s);
            else step(std::integral_constant<int, 1>{}, t, s);
            KWD_STEP_END
            if constexpr (!kwd_no_dma) {
                if (wave == 0) dma_piece(s + KW_LEAD, 1);
            }
            KWD_WAIT_BEGIN
            if (kwd_no_wait || kwd_no_dma || kwd_no_epi) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else if (wave == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((KW_LEAD - 2) * kw_vm_per_step(0)) : "memory");
            else if (wave == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((KW_LEAD - 2) * kw_vm_per_step(1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((KW_LEAD - 2) * kw_vm_per_step(2)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            KWD_WAIT_END(true)
        }
        u += G;
        if (u >= a.n_units) break;
        unit_setup(u);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    KWD_EXIT
}

template __global__ void k_wino<false>(const PairArgs);
template __global__ void k_wino<true>(const PairArgs);

int wino_lds_bytes() { return KW_LDS; }

int prepare_wino_kernels()
{
    int rc = 0;
    for (const void* f : {(const void*)k_wino<false>, (const void*)k_wino<true>})
        rc |= (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, KW_LDS);
    return rc;
}

int launch_wino(const PairArgs& a, int grid, void* stream)
{
    launch_prepare();
    if (a.unit_slopes) hipLaunchKernelGGL((k_wino<true>), dim3(grid), dim3(64 * KW_NW), KW_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_wino<false>), dim3(grid), dim3(64 * KW_NW), KW_LDS, (hipStream_t)stream, a);
    return launch_status();
}

}  // namespace reve
