// conv_last (64 -> 12 / 27 / 48 3x3 conv, PixelShuffle(2 / 3 / 4), + nearest-upsampled input, post-process -> u8 RGB) for whole
// frames on gfx950, as a ROLLING STRIP kernel (round 3).  Same arithmetic and summation order as k_body<ORDER, 2 / 3 / 4>
// (kernels.hip), which stays for tiled frames: identical output bytes.  Replaces nothing in the reference beyond what that kernel
// replaces (reve-shared/src/lib.rs:134-147: the realesrgan-ncnn-vulkan subprocess).
//
// Why: a tile of k_body has one tile's LDS-DMA in flight per CU — issued while the previous tile computes, waited for at its end.
// Here a workgroup owns a vertical strip of 62 output columns and rolls down a segment of rows like k_pair (kernels_pair.hip): the
// input rows arrive as a CONTINUOUS stream of LDS-DMA pieces two steps (eight rows, 64 KiB) ahead of their use in a 16-row ring,
// never drained inside a unit, loaded as streaming data (nt: read once; DESIGN.md §4 "Cache policy").  The four waves take one
// px-block (16 columns) each.  A step is four rows per wave (72 MFMAs per co-block) and one s_barrier; a fragment read for ring row
// i feeds output rows i, i - 1, i - 2 (36 reads per step); a row's epilogue — PixelShuffle by store order (pack_last), residual,
// quantisation, the scale's store format — runs under the next input row's MFMAs.  x2: 50 us against the tile kernel's 64 us at
// 1080p; x3 81 / 85; x4 103 / 105 (MFMA-bound).  Measurements and timing-only variants: DESIGN.md §4, profiles/r03/.
//
// Round 6: the CANVAS instantiation does the same for tiled frames (the binary's tiling: planes with a 10-pixel apron on one canvas,
// Engine::configure) — a unit is a strip of ONE plane's un-padded interior, read at the plane's place in the arena and written at
// the plane's place in the frame; the apron, which the tile kernel computes and drops, is not computed.  Units come from a list
// (LastStripArgs::units).  The whole-frame instantiation is the code it was.
#include <algorithm>
#include <type_traits>

#include "kernels_dev.h"

#ifndef KL_DMA_AUX
#define KL_DMA_AUX 2            // cache policy of the input rows' LDS-DMA loads (1 sc0, 2 nt, 16 sc1): read once, streaming
#endif
#ifndef KL_B_AHEAD
#define KL_B_AHEAD 2            // operand reads issued this many fragments ahead of their first MFMA
#endif
#ifndef KL_LEAD_STEPS
#define KL_LEAD_STEPS 2         // (3: an 18-row ring, 96 KiB in flight per CU — measured no faster, profiles/r03/ab_conv_last_strip.txt)
#endif
#ifndef KL_VALU_PER_MFMA
#define KL_VALU_PER_MFMA 12     // epilogue instructions the scheduler places behind each MFMA
#endif
#if (defined(KL_ABL_ROLL_UP) || defined(KL_ABL_NO_EPI) || defined(KL_ABL_NO_WAIT) || defined(KL_ABL_NO_LDS) || defined(KL_ABL_NO_DMA) || defined(KL_ABL_NO_STORE) || defined(KL_ABL_NO_MFMA)) && !defined(REVE_DIAGNOSTIC_BUILD)
#error "KL_ABL_* are timing-only ablations (wrong results): build them with -DREVE_DIAGNOSTIC_BUILD, never into the product library"
#endif

// The timing-only ablations (KL_ABL_*) live in kernels_last_diag.inc and exist in diagnostic builds only (scripts/ablate_pair.sh,
// KFILE=kernels_last.hip); a product build sees the plain hooks below and compiles with that file absent.
#ifdef REVE_DIAGNOSTIC_BUILD
#include "kernels_last_diag.inc"
#else
#define KL_ABL_DMA_N KL_DMA_PER_WAVE
#define KL_ABL_STORE_N KL_RPS            // (x the store instructions per row)
#define KL_MFMA(a, b, c) MFMA16(a, b, c)
#define KL_ROW(R) (y0 + (R))             // image row of the unit's R-th row; KL_ABL_ROLL_UP walks the strip from its last row up
#define KL_ROW_OK(y) ((y) < y1)
#define KL_DY(d) (d)
#define KL_DMA_ROW(rho) (y0 + (rho))
namespace reve { constexpr bool kld_no_epi = false, kld_no_store = false, kld_no_dma = false, kld_no_wait = false; }
#define KLD_KEEP(...)
#define KLD_OPERANDS
#define KLD_LOAD_B(i, hf)
#define KLD_CLAMP_ROW(fy)
#endif

namespace reve {

namespace {
constexpr int KL_NW = 4;
constexpr int KL_COLS = PAIR_COLS;                       // ring columns (64): 62 valid output columns + the convolution's halo
constexpr int KL_VALID = KL_COLS - 2;
constexpr int KL_ROW_BYTES = KL_COLS * PIX_BYTES;        // 8,192: eight DMA pieces
constexpr int KL_LEAD = KL_LEAD_STEPS;                   // a step's rows are requested this many steps before it runs
constexpr int KL_RING = KL_LEAD == 2 ? 16 : 6 + 4 * KL_LEAD;      // rows in the ring: the step's six + those in flight (14 -> 16: a mask, not a modulo)
constexpr int KL_RPS = 4;                                // rows per step
constexpr int KL_LDS = KL_RING * KL_ROW_BYTES + 1024;    // (+ slack: the last px-block reads two columns past its row)
constexpr int KL_DMA_PER_WAVE = KL_RPS * (KL_ROW_BYTES / 1024) / KL_NW;     // 8 pieces per wave and step
static_assert(KL_LDS <= 160 * 1024 && KL_DMA_PER_WAVE == 8 && KL_LEAD >= 2 && (KL_LEAD - 1) * (KL_RPS + 8 + 4 * KL_RPS) < 64, "geometry");

// f(integral_constant<int, I>) for I = 0 .. N-1, written out at compile time
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
}  // namespace

// SC: the graph's upscale factor.  x2: 12 output channels = one co-block, a lane's four accumulator rows are bytes 4 * (g & 1) + r of
// the 6-byte run of output sub-row g >> 1 (a dword store from even lane groups, a short from odd ones); x3: 27 channels = two
// co-blocks, lane groups 0..2 hold bytes 0..7 of sub-row g (one 8-byte store), group 3 the ninth byte of the three sub-rows (three
// byte stores); x4: 48 channels = three co-blocks, a lane's three words are the 12-byte run of sub-row g (one 12-byte store).
// The channel orders are pack_last(store_order)'s (model.cpp), the epilogue arithmetic k_body's.
template <int SC, bool CANVAS>
__global__ void __launch_bounds__(64 * KL_NW, 1) k_last_strip(const LastStripArgs a)
{
    constexpr int NCOB = SC == 2 ? 1 : (SC == 3 ? 2 : 3);      // co-blocks computed
    constexpr int NPACK = SC == 2 ? 1 : (SC == 3 ? 2 : 4);     // co-blocks per k-step in a.wpack (x4: the fourth is all zero)
    constexpr int NSTORE = SC == 2 ? 2 : (SC == 3 ? 4 : 1);    // store instructions per row
    constexpr int NSLOT = SC == 4 ? 3 : 2;                     // reads under which a row's epilogue VALU runs; its stores 2 reads later
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;

    // weights: 18 fragments per co-block, straight from global memory (every wave all of them)
    h8 wf[KSTEPS][NCOB];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < NCOB; ++m) wf[s][m] = ((const h8*)a.wpack)[(s * NPACK + m) * 64 + lane];
    float bias[NCOB][4];
#pragma unroll
    for (int m = 0; m < NCOB; ++m) {
        const h4 b = *(const h4*)(a.bias + 16 * m + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[m][r] = (float)b[r];
    }
    // operand reads: output column c = 16 * wave + pl reads ring columns c + dx
    int roff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            roff[dx][hf] = (16 * wave + pl + dx) * PIX_BYTES + 16 * ((4 * hf + g) ^ ((pl + dx) & 6));

    const int plane_bytes = a.Hp * a.Wp * PIX_BYTES;
    // (several frames per launch: the three descriptors are the unit's frame's, set again by unit_setup)
    auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, plane_bytes, 0x00020000);
    auto no_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, 0, 0x00020000);
    auto srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, (((int)(a.src_stride * a.H) + 3) & ~3), 0x00020000);
    auto drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, (int)(a.dst_stride * a.H * SC), 0x00020000);

    const int G = gridDim.x;
    const int bid = blockIdx.x;
    int u = ((G & 7) == 0) ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;

    // residual pixel (RGB in one dword) of this lane's column in image row fy, clamped into the frame (masked lanes, rows past the
    // segment).  At the very end of the frame buffer the load is moved back inside it and the pixel sits sh_lane bits up in the
    // dword: the last row only (the shift is applied in the epilogue, not here: the load's wait would land at the top of the step).
    // Offsets are a per-unit lane part plus a scalar row part (no per-row vector multiplies).
    const int src_lim = (int)(a.src_stride * a.H) - 4 > 0 ? (int)(a.src_stride * a.H) - 4 : 0;
    int res_lane = 0, sh_lane = 0;
    int fx0 = 0, fy0 = 0;          // CANVAS: frame coordinates of the unit's plane's interior pixel (0, 0); the unit's x0 / y0 / y1 are interior coordinates
    int iw = a.W;                  // CANVAS: width of that interior (whole frame: the frame's)
    auto fetch_resid = [&](int fy) -> unsigned {
        if constexpr (CANVAS) fy += fy0;
        fy = fy >= a.H ? a.H - 1 : fy;
        KLD_CLAMP_ROW(fy)
        const int off = fy * (int)a.src_stride + res_lane;
        return __builtin_amdgcn_raw_buffer_load_b32(srsrc, off < src_lim ? off : src_lim, 0, 0);
    };
    // store offsets: lane parts of the row's main store (x2: the dword of even lane groups; x3: the 8 bytes of groups 0..2; x4:
    // the 12 bytes) and of its second kind (x2: the short of odd groups; x3: group 3's bytes), OOB_OFF where the lane stores nothing
    constexpr int OOB_OFF = 0x40000000;           // (the engine uses this kernel for output frames below 1 GiB)
    int st_lane_a = OOB_OFF, st_lane_b = OOB_OFF;
    int x0 = 0, y0 = 0, y1 = 0, n_steps = 0;
    int vcol[2] = {0, 0};        // DMA source column part of the wave's two column groups (8 px each): ring column j <-> arena column x0 + j
    // ring row rho <-> image row y0 - 1 + rho <-> arena row y0 + rho (clamped: the arena's border rows are zero)
    // (CANVAS: rows and columns count from the plane's interior — the descriptor's base is moved there — and what a strip reads past its
    // plane lies in the neighbouring plane or the canvas' slack rows and feeds dropped outputs only.  The row rides in the scalar
    // offset, which the descriptor's range check does not see: it is clamped to the rows the arena really has below the plane,
    // ar_lim — one row short of the last, since a strip's columns may run on into the next row)
    int ar_lim = a.Hp - 1;
    auto dma_piece = [&](int rho, int i, bool needed) {
        int ar = KL_DMA_ROW(rho);
        ar = ar > ar_lim ? ar_lim : ar;
        dma16a<KL_DMA_AUX>(needed ? in_rsrc : no_rsrc, to_lds(smem + (int)((unsigned)rho % KL_RING) * KL_ROW_BYTES + (wave + KL_NW * i) * 1024), vcol[i], ar * a.Wp * PIX_BYTES);
    };
    const int ox_lane = 16 * wave + pl;       // this lane's output column inside the strip
    unsigned resid2[2][KL_RPS];      // residual pixels of the four rows of even / odd steps, fetched one step ahead
    auto unit_setup = [&](int un) {
        int uu = a.reverse ? a.n_units - 1 - un : un;
        if constexpr (CANVAS) {
            // (pointers inside the argument struct are generic: what is loaded through them counts as divergent, and a descriptor built from
            // it would be re-read lane by lane at every use — so every word comes back to the scalar side at once)
            const unsigned code = __builtin_amdgcn_readfirstlane(a.units[__builtin_amdgcn_readfirstlane(uu)]);      // plane | strip << 12 | segment << 20
            const int* pw = (const int*)(a.planes + (code & 0xfffu));             // PlaneDesc: w, h, x0, y0, base (64 bits), span
            const int p_w = __builtin_amdgcn_readfirstlane(pw[0]), p_h = __builtin_amdgcn_readfirstlane(pw[1]);
            const int p_x0 = __builtin_amdgcn_readfirstlane(pw[2]), p_y0 = __builtin_amdgcn_readfirstlane(pw[3]);
            const unsigned long long p_base = (unsigned)__builtin_amdgcn_readfirstlane(pw[4]) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(pw[5]) << 32);
            const int p_span = __builtin_amdgcn_readfirstlane(pw[6]);
            ar_lim = __builtin_amdgcn_readfirstlane(pw[7]) - a.pad - 2;           // PlaneDesc::reserved = arena rows from the plane's border row to the arena's end
            const int shift = (a.pad * a.Wp + a.pad) * PIX_BYTES;                   // from the plane's border pixel to its interior's
            in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + p_base + shift), 0, p_span - shift, 0x00020000);
            fx0 = p_x0 + a.pad; fy0 = p_y0 + a.pad;
            iw = p_w - 2 * a.pad;
            const int ih = p_h - 2 * a.pad;
            x0 = (int)((code >> 12) & 0xffu) * KL_VALID;
            y0 = (int)(code >> 20) * a.seg_h;
            y1 = y0 + a.seg_h < ih ? y0 + a.seg_h : ih;
        } else {
        if (a.n_frames) {
            int f = __builtin_amdgcn_readfirstlane(uu / a.units_per_frame);      // (the division runs on the vector unit: back to scalars at once)
            uu -= f * a.units_per_frame;
            f = f < MAX_BATCH ? f : 0;
            in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + f * a.in_frame_stride), 0, plane_bytes, 0x00020000);
            srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.src_tab[f], 0, (((int)(a.src_stride * a.H) + 3) & ~3), 0x00020000);
            drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst_tab[f], 0, (int)(a.dst_stride * a.H * SC), 0x00020000);
        }
        const int sy = __builtin_amdgcn_readfirstlane(uu / a.n_strips), sx = uu - sy * a.n_strips;
        x0 = sx * KL_VALID;
        y0 = sy * a.seg_h;
        y1 = y0 + a.seg_h < a.H ? y0 + a.seg_h : a.H;
        }
        n_steps = (y1 - y0 + KL_RPS - 1) / KL_RPS;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = 8 * (wave + KL_NW * i) + (lane >> 3);
            int ac = x0 + j;
            if constexpr (!CANVAS) ac = ac > a.Wp - 1 ? a.Wp - 1 : ac;
            vcol[i] = ac * PIX_BYTES + 16 * ((lane & 7) ^ (j & 6));
        }
        {
            const int fxl = (CANVAS ? fx0 : 0) + x0 + ox_lane;        // this lane's column in the frame
            const int fx = fxl < a.W ? fxl : a.W - 1;
            res_lane = fx * 3;
            const int off = (a.H - 1) * (int)a.src_stride + res_lane;
            sh_lane = off < src_lim ? 0 : 8 * (off - src_lim);
            const bool col_ok = ox_lane < KL_VALID && x0 + ox_lane < iw;
            if constexpr (SC == 2) {
                const int o = (g >> 1) * (int)a.dst_stride + fxl * 6 + 4 * (g & 1);
                st_lane_a = (col_ok && !(g & 1)) ? o : OOB_OFF;
                st_lane_b = (col_ok && (g & 1)) ? o : OOB_OFF;
            } else if constexpr (SC == 3) {
                st_lane_a = (col_ok && g < 3) ? g * (int)a.dst_stride + fxl * 9 : OOB_OFF;
                st_lane_b = (col_ok && g == 3) ? fxl * 9 + 8 : OOB_OFF;
            } else {
                st_lane_a = col_ok ? g * (int)a.dst_stride + fxl * 12 : OOB_OFF;
            }
        }
        // (before the pieces: the compiler's own wait at their first use then counts those as younger and lets them fly)
#pragma unroll
        for (int j = 0; j < KL_RPS; ++j) resid2[0][j] = fetch_resid(KL_ROW(j));
        // rows of steps 0 and 1 (and the two halo rows of step 1's last row): ring rows 0..9 -> ten rows, 20 pieces per wave
#pragma unroll
        for (int rho = 0; rho < KL_LEAD * KL_RPS + 2; ++rho)
#pragma unroll
            for (int i = 0; i < 2; ++i) dma_piece(rho, i, true);
    };

    unit_setup(u);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * KL_RPS * (KL_LEAD - 2)) : "memory");      // rows 0 .. 9 have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // (x3 / x4: the weights are parked in the accumulator file, the MFMA reads its A operand from there: -mllvm -amdgpu-mfma-vgpr-form=1)
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < NCOB; ++m) {
            if (NCOB > 1 && s * NCOB + m < 64) asm volatile("" : "+a"(wf[s][m]));
            else asm volatile("" : "+v"(wf[s][m]));
        }

    for (;;) {
        f4 racc[NCOB];                            // the row whose epilogue is pending
#pragma unroll
        for (int m = 0; m < NCOB; ++m) racc[m] = (f4){0.f, 0.f, 0.f, 0.f};
        // a finished row's epilogue: VALU in pieces (in the shadow of MFMAs: x2 the two halves of the lane's four bytes, x3 / x4 a
        // co-block's four bytes per piece), then the stores
        unsigned pend[NCOB];
#pragma unroll
        for (int m = 0; m < NCOB; ++m) pend[m] = 0;
        int pend_row = OOB_OFF;          // scalar row part of the store offsets
        unsigned pend_rb = 0;
        auto epi_bytes = [&](const f4& ac, int m, int r_lo, int r_hi) {
            const unsigned rb = pend_rb;
            if constexpr (kld_no_epi) { KLD_KEEP("v"(ac), "v"(rb)) return; }
#pragma unroll
            for (int r = r_lo; r < r_hi; ++r) {
                // x3, lane group 3: rows 0..2 of co-block 0 are byte 8 (colour 2) of the three sub-rows
                const int c = SC == 2 ? (r + (g & 1)) % 3 : ((SC == 3 && g == 3) ? 2 : (4 * m + r) % 3);
                const float res = (float)(_Float16)((float)((rb >> (8 * c)) & 0xffu) * (1.0f / 255.0f));
                const float v = (float)(_Float16)ac[r];
                const float o = (float)(_Float16)(v + res);
                pend[m] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(o * 255.0f + 0.5f), r, pend[m]);
            }
        };
        // piece k (0 .. NSLOT - 1) of the row in `ac`
        auto epi_piece = [&](const f4 (&ac)[NCOB], int k) {
            if constexpr (SC == 2) epi_bytes(ac[0], 0, 2 * k, 2 * k + 2);
            else epi_bytes(ac[k], k, 0, 4);
        };
        auto epi_where = [&](unsigned rb, int y, bool ok) {
            if constexpr (CANVAS) y += fy0;          // (the unit's rows are the plane interior's; stores and the residual's last-row fix-up go by the frame's)
            pend_rb = rb >> (y == a.H - 1 ? sh_lane : 0);
            pend_row = __builtin_amdgcn_readfirstlane(ok ? y * SC * (int)a.dst_stride : OOB_OFF);      // (wave-uniform: a scalar offset, no waterfall loop)
        };
        auto put = [&]() {
            if constexpr (kld_no_store) { KLD_KEEP("v"(pend[0]), "s"(pend_row)) return; }
            if constexpr (SC == 2) {
                __builtin_amdgcn_raw_buffer_store_b32(pend[0], drsrc, st_lane_a, pend_row, 0);
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)pend[0], drsrc, st_lane_b, pend_row, 0);
            } else if constexpr (SC == 3) {
                __builtin_amdgcn_raw_buffer_store_b64((u32x2){pend[0], pend[1]}, drsrc, st_lane_a, pend_row, 0);
                const int stride = __builtin_amdgcn_readfirstlane((int)a.dst_stride);
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(pend[0] >> (8 * i)), drsrc, st_lane_b, pend_row == OOB_OFF ? OOB_OFF : pend_row + i * stride, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b96((u32x3){pend[0], pend[1], pend[2]}, drsrc, st_lane_a, pend_row, 0);
            }
        };
        // (even and odd steps written out: their residual registers alternate by NAME — a rotating copy at the end of a step
        // would wait for the loads just issued, and with them, in order, for every piece requested before)
        auto step = [&](int s, auto par_c) __attribute__((always_inline)) {
            constexpr int par = decltype(par_c)::value;
            unsigned (&resid)[KL_RPS] = resid2[par];
            unsigned (&resid_next)[KL_RPS] = resid2[par ^ 1];
            const int R0 = KL_RPS * s;                       // first output row of the step (relative to y0); reads ring rows R0 .. R0 + 5
            const bool dma_needed = s + KL_LEAD < n_steps;   // this step requests the last four rows of step s + KL_LEAD
            // (the fourth register still holds the residual of the previous step's last row, whose epilogue runs at reads 6, 7)
#pragma unroll
            for (int j = 0; j < KL_RPS - 1; ++j) resid_next[j] = fetch_resid(KL_ROW(R0 + KL_RPS + j));
            int rb[KL_RPS + 2];
#pragma unroll
            for (int i = 0; i < KL_RPS + 2; ++i) rb[i] = (int)((unsigned)(R0 + i) % KL_RING) * KL_ROW_BYTES;
            // flat read L = 6 * i + 2 * dx + hf: the fragment (ring row R0 + i, tap column dx, channel half hf) feeds output rows
            // j = i - dy (dy = 0..2) of the step — every accumulator still adds its 18 products in k-step order (dy, dx, hf)
            KLD_OPERANDS
            auto load_b = [&](int L) {
                const int i = L / 6, dx = (L - 6 * i) >> 1, hf = L & 1;
                KLD_LOAD_B(i, hf)
                return *(const h8*)(smem + rb[i] + roff[dx][hf]);
            };
            constexpr int NL = 6 * (KL_RPS + 2), AH = KL_B_AHEAD;
            h8 Bb[AH + 1];
#pragma unroll
            for (int L = 0; L < AH; ++L) Bb[L] = load_b(L);
            f4 acc[KL_RPS][NCOB];
#pragma unroll
            for (int j = 0; j < KL_RPS; ++j)
#pragma unroll
                for (int m = 0; m < NCOB; ++m) acc[j][m] = (f4){bias[m][0], bias[m][1], bias[m][2], bias[m][3]};
            static_for<0, NL>([&](auto Lc) __attribute__((always_inline)) {
                constexpr int L = decltype(Lc)::value;
                constexpr int i = L / 6, dx = (L - 6 * i) >> 1, hf = L & 1;
                if constexpr (L + AH < NL) Bb[(L + AH) % (AH + 1)] = load_b(L + AH);
                if constexpr (!kld_no_dma && (L & 3) == 2 && L < 32) {      // the step's eight DMA pieces: rows R0 + 10 .. R0 + 13, two column groups each
                    constexpr int k = L >> 2;
                    dma_piece(R0 + KL_LEAD * KL_RPS + 2 + (k >> 1), k & 1, dma_needed);
                }
                // a finished row leaves under the MFMAs of the next input row (row 3 of the previous step under input row 1):
                // VALU at reads E .. E + NSLOT - 1 of that row, stores at E + NSLOT + 1
                if constexpr (L == 6 + NSLOT + 1 || L == 18 + NSLOT + 1 || L == 24 + NSLOT + 1 || L == 30 + NSLOT + 1) put();
                if constexpr (L == 8) resid_next[KL_RPS - 1] = fetch_resid(KL_ROW(R0 + 2 * KL_RPS - 1));
                // reads of a later fragment and this one's vector-memory instructions above the fence, MFMAs and VALU below it
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (L >= 6 && L < 6 + NSLOT) {
                    if constexpr (L == 6) epi_where(resid_next[KL_RPS - 1], KL_ROW(R0 - 1), s > 0 && KL_ROW_OK(KL_ROW(R0 - 1)));      // (the previous step's last row)
                    epi_piece(racc, L - 6);
                }
                if constexpr (L >= 18 && (L % 6) < NSLOT) {
                    constexpr int j = L / 6 - 3;
                    if constexpr ((L % 6) == 0) epi_where(resid[j], KL_ROW(R0 + j), KL_ROW_OK(KL_ROW(R0 + j)));
                    epi_piece(acc[j], L % 6);
                }
                constexpr int n_rows = (i >= 2 ? 1 : 0) + (i >= 1 && i <= KL_RPS ? 1 : 0) + (i < KL_RPS ? 1 : 0);
                static_for<0, NCOB>([&](auto mc) __attribute__((always_inline)) {
                    constexpr int m = decltype(mc)::value;
                    if constexpr (i >= 2) acc[i - 2][m] = KL_MFMA(wf[(KL_DY(2) * 3 + dx) * 2 + hf][m], Bb[L % (AH + 1)], acc[i - 2][m]);
                    if constexpr (i >= 1 && i <= KL_RPS) acc[i - 1][m] = KL_MFMA(wf[(KL_DY(1) * 3 + dx) * 2 + hf][m], Bb[L % (AH + 1)], acc[i - 1][m]);
                    if constexpr (i < KL_RPS) acc[i][m] = KL_MFMA(wf[(KL_DY(0) * 3 + dx) * 2 + hf][m], Bb[L % (AH + 1)], acc[i][m]);
                });
                static_for<0, n_rows * NCOB>([&](auto) __attribute__((always_inline)) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, SC == 2 ? KL_VALU_PER_MFMA : (KL_VALU_PER_MFMA + 1) / 2, 0);
                });
                if constexpr (L == 17 || L == 23 || L == 29) {      // the finished row in VGPRs
#pragma unroll
                    for (int m = 0; m < NCOB; ++m) asm volatile("" : "+v"(acc[(L - 17) / 6][m]));
                }
            });
#pragma unroll
            for (int m = 0; m < NCOB; ++m) {
                asm volatile("" : "+v"(acc[KL_RPS - 1][m]));
                racc[m] = acc[KL_RPS - 1][m];
            }
            // every piece of the PREVIOUS step has landed: this step's 4 residual loads, 8 pieces and 8 stores may stay in flight
            if constexpr (kld_no_wait) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((KL_LEAD - 1) * (KL_RPS + KL_ABL_DMA_N + KL_ABL_STORE_N * NSTORE)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        // (no path from an even step to an even step: the compiler's wait counts take the minimum over all paths)
        for (int s = 0;; s += 2) {
            step(s, std::integral_constant<int, 0>{});
            if (s + 1 >= n_steps) break;
            step(s + 1, std::integral_constant<int, 1>{});
            if (s + 2 >= n_steps) break;
        }
        // the unit's last row
        epi_where((n_steps & 1) ? resid2[0][KL_RPS - 1] : resid2[1][KL_RPS - 1], KL_ROW(KL_RPS * n_steps - 1), KL_ROW_OK(KL_ROW(KL_RPS * n_steps - 1)));
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) epi_piece(racc, k);
        put();
        u += G;
        if (u >= a.n_units) break;
        unit_setup(u);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * KL_RPS * (KL_LEAD - 2)) : "memory");      // rows 0 .. 9 have landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

int prepare_last_strip_kernels()
{
    int rc = 0;
    for (const void* f : {(const void*)k_last_strip<2, false>, (const void*)k_last_strip<3, false>, (const void*)k_last_strip<4, false>,
                          (const void*)k_last_strip<2, true>, (const void*)k_last_strip<3, true>, (const void*)k_last_strip<4, true>})
        rc |= (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, KL_LDS);
    return rc;
}

int launch_last_strip(const LastStripArgs& a, int scale, int grid, void* stream)
{
    launch_prepare();
    if (a.units) {          // a canvas of planes: units from the list
        if (scale == 2) hipLaunchKernelGGL((k_last_strip<2, true>), dim3(grid), dim3(64 * KL_NW), KL_LDS, (hipStream_t)stream, a);
        else if (scale == 3) hipLaunchKernelGGL((k_last_strip<3, true>), dim3(grid), dim3(64 * KL_NW), KL_LDS, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((k_last_strip<4, true>), dim3(grid), dim3(64 * KL_NW), KL_LDS, (hipStream_t)stream, a);
        return launch_status();
    }
    if (scale == 2) hipLaunchKernelGGL((k_last_strip<2, false>), dim3(grid), dim3(64 * KL_NW), KL_LDS, (hipStream_t)stream, a);
    else if (scale == 3) hipLaunchKernelGGL((k_last_strip<3, false>), dim3(grid), dim3(64 * KL_NW), KL_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((k_last_strip<4, false>), dim3(grid), dim3(64 * KL_NW), KL_LDS, (hipStream_t)stream, a);
    return launch_status();
}

}  // namespace reve
