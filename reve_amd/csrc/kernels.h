// Shared host/device declarations for the gfx950 SRVGGNetCompact kernels (kernels.hip).
#pragma once
#include <stdint.h>

namespace reve {

// Geometry shared by every kernel: a workgroup owns a TILE_H x TILE_W block of output pixels.
constexpr int TILE_H = 16;
constexpr int TILE_W = 32;
constexpr int LDS_H = TILE_H + 2;          // input rows incl. the 1-pixel halo
constexpr int LDS_W = TILE_W + 2;          // LDS row pitch in pixels
constexpr int PIX_BYTES = 128;             // 64 channels x fp16
constexpr int FEAT = 64;
constexpr int LDS_TILE_BYTES = LDS_H * LDS_W * PIX_BYTES;   // 78,336 B
constexpr int KSTEPS = 18;                 // 9 taps x 2 halves of 32 input channels
constexpr int DMA_SEGS = 5;                // 8-pixel segments per LDS row (last one holds 2 px)

// One plane = one independently zero-padded image in the activation arena (the whole frame,
// or one ncnn-compat tile incl. its apron).  (x0, y0) is the frame coordinate of plane pixel (0,0).
// `base`: byte offset in an arena of the plane's border pixel (-1, -1); rows are ConvArgs::Wp pixels apart; `span`: bytes
// addressable from there (buffer bound of the plane's loads and stores).  One plane: base 0.  Several planes share ONE canvas
// (Engine::configure): neighbours share their 1-pixel zero border, so that the pair kernel can treat the canvas as one frame
// whose border columns / rows ("gutters") stay zero.
struct PlaneDesc { int w, h, x0, y0; unsigned long long base; unsigned span, reserved; };

// Physical position of logical channel c inside a 128-byte activation pixel.  The MFMA C/D layout
// leaves lane (pixel p, group g) of the wave that owns channel half ch = c>>5 holding channels
// 32ch + 16m + 4g + r (m = 0,1; r = 0..3); it stores them as ONE 16-byte piece at byte 64ch + 16g,
// i.e. position 32ch + 8g + 4m + r.
constexpr int chan_phys(int c) { return 32 * (c >> 5) + 8 * ((c >> 2) & 3) + 4 * ((c >> 4) & 1) + (c & 3); }
constexpr int chan_logical(int p) { return 32 * (p >> 5) + 16 * ((p >> 2) & 1) + 4 * ((p >> 3) & 3) + (p & 3); }

struct ConvArgs {
    const char* in;                  // activation arena read by this layer (plane 0 base)
    char* out;                       // activation arena written (body layers)
    const void* wpack;               // A fragments [KSTEPS][n co-blocks][64 lanes] x 16 B
    const uint16_t* bias;            // fp16 [n co-blocks * 16], logical channel order (zero padded)
    const uint16_t* slope;           // fp16 [64] (body layers)
    const PlaneDesc* planes;
    unsigned long long plane_stride; // bytes between planes in an arena
    int n_planes, tiles_x, tiles_y, n_items;
    int Wp;                          // arena row pitch in pixels (= tiles_x*TILE_W + 2)
    int reverse;                     // walk the work items backwards (Infinity-Cache reuse)
    int blocked;                     // items == nullptr, one plane: work order in 4x8 blocks of tiles, computed in the kernel
    int unit_slopes;                 // body layers: every PReLU slope of the layer lies in [0, 1] (selects prelu8_unit_slopes)
    const uint32_t* items;           // optional work list: tx | ty << 10 | plane << 20 (planes of unequal size:
                                     // only their non-empty tiles); nullptr = every tile of every plane
    // conv_last only
    const uint8_t* src; long long src_stride;
    uint8_t* dst; long long dst_stride;
    int frame_w, frame_h, pad;
};

// Two body layers per launch (kernels_pair.hip): a workgroup rolls down a strip of PAIR_COLS columns per layer, of which
// PAIR_VALID are valid output columns of the second layer (its one halo column per side is recomputed by the first).
constexpr int PAIR_COLS = 64;
constexpr int PAIR_VALID = PAIR_COLS - 2;        // the first layer reads PAIR_COLS + 2 input columns, so all of its 64 are valid

struct PairArgs {
    const char* in;                  // activation arena read by the first layer (one plane: the whole frame)
    char* out;                       // arena written by the second layer
    const void* wpack[2];            // A fragments of the two layers, as ConvArgs::wpack
    const uint16_t* bias[2];
    const uint16_t* slope[2];
    int W, H;                        // frame size
    int Wp, Hp;                      // arena pitch and height in pixels (1-pixel zero border included)
    int n_strips, n_segs, seg_h;     // units = strips of PAIR_VALID columns x segments of seg_h rows
    int n_units;
    int reverse;                     // walk the units backwards
    // a canvas of several planes (tiled frames): frame columns that are gutters between planes (col_ok[x] == 0; nullptr: none) and
    // gutter rows gut_first + k * gut_period, k = 0, 1, ... (gut_period 0: none; >= 4); both stay zero in every layer.
    const unsigned char* col_ok;
    int gut_first, gut_period;
    int unit_slopes;                 // every PReLU slope of BOTH layers lies in [0, 1]
};

// Several small frames per launch (Engine::configure): up to MAX_BATCH frames of one size lie one below the other on a canvas
// (each a plane; neighbours share their 1-pixel zero border row, a "gutter row" to the pair kernel); the kernels that touch the
// u8 frames take one source / destination pointer per plane.
constexpr int MAX_BATCH = 16;

// conv_last over a whole frame as a rolling-strip kernel (kernels_last.hip): units as in PairArgs
struct LastStripArgs {
    const char* in;                  // arena holding the last body layer's output (one plane)
    const void* wpack;               // A fragments of conv_last (pack_last, store order), as ConvArgs::wpack
    const uint16_t* bias;
    const uint8_t* src;              // the u8 RGB input frame (residual) and the u8 RGB output frame
    uint8_t* dst;
    long long src_stride, dst_stride;
    int W, H, Wp, Hp;
    int n_strips, seg_h, n_units;
    int reverse;
    // several frames per launch: n_units = n_frames * units_per_frame, unit u belongs to frame u / units_per_frame; frame f reads
    // the arena at in + f * in_frame_stride (its plane: H + 2 rows) and src_tab[f] / dst_tab[f].  n_frames 0: one frame, src / dst
    int n_frames, units_per_frame;
    long long in_frame_stride;
    const uint8_t* src_tab[MAX_BATCH];
    uint8_t* dst_tab[MAX_BATCH];
    // a canvas of planes (tiled frames; units != nullptr selects the CANVAS instantiation): unit u = units[u] = plane | strip << 12 |
    // segment << 20 — a strip of PAIR_VALID columns x seg_h rows of the plane's INTERIOR (the plane without its `pad`-pixel apron),
    // read at the plane's place in the arena (planes[p].base, pitch Wp) and written, with the residual of src, at frame position
    // (planes[p].x0 + pad, planes[p].y0 + pad); W, H: the frame's; n_strips / Hp unused
    const PlaneDesc* planes;
    const uint32_t* units;
    int pad;
};

#ifndef FIRST_NT_VALUE
#define FIRST_NT_VALUE 8
#endif
constexpr int FIRST_NT = FIRST_NT_VALUE;   // tiles per group in k_first (LDS images per workgroup)

struct FirstArgs {
    const uint8_t* src; long long src_stride;
    int frame_w, frame_h;
    char* out;
    const void* wpack;               // A fragments [2][4][64 lanes] x 16 B
    const uint16_t* bias; const uint16_t* slope;
    const PlaneDesc* planes;
    unsigned long long plane_stride;
    int n_planes, tiles_x, tiles_y, Wp;
    int n_items; const uint32_t* items;   // as in ConvArgs
    int blocked;
    int n_src;                            // > 0: plane p of the work list is a frame of its own, read from src_tab[p] (planes' x0 = y0 = 0)
    const uint8_t* src_tab[MAX_BATCH];
};

// launchers (kernels.hip); stream is a hipStream_t
// per-device set-up of the kernels' function attributes (dynamic LDS sizes); call with the device current
int prepare_body_kernels();
void debug_blocked_order(int tiles_x, int tiles_y, uint32_t* out);   // tx | ty << 10 per work item (host-side, tests)
int launch_first(const FirstArgs& a, int grid, void* stream);
int launch_body(const ConvArgs& a, int grid, void* stream);
int launch_last(const ConvArgs& a, int scale, int grid, void* stream);
int launch_last_probe(const ConvArgs& a, int scale, int grid, void* stream);   // fp16 conv_last output, store order (debug probe)
int conv_lds_bytes();
int prepare_pair_kernels();
int launch_pair(const PairArgs& a, int grid, void* stream);
int pair_lds_bytes();
// Winograd F(2,3)-along-the-row body pairs (kernels_wino.hip); PairArgs::wpack[] = pack_body_wino() fragments
int prepare_wino_kernels();
int launch_wino(const PairArgs& a, int grid, void* stream);
int wino_lds_bytes();
int wino_ring_offset(int column, int chunk);      // kw_ring_off, for the layout test (no GPU needed)
int prepare_last_strip_kernels();
int launch_last_strip(const LastStripArgs& a, int scale, int grid, void* stream);

}  // namespace reve
