// Layout arithmetic of the engine that needs no GPU: the address budget of a frame's planes and how many small frames share a
// launch.  Engine::configure (engine.cpp) uses both; the CPU tests reach them through reve_debug_geometry /
// reve_debug_frames_per_launch (include/reve_hip_debug.h), and the CPU sanitizer builds link this file unchanged.
#include <algorithm>

#include "../../include/reve_hip.h"
#include "engine.h"

namespace reve {

// What the layout of a frame costs in addresses, without touching the GPU (Engine::configure uses it; reve_debug_geometry
// exposes it to the CPU tests).  out = {planes, canvas pitch Wp, canvas height Hp, bytes of one arena, largest byte offset a tile
// kernel forms INSIDE a plane}.  The tile kernels (k_first, k_body, conv_last) address a plane through a 64-bit base and
// 32-bit offsets ((row * Wp + column) * 128 as int), the pair kernel the whole canvas through 32-bit offsets: returns
// REVE_E_UNSUPPORTED when a plane's rows (its height rounded up to whole tiles, + border) times the canvas pitch reach 2 GiB —
// rows beyond that would fall outside the buffer bound: loads return 0, stores are dropped, no error.
int frame_geometry(int w, int h, int tile, int prepad, long long out[5])
{
    if (w <= 0 || h <= 0 || tile < 0 || (tile > 0 && tile < 32)) return REVE_E_INVALID;
    if (prepad <= 0) prepad = 10;
    long long n_planes = 1, Wp, Hp, maxh = h, maxw = w;
    if (tile > 0) {
        const long long xt = (w + tile - 1) / tile, yt = (h + tile - 1) / tile;
        n_planes = xt * yt;
        maxw = std::min<long long>(tile, w) + 2 * prepad;
        maxh = std::min<long long>(tile, h) + 2 * prepad;
        if (n_planes > 1) {
            Wp = 1; Hp = 1;
            for (long long xi = 0; xi < xt; ++xi) Wp += std::min<long long>((xi + 1) * tile, w) - xi * tile + 2 * prepad + 1;
            for (long long yi = 0; yi < yt; ++yi) Hp += std::min<long long>((yi + 1) * tile, h) - yi * tile + 2 * prepad + 1;
        }
    }
    const long long tiles_x = (maxw + TILE_W - 1) / TILE_W, tiles_y = (maxh + TILE_H - 1) / TILE_H;
    if (n_planes == 1) { Wp = tiles_x * TILE_W + 2; Hp = tiles_y * TILE_H + 2; }
    const long long canvas = Hp * Wp * PIX_BYTES;
    out[0] = n_planes; out[1] = Wp; out[2] = Hp;
    out[3] = n_planes == 1 ? canvas : canvas + (long long)(TILE_H + 2) * Wp * PIX_BYTES;
    out[4] = (tiles_y * TILE_H + 2) * Wp * PIX_BYTES;
    return out[4] >= (1ll << 31) ? REVE_E_UNSUPPORTED : 0;
}

// Whole frames too small to fill the chip alone: how many of them share a launch (1: every frame has its own).  The pair kernel
// cuts a frame into strips of 62 columns x segments of rows, one unit per CU where the frame allows it, and a segment pays ~8 rows
// of pipeline fill and halo whatever its height: frames whose segments would be under 64 rows (960x540 and below) are stacked
// until a strip's segments come to ~128 rows (1080p alone: 135).  Only the segment height decides: a LARGE frame with few units
// (5400x2700: 88 strips x 2 segments) has strips x floor(CUs / strips) units however many frames are stacked, so stacking buys it
// nothing and would double its arenas (such frames used to get batch 2, and past 2 GiB of canvas lost the fused
// kernels for it).  The stacked canvas must stay inside the pair kernel's 32-bit offsets; fewer frames are stacked until it does.
int frames_per_launch(int w, int h, int n_cu)
{
    if (w <= 0 || h < 4 || n_cu <= 0) return 1;
    const int strips = (w + PAIR_VALID - 1) / PAIR_VALID, segs = std::max(1, n_cu / strips);
    const int seg_h = std::max(16, ((h + segs - 1) / segs + 1) & ~1);
    if (seg_h >= 64) return 1;
    int batch = std::min(MAX_BATCH, std::max(2, (128 * segs + h) / (h + 1)));
    const long long Wp = (long long)((w + TILE_W - 1) / TILE_W) * TILE_W + 2;
    while (batch > 1 && ((long long)batch * (h + 1) + 1 + TILE_H + 2) * Wp * PIX_BYTES >= (1ll << 31)) --batch;
    return batch;
}

}  // namespace reve
