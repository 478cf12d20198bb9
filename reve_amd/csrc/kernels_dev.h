// Device-side helpers shared by the kernels (not a public header).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace reve {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ lds_void_t* to_lds(char* p)
{
    return (lds_void_t*)(__attribute__((address_space(3))) char*)p;
}

// PReLU on fp16 storage values with fp32 arithmetic, result rounded to fp16 (ncnn fp16-storage):
// x >= 0 ? x : RNE(x*slope).  max(x,0) + slope*min(x,0) as one packed fma is exactly that, because
// one of the two terms is always zero and the fp16 product is correctly rounded.
__device__ __forceinline__ h8 prelu8(h8 x, h8 slope)
{
    const h8 z = (h8)(_Float16)0;
    h8 pos = __builtin_elementwise_max(x, z);
    h8 neg = __builtin_elementwise_min(x, z);
    return __builtin_elementwise_fma(slope, neg, pos);
}

// The same for a layer whose 64 slopes all lie in [0, 1] (checked on the host, Engine::init): max(x, RNE(slope * x)) — for
// x >= 0 the product cannot exceed x, for x < 0 it cannot fall below it — two packed instructions instead of three (0.9 % of a
// body launch).  Bit-identical to prelu8 except that -0 stays -0 (as in the oracle's `x >= 0 ? x : x * slope`).
__device__ __forceinline__ h8 prelu8_unit_slopes(h8 x, h8 slope)
{
    return __builtin_elementwise_max(x, x * slope);
}

// work item -> (plane, ty, tx), shared by the persistent kernels
struct Item { int plane, ty, tx; };

// Whole-frame work order (one plane): 4-wide x 8-tall blocks of tiles, row-major inside a block and over
// the blocks, so that 32 consecutive items (what one XCD has in flight) are one block and a tile's
// vertical and horizontal halo neighbours hit that XCD's L2.  Computed instead of looked up: a scalar load
// of a list entry that the previous launch's 550 MB have long evicted costs ~3 us at the head of every
// launch.  Must match Engine::configure()'s list order (used for planes of unequal size).
__host__ __device__ __forceinline__ Item decode_blocked(int it, int tiles_x, int tiles_y)
{
    const int full_rows = tiles_y >> 3, per_row = 8 * tiles_x;
    int r = it / per_row, rem = it - r * per_row, bh = 8;
    if (r >= full_rows) { r = full_rows; rem = it - full_rows * per_row; bh = tiles_y - 8 * full_rows; }
    const int per_block = 4 * bh, full_cols = tiles_x >> 2;
    int k = rem / per_block, rem2 = rem - k * per_block, bw = 4;
    if (k >= full_cols) { k = full_cols; rem2 = rem - full_cols * per_block; bw = tiles_x - 4 * full_cols; }
    const int ty = rem2 / bw, tx = rem2 - ty * bw;
    return Item{0, 8 * r + ty, 4 * k + tx};
}

template <typename Args>
__device__ __forceinline__ Item decode_any(int it, const Args& a, const uint32_t* __restrict__ items)
{
    if (items) {
        const uint32_t v = items[it];
        return Item{(int)(v >> 20), (int)((v >> 10) & 1023u), (int)(v & 1023u)};
    }
    if (a.blocked) return decode_blocked(it, a.tiles_x, a.tiles_y);
    const int per = a.tiles_x * a.tiles_y;
    Item r;
    r.plane = it / per;
    const int rem = it - r.plane * per;
    r.ty = rem / a.tiles_x;
    r.tx = rem - r.ty * a.tiles_x;
    return r;
}

__device__ __forceinline__ Item decode_item(int it, const ConvArgs& a, const uint32_t* __restrict__ items)
{
    if (a.reverse) it = a.n_items - 1 - it;
    return decode_any(it, a, items);
}

// -------------------------------------------------------------------------------------------
// LDS image of one (16+2)x(32+2)-pixel input tile: pixel q = row*34 + col at byte 128*q, its eight
// 16-byte channel chunks XOR-swizzled by (col & 6) — with that mask every ds_read_b128 of a B
// fragment (16 consecutive pixels x 2 chunks per 16-lane group) is bank-conflict free for all
// three horizontal taps.  The image is filled by LDS-DMA in 77 linear 1-KiB pieces (8 pixels per
// wave-instruction): lane l -> pixel 8*piece + (l>>3), slot (l&7); the swizzle is applied on the
// per-lane SOURCE address because a DMA's LDS destination is always base + lane*16.
// Each wave owns pieces w, w+4, ...; their per-lane source offsets are tile-invariant and are
// computed once per launch (DMA_PER_WAVE registers).
// -------------------------------------------------------------------------------------------
constexpr int NWAVES = 4;

constexpr int LDS_PIX = LDS_H * LDS_W;                      // 612
constexpr int DMA_PIECES = (LDS_PIX + 7) / 8;               // 77
constexpr int DMA_PER_WAVE = (DMA_PIECES + NWAVES - 1) / NWAVES;   // 20 (pieces past 76 re-load piece 76)
constexpr int LDS_BUF_BYTES = DMA_PIECES * 1024;            // 78,848: tile image + 512 B of slack

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// cache-policy bits of the LDS-DMA loads / activation stores (buffer builtin aux: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef DMA_AUX
#define DMA_AUX 0
#endif
#ifndef STORE_AUX
#define STORE_AUX 0
#endif

// one LDS-DMA piece: 64 lanes x 16 B from rsrc[voff + soff] to LDS base + lane*16 (a plain device function:
// used directly inside a kernel template with lambdas the builtin breaks the host-side instantiation)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, lds_void_t* dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, DMA_AUX);
}
// (the same with the cache policy chosen per call site)
template <int AUX>
__device__ __forceinline__ void dma16a(__amdgpu_buffer_rsrc_t rsrc, lds_void_t* dst, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, AUX);
}

// Status of the launch just issued.  hipGetLastError() is sticky per host thread: a failed HIP call of ANYONE in the process
// (a probe with a bad device ordinal, another library) stays there until read, and would be reported as this launch's failure.
// The launchers therefore drop whatever is pending with launch_prepare() first and read only their own launch's status.
inline void launch_prepare() { (void)hipGetLastError(); }
inline int launch_status() { return (int)hipGetLastError(); }

__device__ __forceinline__ int dma_piece(int k, int wave)
{
    const int c = k * NWAVES + wave;
    return c < DMA_PIECES ? c : DMA_PIECES - 1;
}


}  // namespace reve
