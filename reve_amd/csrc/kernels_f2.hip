// Fused two-layer kernels for gfx950: two 3x3 convolutions per launch, the intermediate tile in LDS.
//
// Why: a body layer moves 531 MB of fp16 activations through HBM for 153 GFLOP, and what bounds the
// layer-per-launch kernel is the CU's vector-memory path (~7-10 B/clk/CU for its loads + stores,
// DESIGN.md §4).  Keeping every other activation on chip cuts the bytes per layer to ~0.6x:
//     k_f2<FIRST, 0>   conv_first (+pre-process)  ->  body conv      (u8 frame in, arena out)
//     k_f2<BODY, 0>    body conv                  ->  body conv      (arena in, arena out)
//     k_f2<BODY, s>    body conv                  ->  conv_last + PixelShuffle + residual + post-process
// Same ncnn layers as kernels.hip (reve-shared/src/lib.rs:134-147 spawns the binary that runs them).
//
// Geometry: a workgroup (4 waves, one per SIMD) owns an 8x30 OUTPUT tile of the second layer.
//   IN  = 12 x 34 input pixels of the first layer   (LDS, 2 x 52,224 B: double-buffered LDS-DMA, so
//         the next tile's loads spread over BOTH phases of the current tile)
//   MID = 10 x 32 output pixels of the first layer  (LDS, 40,960 B; 32 = exactly two 16-pixel MFMA
//         column blocks; pixels outside the image are stored as ZERO = the second layer's padding)
//   OUT = 8 x 30 pixels (the 2 surplus columns of each 32-lane row are computed and dropped)
// MFMA work is 1.2x the unfused path (halo recompute), memory traffic per layer about 0.6x.
// Arena layout for this path: 2-pixel zero border, image pixel (0,0) at arena pixel (2,2),
// pitch tiles_x*30+4 pixels.  Channel order inside a pixel: chan_phys() as in kernels.hip.
//
// Both layers' A fragments (weights) stay in registers for the whole persistent launch: the first
// layer's 144 in VGPRs, the second layer's 144 parked in AGPRs (pinned with an "a" constraint once;
// hipcc then feeds them to v_mfma straight from the accumulator file).
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace reve {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

#define MFMA_INIT(acc, w, b, bias) (acc) = __builtin_amdgcn_mfma_f32_16x16x32_f16((w), (b), (bias), 0, 0, 0)
#define MFMA_ACC(acc, w, b) (acc) = __builtin_amdgcn_mfma_f32_16x16x32_f16((w), (b), (acc), 0, 0, 0)

namespace {

// one LDS-DMA piece: 64 lanes x 16 B from rsrc[voff + soff] to LDS base + lane*16 (wrapped in a plain
// device function: used directly inside the kernel template the builtin breaks host-side instantiation)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, lds_void_t* dst, int /*bytes*/, int voff, int soff, int, int)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ lds_void_t* to_lds(char* p)
{
    return (lds_void_t*)(__attribute__((address_space(3))) char*)p;
}

__device__ __forceinline__ h8 prelu8(h8 x, h8 slope)
{
    const h8 z = (h8)(_Float16)0;
    return __builtin_elementwise_fma(slope, __builtin_elementwise_min(x, z), __builtin_elementwise_max(x, z));
}

constexpr int IN_H = F2_TILE_H + 4, IN_W = F2_TILE_W + 4;       // 12 x 34
constexpr int MID_H = F2_TILE_H + 2, MID_W = F2_TILE_W + 2;     // 10 x 32
constexpr int IN_PIX = IN_H * IN_W;                             // 408
constexpr int IN_BYTES = IN_PIX * PIX_BYTES;                    // 52,224
constexpr int MID_BYTES = MID_H * MID_W * PIX_BYTES + 2 * PIX_BYTES;   // 40,960 + over-read slack
constexpr int F2_PIECES = IN_PIX / 8;                           // 51 one-KiB DMA pieces, exact
constexpr int F2_DMA_PER_WAVE = (F2_PIECES + 3) / 4;            // 13 (pieces past 50 re-load piece 50)
constexpr int FIRST_IN_BYTES = IN_PIX * 8;                      // fp16x4 per pixel
constexpr int RA = MID_H / 2;                                   // first-layer rows per wave half (5)
constexpr int PBA = 5;                                          // px-blocks per first-layer sub-iteration
constexpr int NSUB_A = RA * 2 / PBA;                            // 2
static_assert(IN_PIX % 8 == 0 && RA * 2 == NSUB_A * PBA, "tile geometry");
static_assert(2 * IN_BYTES + MID_BYTES <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int f2_piece(int k, int wave)
{
    const int c = k * 4 + wave;
    return c < F2_PIECES ? c : F2_PIECES - 1;
}

struct Item2 { int plane, ty, tx; };
__device__ __forceinline__ Item2 decode2(int it, const F2Args& a)
{
    if (a.reverse) it = a.n_items - 1 - it;
    const int per = a.tiles_x * a.tiles_y;
    Item2 r;
    r.plane = it / per;
    const int rem = it - r.plane * per;
    r.ty = rem / a.tiles_x;
    r.tx = rem - r.ty * a.tiles_x;
    return r;
}

}  // namespace

// LA: 0 = conv_first (3->64, from the u8 frame), 1 = body conv (64->64, from the arena)
// SCALE: 0 = second layer is a body conv; 2/3/4 = second layer is conv_last of that scale
template <int LA, int SCALE>
__global__ void __launch_bounds__(256, 1) k_f2(const F2Args a, const PlaneDesc* __restrict__ planes)
{
    constexpr int NCOB_B = SCALE == 0 ? 4 : (SCALE == 2 ? 1 : (SCALE == 3 ? 2 : 4));
    constexpr int COSPLIT_B = NCOB_B >= 2 ? 2 : 1;
    constexpr int CPW_B = NCOB_B / COSPLIT_B;
    constexpr int ROWS_B = F2_TILE_H / (COSPLIT_B == 2 ? 2 : 4);   // output rows per wave: 4 or 2
    constexpr int NSUB_B = ROWS_B / 2;                             // sub-iterations of 4 px-blocks: 2 or 1
    constexpr int KS_A = LA == 0 ? 2 : KSTEPS;
    constexpr int KS_TOTAL = NSUB_A * KSTEPS + NSUB_B * KSTEPS;    // k-step slots of one tile
    constexpr int DMA_SPAN = KS_TOTAL - KSTEPS;                    // DMA issue window: all but the last sub-iteration

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const MID = smem + 2 * IN_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    const int rhA = wave & 1, chA = wave >> 1;                               // first layer: rows RA*rhA.., channels 32chA..
    const int rowB0 = COSPLIT_B == 2 ? ROWS_B * (wave & 1) : ROWS_B * wave;  // second layer: first output row
    const int chB = COSPLIT_B == 2 ? (wave >> 1) : 0;

    // ---- register-stationary weights of both layers
    h8 wA[KS_A][2];
    h8 wB[KSTEPS][CPW_B];
    {
        const h8* wp = (const h8*)a.wA;
#pragma unroll
        for (int s = 0; s < KS_A; ++s)
#pragma unroll
            for (int m = 0; m < 2; ++m) wA[s][m] = wp[(s * 4 + 2 * chA + m) * 64 + lane];
        const h8* wq = (const h8*)a.wB;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
            for (int m = 0; m < CPW_B; ++m) wB[s][m] = wq[(s * NCOB_B + chB * CPW_B + m) * 64 + lane];
    }
    f4 biasA[2], biasB[CPW_B];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const h4 b = *(const h4*)(a.biasA + 16 * (2 * chA + m) + 4 * g);
        biasA[m] = (f4){(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
    }
#pragma unroll
    for (int m = 0; m < CPW_B; ++m) {
        const h4 b = *(const h4*)(a.biasB + 16 * (chB * CPW_B + m) + 4 * g);
        biasB[m] = (f4){(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
    }
    h8 slopeA, slopeB = (h8)(_Float16)0;
    {
        const h4 s0 = *(const h4*)(a.slopeA + 32 * chA + 4 * g), s1 = *(const h4*)(a.slopeA + 32 * chA + 16 + 4 * g);
        slopeA = __builtin_shufflevector(s0, s1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    if constexpr (SCALE == 0) {
        const h4 s0 = *(const h4*)(a.slopeB + 32 * chB + 4 * g), s1 = *(const h4*)(a.slopeB + 32 * chB + 16 + 4 * g);
        slopeB = __builtin_shufflevector(s0, s1, 0, 1, 2, 3, 4, 5, 6, 7);
    }

    // ---- lane-constant LDS offsets
    int roffA[3][2], roffB[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int sw = 16 * ((4 * hf + g) ^ ((pl + dx) & 6));
            roffA[dx][hf] = (RA * rhA * IN_W + pl + dx) * PIX_BYTES + sw;
            roffB[dx][hf] = (rowB0 * MID_W + pl + dx) * PIX_BYTES + sw;
        }
    const int woffA = (RA * rhA * MID_W + pl) * PIX_BYTES + 16 * ((4 * chA + g) ^ (pl & 6));   // MID write

    // ---- lane-constant DMA source offsets (body input)
    int voff[LA == 1 ? F2_DMA_PER_WAVE : 1];
    if constexpr (LA == 1) {
#pragma unroll
        for (int k = 0; k < F2_DMA_PER_WAVE; ++k) {
            const int q = f2_piece(k, wave) * 8 + (lane >> 3);
            const int iy = q / IN_W, ix = q - iy * IN_W;
            voff[k] = (iy * a.Wp + ix) * PIX_BYTES + 16 * ((lane & 7) ^ (ix & 6));
        }
    }

    const int G = gridDim.x, b = blockIdx.x;
    int it = ((G & 7) == 0) ? (b & 7) * (G >> 3) + (b >> 3) : b;
    int cur = 0;   // input double buffer

    // u8 frame -> fp16x4 staging of a 12x34 window (conv_first input), 2 pixels per thread
    constexpr int STG = (IN_PIX + 255) / 256;
    auto stage_load = [&](const Item2& t, const PlaneDesc& pd, h4 (&v)[STG]) {
#pragma unroll
        for (int j = 0; j < STG; ++j) {
            const int q = tid + 256 * j;
            const int iy = q / IN_W, ix = q - iy * IN_W;
            const int py = t.ty * F2_TILE_H + iy - 2, px = t.tx * F2_TILE_W + ix - 2;
            v[j] = (h4)(_Float16)0;
            if (q < IN_PIX && py >= 0 && py < pd.h && px >= 0 && px < pd.w) {
                int fy = pd.y0 + py, fx = pd.x0 + px;   // ncnn-compat apron: replicate the frame border
                fy = fy < 0 ? 0 : (fy >= a.frame_h ? a.frame_h - 1 : fy);
                fx = fx < 0 ? 0 : (fx >= a.frame_w ? a.frame_w - 1 : fx);
                const uint8_t* sp = a.src + (long long)fy * a.src_stride + fx * 3;
                v[j][0] = (_Float16)((float)sp[0] * (1.0f / 255.0f));
                v[j][1] = (_Float16)((float)sp[1] * (1.0f / 255.0f));
                v[j][2] = (_Float16)((float)sp[2] * (1.0f / 255.0f));
            }
        }
    };
    auto stage_store = [&](int buf, const h4 (&v)[STG]) {
#pragma unroll
        for (int j = 0; j < STG; ++j) {
            const int q = tid + 256 * j;
            if (q < IN_PIX) *(h4*)(smem + buf * FIRST_IN_BYTES + q * 8) = v[j];
        }
    };

    if (it < a.n_items) {
        const Item2 t0 = decode2(it, a);
        if constexpr (LA == 1) {
            auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (unsigned long long)t0.plane * a.plane_stride),
                                                          0, (int)a.plane_stride, 0x00020000);
            const int org = ((t0.ty * F2_TILE_H) * a.Wp + t0.tx * F2_TILE_W) * PIX_BYTES;
#pragma unroll
            for (int k = 0; k < F2_DMA_PER_WAVE; ++k)
                dma16(rsrc, to_lds(smem + f2_piece(k, wave) * 1024), 16, voff[k], org, 0, 0);
        } else {
            h4 v[STG];
            stage_load(t0, planes[t0.plane], v);
            stage_store(0, v);
        }
    }
    // pin the waits for the weight loads before the loop (see kernels.hip); the second layer's
    // fragments are parked in the accumulator file
#pragma unroll
    for (int s = 0; s < KS_A; ++s)
#pragma unroll
        for (int m = 0; m < 2; ++m) asm volatile("" : "+v"(wA[s][m]));
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int m = 0; m < CPW_B; ++m) asm volatile("" : "+a"(wB[s][m]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    while (it < a.n_items) {
        const Item2 itm = decode2(it, a);
        const PlaneDesc pd = planes[itm.plane];
        const int nxt = it + G;
        __builtin_amdgcn_s_barrier();      // IN[cur] of this tile is complete; MID and IN[cur^1] are free
        asm volatile("" ::: "memory");
        char* const IN = smem + (LA == 1 ? cur * IN_BYTES : 0);
        // next tile's input: on the last tile the (unused) re-load of the same tile keeps the body branch-free
        const Item2 nitm = decode2(nxt < a.n_items ? nxt : it, a);
        auto nrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (unsigned long long)nitm.plane * a.plane_stride),
                                                       0, (int)a.plane_stride, 0x00020000);
        const int norg = ((nitm.ty * F2_TILE_H) * a.Wp + nitm.tx * F2_TILE_W) * PIX_BYTES;
        char* const NIN = smem + (cur ^ 1) * IN_BYTES;
        // DMA pieces of the next tile are spread over the k-step slots of BOTH layers (slot gs)
        auto dma_slot = [&](int gs) {
#ifdef ABL_NO_DMA
            return;
#endif
            if constexpr (LA == 1) {
#pragma unroll
                for (int k = 0; k < F2_DMA_PER_WAVE; ++k)
                    if (k * DMA_SPAN / F2_DMA_PER_WAVE == gs)
                        dma16(nrsrc, to_lds(NIN + f2_piece(k, wave) * 1024), 16, voff[k], norg, 0, 0);
            }
        };
        h4 stg[STG];
        if constexpr (LA == 0) stage_load(nitm, planes[nitm.plane], stg);   // next tile's u8 window -> registers

        // ================= first layer: IN -> MID (10 x 32 pixels, 5 rows per wave) =================
        const int ybase = itm.ty * F2_TILE_H - 1 + RA * rhA;     // image row of this wave's MID row 0
        const int xlane = itm.tx * F2_TILE_W - 1 + pl;           // image column of MID column pl
#pragma unroll
        for (int si = 0; si < NSUB_A; ++si) {
            f4 acc[2][PBA];
            if constexpr (LA == 1) {
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    const int t = ks >> 1, hf = ks & 1, dy = t / 3, dx = t % 3;
                    h8 B[PBA];
#pragma unroll
                    for (int q = 0; q < PBA; ++q) {
                        const int pb = si * PBA + q, r = pb >> 1, xb = pb & 1;
                        B[q] = *(const h8*)(IN + roffA[dx][hf] + ((r + dy) * IN_W + 16 * xb) * PIX_BYTES);
                    }
                    dma_slot(si * KSTEPS + ks);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int q = 0; q < PBA; ++q) {
                            if (ks == 0) MFMA_INIT(acc[m][q], wA[0][m], B[q], biasA[m]);
                            else MFMA_ACC(acc[m][q], wA[ks][m], B[q]);
                        }
                }
            } else {
                // conv_first: k = 8g + j <-> tap 2g + (j>>2), channel j&3; tap 8 lives in g == 0 of k-step 1
                const h4* tile = (const h4*)(smem + cur * FIRST_IN_BYTES);
                const int t0 = 2 * g, t1 = 2 * g + 1;
                const int q0 = (t0 / 3) * IN_W + (t0 % 3), q1 = (t1 / 3) * IN_W + (t1 % 3), q8 = 2 * IN_W + 2;
#pragma unroll
                for (int q = 0; q < PBA; ++q) {
                    const int pb = si * PBA + q, r = RA * rhA + (pb >> 1), xb = pb & 1;
                    const int qb = r * IN_W + 16 * xb + pl;
                    const h4 lo = tile[qb + q0], hi = tile[qb + q1];
                    h4 l8 = tile[qb + q8];
                    if (g != 0) l8 = (h4)(_Float16)0;
                    const h8 B0 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    const h8 B1 = __builtin_shufflevector(l8, (h4)(_Float16)0, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        MFMA_INIT(acc[m][q], wA[0][m], B0, biasA[m]);
                        MFMA_ACC(acc[m][q], wA[1][m], B1);
                    }
                }
            }
            // epilogue: fp16 round, PReLU, ZERO outside the image, 16 bytes per lane into MID
#pragma unroll
            for (int q = 0; q < PBA; ++q) {
                const int pb = si * PBA + q, r = pb >> 1, xb = pb & 1;
                h8 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    o[c] = (_Float16)acc[0][q][c];
                    o[4 + c] = (_Float16)acc[1][q][c];
                }
                o = prelu8(o, slopeA);
                const int y = ybase + r, x = xlane + 16 * xb;
                const bool ok = y >= 0 && y < pd.h && x >= 0 && x < pd.w;
                u32x4 ov = __builtin_bit_cast(u32x4, o);
                if (!ok) ov = (u32x4){0u, 0u, 0u, 0u};
                *(u32x4*)(MID + woffA + (r * MID_W + 16 * xb) * PIX_BYTES) = ov;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // MID complete
        asm volatile("" ::: "memory");

        // ================= second layer: MID -> arena / frame (8 x 30 pixels) =================
        auto orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (unsigned long long)itm.plane * a.plane_stride),
                                                       0, (int)a.plane_stride, 0x00020000);
        auto drsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.dst, 0, SCALE ? (int)(a.dst_stride * a.frame_h * (SCALE ? SCALE : 1)) : 0, 0x00020000);
        // conv_last: residual (nearest-upsampled input) bytes of this wave's pixels, requested early
        unsigned resid[SCALE ? NSUB_B * 4 : 1][SCALE == 2 ? 1 : 3];
        if constexpr (SCALE != 0) {
            auto srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, (int)(a.src_stride * a.frame_h), 0x00020000);
#pragma unroll
            for (int pb = 0; pb < NSUB_B * 4; ++pb) {
                const int oy = itm.ty * F2_TILE_H + rowB0 + (pb >> 1), ox = itm.tx * F2_TILE_W + 16 * (pb & 1) + pl;
                int fy = pd.y0 + oy, fx = pd.x0 + ox;
                fy = fy < 0 ? 0 : (fy >= a.frame_h ? a.frame_h - 1 : fy);
                fx = fx < 0 ? 0 : (fx >= a.frame_w ? a.frame_w - 1 : fx);
                const int off = fy * (int)a.src_stride + fx * 3;
                if constexpr (SCALE == 2) {
                    resid[pb][0] = __builtin_amdgcn_raw_buffer_load_b8(srsrc, off + (g < 3 ? g : 0), 0, 0);
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) resid[pb][c] = __builtin_amdgcn_raw_buffer_load_b8(srsrc, off + c, 0, 0);
                }
            }
        }

#pragma unroll
        for (int si = 0; si < NSUB_B; ++si) {
            f4 acc[CPW_B][4];
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int t = ks >> 1, hf = ks & 1, dy = t / 3, dx = t % 3;
                h8 B[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 2 * si + (q >> 1), xb = q & 1;
                    B[q] = *(const h8*)(MID + roffB[dx][hf] + ((r + dy) * MID_W + 16 * xb) * PIX_BYTES);
                }
                dma_slot((NSUB_A + si) * KSTEPS + ks);
#pragma unroll
                for (int m = 0; m < CPW_B; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (ks == 0) MFMA_INIT(acc[m][q], wB[0][m], B[q], biasB[m]);
                        else MFMA_ACC(acc[m][q], wB[ks][m], B[q]);
                    }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 2 * si + (q >> 1), xb = q & 1;
                const int oy = itm.ty * F2_TILE_H + rowB0 + r;
                const int ox = itm.tx * F2_TILE_W + 16 * xb + pl;
                const bool col_ok = (16 * xb + pl) < F2_TILE_W;
                if constexpr (SCALE == 0) {
                    h8 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        o[c] = (_Float16)acc[0][q][c];
                        o[4 + c] = (_Float16)acc[1][q][c];
                    }
                    o = prelu8(o, slopeB);
                    const bool ok = col_ok && oy < pd.h && ox < pd.w;
                    const int off = ((oy + 2) * a.Wp + (ox + 2)) * PIX_BYTES + 64 * chB + 16 * g;
#ifdef ABL_EPI_NOSTORE
                    asm volatile("" ::"v"(o), "v"(ok ? off : 0x7fffffff));
#else
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc, ok ? off : 0x7fffffff, 0, 0);
#endif
                } else {
                    const bool inside = col_ok && oy >= a.pad && oy < pd.h - a.pad && ox >= a.pad && ox < pd.w - a.pad;
                    const int fy = pd.y0 + oy, fx = pd.x0 + ox;
                    const int pbi = si * 4 + q;
#pragma unroll
                    for (int m = 0; m < CPW_B; ++m)
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {
                            const int co = 16 * (chB * CPW_B + m) + 4 * g + c4;
                            const int c = co / (SCALE * SCALE), ij = co % (SCALE * SCALE);
                            const int i = ij / SCALE, j = ij % SCALE;
                            const float v = (float)(_Float16)acc[m][q][c4];
                            const unsigned rb = (SCALE == 2) ? resid[pbi][0] : (c == 0 ? resid[pbi][0] : (c == 1 ? resid[pbi][1] : resid[pbi][2]));
                            const float res = (float)(_Float16)((float)rb * (1.0f / 255.0f));
                            const float o = (float)(_Float16)(v + res);
                            float qv = o * 255.0f + 0.5f;
                            qv = qv > 0.f ? qv : 0.f;
                            qv = qv > 255.f ? 255.f : qv;
                            const bool ok = inside && co < 3 * SCALE * SCALE;
                            const int off = (fy * SCALE + i) * (int)a.dst_stride + (fx * SCALE + j) * 3 + c;
                            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)qv, drsrc, ok ? off : 0x7fffffff, 0, 0);
                        }
                }
            }
        }
        if constexpr (LA == 0) {
            stage_store(cur ^ 1, stg);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (SCALE == 0 && NSUB_B > 1) {
            // every DMA piece was issued before the second layer's 8 stores (none in its last sub-iteration)
#if defined(ABL_EPI_NOSTORE)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        cur ^= 1;
        it = nxt;
    }
}

template __global__ void k_f2<0, 0>(const F2Args, const PlaneDesc* __restrict__);
template __global__ void k_f2<1, 0>(const F2Args, const PlaneDesc* __restrict__);
template __global__ void k_f2<1, 2>(const F2Args, const PlaneDesc* __restrict__);
template __global__ void k_f2<1, 3>(const F2Args, const PlaneDesc* __restrict__);
template __global__ void k_f2<1, 4>(const F2Args, const PlaneDesc* __restrict__);

// -------------------------------------------------------------------------------------------
template <typename K>
static int set_lds2(K k)
{
    return (int)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IN_BYTES + MID_BYTES);
}

int prepare_f2_kernels()
{
    return set_lds2(k_f2<0, 0>) | set_lds2(k_f2<1, 0>) | set_lds2(k_f2<1, 2>) | set_lds2(k_f2<1, 3>) | set_lds2(k_f2<1, 4>);
}

int launch_f2(const F2Args& a, int first_is_conv_first, int scale_last, int grid, void* stream)
{
    const size_t lds = 2 * IN_BYTES + MID_BYTES;
    hipStream_t st = (hipStream_t)stream;
    if (first_is_conv_first) {
        if (scale_last != 0) return -1;
        hipLaunchKernelGGL((k_f2<0, 0>), dim3(grid), dim3(256), lds, st, a, a.planes);
    } else {
        switch (scale_last) {
        case 0: hipLaunchKernelGGL((k_f2<1, 0>), dim3(grid), dim3(256), lds, st, a, a.planes); break;
        case 2: hipLaunchKernelGGL((k_f2<1, 2>), dim3(grid), dim3(256), lds, st, a, a.planes); break;
        case 3: hipLaunchKernelGGL((k_f2<1, 3>), dim3(grid), dim3(256), lds, st, a, a.planes); break;
        case 4: hipLaunchKernelGGL((k_f2<1, 4>), dim3(grid), dim3(256), lds, st, a, a.planes); break;
        default: return -1;
        }
    }
    return (int)hipGetLastError();
}

}  // namespace reve
