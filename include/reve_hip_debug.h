/*
 * reve_hip_debug.h — test probes of libreve_hip.so.  Exported by the same library as include/reve_hip.h, kept out of that
 * header because nothing a reve binding needs is here: parity tests at layer granularity and the layout
 * arithmetic the CPU tests check without a GPU.  Plain C like the main header.
 */
#ifndef REVE_HIP_DEBUG_H
#define REVE_HIP_DEBUG_H

#include "reve_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Parity probe for kernel-level tests: runs conv_first and the first `layer` body layers on the
 * frame (whole-frame geometry, tile ignored) and returns the activation after layer `layer`
 * (0 = conv_first+PReLU, 1..16 = body conv+PReLU) as w*h*64 floats, logical channel order;
 * layer 17 = conv_last's fp16 output before PixelShuffle, residual and quantisation: w*h*3*scale^2 floats. */
int reve_debug_run_layers(reve_ctx* ctx, const uint8_t* src, int w, int h, ptrdiff_t src_stride,
                          int layer, float* out, size_t out_floats);


/* Test probe, needs no GPU: the order in which the kernels visit the tiles of a whole frame of tiles_x x tiles_y
 * tiles (4x8 blocks; the kernels compute it, the engine's work lists for tiled frames are built the same way).
 * out[i] = tx | ty << 10 of work item i, tiles_x*tiles_y entries. */
int reve_debug_blocked_order(int tiles_x, int tiles_y, uint32_t* out);

/* Test probe, needs no GPU: the address budget of the layout a w x h frame gets with `tile` / `prepad` (tile 0: one plane).
 * out5 = {planes, canvas pitch in pixels, canvas height in pixels, bytes of one activation arena, largest byte offset a
 * kernel forms inside one plane}.  Returns 0, REVE_E_INVALID, or REVE_E_UNSUPPORTED when that offset reaches 2 GiB — the
 * same answer reve_upscale_* gives for the geometry (e.g. 7680x4320 with tile 2160). */
int reve_debug_geometry(int w, int h, int tile, int prepad, long long* out5);

/* Test probe, needs no GPU: byte offset, inside a ring row of the Winograd pair kernel (option "winograd"), of the 16-byte
 * chunk `chunk` (0..7) of pixel column `column` (0..65) — the layout whose tile reads are free of LDS bank conflicts
 * (reve_amd/csrc/kernels_wino.hip, kw_ring_off).  Negative on arguments outside those ranges. */
int reve_debug_wino_ring_offset(int column, int chunk);

/* Test probe, needs no GPU: how many frames of w x h share one kernel chain on a device with `compute_units` CUs (256 on an
 * MI355X) under option "batch" — 1: every frame has its own launches (1080p and every larger frame); up to 16 for frames whose
 * pair-kernel segments would be under 64 rows.  The stacked canvas never reaches 2 GiB. */
int reve_debug_frames_per_launch(int w, int h, int compute_units);

/* Test probe, needs no GPU: the conditioning estimate behind option "winograd" = 2 (auto) for an in-memory ncnn model — the fp16
 * storage noise the weights carry to the 8-bit output, in LSB rms (reve_amd/csrc/model.h: conditioning_kappa); auto enables
 * the Winograd pairs below *limit (0.5).  Returns 0, REVE_E_INVALID or REVE_E_MODEL. */
int reve_debug_model_conditioning(const void* param_data, size_t param_len, const void* bin_data, size_t bin_len,
                                  double* kappa, double* limit);

#ifdef __cplusplus
}
#endif
#endif /* REVE_HIP_DEBUG_H */
