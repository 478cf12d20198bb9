/*
 * reve_hip.h — C ABI of libreve_hip.so, the MI355X-native replacement for the
 * `realesrgan-ncnn-vulkan` child process that ONdraid/reve spawns per segment.
 *
 * Reference interface replaced (paths relative to the reference repo):
 *   - reve-shared/src/lib.rs:129-155  Video::upscale_segment: builds
 *       `realesrgan-ncnn-vulkan -i temp\tmp_frames\<i> -o temp\out_frames\<i>
 *        -n realesr-animevideov3-x2 -s <ratio> -f png -v`, returns the child's stderr.
 *   - reve-cli/src/main.rs:262-273    the caller: counts stderr lines containing "done".
 *   - reve-gui/src-tauri/src/commands.rs:52-65  single-file variant
 *       (`-i file -o file -m models -n realesr-animevideov3-x<f> -s <f>`).
 * The process boundary (argv + directories + stderr text) becomes an in-process
 * library: plain pointers and sizes, no C++ or torch types, no exceptions, never aborts.
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Threading: a reve_ctx may be used by one thread at a time. reve_submit / reve_wait are
 * single-producer / single-consumer on one ctx. Different contexts are independent.
 */
#ifndef REVE_HIP_H
#define REVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REVE_ABI_VERSION 7   /* 2: + reve_create_group, reve_upscale_dir_multi; 3: reve_stats grew (per-stage times), + reve_resolve_model_name; 4: + reve_set_option / reve_get_option, reve_upscale_stream_multi, reve_device_cpulist, reve_bind_thread_to_device, reve_trim; 5: + reve_debug_geometry, reve_debug_wino_ring_offset, reve_upscale_rgb8_device_batch, options "winograd", "batch"; 6: reve_debug_* moved to reve_hip_debug.h (+ reve_debug_frames_per_launch), options "updown" and "xcd_balance" removed, "winograd" 2 = auto, read-only "pair_*" options, reve_stats.frames_done counted at retirement; 7: option "winograd" defaults to 2 (auto), + reve_model_report, reve_wait returns the launch error of a frame whose batch failed */

/* error codes: 0 = success, negative = failure (reve_strerror gives the text) */
enum {
    REVE_OK = 0,
    REVE_E_INVALID = -1,    /* bad argument / bad config                                   */
    REVE_E_MODEL = -2,      /* model files missing, malformed or not SRVGGNetCompact 64x16  */
    REVE_E_NODEVICE = -3,   /* no usable HIP device (the library has NO CPU fallback)       */
    REVE_E_HIP = -4,        /* a HIP runtime call or kernel launch failed                   */
    REVE_E_NOMEM = -5,      /* host or device allocation failed                             */
    REVE_E_IO = -6,         /* file / directory / PNG error                                 */
    REVE_E_BUSY = -7,       /* reve_submit: ring full; reve_wait: nothing in flight         */
    REVE_E_UNSUPPORTED = -8 /* valid request this build does not implement                  */
};

typedef struct reve_ctx reve_ctx;

/*
 * Configuration. Zero-initialise, set struct_size = sizeof(reve_config), fill what you need.
 * Mirrors the command-line options reve passes (lib.rs:136-146) plus what the binary defaults:
 *   scale      <-> -s   (2, 3, 4; selects the matching x2/x3/x4 graph — reve-cli always names
 *                        the x2 model, lib.rs:141, a bug this library does not reproduce)
 *   model_dir  <-> -m   (default "models"), model_name <-> -n   (resolved by reve_resolve_model_name)
 *   tile       <-> -t   0 = whole frame, seam-free (default); N > 0 = the binary's N-pixel
 *                        tiles with a `prepad` apron (the binary auto-picks 200 on large GPUs)
 *   device     <-> -g
 */
typedef struct reve_config {
    uint32_t struct_size;
    int32_t scale;
    int32_t device;          /* HIP device ordinal */
    int32_t tile;
    int32_t prepad;          /* apron in pixels for tile > 0; <= 0 means the binary's 10 */
    int32_t ring_depth;      /* slots of the async submit/wait ring; <= 0: the library chooses — 3, or twice the batch where small
                              * frames share their kernel launches (option "batch": up to 32) */
    const char* model_dir;   /* directory holding <model_name>.param / .bin (NULL if *_data given) */
    const char* model_name;  /* NULL or "realesr-animevideov3" -> "realesr-animevideov3-x<scale>" */
    const void* param_data;  /* optional in-memory model (e.g. received by an RCCL broadcast) */
    size_t param_len;
    const void* bin_data;
    size_t bin_len;
} reve_config;

typedef struct reve_stats {
    uint32_t struct_size;
    uint64_t frames_done;         /* frames whose completion the library has OBSERVED since the last reset: ring frames
                                   * when reve_wait returns them, the others at reve_sync / a synchronous call's return */
    uint64_t body_launches;       /* body-conv launches timed since the last reset (each covers    */
                                  /* body_layers_per_launch 64->64 layers)                         */
    double body_ms_total;         /* sum of their durations (HIP events on the ctx's stream)  */
    double frame_ms_last;         /* device time of the last frame's kernel chain (10 launches) */
    uint64_t h2d_bytes, d2h_bytes;
    int32_t compute_units;        /* multiProcessorCount of the device                        */
    int32_t frame_w, frame_h;     /* geometry the arenas are currently sized for              */
    int32_t planes, tiles_per_plane;
    int32_t body_layers_per_launch; /* 64->64 layers per body launch: 2 (fused pairs), 1 ("fuse_pairs" 0) */
    /* ABI 3 (present when struct_size covers them).  Profiling on: */
    uint64_t frames_timed;        /* frames whose chain was split by events: conv_first | body | conv_last  */
    double first_ms_total, last_ms_total, frame_ms_total;   /* sums over frames_timed frames (device time)    */
    /* the reve_submit / reve_wait ring, per stage (reve-cli shows one progress bar per stage,            */
    /* reve-cli/src/main.rs:176-189): device time of the upload, the kernel chain and the download of      */
    /* each frame, summed over ring_frames frames, and the host wall time from the first reve_submit to    */
    /* the last reve_wait since the last reset.  Overlap efficiency = slowest stage's total / ring_wall_ms. */
    uint64_t ring_frames;
    double h2d_ms_total, chain_ms_total, d2h_ms_total, ring_wall_ms;
} reve_stats;

/* progress callback of directory mode: called once per finished frame, from the calling thread */
typedef void (*reve_progress_cb)(void* user, int frame_index, const char* in_path, const char* out_path);

int reve_abi_version(void);
/* "key=value" pairs separated by blanks: abi, arch, pair_src_sha256 (sha256 of the dominant kernel's sources at build time: what
 * a profile or a traffic measurement quotes to say which code it was taken on).  Static storage. */
const char* reve_build_info(void);
const char* reve_strerror(int code);
int reve_device_count(void);                       /* >= 0, or a negative REVE_E_* */

/* The model file a (name, scale) pair selects, as reve_create resolves it:
 *   NULL / "realesr-animevideov3"      -> "realesr-animevideov3-x<scale>"   (what the binary does)
 *   "realesr-animevideov3-x<k>", k != scale -> "realesr-animevideov3-x<scale>": reve-cli names the x2 model for every
 *       --scale (reve-shared/src/lib.rs:140-143) and the binary then runs the x2 graph on an x3/x4 canvas; the graph
 *       that matches -s is loaded instead.  Returns 1 in this case (callers may want to say so).
 *   anything else                      -> verbatim.
 * Writes the NUL-terminated name to out[0..cap); returns 0 / 1 as above or REVE_E_INVALID. Needs no GPU. */
int reve_resolve_model_name(const char* model_name, int scale, char* out, size_t cap);

/* What the library sees in a model before any frame is upscaled — needs no GPU.  Loads <model_dir>/<resolved name>.param/.bin
 * (the files reve names at reve-shared/src/lib.rs:140-141, reve-gui/src-tauri/src/commands.rs:58-63) and writes a JSON text to
 * out[0..cap): per layer its gain ||W||_F / sqrt(c_out), weight / bias rms, PReLU slope range and the activation rms the estimate
 * carries; "kappa" (the fp16 storage noise the weights carry to the 8-bit output, LSB rms), the limit, and the evaluation option
 * "winograd" = auto (the default) chooses for these weights.  `realesrgan-hip --model-report -m DIR -n NAME -s S` prints it.
 * Returns 0, REVE_E_MODEL (reve_last_error(NULL) says why), or REVE_E_INVALID (bad scale, or cap too small: 16 KiB is enough). */
int reve_model_report(const char* model_dir, const char* model_name, int scale, char* out, size_t cap);

int reve_create(const reve_config* cfg, reve_ctx** out);
/* Multi-GPU (`-g 0,1,2` of realesrgan-ncnn-vulkan, which lib.rs:134-147 does not pass but the binary
 * accepts): one context per entry of devices[0..n), all from ONE parse of the model;
 * the packed weights (one device blob, ~1.3 MB) are uploaded to devices[0] and reach the other GPUs by
 * one RCCL broadcast over xGMI (librccl is loaded on first use; contexts that share a device get a
 * device-to-device copy; REVE_GROUP_BCAST=rccl|peer forces either, `rccl` with a device listed twice is
 * REVE_E_INVALID).  A group of distinct GPUs whose
 * broadcast cannot be set up fails with REVE_E_HIP.  cfg->device is ignored.  out[] receives n contexts,
 * each destroyed with reve_destroy; on failure none is left. */
int reve_create_group(const reve_config* cfg, const int* devices, int n, reve_ctx** out);
void reve_destroy(reve_ctx* ctx);
const char* reve_last_error(reve_ctx* ctx);        /* detail text of the last failure on ctx */

/* One frame, host buffers (8-bit RGB, HWC, row strides in bytes), synchronous. */
int reve_upscale_rgb8(reve_ctx* ctx, const uint8_t* src, int w, int h, ptrdiff_t src_stride,
                      uint8_t* dst, ptrdiff_t dst_stride);

/* One frame, DEVICE buffers on the ctx's device; enqueued on the ctx's stream, returns at once. */
int reve_upscale_rgb8_device(reve_ctx* ctx, const void* d_src, int w, int h, ptrdiff_t src_stride,
                             void* d_dst, ptrdiff_t dst_stride);
/* n frames of ONE size, DEVICE buffers (d_srcs[i] -> d_dsts[i]), enqueued like the call above.  Frames too small to fill the
 * GPU alone (960x540 and below) share their kernel launches, up to 16 at a time, as the async ring does by itself ("batch" below);
 * the bytes are those of n single calls.  (reve's callers never see this: their frames arrive through the ring or a directory.) */
int reve_upscale_rgb8_device_batch(reve_ctx* ctx, int n, const void* const* d_srcs, void* const* d_dsts, int w, int h,
                                   ptrdiff_t src_stride, ptrdiff_t dst_stride);
int reve_sync(reve_ctx* ctx);                      /* wait for everything enqueued on ctx */

/* Async ring (decode/upload, inference, download/encode overlap on separate HIP streams).
 * src/dst must stay valid from submit until the matching wait; pinned memory
 * (reve_alloc_pinned) gives true overlap. Completion order == submission order.
 * reve_submit: REVE_E_BUSY = ring full (the frame was NOT taken: call reve_wait and submit again).  Any other error on a frame whose
 * batch launch failed (small frames share launches): the frame and the others of its batch stay in the ring marked failed, none of
 * them reads `src` any more, and reve_wait returns that error once per such frame (with its id in *id), writes nothing to `dst` and
 * does not count it in reve_stats.frames_done; the context stays usable. */
int reve_submit(reve_ctx* ctx, uint64_t id, const uint8_t* src, int w, int h, ptrdiff_t src_stride,
                uint8_t* dst, ptrdiff_t dst_stride);
int reve_wait(reve_ctx* ctx, uint64_t* id);
void* reve_alloc_pinned(size_t bytes);
void reve_free_pinned(void* p);

/* The directory contract of lib.rs:130-147: every *.png in in_dir -> out_dir/<stem>.png,
 * callback once per finished frame (the executable prints "<in> -> <out> done" from it). */
int reve_upscale_dir(reve_ctx* ctx, const char* in_dir, const char* out_dir,
                     reve_progress_cb cb, void* user);

/* The same over several contexts (one per GPU, same scale): frame f of the sorted directory goes to
 * ctxs[f mod n]; frames are independent, so there is no communication between the GPUs. Callbacks
 * still come once per frame, in name order, from the calling thread. Errors: reve_last_error(ctxs[0]). */
int reve_upscale_dir_multi(reve_ctx* const* ctxs, int n, const char* in_dir, const char* out_dir,
                           reve_progress_cb cb, void* user);

/* Single-file contract of reve-gui (commands.rs:52-65: `-i <file> -o <file>`): one PNG in, one PNG out. */
int reve_upscale_file(reve_ctx* ctx, const char* in_path, const char* out_path);

/* Raw-frame stream through the same host pipeline as directory mode (what an in-process host that decodes video itself
 * would call instead of writing PNGs — reve-shared/src/lib.rs:89-127 exports them with ffmpeg only because the child
 * process reads files): n_frames frames of w x h RGB pixels, frame f on ctxs[f mod n] (one feeder thread, ring and pinned
 * pools per context, bound to the CPUs next to its GPU).
 *   read(user, index, rgb)    fill rgb[0 .. w*h*3) — a PINNED buffer whenever one is free — with frame `index`;
 *   write(user, index, rgb)   consume the upscaled frame, (w*scale) x (h*scale) tightly packed; the buffer is reused afterwards.
 * Both are called from pool threads, concurrently for different frames, and return 0 or non-zero (= that frame failed, REVE_E_IO).
 *   done(user, index, NULL, NULL)   optional, on the calling thread, once per finished frame, in frame order.
 * Returns 0 or the first failure. */
typedef int (*reve_read_frame_cb)(void* user, int index, uint8_t* rgb);
typedef int (*reve_write_frame_cb)(void* user, int index, const uint8_t* rgb);
int reve_upscale_stream_multi(reve_ctx* const* ctxs, int n, int n_frames, int w, int h, reve_read_frame_cb read,
                              reve_write_frame_cb write, reve_progress_cb done, void* user);

/* Host placement helpers ("pinned to the GPU's NUMA node").  reve_device_cpulist writes the kernel's
 * local_cpulist of the GPU's PCI device ("0-15,128-143"; empty string if the box exposes none); reve_bind_thread_to_device
 * restricts the CALLING thread to those CPUs (never beyond its current affinity mask) and returns how many it is bound to
 * (0 = left alone).  A process-per-GPU host calls it before its first pinned allocation; the library's own feeder threads do. */
int reve_device_cpulist(int device, char* out, size_t cap);
int reve_bind_thread_to_device(int device);
/* Pinned host buffers parked by finished directory / stream calls are kept for the next call (pinning 25 MB takes ~8 ms);
 * reve_trim frees them now and returns the number of bytes released.  The last reve_destroy of a process does the same. */
size_t reve_trim(void);

/* Frame-file helpers used by directory mode (8-bit RGB PNG, the format of `frame%08d.png` at
 * lib.rs:93 and main.rs:297-300). reve_png_read allocates *rgb (w*h*3 bytes, tightly packed);
 * release it with reve_free. Both return 0 or REVE_E_IO / REVE_E_NOMEM. Need no GPU. */
int reve_png_read(const char* path, uint8_t** rgb, int* w, int* h);
int reve_png_write(const char* path, const uint8_t* rgb, int w, int h, ptrdiff_t stride);
void reve_free(void* p);

/* Stats / profiling: with profiling on, the 16 body-layer launches of each frame are bracketed
 * by HIP events on the launch stream. */
int reve_set_profiling(reve_ctx* ctx, int enabled);
int reve_get_stats(reve_ctx* ctx, reve_stats* out);
int reve_reset_stats(reve_ctx* ctx);

/* Run-time switches of a context (the binary has no counterpart; reve's callers never need them).  Every switch but "winograd"
 * changes only the launch structure: the output bytes are the same whatever they are set to.  Not to be changed with frames in
 * flight on the ring (REVE_E_BUSY).  The environment does not reach them, REVE_WINOGRAD excepted; a lab session
 * (A/B scripts) may set REVE_LAB=1 to have REVE_FUSE_PAIRS / REVE_STRIP_LAST / REVE_BATCH / REVE_GRAPH read as initial values.
 *   "fuse_pairs"  0 / 1   (default 1) body layers two per launch, the layer between them kept in LDS (whole frames, and tiled
 *                         frames — their planes lie on one canvas with shared zero borders; 0: one layer per launch).
 *   "graph"       0 / 1   (default 0) reve_submit launches each frame's kernel chain as ONE captured hipGraph (per ring slot and
 *                         geometry) instead of 10-18 kernel launches.
 *   "strip_last"  0 / 1   (default 1) conv_last as a rolling-strip kernel instead of the tile kernel (whole frames; tiled frames of
 *                         the x2 / x3 graphs: strips of the planes' interiors).
 *   "batch"       0 / 1   (default 1) frames whose strips x segments cannot fill the GPU (960x540 and below) go through the kernel
 *                         chain several at a time, up to 16, laid one below the other on one canvas: reve_submit holds a frame
 *                         (uploaded) while the GPU is busy, until its batch is full or reve_wait asks for it; with
 *                         reve_config.ring_depth <= 0 the ring then takes twice the batch before it answers REVE_E_BUSY (an explicit
 *                         depth is kept, and caps the batch).  Read-only: "batch_frames" (of the current frame size).
 *   "winograd"    0 / 1 / 2   (default 2 = auto; env REVE_WINOGRAD=0|1|auto) how the fused pairs evaluate their layers: 1 = by
 *                         Winograd F(2,3) along the row (two thirds of the MFMAs; whole frames, tiled frames and batched small
 *                         frames alike), 0 = by the direct sums.  The ONE switch that is not bit-neutral: a different sum, within
 *                         the same tolerance of the CPU oracle (<= 1 LSB per sample, ~0.2 % of the samples) for well-conditioned
 *                         weights; 8-10 % more frames/s on noise frames, 16 % on flat content.  2 = auto, the default since ABI 7:
 *                         Winograd if and only if the loaded weights pass the conditioning rule of DESIGN.md §3 (the fp16 storage
 *                         noise the weights carry to the output, estimated at load, under 0.5 LSB), said once per process on
 *                         stderr in a line reve's frame counter ignores.  REVE_WINOGRAD=0 / reve_set_option("winograd", 0) pins
 *                         the direct kernels.  reve_get_option answers the evaluation in force (0 or 1), "winograd_mode" the
 *                         setting (0 / 1 / 2) and "winograd_kappa_permille" the estimate the rule compared (limit 500).
 *   "debug_fail_launch" 0 / 1  test hook: the next batch launch of the ring fails after its kernel chain was enqueued (REVE_E_HIP), so
 *                         that the error path of reve_wait can be exercised; never set by a caller with real work.
 *   read-only geometry of the fused-pair launch at the current frame size (what bench.py derives its executed-FLOP figure from):
 *                         "pair_units", "pair_strips", "pair_segments", "pair_seg_rows", "pair_mfma_per_launch" (MFMA
 *                         instructions, 16,384 FLOP each, that one body-pair launch executes: strips x segments x steps x waves).
 * Unknown names: REVE_E_INVALID. */
int reve_set_option(reve_ctx* ctx, const char* name, int value);
int reve_get_option(reve_ctx* ctx, const char* name, int* value);

/* Test probes (reve_debug_*: parity at layer granularity, layout arithmetic) are declared in reve_hip_debug.h — exported by the
 * same library, not part of the interface a reve binding needs. */

#ifdef __cplusplus
}
#endif
#endif /* REVE_HIP_H */
