// Per-context engine: device weights, activation arenas, the 18-launch frame chain and the
// 3-stream (H2D / compute / D2H) frame ring.  Host-side C++ above the HIP runtime; the C ABI in
// capi.cpp is a thin shell over this class.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "kernels.h"
#include "model.h"

#include <mutex>

namespace reve {

std::mutex& unsafe_calls_mutex();     // engine.cpp: serialises stream captures with the library's own allocations / synchronous copies

// address budget of a frame layout, no GPU needed (engine.cpp)
int frame_geometry(int w, int h, int tile, int prepad, long long out[5]);
// frames of one size that share a kernel chain on a device with n_cu compute units (1: none do), no GPU needed (engine.cpp)
int frames_per_launch(int w, int h, int n_cu);

struct EngineConfig {
    int scale = 2, device = 0, tile = 0, prepad = 10, ring_depth = 3;
};

struct Stats {
    uint64_t frames_done = 0, body_launches = 0;
    double body_ms_total = 0, frame_ms_last = 0;
    uint64_t h2d_bytes = 0, d2h_bytes = 0;
    int compute_units = 0, frame_w = 0, frame_h = 0, planes = 0, tiles_per_plane = 0, body_layers_per_launch = 1;
    // per-kernel split of the chain (profiling on): conv_first, conv_last and the whole chain, summed over frames_timed frames
    uint64_t frames_timed = 0;
    double first_ms_total = 0, last_ms_total = 0, frame_ms_total = 0;
    // the submit/wait ring (profiling on): per-stage device time summed over ring_frames frames, and the host wall
    // time from the first submit to the last wait since the last reset
    uint64_t ring_frames = 0;
    double h2d_ms_total = 0, chain_ms_total = 0, d2h_ms_total = 0, ring_wall_ms = 0;
};

class Engine {
public:
    Engine() = default;
    ~Engine();
    // all methods return 0 or a negative REVE_E_* code; err() has the detail text
    // upload_weights = false: the blob of packed weights is allocated but left empty; a group's creator fills it from
    // the first engine's copy (broadcast_weights / copy_weights_from)
    int init(const EngineConfig& cfg, const Model& model, bool upload_weights = true);
    // The packed weights of all 18 layers are ONE device allocation (~1.3 MB): what a multi-GPU group broadcasts.
    void* weights_ptr() const { return d_weights_; }
    size_t weights_bytes() const { return weights_bytes_; }
    int device() const { return cfg_.device; }
    // PCI address of the GPU ("0000:c1:00.0"; "" if unknown): where the host pipeline looks up the CPUs and NUMA node next to it
    const std::string& pci_bus_id() const { return bus_id_; }
    int copy_weights_from(const Engine& src);   // hipMemcpyPeer (also device-to-device on one GPU)
    int upscale_host(const uint8_t* src, int w, int h, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds);
    int upscale_device(const void* d_src, int w, int h, ptrdiff_t ss, void* d_dst, ptrdiff_t ds);
    // n frames of one size, all resident on the device: small frames go through the kernel chain several at a time (batch_)
    int upscale_device_batch(int n, const void* const* d_srcs, void* const* d_dsts, int w, int h, ptrdiff_t ss, ptrdiff_t ds);
    int sync();
    int submit(uint64_t id, const uint8_t* src, int w, int h, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds);
    int wait(uint64_t* id);
    int debug_run_layers(const uint8_t* src, int w, int h, ptrdiff_t ss, int layer, float* out, size_t n);
    void set_profiling(bool on) { profiling_ = on; }
    // run-time switches (reve_set_option): "fuse_pairs" 0/1 — body layers two per launch (kernels_pair.hip, whole-frame mode);
    // "graph" 0/1 — the submit/wait ring launches each frame's kernel chain as one captured hipGraph; ... (include/reve_hip.h)
    int set_option(const std::string& name, int value);
    int get_option(const std::string& name, int* value) const;
    bool profiling() const { return profiling_; }
    int get_stats(Stats& s);
    int reset_stats();
    const std::string& err() const { return err_; }
    int scale() const { return cfg_.scale; }

private:
    struct Slot {
        void* d_in = nullptr; void* d_out = nullptr;
        size_t in_cap = 0, out_cap = 0;
        void* ev_h2d = nullptr; void* ev_comp = nullptr; void* ev_d2h = nullptr;   // stage ends (chain the streams)
        void* ev_h2d0 = nullptr; void* ev_comp0 = nullptr; void* ev_d2h0 = nullptr; // stage starts (profiling only)
        bool timed = false;
        uint8_t* dst = nullptr; ptrdiff_t dst_stride = 0; int w = 0, h = 0;      // where its download goes (kept for a deferred launch)
        bool launched = true;        // false: uploaded, waiting for its batch to fill (or for reve_wait) before the chain is launched
        int batch_k = 1;             // frames that shared its kernel chain (its chain time is a k-th of the events' distance)
        int failed = 0;              // REVE_E_*: the launch of its batch failed (flush_pending); reve_wait returns it for this frame
        uint64_t id = 0;
        // the slot's kernel chain captured as a hipGraph (option "graph"): one launch per frame instead of 10-18; valid for
        // the geometry, buffers and switches it was captured with
        void* graph_exec = nullptr;
        int g_w = 0, g_h = 0, g_tile = -1;
        bool g_fuse = false, g_wino = false;
    };
    struct DevLayer { void* wpack = nullptr; uint16_t* bias = nullptr; uint16_t* slope = nullptr; };   // into d_weights_

    int fail(int code, const std::string& what);
    int hipfail(int hiperr, const char* what);
    int configure(int w, int h, bool whole_frame_only);
    int enqueue_chain(const uint8_t* d_src, ptrdiff_t ss, uint8_t* d_dst, ptrdiff_t ds, int stop_after_layer);
    int enqueue_chain_k(const uint8_t* const* d_srcs, uint8_t* const* d_dsts, int k, ptrdiff_t ss, ptrdiff_t ds, int stop_after_layer);
    int submit_frame(uint64_t id, const uint8_t* src, int w, int h, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds);
    int flush_pending();             // launches the chain of the frames submitted but not yet launched, and their downloads
    int launch_batch(const std::vector<size_t>& batch);
    size_t ring_cap() const;         // frames the ring takes before reve_submit answers REVE_E_BUSY
    int ensure_slot(Slot& s, size_t in_bytes, size_t out_bytes);
    void harvest_events(bool all);
    void release_geometry();

    EngineConfig cfg_;
    std::string err_;
    std::string launch_err_;         // the error text of the last failed batch launch (reve_wait quotes it for each of its frames)
    std::string bus_id_;
    bool inited_ = false, profiling_ = false;
    // body layers (2k, 2k+1) in one launch: whole frames, and tiled frames on their canvas of planes.  On by default: faster on
    // every geometry measured against layer-per-launch (1080p 4-8 %, 4K 4 %, tile 200 / 100 / 400 at 1080p 5 / 9 / 8 %; profiles/r03)
    bool fuse_pairs_ = true;
    // conv_last as a rolling-strip kernel (kernels_last.hip) instead of the tile kernel: same bytes, x2 53 us against 63-68 us at
    // 1080p (profiles/r03/ablation_table_last_strip.txt); round 6: tiled frames too (strips of the planes' interiors).  On by default
    bool strip_last_ = true;
    // body pairs by Winograd F(2,3) along the row (kernels_wino.hip): two thirds of the MFMAs of the direct kernel, results
    // within the oracle's tolerance but not bit-identical to the direct path.  Whole frames and canvases, with fuse_pairs on.
    // The DEFAULT is auto (round 6): the Winograd pairs wherever the loaded weights' conditioning leaves room for a second
    // summation order (8-9 % more frames/s on every geometry measured, profiles/r05/bench_box_spread.txt), the direct pairs
    // otherwise; REVE_WINOGRAD=0 / reve_set_option("winograd", 0) pins the direct kernels.
    bool winograd_ = false;         // the evaluation in force
    int winograd_mode_ = 2;         // the setting: 0 off, 1 on, 2 auto (on iff kappa_ < WINOGRAD_KAPPA_LIMIT, model.h)
    double kappa_ = 0;              // conditioning_kappa() of the loaded model
    void apply_winograd_mode(bool announce);
    std::vector<void*> body_wino_;  // per body layer: its Winograd-domain fragments (pack_body_wino)
    // Several small frames per launch (option "batch", env REVE_BATCH, on by default): a frame whose strips x segments would leave
    // the pair kernel's segments under 64 rows or fewer than 200 units (960x540 and below) is laid with up to MAX_BATCH - 1 others
    // of the same size on ONE canvas, one below the other, their 1-pixel borders shared (gutter rows that stay zero: the mechanism
    // of tiled frames), and the chain runs once for all of them: conv_first over all planes, the pairs over the canvas as one tall
    // frame, conv_last's strips per frame.  Same bytes as one frame per launch.  The ring (reve_submit) collects the frames: a
    // chain is launched when batch_ frames are uploaded or reve_wait asks for one of them.
    bool batching_ = true;
    bool ring_auto_ = false;        // EngineConfig::ring_depth was <= 0: the ring's depth follows the batch size
    int batch_ = 1;                 // frames per launch of the current geometry (1: as before)
    int items_per_plane_ = 0;
    std::vector<size_t> pending_;   // ring slots uploaded, chain not launched yet
    bool inject_launch_failure_ = false;   // option "debug_fail_launch": the next batch launch fails after its chain was enqueued (tests)
    bool ring_chain_ = false;       // the chain being enqueued belongs to ring slots (reve_wait counts those frames as done)
    uint64_t unretired_ = 0;        // frames enqueued outside the ring whose completion has not been observed yet (frames_done counts at retire)
    bool use_graph_ = false;        // ring slots replay their chain as a captured hipGraph
    bool capturing_ = false;        // enqueue_chain is being recorded into a graph
    void drop_graphs();
    int pair_strips_ = 0, pair_segs_ = 0, pair_seg_h_ = 0;   // units of the fused-pair kernel for the current geometry
    int pair_w_ = 0, pair_h_ = 0;                             // size of the one plane (or of the canvas of planes) it works on
    uint32_t* d_last_units_ = nullptr;                        // tiled frames: conv_last's strip units over the planes' interiors (LastStripArgs::units)
    int n_last_units_ = 0, last_seg_h_ = 0;
    unsigned char* d_col_ok_ = nullptr;                       // canvas: per frame column, 0 = gutter between planes
    int pair_gut_first_ = 0, pair_gut_period_ = 0;            // canvas: gutter rows first + k * period (frame coordinates)
    int n_cu_ = 0;
    void* stream_ = nullptr; void* s_h2d_ = nullptr; void* s_d2h_ = nullptr;
    void* d_weights_ = nullptr;
    size_t weights_bytes_ = 0;
    DevLayer first_, last_;
    std::vector<DevLayer> body_;
    std::vector<char> body_unit_slopes_;   // per body layer: all 64 PReLU slopes (as stored: fp16) lie in [0, 1]
    int n_body_ = 0;

    // geometry (valid when geo_w_ > 0)
    int geo_w_ = 0, geo_h_ = 0, geo_tile_ = -1;
    bool geo_batching_ = true;
    int n_planes_ = 0, tiles_x_ = 0, tiles_y_ = 0, Wp_ = 0, Hp_ = 0, pad_ = 0;
    size_t plane_stride_ = 0;
    PlaneDesc* d_planes_ = nullptr;
    uint32_t* d_items_ = nullptr;   // work list of non-empty tiles (tile mode, layer-per-launch path)
    int n_items_ = 0;
    bool blocked_order_ = false;    // whole frame: no list, the kernels compute the blocked order
    bool blocked_env_ = true;       // (REVE_LAB=1 REVE_NO_BLOCKED_ORDER=1: row-major work order, an A/B control)
    char* arena_[2] = {nullptr, nullptr};
    int last_arena_ = 0;   // arena holding the output of the last body layer run

    // sync-path staging and the async ring
    Slot sync_slot_;
    std::vector<Slot> ring_;
    size_t ring_head_ = 0, ring_count_ = 0;

    // profiling events (pairs bracketing the body chain / whole chain of a frame)
    struct EvRec { void* b0; void* b1; void* f0; void* f1; bool used; int k; };      // k: frames that shared the chain
    std::vector<EvRec> evpool_;
    size_t ev_next_ = 0;
    Stats stats_;
    double ring_t0_ = 0;   // host clock (ms) of the first submit since the last reset
};

}  // namespace reve
