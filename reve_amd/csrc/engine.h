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

namespace reve {

struct EngineConfig {
    int scale = 2, device = 0, tile = 0, prepad = 10, ring_depth = 3;
    int body = 2;        // body kernel: 2 = k_body2 (row-pipelined, shipped), 1 = k_body (REVE_BODY overrides; A/B reference)
    bool fused = false;  // EXPERIMENTAL (REVE_FUSED=1): two convolutions per launch (kernels_f2.hip)
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
    // weights_from != nullptr: the packed weights are not uploaded from the host again but copied
    // device-to-device (over xGMI between two GPUs) from an initialised engine of the same model
    int init(const EngineConfig& cfg, const Model& model, const Engine* weights_from = nullptr);
    int upscale_host(const uint8_t* src, int w, int h, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds);
    int upscale_device(const void* d_src, int w, int h, ptrdiff_t ss, void* d_dst, ptrdiff_t ds);
    int sync();
    int submit(uint64_t id, const uint8_t* src, int w, int h, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds);
    int wait(uint64_t* id);
    int debug_run_layers(const uint8_t* src, int w, int h, ptrdiff_t ss, int layer, float* out, size_t n);
    void set_profiling(bool on) { profiling_ = on; }
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
        uint64_t id = 0;
    };
    struct DevLayer {
        void* wpack = nullptr; uint16_t* bias = nullptr; uint16_t* slope = nullptr;
        size_t w_bytes = 0, bias_bytes = 0, slope_bytes = 0;
    };

    int fail(int code, const std::string& what);
    int hipfail(int hiperr, const char* what);
    int configure(int w, int h, bool whole_frame_only, bool fused);
    int enqueue_chain_fused(const uint8_t* d_src, ptrdiff_t ss, uint8_t* d_dst, ptrdiff_t ds, int stop_after_layer);
    int enqueue_chain(const uint8_t* d_src, ptrdiff_t ss, uint8_t* d_dst, ptrdiff_t ds, int stop_after_layer);
    int upload_layer(const PackedLayer& p, DevLayer& d);
    int clone_layer(const DevLayer& s, int src_device, DevLayer& d);
    int ensure_slot(Slot& s, size_t in_bytes, size_t out_bytes);
    void harvest_events(bool all);
    void release_geometry();

    EngineConfig cfg_;
    std::string err_;
    bool inited_ = false, profiling_ = false;
    int n_cu_ = 0;
    void* stream_ = nullptr; void* s_h2d_ = nullptr; void* s_d2h_ = nullptr;
    DevLayer first_, last_;
    DevLayer last_f2_;   // conv_last in natural channel order for the experimental fused path
    std::vector<DevLayer> body_;
    int n_body_ = 0;

    // geometry (valid when geo_w_ > 0)
    int geo_w_ = 0, geo_h_ = 0, geo_tile_ = -1;
    bool geo_fused_ = false;
    int border_ = 1;
    int n_planes_ = 0, tiles_x_ = 0, tiles_y_ = 0, Wp_ = 0, Hp_ = 0, pad_ = 0;
    size_t plane_stride_ = 0;
    PlaneDesc* d_planes_ = nullptr;
    uint32_t* d_items_ = nullptr;   // work list of non-empty tiles (tile mode, layer-per-launch path)
    int n_items_ = 0;
    bool blocked_order_ = false;    // whole frame: no list, the kernels compute the blocked order
    char* arena_[2] = {nullptr, nullptr};
    int last_arena_ = 0;   // arena holding the output of the last body layer run

    // sync-path staging and the async ring
    Slot sync_slot_;
    std::vector<Slot> ring_;
    size_t ring_head_ = 0, ring_count_ = 0;

    // profiling events (pairs bracketing the body chain / whole chain of a frame)
    struct EvRec { void* b0; void* b1; void* f0; void* f1; bool used; };
    std::vector<EvRec> evpool_;
    size_t ev_next_ = 0;
    Stats stats_;
    double ring_t0_ = 0;   // host clock (ms) of the first submit since the last reset
};

}  // namespace reve
