// Directory mode: the file contract Video::upscale_segment relies on
// (reve-shared/src/lib.rs:130-147): every frame image in `in_dir` is upscaled to
// `out_dir/<same stem>.png`; one progress callback per finished frame, in name order.
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "../../include/reve_hip.h"
#include "engine.h"

namespace reve {
// engs: one engine per GPU (same scale); frame f of the sorted directory goes to engs[f mod G]
int upscale_dir(const std::vector<Engine*>& engs, const std::string& in_dir, const std::string& out_dir, reve_progress_cb cb,
                void* user, std::string& err);
// CPUs the process may use: affinity mask and control-group CPU quota (dirmode.cpp).  `root` is prefixed to /proc/self/cgroup and
// /sys/fs/cgroup (tests point it at a directory of their own).
int effective_cpus(const std::string& root = "");

int upscale_file(Engine& eng, const std::string& in_path, const std::string& out_path, std::string& err);

// The host pipeline itself (decode pool -> one feeder thread per engine -> encode pool; callbacks in frame order on the
// caller's thread), with the two ends supplied by the caller.  Both functions are called from pool threads, concurrently
// for different frames, and return "" or an error text.
struct FrameIO {
    // produce frame i: call sink(w, h) ONCE to get the w*h*3-byte buffer (pinned when one is free) and fill it with RGB pixels
    std::function<std::string(int i, const std::function<uint8_t*(int, int)>& sink)> decode;
    // consume the upscaled frame i (w x h pixels, tightly packed RGB); the buffer is reused after the call returns
    std::function<std::string(int i, const uint8_t* rgb, int w, int h)> encode;
};
int run_pipeline(const std::vector<Engine*>& engs, int n_frames, const FrameIO& io, const std::function<void(int)>& on_done, std::string& err);
// frees every pinned buffer parked by earlier calls (the last reve_destroy does this); returns the bytes released
size_t pinned_cache_trim();
}  // namespace reve
