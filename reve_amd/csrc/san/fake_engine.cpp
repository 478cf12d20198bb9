// CPU sanitizer builds only (make san): the host code that parses untrusted files (png.cpp, model.cpp) and runs the
// 70-thread directory pipeline (dirmode.cpp) is compiled unchanged with g++ -fsanitize=... and linked against THIS
// stand-in for engine.cpp: an "engine" whose submit() does a nearest-neighbour upscale on the calling thread and whose
// wait() hands frames back in submission order, plus malloc-backed stand-ins for the two HIP calls dirmode.cpp makes.
// It is not part of libreve_hip.so and computes nothing of the product path.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>

#include "../../../include/reve_hip.h"
#include "../engine.h"

extern "C" hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
extern "C" hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }

namespace reve {

std::mutex& unsafe_calls_mutex()
{
    static std::mutex m;
    return m;
}

Engine::~Engine() {}
int Engine::fail(int code, const std::string& what) { err_ = what; return code; }

int Engine::init(const EngineConfig& cfg, const Model&, bool)
{
    cfg_ = cfg;
    if (cfg_.ring_depth <= 0) cfg_.ring_depth = 3;
    ring_.resize(cfg_.ring_depth);
    inited_ = true;
    // (the capacity test gives its fake engines PCI addresses of a fake sysfs tree: REVE_FAKE_BUS_ID_<device>)
    if (const char* e = std::getenv(("REVE_FAKE_BUS_ID_" + std::to_string(cfg_.device)).c_str())) bus_id_ = e;
    return 0;
}

static void nearest(const uint8_t* src, int w, int h, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds, int s)
{
    for (int y = 0; y < h * s; ++y)
        for (int x = 0; x < w * s; ++x) std::memcpy(dst + (size_t)y * ds + (size_t)x * 3, src + (size_t)(y / s) * ss + (size_t)(x / s) * 3, 3);
}

int Engine::upscale_host(const uint8_t* src, int w, int h, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds)
{
    if (!src || !dst || w <= 0 || h <= 0) return fail(REVE_E_INVALID, "bad frame arguments");
    // REVE_FAKE_ENGINE_NOOP=1: a GPU that takes no time and no host cycles (the capacity test of the host pipeline)
    static const bool noop = std::getenv("REVE_FAKE_ENGINE_NOOP") && std::getenv("REVE_FAKE_ENGINE_NOOP")[0] == '1';
    if (!noop) nearest(src, w, h, ss, dst, ds, cfg_.scale);
    stats_.frames_done++;
    return 0;
}

int Engine::submit(uint64_t id, const uint8_t* src, int w, int h, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds)
{
    if (ring_count_ == ring_.size()) return fail(REVE_E_BUSY, "ring full: call reve_wait first");
    if ((w != geo_w_ || h != geo_h_) && ring_count_) return fail(REVE_E_BUSY, "frame size changed with frames in flight");
    geo_w_ = w; geo_h_ = h;
    int rc = upscale_host(src, w, h, ss, dst, ds);
    if (rc) return rc;
    ring_[(ring_head_ + ring_count_) % ring_.size()].id = id;
    ring_count_++;
    return 0;
}

int Engine::wait(uint64_t* id)
{
    if (!ring_count_) return fail(REVE_E_BUSY, "nothing in flight");
    if (id) *id = ring_[ring_head_].id;
    ring_head_ = (ring_head_ + 1) % ring_.size();
    ring_count_--;
    return 0;
}

}  // namespace reve

namespace reve {
int Engine::get_stats(Stats& s) { s = stats_; return 0; }
int Engine::reset_stats() { stats_ = Stats(); return 0; }
}  // namespace reve
