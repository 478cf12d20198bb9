// CPU sanitizer builds only (make san): the host code that parses untrusted files (png.cpp, model.cpp) and runs the
// 70-thread directory pipeline (dirmode.cpp) is compiled unchanged with g++ -fsanitize=... and linked against THIS
// stand-in for engine.cpp: an "engine" whose submit() does a nearest-neighbour upscale on the calling thread and whose
// wait() hands frames back in submission order, plus malloc-backed stand-ins for the two HIP calls dirmode.cpp makes.
// It is not part of libreve_hip.so and computes nothing of the product path.
//
// Round 5: the same stand-in carries the C ABI (capi.cpp, unchanged) and the two executables (reve_cli.cpp,
// main_realesrgan.cpp, unchanged) in the `fake` builds of `make san`: BASELINE config 1 — "plumbing, no GPU" — runs as a CPU test
// with NO CPU compute path in the product; and reve_create_group over REVE_FAKE_DEVICES "GPUs" drives the weights broadcast
// against the recording RCCL table of san/fake_rccl.cpp.  A fake "device blob" is 4 KiB of host memory: the root's is filled
// from the model, the others' start as zeros, so a test can tell whether the broadcast reached them.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <string>

#include "../../../include/reve_hip.h"
#include "../engine.h"

extern "C" hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
extern "C" hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
// the rest of the HIP surface capi.cpp touches: REVE_FAKE_DEVICES devices (default 1) at PCI addresses 0000:<device>1:00.0
static int fake_device_count() { const char* e = std::getenv("REVE_FAKE_DEVICES"); return e ? std::atoi(e) : 1; }
extern "C" hipError_t hipGetDeviceCount(int* n) { *n = fake_device_count(); return *n > 0 ? hipSuccess : hipErrorNoDevice; }
extern "C" hipError_t hipGetLastError(void) { return hipSuccess; }
extern "C" hipError_t hipDeviceGetPCIBusId(char* out, int cap, int device)
{
    if (device < 0 || device >= fake_device_count()) return hipErrorInvalidDevice;
    std::snprintf(out, (size_t)cap, "0000:%x1:00.0", device);
    return hipSuccess;
}

namespace reve {

std::mutex& unsafe_calls_mutex()
{
    static std::mutex m;
    return m;
}

int Engine::fail(int code, const std::string& what) { err_ = what; return code; }

// Context accounting for the tests: how many fake engines are alive (reve_create_group must leave none behind when it fails)
static std::atomic<int> g_fake_engines{0};
extern "C" int reve_fake_live_engines(void) { return g_fake_engines.load(); }
Engine::~Engine()
{
    if (inited_) --g_fake_engines;
    std::free(d_weights_);
}

int Engine::init(const EngineConfig& cfg, const Model& model, bool upload_weights)
{
    cfg_ = cfg;
    if (cfg_.device < 0 || cfg_.device >= fake_device_count()) return fail(REVE_E_NODEVICE, "device ordinal out of range");
    if (model.scale && cfg_.scale != model.scale) return fail(REVE_E_MODEL, "model upscale factor does not match config.scale");
    if (const char* e = std::getenv("REVE_FAKE_INIT_FAILS_ON"); e && std::atoi(e) == cfg_.device) return fail(REVE_E_NOMEM, "injected: init failure");
    if (cfg_.ring_depth <= 0) cfg_.ring_depth = 3;
    ring_.resize(cfg_.ring_depth);
    weights_bytes_ = 4096;
    d_weights_ = std::calloc(1, weights_bytes_);
    if (!d_weights_) return fail(REVE_E_NOMEM, "calloc");
    if (upload_weights)
        for (size_t i = 0; i < weights_bytes_; ++i)
            ((uint8_t*)d_weights_)[i] = (uint8_t)(i * 131 + (model.w_last.empty() ? 7 : (int)(model.w_last[i % model.w_last.size()] * 1024)));
    inited_ = true;
    ++g_fake_engines;
    char bus[32];
    std::snprintf(bus, sizeof bus, "0000:%x1:00.0", cfg_.device);
    bus_id_ = bus;
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

// ---- what capi.cpp links beyond the pipeline's needs
int Engine::copy_weights_from(const Engine& src)
{
    if (src.weights_bytes_ != weights_bytes_) return fail(REVE_E_INVALID, "weight blobs do not match");
    std::memcpy(d_weights_, src.d_weights_, weights_bytes_);
    return 0;
}
int Engine::upscale_device(const void*, int, int, ptrdiff_t, void*, ptrdiff_t) { return fail(REVE_E_UNSUPPORTED, "fake engine: no device frames"); }
int Engine::upscale_device_batch(int, const void* const*, void* const*, int, int, ptrdiff_t, ptrdiff_t) { return fail(REVE_E_UNSUPPORTED, "fake engine: no device frames"); }
int Engine::sync() { return 0; }
int Engine::debug_run_layers(const uint8_t*, int, int, ptrdiff_t, int, float*, size_t) { return fail(REVE_E_UNSUPPORTED, "fake engine: no layers"); }
int Engine::set_option(const std::string& name, int) { return fail(REVE_E_INVALID, "unknown option " + name); }
int Engine::get_option(const std::string&, int*) const { return REVE_E_INVALID; }
void debug_blocked_order(int, int, uint32_t*) {}
int wino_ring_offset(int, int) { return 0; }
}  // namespace reve
