// extern "C" surface of libreve_hip.so (declared in include/reve_hip.h).
// Replaces the process boundary of reve-shared/src/lib.rs:129-155 (spawn realesrgan-ncnn-vulkan,
// read its stderr) with plain function calls; never throws, never aborts, every failure is a code.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <new>
#include <atomic>
#include <exception>
#include <functional>
#include <mutex>
#include <string>

#include "../../include/reve_hip.h"
#include "../../include/reve_hip_debug.h"
#include "dirmode.h"
#include "hostbind.h"
#include "engine.h"
#include "model.h"
#include "png.h"

namespace { std::atomic<int> g_live_contexts{0}; }
struct reve_ctx {
    reve::Engine engine;
    std::string last_error;
    reve_ctx() { ++g_live_contexts; }
    // the last context of a process takes the parked pinned buffers with it (a long-lived host must not keep gigabytes of
    // page-locked memory for an upscaler it no longer has)
    ~reve_ctx() { if (--g_live_contexts == 0) (void)reve::pinned_cache_trim(); }
};

namespace {
thread_local std::string g_create_error;

int done(reve_ctx* c, int rc)
{
    if (rc != 0) c->last_error = c->engine.err();
    return rc;
}
}  // namespace

// ---- the one collective of the path: the packed weights, uploaded to devices[0], reach the other GPUs of
// a group by ONE ncclBroadcast over xGMI (RCCL keeps NCCL's API names).  librccl is loaded on first use, so a single-GPU
// caller never needs it; a group of distinct GPUs that cannot load or initialise it FAILS (no silent other path).
// REVE_GROUP_BCAST=peer selects hipMemcpyPeer copies instead (what contexts sharing one device always use).
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;         // why the library is unusable ("" = loaded and complete); set once
    std::once_flag once;
    // thread-safe and sticky: two threads creating groups at once load it once, and a librccl that lacks a symbol stays an
    // error on every later call (it used to look "loaded" the second time and jump through a null pointer)
    std::string load()
    {
        std::call_once(once, [this] {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
            if (!lib) { const char* why = dlerror(); error = std::string("cannot load librccl: ") + (why ? why : "unknown error"); return; }
            auto sym = [&](const char* n) { return dlsym(lib, n); };
            CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
            CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
            GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
            GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
            Broadcast = (decltype(Broadcast))sym("ncclBroadcast");
            GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
            if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Broadcast || !GetErrorString) {
                error = "librccl lacks the NCCL entry points";
                dlclose(lib);
                lib = nullptr;
            }
        });
        return error;
    }
};

// one broadcast of engines[0]'s weights blob to engines[1..n) (in place on the root), every rank driven from this thread
std::string rccl_broadcast_weights(const std::vector<reve::Engine*>& engs)
{
    static Rccl r;
    std::string e = r.load();
    if (!e.empty()) return e;
    const int n = (int)engs.size();
    std::vector<int> devs;
    for (auto* g : engs) devs.push_back(g->device());
    std::vector<void*> comms(n, nullptr);
    int rc = r.CommInitAll(comms.data(), n, devs.data());
    if (rc != 0) return std::string("ncclCommInitAll: ") + r.GetErrorString(rc);
    std::vector<hipStream_t> streams(n, nullptr);
    for (int i = 0; i < n && e.empty(); ++i)
        if (hipSetDevice(devs[i]) != hipSuccess || hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking) != hipSuccess) e = "hipStreamCreate for the broadcast failed";
    if (e.empty()) {
        constexpr int kUint8 = 1;   // ncclUint8
        rc = r.GroupStart();
        for (int i = 0; i < n && rc == 0; ++i)
            rc = r.Broadcast(engs[0]->weights_ptr(), engs[i]->weights_ptr(), engs[0]->weights_bytes(), kUint8, 0, comms[i], streams[i]);
        const int rc2 = r.GroupEnd();
        if (rc == 0) rc = rc2;
        if (rc != 0) e = std::string("ncclBroadcast: ") + r.GetErrorString(rc);
    }
    for (int i = 0; i < n; ++i)
        if (streams[i]) {
            (void)hipSetDevice(devs[i]);
            if (hipStreamSynchronize(streams[i]) != hipSuccess && e.empty()) e = "broadcast stream failed";
            (void)hipStreamDestroy(streams[i]);
        }
    for (void* c : comms)
        if (c) (void)r.CommDestroy(c);
    return e;
}
}  // namespace

extern "C" {

int reve_abi_version(void) { return REVE_ABI_VERSION; }

const char* reve_strerror(int code)
{
    switch (code) {
    case REVE_OK: return "success";
    case REVE_E_INVALID: return "invalid argument";
    case REVE_E_MODEL: return "model files missing, malformed or unsupported";
    case REVE_E_NODEVICE: return "no usable gfx950 HIP device (no CPU fallback)";
    case REVE_E_HIP: return "HIP runtime error";
    case REVE_E_NOMEM: return "out of memory";
    case REVE_E_IO: return "I/O error";
    case REVE_E_BUSY: return "ring full or empty";
    case REVE_E_UNSUPPORTED: return "unsupported request";
    default: return "unknown error";
    }
}

int reve_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int reve_resolve_model_name(const char* model_name, int scale, char* out, size_t cap)
{
    if (!out || cap == 0 || scale < 2 || scale > 4) return REVE_E_INVALID;
    static const char base[] = "realesr-animevideov3";
    std::string name = model_name && model_name[0] ? model_name : base;
    int swapped = 0;
    if (name == base) {
        name += "-x" + std::to_string(scale);           // like the binary
    } else if (name.size() == sizeof base - 1 + 3 && name.compare(0, sizeof base - 1, base) == 0 &&
               name[sizeof base - 1] == '-' && name[sizeof base] == 'x' && name.back() >= '2' && name.back() <= '4' &&
               name.back() != (char)('0' + scale)) {
        name.back() = (char)('0' + scale);              // reve-cli's always-x2 name (lib.rs:141) with -s 3 / 4
        swapped = 1;
    }
    if (name.size() + 1 > cap) return REVE_E_INVALID;
    std::memcpy(out, name.c_str(), name.size() + 1);
    return swapped;
}

static int load_model(const reve_config* cfg, reve::Model& model)
{
    std::string e;
    if (cfg->param_data && cfg->bin_data) {
        e = reve::parse_ncnn(std::string((const char*)cfg->param_data, cfg->param_len),
                             (const uint8_t*)cfg->bin_data, cfg->bin_len, model);
    } else {
        char name[512];
        if (reve_resolve_model_name(cfg->model_name, cfg->scale, name, sizeof name) < 0) {
            g_create_error = "model name too long";
            return REVE_E_INVALID;
        }
        e = reve::load_ncnn_files(cfg->model_dir ? cfg->model_dir : "models", name, model);
    }
    if (!e.empty()) {
        g_create_error = e;
        return REVE_E_MODEL;
    }
    return REVE_OK;
}

int reve_create_group(const reve_config* cfg, const int* devices, int n, reve_ctx** out)
{
    if (!cfg || !out || !devices || n <= 0 || n > 64 || cfg->struct_size < sizeof(reve_config)) return REVE_E_INVALID;
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    if (cfg->scale < 2 || cfg->scale > 4) return REVE_E_INVALID;
    reve::Model model;
    int rc = load_model(cfg, model);
    if (rc != REVE_OK) return rc;
    reve::EngineConfig ec;
    ec.scale = cfg->scale; ec.tile = cfg->tile;
    ec.prepad = cfg->prepad; ec.ring_depth = cfg->ring_depth;
    for (int i = 0; i < n && rc == REVE_OK; ++i) {
        reve_ctx* c = new (std::nothrow) reve_ctx();
        if (!c) { rc = REVE_E_NOMEM; break; }
        ec.device = devices[i];
        rc = c->engine.init(ec, model, /*upload_weights=*/i == 0);   // the others receive devices[0]'s copy below
        if (rc != 0) {
            g_create_error = c->engine.err();
            delete c;
        } else {
            out[i] = c;
        }
    }
    if (rc == REVE_OK) {
        // transport of the weights to out[1..n): RCCL broadcast when the devices are distinct (or when asked for: a
        // one-device "group" then exercises the same code), device-to-device copies when contexts share a device
        bool distinct = true;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j) distinct &= devices[i] != devices[j];
        const char* env = std::getenv("REVE_GROUP_BCAST");
        const bool want_rccl = env ? std::strcmp(env, "rccl") == 0 : n > 1;
        if (env && std::strcmp(env, "rccl") == 0 && !distinct) {
            // RCCL has one rank per device: "forcing" it for contexts that share a device cannot be honoured, and doing the
            // peer copy instead would be the silent other path the header rules out
            g_create_error = "REVE_GROUP_BCAST=rccl needs distinct devices (contexts that share a GPU are filled by a device-to-device copy)";
            rc = REVE_E_INVALID;
        } else if (want_rccl && distinct) {
            std::vector<reve::Engine*> engs;
            for (int i = 0; i < n; ++i) engs.push_back(&out[i]->engine);
            const std::string e = rccl_broadcast_weights(engs);
            if (!e.empty()) { g_create_error = "weights broadcast: " + e; rc = REVE_E_HIP; }
        } else {
            for (int i = 1; i < n && rc == REVE_OK; ++i)
                if ((rc = out[i]->engine.copy_weights_from(out[0]->engine)) != 0) g_create_error = out[i]->engine.err();
        }
    }
    if (rc != REVE_OK)
        for (int i = 0; i < n; ++i) { delete out[i]; out[i] = nullptr; }
    return rc;
}

int reve_create(const reve_config* cfg, reve_ctx** out)
{
    if (!cfg || !out) return REVE_E_INVALID;
    const int dev = cfg->device;
    return reve_create_group(cfg, &dev, 1, out);
}

void reve_destroy(reve_ctx* ctx) { delete ctx; }

const char* reve_last_error(reve_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_create_error.c_str(); }

int reve_upscale_rgb8(reve_ctx* c, const uint8_t* src, int w, int h, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds)
{
    if (!c) return REVE_E_INVALID;
    return done(c, c->engine.upscale_host(src, w, h, ss, dst, ds));
}

int reve_upscale_rgb8_device(reve_ctx* c, const void* d_src, int w, int h, ptrdiff_t ss, void* d_dst, ptrdiff_t ds)
{
    if (!c) return REVE_E_INVALID;
    return done(c, c->engine.upscale_device(d_src, w, h, ss, d_dst, ds));
}

int reve_upscale_rgb8_device_batch(reve_ctx* c, int n, const void* const* d_srcs, void* const* d_dsts, int w, int h, ptrdiff_t ss, ptrdiff_t ds)
{
    if (!c) return REVE_E_INVALID;
    return done(c, c->engine.upscale_device_batch(n, d_srcs, d_dsts, w, h, ss, ds));
}

int reve_sync(reve_ctx* c) { return c ? done(c, c->engine.sync()) : REVE_E_INVALID; }

int reve_submit(reve_ctx* c, uint64_t id, const uint8_t* src, int w, int h, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds)
{
    if (!c) return REVE_E_INVALID;
    return done(c, c->engine.submit(id, src, w, h, ss, dst, ds));
}

int reve_wait(reve_ctx* c, uint64_t* id) { return c ? done(c, c->engine.wait(id)) : REVE_E_INVALID; }

void* reve_alloc_pinned(size_t bytes)
{
    void* p = nullptr;
    std::lock_guard<std::mutex> lk(reve::unsafe_calls_mutex());       // (not while another thread of the process captures a graph)
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void reve_free_pinned(void* p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(reve::unsafe_calls_mutex());
    (void)hipHostFree(p);
}

int reve_upscale_dir(reve_ctx* c, const char* in_dir, const char* out_dir, reve_progress_cb cb, void* user)
{
    if (!c || !in_dir || !out_dir) return REVE_E_INVALID;
    std::string err;
    std::vector<reve::Engine*> one{&c->engine};
    int rc = reve::upscale_dir(one, in_dir, out_dir, cb, user, err);
    if (rc != 0) c->last_error = err;
    return rc;
}

int reve_upscale_dir_multi(reve_ctx* const* ctxs, int n, const char* in_dir, const char* out_dir, reve_progress_cb cb, void* user)
{
    if (!ctxs || n <= 0 || n > 64 || !in_dir || !out_dir) return REVE_E_INVALID;
    std::vector<reve::Engine*> engs;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || ctxs[i]->engine.scale() != ctxs[0]->engine.scale()) return REVE_E_INVALID;
        engs.push_back(&ctxs[i]->engine);
    }
    std::string err;
    int rc = reve::upscale_dir(engs, in_dir, out_dir, cb, user, err);
    if (rc != 0) ctxs[0]->last_error = err;
    return rc;
}

int reve_upscale_stream_multi(reve_ctx* const* ctxs, int n, int n_frames, int w, int h, reve_read_frame_cb read,
                              reve_write_frame_cb write, reve_progress_cb done_cb, void* user)
{
    if (!ctxs || n <= 0 || n > 64 || n_frames < 0 || w <= 0 || h <= 0 || !read || !write) return REVE_E_INVALID;
    std::vector<reve::Engine*> engs;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || ctxs[i]->engine.scale() != ctxs[0]->engine.scale()) return REVE_E_INVALID;
        engs.push_back(&ctxs[i]->engine);
    }
    reve::FrameIO io;
    io.decode = [&](int i, const std::function<uint8_t*(int, int)>& sink) -> std::string {
        return read(user, i, sink(w, h)) == 0 ? "" : "frame " + std::to_string(i) + ": the frame source failed";
    };
    io.encode = [&](int i, const uint8_t* rgb, int, int) -> std::string {
        return write(user, i, rgb) == 0 ? "" : "frame " + std::to_string(i) + ": the frame sink failed";
    };
    std::string err;
    int rc = REVE_E_NOMEM;
    try {
        rc = reve::run_pipeline(engs, n_frames, io, [&](int i) { if (done_cb) done_cb(user, i, nullptr, nullptr); }, err);
    } catch (const std::exception& e) {
        err = e.what();
    }
    if (rc != 0) ctxs[0]->last_error = err;
    return rc;
}

int reve_device_cpulist(int device, char* out, size_t cap)
{
    if (!out || cap == 0) return REVE_E_INVALID;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return REVE_E_NODEVICE; }
    const std::string l = reve::pci_local_cpulist(bus);
    if (l.size() + 1 > cap) return REVE_E_INVALID;
    std::memcpy(out, l.c_str(), l.size() + 1);
    return REVE_OK;
}

int reve_bind_thread_to_device(int device)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return REVE_E_NODEVICE; }
    return reve::bind_this_thread(reve::pci_local_cpulist(bus));
}

size_t reve_trim(void) { return reve::pinned_cache_trim(); }

int reve_upscale_file(reve_ctx* c, const char* in_path, const char* out_path)
{
    if (!c || !in_path || !out_path) return REVE_E_INVALID;
    std::string err;
    int rc = reve::upscale_file(c->engine, in_path, out_path, err);
    if (rc != 0) c->last_error = err;
    return rc;
}

int reve_png_read(const char* path, uint8_t** rgb, int* w, int* h)
{
    if (!path || !rgb || !w || !h) return REVE_E_INVALID;
    std::vector<uint8_t> file, px;
    std::string e = reve::read_file(path, file);
    if (e.empty()) e = reve::png_decode_rgb8(file, px, *w, *h);
    if (!e.empty()) { g_create_error = e; return REVE_E_IO; }
    *rgb = (uint8_t*)std::malloc(px.size());
    if (!*rgb) return REVE_E_NOMEM;
    std::memcpy(*rgb, px.data(), px.size());
    return REVE_OK;
}

int reve_png_write(const char* path, const uint8_t* rgb, int w, int h, ptrdiff_t stride)
{
    if (!path || !rgb || w <= 0 || h <= 0 || stride < (ptrdiff_t)w * 3) return REVE_E_INVALID;
    std::vector<uint8_t> file;
    std::string e = reve::png_encode_rgb8(rgb, w, h, (size_t)stride, 1, file);
    if (e.empty()) e = reve::write_file(path, file);
    if (!e.empty()) { g_create_error = e; return REVE_E_IO; }
    return REVE_OK;
}

void reve_free(void* p) { std::free(p); }

int reve_set_profiling(reve_ctx* c, int enabled)
{
    if (!c) return REVE_E_INVALID;
    c->engine.set_profiling(enabled != 0);
    return REVE_OK;
}

int reve_set_option(reve_ctx* c, const char* name, int value)
{
    if (!c || !name) return REVE_E_INVALID;
    return done(c, c->engine.set_option(name, value));
}

int reve_get_option(reve_ctx* c, const char* name, int* value)
{
    if (!c || !name || !value) return REVE_E_INVALID;
    return c->engine.get_option(name, value);
}

int reve_get_stats(reve_ctx* c, reve_stats* out)
{
    // a caller built against ABI 2 passes the shorter struct: it gets the fields it knows
    if (!c || !out || out->struct_size < offsetof(reve_stats, frames_timed)) return REVE_E_INVALID;
    reve::Stats s;
    c->engine.get_stats(s);
    reve_stats full;
    std::memset(&full, 0, sizeof full);
    full.frames_done = s.frames_done; full.body_launches = s.body_launches;
    full.body_ms_total = s.body_ms_total; full.frame_ms_last = s.frame_ms_last;
    full.h2d_bytes = s.h2d_bytes; full.d2h_bytes = s.d2h_bytes;
    full.compute_units = s.compute_units; full.frame_w = s.frame_w; full.frame_h = s.frame_h;
    full.planes = s.planes; full.tiles_per_plane = s.tiles_per_plane;
    full.body_layers_per_launch = s.body_layers_per_launch;
    full.frames_timed = s.frames_timed; full.first_ms_total = s.first_ms_total;
    full.last_ms_total = s.last_ms_total; full.frame_ms_total = s.frame_ms_total;
    full.ring_frames = s.ring_frames; full.h2d_ms_total = s.h2d_ms_total; full.chain_ms_total = s.chain_ms_total;
    full.d2h_ms_total = s.d2h_ms_total; full.ring_wall_ms = s.ring_wall_ms;
    const uint32_t n = out->struct_size < sizeof full ? out->struct_size : (uint32_t)sizeof full;
    full.struct_size = out->struct_size;
    std::memcpy(out, &full, n);
    return REVE_OK;
}

int reve_reset_stats(reve_ctx* c) { return c ? c->engine.reset_stats() : REVE_E_INVALID; }

int reve_debug_blocked_order(int tiles_x, int tiles_y, uint32_t* out)
{
    if (tiles_x <= 0 || tiles_y <= 0 || tiles_x >= 1024 || tiles_y >= 1024 || !out) return REVE_E_INVALID;
    reve::debug_blocked_order(tiles_x, tiles_y, out);
    return REVE_OK;
}

int reve_debug_geometry(int w, int h, int tile, int prepad, long long* out5)
{
    if (!out5) return REVE_E_INVALID;
    return reve::frame_geometry(w, h, tile, prepad, out5);
}

int reve_debug_model_conditioning(const void* param, size_t plen, const void* bin, size_t blen, double* kappa, double* limit)
{
    if (!param || !bin || !kappa) return REVE_E_INVALID;
    reve::Model model;
    const std::string e = reve::parse_ncnn(std::string((const char*)param, plen), (const uint8_t*)bin, blen, model);
    if (!e.empty()) { g_create_error = e; return REVE_E_MODEL; }
    *kappa = reve::conditioning_kappa(model);
    if (limit) *limit = reve::WINOGRAD_KAPPA_LIMIT;
    return REVE_OK;
}

int reve_debug_frames_per_launch(int w, int h, int compute_units) { return reve::frames_per_launch(w, h, compute_units); }

int reve_debug_wino_ring_offset(int column, int chunk)
{
    if (column < 0 || column > 65 || chunk < 0 || chunk > 7) return REVE_E_INVALID;
    return reve::wino_ring_offset(column, chunk);
}

int reve_debug_run_layers(reve_ctx* c, const uint8_t* src, int w, int h, ptrdiff_t ss, int layer, float* out, size_t n)
{
    if (!c) return REVE_E_INVALID;
    return done(c, c->engine.debug_run_layers(src, w, h, ss, layer, out, n));
}

}  // extern "C"
