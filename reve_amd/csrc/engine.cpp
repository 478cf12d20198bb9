#include "engine.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/reve_hip.h"
#include "trace.h"

namespace reve {

#define HIPCHK(call, what)                                     \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return hipfail((int)e_, what);   \
    } while (0)

// Runtime calls that would invalidate another thread's stream capture (allocations, frees, synchronous copies and memsets) and
// the captures themselves take turns on this mutex — within the library; see Engine::submit.
std::mutex& unsafe_calls_mutex()
{
    static std::mutex* m = new std::mutex;     // (never destroyed: contexts may be torn down during process exit)
    return *m;
}

int Engine::fail(int code, const std::string& what)
{
    err_ = what;
    return code;
}

int Engine::hipfail(int e, const char* what)
{
    err_ = std::string(what) + ": " + hipGetErrorString((hipError_t)e);
    return (e == (int)hipErrorOutOfMemory) ? REVE_E_NOMEM : REVE_E_HIP;
}

Engine::~Engine()
{
    if (!inited_) return;
    (void)hipSetDevice(cfg_.device);
    (void)hipDeviceSynchronize();
    release_geometry();
    if (d_weights_) (void)hipFree(d_weights_);
    drop_graphs();
    auto free_slot = [](Slot& s) {
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.ev_h2d) (void)hipEventDestroy((hipEvent_t)s.ev_h2d);
        if (s.ev_comp) (void)hipEventDestroy((hipEvent_t)s.ev_comp);
        if (s.ev_d2h) (void)hipEventDestroy((hipEvent_t)s.ev_d2h);
        if (s.ev_h2d0) (void)hipEventDestroy((hipEvent_t)s.ev_h2d0);
        if (s.ev_comp0) (void)hipEventDestroy((hipEvent_t)s.ev_comp0);
        if (s.ev_d2h0) (void)hipEventDestroy((hipEvent_t)s.ev_d2h0);
    };
    free_slot(sync_slot_);
    for (auto& s : ring_) free_slot(s);
    for (auto& e : evpool_) {
        (void)hipEventDestroy((hipEvent_t)e.b0); (void)hipEventDestroy((hipEvent_t)e.b1);
        (void)hipEventDestroy((hipEvent_t)e.f0); (void)hipEventDestroy((hipEvent_t)e.f1);
    }
    if (stream_) (void)hipStreamDestroy((hipStream_t)stream_);
    if (s_h2d_) (void)hipStreamDestroy((hipStream_t)s_h2d_);
    if (s_d2h_) (void)hipStreamDestroy((hipStream_t)s_d2h_);
}

// The "broadcast" of a multi-GPU group when RCCL is not the transport (two contexts on ONE device, or
// REVE_GROUP_BCAST=peer): the first engine's blob goes GPU to GPU.
int Engine::copy_weights_from(const Engine& src)
{
    if (!inited_ || !src.inited_ || src.weights_bytes_ != weights_bytes_) return fail(REVE_E_INVALID, "weight blobs do not match");
    HIPCHK(hipSetDevice(cfg_.device), "hipSetDevice");
    HIPCHK(hipMemcpyPeer(d_weights_, cfg_.device, src.d_weights_, src.cfg_.device, weights_bytes_), "peer copy of the weights");
    return 0;
}

int Engine::init(const EngineConfig& cfg, const Model& model, bool upload_weights)
{
    cfg_ = cfg;
    if (cfg_.scale != model.scale) return fail(REVE_E_MODEL, "model upscale factor does not match config.scale");
    if (cfg_.tile < 0) return fail(REVE_E_INVALID, "tile must be >= 0");
    if (cfg_.tile > 0 && cfg_.tile < 32) return fail(REVE_E_INVALID, "tile must be 0 or >= 32");
    if (cfg_.prepad <= 0) cfg_.prepad = 10;
    ring_auto_ = cfg_.ring_depth <= 0;          // the library chooses: 3, or two batches where frames share launches
    if (cfg_.ring_depth <= 0) cfg_.ring_depth = 3;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(REVE_E_NODEVICE, "no HIP device visible (libreve_hip has no CPU fallback)");
    if (cfg_.device < 0 || cfg_.device >= ndev) return fail(REVE_E_NODEVICE, "device ordinal out of range");
    HIPCHK(hipSetDevice(cfg_.device), "hipSetDevice");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg_.device), "hipGetDeviceProperties");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(REVE_E_NODEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    n_cu_ = prop.multiProcessorCount;
    {
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), cfg_.device) == hipSuccess) bus_id_ = bus;
    }
    if (int e = prepare_body_kernels())
        return hipfail(e, "hipFuncSetAttribute(dynamic LDS size)");
    if (int e = prepare_pair_kernels())
        return hipfail(e, "hipFuncSetAttribute(dynamic LDS size, fused pair)");
    if (int e = prepare_wino_kernels())
        return hipfail(e, "hipFuncSetAttribute(dynamic LDS size, Winograd pair)");
    if (int e = prepare_last_strip_kernels())
        return hipfail(e, "hipFuncSetAttribute(dynamic LDS size, conv_last strips)");
    // Environment: REVE_WINOGRAD is the one switch a deployment may want without rebuilding its caller; the
    // launch-structure switches below are lab controls (A/B scripts, bisecting a suspected kernel) and are read only when
    // REVE_LAB=1 says the process is such a session — a production host's stray environment cannot change the launch structure.
    kappa_ = conditioning_kappa(model);
    if (const char* e = std::getenv("REVE_WINOGRAD"); e && e[0]) {       // 0 | 1 | auto (the default); also off / on / direct / winograd
        const std::string v = e;
        if (v == "0" || v == "off" || v == "direct") winograd_mode_ = 0;
        else if (v == "1" || v == "on" || v == "winograd") winograd_mode_ = 1;
        else if (v == "2" || v == "auto") winograd_mode_ = 2;
        else {
            static std::once_flag warned;        // (a value nobody defined must not silently pin an evaluation; no "d-o-n-e" in the text)
            std::call_once(warned, [&] { std::fprintf(stderr, "libreve_hip: REVE_WINOGRAD=%s is not 0, 1 or auto: ignored, the evaluation stays auto\n", e); });
        }
    }
    apply_winograd_mode(true);
    if (const char* lab = std::getenv("REVE_LAB"); lab && lab[0] == '1') {
        if (const char* e = std::getenv("REVE_FUSE_PAIRS")) fuse_pairs_ = e[0] == '1';
        if (const char* e = std::getenv("REVE_STRIP_LAST")) strip_last_ = e[0] == '1';
        if (const char* e = std::getenv("REVE_BATCH")) batching_ = e[0] == '1';
        if (const char* e = std::getenv("REVE_GRAPH")) use_graph_ = e[0] == '1';
        if (const char* e = std::getenv("REVE_NO_BLOCKED_ORDER")) blocked_env_ = e[0] != '1';
    }
    stats_.compute_units = n_cu_;
    inited_ = true;
    hipStream_t s;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate"); stream_ = s;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate"); s_h2d_ = s;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate"); s_d2h_ = s;

    // ---- packed weights: one blob, every layer's fragments / bias / slopes at 256-byte aligned offsets (the same on
    // every device of a group, so a group needs ONE broadcast of it)
    n_body_ = model.n_body;
    body_.resize(n_body_);
    std::vector<PackedLayer> packed;
    packed.push_back(pack_first(model));
    body_unit_slopes_.assign(n_body_, 1);
    for (int l = 0; l < n_body_; ++l) {
        packed.push_back(pack_body(model, l));
        for (uint16_t hs : packed.back().slope) {
            const float sl = f16_to_f32(hs);
            if (!(sl >= 0.0f && sl <= 1.0f)) body_unit_slopes_[l] = 0;    // NaN fails too: the general PReLU form then
        }
    }
    packed.push_back(pack_last(model, true));
    // (behind everything else: the Winograd-domain fragments of the body layers)
    for (int l = 0; l < n_body_; ++l) {
        PackedLayer f = pack_body_wino(model, l);
        f.bias.clear(); f.slope.clear();
        packed.push_back(std::move(f));
    }
    std::vector<uint8_t> host;
    struct Off { size_t w, b, s; };
    std::vector<Off> offs;
    auto put = [&](const std::vector<uint16_t>& v, size_t min_elems) {
        const size_t at = (host.size() + 255) & ~(size_t)255;
        host.resize(at + std::max(v.size(), min_elems) * 2, 0);
        std::memcpy(host.data() + at, v.data(), v.size() * 2);
        return at;
    };
    for (const PackedLayer& p : packed) {
        Off o;
        o.w = put(p.wpack, 0);
        o.b = p.bias.empty() ? (size_t)-1 : put(p.bias, 64);          // the kernels read up to 64 bias entries (zero padded)
        o.s = p.slope.empty() ? (size_t)-1 : put(p.slope, 0);
        offs.push_back(o);
    }
    weights_bytes_ = (host.size() + 255) & ~(size_t)255;
    host.resize(weights_bytes_, 0);
    HIPCHK(hipMalloc(&d_weights_, weights_bytes_), "hipMalloc(weights)");
    if (upload_weights) HIPCHK(hipMemcpy(d_weights_, host.data(), weights_bytes_, hipMemcpyHostToDevice), "upload weights");
    auto at = [&](const Off& o) {
        DevLayer d;
        d.wpack = (char*)d_weights_ + o.w;
        d.bias = (uint16_t*)((char*)d_weights_ + o.b);
        d.slope = o.s == (size_t)-1 ? nullptr : (uint16_t*)((char*)d_weights_ + o.s);
        return d;
    };
    first_ = at(offs[0]);
    for (int l = 0; l < n_body_; ++l) body_[l] = at(offs[1 + l]);
    last_ = at(offs[1 + n_body_]);
    body_wino_.resize(n_body_);
    for (int l = 0; l < n_body_; ++l) body_wino_[l] = (char*)d_weights_ + offs[2 + n_body_ + l].w;

    ring_.resize(std::max(cfg_.ring_depth, 2 * MAX_BATCH));      // (slots are filled on demand; how many are used: ring_cap())
    evpool_.resize(64);
    for (auto& e : evpool_) {
        hipEvent_t ev[4];
        for (auto& x : ev) HIPCHK(hipEventCreate(&x), "hipEventCreate");
        e = {ev[0], ev[1], ev[2], ev[3], false, 1};
    }
    return 0;
}

void Engine::drop_graphs()
{
    for (Slot& s : ring_)
        if (s.graph_exec) { (void)hipGraphExecDestroy((hipGraphExec_t)s.graph_exec); s.graph_exec = nullptr; }
}

void Engine::release_geometry()
{
    drop_graphs();
    if (arena_[0]) (void)hipFree(arena_[0]);
    if (arena_[1]) (void)hipFree(arena_[1]);
    if (d_planes_) (void)hipFree(d_planes_);
    if (d_items_) (void)hipFree(d_items_);
    if (d_col_ok_) (void)hipFree(d_col_ok_);
    if (d_last_units_) (void)hipFree(d_last_units_);
    d_last_units_ = nullptr;
    n_last_units_ = 0;
    d_col_ok_ = nullptr;
    d_items_ = nullptr;
    arena_[0] = arena_[1] = nullptr;
    d_planes_ = nullptr;
    geo_w_ = geo_h_ = 0;
}

// Lay the frame out as planes: one for the whole frame, or one per ncnn-compat tile (the binary's
// tiling, SURVEY.md §2.3.1 S2: ceil(w/T) x ceil(h/T) tiles, each with a `prepad` apron).
int Engine::configure(int w, int h, bool whole_frame_only)
{
    const int tile = whole_frame_only ? 0 : cfg_.tile;
    if (w == geo_w_ && h == geo_h_ && tile == geo_tile_ && batching_ == geo_batching_) return 0;
    const int batch = (tile == 0 && batching_) ? frames_per_launch(w, h, n_cu_) : 1;
    {
        long long geo[5];
        if (frame_geometry(w, h, tile, cfg_.prepad, geo) == REVE_E_UNSUPPORTED)
            return fail(REVE_E_UNSUPPORTED, tile ? "planes too large for 32-bit offsets on this canvas (use a smaller tile)" : "frame too large for one plane (use tile > 0)");
    }
    std::lock_guard<std::mutex> unsafe_lk(unsafe_calls_mutex());
    HIPCHK(hipStreamSynchronize((hipStream_t)stream_), "sync before re-configure");
    release_geometry();
    std::vector<PlaneDesc> planes;
    int maxw = 0, maxh = 0;
    std::vector<int> col_x, row_y;            // canvas position of each plane column's / row's border pixel (several planes)
    int xt = 1, yt = 1;
    if (tile == 0) {
        for (int f = 0; f < batch; ++f) planes.push_back({w, h, 0, 0, 0ull, 0u, 0u});      // (batch > 1: a column of planes, one per frame)
        maxw = w; maxh = h; pad_ = 0;
        yt = batch;
    } else {
        pad_ = cfg_.prepad;
        xt = (w + tile - 1) / tile; yt = (h + tile - 1) / tile;
        for (int yi = 0; yi < yt; ++yi)
            for (int xi = 0; xi < xt; ++xi) {
                const int x0 = xi * tile - pad_, x1 = std::min((xi + 1) * tile, w) + pad_;
                const int y0 = yi * tile - pad_, y1 = std::min((yi + 1) * tile, h) + pad_;
                planes.push_back({x1 - x0, y1 - y0, x0, y0, 0ull, 0u, 0u});
                maxw = std::max(maxw, x1 - x0); maxh = std::max(maxh, y1 - y0);
            }
    }
    n_planes_ = (int)planes.size();
    const int tw = TILE_W, th = TILE_H;       // 16 x 32 tiles, 1-pixel zero border
    tiles_x_ = (maxw + tw - 1) / tw;
    tiles_y_ = (maxh + th - 1) / th;
    size_t arena_bytes;
    if (n_planes_ == 1) {
        Wp_ = tiles_x_ * tw + 2;
        Hp_ = tiles_y_ * th + 2;
        plane_stride_ = (size_t)Hp_ * Wp_ * PIX_BYTES;
        if (plane_stride_ >= ((size_t)1 << 31))
            return fail(REVE_E_UNSUPPORTED, "frame too large for one plane (use tile > 0)");
        arena_bytes = plane_stride_;
        planes[0].span = (unsigned)plane_stride_;
        planes[0].reserved = (unsigned)Hp_;          // arena rows from the plane's border row to the arena's end (k_last_strip's CANVAS instantiation clamps its reads there)
    } else {
        // Several planes (the binary's tiles with their aprons) lie on ONE canvas, a grid in which neighbours share their
        // 1-pixel zero border: plane column xi starts (border pixel) at canvas column col_x[xi].  The tile kernels address a plane
        // through its base offset and the canvas pitch and never write outside a plane's w x h pixels, so the shared borders
        // ("gutters") stay zero — and the pair kernel can take the canvas for one frame (col_ok / gut in PairArgs).  A tile
        // that hangs over its plane's edge reads into the neighbour (or wraps into the next canvas row): those inputs feed
        // only outputs the kernels drop.
        col_x.assign(xt + 1, 0); row_y.assign(yt + 1, 0);
        for (int xi = 0; xi < xt; ++xi) col_x[xi + 1] = col_x[xi] + planes[xi].w + 1;
        for (int yi = 0; yi < yt; ++yi) row_y[yi + 1] = row_y[yi] + planes[(size_t)yi * xt].h + 1;
        Wp_ = col_x[xt] + 1;
        Hp_ = row_y[yt] + 1;
        plane_stride_ = (size_t)Hp_ * Wp_ * PIX_BYTES;       // (the canvas)
        arena_bytes = plane_stride_ + (size_t)(th + 2) * Wp_ * PIX_BYTES;      // + the rows the last tiles hang over
        for (int yi = 0; yi < yt; ++yi)
            for (int xi = 0; xi < xt; ++xi) {
                PlaneDesc& p = planes[(size_t)yi * xt + xi];
                p.base = ((unsigned long long)row_y[yi] * Wp_ + col_x[xi]) * PIX_BYTES;
                p.span = (unsigned)std::min<unsigned long long>(arena_bytes - p.base, 0x7fffffffull);
                p.reserved = (unsigned)(Hp_ + th + 2 - row_y[yi]);
            }
    }
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipMalloc((void**)&arena_[i], arena_bytes), "hipMalloc(activation arena)");
        // the border and everything outside the image must stay zero for the arena's whole life
        HIPCHK(hipMemsetAsync(arena_[i], 0, arena_bytes, (hipStream_t)stream_), "memset arena");
    }
    HIPCHK(hipMalloc((void**)&d_planes_, sizeof(PlaneDesc) * n_planes_), "hipMalloc(planes)");
    HIPCHK(hipMemcpy(d_planes_, planes.data(), sizeof(PlaneDesc) * n_planes_, hipMemcpyHostToDevice), "upload planes");
    n_items_ = n_planes_ * tiles_x_ * tiles_y_;
    items_per_plane_ = ((w + tw - 1) / tw) * ((h + th - 1) / th);
    batch_ = batch;
    blocked_order_ = false;
    const bool blocked_env = blocked_env_;
    if (n_planes_ == 1 && blocked_env) {
        blocked_order_ = true;   // one plane: the kernels compute the 4x8-blocked order themselves (decode_blocked)
    } else if (n_planes_ < 4096 && tiles_x_ < 1024 && tiles_y_ < 1024) {
        // Work list for the persistent kernels (32 consecutive items run together on one XCD):
        //  * only the non-empty tiles of planes smaller than their slot (edge tiles of the frame);
        //  * in 4-wide x 8-tall blocks, so that a tile's vertical AND horizontal halo neighbours are
        //    in flight on the same XCD at the same time and the halo re-reads hit that XCD's L2.
        const bool blocked = blocked_env;
        const int bw = blocked ? 4 : 1024, bh = blocked ? 8 : 1;
        std::vector<uint32_t> items;
        for (int p = 0; p < n_planes_; ++p) {
            const int ptx = (planes[p].w + tw - 1) / tw, pty = (planes[p].h + th - 1) / th;
            for (int by = 0; by < pty; by += bh)
                for (int bx = 0; bx < ptx; bx += bw)
                    for (int ty = by; ty < std::min(by + bh, pty); ++ty)
                        for (int tx = bx; tx < std::min(bx + bw, ptx); ++tx)
                            items.push_back((uint32_t)tx | ((uint32_t)ty << 10) | ((uint32_t)p << 20));
        }
        n_items_ = (int)items.size();
        HIPCHK(hipMalloc((void**)&d_items_, items.size() * 4), "hipMalloc(items)");
        HIPCHK(hipMemcpy(d_items_, items.data(), items.size() * 4, hipMemcpyHostToDevice), "upload items");
    }
    // fused-pair kernel (whole frame only): strips of PAIR_VALID columns x segments of rows, as many units as CUs if the frame
    // allows it (1080p: 32 x 8 = 256); segments are an even number of rows (the kernel steps two rows at a time), >= 16
    pair_strips_ = pair_segs_ = pair_seg_h_ = 0;
    pair_gut_first_ = pair_gut_period_ = 0;
    const bool canvas = n_planes_ > 1;
    if (!canvas || plane_stride_ < ((size_t)1 << 31)) {          // (the pair kernel's offsets are 32-bit: a canvas of 2 GiB or more keeps one layer per launch)
        // (one plane is the whole frame, or — a frame smaller than the ncnn-compat tile — the frame with its apron: the kernel
        // works on the PLANE, whatever it stands for.  Several planes: on the canvas as one frame whose gutters stay zero)
        pair_w_ = canvas ? Wp_ - 2 : planes[0].w; pair_h_ = canvas ? Hp_ - 2 : planes[0].h;
        if (canvas) {
            if (xt > 1) {        // (a column of frames has gutter rows only)
                std::vector<unsigned char> ok((size_t)pair_w_, 1);
                for (int xi = 1; xi < xt; ++xi) ok[(size_t)col_x[xi] - 1] = 0;          // frame column = canvas column - 1
                HIPCHK(hipMalloc((void**)&d_col_ok_, ok.size()), "hipMalloc(gutter columns)");
                HIPCHK(hipMemcpy(d_col_ok_, ok.data(), ok.size(), hipMemcpyHostToDevice), "upload gutter columns");
            }
            // (every row of planes but the last is tile + 2 * prepad tall: the borders they share are one period apart)
            if (yt > 1) { pair_gut_first_ = row_y[1] - 1; pair_gut_period_ = row_y[1]; }
        }
        pair_strips_ = (pair_w_ + PAIR_VALID - 1) / PAIR_VALID;
        int segs = std::max(1, n_cu_ / pair_strips_);          // never more units than CUs: a workgroup with two units would double the launch
        int seg_h = (pair_h_ + segs - 1) / segs;
        seg_h = std::max(16, (seg_h + 1) & ~1);
        pair_seg_h_ = seg_h;
        pair_segs_ = (pair_h_ + seg_h - 1) / seg_h;
    }
    // conv_last of tiled frames on rolling strips (kernels_last.hip, the CANVAS instantiation; round 6): one unit = a strip of
    // PAIR_VALID columns x last_seg_h_ rows of ONE plane's interior (the plane without its apron: the apron's conv_last outputs
    // are dropped anyway), as many units as CUs if the planes allow it (1080p, tile 200: 228 strips of up to 200 rows)
    if (tile != 0) {
        long long strips_total = 0;
        int max_ih = 0, max_strips = 0;
        for (const PlaneDesc& p : planes) {
            const int st = (p.w - 2 * pad_ + PAIR_VALID - 1) / PAIR_VALID;
            strips_total += st;
            max_strips = std::max(max_strips, st);
            max_ih = std::max(max_ih, p.h - 2 * pad_);
        }
        const int segs = (int)std::max<long long>(1, n_cu_ / std::max<long long>(strips_total, 1));
        last_seg_h_ = std::max(16, ((max_ih + segs - 1) / segs + 3) & ~3);       // whole steps of four rows
        if (n_planes_ <= 4096 && max_strips <= 255 && (max_ih + last_seg_h_ - 1) / last_seg_h_ <= 4095) {
            std::vector<uint32_t> units;
            for (int p = 0; p < n_planes_; ++p) {
                const int st = (planes[p].w - 2 * pad_ + PAIR_VALID - 1) / PAIR_VALID, ih = planes[p].h - 2 * pad_;
                for (int sy = 0; sy * last_seg_h_ < ih; ++sy)
                    for (int sx = 0; sx < st; ++sx) units.push_back((uint32_t)p | ((uint32_t)sx << 12) | ((uint32_t)sy << 20));
            }
            n_last_units_ = (int)units.size();
            HIPCHK(hipMalloc((void**)&d_last_units_, units.size() * 4), "hipMalloc(conv_last units)");
            HIPCHK(hipMemcpy(d_last_units_, units.data(), units.size() * 4, hipMemcpyHostToDevice), "upload conv_last units");
        }
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream_), "sync after configure");
    geo_w_ = w; geo_h_ = h; geo_tile_ = tile; geo_batching_ = batching_;
    stats_.body_layers_per_launch = 1;
    stats_.frame_w = w; stats_.frame_h = h; stats_.planes = n_planes_;
    stats_.tiles_per_plane = tiles_x_ * tiles_y_;
    return 0;
}

void Engine::harvest_events(bool all)
{
    for (auto& e : evpool_) {
        if (!e.used) continue;
        if (!all && hipEventQuery((hipEvent_t)e.f1) != hipSuccess) continue;
        (void)hipEventSynchronize((hipEvent_t)e.f1);
        float ms = 0;
        if (hipEventElapsedTime(&ms, (hipEvent_t)e.b0, (hipEvent_t)e.b1) == hipSuccess) {
            stats_.body_ms_total += ms;
            stats_.body_launches += n_body_;
        }
        if (hipEventElapsedTime(&ms, (hipEvent_t)e.f0, (hipEvent_t)e.f1) == hipSuccess) {
            stats_.frame_ms_last = ms;
            float first = 0, last = 0;
            if (hipEventElapsedTime(&first, (hipEvent_t)e.f0, (hipEvent_t)e.b0) == hipSuccess &&
                hipEventElapsedTime(&last, (hipEvent_t)e.b1, (hipEvent_t)e.f1) == hipSuccess) {
                stats_.frames_timed += e.k;            // (per-frame figures = totals / frames_timed: a chain's time is shared by its k frames)
                stats_.frame_ms_total += ms; stats_.first_ms_total += first; stats_.last_ms_total += last;
            }
        }
        e.used = false;
    }
}

// conv_first -> 16 x body -> conv_last on the compute stream.  Consecutive layers walk the tiles in
// opposite directions so that a layer starts on the data its producer wrote last (still in the
// 256 MiB Infinity Cache when one activation does not fit).
int Engine::enqueue_chain(const uint8_t* d_src, ptrdiff_t ss, uint8_t* d_dst, ptrdiff_t ds, int stop_after)
{
    return enqueue_chain_k(&d_src, &d_dst, 1, ss, ds, stop_after);
}

// k frames of the current geometry through ONE chain (k <= batch_; k > 1 only on a geometry configured for several frames per
// launch: frame f is plane f of the canvas)
int Engine::enqueue_chain_k(const uint8_t* const* d_srcs, uint8_t* const* d_dsts, int k, ptrdiff_t ss, ptrdiff_t ds, int stop_after)
{
    if (k < 1 || k > batch_) return fail(REVE_E_INVALID, "more frames than the geometry takes per launch");
    const bool stacked = batch_ > 1;                 // the planes are frames, one below the other
    const uint8_t* const d_src = d_srcs[0];
    uint8_t* const d_dst = d_dsts[0];
    const int n_planes = stacked ? k : n_planes_, n_items = stacked ? k * items_per_plane_ : n_items_;
    hipStream_t st = (hipStream_t)stream_;
    EvRec* rec = nullptr;
    if (profiling_ && stop_after < 0) {
        rec = &evpool_[ev_next_];
        if (rec->used) { harvest_events(false); if (rec->used) { (void)hipEventSynchronize((hipEvent_t)rec->f1); harvest_events(false); } }
        ev_next_ = (ev_next_ + 1) % evpool_.size();
        (void)hipEventRecord((hipEvent_t)rec->f0, st);
    }
    FirstArgs fa{};
    fa.src = d_src; fa.src_stride = ss; fa.frame_w = geo_w_; fa.frame_h = geo_h_;
    fa.out = arena_[0]; fa.wpack = first_.wpack; fa.bias = first_.bias; fa.slope = first_.slope;
    fa.planes = d_planes_; fa.plane_stride = plane_stride_;
    fa.n_planes = n_planes; fa.tiles_x = tiles_x_; fa.tiles_y = tiles_y_; fa.Wp = Wp_;
    fa.n_items = n_items; fa.items = d_items_; fa.blocked = blocked_order_;
    if (stacked) {
        fa.n_src = k;
        for (int f = 0; f < k; ++f) fa.src_tab[f] = d_srcs[f];
    }
    // two workgroups per CU, each fetching the source of FIRST_NT tiles ahead of its stores (measured best)
    int rc = launch_first(fa, std::min(n_items, n_cu_ * 2), st);
    if (rc) return hipfail(rc, "launch conv_first");

    ConvArgs ca{};
    ca.planes = d_planes_; ca.plane_stride = plane_stride_;
    ca.n_planes = n_planes; ca.tiles_x = tiles_x_; ca.tiles_y = tiles_y_;
    ca.n_items = n_items; ca.items = d_items_; ca.blocked = blocked_order_; ca.Wp = Wp_;
    ca.src = d_src; ca.src_stride = ss; ca.dst = d_dst; ca.dst_stride = ds;
    ca.frame_w = geo_w_; ca.frame_h = geo_h_; ca.pad = pad_;
    const int grid = std::min(n_cu_, ca.n_items);
    int cur = 0;
    const int nb = stop_after < 0 ? n_body_ : std::min(stop_after, n_body_);
    if (rec) (void)hipEventRecord((hipEvent_t)rec->b0, st);
    for (int l = 0; l < nb; ++l) {
        if (fuse_pairs_ && pair_strips_ > 0 && l + 1 < nb) {
            // layers l and l+1 in one launch: the activation between them stays in LDS (kernels_pair.hip)
            PairArgs pa{};
            pa.in = arena_[cur]; pa.out = arena_[cur ^ 1];
            const bool wino = winograd_;
            for (int k = 0; k < 2; ++k) {
                pa.wpack[k] = wino ? body_wino_[l + k] : body_[l + k].wpack;
                pa.bias[k] = body_[l + k].bias; pa.slope[k] = body_[l + k].slope;
            }
            pa.W = pair_w_; pa.H = pair_h_; pa.Wp = Wp_; pa.Hp = Hp_;
            pa.col_ok = d_col_ok_; pa.gut_first = pair_gut_first_; pa.gut_period = pair_gut_period_;
            pa.n_strips = pair_strips_; pa.n_segs = pair_segs_; pa.seg_h = pair_seg_h_;
            pa.n_units = pair_strips_ * pair_segs_;
            if (stacked) {
                // the canvas as far as this launch's k frames reach: k planes of geo_h_ rows and the k - 1 gutter rows between them
                pa.H = k * (geo_h_ + 1) - 1; pa.Hp = pa.H + 2;
                if (k == 1) { pa.gut_first = pa.gut_period = 0; }
                const int segs = std::max(1, n_cu_ / pa.n_strips);
                pa.seg_h = std::max(16, ((pa.H + segs - 1) / segs + 1) & ~1);
                pa.n_segs = (pa.H + pa.seg_h - 1) / pa.seg_h;
                pa.n_units = pa.n_strips * pa.n_segs;
            }
            pa.reverse = ((l >> 1) & 1) ^ 1;
            pa.unit_slopes = body_unit_slopes_[l] && body_unit_slopes_[l + 1];
            rc = wino ? launch_wino(pa, std::min(n_cu_, pa.n_units), st) : launch_pair(pa, std::min(n_cu_, pa.n_units), st);
            if (rc) return hipfail(rc, "launch fused body pair");
            cur ^= 1;
            ++l;
            continue;
        }
        ca.in = arena_[cur]; ca.out = arena_[cur ^ 1];
        ca.wpack = body_[l].wpack; ca.bias = body_[l].bias; ca.slope = body_[l].slope;
        ca.unit_slopes = body_unit_slopes_[l];
        ca.reverse = (l & 1) ^ 1;
        rc = launch_body(ca, grid, st);
        if (rc) return hipfail(rc, "launch body conv");
        cur ^= 1;
    }
    if (rec) (void)hipEventRecord((hipEvent_t)rec->b1, st);
    last_arena_ = cur;
    if (stop_after >= 0) return 0;
    ca.in = arena_[cur]; ca.out = nullptr;
    ca.wpack = last_.wpack; ca.bias = last_.bias; ca.slope = nullptr;
    ca.reverse = (nb & 1) ^ 1;
    // (its store offsets use 0x40000000 as "nowhere": output frames below 1 GiB)
    // (several frames per launch: always the strip kernel — it takes one source / destination per frame, the tile kernel does not)
    const bool strip_ok = pad_ == 0 && (long long)ds * geo_h_ * cfg_.scale < 0x40000000ll;
    // (the tile kernel below takes ONE source / destination: with several frames in the launch it would read frame 0's residual
    // and write frame 0's output for all of them.  Small frames with an output stride that large are refused, not mis-written.)
    if (stacked && k > 1 && !strip_ok)
        return fail(REVE_E_UNSUPPORTED, "output stride too large for frames that share a launch (set option \"batch\" 0)");
    if (strip_last_ && !stacked && d_last_units_ && cfg_.scale != 4 && (long long)ds * geo_h_ * cfg_.scale < 0x40000000ll) {
        // tiled frame: conv_last rolls down strips of the planes' interiors (kernels_last.hip, CANVAS): same bytes as the tile kernel.
        // Measured at 1080p with the binary's 200-pixel tiles (profiles/r06/ab_conv_last_canvas_strips.txt): x2 90.7 -> 77.4 us
        // (tile 100: 117.8 -> 85.2; 4K: 345 -> 283), x3 118.7 -> 114.0; x4 — MFMA-bound, where the strips' 256 computed columns per
        // 200 cost more than the tile kernel's apron — 142.1 -> 150.4: x4 keeps the tile kernel
        LastStripArgs la{};
        la.in = arena_[cur]; la.wpack = last_.wpack; la.bias = last_.bias;
        la.src = d_src; la.src_stride = ss; la.dst = d_dst; la.dst_stride = ds;
        la.W = geo_w_; la.H = geo_h_; la.Wp = Wp_; la.Hp = Hp_;
        la.seg_h = last_seg_h_; la.n_units = n_last_units_;
        la.planes = d_planes_; la.units = d_last_units_; la.pad = pad_;
        la.reverse = (nb & 1) ^ 1;
        rc = launch_last_strip(la, cfg_.scale, std::min(n_cu_, la.n_units), st);
    } else if ((strip_last_ || stacked) && (n_planes_ == 1 || stacked) && strip_ok) {
        // whole frame: conv_last rolls down strips with its input streamed through a ring of rows (kernels_last.hip)
        LastStripArgs la{};
        la.in = arena_[cur]; la.wpack = last_.wpack; la.bias = last_.bias;
        la.src = d_src; la.src_stride = ss; la.dst = d_dst; la.dst_stride = ds;
        la.W = geo_w_; la.H = geo_h_; la.Wp = Wp_; la.Hp = stacked ? geo_h_ + 2 : Hp_;
        la.n_strips = (geo_w_ + PAIR_VALID - 1) / PAIR_VALID;
        const int segs = std::max(1, n_cu_ / (la.n_strips * k));
        la.seg_h = std::max(16, ((geo_h_ + segs - 1) / segs + 3) & ~3);       // whole steps of four rows
        la.n_units = la.n_strips * ((geo_h_ + la.seg_h - 1) / la.seg_h);
        if (stacked && k > 1) {
            la.n_frames = k; la.units_per_frame = la.n_units; la.n_units *= k;
            la.in_frame_stride = (long long)(geo_h_ + 1) * Wp_ * PIX_BYTES;
            for (int f = 0; f < k; ++f) { la.src_tab[f] = d_srcs[f]; la.dst_tab[f] = d_dsts[f]; }
        }
        la.reverse = (nb & 1) ^ 1;
        rc = launch_last_strip(la, cfg_.scale, std::min(n_cu_, la.n_units), st);
    } else {
        ca.n_planes = n_planes; ca.n_items = n_items;      // (stacked geometry, k == 1: plane 0 is the frame)
        rc = launch_last(ca, cfg_.scale, grid, st);
    }
    if (rc) return hipfail(rc, "launch conv_last");
    if (rec) { (void)hipEventRecord((hipEvent_t)rec->f1, st); rec->used = true; rec->k = k; }
    if (!capturing_ && !ring_chain_) unretired_ += k;      // (ring frames are counted by reve_wait; the others when the stream is known to have drained)
    return 0;
}

int Engine::ensure_slot(Slot& s, size_t in_bytes, size_t out_bytes)
{
    if (s.in_cap >= in_bytes && s.out_cap >= out_bytes && s.ev_h2d) return 0;
    std::lock_guard<std::mutex> unsafe_lk(unsafe_calls_mutex());
    if ((s.in_cap < in_bytes || s.out_cap < out_bytes) && s.graph_exec) {
        (void)hipGraphExecDestroy((hipGraphExec_t)s.graph_exec);      // captured with the old buffers
        s.graph_exec = nullptr;
    }
    if (s.in_cap < in_bytes) {
        if (s.d_in) (void)hipFree(s.d_in);
        s.d_in = nullptr; s.in_cap = 0;
        HIPCHK(hipMalloc(&s.d_in, in_bytes), "hipMalloc(input frame)");
        s.in_cap = in_bytes;
    }
    if (s.out_cap < out_bytes) {
        if (s.d_out) (void)hipFree(s.d_out);
        s.d_out = nullptr; s.out_cap = 0;
        HIPCHK(hipMalloc(&s.d_out, out_bytes), "hipMalloc(output frame)");
        s.out_cap = out_bytes;
    }
    if (!s.ev_h2d) {
        hipEvent_t e;
        // timing-capable: with profiling on, reve_wait reads the three stages' device times from them
        HIPCHK(hipEventCreate(&e), "hipEventCreate"); s.ev_h2d = e;
        HIPCHK(hipEventCreate(&e), "hipEventCreate"); s.ev_comp = e;
        // the event reve_wait sleeps on: blocking sync, so that the feeder threads of a multi-GPU host (one per GPU, dirmode.cpp)
        // yield their CPUs while their frames are in flight instead of spinning on them (a box may allow 16 CPUs for 8 GPUs);
        // with three frames on the ring the wake-up latency is hidden
        HIPCHK(hipEventCreateWithFlags(&e, hipEventBlockingSync), "hipEventCreate"); s.ev_d2h = e;
        HIPCHK(hipEventCreate(&e), "hipEventCreate"); s.ev_h2d0 = e;
        HIPCHK(hipEventCreate(&e), "hipEventCreate"); s.ev_comp0 = e;
        HIPCHK(hipEventCreate(&e), "hipEventCreate"); s.ev_d2h0 = e;
    }
    return 0;
}

static bool bad_frame(const void* src, int w, int h, ptrdiff_t ss, const void* dst, ptrdiff_t ds, int scale)
{
    if (!src || !dst || w <= 0 || h <= 0 || ss < (ptrdiff_t)w * 3 || ds < (ptrdiff_t)w * 3 * scale) return true;
    // the kernels address both frames with 32-bit byte offsets through buffer descriptors
    const unsigned long long lim = 1ull << 31;
    return (unsigned long long)ss * h >= lim || (unsigned long long)ds * h * scale >= lim;
}

int Engine::upscale_device(const void* d_src, int w, int h, ptrdiff_t ss, void* d_dst, ptrdiff_t ds)
{
    if (!inited_) return fail(REVE_E_INVALID, "context not initialised");
    if (bad_frame(d_src, w, h, ss, d_dst, ds, cfg_.scale)) return fail(REVE_E_INVALID, "bad frame arguments");
    if (ring_count_) return fail(REVE_E_BUSY, "frames in flight on the async ring");      // (a re-configure would pull the geometry from under them)
    HIPCHK(hipSetDevice(cfg_.device), "hipSetDevice");
    int rc = configure(w, h, false);
    if (rc) return rc;
    return enqueue_chain((const uint8_t*)d_src, ss, (uint8_t*)d_dst, ds, -1);
}

int Engine::upscale_device_batch(int n, const void* const* d_srcs, void* const* d_dsts, int w, int h, ptrdiff_t ss, ptrdiff_t ds)
{
    if (!inited_) return fail(REVE_E_INVALID, "context not initialised");
    if (n <= 0 || !d_srcs || !d_dsts) return fail(REVE_E_INVALID, "bad batch arguments");
    for (int i = 0; i < n; ++i)
        if (bad_frame(d_srcs[i], w, h, ss, d_dsts[i], ds, cfg_.scale)) return fail(REVE_E_INVALID, "bad frame arguments");
    if (ring_count_) return fail(REVE_E_BUSY, "frames in flight on the async ring");
    HIPCHK(hipSetDevice(cfg_.device), "hipSetDevice");
    int rc = configure(w, h, false);
    if (rc) return rc;
    for (int i = 0; i < n; i += batch_) {
        const int k = std::min(batch_, n - i);
        if ((rc = enqueue_chain_k((const uint8_t* const*)d_srcs + i, (uint8_t* const*)d_dsts + i, k, ss, ds, -1))) return rc;
    }
    return 0;
}

int Engine::sync()
{
    if (!inited_) return fail(REVE_E_INVALID, "context not initialised");
    if (int rc = flush_pending()) return rc;        // (frames of the ring that still wait for their batch: "everything enqueued" includes them)
    HIPCHK(hipStreamSynchronize((hipStream_t)stream_), "hipStreamSynchronize");
    stats_.frames_done += unretired_;
    unretired_ = 0;
    return 0;
}

int Engine::upscale_host(const uint8_t* src, int w, int h, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds)
{
    if (!inited_) return fail(REVE_E_INVALID, "context not initialised");
    if (bad_frame(src, w, h, ss, dst, ds, cfg_.scale)) return fail(REVE_E_INVALID, "bad frame arguments");
    if (ring_count_) return fail(REVE_E_BUSY, "frames in flight on the async ring");
    HIPCHK(hipSetDevice(cfg_.device), "hipSetDevice");
    int rc = configure(w, h, false);
    if (rc) return rc;
    const int s = cfg_.scale;
    const size_t in_row = (size_t)w * 3, out_row = in_row * s;
    if ((rc = ensure_slot(sync_slot_, in_row * h, out_row * h * s))) return rc;
    hipStream_t st = (hipStream_t)stream_;
    HIPCHK(hipMemcpy2DAsync(sync_slot_.d_in, in_row, src, ss, in_row, h, hipMemcpyHostToDevice, st), "H2D");
    if ((rc = enqueue_chain((const uint8_t*)sync_slot_.d_in, in_row, (uint8_t*)sync_slot_.d_out, out_row, -1))) return rc;
    HIPCHK(hipMemcpy2DAsync(dst, ds, sync_slot_.d_out, out_row, out_row, (size_t)h * s, hipMemcpyDeviceToHost, st), "D2H");
    HIPCHK(hipStreamSynchronize(st), "hipStreamSynchronize");
    stats_.frames_done += unretired_;
    unretired_ = 0;
    stats_.h2d_bytes += in_row * h;
    stats_.d2h_bytes += out_row * h * s;
    return 0;
}

// Frame ring: upload on s_h2d_, the 18-kernel chain on stream_, download on s_d2h_, chained by
// events, so frame n+1's upload and frame n-1's download run under frame n's kernels.
int Engine::submit(uint64_t id, const uint8_t* src, int w, int h, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds)
{
    const int rc = submit_frame(id, src, w, h, ss, dst, ds);
    // A failure behind the frame's upload (a launch, an event) must not leave that upload reading `src`: the caller may take the
    // failed call for "not taken" and release the frame.  (REVE_E_BUSY — the ring is full, the steady state of a pipelined caller —
    // and argument errors return before anything is queued.)
    if (rc && rc != REVE_E_BUSY && rc != REVE_E_INVALID && s_h2d_) (void)hipStreamSynchronize((hipStream_t)s_h2d_);
    return rc;
}

int Engine::submit_frame(uint64_t id, const uint8_t* src, int w, int h, ptrdiff_t ss, uint8_t* dst, ptrdiff_t ds)
{
    if (!inited_) return fail(REVE_E_INVALID, "context not initialised");
    if (bad_frame(src, w, h, ss, dst, ds, cfg_.scale)) return fail(REVE_E_INVALID, "bad frame arguments");
    HIPCHK(hipSetDevice(cfg_.device), "hipSetDevice");
    if ((w != geo_w_ || h != geo_h_) && ring_count_) return fail(REVE_E_BUSY, "frame size changed with frames in flight");
    int rc = configure(w, h, false);
    if (rc) return rc;
    if (ring_count_ >= ring_cap()) {
        // (frames that wait for their batch to fill do not block the ring: their chain is launched now, the caller retires one and comes back)
        if (!pending_.empty() && (rc = flush_pending())) return rc;
        return fail(REVE_E_BUSY, "ring full: call reve_wait first");
    }
    const int s = cfg_.scale;
    const size_t in_row = (size_t)w * 3, out_row = in_row * s;
    Slot& sl = ring_[(ring_head_ + ring_count_) % ring_.size()];
    if ((rc = ensure_slot(sl, in_row * h, out_row * h * s))) return rc;
    sl.id = id;
    sl.launched = true; sl.batch_k = 1; sl.failed = 0;
    hipStream_t sc = (hipStream_t)stream_, su = (hipStream_t)s_h2d_, sd = (hipStream_t)s_d2h_;
    // stage-start events sit behind the stream's wait, so a stage's time is its own work, not its queueing
    sl.timed = profiling_;
    if (sl.timed && stats_.ring_frames == 0 && ring_count_ == 0)
        ring_t0_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    {
        TraceRange tr("reve:upload");
        if (sl.timed) HIPCHK(hipEventRecord((hipEvent_t)sl.ev_h2d0, su), "record h2d start");
        HIPCHK(hipMemcpy2DAsync(sl.d_in, in_row, src, ss, in_row, h, hipMemcpyHostToDevice, su), "H2D");
        HIPCHK(hipEventRecord((hipEvent_t)sl.ev_h2d, su), "record h2d");
    }
    if (batch_ > 1) {
        // several frames per launch: the frame waits (uploaded) until batch_ frames are there or reve_wait asks for one of them
        sl.launched = false;
        sl.dst = dst; sl.dst_stride = ds; sl.w = w; sl.h = h;
        pending_.push_back((ring_head_ + ring_count_) % ring_.size());
        ring_count_++;
        stats_.h2d_bytes += in_row * h;
        stats_.d2h_bytes += out_row * h * s;
        // Launch now when the batch is full.  A caller whose ring cannot hold two batches (an explicit ring_depth under 2 x batch: the
        // `reve` CLI's lanes keep three frames in flight) would otherwise leave the chip idle while it reads its next frames — its
        // chain used to start only when reve_wait reached the frame — so for such a ring a partial batch is launched
        // as soon as the compute stream has nothing to do.  A ring of two batches (the default) never asks: hipStreamQuery costs
        // ~50 us on a stream with work in flight, three times the whole per-frame cost of a 100x100 frame (34,000 -> 11,500
        // frames/s when every submit asked, profiles/r05/ab_batch_query_every_submit.txt).
        if (pending_.size() >= std::min((size_t)batch_, ring_cap())) return flush_pending();
        if (ring_cap() < 2 * (size_t)batch_ && hipStreamQuery(sc) == hipSuccess) return flush_pending();
        return 0;
    }
    HIPCHK(hipStreamWaitEvent(sc, (hipEvent_t)sl.ev_h2d, 0), "wait h2d");
    TraceRange tr_chain("reve:chain");
    if (sl.timed) HIPCHK(hipEventRecord((hipEvent_t)sl.ev_comp0, sc), "record compute start");
    bool launched = false;
    if (use_graph_ && !profiling_ && batch_ == 1) {
        // one launch per frame: the chain of this slot (its buffers are the kernels' arguments) is captured once per geometry
        if (sl.graph_exec && (sl.g_w != w || sl.g_h != h || sl.g_tile != geo_tile_ || sl.g_fuse != fuse_pairs_ || sl.g_wino != winograd_)) {
            (void)hipGraphExecDestroy((hipGraphExec_t)sl.graph_exec);
            sl.graph_exec = nullptr;
        }
        if (!sl.graph_exec) {
            // A capture is invalidated by "unsafe" runtime calls made meanwhile (allocations, synchronous copies — e.g. a second
            // context of the same process configuring its arenas on another thread; seen as "operation failed due to a previous
            // error during capture" with two lanes on one GPU): the library's own such calls (configure, pinned allocations) and
            // its captures take turns on one process-wide mutex, and a capture that fails all the same is abandoned — the frame
            // is launched kernel by kernel and this context stops using graphs.
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            bool ok = false;
            {
                std::lock_guard<std::mutex> lk(unsafe_calls_mutex());
                if (hipStreamBeginCapture(sc, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    capturing_ = true;
                    const int crc = enqueue_chain((const uint8_t*)sl.d_in, in_row, (uint8_t*)sl.d_out, out_row, -1);
                    capturing_ = false;
                    const hipError_t e = hipStreamEndCapture(sc, &graph);
                    ok = crc == 0 && e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess && exec;
                }
                if (graph) (void)hipGraphDestroy(graph);
            }
            if (ok) {
                sl.graph_exec = exec;
                sl.g_w = w; sl.g_h = h; sl.g_tile = geo_tile_; sl.g_fuse = fuse_pairs_; sl.g_wino = winograd_;
            } else {
                (void)hipGetLastError();
                use_graph_ = false;
            }
        }
        if (sl.graph_exec) {
            HIPCHK(hipGraphLaunch((hipGraphExec_t)sl.graph_exec, sc), "hipGraphLaunch");
            launched = true;
        }
    }
    if (!launched) {
        ring_chain_ = true;
        rc = enqueue_chain((const uint8_t*)sl.d_in, in_row, (uint8_t*)sl.d_out, out_row, -1);
        ring_chain_ = false;
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord((hipEvent_t)sl.ev_comp, sc), "record compute");
    tr_chain.end();
    TraceRange tr_down("reve:download");
    HIPCHK(hipStreamWaitEvent(sd, (hipEvent_t)sl.ev_comp, 0), "wait compute");
    if (sl.timed) HIPCHK(hipEventRecord((hipEvent_t)sl.ev_d2h0, sd), "record d2h start");
    HIPCHK(hipMemcpy2DAsync(dst, ds, sl.d_out, out_row, out_row, (size_t)h * s, hipMemcpyDeviceToHost, sd), "D2H");
    HIPCHK(hipEventRecord((hipEvent_t)sl.ev_d2h, sd), "record d2h");
    ring_count_++;
    stats_.h2d_bytes += in_row * h;
    stats_.d2h_bytes += out_row * h * s;
    return 0;
}

size_t Engine::ring_cap() const
{
    // the depth the caller asked for, always; left to the library (ring_depth <= 0): 3, or — frames that share launches — a batch
    // computing and a batch filling
    return batch_ > 1 && ring_auto_ ? std::min(ring_.size(), std::max((size_t)cfg_.ring_depth, (size_t)2 * batch_)) : (size_t)cfg_.ring_depth;
}

int Engine::flush_pending()
{
    if (pending_.empty()) return 0;
    // The frames leave the pending list HERE: whatever fails below, a later reve_wait / reve_sync must not launch their chain a
    // second time (an error after the chain was enqueued used to leave them pending).  A failure is fatal for these slots: their
    // completion events may never have been recorded — hipEventSynchronize on such an event returns at once — so every slot of the
    // batch remembers the error and reve_wait hands it out for that frame instead of reporting bytes that were never written.
    const std::vector<size_t> batch = std::move(pending_);
    pending_.clear();
    const int rc = launch_batch(batch);
    if (rc) {
        launch_err_ = err_;
        for (size_t i : batch) ring_[i].failed = rc;
        // the frames' uploads were queued when they were submitted: none of them may still be reading a caller's buffer when the
        // error comes back (a caller may take a failed reve_submit for "not taken" and release the frame)
        (void)hipStreamSynchronize((hipStream_t)s_h2d_);
    }
    return rc;
}

int Engine::launch_batch(const std::vector<size_t>& batch)
{
    hipStream_t sc = (hipStream_t)stream_, sd = (hipStream_t)s_d2h_;
    const int k = (int)batch.size(), s = cfg_.scale;
    const uint8_t* srcs[MAX_BATCH];
    uint8_t* dsts[MAX_BATCH];
    for (int i = 0; i < k; ++i) {
        Slot& sl = ring_[batch[i]];
        sl.launched = true; sl.batch_k = k;
        srcs[i] = (const uint8_t*)sl.d_in; dsts[i] = (uint8_t*)sl.d_out;
    }
    for (int i = 0; i < k; ++i) HIPCHK(hipStreamWaitEvent(sc, (hipEvent_t)ring_[batch[i]].ev_h2d, 0), "wait h2d");
    const size_t in_row = (size_t)geo_w_ * 3, out_row = in_row * s;
    int rc;
    {
        TraceRange tr_chain("reve:chain");
        for (int i = 0; i < k; ++i)
            if (ring_[batch[i]].timed) HIPCHK(hipEventRecord((hipEvent_t)ring_[batch[i]].ev_comp0, sc), "record compute start");
        ring_chain_ = true;
        rc = enqueue_chain_k(srcs, dsts, k, in_row, out_row, -1);
        ring_chain_ = false;
        if (rc) return rc;
        if (inject_launch_failure_) {       // (test hook, option "debug_fail_launch": the error path between the chain and its events)
            inject_launch_failure_ = false;
            return fail(REVE_E_HIP, "injected launch failure (option debug_fail_launch)");
        }
        for (int i = 0; i < k; ++i) HIPCHK(hipEventRecord((hipEvent_t)ring_[batch[i]].ev_comp, sc), "record compute");
    }
    TraceRange tr_down("reve:download");
    for (int i = 0; i < k; ++i) {
        Slot& sl = ring_[batch[i]];
        HIPCHK(hipStreamWaitEvent(sd, (hipEvent_t)sl.ev_comp, 0), "wait compute");
        if (sl.timed) HIPCHK(hipEventRecord((hipEvent_t)sl.ev_d2h0, sd), "record d2h start");
        HIPCHK(hipMemcpy2DAsync(sl.dst, sl.dst_stride, sl.d_out, out_row, out_row, (size_t)sl.h * s, hipMemcpyDeviceToHost, sd), "D2H");
        HIPCHK(hipEventRecord((hipEvent_t)sl.ev_d2h, sd), "record d2h");
    }
    return 0;
}

int Engine::wait(uint64_t* id)
{
    if (!inited_) return fail(REVE_E_INVALID, "context not initialised");
    if (!ring_count_) return fail(REVE_E_BUSY, "nothing in flight");
    if (!ring_[ring_head_].launched) (void)flush_pending();      // the batch it waits in will not fill by itself (a failure stays on its slots)
    Slot& sl = ring_[ring_head_];
    if (sl.failed) {
        // its chain or its download was never (completely) queued: the frame leaves the ring with the error, nothing was written
        const int rc = sl.failed;
        sl.failed = 0; sl.timed = false;
        if (id) *id = sl.id;
        ring_head_ = (ring_head_ + 1) % ring_.size();
        ring_count_--;
        return fail(rc, "frame " + std::to_string(sl.id) + " was not upscaled: its launch failed (" + launch_err_ + ")");
    }
    {
        TraceRange tr("reve:wait");
        HIPCHK(hipEventSynchronize((hipEvent_t)sl.ev_d2h), "hipEventSynchronize");
    }
    if (sl.timed) {
        float a = 0, b = 0, c = 0;
        if (hipEventElapsedTime(&a, (hipEvent_t)sl.ev_h2d0, (hipEvent_t)sl.ev_h2d) == hipSuccess &&
            hipEventElapsedTime(&b, (hipEvent_t)sl.ev_comp0, (hipEvent_t)sl.ev_comp) == hipSuccess &&
            hipEventElapsedTime(&c, (hipEvent_t)sl.ev_d2h0, (hipEvent_t)sl.ev_d2h) == hipSuccess) {
            stats_.ring_frames++;
            stats_.h2d_ms_total += a; stats_.chain_ms_total += b / sl.batch_k; stats_.d2h_ms_total += c;
            stats_.ring_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - ring_t0_;
        }
        sl.timed = false;
    }
    if (id) *id = sl.id;
    stats_.frames_done++;                 // its bytes are in the caller's buffer
    ring_head_ = (ring_head_ + 1) % ring_.size();
    ring_count_--;
    return 0;
}

int Engine::debug_run_layers(const uint8_t* src, int w, int h, ptrdiff_t ss, int layer, float* out, size_t n)
{
    if (!inited_) return fail(REVE_E_INVALID, "context not initialised");
    const bool last = layer == n_body_ + 1;              // conv_last's fp16 output, before PixelShuffle / residual / quantisation
    const int ch = last ? 3 * cfg_.scale * cfg_.scale : FEAT;
    if (!src || !out || w <= 0 || h <= 0 || layer < 0 || layer > n_body_ + 1 || n < (size_t)w * h * ch)
        return fail(REVE_E_INVALID, "bad debug_run_layers arguments");
    if (ring_count_) return fail(REVE_E_BUSY, "frames in flight on the async ring");
    HIPCHK(hipSetDevice(cfg_.device), "hipSetDevice");
    int rc = configure(w, h, true);
    if (rc) return rc;
    const size_t in_row = (size_t)w * 3;
    if ((rc = ensure_slot(sync_slot_, in_row * h, in_row * h * cfg_.scale * cfg_.scale))) return rc;
    hipStream_t st = (hipStream_t)stream_;
    HIPCHK(hipMemcpy2DAsync(sync_slot_.d_in, in_row, src, ss, in_row, h, hipMemcpyHostToDevice, st), "H2D");
    if ((rc = enqueue_chain((const uint8_t*)sync_slot_.d_in, in_row, nullptr, 0, last ? n_body_ : layer))) return rc;
    if (last) {
        // the probe instantiation of the conv_last kernel: [pixel][16 * co-blocks] fp16 in pack_last()'s store order
        const int ncob = cfg_.scale == 2 ? 1 : (cfg_.scale == 3 ? 2 : 3), nrow = 16 * ncob;
        const size_t bytes = (size_t)w * h * nrow * 2;
        if (bytes >= ((size_t)1 << 31)) return fail(REVE_E_UNSUPPORTED, "frame too large for the conv_last probe");
        void* d_probe = nullptr;
        {
            std::lock_guard<std::mutex> unsafe_lk(unsafe_calls_mutex());
            HIPCHK(hipMalloc(&d_probe, bytes), "hipMalloc(probe)");
        }
        ConvArgs ca{};
        ca.planes = d_planes_; ca.plane_stride = plane_stride_;
        // (a geometry laid out for several frames per launch: the probe looks at the first plane only)
        ca.n_planes = batch_ > 1 ? 1 : n_planes_; ca.tiles_x = tiles_x_; ca.tiles_y = tiles_y_;
        ca.n_items = ca.n_planes * tiles_x_ * tiles_y_; ca.items = nullptr; ca.blocked = 0; ca.Wp = Wp_;
        ca.src = (const uint8_t*)sync_slot_.d_in; ca.src_stride = in_row; ca.dst = (uint8_t*)d_probe; ca.dst_stride = 0;
        ca.frame_w = w; ca.frame_h = h; ca.pad = 0;
        ca.in = arena_[last_arena_]; ca.out = nullptr;
        ca.wpack = last_.wpack; ca.bias = last_.bias; ca.slope = nullptr;
        rc = launch_last_probe(ca, cfg_.scale, std::min(n_cu_, ca.n_items), st);
        std::vector<uint16_t> host((size_t)w * h * nrow);
        hipError_t e = rc ? hipSuccess : hipMemcpyAsync(host.data(), d_probe, bytes, hipMemcpyDeviceToHost, st);
        if (!rc && e == hipSuccess) e = hipStreamSynchronize(st);
        {
            std::lock_guard<std::mutex> unsafe_lk(unsafe_calls_mutex());
            (void)hipFree(d_probe);
        }
        if (rc) return hipfail(rc, "launch conv_last probe");
        if (e != hipSuccess) return hipfail((int)e, "conv_last probe read-back");
        const std::vector<int> rows = last_rows(cfg_.scale, ch, true);
        for (size_t p = 0; p < (size_t)w * h; ++p)
            for (int r = 0; r < nrow; ++r)
                if (rows[r] >= 0) out[p * ch + rows[r]] = f16_to_f32(host[p * nrow + r]);
        return 0;
    }
    std::vector<uint16_t> host(plane_stride_ / 2);
    HIPCHK(hipMemcpyAsync(host.data(), arena_[last_arena_], plane_stride_, hipMemcpyDeviceToHost, st), "D2H arena");
    HIPCHK(hipStreamSynchronize(st), "sync");
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const uint16_t* px = host.data() + ((size_t)(y + 1) * Wp_ + (x + 1)) * FEAT;
            float* o = out + ((size_t)y * w + x) * FEAT;
            for (int c = 0; c < FEAT; ++c) o[c] = f16_to_f32(px[chan_phys(c)]);
        }
    return 0;
}

// "winograd" 2 = auto: the second numeric path runs only where the weights' conditioning leaves room for it (model.h).  The
// choice is announced once per process on stderr — in words that do not contain "done": reve counts stderr lines with that
// substring as finished frames (reve-cli/src/main.rs:266-273).
void Engine::apply_winograd_mode(bool announce)
{
    winograd_ = winograd_mode_ == 1 || (winograd_mode_ == 2 && kappa_ < WINOGRAD_KAPPA_LIMIT);
    static std::once_flag said;
    if (announce && winograd_mode_ == 2)
        std::call_once(said, [&] {
            std::fprintf(stderr, "libreve_hip: winograd auto -> %s (weights' conditioning estimate %.3f, limit %.2f)\n",
                         winograd_ ? "on" : "off", kappa_, WINOGRAD_KAPPA_LIMIT);
        });
}

int Engine::set_option(const std::string& name, int value)
{
    if (name == "winograd") {
        if (ring_count_) return fail(REVE_E_BUSY, "frames in flight on the async ring");
        if (value < 0 || value > 2) return fail(REVE_E_INVALID, "winograd: 0 (off), 1 (on) or 2 (auto)");
        winograd_mode_ = value;
        apply_winograd_mode(true);
        drop_graphs();
        return 0;
    }
    if (name == "debug_fail_launch") { inject_launch_failure_ = value != 0; return 0; }
    bool* sw = name == "fuse_pairs" ? &fuse_pairs_ : name == "graph" ? &use_graph_ : name == "strip_last" ? &strip_last_
             : name == "batch" ? &batching_ : nullptr;
    if (!sw) return fail(REVE_E_INVALID, "unknown option " + name);
    if (ring_count_) return fail(REVE_E_BUSY, "frames in flight on the async ring");
    *sw = value != 0;         // ("batch" takes effect at the next frame: the geometry is laid out again)
    drop_graphs();            // (captured with the old switches)
    return 0;
}

int Engine::get_option(const std::string& name, int* value) const
{
    if (!value) return REVE_E_INVALID;
    if (name == "fuse_pairs") { *value = fuse_pairs_ ? 1 : 0; return 0; }
    if (name == "graph") { *value = use_graph_ ? 1 : 0; return 0; }
    if (name == "winograd") { *value = winograd_ ? 1 : 0; return 0; }
    if (name == "batch") { *value = batching_ ? 1 : 0; return 0; }
    if (name == "batch_frames") { *value = batch_; return 0; }          // (read-only: frames per launch of the current geometry)
    if (name == "strip_last") { *value = strip_last_ ? 1 : 0; return 0; }
    if (name == "winograd_mode") { *value = winograd_mode_; return 0; }
    if (name == "winograd_kappa_permille") { *value = (int)std::min(kappa_ * 1000.0 + 0.5, 2.0e9); return 0; }
    // read-only: the fused-pair launch of the current geometry (one frame per launch: the full canvas; `batch_frames` share it)
    if (name.rfind("pair_", 0) == 0) {
        int H = pair_h_, seg_h = pair_seg_h_, segs = pair_segs_;
        if (batch_ > 1 && pair_strips_ > 0) {        // (as enqueue_chain_k lays a full batch out)
            H = batch_ * (geo_h_ + 1) - 1;
            const int per = std::max(1, n_cu_ / pair_strips_);
            seg_h = std::max(16, ((H + per - 1) / per + 1) & ~1);
            segs = (H + seg_h - 1) / seg_h;
        }
        if (name == "pair_strips") { *value = pair_strips_; return 0; }
        if (name == "pair_segments") { *value = segs; return 0; }
        if (name == "pair_seg_rows") { *value = seg_h; return 0; }
        if (name == "pair_units") { *value = pair_strips_ * segs; return 0; }
        if (name == "pair_mfma_per_launch") {
            // MFMA instructions (v_mfma_f32_16x16x32_f16: 16,384 FLOP each) one body-pair launch executes: per unit of NB rows the
            // first layer's two waves run ceil((NB + 2) / 2) active steps, the second layer's two ceil(NB / 2), and a step is 288
            // MFMAs per wave (direct; 192 in the Winograd kernel): kernels_pair.hip / kernels_wino.hip
            long long steps = 0;
            for (int sy = 0; sy < segs; ++sy) {
                const int nb = std::min(seg_h, H - sy * seg_h);
                steps += (nb + 2 + 1) / 2 + (nb + 1) / 2;
            }
            const long long n = steps * pair_strips_ * 2 * (winograd_ ? 192 : 288);
            *value = (int)std::min<long long>(n, 0x7fffffff);
            return 0;
        }
    }
    return REVE_E_INVALID;
}

int Engine::get_stats(Stats& s)
{
    harvest_events(false);
    if (unretired_ && hipStreamQuery((hipStream_t)stream_) == hipSuccess) { stats_.frames_done += unretired_; unretired_ = 0; }
    stats_.body_layers_per_launch = (fuse_pairs_ && pair_strips_ > 0) ? 2 : 1;
    s = stats_;
    return 0;
}

int Engine::reset_stats()
{
    harvest_events(true);
    const Stats keep = stats_;
    stats_ = Stats();
    stats_.compute_units = keep.compute_units;
    stats_.frame_w = keep.frame_w; stats_.frame_h = keep.frame_h;
    stats_.planes = keep.planes; stats_.tiles_per_plane = keep.tiles_per_plane;
    return 0;
}

}  // namespace reve
