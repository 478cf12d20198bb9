#include "dirmode.h"

#include <dirent.h>
#include <sched.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "alpha.h"
#include "hostbind.h"
#include "png.h"
#include "trace.h"

namespace reve {

static bool has_ext(const std::string& n, const char* ext)
{
    const size_t l = std::strlen(ext);
    if (n.size() < l) return false;
    for (size_t i = 0; i < l; ++i)
        if (std::tolower((unsigned char)n[n.size() - l + i]) != ext[i]) return false;
    return true;
}

// CPUs this process may actually use: its affinity mask, cut down to the CPU-bandwidth quota of its control group when it
// has one (a container that shows 256 CPUs may be allowed 16 of them per 100 ms period).  The codec pools are sized from
// this: a process that runs more busy threads than its quota is frozen as a whole for the rest of each period — the thread
// that feeds the GPU included, so the ring drains and the GPU idles 25-50 ms at a time (22 such freezes per
// 1000 frames on the 16-CPU GPU boxes of this project, 20 % of the wall time).
int effective_cpus(const std::string& root)
{
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
    auto quota_of = [](const std::string& path) -> double {   // CPUs, or 0 when the file sets no limit
        FILE* f = std::fopen(path.c_str(), "r");
        if (!f) return 0;
        char a[64] = {0};
        long long period = 0;
        const int k = std::fscanf(f, "%63s %lld", a, &period);
        std::fclose(f);
        if (k != 2 || period <= 0 || a[0] < '0' || a[0] > '9') return 0;   // "max 100000"
        return (double)std::atoll(a) / (double)period;
    };
    double q = 0;
    auto take = [&](double v) { if (v > 0 && (q == 0 || v < q)) q = v; };
    // cgroup v2: the mount's root (a container sees its own group there) and every level of this process's path
    take(quota_of(root + "/sys/fs/cgroup/cpu.max"));
    if (FILE* f = std::fopen((root + "/proc/self/cgroup").c_str(), "r")) {
        char line[4096];
        while (std::fgets(line, sizeof(line), f)) {
            std::string l(line);
            while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back();
            if (l.compare(0, 3, "0::") != 0) continue;
            std::string path = l.substr(3);
            while (path.size() > 1) {
                take(quota_of(root + "/sys/fs/cgroup" + path + "/cpu.max"));
                const size_t slash = path.find_last_of('/');
                path = slash == std::string::npos || slash == 0 ? "" : path.substr(0, slash);
            }
        }
        std::fclose(f);
    }
    // cgroup v1
    for (const char* dir : {"/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"}) {
        long long quota = -1, period = 0;
        if (FILE* f = std::fopen((root + dir + "/cpu.cfs_quota_us").c_str(), "r")) { if (std::fscanf(f, "%lld", &quota) != 1) quota = -1; std::fclose(f); }
        if (FILE* f = std::fopen((root + dir + "/cpu.cfs_period_us").c_str(), "r")) { if (std::fscanf(f, "%lld", &period) != 1) period = 0; std::fclose(f); }
        if (quota > 0 && period > 0) take((double)quota / (double)period);
    }
    if (q > 0 && q < n) n = std::max(1, (int)q);
    return n;
}

int upscale_file(Engine& eng, const std::string& in_path, const std::string& out_path, std::string& err)
{
    std::vector<uint8_t> file, rgb, alpha, out, png;
    int w = 0, h = 0;
    err = read_file(in_path, file);
    if (err.empty()) err = png_decode_rgba8(file, rgb, alpha, w, h);
    if (!err.empty()) { err = in_path + ": " + err; return REVE_E_IO; }
    const int s = eng.scale();
    out.resize((size_t)w * s * h * s * 3);
    int rc = eng.upscale_host(rgb.data(), w, h, (ptrdiff_t)w * 3, out.data(), (ptrdiff_t)w * s * 3);
    if (rc != 0) { err = eng.err(); return rc; }
    if (!alpha.empty()) {
        // an image with transparency (the GUI's single-file call, commands.rs:52-65): the network sees RGB, the alpha plane is scaled
        // beside it as the binary does — bicubic, alpha.h — and the result is an RGBA file
        std::vector<uint8_t> a_out((size_t)w * s * h * s);
        alpha_bicubic(alpha.data(), w, h, s, a_out.data());
        err = png_encode_rgba8(out.data(), a_out.data(), w * s, h * s, png);
    } else
        err = png_encode_rgb8(out.data(), w * s, h * s, (size_t)w * s * 3, 1, png);
    if (err.empty()) err = write_file(out_path, png);
    if (!err.empty()) { err = out_path + ": " + err; return REVE_E_IO; }
    return 0;
}

// The host pipeline behind directory mode and the raw-frame stream entry point.
//
// Multi-GPU: frames are independent, so frame i simply goes to engine i mod G.  Shape (the binary's own is
// 1 load : 2 proc : 2 save threads):
//   decode pool  -> frame i into a pinned buffer of lane i mod G, a bounded distance ahead of the feeders;
//   lane g       =  engine g's FEEDER thread: its own submit/wait ring (hipMemcpyAsync H2D | kernel chain | D2H on the
//                   engine's three streams), its own retire loop and its own pinned buffer pools, allocated by a helper
//                   thread of the lane.  Feeder and allocator bind themselves to the CPUs next to their GPU
//                   (sysfs local_cpulist of its PCI device, hostbind.cpp) before the first pinned allocation, so the pinned
//                   pages and the thread that issues the copies sit on the GPU's NUMA node (SURVEY.md §8e partitioning row);
//   encode pool  <- retired frames from every lane;
//   the CALLER's thread only reports: the progress callback fires there, once per frame, in frame order, after the frame has
//                   been written.
// Round 2 fed all G rings from the calling thread (submit -> wait on engine g's oldest frame blocked while other GPUs idled).
// Frames travel in PINNED host buffers (hipHostMalloc, sized by the first frame): with pageable std::vectors every frame paid
// a first-touch page-fault pass over its 25 MB output and a staged, synchronous copy on the feeding thread, which capped the
// mode at ~150 frames/s.  A frame whose size does not fit its lane's buffers (mixed sizes in one directory) travels in
// pageable vectors of its own.
namespace {
struct Job {
    std::vector<uint8_t> rgb, out;       // pageable fallback
    uint8_t* in_p = nullptr;             // pinned buffers (from the lane's pools), or nullptr
    uint8_t* out_p = nullptr;
    int w = 0, h = 0;
    std::string err;
    int rc = 0;                          // REVE_E_* of a failed frame (REVE_E_IO for the codecs, the engine's code for the GPU side)
    bool decoded = false, encoded = false, submitted = false;
};

// Pinned buffers outlive a call: reve upscales a video segment by segment (one call each, reve-cli/src/main.rs:249-274) and
// pinning a 25 MB buffer takes ~8 ms, so a 1000-frame segment would spend its first half second allocating while its first
// frames pass through pageable memory.  Buffers of a finished call are parked here, keyed by size and NUMA node, and taken
// back by the next call that asks for the same; a call that works with other sizes evicts what it cannot use (a long-lived
// host with varying frame sizes used to collect up to 3 GiB of dead page-locked memory); the total is bounded; pinned_cache_trim()
// — the last reve_destroy calls it — frees everything.  What the cache still holds when the process ends is left to the OS
// (a static destructor would run hipHostFree after the HIP runtime's own teardown).
// every hipHostFree of the pipeline: like the allocations, not while another context captures its chain into a graph
static void pinned_free(void* p)
{
    std::lock_guard<std::mutex> ulk(unsafe_calls_mutex());
    (void)hipHostFree(p);
}

struct PinnedCache {
    struct Entry { size_t cap; int node; uint8_t* p; };
    std::mutex mu;
    std::vector<Entry> parked;
    static constexpr size_t kMaxBytes = (size_t)6 << 30;      // 8 lanes x (28 x 6 MB + 18 x 25 MB) = 5 GB at 1080p x2
    size_t bytes = 0;
    uint8_t* take(size_t cap, int node)
    {
        std::lock_guard<std::mutex> lk(mu);
        int best = -1;
        for (size_t i = 0; i < parked.size(); ++i)
            if (parked[i].cap == cap && (best < 0 || parked[i].node == node)) { best = (int)i; if (parked[i].node == node) break; }
        if (best < 0) return nullptr;
        uint8_t* p = parked[best].p;
        parked[best] = parked.back();
        parked.pop_back();
        bytes -= cap;
        return p;
    }
    bool park(size_t cap, int node, uint8_t* p)      // false: cache full, the caller frees the buffer
    {
        std::lock_guard<std::mutex> lk(mu);
        if (bytes + cap > kMaxBytes) return false;
        parked.push_back({cap, node, p});
        bytes += cap;
        return true;
    }
    void evict_except(size_t cap_a, size_t cap_b)      // a call that uses these two sizes: every other parked buffer is dead weight
    {
        std::vector<uint8_t*> dead;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < parked.size();) {
                if (parked[i].cap != cap_a && parked[i].cap != cap_b) {
                    dead.push_back(parked[i].p);
                    bytes -= parked[i].cap;
                    parked[i] = parked.back();
                    parked.pop_back();
                } else ++i;
            }
        }
        for (uint8_t* p : dead) pinned_free(p);
    }
    size_t trim()
    {
        std::vector<Entry> all;
        {
            std::lock_guard<std::mutex> lk(mu);
            all.swap(parked);
            bytes = 0;
        }
        size_t n = 0;
        for (const Entry& e : all) { pinned_free(e.p); n += e.cap; }
        return n;
    }
};
PinnedCache& g_pinned_cache = *new PinnedCache;   // never destroyed (see above); stays reachable, so leak checkers stay quiet

// Fixed-size pinned buffers of one lane; all state is guarded by the pipeline's one mutex (callers hold it).
struct PinnedPool {
    size_t cap = 0;          // buffer size, fixed by the first frame
    int total = 0, limit = 0;
    std::vector<uint8_t*> free_list;
    uint8_t* get(size_t bytes)   // nullptr: nothing free right now, or the request does not fit
    {
        if (bytes > cap || free_list.empty()) return nullptr;
        uint8_t* p = free_list.back();
        free_list.pop_back();
        return p;
    }
    bool complete() const { return total >= limit; }   // no more buffers will appear by allocation
    void put(uint8_t* p) { if (p) free_list.push_back(p); }
    void destroy(int node)   // end of the call: park the buffers for the next call (or free them if the cache is full)
    {
        for (uint8_t* p : free_list)
            if (!g_pinned_cache.park(cap, node, p)) pinned_free(p);
        free_list.clear();
    }
};

// One GPU's share of the pipeline.
struct Lane {
    Engine* eng = nullptr;
    int node = -1;                       // NUMA node of the GPU (-1 unknown)
    std::string cpulist;                 // CPUs next to it ("" unknown: threads stay where the scheduler puts them)
    PinnedPool in_pool, out_pool;
    std::deque<int> inflight;            // frames on this GPU's ring, in submission order (feeder thread only)
    int next = 0;                        // the frame this feeder takes next (g, g + G, ...); > n when done
    std::condition_variable cv;          // wakes this lane's feeder: its next frame was decoded / an output buffer came back
    long long us_wait_dec = 0, us_wait_buf = 0, us_gpu_wait = 0, us_submit = 0;
    long long n_retire_buf = 0, n_retire_full = 0, n_pageable_in = 0, n_pageable_out = 0;
    int bound_cpus = 0;
};
}  // namespace

size_t pinned_cache_trim() { return g_pinned_cache.trim(); }

int run_pipeline(const std::vector<Engine*>& engs, int n, const FrameIO& io, const std::function<void(int)>& on_done, std::string& err)
{
    const int G = (int)engs.size();
    if (G == 0) { err = "no engine"; return REVE_E_INVALID; }
    if (n <= 0) return 0;
    std::vector<Job> jobs(n);
    const int s = engs[0]->scale();
    const int lookahead = 24 * G;
    // codec threads: what the process may use minus the feeders and the runtime's own, never more busy threads than CPUs (a
    // feeder sleeps in its ring's blocking event most of the time: half a CPU each).  ONE pool (round 5): a thread takes whatever
    // there is, an encode job first (it frees a pinned output buffer the GPU side is waiting for), else the next frame to decode.
    // The split between the two kinds of work therefore follows the content by itself — flat frames want 1 ms of decoding per
    // 10 ms of encoding, decoded video 6 per 16, raw frames 6 MB in per 25 MB out — where rounds 2-4 fixed it at three threads in ten.
    const int budget = std::max(2, effective_cpus() - 1 - (G + 1) / 2);
    int n_codec = std::max(2, std::min<int>(40 * G, budget));
    int n_dec = std::min<int>(8 * G, n_codec), n_enc = n_codec;          // at most this many decoding / encoding at once
    // tuning / diagnosis: REVE_DIR_THREADS sets the pool size, REVE_DIR_DEC / REVE_DIR_ENC cap how many of its threads decode / encode at
    // a time, REVE_DIR_STATS=1 prints where the time went, REVE_DIR_BIND=0 leaves the lanes' threads unbound
    if (const char* e = std::getenv("REVE_DIR_THREADS")) { n_codec = std::max(1, std::atoi(e)); n_dec = std::min(n_dec, n_codec); n_enc = n_codec; }
    if (const char* e = std::getenv("REVE_DIR_DEC")) n_dec = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("REVE_DIR_ENC")) n_enc = std::max(1, std::atoi(e));
    n_codec = std::max(n_codec, std::max(n_dec, n_enc));
    const bool stats = std::getenv("REVE_DIR_STATS") && std::getenv("REVE_DIR_STATS")[0] == '1';
    const bool bind = !(std::getenv("REVE_DIR_BIND") && std::getenv("REVE_DIR_BIND")[0] == '0');
    std::atomic<long long> us_dec{0}, us_enc{0};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us_since = [](std::chrono::steady_clock::time_point t) {
        return (long long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t).count();
    };
    const auto t_start = now();
    std::vector<bool> was_profiling;
    if (stats)
        for (Engine* e : engs) { was_profiling.push_back(e->profiling()); e->set_profiling(true); e->reset_stats(); }

    std::mutex mu;
    // One condition variable per kind of waiter: with a single one every finished decode or encode woke all ~70 pool threads, which
    // then queued on the mutex in front of the feeding thread (64 + 8 codec threads: 318 frames/s, 110 + 16: 232, 24 + 4: 340).
    std::condition_variable cv_codec, cv_alloc, cv_main;         // the codec pool / buffer allocators / the calling thread
    int active_dec = 0, active_enc = 0;                           // pool threads decoding / encoding right now
    std::vector<Lane> lanes(G);
    for (int g = 0; g < G; ++g) {
        Lane& L = lanes[g];
        L.eng = engs[g];
        L.next = g;
        const std::string bus = engs[g]->pci_bus_id();
        L.cpulist = pci_local_cpulist(bus);
        L.node = pci_numa_node(bus);
        const int mine = (n - g + G - 1) / G;                                   // frames of this lane
        L.in_pool.limit = std::min(mine, lookahead / G + 4);
        L.out_pool.limit = std::min(mine, std::min(n_enc, 28 * G) / G + 4);     // encoders at work + ring slots (pinning 25 MB takes ~7 ms, and a short run needs few)
    }
    size_t cap_in = 0, cap_out = 0;      // buffer sizes, fixed by the first decoded frame
    int next_decode = 0;                 // decode may run up to `lookahead` frames ahead of the slowest feeder
    auto consumed = [&] { int m = n; for (const Lane& L : lanes) m = std::min(m, L.next); return m; };
    std::deque<int> enc_queue;
    bool stop = false;
    std::vector<long long> t_retired;    // (statistics) when each frame left its ring, microseconds since the start of the call

    auto do_decode = [&](int i) {
        {
            Job& j = jobs[i];
            Lane& L = lanes[i % G];
            const auto td = now();
            // where the frame's pixels go: a pinned buffer of the lane if one is free and fits, else the job's own vector
            uint8_t* pin = nullptr;
            auto sink = [&](int w, int h) -> uint8_t* {
                const size_t bytes = (size_t)w * h * 3;
                j.w = w; j.h = h;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (cap_out == 0) {   // first frame: the pools' buffer sizes
                        cap_in = bytes; cap_out = bytes * s * s;
                        for (Lane& l : lanes) { l.in_pool.cap = cap_in; l.out_pool.cap = cap_out; }
                        cv_alloc.notify_all();
                    }
                    pin = L.in_pool.get(bytes);
                }
                if (pin) return pin;
                j.rgb.resize(bytes);
                return j.rgb.data();
            };
            std::string e;
            {
                TraceRange tr("reve:decode");
                e = io.decode(i, sink);
            }
            if (e.empty() && (j.w <= 0 || j.h <= 0)) e = "frame source delivered no pixels";
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!e.empty()) { j.err = e; j.rc = REVE_E_IO; }
                j.in_p = pin;
                j.decoded = true;
            }
            us_dec += us_since(td);
            L.cv.notify_one();
            cv_main.notify_one();       // (a frame that failed to decode is reported without passing a lane)
        }
    };
    auto do_encode = [&](int i) {
        {
            Job& j = jobs[i];
            Lane& L = lanes[i % G];
            const auto te = now();
            std::string e;
            {
                TraceRange tr("reve:encode");
                e = io.encode(i, j.out_p ? j.out_p : j.out.data(), j.w * s, j.h * s);
            }
            std::vector<uint8_t>().swap(j.out);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!e.empty()) { j.err = e; j.rc = REVE_E_IO; }
                L.out_pool.put(j.out_p);
                j.out_p = nullptr;
                j.encoded = true;
            }
            us_enc += us_since(te);
            L.cv.notify_one();          // an output buffer of this lane is free again
            cv_main.notify_one();       // a frame can be reported
        }
    };
    // a thread of the codec pool: encode what is queued, else decode the next frame inside the window, else sleep
    auto worker = [&] {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            const bool can_enc = !enc_queue.empty() && active_enc < n_enc;
            const bool can_dec = !stop && next_decode < n && next_decode < consumed() + lookahead && active_dec < n_dec;
            if (can_enc) {
                const int i = enc_queue.front();
                enc_queue.pop_front();
                ++active_enc;
                lk.unlock();
                do_encode(i);
                lk.lock();
                --active_enc;
            } else if (can_dec) {
                const int i = next_decode++;
                ++active_dec;
                lk.unlock();
                do_decode(i);
                lk.lock();
                --active_dec;
            } else {
                if (stop && enc_queue.empty()) return;
                cv_codec.wait(lk);
            }
        }
    };
    // fills both pools of ONE lane, output buffers first, once the first frame has fixed the sizes; bound to the GPU's CPUs so
    // that the pinned pages are first touched (and therefore placed) on its NUMA node
    auto allocator = [&](int g) {
        Lane& L = lanes[g];
        if (bind) (void)bind_this_thread(L.cpulist);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_alloc.wait(lk, [&] { return stop || cap_out != 0; });
            if (stop) return;
        }
        g_pinned_cache.evict_except(cap_in, cap_out);
        for (;;) {
            PinnedPool* p;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (stop) return;
                p = !L.out_pool.complete() && (L.out_pool.total <= L.in_pool.total || L.in_pool.complete()) ? &L.out_pool
                    : (!L.in_pool.complete() ? &L.in_pool : nullptr);
                if (!p) return;
            }
            void* mem = g_pinned_cache.take(p->cap, L.node);      // a buffer parked by an earlier call, else a new one
            static const unsigned pin_flags = std::getenv("REVE_DIR_PIN_FLAGS") ? (unsigned)std::atoi(std::getenv("REVE_DIR_PIN_FLAGS")) : (unsigned)hipHostMallocPortable;
            bool ok = mem != nullptr;
            if (!ok) {
                std::lock_guard<std::mutex> ulk(unsafe_calls_mutex());      // (not while a context captures its chain into a graph)
                ok = hipHostMalloc(&mem, p->cap, pin_flags) == hipSuccess && mem;
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (ok) { p->put((uint8_t*)mem); p->total++; }
                else p->limit = p->total;   // out of pinnable memory: live with what there is
            }
            L.cv.notify_one();
        }
    };

    // engine g's oldest frame leaves its ring and goes to the encoders (lane g's feeder only)
    auto retire_one = [&](int g) {
        Lane& L = lanes[g];
        uint64_t id = 0;
        const auto tw = now();
        const int rc = L.eng->wait(&id);
        L.us_gpu_wait += us_since(tw);
        const int i = L.inflight.front();
        L.inflight.pop_front();
        std::vector<uint8_t>().swap(jobs[i].rgb);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (stats) t_retired.push_back(us_since(t_start));
            L.in_pool.put(jobs[i].in_p);
            jobs[i].in_p = nullptr;
            if (rc != 0) {
                jobs[i].err = L.eng->err(); jobs[i].rc = rc; jobs[i].encoded = true;
                L.out_pool.put(jobs[i].out_p); jobs[i].out_p = nullptr;
                cv_main.notify_one();
            } else {
                enc_queue.push_back(i);
            }
        }
        cv_codec.notify_one();
    };
    auto feeder = [&](int g) {
        Lane& L = lanes[g];
        if (bind) L.bound_cpus = bind_this_thread(L.cpulist);
        Engine& eng = *L.eng;
        for (int i = g; i < n; i += G) {
            Job& j = jobs[i];
            {
                const auto tw = now();
                std::unique_lock<std::mutex> lk(mu);
                // while the next frame is still being decoded, frames already on the ring are retired: the encoders get work and
                // the ring's slots free up, instead of the lane sleeping on a frame that is not there yet
                while (!j.decoded) {
                    if (!L.inflight.empty()) { lk.unlock(); retire_one(g); lk.lock(); continue; }
                    L.cv.wait(lk);
                }
                L.next = i + G;
                L.us_wait_dec += us_since(tw);
            }
            cv_codec.notify_one();          // the decode window moved on
            if (!j.err.empty()) {           // undecodable: nothing to submit; reported by the caller's thread in its turn
                std::lock_guard<std::mutex> lk(mu);
                L.in_pool.put(j.in_p);
                j.in_p = nullptr;
                cv_main.notify_one();
                continue;
            }
            const size_t out_bytes = (size_t)j.w * s * j.h * s * 3;
            // a pinned output buffer: free ones come back from the encoders; while there is none, frames that
            // are still on this GPU's ring are retired so that the encoders have something to do
            for (;;) {
                std::unique_lock<std::mutex> lk(mu);
                if (out_bytes > L.out_pool.cap || (j.out_p = L.out_pool.get(out_bytes)) != nullptr) break;
                if (!L.out_pool.complete() || L.out_pool.total == 0) break;   // still being allocated: pageable this time
                if (!L.inflight.empty()) { lk.unlock(); ++L.n_retire_buf; retire_one(g); continue; }
                const auto tw = now();
                L.cv.wait(lk);
                L.us_wait_buf += us_since(tw);
            }
            if (!j.out_p) { j.out.resize(out_bytes); ++L.n_pageable_out; }
            if (!j.in_p) ++L.n_pageable_in;
            const uint8_t* src = j.in_p ? j.in_p : j.rgb.data();
            uint8_t* dst = j.out_p ? j.out_p : j.out.data();
            const auto ts = now();
            int rc = eng.submit((uint64_t)i, src, j.w, j.h, (ptrdiff_t)j.w * 3, dst, (ptrdiff_t)j.w * s * 3);
            while (rc == REVE_E_BUSY && !L.inflight.empty()) {   // ring full, or the frame size changed
                ++L.n_retire_full;
                retire_one(g);
                rc = eng.submit((uint64_t)i, src, j.w, j.h, (ptrdiff_t)j.w * 3, dst, (ptrdiff_t)j.w * s * 3);
            }
            if (rc != 0) {
                std::lock_guard<std::mutex> lk(mu);
                j.err = eng.err(); j.rc = rc;
                j.encoded = true;
                L.in_pool.put(j.in_p); j.in_p = nullptr;
                L.out_pool.put(j.out_p); j.out_p = nullptr;
                cv_main.notify_one();
            } else {
                j.submitted = true;
                L.inflight.push_back(i);
            }
            L.us_submit += us_since(ts);
        }
        while (!L.inflight.empty()) retire_one(g);
        {
            std::lock_guard<std::mutex> lk(mu);
            L.next = n + G;                 // this lane no longer holds the decode window back
        }
        cv_codec.notify_all();
    };

    std::vector<std::thread> pool;
    for (int g = 0; g < G; ++g) pool.emplace_back(allocator, g);
    for (int t = 0; t < n_codec; ++t) pool.emplace_back(worker);
    for (int g = 0; g < G; ++g) pool.emplace_back(feeder, g);

    // ---- the caller's thread: callbacks in frame order
    int first_rc = 0;
    {
        std::unique_lock<std::mutex> lk(mu);
        for (int i = 0; i < n; ++i) {
            Job& j = jobs[i];
            cv_main.wait(lk, [&] { return j.encoded || (j.decoded && !j.err.empty() && !j.submitted); });
            if (!j.err.empty()) {
                if (!first_rc) { first_rc = j.rc ? j.rc : REVE_E_IO; err = j.err; }
            } else if (on_done) {
                lk.unlock();
                on_done(i);
                lk.lock();
            }
        }
        // (every frame is accounted for: the feeders have retired everything they submitted)
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
    }
    cv_codec.notify_all(); cv_alloc.notify_all();
    for (Lane& L : lanes) L.cv.notify_all();
    for (auto& t : pool) t.join();
    for (Lane& L : lanes) { L.in_pool.destroy(L.node); L.out_pool.destroy(L.node); }

    if (stats) {
        std::fprintf(stderr, "[dir] %d CPUs usable; %d frames in %.3f s on %d lane(s); %d codec threads: decoding %.3f ms, encoding %.3f ms of one CPU per frame, "
                     "%.3f s busy each\n", effective_cpus(), n, us_since(t_start) / 1e6, G, n_codec, us_dec / 1e3 / n, us_enc / 1e3 / n, (us_dec + us_enc) / 1e6 / n_codec);
        for (int g = 0; g < G; ++g) {
            const Lane& L = lanes[g];
            std::fprintf(stderr, "[dir] lane %d (numa node %d, feeder bound to %d CPUs%s%s): waited %.3f s for decode, %.3f s for an output buffer, %.3f s for the GPU; "
                         "%.3f s in submit (incl. ring-full waits); frames retired because the ring was full %lld / because no pinned output buffer was free %lld; "
                         "frames through pageable memory in %lld / out %lld\n", g, L.node, L.bound_cpus, L.cpulist.empty() ? "" : ": ", L.cpulist.c_str(),
                         L.us_wait_dec / 1e6, L.us_wait_buf / 1e6, L.us_gpu_wait / 1e6, L.us_submit / 1e6, L.n_retire_full, L.n_retire_buf, L.n_pageable_in, L.n_pageable_out);
        }
    }
    if (stats && t_retired.size() >= 20) {
        // steady state: the middle 80 % of the frames; stalls: gaps between consecutive frames leaving a ring above 3x the median
        std::sort(t_retired.begin(), t_retired.end());
        const size_t a = t_retired.size() / 10, b = t_retired.size() - a;
        std::vector<long long> gaps;
        for (size_t i = a + 1; i < b; ++i) gaps.push_back(t_retired[i] - t_retired[i - 1]);
        std::vector<long long> sorted = gaps;
        std::sort(sorted.begin(), sorted.end());
        const long long med = sorted[sorted.size() / 2];
        long long stall_us = 0, n_stall = 0, worst = 0;
        for (long long g : gaps) { if (g > 3 * med && g > 100) { stall_us += g - med; ++n_stall; } worst = std::max(worst, g); }
        std::fprintf(stderr, "[dir] first frame off a ring after %.1f ms, last after %.1f ms (call: %.1f ms); middle 80 %%: %.1f frames/s, median gap %.3f ms, "
                     "%lld gaps above 3x the median (worst %.1f ms) cost %.1f ms\n", t_retired.front() / 1e3, t_retired.back() / 1e3, us_since(t_start) / 1e3,
                     (double)(b - a - 1) * 1e6 / (double)std::max<long long>(1, t_retired[b - 1] - t_retired[a]), med / 1e3, n_stall, worst / 1e3, stall_us / 1e3);
    }
    if (stats)
        for (int g = 0; g < G; ++g) {
            Stats st;
            engs[g]->get_stats(st);
            const double f = st.ring_frames ? (double)st.ring_frames : 1.0;
            std::fprintf(stderr, "[dir] gpu %d: %llu frames through the ring, device time per frame: upload %.3f ms, chain %.3f ms, download %.3f ms; "
                         "ring wall %.3f ms per frame\n", g, (unsigned long long)st.ring_frames, st.h2d_ms_total / f, st.chain_ms_total / f,
                         st.d2h_ms_total / f, st.ring_wall_ms / f);
            engs[g]->set_profiling(was_profiling[g]);
        }
    return first_rc;
}

// Directory mode = the pipeline with PNG files at both ends.
int upscale_dir(const std::vector<Engine*>& engs, const std::string& in_dir, const std::string& out_dir, reve_progress_cb cb,
                void* user, std::string& err)
{
    if (engs.empty()) { err = "no engine"; return REVE_E_INVALID; }
    DIR* d = opendir(in_dir.c_str());
    if (!d) { err = "cannot open directory " + in_dir; return REVE_E_IO; }
    std::vector<std::string> names;
    while (dirent* e = readdir(d)) {
        std::string n = e->d_name;
        if (has_ext(n, ".png")) names.push_back(n);
    }
    closedir(d);
    std::sort(names.begin(), names.end());
    struct stat st;
    if (stat(out_dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) { err = "output directory missing: " + out_dir; return REVE_E_IO; }
    const int n = (int)names.size();
    if (n == 0) return 0;
    std::vector<std::string> in_path(n), out_path(n);
    for (int i = 0; i < n; ++i) {
        in_path[i] = in_dir + "/" + names[i];
        out_path[i] = out_dir + "/" + names[i].substr(0, names[i].size() - 4) + ".png";
    }
    FrameIO io;
    io.decode = [&](int i, const std::function<uint8_t*(int, int)>& sink) -> std::string {
        // per-thread scratch (file bytes, decoded pixels): no allocation per frame once warm
        static thread_local std::vector<uint8_t> file;
        int w = 0, h = 0;
        std::string e = read_file(in_path[i], file);
        if (e.empty()) e = png_decode_rgb8_to(file, sink, w, h);   // scanlines are un-filtered straight into the lane's pinned buffer
        if (!e.empty()) return in_path[i] + ": " + e;
        return "";
    };
    io.encode = [&](int i, const uint8_t* rgb, int w, int h) -> std::string {
        static thread_local std::vector<uint8_t> png;
        std::string e = png_encode_rgb8(rgb, w, h, (size_t)w * 3, 1, png);
        if (e.empty()) e = write_file(out_path[i], png);
        return e.empty() ? "" : out_path[i] + ": " + e;
    };
    return run_pipeline(engs, n, io, [&](int i) { if (cb) cb(user, i, in_path[i].c_str(), out_path[i].c_str()); }, err);
}

}  // namespace reve
