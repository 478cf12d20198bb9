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
#include <mutex>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "png.h"

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
    std::vector<uint8_t> file, rgb, out, png;
    int w = 0, h = 0;
    err = read_file(in_path, file);
    if (err.empty()) err = png_decode_rgb8(file, rgb, w, h);
    if (!err.empty()) { err = in_path + ": " + err; return REVE_E_IO; }
    const int s = eng.scale();
    out.resize((size_t)w * s * h * s * 3);
    int rc = eng.upscale_host(rgb.data(), w, h, (ptrdiff_t)w * 3, out.data(), (ptrdiff_t)w * s * 3);
    if (rc != 0) { err = eng.err(); return rc; }
    err = png_encode_rgb8(out.data(), w * s, h * s, (size_t)w * s * 3, 1, png);
    if (err.empty()) err = write_file(out_path, png);
    if (!err.empty()) { err = out_path + ": " + err; return REVE_E_IO; }
    return 0;
}

// Multi-GPU: frames are independent, so frame i simply goes to engine i mod G; every
// engine has its own ring and streams, the calling thread feeds them round-robin.
// Directory mode as a 3-stage pipeline (the binary's own shape is 1 load : 2 proc : 2 save threads,
// SURVEY.md §2.3.1): PNG decode on a small thread pool running a bounded distance ahead, the GPU
// through the engine's submit/wait ring on the calling thread, PNG encode + write on a second pool.
// The progress callback fires on the calling thread, once per frame, in name order, only after
// the frame's file is on disk.
// Frames travel in PINNED host buffers taken from two pools (hipHostMalloc, sized by the first frame): with
// pageable std::vectors every frame paid a first-touch page-fault pass over its 25 MB output and a staged,
// synchronous copy on the feeding thread, which capped the mode at ~150 frames/s.  A frame whose size does
// not fit the pools' buffers (mixed sizes in one directory) falls back to its own pageable vectors.
namespace {
struct Job {
    std::string in_path, out_path;
    std::vector<uint8_t> rgb, out;       // pageable fallback
    uint8_t* in_p = nullptr;             // pinned buffers (from the pools), or nullptr
    uint8_t* out_p = nullptr;
    int w = 0, h = 0;
    std::string err;
    bool decoded = false, encoded = false, submitted = false;
};

// Fixed-size pinned buffers; all state is guarded by the pipeline's one mutex (callers hold it).  Pinning
// 25 MB takes several milliseconds, so the buffers are allocated by a helper thread while the pipeline is
// already running (frames that find the pool still empty use pageable memory).
// Pinned buffers outlive a call: reve upscales a video segment by segment (one directory call each, reve-cli/src/main.rs:249-274)
// and pinning a 25 MB buffer takes ~8 ms, so a 1000-frame segment would spend its first half second allocating while its
// first frames pass through pageable memory.  Buffers of a finished call are parked here (by size) and taken back by the next
// call that asks for the same size; the cache is bounded; what it still holds when the process ends is left to the OS (a
// static destructor would run hipHostFree after the HIP runtime's own teardown).
struct PinnedCache {
    std::mutex mu;
    std::vector<std::pair<size_t, uint8_t*>> parked;
    static constexpr size_t kMaxBytes = (size_t)3 << 30;
    size_t bytes = 0;
    uint8_t* take(size_t cap)
    {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < parked.size(); ++i)
            if (parked[i].first == cap) {
                uint8_t* p = parked[i].second;
                parked[i] = parked.back();
                parked.pop_back();
                bytes -= cap;
                return p;
            }
        return nullptr;
    }
    bool park(size_t cap, uint8_t* p)      // false: cache full, the caller frees the buffer
    {
        std::lock_guard<std::mutex> lk(mu);
        if (bytes + cap > kMaxBytes) return false;
        parked.emplace_back(cap, p);
        bytes += cap;
        return true;
    }
};
PinnedCache& g_pinned_cache = *new PinnedCache;   // never destroyed (see above); stays reachable, so leak checkers stay quiet

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
    void destroy()           // end of the call: park the buffers for the next call (or free them if the cache is full)
    {
        for (uint8_t* p : free_list)
            if (!g_pinned_cache.park(cap, p)) (void)hipHostFree(p);
        free_list.clear();
    }
};
}  // namespace

int upscale_dir(const std::vector<Engine*>& engs, const std::string& in_dir, const std::string& out_dir, reve_progress_cb cb,
                void* user, std::string& err)
{
    const int G = (int)engs.size();
    if (G == 0) { err = "no engine"; return REVE_E_INVALID; }
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

    std::vector<Job> jobs(n);
    for (int i = 0; i < n; ++i) {
        jobs[i].in_path = in_dir + "/" + names[i];
        jobs[i].out_path = out_dir + "/" + names[i].substr(0, names[i].size() - 4) + ".png";
    }
    const int s = engs[0]->scale();
    const int lookahead = 24 * G;
    // codec threads: what the process may use minus the feeding thread and the runtime's own, a quarter of it decoders (a 1080p
    // frame decodes in ~5 ms, its 4K result encodes in 8-20 ms with fastdeflate.cpp), never more busy threads than CPUs
    const int budget = std::max(2, effective_cpus() - 1 - G);
    int n_dec = std::max(1, std::min<int>(8 * G, budget / 4)), n_enc = std::max(1, std::min<int>(32 * G, budget - budget / 4));
    // tuning / diagnosis: REVE_DIR_DEC, REVE_DIR_ENC override the pool sizes, REVE_DIR_STATS=1 prints where the time went
    if (const char* e = std::getenv("REVE_DIR_DEC")) n_dec = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("REVE_DIR_ENC")) n_enc = std::max(1, std::atoi(e));
    const bool stats = std::getenv("REVE_DIR_STATS") && std::getenv("REVE_DIR_STATS")[0] == '1';
    std::atomic<long long> us_dec{0}, us_enc{0}, us_wait_dec{0}, us_wait_buf{0}, us_gpu_wait{0}, us_submit{0}, us_report{0};
    long long n_retire_buf = 0, n_retire_full = 0, n_pageable_in = 0, n_pageable_out = 0;   // (feeding thread only)
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
    std::condition_variable cv_dec, cv_enc, cv_alloc, cv_main;   // decoders / encoders / buffer allocators / the calling thread
    PinnedPool in_pool, out_pool;
    in_pool.limit = std::min(n, lookahead + 4 * G);
    out_pool.limit = std::min(n, std::min(n_enc, 28 * G) + 4 * G);   // encoders at work + ring slots (pinning 25 MB takes ~7 ms, and a short directory needs few)
    int next_decode = 0, consumed = 0;   // decode may run up to `lookahead` frames ahead of `consumed`
    std::deque<int> enc_queue;
    bool stop = false;

    auto decoder = [&] {
        for (;;) {
            int i;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_dec.wait(lk, [&] { return stop || (next_decode < n && next_decode < consumed + lookahead); });
                if (stop || next_decode >= n) return;
                i = next_decode++;
            }
            Job& j = jobs[i];
            const auto td = now();
            // per-thread scratch (file bytes, decoded pixels): no allocation per frame once warm
            static thread_local std::vector<uint8_t> file, rgb;
            std::string e = read_file(j.in_path, file);
            if (e.empty()) e = png_decode_rgb8(file, rgb, j.w, j.h);
            uint8_t* pin = nullptr;
            if (e.empty()) {
                std::lock_guard<std::mutex> lk(mu);
                if (out_pool.cap == 0) {   // first decoded frame: the pools' buffer sizes
                    in_pool.cap = rgb.size();
                    out_pool.cap = rgb.size() * s * s;
                    cv_alloc.notify_all();
                }
                pin = in_pool.get(rgb.size());
            }
            if (e.empty()) {
                if (pin) std::memcpy(pin, rgb.data(), rgb.size());   // hand the frame over in pinned memory (a 6 MB copy on this pool thread)
                else j.rgb = rgb;                                     // no pinned buffer free (yet): the frame travels in its own pageable vector
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!e.empty()) j.err = j.in_path + ": " + e;
                j.in_p = pin;
                j.decoded = true;
            }
            us_dec += us_since(td);
            cv_main.notify_one();
        }
    };
    auto encoder = [&] {
        for (;;) {
            int i;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_enc.wait(lk, [&] { return stop || !enc_queue.empty(); });
                if (enc_queue.empty()) return;
                i = enc_queue.front();
                enc_queue.pop_front();
            }
            Job& j = jobs[i];
            const auto te = now();
            static thread_local std::vector<uint8_t> png;
            std::string e = png_encode_rgb8(j.out_p ? j.out_p : j.out.data(), j.w * s, j.h * s, (size_t)j.w * s * 3, 1, png);
            if (e.empty()) e = write_file(j.out_path, png);
            std::vector<uint8_t>().swap(j.out);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!e.empty()) j.err = j.out_path + ": " + e;
                out_pool.put(j.out_p);
                j.out_p = nullptr;
                j.encoded = true;
            }
            us_enc += us_since(te);
            cv_main.notify_one();       // an output buffer is free again / a frame can be reported
        }
    };
    auto allocator = [&] {   // fills both pools, output buffers first, once the first frame has fixed the sizes
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_alloc.wait(lk, [&] { return stop || out_pool.cap != 0; });
        }
        for (;;) {
            PinnedPool* p;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (stop) return;
                p = !out_pool.complete() && (out_pool.total <= in_pool.total || in_pool.complete()) ? &out_pool
                    : (!in_pool.complete() ? &in_pool : nullptr);
                if (!p) return;
            }
            void* mem = g_pinned_cache.take(p->cap);      // a buffer parked by an earlier call, else a new one
            static const unsigned pin_flags = std::getenv("REVE_DIR_PIN_FLAGS") ? (unsigned)std::atoi(std::getenv("REVE_DIR_PIN_FLAGS")) : (unsigned)hipHostMallocPortable;
            const bool ok = mem || (hipHostMalloc(&mem, p->cap, pin_flags) == hipSuccess && mem);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (ok) { p->put((uint8_t*)mem); p->total++; }
                else p->limit = p->total;   // out of pinnable memory: live with what there is
            }
            cv_main.notify_one();
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < 3; ++t) pool.emplace_back(allocator);   // (pinning is the slow part of hipHostMalloc and runs in parallel)
    for (int t = 0; t < n_dec; ++t) pool.emplace_back(decoder);
    for (int t = 0; t < n_enc; ++t) pool.emplace_back(encoder);

    int first_rc = 0, reported = 0;
    std::vector<std::deque<int>> inflight(G);   // frames on each GPU's ring, in submission order
    auto fail = [&](int rc, const std::string& what) { if (!first_rc) { first_rc = rc; err = what; } };
    auto report_ready = [&](bool wait_all) {   // callbacks in name order, on this thread
        std::unique_lock<std::mutex> lk(mu);
        while (reported < n) {
            Job& j = jobs[reported];
            const bool dead = j.decoded && !j.err.empty() && !j.submitted;
            if (!dead && !j.encoded) {
                if (!wait_all) break;
                cv_main.wait(lk, [&] { return jobs[reported].encoded || (jobs[reported].decoded && !jobs[reported].err.empty() && !jobs[reported].submitted); });
                continue;
            }
            const std::string e = j.err;
            lk.unlock();
            if (!e.empty()) fail(REVE_E_IO, e);
            else if (cb) cb(user, reported, j.in_path.c_str(), j.out_path.c_str());
            lk.lock();
            ++reported;
        }
    };
    std::vector<long long> t_retired;   // (statistics) when each frame left its ring, microseconds since the start of the call
    auto retire_one = [&](int g) {   // engine g's oldest frame leaves its ring and goes to the encoders
        uint64_t id = 0;
        const auto tw = now();
        int rc = engs[g]->wait(&id);
        us_gpu_wait += us_since(tw);
        if (stats) t_retired.push_back(us_since(t_start));
        const int i = inflight[g].front();
        inflight[g].pop_front();
        std::vector<uint8_t>().swap(jobs[i].rgb);
        std::lock_guard<std::mutex> lk(mu);
        in_pool.put(jobs[i].in_p);
        jobs[i].in_p = nullptr;
        if (rc != 0) { jobs[i].err = engs[g]->err(); jobs[i].encoded = true; out_pool.put(jobs[i].out_p); jobs[i].out_p = nullptr; }
        else enc_queue.push_back(i);
        cv_enc.notify_one();
    };

    for (int i = 0; i < n; ++i) {
        Job& j = jobs[i];
        {
            const auto tw = now();
            std::unique_lock<std::mutex> lk(mu);
            cv_main.wait(lk, [&] { return j.decoded; });
            consumed = i + 1;
            us_wait_dec += us_since(tw);
        }
        cv_dec.notify_one();            // the decode window moved on by one frame
        if (!j.err.empty()) {
            std::lock_guard<std::mutex> lk(mu);
            in_pool.put(j.in_p);
            j.in_p = nullptr;
        }
        if (!j.err.empty()) { report_ready(false); continue; }
        const size_t out_bytes = (size_t)j.w * s * j.h * s * 3;
        const int g = i % G;
        Engine& eng = *engs[g];
        // a pinned output buffer: free ones come back from the encoders; while there is none, frames that
        // are still on a GPU ring are retired so that the encoders have something to do
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            if (out_bytes > out_pool.cap || (j.out_p = out_pool.get(out_bytes)) != nullptr) break;
            if (!out_pool.complete() || out_pool.total == 0) break;   // still being allocated: pageable this time
            int busy = -1;
            for (int k = 0; k < G; ++k)
                if (!inflight[k].empty() && (busy < 0 || inflight[k].front() < inflight[busy].front())) busy = k;
            if (busy >= 0) { lk.unlock(); ++n_retire_buf; retire_one(busy); continue; }
            const auto tw = now();
            cv_main.wait(lk);
            us_wait_buf += us_since(tw);
        }
        if (!j.out_p) { j.out.resize(out_bytes); ++n_pageable_out; }
        if (!j.in_p) ++n_pageable_in;
        const uint8_t* src = j.in_p ? j.in_p : j.rgb.data();
        uint8_t* dst = j.out_p ? j.out_p : j.out.data();
        const auto ts = now();
        int rc = eng.submit((uint64_t)i, src, j.w, j.h, (ptrdiff_t)j.w * 3, dst, (ptrdiff_t)j.w * s * 3);
        while (rc == REVE_E_BUSY && !inflight[g].empty()) {   // ring full, or the frame size changed
            ++n_retire_full;
            retire_one(g);
            rc = eng.submit((uint64_t)i, src, j.w, j.h, (ptrdiff_t)j.w * 3, dst, (ptrdiff_t)j.w * s * 3);
        }
        if (rc != 0) {
            std::lock_guard<std::mutex> lk(mu);
            j.err = eng.err();
            j.encoded = true;
            in_pool.put(j.in_p); j.in_p = nullptr;
            out_pool.put(j.out_p); j.out_p = nullptr;
        } else {
            j.submitted = true;
            inflight[g].push_back(i);
        }
        us_submit += us_since(ts);
        const auto tr = now();
        report_ready(false);
        us_report += us_since(tr);
    }
    for (;;) {   // drain in frame order
        int g = -1;
        for (int k = 0; k < G; ++k)
            if (!inflight[k].empty() && (g < 0 || inflight[k].front() < inflight[g].front())) g = k;
        if (g < 0) break;
        retire_one(g);
    }
    report_ready(true);
    {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
    }
    cv_dec.notify_all(); cv_enc.notify_all(); cv_alloc.notify_all();
    for (auto& t : pool) t.join();
    in_pool.destroy();
    out_pool.destroy();
    if (stats)
        std::fprintf(stderr, "[dir] %d CPUs usable; %d frames in %.3f s; %d decode threads busy %.3f s each, %d encode threads busy %.3f s each; "
                     "feeder waited %.3f s for decode, %.3f s for an output buffer, %.3f s for the GPU; %.3f s in submit (incl. ring-full waits), %.3f s reporting; "
                     "frames retired because the ring was full %lld / because no pinned output buffer was free %lld; frames through pageable memory in %lld / out %lld\n",
                     effective_cpus(), n, us_since(t_start) / 1e6, n_dec, us_dec / 1e6 / n_dec, n_enc, us_enc / 1e6 / n_enc,
                     us_wait_dec / 1e6, us_wait_buf / 1e6, us_gpu_wait / 1e6, us_submit / 1e6, us_report / 1e6, n_retire_full, n_retire_buf, n_pageable_in, n_pageable_out);
    if (stats && t_retired.size() >= 20) {
        // steady state: the middle 80 % of the frames; stalls: gaps between consecutive frames leaving the ring above 3x the median
        const size_t a = t_retired.size() / 10, b = t_retired.size() - a;
        std::vector<long long> gaps;
        for (size_t i = a + 1; i < b; ++i) gaps.push_back(t_retired[i] - t_retired[i - 1]);
        std::vector<long long> sorted = gaps;
        std::sort(sorted.begin(), sorted.end());
        const long long med = sorted[sorted.size() / 2];
        long long stall_us = 0, n_stall = 0, worst = 0;
        for (long long g : gaps) { if (g > 3 * med) { stall_us += g - med; ++n_stall; } worst = std::max(worst, g); }
        std::fprintf(stderr, "[dir] first frame off the ring after %.1f ms, last after %.1f ms (call: %.1f ms); middle 80 %%: %.1f frames/s, median gap %.3f ms, "
                     "%lld gaps above 3x the median (worst %.1f ms) cost %.1f ms\n", t_retired.front() / 1e3, t_retired.back() / 1e3, us_since(t_start) / 1e3,
                     (double)(b - a - 1) * 1e6 / (double)(t_retired[b - 1] - t_retired[a]), med / 1e3, n_stall, worst / 1e3, stall_us / 1e3);
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

}  // namespace reve
