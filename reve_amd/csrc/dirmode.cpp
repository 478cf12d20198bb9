#include "dirmode.h"

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

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
namespace {
struct Job {
    std::string in_path, out_path;
    std::vector<uint8_t> rgb, out;
    int w = 0, h = 0;
    std::string err;
    bool decoded = false, encoded = false, submitted = false;
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
    unsigned hw = std::thread::hardware_concurrency();
    const int n_dec = std::max(1, std::min<int>(8 * G, hw ? hw / 4 : 1)), n_enc = std::max(1, std::min<int>(32 * G, hw ? hw / 2 : 2));

    std::mutex mu;
    std::condition_variable cv;
    int next_decode = 0, consumed = 0;   // decode may run up to `lookahead` frames ahead of `consumed`
    std::deque<int> enc_queue;
    bool stop = false;

    auto decoder = [&] {
        for (;;) {
            int i;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || (next_decode < n && next_decode < consumed + lookahead); });
                if (stop || next_decode >= n) return;
                i = next_decode++;
            }
            Job& j = jobs[i];
            std::vector<uint8_t> file;
            std::string e = read_file(j.in_path, file);
            if (e.empty()) e = png_decode_rgb8(file, j.rgb, j.w, j.h);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!e.empty()) j.err = j.in_path + ": " + e;
                j.decoded = true;
            }
            cv.notify_all();
        }
    };
    auto encoder = [&] {
        for (;;) {
            int i;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !enc_queue.empty(); });
                if (enc_queue.empty()) return;
                i = enc_queue.front();
                enc_queue.pop_front();
            }
            Job& j = jobs[i];
            std::vector<uint8_t> png;
            std::string e = png_encode_rgb8(j.out.data(), j.w * s, j.h * s, (size_t)j.w * s * 3, 1, png);
            if (e.empty()) e = write_file(j.out_path, png);
            std::vector<uint8_t>().swap(j.out);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!e.empty()) j.err = j.out_path + ": " + e;
                j.encoded = true;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
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
                cv.wait(lk, [&] { return jobs[reported].encoded || (jobs[reported].decoded && !jobs[reported].err.empty() && !jobs[reported].submitted); });
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
    auto retire_one = [&](int g) {   // engine g's oldest frame leaves its ring and goes to the encoders
        uint64_t id = 0;
        int rc = engs[g]->wait(&id);
        const int i = inflight[g].front();
        inflight[g].pop_front();
        std::vector<uint8_t>().swap(jobs[i].rgb);
        std::lock_guard<std::mutex> lk(mu);
        if (rc != 0) { jobs[i].err = engs[g]->err(); jobs[i].encoded = true; }
        else enc_queue.push_back(i);
        cv.notify_all();
    };

    for (int i = 0; i < n; ++i) {
        Job& j = jobs[i];
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return j.decoded; });
            consumed = i + 1;
        }
        cv.notify_all();
        if (!j.err.empty()) { report_ready(false); continue; }
        j.out.resize((size_t)j.w * s * j.h * s * 3);
        const int g = i % G;
        Engine& eng = *engs[g];
        int rc = eng.submit((uint64_t)i, j.rgb.data(), j.w, j.h, (ptrdiff_t)j.w * 3, j.out.data(), (ptrdiff_t)j.w * s * 3);
        while (rc == REVE_E_BUSY && !inflight[g].empty()) {   // ring full, or the frame size changed
            retire_one(g);
            rc = eng.submit((uint64_t)i, j.rgb.data(), j.w, j.h, (ptrdiff_t)j.w * 3, j.out.data(), (ptrdiff_t)j.w * s * 3);
        }
        if (rc != 0) {
            std::lock_guard<std::mutex> lk(mu);
            j.err = eng.err();
            j.encoded = true;
        } else {
            j.submitted = true;
            inflight[g].push_back(i);
        }
        report_ready(false);
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
    cv.notify_all();
    for (auto& t : pool) t.join();
    return first_rc;
}

}  // namespace reve
