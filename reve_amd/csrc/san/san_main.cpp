// Harness of the sanitizer builds (make san; tests/test_sanitizers.py).  Every command returns 0 unless a sanitizer
// stops the process: rejected inputs are the expected outcome for a corpus of truncated and bit-flipped files.
//   san_harness png <dir>           decode every file in <dir>; re-encode and re-decode what decodes
//   san_harness model <dir>         parse every <stem>.param + <stem>.bin pair in <dir>, pack what parses
//   san_harness dir <in> <out> <G> [twice]  the directory pipeline over G fake engines (x2 nearest), checks the outputs
//   san_harness stream <G> <frames> <w> <h>  the same pipeline on raw frames: its own capacity in frames/s
//   san_harness cpulist <root> <bus id>      GPU placement lookup against a fake sysfs tree
#include <dirent.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../dirmode.h"
#include "../engine.h"
#include "../fastdeflate.h"
#include "../hostbind.h"
#include "../model.h"
#include "../png.h"

using namespace reve;

static std::vector<std::string> list(const std::string& d)
{
    std::vector<std::string> v;
    if (DIR* dp = opendir(d.c_str())) {
        while (dirent* e = readdir(dp))
            if (e->d_name[0] != '.') v.push_back(e->d_name);
        closedir(dp);
    }
    std::sort(v.begin(), v.end());
    return v;
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const std::string cmd = argv[1];
    if (cmd == "png") {
        int ok = 0, bad = 0;
        for (const std::string& n : list(argv[2])) {
            std::vector<uint8_t> file, rgb, again, rgb2;
            int w = 0, h = 0, w2 = 0, h2 = 0;
            if (!read_file(std::string(argv[2]) + "/" + n, file).empty()) continue;
            if (!png_decode_rgb8(file, rgb, w, h).empty()) { ++bad; continue; }
            ++ok;
            for (int level : {1, 6}) {
                if (!png_encode_rgb8(rgb.data(), w, h, (size_t)w * 3, level, again).empty()) return 3;
                if (!png_decode_rgb8(again, rgb2, w2, h2).empty() || w2 != w || h2 != h || rgb2 != rgb) return 4;   // codec round trip
            }
        }
        std::printf("png: %d decoded, %d rejected\n", ok, bad);
        return 0;
    }
    if (cmd == "cpus") {          // the codec pools' CPU budget as read from a (fake) /proc and /sys tree
        std::printf("cpus: %d\n", effective_cpus(argv[2]));
        return 0;
    }
    if (cmd == "deflate") {
        // the fast path's deflate encoder on every kind of content and on the sizes around its block and tail limits,
        // each stream inflated by zlib and compared
        uint64_t x = 88172645463325252ull;
        auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 11); };
        const int rounds = std::atoi(argv[2]);
        std::vector<uint8_t> z, back;
        for (int t = 0; t < rounds; ++t) {
            const size_t n = t < 80 ? (size_t)t : (t % 16 == 0 ? rnd() % 2500000 : rnd() % 90000);
            std::vector<uint8_t> b(n);
            for (size_t i = 0; i < n; ++i)
                switch (t % 6) {
                case 0: b[i] = (uint8_t)rnd(); break;
                case 1: b[i] = 0; break;
                case 2: b[i] = (uint8_t)((i % 6) * 40); break;
                case 3: b[i] = (rnd() % 16 == 0 || !i) ? (uint8_t)rnd() : b[i - 1]; break;
                case 4: b[i] = (uint8_t)(rnd() % 4); break;
                default: b[i] = (i >= 5000 && rnd() % 64) ? b[i - 5000] : (uint8_t)rnd(); break;
                }
            const size_t zn = fast_zlib_compress(b.data(), n, z);
            back.resize(n + 1);
            uLongf bn = (uLongf)back.size();
            if (!zn || uncompress(back.data(), &bn, z.data(), (uLong)zn) != Z_OK || bn != n || (n && std::memcmp(back.data(), b.data(), n) != 0)) return 8;
            if (n && fast_adler32(1, b.data(), n) != adler32(1, b.data(), (uInt)n)) return 9;
            if (n && fast_crc32(0, b.data(), n) != (uint32_t)crc32(0, b.data(), (uInt)n)) return 10;
        }
        std::printf("deflate: %d streams round-tripped\n", rounds);
        return 0;
    }
    if (cmd == "model") {
        int ok = 0, bad = 0;
        for (const std::string& n : list(argv[2])) {
            if (n.size() < 7 || n.substr(n.size() - 6) != ".param") continue;
            const std::string stem = n.substr(0, n.size() - 6);
            Model m;
            if (!load_ncnn_files(argv[2], stem, m).empty()) { ++bad; continue; }
            ++ok;
            (void)pack_first(m);
            for (int l = 0; l < m.n_body; ++l) (void)pack_body(m, l);
            (void)pack_last(m, true);
            (void)pack_last(m, false);
        }
        std::printf("model: %d parsed, %d rejected\n", ok, bad);
        return 0;
    }
    if (cmd == "cpulist") {       // a GPU's local CPUs / NUMA node as read from a (fake) sysfs tree: <root> <bus id>
        if (argc < 4) return 2;
        const std::string l = pci_local_cpulist(argv[3], argv[2]);
        std::printf("cpulist: '%s' (%zu CPUs), node %d\n", l.c_str(), parse_cpulist(l).size(), pci_numa_node(argv[3], argv[2]));
        return 0;
    }
    if (cmd == "stream" && argc >= 6) {
        // Capacity of the host pipeline itself: G engines that take no time (REVE_FAKE_ENGINE_NOOP=1) or upscale on the CPU, raw
        // frames — every frame is COPIED into its (pinned) buffer from a template, as a decoder would deliver it, and the sink
        // reads one byte per page of the result — no PNG anywhere.  Prints frames/s: what the feeder / pool / callback machinery
        // can push with N engines before any GPU is the limit.
        const int G = std::atoi(argv[2]), n = std::atoi(argv[3]), w = std::atoi(argv[4]), h = std::atoi(argv[5]);
        std::vector<Engine> engs(G);
        std::vector<Engine*> ptrs;
        EngineConfig ec;
        ec.scale = 2;
        for (int g = 0; g < G; ++g) { ec.device = g; engs[g].init(ec, Model()); ptrs.push_back(&engs[g]); }
        std::vector<uint8_t> tmpl((size_t)w * h * 3);
        for (size_t i = 0; i < tmpl.size(); ++i) tmpl[i] = (uint8_t)(i * 2654435761u >> 24);
        std::atomic<long long> sum{0};
        FrameIO io;
        io.decode = [&](int i, const std::function<uint8_t*(int, int)>& sink) -> std::string {
            uint8_t* p = sink(w, h);
            std::memcpy(p, tmpl.data(), tmpl.size());
            p[0] = (uint8_t)i;
            return "";
        };
        io.encode = [&](int, const uint8_t* rgb, int ow, int oh) -> std::string {
            long long s = 0;
            for (size_t o = 0; o < (size_t)ow * oh * 3; o += 4096) s += rgb[o];
            sum += s;
            return "";
        };
        int last = -1, n_done = 0;
        bool ordered = true;
        std::string err;
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = run_pipeline(ptrs, n, io, [&](int i) { ordered &= i > last; last = i; ++n_done; }, err);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("stream: rc %d, %d engines, %d of %d frames %dx%d in %.3f s = %.0f frames/s (%s)%s%s\n", rc, G, n_done, n, w, h, dt, n_done / dt,
                    ordered ? "in order" : "OUT OF ORDER", err.empty() ? "" : ", first error: ", err.c_str());
        return ordered && n_done == n && rc == 0 ? 0 : 9;
    }
    if (cmd == "dir" && argc >= 5) {
        const int G = std::atoi(argv[4]);
        std::vector<Engine> engs(G);
        std::vector<Engine*> ptrs;
        EngineConfig ec;
        ec.scale = 2;
        for (auto& e : engs) { e.init(ec, Model()); ptrs.push_back(&e); }
        struct Seen { int n = 0, last = -1; bool ordered = true; } seen;   // callbacks come in name order; a frame that failed has none
        std::string err;
        if (argc >= 6 && std::string(argv[5]) == "twice") {   // a first pass parks its pinned buffers; the second one re-uses them
            std::string e0;
            (void)upscale_dir(ptrs, argv[2], argv[3], nullptr, nullptr, e0);
        }
        int rc = upscale_dir(ptrs, argv[2], argv[3],
                             [](void* u, int i, const char*, const char*) { auto* s = (Seen*)u; s->ordered &= (i > s->last); s->last = i; s->n++; },
                             &seen, err);
        // every frame that was decodable must have come out as the 2x nearest upscale of its input
        int checked = 0;
        for (const std::string& n : list(argv[2])) {
            std::vector<uint8_t> fi, fo, a, b;
            int w, h, w2, h2;
            if (!read_file(std::string(argv[2]) + "/" + n, fi).empty() || !png_decode_rgb8(fi, a, w, h).empty()) continue;
            if (!read_file(std::string(argv[3]) + "/" + n, fo).empty() || !png_decode_rgb8(fo, b, w2, h2).empty()) return 5;
            if (w2 != 2 * w || h2 != 2 * h) return 6;
            for (int y = 0; y < h2; y += 3)
                for (int x = 0; x < w2; x += 5)
                    if (std::memcmp(&b[((size_t)y * w2 + x) * 3], &a[((size_t)(y / 2) * w + x / 2) * 3], 3) != 0) return 7;
            ++checked;
        }
        std::printf("dir: rc %d, %d callbacks (%s), %d outputs checked%s%s\n", rc, seen.n, seen.ordered ? "in order" : "OUT OF ORDER", checked,
                    err.empty() ? "" : ", first error: ", err.c_str());
        return seen.ordered ? 0 : 8;
    }
    return 2;
}
