// Harness of the sanitizer builds (make san; tests/test_sanitizers.py).  Every command returns 0 unless a sanitizer
// stops the process: rejected inputs are the expected outcome for a corpus of truncated and bit-flipped files.
//   san_harness png <dir>           decode every file in <dir>; re-encode and re-decode what decodes
//   san_harness model <dir>         parse every <stem>.param + <stem>.bin pair in <dir>, pack what parses
//   san_harness dir <in> <out> <G> [twice]  the directory pipeline over G fake engines (x2 nearest), checks the outputs
//   san_harness stream <G> <frames> <w> <h>  the same pipeline on raw frames: its own capacity in frames/s
//   san_harness cpulist <root> <bus id>      GPU placement lookup against a fake sysfs tree
//   san_harness inflate <rounds>             the library's inflate (fastinflate.cpp) against zlib's: streams of every block type and
//                                            content written by zlib at every level and strategy and by fastdeflate.cpp, then the
//                                            same streams truncated and with bits flipped — same verdict, same bytes, no overrun
//   san_harness bcast <n>                    the weights broadcast (groupcast.cpp) against the recording RCCL table: call sequence,
//                                            delivery, and every unwinding path under injected failures
//   san_harness group <n> <model dir> <name> reve_create_group over n fake GPUs through the C ABI (capi.cpp, unchanged): same, plus
//                                            "on failure none is left"
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

#include "../../../include/reve_hip.h"
#include "../dirmode.h"
#include "../groupcast.h"
#include "../engine.h"
#include "../fastdeflate.h"
#include "../fastinflate.h"
#include "../hostbind.h"
#include "../model.h"
#include "../png.h"

using namespace reve;

namespace reve { std::vector<std::string> fake_rccl_take_log(int* live_comms, int* live_streams); }      // san/fake_rccl.cpp
extern "C" int reve_fake_live_engines(void);                                                             // san/fake_engine.cpp

// ---- the broadcast's call sequence as it must be for devices devs[0..n) and `bytes` per blob
static std::vector<std::string> expected_bcast_log(const std::vector<int>& devs, size_t bytes)
{
    const int n = (int)devs.size();
    std::vector<std::string> e;
    std::string l = "comm_init_all " + std::to_string(n) + " devs";
    for (int d : devs) l += " " + std::to_string(d);
    e.push_back(l);
    for (int i = 0; i < n; ++i) e.push_back("stream_create on device " + std::to_string(devs[i]));
    e.push_back("group_start");
    for (int i = 0; i < n; ++i)
        e.push_back("broadcast count " + std::to_string(bytes) + " dtype 1 root 0 comm" + std::to_string(i) + " stream" + std::to_string(i) + (i == 0 ? " in-place" : ""));
    e.push_back("group_end");
    for (int i = 0; i < n; ++i) {
        e.push_back("stream_sync stream" + std::to_string(i) + " on device " + std::to_string(devs[i]));
        e.push_back("stream_destroy stream" + std::to_string(i));
    }
    for (int i = 0; i < n; ++i) e.push_back("comm_destroy comm" + std::to_string(i));
    return e;
}

static int count_of(const std::vector<std::string>& log, const std::string& prefix)
{
    int k = 0;
    for (const std::string& l : log) k += l.compare(0, prefix.size(), prefix) == 0;
    return k;
}

// after ANY outcome: nothing used after its release or released twice ("BAD" in the log), no communicator or stream left, as
// many destroys as creations
static int check_released(const std::vector<std::string>& log, int live_c, int live_s, const char* what)
{
    for (const std::string& l : log)
        if (l.find("BAD") != std::string::npos) { std::printf("%s: %s\n", what, l.c_str()); return 21; }
    if (live_c || live_s) { std::printf("%s: %d communicators and %d streams left\n", what, live_c, live_s); return 22; }
    return 0;
}

static const char* const kInjections[] = {"comm_init_all", "stream_create:0", "stream_create:last", "group_start", "broadcast:0", "broadcast:last",
                                          "group_end", "stream_sync:0", "stream_sync:last"};
static std::string injection(const char* spec, int n)
{
    std::string s = spec;
    const size_t p = s.find(":last");
    if (p != std::string::npos) s = s.substr(0, p) + ":" + std::to_string(n - 1);
    return s;
}

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
    setenv("REVE_FAKE_DEVICES", "64", 0);          // (the commands below address their fake engines as devices 0 .. G-1)
    if (cmd == "bcast") {
        const int n = std::atoi(argv[2]);
        const size_t bytes = 1234567;              // (not a multiple of anything: count is BYTES, dtype ncclUint8)
        std::vector<int> devs;
        for (int i = 0; i < n; ++i) devs.push_back(n - 1 - i + 3);          // (not 0..n-1 and not ascending: ordinals are passed through, never assumed)
        std::string err;
        const BcastApi* api = system_bcast_api(err);
        if (!api) return 20;
        auto blobs = [&](std::vector<std::vector<uint8_t>>& store, std::vector<void*>& ptrs) {
            store.assign(n, std::vector<uint8_t>(bytes, 0));
            for (size_t i = 0; i < bytes; ++i) store[0][i] = (uint8_t)(i * 2654435761u >> 24);
            ptrs.clear();
            for (auto& b : store) ptrs.push_back(b.data());
        };
        std::vector<std::vector<uint8_t>> store;
        std::vector<void*> ptrs;
        int lc = 0, ls = 0;
        // 1. the sequence and the delivery
        blobs(store, ptrs);
        unsetenv("REVE_FAKE_RCCL_FAIL");
        std::string e = broadcast_blob(*api, devs, ptrs, bytes);
        std::vector<std::string> log = fake_rccl_take_log(&lc, &ls);
        if (!e.empty()) { std::printf("bcast: %s\n", e.c_str()); return 23; }
        if (log != expected_bcast_log(devs, bytes)) {
            for (const std::string& l : log) std::printf("  got: %s\n", l.c_str());
            return 24;
        }
        if (int rc = check_released(log, lc, ls, "success")) return rc;
        for (int i = 1; i < n; ++i)
            if (store[i] != store[0]) { std::printf("bcast: rank %d did not receive the blob\n", i); return 25; }
        // 2. every unwinding path
        int injected = 0;
        for (const char* spec : kInjections) {
            const std::string inj = injection(spec, n);
            setenv("REVE_FAKE_RCCL_FAIL", inj.c_str(), 1);
            blobs(store, ptrs);
            e = broadcast_blob(*api, devs, ptrs, bytes);
            log = fake_rccl_take_log(&lc, &ls);
            if (e.empty()) { std::printf("bcast: injected %s went unnoticed\n", inj.c_str()); return 26; }
            if (int rc = check_released(log, lc, ls, inj.c_str())) return rc;
            if (count_of(log, "comm_init_all") != 1 || count_of(log, "group_start") > 1) return 27;
            // (a group that was opened is closed, whatever failed inside it; one that failed to open is not "closed")
            if (count_of(log, "group_end") != (inj == "group_start" ? 0 : count_of(log, "group_start"))) { std::printf("bcast: %s left a group open\n", inj.c_str()); return 28; }
            if (inj == "comm_init_all" ? log.size() != 1 : count_of(log, "comm_destroy") != n) return 29;
            if (count_of(log, "stream_destroy") != count_of(log, "stream_create") - (inj.compare(0, 13, "stream_create") == 0 ? 1 : 0)) return 30;
            std::printf("bcast: %-16s -> \"%s\", %zu calls, all released\n", inj.c_str(), e.c_str(), log.size());
            ++injected;
        }
        unsetenv("REVE_FAKE_RCCL_FAIL");
        // 3. arguments that must never reach RCCL
        std::vector<void*> with_null = ptrs;
        with_null.back() = nullptr;
        if (broadcast_blob(*api, devs, with_null, bytes).empty() || broadcast_blob(*api, devs, ptrs, 0).empty() || broadcast_blob(*api, {}, {}, bytes).empty()) return 31;
        if (!fake_rccl_take_log(&lc, &ls).empty()) return 32;
        std::printf("bcast: n = %d ok: %zu calls in order, %d ranks hold the blob, %d injected failures unwound\n", n, expected_bcast_log(devs, bytes).size(), n, injected);
        return 0;
    }
    if (cmd == "group" && argc >= 5) {
        const int n = std::atoi(argv[2]);
        reve_config cfg;
        std::memset(&cfg, 0, sizeof cfg);
        cfg.struct_size = sizeof cfg;
        cfg.scale = 2;
        cfg.model_dir = argv[3];
        cfg.model_name = argv[4];
        std::vector<int> devs;
        for (int i = 0; i < n; ++i) devs.push_back(i);
        std::vector<reve_ctx*> ctx(n, (reve_ctx*)0x1);
        setenv("REVE_FAKE_DEVICES", std::to_string(n).c_str(), 1);
        int lc = 0, ls = 0;
        auto none_left = [&](const char* what) {
            for (reve_ctx* c : ctx)
                if (c) { std::printf("group: %s left a context in out[]\n", what); return false; }
            if (reve_fake_live_engines() != 0) { std::printf("group: %s left %d engines alive\n", what, reve_fake_live_engines()); return false; }
            return true;
        };
        // 1. n distinct devices: one RCCL broadcast, n contexts
        unsetenv("REVE_FAKE_RCCL_FAIL");
        int rc = reve_create_group(&cfg, devs.data(), n, ctx.data());
        std::vector<std::string> log = fake_rccl_take_log(&lc, &ls);
        if (rc != 0) { std::printf("group: create failed: %s\n", reve_last_error(nullptr)); return 40; }
        if (n > 1 && (log != expected_bcast_log(devs, 4096) || check_released(log, lc, ls, "group"))) {
            for (const std::string& l : log) std::printf("  got: %s\n", l.c_str());
            return 41;
        }
        if (n == 1 && !log.empty()) return 42;          // a single context never touches RCCL
        if (reve_fake_live_engines() != n) return 43;
        for (reve_ctx*& c : ctx) { reve_destroy(c); c = (reve_ctx*)0x1; }
        if (reve_fake_live_engines() != 0) return 44;
        if (n > 1) {
            // 2. every failure of the broadcast: REVE_E_HIP, the text names it, nothing is left
            for (const char* spec : kInjections) {
                const std::string inj = injection(spec, n);
                setenv("REVE_FAKE_RCCL_FAIL", inj.c_str(), 1);
                rc = reve_create_group(&cfg, devs.data(), n, ctx.data());
                log = fake_rccl_take_log(&lc, &ls);
                if (rc != REVE_E_HIP || std::string(reve_last_error(nullptr)).find("weights broadcast") == std::string::npos) { std::printf("group: %s -> %d %s\n", inj.c_str(), rc, reve_last_error(nullptr)); return 45; }
                if (check_released(log, lc, ls, inj.c_str()) || !none_left(inj.c_str())) return 46;
                for (reve_ctx*& c : ctx) c = (reve_ctx*)0x1;
            }
            unsetenv("REVE_FAKE_RCCL_FAIL");
            // 3. librccl absent: a group of distinct GPUs FAILS (no silent other path)
            setenv("REVE_FAKE_RCCL_MISSING", "1", 1);
            rc = reve_create_group(&cfg, devs.data(), n, ctx.data());
            unsetenv("REVE_FAKE_RCCL_MISSING");
            if (rc != REVE_E_HIP || !none_left("missing librccl") || !fake_rccl_take_log(&lc, &ls).empty()) return 47;
            for (reve_ctx*& c : ctx) c = (reve_ctx*)0x1;
            // 4. a context that fails to initialise half-way: the ones before it are released, RCCL is never reached
            setenv("REVE_FAKE_INIT_FAILS_ON", std::to_string(n / 2).c_str(), 1);
            rc = reve_create_group(&cfg, devs.data(), n, ctx.data());
            unsetenv("REVE_FAKE_INIT_FAILS_ON");
            if (rc != REVE_E_NOMEM || !none_left("init failure") || !fake_rccl_take_log(&lc, &ls).empty()) return 48;
            for (reve_ctx*& c : ctx) c = (reve_ctx*)0x1;
            // 5. contexts that share a device (and REVE_GROUP_BCAST=peer): device-to-device copies, RCCL untouched; forcing rccl
            // onto a shared device is refused
            std::vector<int> shared(n, 0);
            rc = reve_create_group(&cfg, shared.data(), n, ctx.data());
            if (rc != 0 || !fake_rccl_take_log(&lc, &ls).empty() || reve_fake_live_engines() != n) return 49;
            for (reve_ctx*& c : ctx) { reve_destroy(c); c = (reve_ctx*)0x1; }
            setenv("REVE_GROUP_BCAST", "peer", 1);
            rc = reve_create_group(&cfg, devs.data(), n, ctx.data());
            if (rc != 0 || !fake_rccl_take_log(&lc, &ls).empty()) return 50;
            for (reve_ctx*& c : ctx) { reve_destroy(c); c = (reve_ctx*)0x1; }
            setenv("REVE_GROUP_BCAST", "rccl", 1);
            rc = reve_create_group(&cfg, shared.data(), n, ctx.data());
            unsetenv("REVE_GROUP_BCAST");
            if (rc != REVE_E_INVALID || !none_left("rccl forced onto a shared device")) return 51;
        }
        std::printf("group: n = %d ok\n", n);
        return 0;
    }
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
    if (cmd == "codec" && argc >= 6) {
        // timing aid (san_harness_opt): <raw rgb file> <w> <h> <reps> — PNG encode (fast path) and decode of that frame on one core
        const int w = std::atoi(argv[3]), h = std::atoi(argv[4]), reps = std::atoi(argv[5]);
        std::vector<uint8_t> raw, file, back;
        if (!read_file(argv[2], raw).empty() || raw.size() != (size_t)w * h * 3) return 2;
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double te = 1e30, td = 1e30;
        int bw = 0, bh = 0;
        for (int r = 0; r < reps; ++r) {
            double t0 = now();
            if (!png_encode_rgb8(raw.data(), w, h, (size_t)w * 3, 1, file).empty()) return 3;
            te = std::min(te, now() - t0);
            t0 = now();
            if (!png_decode_rgb8(file, back, bw, bh).empty()) return 4;
            td = std::min(td, now() - t0);
        }
        if (back != raw) return 5;
        std::printf("codec: %dx%d encode %.2f ms, decode %.2f ms, file %.2f MB (ratio %.3f)\n", w, h, te, td, file.size() / 1e6, (double)file.size() / raw.size());
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
            // the same bytes fed ROW BY ROW through the encoder's sliding window (what the PNG fast path does): rows of 1 byte up to
            // more than the window's 64 KB read-ahead, offsets into the output vector — the stream must be byte-identical to the
            // contiguous one, each row asked for exactly once and in order
            if (n) {
                const size_t rb = t % 5 == 0 ? 1 + rnd() % 7 : (t % 5 == 1 ? 60000 + rnd() % 90000 : 1 + rnd() % 12000);
                const size_t rows = n / rb;
                if (rows) {
                    const size_t off = rnd() % 100;
                    std::vector<uint8_t> z2, z1;
                    size_t next = 0;
                    bool in_order = true;
                    const size_t zr = fast_zlib_compress_rows(rows, rb, [&](uint8_t* dst, size_t r0, size_t k) {
                        in_order = in_order && r0 == next && k >= 1 && r0 + k <= rows;
                        next = r0 + k;
                        std::memcpy(dst, b.data() + r0 * rb, k * rb);
                    }, z2, off);
                    const size_t zc = fast_zlib_compress(b.data(), rows * rb, z1);
                    if (!in_order || next != rows || zr != zc || std::memcmp(z2.data() + off, z1.data(), zc) != 0) { std::printf("deflate: rows (%zu x %zu) differ from the contiguous stream\n", rows, rb); return 11; }
                }
            }
            if (n && fast_crc32(0, b.data(), n) != (uint32_t)crc32(0, b.data(), (uInt)n)) return 10;
        }
        // Streams built against the literal-only blocks' SAMPLED histogram (64 bytes of every 256 are looked at): noise first, so that
        // the encoder stops searching for matches, then blocks whose sampled bytes are a few common values and whose unsampled bytes are
        // everything else — the code built from the sample gives those 12 bits each.  The output must still fit the stored-form bound
        // (a fresh vector each time: ASan sees one byte past it) and round-trip.
        for (int variant = 0; variant < 6; ++variant) {
            const size_t head = (size_t)512 << 10, tail = (size_t)(7 - variant % 2) * (512 << 10) + (variant >= 4 ? 12345 : 0);
            const size_t n = head + tail;
            std::vector<uint8_t> b(n);
            for (size_t i = 0; i < head; ++i) b[i] = (uint8_t)rnd();
            for (size_t i = head; i < n; ++i) {
                const size_t ph = (i - head + (variant == 3 ? 17 : 0)) % 256;
                if (ph < 64) { int g = 0; while (g < 15 && rnd() % 2) ++g; b[i] = (uint8_t)g; }
                else b[i] = (uint8_t)(16 + rnd() % 240);
            }
            std::vector<uint8_t> zf;
            const size_t off = variant == 2 ? 57 : 0;
            const size_t zn = variant == 1 ? fast_zlib_compress_rows(n / 256, 256, [&](uint8_t* dst, size_t r0, size_t k) { std::memcpy(dst, b.data() + r0 * 256, k * 256); }, zf, off)
                                           : fast_zlib_compress(b.data(), n, zf, off);
            const size_t nn = variant == 1 ? n / 256 * 256 : n;
            if (!zn || off + zn > zf.size() || zn > nn + nn / 2048 + 4096) { std::printf("deflate: adversarial stream %d: %zu bytes from %zu\n", variant, zn, nn); return 12; }
            back.resize(nn + 1);
            uLongf bn = (uLongf)back.size();
            if (uncompress(back.data(), &bn, zf.data() + off, (uLong)zn) != Z_OK || bn != nn || std::memcmp(back.data(), b.data(), nn) != 0) return 13;
        }
        std::printf("deflate: %d streams round-tripped\n", rounds);
        return 0;
    }
    if (cmd == "inflate") {
        uint64_t x = 0x9E3779B97F4A7C15ull;
        auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 11); };
        const int rounds = std::atoi(argv[2]);
        long valid = 0, damaged = 0, damaged_ok = 0;
        std::vector<uint8_t> z, mine, ref;
        for (int t = 0; t < rounds; ++t) {
            const size_t n = t < 40 ? (size_t)t : (t % 13 == 0 ? 300000 + rnd() % 900000 : rnd() % 70000);
            std::vector<uint8_t> b(n);
            for (size_t i = 0; i < n; ++i)
                switch (t % 7) {
                case 0: b[i] = (uint8_t)rnd(); break;                                           // noise: stored blocks
                case 1: b[i] = 0; break;                                                        // one long run (distance 1)
                case 2: b[i] = (uint8_t)((i % 6) * 40); break;                                  // period 6
                case 3: b[i] = (rnd() % 16 == 0 || !i) ? (uint8_t)rnd() : b[i - 1]; break;
                case 4: b[i] = (uint8_t)(rnd() % 4 + (rnd() % 64 == 0 ? 100 : 0)); break;       // grain: short codes, pairs of literals
                case 5: b[i] = (i >= 5000 && rnd() % 64) ? b[i - 5000] : (uint8_t)rnd(); break;  // far matches
                default: b[i] = (uint8_t)((i % 3 == 0) ? rnd() % 3 : (i >= 3 ? b[i - 3] + (rnd() % 5 == 0) : 7)); break;   // RGB-like, period 3
                }
            // the stream: zlib at a level / strategy picked by the round, or the library's own encoder
            const int how = t % 9;
            size_t zn = 0;
            if (how == 8) {
                zn = fast_zlib_compress(b.data(), n, z);
            } else {
                z_stream zs;
                std::memset(&zs, 0, sizeof zs);
                static const int levels[8] = {0, 1, 6, 9, 1, 6, 9, 4};
                const int strategy = how == 4 ? Z_FIXED : how == 5 ? Z_HUFFMAN_ONLY : how == 6 ? Z_RLE : how == 7 ? Z_FILTERED : Z_DEFAULT_STRATEGY;
                if (deflateInit2(&zs, levels[how], Z_DEFLATED, 9 + (int)(rnd() % 7), 1 + (int)(rnd() % 9), strategy) != Z_OK) return 60;
                z.resize(deflateBound(&zs, (uLong)n) + 64);
                zs.next_in = b.data(); zs.avail_in = (uInt)n;
                zs.next_out = z.data(); zs.avail_out = (uInt)z.size();
                // (a flush in the middle: an empty stored block, several blocks)
                if (n > 1000 && t % 4 == 0) { zs.avail_in = (uInt)(n / 2); if (deflate(&zs, Z_FULL_FLUSH) != Z_OK) return 61; zs.avail_in = (uInt)(n - n / 2); }
                if (deflate(&zs, Z_FINISH) != Z_STREAM_END) return 62;
                zn = zs.total_out;
                deflateEnd(&zs);
            }
            mine.assign(n + 1, 0xAA);                      // (one guard byte behind the destination)
            std::string e = fast_zlib_uncompress(z.data(), zn, mine.data(), n);
            if (!e.empty() || (n && std::memcmp(mine.data(), b.data(), n) != 0) || mine[n] != 0xAA) { std::printf("inflate: round %d (how %d, n %zu): %s\n", t, how, n, e.c_str()); return 63; }
            // wrong expected sizes are errors, not overruns
            if (n && fast_zlib_uncompress(z.data(), zn, mine.data(), n - 1).empty()) return 64;
            mine.resize(n + 2, 0xAA);
            if (fast_zlib_uncompress(z.data(), zn, mine.data(), n + 1).empty() || mine[n + 1] != 0xAA) return 65;
            ++valid;
            // damaged copies: zlib is the referee — what it accepts with the right size and checksum must come out the same here, and
            // what this decoder accepts zlib must accept too
            for (int m = 0; m < 12; ++m) {
                std::vector<uint8_t> d(z.begin(), z.begin() + (long)zn);
                if (m % 3 == 0 && zn > 1) d.resize(rnd() % zn);
                else for (int k = 0; k < 1 + (int)(rnd() % 3); ++k) d[rnd() % d.size()] ^= (uint8_t)(1u << (rnd() % 8));
                mine.assign(n + 1, 0xAA);
                ref.assign(n + 1, 0);
                const bool ok_mine = fast_zlib_uncompress(d.data(), d.size(), mine.data(), n).empty();
                uLongf rn = (uLongf)n;
                const bool ok_ref = uncompress(ref.data(), &rn, d.data(), (uLong)d.size()) == Z_OK && rn == n;
                if (mine[n] != 0xAA) return 66;
                if (ok_mine != ok_ref) { std::printf("inflate: round %d mutation %d: mine %d zlib %d\n", t, m, (int)ok_mine, (int)ok_ref); return 67; }
                if (ok_mine && n && std::memcmp(mine.data(), ref.data(), n) != 0) return 68;
                ++damaged;
                damaged_ok += ok_mine;
            }
        }
        std::printf("inflate: %ld streams decoded, %ld damaged copies judged like zlib (%ld of them still valid)\n", valid, damaged, damaged_ok);
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
