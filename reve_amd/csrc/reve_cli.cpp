// reve — resumable segment scheduler around the in-process upscaler.
//
// Re-creates reve-cli's surface and semantics above libreve_hip.so instead of a child process:
//   flags            reve-shared/src/lib.rs:209-247  (-i/--inputpath, <outputpath>, -s/--scale 2..4,
//                    -S/--segmentsize=1000 [README documents -P: both accepted], -c/--crf=15,
//                    -p/--preset=slow, -x/--x265params) and their validators (lib.rs:249-280)
//   state files      temp/args.temp, temp/video.temp (JSON, same keys; reve-cli/src/main.rs:112-121)
//   segmentation     Video::new (lib.rs:59-86) — without its off-by-one: every frame exactly once
//   resume           main.rs:43-102,142-159: segment-granular; a segment leaves video.temp only after
//                    its part file is complete, the first remaining segment's partial part is deleted
//   3-stage overlap  main.rs:218-345: export(i+1) || upscale(i) || merge(i-1)
//   tools            ffmpeg / mediainfo command lines of lib.rs:30-50,100-119,181-204 and
//                    main.rs:306-326 (verbatim options, portable paths)
// Deliberate divergences from the reference are the SURVEY.md §9.1 items (A-F, I).
// Worker threads (export / merge) never leave the process: they hand a status back, the main thread joins
// them, destroys the GPU contexts and only then exits non-zero with the state files kept.
#include <fcntl.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <deque>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/reve_hip.h"

extern char** environ;

namespace {

struct Args {
    std::string inputpath, outputpath, preset = "slow", x265params = "psy-rd=2:aq-strength=1:deblock=0,0:bframes=8";
    int scale = 0, segmentsize = 1000, crf = 15;
};
struct Segment { int index, size; };
struct Video {
    std::string path, output_path;
    std::vector<Segment> segments;   // segments still to do
    double frame_rate = 0;
    int frame_count = 0, segment_size = 0, segment_count = 0, upscale_ratio = 0;
    // not persisted: the container's exact rate when mediainfo reports it (24000/1001), used for seeking only
    long rate_num = 0, rate_den = 0;
    double seek_rate() const { return rate_num > 0 && rate_den > 0 ? (double)rate_num / (double)rate_den : frame_rate; }
};
struct Options {   // not persisted
    std::string temp_dir, model_dir, ffmpeg = "ffmpeg", mediainfo = "mediainfo";
    int tile = 200;
    std::vector<int> devices{0};   // --gpu 0 or --gpu 0,1,...: frames of a segment are dealt round-robin
    // reve passes no -t: the binary's auto tile size (200 on a large GPU)
    int answer = -1;   // -1 ask, 1 resume, 0 start over
    bool plan = false; // --plan: write the state files, print video.temp and stop (no GPU needed)
    bool pipes = false; // --io pipes: raw RGB over pipes to/from ffmpeg instead of PNG files
};

[[noreturn]] void die(const std::string& m)
{
    std::fprintf(stderr, "error: %s\n", m.c_str());
    std::exit(1);
}

bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
long file_size(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 ? (long)st.st_size : -1; }
std::string ext_of(const std::string& p) { auto d = p.find_last_of('.'); return d == std::string::npos ? "" : p.substr(d + 1); }
std::string abspath(const std::string& p)
{
    if (!p.empty() && p[0] == '/') return p;
    char buf[4096];
    return std::string(getcwd(buf, sizeof buf) ? buf : ".") + "/" + p;
}
void mkdirs(const std::string& p)
{
    for (size_t i = 1; i <= p.size(); ++i)
        if (i == p.size() || p[i] == '/') mkdir(p.substr(0, i).c_str(), 0777);
}
void rm_rf(const std::string& p)
{
    if (p.size() < 4) return;   // never "/" or ""
    const char* argv[] = {"rm", "-rf", p.c_str(), nullptr};
    pid_t pid;
    if (posix_spawnp(&pid, "rm", nullptr, nullptr, (char* const*)argv, environ) == 0) { int st; waitpid(pid, &st, 0); }
}

// Removes what reve itself puts into its temp directory (args.temp, video.temp, parts.txt, tools.log,
// tmp_frames/, out_frames/, video_parts/) and then the directory if that left it empty.  --temp-dir may name any
// directory (/tmp, ~/work): nothing else in it is touched, unlike the reference's remove_dir_all on its own fixed
// exe-relative `temp` (reve-shared/src/lib.rs:291-312).
void remove_own_temp(const std::string& temp)
{
    for (const char* f : {"args.temp", "video.temp", "parts.txt", "tools.log"}) unlink((temp + "/" + f).c_str());
    for (const char* d : {"tmp_frames", "out_frames", "video_parts"}) rm_rf(temp + "/" + d);
    rmdir(temp.c_str());   // fails, harmlessly, when the user keeps other things there
}

// Start time of segment `index` for `ffmpeg -ss` (accurate seek drops every frame whose pts is below it).  The
// reference seeks to (index*segsize - 1)/fps (lib.rs:94-98), one frame early, which re-reads a frame; seeking to
// exactly index*segsize/fps with mediainfo's 3-decimal rate ("23.976" for 24000/1001) lands a few microseconds
// AFTER frame N's pts and loses it.  Half a frame of slack is immune to any rounding of the rate below 0.5/N.
std::string seek_time(int index, int segment_size, double rate)
{
    if (index == 0) return "0";
    char ss[64];
    std::snprintf(ss, sizeof ss, "%.6f", ((double)index * segment_size - 0.5) / rate);
    return ss;
}

// run a tool, optionally capturing stdout; stderr goes to `log` (appended) or /dev/null
int run_tool(const std::vector<std::string>& argv, std::string* out, const std::string& log)
{
    int pfd[2] = {-1, -1};
    if (out && pipe2(pfd, O_CLOEXEC) != 0) return -1;
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
    if (out) { posix_spawn_file_actions_adddup2(&fa, pfd[1], 1); posix_spawn_file_actions_addclose(&fa, pfd[0]); }
    else posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0);
    posix_spawn_file_actions_addopen(&fa, 2, log.empty() ? "/dev/null" : log.c_str(), O_WRONLY | O_CREAT | O_APPEND, 0644);
    std::vector<char*> av;
    for (auto& s : argv) av.push_back(const_cast<char*>(s.c_str()));
    av.push_back(nullptr);
    pid_t pid;
    int rc = posix_spawnp(&pid, av[0], &fa, nullptr, av.data(), environ);
    posix_spawn_file_actions_destroy(&fa);
    if (out) close(pfd[1]);
    if (rc != 0) { if (out) close(pfd[0]); return -1; }
    if (out) {
        char buf[4096];
        ssize_t n;
        while ((n = read(pfd[0], buf, sizeof buf)) > 0) out->append(buf, (size_t)n);
        close(pfd[0]);
    }
    int st = 0;
    waitpid(pid, &st, 0);
    return WIFEXITED(st) ? WEXITSTATUS(st) : -1;
}

// start a tool with its stdout (want_stdout) or stdin (!want_stdout) connected to a pipe; returns pid, fd in *fd
pid_t spawn_piped(const std::vector<std::string>& argv, bool want_stdout, int* fd, const std::string& log)
{
    int pfd[2];
    // close-on-exec: several tools run at once (decoder of segment i+1, encoder of segment i); a later child that
    // inherited the write end of an earlier child's stdin pipe would keep that child from ever seeing end-of-file
    if (pipe2(pfd, O_CLOEXEC) != 0) return -1;
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    if (want_stdout) {
        posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
        posix_spawn_file_actions_adddup2(&fa, pfd[1], 1);
    } else {
        posix_spawn_file_actions_adddup2(&fa, pfd[0], 0);
        posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0);
    }
    posix_spawn_file_actions_addclose(&fa, pfd[0]);
    posix_spawn_file_actions_addclose(&fa, pfd[1]);
    posix_spawn_file_actions_addopen(&fa, 2, log.empty() ? "/dev/null" : log.c_str(), O_WRONLY | O_CREAT | O_APPEND, 0644);
    std::vector<char*> av;
    for (auto& s : argv) av.push_back(const_cast<char*>(s.c_str()));
    av.push_back(nullptr);
    pid_t pid;
    int rc = posix_spawnp(&pid, av[0], &fa, nullptr, av.data(), environ);
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) { close(pfd[0]); close(pfd[1]); return -1; }
    if (want_stdout) { close(pfd[1]); *fd = pfd[0]; } else { close(pfd[0]); *fd = pfd[1]; }
    return pid;
}
bool read_full(int fd, uint8_t* p, size_t n)
{
    while (n) { ssize_t r = read(fd, p, n); if (r <= 0) return false; p += r; n -= (size_t)r; }
    return true;
}
bool write_full(int fd, const uint8_t* p, size_t n)
{
    while (n) { ssize_t r = write(fd, p, n); if (r <= 0) return false; p += r; n -= (size_t)r; }
    return true;
}

// ---- the two JSON state files (flat objects + one array of {index,size}) ----
std::string jstr(const std::string& s)
{
    std::string o = "\"";
    for (char c : s) { if (c == '"' || c == '\\') o += '\\'; o += c; }
    return o + "\"";
}
std::string to_json(const Args& a)
{
    std::ostringstream o;
    o << "{\"inputpath\":" << jstr(a.inputpath) << ",\"outputpath\":" << jstr(a.outputpath) << ",\"scale\":" << a.scale
      << ",\"segmentsize\":" << a.segmentsize << ",\"crf\":" << a.crf << ",\"preset\":" << jstr(a.preset)
      << ",\"x265params\":" << jstr(a.x265params) << "}";
    return o.str();
}
std::string to_json(const Video& v)
{
    std::ostringstream o;
    o << "{\"path\":" << jstr(v.path) << ",\"output_path\":" << jstr(v.output_path) << ",\"segments\":[";
    for (size_t i = 0; i < v.segments.size(); ++i)
        o << (i ? "," : "") << "{\"index\":" << v.segments[i].index << ",\"size\":" << v.segments[i].size << "}";
    o.precision(9);
    o << "],\"frame_rate\":" << v.frame_rate << ",\"frame_count\":" << v.frame_count << ",\"segment_size\":" << v.segment_size
      << ",\"segment_count\":" << v.segment_count << ",\"upscale_ratio\":" << v.upscale_ratio << "}";
    return o.str();
}
// value of "key" in a flat JSON text: string contents or the raw number token
bool jget(const std::string& j, const std::string& key, std::string& out)
{
    size_t p = j.find("\"" + key + "\":");
    if (p == std::string::npos) return false;
    p += key.size() + 3;
    out.clear();
    if (j[p] == '"') {
        for (++p; p < j.size() && j[p] != '"'; ++p) { if (j[p] == '\\' && p + 1 < j.size()) ++p; out += j[p]; }
    } else {
        while (p < j.size() && (std::isdigit((unsigned char)j[p]) || std::strchr("+-.eE", j[p]))) out += j[p++];
    }
    return true;
}
bool parse_args_json(const std::string& j, Args& a)
{
    std::string v;
    if (!jget(j, "inputpath", a.inputpath) || !jget(j, "outputpath", a.outputpath) || !jget(j, "preset", a.preset) ||
        !jget(j, "x265params", a.x265params)) return false;
    if (!jget(j, "scale", v)) return false; a.scale = std::atoi(v.c_str());
    if (!jget(j, "segmentsize", v)) return false; a.segmentsize = std::atoi(v.c_str());
    if (!jget(j, "crf", v)) return false; a.crf = std::atoi(v.c_str());
    return true;
}
bool parse_video_json(const std::string& j, Video& v)
{
    std::string t;
    if (!jget(j, "path", v.path) || !jget(j, "output_path", v.output_path)) return false;
    if (!jget(j, "frame_rate", t)) return false; v.frame_rate = std::atof(t.c_str());
    if (!jget(j, "frame_count", t)) return false; v.frame_count = std::atoi(t.c_str());
    if (!jget(j, "segment_size", t)) return false; v.segment_size = std::atoi(t.c_str());
    if (!jget(j, "segment_count", t)) return false; v.segment_count = std::atoi(t.c_str());
    if (!jget(j, "upscale_ratio", t)) return false; v.upscale_ratio = std::atoi(t.c_str());
    size_t p = j.find("\"segments\":[");
    if (p == std::string::npos) return false;
    const size_t end = j.find(']', p);
    v.segments.clear();
    while ((p = j.find("{\"index\":", p)) != std::string::npos && p < end) {
        Segment s;
        s.index = std::atoi(j.c_str() + p + 9);
        size_t q = j.find("\"size\":", p);
        s.size = std::atoi(j.c_str() + q + 7);
        v.segments.push_back(s);
        p = q;
    }
    return true;
}
std::string slurp(const std::string& p) { std::ifstream f(p); std::stringstream s; s << f.rdbuf(); return s.str(); }
void spit(const std::string& p, const std::string& s) { std::ofstream f(p, std::ios::trunc); f << s; }

void usage()
{
    std::printf(
        "Real-ESRGAN Video Enhance (MI355X / HIP)\nReal-ESRGAN video upscaler with resumability\n\n"
        "Usage: reve [OPTIONS] --inputpath <INPUTPATH> --scale <SCALE> <OUTPUTPATH>\n\n"
        "Arguments:\n  <OUTPUTPATH>  output video path (mp4/mkv)\n\nOptions:\n"
        "  -i, --inputpath <INPUTPATH>      input video path (mp4/mkv)\n"
        "  -s, --scale <SCALE>              upscale ratio (2, 3, 4)\n"
        "  -S, --segmentsize <SEGMENTSIZE>  segment size (in frames) [default: 1000]  (-P accepted too)\n"
        "  -c, --crf <CRF>                  video constant rate factor (crf: 51-0) [default: 15]\n"
        "  -p, --preset <PRESET>            video encoding preset [default: slow]\n"
        "  -x, --x265params <X265PARAMS>    x265 encoding parameters [default: psy-rd=2:aq-strength=1:deblock=0,0:bframes=8]\n"
        "      --model-dir <DIR>            ncnn model files [default: $REVE_MODEL_DIR or <exe dir>/models]\n"
        "      --temp-dir <DIR>             state + scratch directory [default: <exe dir>/temp]\n"
        "      --tile <N|full>              N-pixel tiles with a 10-px apron like the binary [default: 200 = its auto choice];\n"
        "                                   full = whole frame, seam-free and faster\n"
        "      --gpu <ID[,ID...]>           HIP device(s); several = frames dealt round-robin [default: 0]\n"
        "      --yes / --fresh              answer the resume prompt: resume / start over\n"
        "      --plan                       probe + segment + write state files, print video.temp, stop\n"
        "      --io <png|pipes>             frame transport to/from ffmpeg: PNG files like reve [default], or raw RGB pipes\n"
        "      --ffmpeg <EXE> --mediainfo <EXE>\n  -h, --help\n"
        "Environment:\n"
        "  REVE_WINOGRAD=0|1|auto           evaluation of the 16 body layers: auto [default] = Winograd F(2,3) where the model's weights are\n"
        "                                   well conditioned, direct sums otherwise; 0 pins the direct kernels (both within 1 LSB of the oracle)\n");
}

bool valid_preset(const std::string& s)
{
    for (const char* p : {"ultrafast", "superfast", "veryfast", "faster", "fast", "medium", "slow", "slower", "veryslow"})
        if (s == p) return true;
    return false;
}

// clap-equivalent parsing + the validators of lib.rs:249-280
void parse_cli(int argc, char** argv, Args& a, Options& o, bool need_positional)
{
    bool have_in = false, have_out = false, have_scale = false;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        auto val = [&](const char* what) -> std::string {
            auto eq = k.find('=');
            if (k.rfind("--", 0) == 0 && eq != std::string::npos) return k.substr(eq + 1);
            if (i + 1 >= argc) die(std::string("a value is required for '") + what + "'");
            return argv[++i];
        };
        auto is = [&](const char* s, const char* l) { return k == s || k == l || k.rfind(std::string(l) + "=", 0) == 0; };
        if (is("-i", "--inputpath")) { a.inputpath = val("--inputpath"); have_in = true; }
        else if (is("-s", "--scale")) { a.scale = std::atoi(val("--scale").c_str()); have_scale = true; }
        else if (is("-S", "--segmentsize") || k == "-P") a.segmentsize = std::atoi(val("--segmentsize").c_str());
        else if (is("-c", "--crf")) a.crf = std::atoi(val("--crf").c_str());
        else if (is("-p", "--preset")) a.preset = val("--preset");
        else if (is("-x", "--x265params")) a.x265params = val("--x265params");
        else if (is("", "--model-dir")) o.model_dir = val("--model-dir");
        else if (is("", "--temp-dir")) o.temp_dir = val("--temp-dir");
        else if (is("", "--tile")) { const std::string t = val("--tile"); o.tile = t == "full" ? 0 : (std::atoi(t.c_str()) == 0 ? 200 : std::atoi(t.c_str())); }
        else if (is("", "--gpu")) {
            o.devices.clear();
            const std::string v = val("--gpu");
            for (size_t b = 0; b <= v.size();) {
                const size_t e = std::min(v.find(',', b), v.size());
                const std::string t = v.substr(b, e - b);
                if (t.empty() || t.find_first_not_of("0123456789") != std::string::npos) die("--gpu takes a device ordinal or a comma-separated list of them");
                o.devices.push_back(std::atoi(t.c_str()));
                b = e + 1;
            }
        }
        else if (is("", "--ffmpeg")) o.ffmpeg = val("--ffmpeg");
        else if (is("", "--mediainfo")) o.mediainfo = val("--mediainfo");
        else if (k == "--yes") o.answer = 1;
        else if (k == "--fresh") o.answer = 0;
        else if (k == "--plan") o.plan = true;
        else if (is("", "--io")) { const std::string v = val("--io"); if (v != "png" && v != "pipes") die("invalid value for '--io': png/pipes"); o.pipes = v == "pipes"; }
        else if (k == "-h" || k == "--help") { usage(); std::exit(0); }
        else if (!k.empty() && k[0] == '-') die("unexpected argument '" + k + "'");
        else { a.outputpath = k; have_out = true; }
    }
    if (!need_positional) return;
    if (!have_in || !have_out || !have_scale) { usage(); die("the following required arguments were not provided: --inputpath, --scale, <OUTPUTPATH>"); }
    if (!exists(a.inputpath)) die("invalid value for '--inputpath': input path not found");
    if (ext_of(a.inputpath) != "mp4" && ext_of(a.inputpath) != "mkv") die("invalid value for '--inputpath': valid input formats: mp4/mkv");
    if (exists(a.outputpath)) die("invalid value for '<OUTPUTPATH>': output path already exists");
    if (ext_of(a.outputpath) != "mp4" && ext_of(a.outputpath) != "mkv") die("invalid value for '<OUTPUTPATH>': valid output formats: mp4/mkv");
    if (a.scale < 2 || a.scale > 4) die("invalid value for '--scale': 2..=4");
    if (a.crf < 0 || a.crf > 51) die("invalid value for '--crf': 0..=51");
    if (a.segmentsize <= 0) die("invalid value for '--segmentsize'");
    if (!valid_preset(a.preset)) die("invalid value for '--preset': valid: ultrafast/superfast/veryfast/faster/fast/medium/slow/slower/veryslow");
    if (ext_of(a.inputpath) == "mkv" && ext_of(a.outputpath) != "mkv")   // main.rs:124-140
        die("Invalid value \"" + a.inputpath + "\" for '--outputpath <OUTPUTPATH>': mkv file can only be exported as mkv file");
}

// the exact rational rate, where the container has one (absent: the fields come back empty and the decimal is used)
void probe_exact_rate(Video& v, const Options& o)
{
    std::string out;
    if (run_tool({o.mediainfo, "--Output=Video;%FrameRate_Num%/%FrameRate_Den%", v.path}, &out, "") != 0) return;
    long num = 0, den = 0;
    if (std::sscanf(out.c_str(), "%ld/%ld", &num, &den) == 2 && num > 0 && den > 0 &&
        std::fabs((double)num / den - v.frame_rate) < 0.01 * v.frame_rate) { v.rate_num = num; v.rate_den = den; }
}

// Video::new (lib.rs:28-87): probe with mediainfo, split into ceil(frames/segsize) segments
Video probe(const Args& a, const Options& o)
{
    Video v;
    v.path = a.inputpath; v.output_path = a.outputpath; v.segment_size = a.segmentsize; v.upscale_ratio = a.scale;
    std::string out;
    if (run_tool({o.mediainfo, "--Output=Video;%FrameCount%", a.inputpath}, &out, "") != 0) die("failed to execute " + o.mediainfo);
    v.frame_count = std::atoi(out.c_str());
    out.clear();
    if (run_tool({o.mediainfo, "--Output=Video;%FrameRate%", a.inputpath}, &out, "") != 0) die("failed to execute " + o.mediainfo);
    v.frame_rate = std::atof(out.c_str());
    if (v.frame_count <= 0 || !(v.frame_rate > 0)) die("could not probe frame count / frame rate of " + a.inputpath);
    probe_exact_rate(v, o);
    v.segment_count = (v.frame_count + a.segmentsize - 1) / a.segmentsize;
    for (int i = 0; i < v.segment_count; ++i)
        v.segments.push_back({i, std::min(a.segmentsize, v.frame_count - i * a.segmentsize)});
    return v;
}

bool ask(const char* prompt, int preset_answer)
{
    if (preset_answer >= 0) return preset_answer == 1;
    std::printf("%s [Y/n] ", prompt);
    std::fflush(stdout);
    char buf[16];
    if (!std::fgets(buf, sizeof buf, stdin)) return true;
    return !(buf[0] == 'n' || buf[0] == 'N');
}

struct Progress { int done, total, segment; };
void on_frame(void* user, int, const char*, const char*)
{
    auto* p = (Progress*)user;
    p->done++;
    std::fprintf(stderr, "\r[upsc] segment %d: %d/%d", p->segment, p->done, p->total);
    if (p->done == p->total) std::fprintf(stderr, "\n");
}

// ---- what the stages of one run share.  Every failure lands in `failure` and is reported by main() AFTER the worker
// threads have been joined: the GPU contexts are destroyed first, then the process leaves with status 1, the state files in place.
struct Run {
    Args args;
    Options opt;
    Video video;
    std::string temp, log, args_path, video_path;
    std::vector<reve_ctx*> ctxs;
    int G = 1;
    std::string failure;
    std::mutex state_mu;   // video.segments / video.temp are touched by the merge thread / the lanes and read by nobody else meanwhile
    static constexpr int kMaxShortfall = 4;   // frames a LAST segment may come up short of mediainfo's FrameCount and still be taken as the end of the stream

    bool is_last(const Segment& s) const { return s.index == video.segment_count - 1; }
    std::string seg_dir(const char* kind, int i) const { return temp + "/" + kind + "/" + std::to_string(i); }
    std::string part_of(int i) const { return temp + "/video_parts/" + std::to_string(i) + ".mp4"; }
    std::string frame_rate_arg() const   // main.rs:302: format!("{}/1", video.frame_rate)
    {
        char fr[64];
        std::snprintf(fr, sizeof fr, "%.9g/1", video.frame_rate);
        return std::string(fr);
    }
    void checkpoint(int seg_index)   // main.rs:340-343: the segment leaves the state file once its part exists
    {
        std::lock_guard<std::mutex> lk(state_mu);
        for (size_t j = 0; j < video.segments.size(); ++j)
            if (video.segments[j].index == seg_index) { video.segments.erase(video.segments.begin() + j); break; }
        spit(video_path, to_json(video));
        std::fprintf(stderr, "[merg] segment %d/%d done\n", seg_index + 1, video.segment_count);
    }
    // export / merge run on worker threads: they RETURN their status ("" = ok), they never exit the process
    std::string export_segment(Segment s) const   // lib.rs:89-127
    {
        rm_rf(seg_dir("tmp_frames", s.index));
        mkdirs(seg_dir("tmp_frames", s.index));
        int r = run_tool({opt.ffmpeg, "-v", "verbose", "-ss", seek_time(s.index, video.segment_size, video.seek_rate()), "-i", video.path,
                          "-qscale:v", "1", "-qmin", "1", "-qmax", "1", "-vsync", "0", "-vframes", std::to_string(s.size),
                          seg_dir("tmp_frames", s.index) + "/frame%08d.png"}, nullptr, log);
        return r == 0 ? "" : "ffmpeg export of segment " + std::to_string(s.index) + " failed (see " + log + ")";
    }
    std::string merge_segment(Segment s) const    // lib.rs:157-171 + main.rs:297-326
    {
        const std::string part = part_of(s.index);
        int r = run_tool({opt.ffmpeg, "-v", "verbose", "-f", "image2", "-framerate", frame_rate_arg(), "-i", seg_dir("out_frames", s.index) + "/frame%08d.png",
                          "-c:v", "libx265", "-pix_fmt", "yuv420p10le", "-crf", std::to_string(args.crf), "-preset", args.preset,
                          "-x265-params", args.x265params, part}, nullptr, log);
        if (r != 0 || file_size(part) <= 0) { unlink(part.c_str()); return "ffmpeg merge of segment " + std::to_string(s.index) + " failed (see " + log + ")"; }
        rm_rf(seg_dir("out_frames", s.index));
        return "";
    }
};

// The plan of the run: a fresh one from the command line (probe, segment list, state files written) or the one the state
// files of an interrupted run hold.  false: the user declined to go on.
bool load_or_resume(int argc, char** argv, Run& R)
{
    bool resumed = false;
    if (exists(R.args_path)) {
        std::printf("found existing temporary files.\n");
        if (ask("resume upscaling previous video?", R.opt.answer)) {
            if (!parse_args_json(slurp(R.args_path), R.args) || !parse_video_json(slurp(R.video_path), R.video)) die("corrupt state files in " + R.temp);
            resumed = true;
            rm_rf(R.temp + "/tmp_frames"); rm_rf(R.temp + "/out_frames");   // rebuild_temp(true), lib.rs:301-311
            unlink((R.temp + "/parts.txt").c_str());
            probe_exact_rate(R.video, R.opt);                                // not in the state file
            std::printf("resuming upscale\n");
        } else if (!ask("all progress will be lost. do you want to continue?", R.opt.answer == 0 ? 1 : R.opt.answer)) {
            return false;
        }
    }
    if (!resumed) {
        Args fresh;
        parse_cli(argc, argv, fresh, R.opt, true);
        R.args = fresh;
        R.args.inputpath = abspath(R.args.inputpath);
        R.args.outputpath = abspath(R.args.outputpath);
        std::printf("%s loaded\n", R.args.inputpath.c_str());
        remove_own_temp(R.temp);                                          // rebuild_temp(false): only reve's own files
        R.video = probe(R.args, R.opt);
        mkdirs(R.temp + "/video_parts");
        spit(R.args_path, to_json(R.args));
        spit(R.video_path, to_json(R.video));
    }
    mkdirs(R.temp + "/tmp_frames"); mkdirs(R.temp + "/out_frames"); mkdirs(R.temp + "/video_parts");
    if (!R.video.segments.empty()) unlink(R.part_of(R.video.segments[0].index).c_str());
    std::printf("total segments: %d, last segment size: %d\n", R.video.segment_count,
                R.video.frame_count - (R.video.segment_count - 1) * R.video.segment_size);
    return true;
}

// ---- PNG transport: the reference's 3-stage pipeline over the remaining segments, export(i+1) || upscale(i) || merge(i-1)
void run_png_segments(Run& R)
{
    const std::vector<Segment> todo = R.video.segments;
    std::thread export_thread, merge_thread;
    std::string export_status, merge_status;     // written by the worker, read after join()
    if (!todo.empty()) R.failure = R.export_segment(todo[0]);
    for (size_t k = 0; k < todo.size() && R.failure.empty(); ++k) {
        const Segment s = todo[k];
        if (k + 1 < todo.size()) export_thread = std::thread([&, k] { export_status = R.export_segment(todo[k + 1]); });
        rm_rf(R.seg_dir("out_frames", s.index));
        mkdirs(R.seg_dir("out_frames", s.index));
        Progress pr{0, s.size, s.index};
        const int rc = reve_upscale_dir_multi(R.ctxs.data(), R.G, R.seg_dir("tmp_frames", s.index).c_str(), R.seg_dir("out_frames", s.index).c_str(), on_frame, &pr);
        if (rc != REVE_OK) R.failure = "upscaling segment " + std::to_string(s.index) + " failed: " + reve_last_error(R.ctxs[0]);
        else if (pr.done != s.size) {
            if (R.is_last(s) && pr.done > 0 && s.size - pr.done <= Run::kMaxShortfall) std::fprintf(stderr, "\nnote: the stream ended %d frame(s) before its declared length\n", s.size - pr.done);
            else R.failure = "upscaling segment " + std::to_string(s.index) + " failed: frame count mismatch (" + std::to_string(pr.done) + " of " + std::to_string(s.size) + ")";
        }
        if (!R.failure.empty()) break;
        rm_rf(R.seg_dir("tmp_frames", s.index));
        if (merge_thread.joinable()) merge_thread.join();
        if (!merge_status.empty()) break;
        merge_thread = std::thread([&, s] {
            merge_status = R.merge_segment(s);
            if (merge_status.empty()) R.checkpoint(s.index);
        });
        if (export_thread.joinable()) export_thread.join();
        if (!export_status.empty()) break;
    }
    if (export_thread.joinable()) export_thread.join();
    if (merge_thread.joinable()) merge_thread.join();
    if (R.failure.empty() && !export_status.empty()) R.failure = export_status;
    if (R.failure.empty() && !merge_status.empty()) R.failure = merge_status;
}

// ---- pipe transport: ffmpeg decodes straight into pinned ring slots and encodes straight out of them; no
// PNG codec, no frame files.  One decoder and one encoder process per segment.
struct SegIO { Segment s; pid_t dec = -1, enc = -1; int dfd = -1, efd = -1; int written = 0, expect = 0; std::string part; bool reaped = false; };
struct Pipes {
    Run& R;
    int fw = 0, fh = 0, sc = 2;
    size_t in_bytes = 0, out_bytes = 0;
    std::string size;              // "WxH" of the upscaled frames
    std::vector<SegIO> io;

    explicit Pipes(Run& r) : R(r) {}
    bool probe()
    {
        std::string out;
        run_tool({R.opt.mediainfo, "--Output=Video;%Width%", R.video.path}, &out, "");
        fw = std::atoi(out.c_str());
        out.clear();
        run_tool({R.opt.mediainfo, "--Output=Video;%Height%", R.video.path}, &out, "");
        fh = std::atoi(out.c_str());
        if (fw <= 0 || fh <= 0) { R.failure = "could not probe the frame size of " + R.video.path; return false; }
        sc = R.args.scale;
        in_bytes = (size_t)fw * fh * 3; out_bytes = in_bytes * sc * sc;
        size = std::to_string(fw * sc) + "x" + std::to_string(fh * sc);
        for (const Segment& s : R.video.segments) { SegIO x; x.s = s; x.expect = s.size; x.part = R.part_of(s.index); io.push_back(x); }
        return true;
    }
    bool start_decoder(SegIO& x) const
    {
        x.dec = spawn_piped({R.opt.ffmpeg, "-v", "error", "-ss", seek_time(x.s.index, R.video.segment_size, R.video.seek_rate()), "-i", R.video.path,
                             "-vsync", "0", "-vframes", std::to_string(x.s.size), "-f", "rawvideo", "-pix_fmt", "rgb24", "-"}, true, &x.dfd, R.log);
        return x.dec >= 0;
    }
    bool start_encoder(SegIO& x) const
    {
        unlink(x.part.c_str());
        x.enc = spawn_piped({R.opt.ffmpeg, "-v", "error", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", size, "-framerate", R.frame_rate_arg(), "-i", "-",
                             "-c:v", "libx265", "-pix_fmt", "yuv420p10le", "-crf", std::to_string(R.args.crf), "-preset", R.args.preset,
                             "-x265-params", R.args.x265params, x.part}, false, &x.efd, R.log);
        return x.enc >= 0;
    }
    // The decoder's pipe ended after k of the segment's frames.  Containers disagree with mediainfo's FrameCount by a frame now
    // and then: a LAST segment that ends a few frames early is accepted as it is (true: x.expect = k) — but only when the decoder
    // itself says the stream was over (clean exit): a decoder that crashed or was killed in the middle of the last segment must not
    // end in a truncated part that is checkpointed and concatenated.  Anywhere else a short read is a lost frame (false, err set).
    bool short_read_is_end_of_stream(SegIO& x, int k, std::string& err) const
    {
        const std::string seg = "segment " + std::to_string(x.s.index) + " failed (decoder ";
        const std::string of = " of " + std::to_string(x.s.size) + " frames; see " + R.log + ")";
        if (R.is_last(x.s) && k > 0 && x.s.size - k <= Run::kMaxShortfall) {
            int dst = 0;
            close(x.dfd); x.dfd = -1;
            const pid_t dr = waitpid(x.dec, &dst, 0);
            x.dec = -1;
            if (dr > 0 && WIFEXITED(dst) && WEXITSTATUS(dst) == 0) {
                std::fprintf(stderr, "\nnote: the stream ended %d frame(s) before its declared length\n", x.s.size - k);
                x.expect = k;
                return true;
            }
            err = seg + "died after " + std::to_string(k) + of;
            return false;
        }
        err = seg + "delivered " + std::to_string(k) + of;
        return false;
    }
    bool encoder_finished_well(SegIO& x, bool block, bool* still_running) const
    {
        int st = 0;
        const pid_t r = waitpid(x.enc, &st, block ? 0 : WNOHANG);
        if (still_running) *still_running = r == 0;
        if (r == 0) return false;
        x.reaped = true;
        return r > 0 && WIFEXITED(st) && WEXITSTATUS(st) == 0 && file_size(x.part) > 0;
    }
};

// Several GPUs: one LANE per GPU, each a thread that takes the next segment that nobody has and runs it whole — its own decoder
// and encoder processes, its own three pinned ring slots, its own context.  One decoder's stdout is a serial stream (a pipe moves
// a few GB/s: ~600 frames/s of 1080p), so frames of ONE segment dealt over G rings from one reader thread cannot feed eight
// GPUs; G segments in flight can.  Segments finish out of order; the state file lists what is still to do (main.rs:340-343),
// so the unit of resume is unchanged.
struct Lanes {
    Pipes& P;
    std::mutex fail_mu;
    std::atomic<size_t> next_seg{0};
    std::atomic<bool> stop{false};
    explicit Lanes(Pipes& p) : P(p) {}
    void fail_with(const std::string& what)
    {
        std::lock_guard<std::mutex> lk(fail_mu);
        if (P.R.failure.empty()) P.R.failure = what;
        stop = true;
    }
    // one segment, whole, on GPU g (a failure in another lane stops new segments from being taken; the one in hand is finished: its part is good)
    std::string run_segment(SegIO& x, int g, uint8_t* const (&ib)[3], uint8_t* const (&ob)[3])
    {
        Run& R = P.R;
        reve_ctx* ctx = R.ctxs[g];
        std::string err;
        uint64_t sub = 0, ret = 0;
        if (!P.start_decoder(x) || !P.start_encoder(x)) err = "could not start ffmpeg";
        auto retire1 = [&] {
            uint64_t id = 0;
            if (reve_wait(ctx, &id) != REVE_OK || id != ret) { err = std::string("upscaling failed: ") + reve_last_error(ctx); return; }
            if (!write_full(x.efd, ob[ret % 3], P.out_bytes)) { err = "segment " + std::to_string(x.s.index) + " failed (encoder closed its input; see " + R.log + ")"; return; }
            ++ret; ++x.written;
        };
        for (int k = 0; k < x.s.size && err.empty(); ++k) {
            if (sub - ret == 3) retire1();
            if (!err.empty()) break;
            if (!read_full(x.dfd, ib[sub % 3], P.in_bytes)) {
                (void)P.short_read_is_end_of_stream(x, k, err);
                break;
            }
            if (reve_submit(ctx, sub, ib[sub % 3], P.fw, P.fh, (ptrdiff_t)P.fw * 3, ob[sub % 3], (ptrdiff_t)P.fw * P.sc * 3) != REVE_OK) {
                err = std::string("upscaling failed: ") + reve_last_error(ctx);
                break;
            }
            ++sub;
        }
        while (err.empty() && ret < sub) retire1();
        while (ret < sub) { uint64_t id; if (reve_wait(ctx, &id) != REVE_OK) break; ++ret; }      // (failure: nothing may stay on the ring)
        if (x.dfd >= 0) { close(x.dfd); x.dfd = -1; }
        if (x.dec > 0) { int st; if (!err.empty()) kill(x.dec, SIGTERM); waitpid(x.dec, &st, 0); x.dec = -1; }
        if (x.efd >= 0) { close(x.efd); x.efd = -1; }
        if (x.enc > 0 && !P.encoder_finished_well(x, true, nullptr) && err.empty()) err = "segment " + std::to_string(x.s.index) + " failed (encoder; see " + R.log + ")";
        return err;
    }
    void lane(int g)
    {
        Run& R = P.R;
        (void)reve_bind_thread_to_device(R.opt.devices[g]);
        uint8_t* ib[3]; uint8_t* ob[3];
        bool ok = true;
        for (int k = 0; k < 3; ++k) { ib[k] = (uint8_t*)reve_alloc_pinned(P.in_bytes); ob[k] = (uint8_t*)reve_alloc_pinned(P.out_bytes); ok &= ib[k] && ob[k]; }
        if (!ok) fail_with("pinned allocation failed");
        while (ok && !stop) {
            const size_t j = next_seg.fetch_add(1);
            if (j >= P.io.size()) break;
            SegIO& x = P.io[j];
            const std::string err = run_segment(x, g, ib, ob);
            if (err.empty() && x.written == x.expect) {
                std::fprintf(stderr, "[upsc] segment %d: %d/%d (gpu %d)\n", x.s.index, x.written, x.expect, R.opt.devices[g]);
                R.checkpoint(x.s.index);
            } else {
                unlink(x.part.c_str());        // incomplete: redone on resume
                if (!err.empty()) fail_with(err);
            }
        }
        for (int k = 0; k < 3; ++k) { reve_free_pinned(ib[k]); reve_free_pinned(ob[k]); }
    }
    void run()
    {
        std::vector<std::thread> ts;
        for (int g = 0; g < P.R.G; ++g) ts.emplace_back([this, g] { lane(g); });
        for (auto& t : ts) t.join();
    }
};

// One GPU: the ring never drains between segments — segment i+1's decoder is started when segment i's first frame is read (it
// runs ahead until its pipe is full) and its frames enter the ring while segment i's last frames are still on the GPU and its
// encoder is still draining; encoders are reaped in segment order without blocking the frame loop.  The unit of resume is
// still the segment.
struct SingleLane {
    Pipes& P;
    Run& R;
    static constexpr int depth = 3;
    uint8_t* in_buf[depth] = {nullptr, nullptr, nullptr};
    uint8_t* out_buf[depth] = {nullptr, nullptr, nullptr};
    size_t next_reap = 0;          // encoders finish in segment order; the checkpoint follows the same order
    std::deque<size_t> fly;        // the segment of each frame on the ring, oldest first
    uint64_t submitted = 0, retired = 0;
    explicit SingleLane(Pipes& p) : P(p), R(p.R) {}

    void reap(bool block)
    {
        while (R.failure.empty() && next_reap < P.io.size()) {
            SegIO& x = P.io[next_reap];
            if (x.enc < 0 || x.efd >= 0) return;              // not started, or still being fed
            bool running = false;
            if (!P.encoder_finished_well(x, block, &running)) {
                if (running) return;
                unlink(x.part.c_str());
                R.failure = "segment " + std::to_string(x.s.index) + " failed (encoder; see " + R.log + ")";
                return;
            }
            R.checkpoint(x.s.index);
            ++next_reap;
        }
    }
    void retire()
    {
        uint64_t id = 0;
        SegIO& x = P.io[fly.front()];
        if (reve_wait(R.ctxs[0], &id) != REVE_OK || id != retired) { R.failure = std::string("upscaling failed: ") + reve_last_error(R.ctxs[0]); return; }
        if (!write_full(x.efd, out_buf[retired % depth], P.out_bytes)) { R.failure = "segment " + std::to_string(x.s.index) + " failed (encoder closed its input; see " + R.log + ")"; return; }
        ++retired;
        fly.pop_front();
        std::fprintf(stderr, "\r[upsc] segment %d: %d/%d", x.s.index, ++x.written, x.expect);
        if (x.written == x.expect) { std::fprintf(stderr, "\n"); close(x.efd); x.efd = -1; }
        reap(false);
    }
    void feed_segment(size_t j)
    {
        SegIO& x = P.io[j];
        if (j + 1 < P.io.size() && !P.start_decoder(P.io[j + 1])) { R.failure = "could not start ffmpeg"; return; }
        if (!P.start_encoder(x)) { R.failure = "could not start ffmpeg"; return; }
        for (int k = 0; k < x.s.size && R.failure.empty(); ++k) {
            if ((int)fly.size() == depth) retire();
            if (!R.failure.empty()) break;
            if (!read_full(x.dfd, in_buf[submitted % depth], P.in_bytes)) {
                (void)P.short_read_is_end_of_stream(x, k, R.failure);
                break;
            }
            if (reve_submit(R.ctxs[0], submitted, in_buf[submitted % depth], P.fw, P.fh, (ptrdiff_t)P.fw * 3, out_buf[submitted % depth],
                            (ptrdiff_t)P.fw * P.sc * 3) != REVE_OK) { R.failure = std::string("upscaling failed: ") + reve_last_error(R.ctxs[0]); break; }
            fly.push_back(j);
            ++submitted;
        }
        if (x.dfd >= 0) { close(x.dfd); x.dfd = -1; }
        if (x.dec > 0) { int st; waitpid(x.dec, &st, 0); x.dec = -1; }
        if (R.failure.empty() && x.written == x.expect && x.efd >= 0) { close(x.efd); x.efd = -1; }   // a short last segment whose frames all left already
    }
    void run()
    {
        for (int k = 0; k < depth && R.failure.empty(); ++k) {
            in_buf[k] = (uint8_t*)reve_alloc_pinned(P.in_bytes);
            out_buf[k] = (uint8_t*)reve_alloc_pinned(P.out_bytes);
            if (!in_buf[k] || !out_buf[k]) R.failure = "pinned allocation failed";
        }
        if (R.failure.empty() && !P.io.empty() && !P.start_decoder(P.io[0])) R.failure = "could not start ffmpeg";
        for (size_t j = 0; j < P.io.size() && R.failure.empty(); ++j) feed_segment(j);
        while (R.failure.empty() && !fly.empty()) retire();
        reap(true);
        // tidy up whatever is still open (failure paths): frames still on the ring are waited for, children are reaped, parts of
        // segments that did not complete are removed
        while (retired < submitted) { uint64_t id; if (reve_wait(R.ctxs[0], &id) != REVE_OK) break; ++retired; }
        for (SegIO& x : P.io) {
            if (x.dfd >= 0) close(x.dfd);
            if (x.efd >= 0) close(x.efd);
            int st;
            if (x.dec > 0) { kill(x.dec, SIGTERM); waitpid(x.dec, &st, 0); }
            if (x.enc > 0 && !x.reaped) { waitpid(x.enc, &st, 0); unlink(x.part.c_str()); }
        }
        for (int k = 0; k < depth; ++k) { if (in_buf[k]) reve_free_pinned(in_buf[k]); if (out_buf[k]) reve_free_pinned(out_buf[k]); }
    }
};

void run_pipe_segments(Run& R)
{
    Pipes P(R);
    if (!P.probe()) return;
    signal(SIGPIPE, SIG_IGN);
    if (R.G > 1) Lanes(P).run();
    else SingleLane(P).run();
}

// ---- concatenate (lib.rs:173-206) and validate (main.rs:355-363)
void concat_and_validate(Run& R)
{
    std::printf("merging video segments\n");
    std::string parts;
    for (int i = 0; i < R.video.segment_count; ++i) parts += std::string(i ? "\n" : "") + "file 'video_parts/" + std::to_string(i) + ".mp4'";
    spit(R.temp + "/parts.txt", parts);
    run_tool({R.opt.ffmpeg, "-f", "concat", "-safe", "0", "-i", R.temp + "/parts.txt", "-i", R.video.path, "-map", "0:v", "-map", "1:a?",
              "-map", "1:s?", "-map_chapters", "1", "-c", "copy", R.video.output_path}, nullptr, R.log);
    unlink((R.temp + "/parts.txt").c_str());
    if (file_size(R.video.output_path) > 0) remove_own_temp(R.temp);
    else die("final file validation error: try running again");
    std::printf("done!\n");
}

}  // namespace

int main(int argc, char** argv)
{
    char exe[4096];
    ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
    std::string exe_dir = ".";
    if (n > 0) { exe[n] = 0; exe_dir = std::string(exe).substr(0, std::string(exe).find_last_of('/')); }

    Run R;
    parse_cli(argc, argv, R.args, R.opt, false);   // options only; positional/validation depends on resume
    R.temp = R.opt.temp_dir.empty() ? exe_dir + "/temp" : abspath(R.opt.temp_dir);
    if (R.opt.model_dir.empty()) R.opt.model_dir = std::getenv("REVE_MODEL_DIR") ? std::getenv("REVE_MODEL_DIR") : exe_dir + "/models";
    R.args_path = R.temp + "/args.temp"; R.video_path = R.temp + "/video.temp"; R.log = R.temp + "/tools.log";
    if (!load_or_resume(argc, argv, R)) return 1;
    if (R.opt.plan) {
        std::printf("%s\n", to_json(R.video).c_str());
        return 0;
    }

    // ---- the upscaler: one context per GPU for the whole run (the reference pays process start + model load + pipeline
    // compile once per segment, lib.rs:134)
    reve_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.scale = R.args.scale; cfg.device = R.opt.devices[0]; cfg.tile = R.opt.tile;
    cfg.model_dir = R.opt.model_dir.c_str(); cfg.model_name = "realesr-animevideov3";
    // the pipe lanes keep three frames in flight (ib / ob[sub % 3]): saying so lets the library launch a partial batch of small
    // frames when the GPU is idle instead of holding them for a batch this ring will never fill (include/reve_hip.h, "batch")
    if (R.opt.pipes) cfg.ring_depth = 3;
    R.G = (int)R.opt.devices.size();
    R.ctxs.assign(R.G, nullptr);
    const int rc = reve_create_group(&cfg, R.opt.devices.data(), R.G, R.ctxs.data());
    if (rc != REVE_OK) die(std::string("upscaler: ") + reve_strerror(rc) + " (" + reve_last_error(nullptr) + ")");

    if (R.opt.pipes) run_pipe_segments(R);
    else run_png_segments(R);
    for (reve_ctx* c : R.ctxs) reve_destroy(c);
    if (!R.failure.empty()) {
        std::fprintf(stderr, "error: %s (state kept; run again to resume)\n", R.failure.c_str());
        return 1;
    }
    concat_and_validate(R);
    return 0;
}
