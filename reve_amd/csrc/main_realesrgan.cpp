// realesrgan-hip — argv/stderr-compatible stand-in for the `realesrgan-ncnn-vulkan` executable that
// reve spawns (reve-shared/src/lib.rs:134-147: `-i DIR -o DIR -n MODEL -s S -f png -v`;
// reve-gui/src-tauri/src/commands.rs:52-65: `-i FILE -o FILE -m models -n MODEL -s S`).
// Progress protocol kept: one stderr line containing "done" per finished frame and nothing else on
// stderr contains "done" (reve-cli/src/main.rs:266-273).  Unlike the reference's caller
// (lib.rs:150-154 drops the Child), the exit status is meaningful: non-zero if any frame failed.
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/reve_hip.h"

static void usage()
{
    std::fprintf(stderr,
                 "Usage: realesrgan-hip -i infile -o outfile [options]...\n"
                 "  -i input-path   input image path (png) or directory\n"
                 "  -o output-path  output image path (png) or directory\n"
                 "  -s scale        upscale ratio (2, 3, 4; default 4)\n"
                 "  -t tile-size    tile size (>=32/0=auto like the original: 200 on this GPU, default=0; \"full\" = whole frame, seam-free);\n"
                 "                  without -t the environment variable REVE_TILE=full|N is honoured\n"
                 "  -m model-path   folder path to the models (default models)\n"
                 "  -n model-name   model name (default realesr-animevideov3)\n"
                 "  -g gpu-id       HIP device to use (default 0), or a list 0,1,2 for multi-GPU\n"
                 "  -j l:p:s        accepted for compatibility, ignored\n"
                 "  -f format       output format (png only)\n"
                 "  -x              TTA mode (rejected)\n"
                 "  -v              verbose output\n"
                 "  --model-report  print what the library sees in the model -m / -n / -s select (per-layer gains, the conditioning\n"
                 "                  estimate kappa, the evaluation it would choose) as JSON on stdout and leave; needs no GPU, no -i / -o\n"
                 "Environment (for callers whose argv cannot change: reve passes no -t and no options, reve-shared/src/lib.rs:134-147):\n"
                 "  REVE_TILE=full|N        without -t: whole frames (seam-free, ~1.3x faster) or N-pixel tiles instead of 200\n"
                 "  REVE_WINOGRAD=0|1|auto  how the 16 body layers are evaluated: auto (default) = Winograd F(2,3) where the model's\n"
                 "                          weights are well conditioned (kappa < 0.5, see --model-report), else direct sums;\n"
                 "                          0 pins the direct kernels, 1 forces Winograd.  Both are within 1 LSB of the CPU oracle\n"
                 "  REVE_DIR_STATS=1        per-stage times of directory mode on stderr\n");
}

static int g_verbose = 0;
static void on_frame(void*, int, const char* in, const char* out)
{
    if (g_verbose) std::fprintf(stderr, "%s -> %s done\n", in, out);
}

int main(int argc, char** argv)
{
    std::string in, out, model_dir = "models", model = "realesr-animevideov3", fmt = "png";
    int scale = 4, tile = 0;
    bool tile_given = false, model_report = false;
    std::vector<int> gpus{0};
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", what); std::exit(2); }
            return argv[++i];
        };
        if (a == "-i") in = need("-i");
        else if (a == "-o") out = need("-o");
        else if (a == "-s") scale = std::atoi(need("-s"));
        else if (a == "-t") { const char* t = need("-t"); tile = std::strcmp(t, "full") == 0 ? -1 : std::atoi(t); tile_given = true; }
        else if (a == "-m") model_dir = need("-m");
        else if (a == "-n") model = need("-n");
        else if (a == "-g") {   // "0" or "0,1,2" (multi-GPU like the original binary: frames are dealt round-robin)
            gpus.clear();
            for (const char* p = need("-g"); *p;) {
                char* e = nullptr;
                const long v = std::strtol(p, &e, 10);
                if (e == p) { std::fprintf(stderr, "bad -g list\n"); return 2; }
                gpus.push_back((int)v);
                p = (*e == ',') ? e + 1 : e;
                if (*e && *e != ',') { std::fprintf(stderr, "bad -g list\n"); return 2; }
            }
            if (gpus.empty()) { std::fprintf(stderr, "bad -g list\n"); return 2; }
        }
        else if (a == "-j") (void)need("-j");
        else if (a == "-f") fmt = need("-f");
        else if (a == "-v") g_verbose = 1;
        else if (a == "-x") { std::fprintf(stderr, "TTA mode is not supported\n"); return 2; }
        else if (a == "--model-report") model_report = true;
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); usage(); return 2; }
    }
    if (model_report) {
        // no GPU, no frames: the report of the model alone (`-s` defaults to 4 like an upscale would)
        static char text[1 << 16];
        const int rc = reve_model_report(model_dir.c_str(), model.c_str(), scale, text, sizeof text);
        if (rc != REVE_OK) { std::fprintf(stderr, "model report failed: %s (%s)\n", reve_strerror(rc), reve_last_error(nullptr)); return 1; }
        std::fputs(text, stdout);
        return 0;
    }
    if (in.empty() || out.empty()) { usage(); return 2; }
    if (fmt != "png") { std::fprintf(stderr, "only -f png is supported\n"); return 2; }
    for (int gpu : gpus)
        if (gpu < 0) { std::fprintf(stderr, "CPU mode (-g -1) does not exist in this build: a gfx950 GPU is required\n"); return 2; }

    // reve never passes -t (reve-shared/src/lib.rs:134-147), so an unmodified reve gets the original's auto choice, 200-pixel tiles
    // and their seams.  REVE_TILE=full|N in the environment chooses otherwise WITHOUT touching reve's argv: "full" = whole frames,
    // seam-free and ~1.3x faster.  An explicit -t wins.  Said on stderr in words reve's line counter ignores.
    if (!tile_given) {
        if (const char* e = std::getenv("REVE_TILE")) {
            if (std::strcmp(e, "full") == 0) tile = -1;
            else if (std::atoi(e) >= 32) tile = std::atoi(e);
            else if (*e) { std::fprintf(stderr, "REVE_TILE must be \"full\" or a tile size >= 32\n"); return 2; }
            if (*e) std::fprintf(stderr, "note: REVE_TILE=%s: %s\n", e, tile < 0 ? "whole frames, no tile seams" : "tile size from the environment");
        }
    }
    reve_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    // the original picks the tile size from the GPU's heap budget (> 1900 MB -> 200);
    // any MI355X is in that class, so "auto" is 200: same seams as the reference's default run
    cfg.scale = scale; cfg.device = gpus[0]; cfg.tile = tile == 0 ? 200 : (tile < 0 ? 0 : tile);
    cfg.model_dir = model_dir.c_str(); cfg.model_name = model.c_str();
    {
        // An unmodified reve-cli names the x2 model for every --scale (reve-shared/src/lib.rs:140-143); the graph that
        // matches -s is loaded instead.  Said on stderr in words reve's line counter ignores
        // (it counts lines containing the letters d-o-n-e, reve-cli/src/main.rs:266-273).
        char resolved[512];
        if (reve_resolve_model_name(model.c_str(), scale, resolved, sizeof resolved) == 1)
            std::fprintf(stderr, "note: -n %s with -s %d: loading %s (the graph that matches the scale)\n", model.c_str(), scale, resolved);
    }
    std::vector<reve_ctx*> ctxs(gpus.size(), nullptr);
    const bool stats = std::getenv("REVE_DIR_STATS") && std::getenv("REVE_DIR_STATS")[0] == '1';
    const auto t_main = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    int rc = reve_create_group(&cfg, gpus.data(), (int)gpus.size(), ctxs.data());
    const double ms_create = ms_since(t_main);
    if (rc != REVE_OK) {
        std::fprintf(stderr, "reve_create failed: %s (%s)\n", reve_strerror(rc), reve_last_error(nullptr));
        return 1;
    }
    reve_ctx* ctx = ctxs[0];
    struct stat st;
    const bool is_dir = stat(in.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
    if (is_dir) {
        rc = reve_upscale_dir_multi(ctxs.data(), (int)ctxs.size(), in.c_str(), out.c_str(), on_frame, nullptr);
    } else {
        rc = reve_upscale_file(ctx, in.c_str(), out.c_str());
        if (rc == REVE_OK) on_frame(nullptr, 0, in.c_str(), out.c_str());
    }
    if (rc != REVE_OK) std::fprintf(stderr, "failed: %s (%s)\n", reve_strerror(rc), reve_last_error(ctx));
    const double ms_work = ms_since(t_main) - ms_create;
    for (reve_ctx* c : ctxs) reve_destroy(c);
    if (stats)
        std::fprintf(stderr, "[main] context creation (runtime start, model, weights, kernels) %.1f ms, upscaling %.1f ms, teardown %.1f ms\n", ms_create, ms_work,
                     ms_since(t_main) - ms_create - ms_work);
    return rc == REVE_OK ? 0 : 1;
}
