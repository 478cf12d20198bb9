#include "dirmode.h"

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstring>
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

int upscale_dir(Engine& eng, const std::string& in_dir, const std::string& out_dir, reve_progress_cb cb,
                void* user, std::string& err)
{
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
    int idx = 0, first_rc = 0;
    for (const std::string& n : names) {
        const std::string ip = in_dir + "/" + n;
        const std::string op = out_dir + "/" + n.substr(0, n.size() - 4) + ".png";
        std::string e;
        int rc = upscale_file(eng, ip, op, e);
        if (rc != 0) {            // keep going like the binary does, but report the first failure
            if (!first_rc) { first_rc = rc; err = e; }
        } else if (cb) {
            cb(user, idx, ip.c_str(), op.c_str());
        }
        ++idx;
    }
    return first_rc;
}

}  // namespace reve
