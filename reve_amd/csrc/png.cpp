#include "png.h"

#include "fastdeflate.h"
#include "fastinflate.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>

namespace reve {

static const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static void put32(std::vector<uint8_t>& v, uint32_t x)
{
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}

static inline int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

std::string read_file(const std::string& path, std::vector<uint8_t>& out)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return "cannot open " + path;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    try {
        out.resize(n > 0 ? (size_t)n : 0);
    } catch (const std::bad_alloc&) {
        std::fclose(f);
        return "out of memory reading " + path;
    }
    size_t got = n > 0 ? std::fread(out.data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    if (got != out.size()) return "short read on " + path;
    return "";
}

std::string write_file(const std::string& path, const std::vector<uint8_t>& data)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return "cannot create " + path;
    size_t put = std::fwrite(data.data(), 1, data.size(), f);
    if (std::fclose(f) != 0 || put != data.size()) return "short write on " + path;
    return "";
}

using PixelSink = std::function<uint8_t*(int w, int h)>;
static std::string decode_impl(const std::vector<uint8_t>& file, const PixelSink& sink, int& w, int& h, std::vector<uint8_t>* alpha);

// Frame files are untrusted input (anything may sit in tmp_frames/): every malformed file is an error string, and so
// is an allocation failure — this function is called from the C ABI and from pool threads, where an escaping
// std::bad_alloc would end the process.
std::string png_decode_rgb8(const std::vector<uint8_t>& file, std::vector<uint8_t>& rgb, int& w, int& h)
{
    return png_decode_rgb8_to(file, [&](int ww, int hh) { rgb.resize((size_t)ww * hh * 3); return rgb.data(); }, w, h);
}

std::string png_decode_rgb8_to(const std::vector<uint8_t>& file, const PixelSink& sink, int& w, int& h)
{
    try {
        return decode_impl(file, sink, w, h, nullptr);
    } catch (const std::bad_alloc&) {
        return "out of memory decoding PNG";
    }
}

std::string png_decode_rgba8(const std::vector<uint8_t>& file, std::vector<uint8_t>& rgb, std::vector<uint8_t>& alpha, int& w, int& h)
{
    alpha.clear();
    try {
        return decode_impl(file, [&](int ww, int hh) { rgb.resize((size_t)ww * hh * 3); return rgb.data(); }, w, h, &alpha);
    } catch (const std::bad_alloc&) {
        return "out of memory decoding PNG";
    }
}

// dst[i] = raw[i] + prev[i] (the Up filter undone), restrict pointers so that the loop vectorises
static void add_rows(uint8_t* __restrict o, const uint8_t* __restrict a, const uint8_t* __restrict b, size_t n)
{
    for (size_t i = 0; i < n; ++i) o[i] = (uint8_t)(a[i] + b[i]);
}

// One scanline of 8-bit RGB un-filtered from `raw` (its filter byte stripped) into `dst`; `prev` = the row above in the
// destination (nullptr: the first row, above which everything is zero).  The serial filters run three chains, one per channel.
static bool unfilter_rgb8(int ft, const uint8_t* raw, uint8_t* dst, const uint8_t* prev, size_t rowb)
{
    switch (ft) {
    case 0: std::memcpy(dst, raw, rowb); return true;
    case 2:
        if (prev) add_rows(dst, raw, prev, rowb);
        else std::memcpy(dst, raw, rowb);
        return true;
    case 1: {
        uint8_t a = 0, b = 0, c = 0;
        size_t i = 0;
        for (; i + 3 <= rowb; i += 3) {
            a = (uint8_t)(a + raw[i]); b = (uint8_t)(b + raw[i + 1]); c = (uint8_t)(c + raw[i + 2]);
            dst[i] = a; dst[i + 1] = b; dst[i + 2] = c;
        }
        return true;
    }
    case 3:
        for (size_t i = 0; i < rowb; ++i) dst[i] = (uint8_t)(raw[i] + (((i >= 3 ? dst[i - 3] : 0) + (prev ? prev[i] : 0)) >> 1));
        return true;
    case 4:
        for (size_t i = 0; i < rowb; ++i)
            dst[i] = (uint8_t)(raw[i] + paeth(i >= 3 ? dst[i - 3] : 0, prev ? prev[i] : 0, (i >= 3 && prev) ? prev[i - 3] : 0));
        return true;
    default: return false;
    }
}

static std::string decode_impl(const std::vector<uint8_t>& file, const PixelSink& sink, int& w, int& h, std::vector<uint8_t>* alpha)
{
    if (file.size() < 8 + 25 || std::memcmp(file.data(), kSig, 8) != 0) return "not a PNG file";
    size_t off = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat_joined, plte, trns;
    const uint8_t* idat = nullptr;          // the image data: in place when the file has ONE IDAT chunk (every encoder of whole frames
    size_t idat_len = 0;                    // writes one or a few), else the chunks' payloads joined
    int n_idat = 0;
    bool have_ihdr = false, have_iend = false;
    w = h = 0;
    while (off + 12 <= file.size()) {
        const uint32_t len = be32(&file[off]);
        const uint8_t* type = &file[off + 4];
        if (off + 12 + (size_t)len > file.size()) return "truncated PNG chunk";
        const uint8_t* data = &file[off + 8];
        const uint32_t crc = be32(&file[off + 8 + len]);
        if (fast_crc32(fast_crc32(0, type, 4), data, len) != crc) return "PNG chunk CRC mismatch";
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return "bad IHDR";
            w = (int)be32(data); h = (int)be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) return "unsupported PNG compression/filter method";
            have_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            if (n_idat == 0) { idat = data; idat_len = len; }
            else {
                if (n_idat == 1) idat_joined.assign(idat, idat + idat_len);
                idat_joined.insert(idat_joined.end(), data, data + len);
            }
            ++n_idat;
        } else if (!std::memcmp(type, "IEND", 4)) {
            have_iend = true;
            break;
        }
        off += 12 + (size_t)len;
    }
    if (n_idat > 1) { idat = idat_joined.data(); idat_len = idat_joined.size(); }
    if (!have_ihdr || !have_iend) return "PNG missing IHDR/IEND";
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535) return "unreasonable PNG dimensions";
    if (interlace > 1) return "bad PNG interlace method";
    int ch;
    switch (ctype) {
    case 0: ch = 1; break;
    case 2: ch = 3; break;
    case 3: ch = 1; break;
    case 4: ch = 2; break;
    case 6: ch = 4; break;
    default: return "bad PNG colour type";
    }
    const bool sub = depth == 1 || depth == 2 || depth == 4;   // packed samples: gray / palette only
    if (!(depth == 8 || (depth == 16 && ctype != 3) || (sub && (ctype == 0 || ctype == 3)))) return "unsupported PNG bit depth";
    if (ctype == 3 && plte.empty()) return "palette PNG without PLTE";
    const int bpp = sub ? 1 : ch * depth / 8;                   // filter distance in bytes
    auto row_bytes = [&](int pw) { return sub ? ((size_t)pw * depth + 7) / 8 : (size_t)pw * bpp; };
    const size_t rowb = row_bytes(w);
    // Adam7 (the binary's stb_image reads interlaced files; ffmpeg never writes them): seven passes, each a smaller image with its
    // own filtered scanlines; pass p holds the pixels (x0 + i * dx, y0 + j * dy)
    static const int ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1}, adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
    struct Pass { int x0, y0, dx, dy, pw, ph; };
    std::vector<Pass> passes;
    if (!interlace) passes.push_back({0, 0, 1, 1, w, h});
    else
        for (int k = 0; k < 7; ++k) {
            const int pw = (w - ax0[k] + adx[k] - 1) / adx[k], ph = (h - ay0[k] + ady[k] - 1) / ady[k];
            if (pw > 0 && ph > 0) passes.push_back({ax0[k], ay0[k], adx[k], ady[k], pw, ph});
        }
    // IHDR alone may claim 65535 x 65535 x 8 B: do not allocate what the IDAT stream cannot possibly inflate to
    // (deflate expands by at most ~1032x) — a 100-byte file must not cost 34 GB
    unsigned long long raw_bytes = 0;
    for (const Pass& ps : passes) raw_bytes += (unsigned long long)(row_bytes(ps.pw) + 1) * (unsigned long long)ps.ph;
    if (raw_bytes > (unsigned long long)idat_len * 1032ull + 65536ull) return "PNG image data shorter than its header claims";
    // scratch that keeps its capacity from frame to frame: a fresh multi-megabyte vector per frame is an mmap + page faults +
    // munmap per frame, and with 70 codec threads in one process those serialise on the address-space lock
    static thread_local std::vector<uint8_t> raw;
    raw.resize((size_t)raw_bytes);
    // the library's own inflate (fastinflate.h): a third of zlib's time on the literal-heavy streams of decoded video
    if (!idat || !fast_zlib_uncompress(idat, idat_len, raw.data(), raw.size()).empty()) return "PNG inflate failed";
    uint8_t* const rgb = sink(w, h);
    if (!rgb) return "no buffer for the decoded frame";
    if (ctype == 2 && depth == 8 && !interlace && !(alpha && trns.size() >= 6)) {
        // the frames reve exports (reve-shared/src/lib.rs:93: ffmpeg's rgb24 PNGs): scanlines un-filtered straight into the
        // destination — the caller's pinned buffer in directory mode — one pass, no intermediate image
        for (int y = 0; y < h; ++y) {
            const uint8_t* line = &raw[(rowb + 1) * y];
            uint8_t* dst = rgb + (size_t)y * rowb;
            if (!unfilter_rgb8(line[0], line + 1, dst, y ? dst - rowb : nullptr, rowb)) return "bad PNG filter type";
        }
        return "";
    }
    // every other layout: per pass, un-filter in place, then pixel by pixel to RGB (+ alpha).  Transparency: an alpha channel (colour
    // types 4, 6), or a tRNS chunk — per palette entry, or one colour key for gray / RGB images (the binary's stb_image turns both
    // into an alpha channel) — 16-bit samples keep their high byte (stb: v >> 8; the key is compared at full width first)
    const int step = depth / 8;
    const bool has_alpha = ctype == 4 || ctype == 6 || (ctype == 3 && !trns.empty()) || (ctype == 0 && trns.size() >= 2) || (ctype == 2 && trns.size() >= 6);
    if (alpha && has_alpha) alpha->assign((size_t)w * h, 255);
    uint8_t* const ap = (alpha && has_alpha) ? alpha->data() : nullptr;
    size_t at = 0;
    for (const Pass& ps : passes) {
        const size_t prb = row_bytes(ps.pw);
        std::vector<uint8_t> zero(prb, 0);
        const uint8_t* prev = zero.data();
        for (int j = 0; j < ps.ph; ++j) {
            uint8_t* row = &raw[at + (prb + 1) * (size_t)j + 1];
            switch (row[-1]) {
            case 0: break;
            case 1: for (size_t i = bpp; i < prb; ++i) row[i] = (uint8_t)(row[i] + row[i - bpp]); break;
            case 2: for (size_t i = 0; i < prb; ++i) row[i] = (uint8_t)(row[i] + prev[i]); break;
            case 3:
                for (size_t i = 0; i < prb; ++i) row[i] = (uint8_t)(row[i] + (((i >= (size_t)bpp ? row[i - bpp] : 0) + prev[i]) >> 1));
                break;
            case 4:
                for (size_t i = 0; i < prb; ++i)
                    row[i] = (uint8_t)(row[i] + paeth(i >= (size_t)bpp ? row[i - bpp] : 0, prev[i], i >= (size_t)bpp ? prev[i - bpp] : 0));
                break;
            default: return "bad PNG filter type";
            }
            prev = row;
            const int y = ps.y0 + j * ps.dy;
            for (int i = 0; i < ps.pw; ++i) {
                const int x = ps.x0 + i * ps.dx;
                uint8_t* o = &rgb[((size_t)y * w + x) * 3];
                const uint8_t* p = row + (size_t)i * bpp;
                unsigned sample = 0;                 // the raw value of a packed sample
                uint8_t packed = 0;
                if (sub) {   // extract the i-th `depth`-bit sample, MSB first
                    const size_t bit = (size_t)i * depth;
                    sample = (unsigned)((row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1));
                    packed = ctype == 0 ? (uint8_t)(sample * 255 / ((1u << depth) - 1)) : (uint8_t)sample;
                    p = &packed;
                }
                uint8_t a = 255;
                switch (ctype) {
                case 0:
                    o[0] = o[1] = o[2] = p[0];
                    if (trns.size() >= 2) {
                        const unsigned key = ((unsigned)trns[0] << 8) | trns[1];
                        const unsigned v = sub ? sample : (depth == 16 ? (((unsigned)p[0] << 8) | p[1]) : p[0]);
                        if (v == key) a = 0;
                    }
                    break;
                case 4: o[0] = o[1] = o[2] = p[0]; a = p[step]; break;
                case 2:
                    o[0] = p[0]; o[1] = p[step]; o[2] = p[2 * step];
                    if (trns.size() >= 6) {
                        bool same = true;
                        for (int c = 0; c < 3; ++c) {
                            const unsigned key = ((unsigned)trns[2 * c] << 8) | trns[2 * c + 1];
                            const unsigned v = depth == 16 ? (((unsigned)p[c * 2] << 8) | p[c * 2 + 1]) : p[c];
                            same = same && v == key;
                        }
                        if (same) a = 0;
                    }
                    break;
                case 6: o[0] = p[0]; o[1] = p[step]; o[2] = p[2 * step]; a = p[3 * step]; break;
                case 3: {
                    const size_t k = (size_t)p[0] * 3;
                    if (k + 2 >= plte.size()) return "palette index out of range";
                    o[0] = plte[k]; o[1] = plte[k + 1]; o[2] = plte[k + 2];
                    if (p[0] < trns.size()) a = trns[p[0]];
                } break;
                }
                if (ap) ap[(size_t)y * w + x] = a;
            }
        }
        at += (prb + 1) * (size_t)ps.ph;
    }
    return "";
}

// o[i] = a[i] - b[i]: the Up filter of a scanline.  (Written as a function over restrict pointers: inside the encoder's row callback
// the compiler cannot prove that the three rows do not overlap and leaves the loop scalar — 12 ms per 4K frame instead of 3.)
static void sub_rows(uint8_t* __restrict o, const uint8_t* __restrict a, const uint8_t* __restrict b, size_t n)
{
    for (size_t i = 0; i < n; ++i) o[i] = (uint8_t)(a[i] - b[i]);
}

static void chunk(std::vector<uint8_t>& f, const char* type, const uint8_t* data, size_t len)
{
    put32(f, (uint32_t)len);
    const size_t at = f.size();
    f.insert(f.end(), type, type + 4);
    if (len) f.insert(f.end(), data, data + len);
    put32(f, fast_crc32(0, &f[at], len + 4));
}

static std::string encode_impl(const uint8_t* rgb, int w, int h, size_t stride, int level, std::vector<uint8_t>& file);

// RGBA8 from an interleaved RGB image and a separate alpha plane (the single-file path of an image with transparency): Up filter,
// the fast path's deflate encoder, one IDAT chunk
std::string png_encode_rgba8(const uint8_t* rgb, const uint8_t* alpha, int w, int h, std::vector<uint8_t>& file)
{
    if (!rgb || !alpha || w <= 0 || h <= 0) return "bad image";
    try {
        const size_t rowb = (size_t)w * 4;
        std::vector<uint8_t> prev(rowb, 0), cur(rowb);
        const size_t zn = fast_zlib_compress_rows((size_t)h, rowb + 1, [&](uint8_t* o, size_t y0, size_t k) {
            for (size_t y = y0; y < y0 + k; ++y, o += rowb + 1) {
                const uint8_t* r = rgb + y * (size_t)w * 3;
                const uint8_t* a = alpha + y * (size_t)w;
                for (int x = 0; x < w; ++x) { cur[4 * x] = r[3 * x]; cur[4 * x + 1] = r[3 * x + 1]; cur[4 * x + 2] = r[3 * x + 2]; cur[4 * x + 3] = a[x]; }
                o[0] = 2;
                for (size_t i = 0; i < rowb; ++i) o[1 + i] = (uint8_t)(cur[i] - prev[i]);
                prev.swap(cur);
            }
        }, file, 41);
        if (!zn) return "PNG deflate failed";
        uint8_t* f = file.data();
        std::memcpy(f, kSig, 8);
        const uint8_t ihdr[25] = {0, 0, 0, 13, 'I', 'H', 'D', 'R', (uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                                  (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 6, 0, 0, 0, 0, 0, 0, 0};
        std::memcpy(f + 8, ihdr, 25);
        const uint32_t c1 = fast_crc32(0, f + 12, 17);
        f[29] = c1 >> 24; f[30] = c1 >> 16; f[31] = c1 >> 8; f[32] = c1;
        f[33] = zn >> 24; f[34] = zn >> 16; f[35] = zn >> 8; f[36] = zn;
        std::memcpy(f + 37, "IDAT", 4);
        const uint32_t c2 = fast_crc32(0, f + 37, zn + 4);
        static const uint8_t iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xae, 0x42, 0x60, 0x82};
        file.resize(41 + zn + 4 + 12);
        f = file.data();
        f[41 + zn] = c2 >> 24; f[42 + zn] = c2 >> 16; f[43 + zn] = c2 >> 8; f[44 + zn] = c2;
        std::memcpy(f + 45 + zn, iend, 12);
        return "";
    } catch (const std::bad_alloc&) {
        return "out of memory encoding PNG";
    }
}

std::string png_encode_rgb8(const uint8_t* rgb, int w, int h, size_t stride, int level, std::vector<uint8_t>& file)
{
    try {
        return encode_impl(rgb, w, h, stride, level, file);
    } catch (const std::bad_alloc&) {
        return "out of memory encoding PNG";
    }
}

static std::string encode_impl(const uint8_t* rgb, int w, int h, size_t stride, int level, std::vector<uint8_t>& file)
{
    if (!rgb || w <= 0 || h <= 0) return "bad image";
    const size_t rowb = (size_t)w * 3;
    static thread_local std::vector<uint8_t> filt, z;     // per-thread scratch, see png_decode_rgb8
    uLongf zcap = 0;
    if (level <= 1) {
        // fast path (directory mode): a fixed filter, Up (Sub on the first row) — on upscaled frames it compresses as well as the
        // per-row search below and takes a third of its time — written straight into the window of the fast path's own deflate
        // encoder (fastdeflate.h: a third to a sixth of zlib level 1's time at its ratio) as it asks for rows: the filtered image
        // never exists as a whole
        // ... and its output goes straight into the file image, behind the signature, IHDR and the IDAT chunk's header (41 bytes)
        zcap = (uLongf)fast_zlib_compress_rows((size_t)h, rowb + 1, [&](uint8_t* o, size_t y0, size_t k) {
            for (size_t y = y0; y < y0 + k; ++y, o += rowb + 1) {
                const uint8_t* row = rgb + y * stride;
                if (y) {
                    const uint8_t* prev = row - stride;
                    o[0] = 2;
                    sub_rows(o + 1, row, prev, rowb);
                } else {
                    o[0] = 1;
                    for (size_t i = 0; i < rowb; ++i) o[1 + i] = (uint8_t)(row[i] - (i >= 3 ? row[i - 3] : 0));
                }
            }
        }, file, 41);
        if (!zcap) return "PNG deflate failed";
        uint8_t* f = file.data();
        std::memcpy(f, kSig, 8);
        const uint8_t ihdr[25] = {0, 0, 0, 13, 'I', 'H', 'D', 'R', (uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                                  (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 2, 0, 0, 0, 0, 0, 0, 0};
        std::memcpy(f + 8, ihdr, 25);
        const uint32_t c1 = fast_crc32(0, f + 12, 17);
        f[29] = c1 >> 24; f[30] = c1 >> 16; f[31] = c1 >> 8; f[32] = c1;
        f[33] = zcap >> 24; f[34] = zcap >> 16; f[35] = zcap >> 8; f[36] = zcap;
        std::memcpy(f + 37, "IDAT", 4);
        const uint32_t c2 = fast_crc32(0, f + 37, zcap + 4);
        static const uint8_t iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xae, 0x42, 0x60, 0x82};
        // (the vector keeps its worst-case capacity from frame to frame; only its size is set to the file's)
        file.resize(41 + zcap + 4 + 12);
        f = file.data();
        f[41 + zcap] = c2 >> 24; f[42 + zcap] = c2 >> 16; f[43 + zcap] = c2 >> 8; f[44 + zcap] = c2;
        std::memcpy(f + 45 + zcap, iend, 12);
        return "";
    } else {
        filt.resize((rowb + 1) * h);
        std::vector<uint8_t> cand[3];
        for (auto& c : cand) c.resize(rowb);
        std::vector<uint8_t> zero(rowb, 0);
        for (int y = 0; y < h; ++y) {
            const uint8_t* row = rgb + (size_t)y * stride;
            const uint8_t* prev = y ? rgb + (size_t)(y - 1) * stride : zero.data();
            uint8_t* o = &filt[(rowb + 1) * y];
            // candidates: Sub, Up, Paeth (plus None); pick the least sum of |signed residual|
            long best = 0;
            int bt = 0;
            for (size_t i = 0; i < rowb; ++i) best += (int8_t)row[i] < 0 ? -(int8_t)row[i] : (int8_t)row[i];
            long s[3] = {0, 0, 0};
            for (size_t i = 0; i < rowb; ++i) {
                const int a = i >= 3 ? row[i - 3] : 0, b = prev[i], c = i >= 3 ? prev[i - 3] : 0;
                const uint8_t v0 = (uint8_t)(row[i] - a), v1 = (uint8_t)(row[i] - b), v2 = (uint8_t)(row[i] - paeth(a, b, c));
                cand[0][i] = v0; cand[1][i] = v1; cand[2][i] = v2;
                s[0] += (int8_t)v0 < 0 ? -(int8_t)v0 : (int8_t)v0;
                s[1] += (int8_t)v1 < 0 ? -(int8_t)v1 : (int8_t)v1;
                s[2] += (int8_t)v2 < 0 ? -(int8_t)v2 : (int8_t)v2;
            }
            static const int ftype[3] = {1, 2, 4};
            const uint8_t* src = row;
            for (int k = 0; k < 3; ++k)
                if (s[k] < best) { best = s[k]; bt = ftype[k]; src = cand[k].data(); }
            o[0] = (uint8_t)bt;
            std::memcpy(o + 1, src, rowb);
        }
        zcap = compressBound(filt.size());
        if (z.size() < zcap) z.resize(zcap);
        if (compress2(z.data(), &zcap, filt.data(), filt.size(), level) != Z_OK) return "PNG deflate failed";
    }
    file.clear();
    file.insert(file.end(), kSig, kSig + 8);
    uint8_t ihdr[13];
    ihdr[0] = w >> 24; ihdr[1] = w >> 16; ihdr[2] = w >> 8; ihdr[3] = w;
    ihdr[4] = h >> 24; ihdr[5] = h >> 16; ihdr[6] = h >> 8; ihdr[7] = h;
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk(file, "IHDR", ihdr, 13);
    chunk(file, "IDAT", z.data(), zcap);
    chunk(file, "IEND", nullptr, 0);
    return "";
}

}  // namespace reve
