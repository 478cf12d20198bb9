#include "model.h"
#include <algorithm>
#include "kernels.h"

#include <cstdio>
#include <cmath>
#include <cstring>
#include <exception>
#include <fstream>
#include <map>
#include <sstream>

namespace reve {

uint16_t f32_to_f16(float f)
{
    _Float16 h = (_Float16)f;   // round-to-nearest-even, same as v_cvt_f16_f32
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}

float f16_to_f32(uint16_t b)
{
    _Float16 h;
    std::memcpy(&h, &b, 2);
    return (float)h;
}

namespace {

struct Cursor {
    const uint8_t* p;
    size_t len, off = 0;
    bool take(void* dst, size_t n)
    {
        if (n > len - off) return false;   // (off + n could wrap for a hostile n)
        std::memcpy(dst, p + off, n);
        off += n;
        return true;
    }
};

// ncnn ModelBin::load(w, type=0): u32 tag, 0x01306B47 -> fp16 payload (padded to 4 B), 0 -> raw fp32
std::string read_weights(Cursor& c, size_t n, std::vector<float>& out)
{
    uint32_t tag;
    if (!c.take(&tag, 4)) return "truncated .bin (weight tag)";
    // sizes come from the .param text: never allocate more than the .bin can still deliver
    if (n > (c.len - c.off) / 2) return "truncated .bin (weights)";
    out.resize(n);
    if (tag == 0x01306B47u) {
        std::vector<uint16_t> h(n);
        if (!c.take(h.data(), n * 2)) return "truncated .bin (fp16 weights)";
        c.off = (c.off + 3) & ~(size_t)3;
        for (size_t i = 0; i < n; ++i) out[i] = f16_to_f32(h[i]);
    } else if (tag == 0) {
        if (!c.take(out.data(), n * 4)) return "truncated .bin (fp32 weights)";
    } else {
        char buf[64];
        std::snprintf(buf, sizeof buf, "unsupported weight storage tag 0x%08x", tag);
        return buf;
    }
    return "";
}

std::string read_raw(Cursor& c, size_t n, std::vector<float>& out)
{
    if (n > (c.len - c.off) / 4) return "truncated .bin (fp32 vector)";
    out.resize(n);
    if (!c.take(out.data(), n * 4)) return "truncated .bin (fp32 vector)";
    return "";
}

struct Layer {
    std::string type;
    std::map<int, std::string> kv;
    int geti(int k, int def = 0) const
    {
        auto it = kv.find(k);
        return it == kv.end() ? def : std::atoi(it->second.c_str());
    }
    double getf(int k, double def = 0) const
    {
        auto it = kv.find(k);
        return it == kv.end() ? def : std::atof(it->second.c_str());
    }
};

}  // namespace

static std::string parse_impl(const std::string& param_text, const uint8_t* bin, size_t bin_len, Model& m);

// Model files are untrusted input: every malformed file is an error text, never an exception or a crash.
std::string parse_ncnn(const std::string& param_text, const uint8_t* bin, size_t bin_len, Model& m)
{
    try {
        return parse_impl(param_text, bin, bin_len, m);
    } catch (const std::exception& e) {
        return std::string("model parse failed: ") + e.what();
    }
}

static std::string parse_impl(const std::string& param_text, const uint8_t* bin, size_t bin_len, Model& m)
{
    std::istringstream in(param_text);
    std::string line;
    if (!std::getline(in, line) || line.find("7767517") == std::string::npos) return "bad .param magic";
    if (!std::getline(in, line)) return "truncated .param";
    std::vector<Layer> layers;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        Layer L;
        std::string name;
        int nin = 0, nout = 0;
        if (!(ls >> L.type >> name >> nin >> nout)) continue;
        if (nin < 0 || nout < 0 || nin > 64 || nout > 64) return "unreasonable blob count in .param";
        std::string tok;
        for (int i = 0; i < nin + nout; ++i) ls >> tok;
        while (ls >> tok) {
            auto eq = tok.find('=');
            if (eq != std::string::npos && eq > 0) L.kv[std::atoi(tok.substr(0, eq).c_str())] = tok.substr(eq + 1);
        }
        layers.push_back(L);
    }

    // Walk the graph in file order; the only layers with data in .bin are Convolution and PReLU.
    Cursor c{bin, bin_len};
    m = Model();
    struct Conv { int co, wsize; std::vector<float> w, b; };
    std::vector<Conv> convs;
    std::vector<std::vector<float>> prelus;
    bool saw_interp = false, saw_add = false;
    for (const Layer& L : layers) {
        if (L.type == "Convolution") {
            if (L.geti(1) != 3 || L.geti(11, L.geti(1)) != 3 || L.geti(3, 1) != 1 || L.geti(2, 1) != 1 || L.geti(4) != 1)
                return "Convolution is not 3x3 stride 1 pad 1";
            if (L.geti(5) != 1) return "Convolution without bias";
            if (L.geti(9) != 0) return "Convolution with fused activation is not part of SRVGGNetCompact";
            Conv cv;
            cv.co = L.geti(0);
            cv.wsize = L.geti(6);
            if (cv.co <= 0 || cv.wsize <= 0) return "Convolution with a non-positive size";
            std::string e = read_weights(c, (size_t)cv.wsize, cv.w);
            if (!e.empty()) return e;
            e = read_raw(c, (size_t)cv.co, cv.b);
            if (!e.empty()) return e;
            convs.push_back(std::move(cv));
        } else if (L.type == "PReLU") {
            std::vector<float> a;
            if (L.geti(0) <= 0) return "PReLU with a non-positive size";
            std::string e = read_raw(c, (size_t)L.geti(0), a);
            if (!e.empty()) return e;
            prelus.push_back(std::move(a));
        } else if (L.type == "PixelShuffle") {
            m.scale = L.geti(0);
            if (L.geti(1, 0) != 0) return "PixelShuffle mode 1 (TensorFlow order) unsupported";
        } else if (L.type == "Interp") {
            if (L.geti(0) != 1) return "Interp is not nearest";
            saw_interp = true;
        } else if (L.type == "BinaryOp") {
            if (L.geti(0) != 0) return "BinaryOp is not add";
            saw_add = true;
        } else if (L.type == "Input" || L.type == "Split") {
        } else {
            return "unexpected layer type '" + L.type + "' (not an SRVGGNetCompact graph)";
        }
    }
    if (c.off != bin_len) return "trailing bytes in .bin";
    if (convs.size() < 3 || prelus.size() != convs.size() - 1 || !saw_interp || !saw_add) return "not an SRVGGNetCompact graph";
    if (m.scale < 2 || m.scale > 4) return "upscale factor must be 2, 3 or 4";
    m.feat = convs[0].co;
    m.n_body = (int)convs.size() - 2;
    m.co_last = convs.back().co;
    if (m.feat != FEAT) return "num_feat must be 64";
    if (m.n_body != 16) return "num_conv must be 16 (realesr-animevideov3)";
    if (m.co_last != 3 * m.scale * m.scale) return "conv_last channels != 3*scale^2";
    if (convs[0].wsize != FEAT * 3 * 9) return "conv_first is not 3->64";
    for (int l = 0; l < m.n_body; ++l)
        if (convs[1 + l].co != FEAT || convs[1 + l].wsize != FEAT * FEAT * 9) return "body conv is not 64->64";
    if (convs.back().wsize != m.co_last * FEAT * 9) return "conv_last is not 64->3*scale^2";
    for (auto& a : prelus)
        if ((int)a.size() != FEAT) return "PReLU must have 64 slopes";
    m.w_first = convs[0].w; m.b_first = convs[0].b; m.a_first = prelus[0];
    for (int l = 0; l < m.n_body; ++l) {
        m.w_body.push_back(convs[1 + l].w);
        m.b_body.push_back(convs[1 + l].b);
        m.a_body.push_back(prelus[1 + l]);
    }
    m.w_last = convs.back().w; m.b_last = convs.back().b;
    return "";
}

std::string load_ncnn_files(const std::string& dir, const std::string& name, Model& out)
{
    const std::string pp = dir + "/" + name + ".param", bp = dir + "/" + name + ".bin";
    std::ifstream pf(pp), bf(bp, std::ios::binary);
    if (!pf) return "cannot open " + pp;
    if (!bf) return "cannot open " + bp;
    try {
        std::stringstream ps;
        ps << pf.rdbuf();
        std::vector<uint8_t> bin((std::istreambuf_iterator<char>(bf)), std::istreambuf_iterator<char>());
        return parse_ncnn(ps.str(), bin.data(), bin.size(), out);
    } catch (const std::exception& e) {
        return "cannot read " + bp + ": " + e.what();
    }
}

// ---------------------------------------------------------------------------------------------
// A fragment of v_mfma_f32_16x16x32_f16: lane l holds A[row = l&15][k = 8*(l>>4) + j], j = 0..7.
// Row = output channel inside the co-block; k walks the 32 PHYSICAL input-channel positions of one
// tap half (see chan_phys() in kernels.h), so that the B fragment is one ds_read_b128 of the pixel.
// ---------------------------------------------------------------------------------------------
// chan_of_row[16*m + row] = the layer's output channel computed by row `row` of co-block m (-1: zero row).
static PackedLayer pack_conv64(const float* w, const float* b, const std::vector<int>& chan_of_row)
{
    const int ncob = (int)chan_of_row.size() / 16;
    PackedLayer P;
    P.ncob = ncob;
    P.ksteps = KSTEPS;
    P.wpack.assign((size_t)KSTEPS * ncob * 64 * 8, 0);
    P.bias.assign((size_t)ncob * 16, 0);
    for (int tap = 0; tap < 9; ++tap)
        for (int hf = 0; hf < 2; ++hf)
            for (int m = 0; m < ncob; ++m)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int co = chan_of_row[16 * m + (lane & 15)];
                        const int ci = chan_logical(32 * hf + 8 * (lane >> 4) + j);
                        const float v = co >= 0 ? w[((size_t)co * FEAT + ci) * 9 + tap] : 0.f;
                        P.wpack[((((size_t)(tap * 2 + hf) * ncob + m) * 64) + lane) * 8 + j] = f32_to_f16(v);
                    }
    for (size_t r = 0; r < chan_of_row.size(); ++r)
        if (chan_of_row[r] >= 0) P.bias[r] = f32_to_f16(b[chan_of_row[r]]);
    return P;
}

static std::vector<int> natural_rows(int co_real, int ncob)
{
    std::vector<int> rows((size_t)ncob * 16, -1);
    for (int c = 0; c < co_real; ++c) rows[c] = c;
    return rows;
}

// (one walk for the estimate and for the report a maintainer reads: `trace`, when given, receives one line of JSON per layer)
static double conditioning_walk(const Model& m, std::string* trace)
{
    auto sq = [](const std::vector<float>& v) { double t = 0; for (float x : v) t += (double)f16_to_f32(f32_to_f16(x)) * f16_to_f32(f32_to_f16(x)); return t; };
    auto slopes = [](const std::vector<float>& a, double& lo, double& hi) {
        lo = 1e30; hi = -1e30;
        for (float x : a) { lo = std::min(lo, (double)x); hi = std::max(hi, (double)x); }
        if (a.empty()) lo = hi = 0;
    };
    bool first_line = true;
    auto note = [&](const char* name, const std::vector<float>& w, const std::vector<float>& b, const std::vector<float>* a, double s2_out) {
        if (!trace) return;
        const double co = (double)std::max<size_t>(b.size(), 1);
        char line[320];
        double lo = 0, hi = 0;
        if (a) slopes(*a, lo, hi);
        std::snprintf(line, sizeof line, "%s\n  {\"layer\": \"%s\", \"gain\": %.6g, \"weight_rms\": %.6g, \"bias_rms\": %.6g, \"slope_min\": %.6g, \"slope_max\": %.6g, \"activation_rms_out\": %.6g}",
                      first_line ? "" : ",", name, std::sqrt(sq(w) / co), std::sqrt(sq(w) / (double)std::max<size_t>(w.size(), 1)), std::sqrt(sq(b) / co), lo, hi, std::sqrt(s2_out));
        first_line = false;
        *trace += line;
    };
    auto conv = [&](const std::vector<float>& w, const std::vector<float>& b, double s2) {
        const double co = (double)std::max<size_t>(b.size(), 1);
        return sq(w) / co * s2 + sq(b) / co;
    };
    auto prelu = [&](const std::vector<float>& a, double s2) { return s2 * (1.0 + sq(a) / (double)std::max<size_t>(a.size(), 1)) / 2.0; };
    double s2 = prelu(m.a_first, conv(m.w_first, m.b_first, 1.0 / 3.0));
    note("conv_first", m.w_first, m.b_first, &m.a_first, s2);
    for (int l = 0; l < m.n_body; ++l) {
        s2 = prelu(m.a_body[l], conv(m.w_body[l], m.b_body[l], s2));
        note(("body" + std::to_string(l)).c_str(), m.w_body[l], m.b_body[l], &m.a_body[l], s2);
    }
    const double g_last = std::sqrt(sq(m.w_last) / (double)std::max<size_t>(m.b_last.size(), 1));
    note("conv_last", m.w_last, m.b_last, nullptr, conv(m.w_last, m.b_last, s2));
    const double blobs = 2.0 * (m.n_body + 1);
    return 255.0 * g_last * std::sqrt(s2) * std::ldexp(1.0, -11) * std::sqrt(blobs);
}

double conditioning_kappa(const Model& m) { return conditioning_walk(m, nullptr); }

std::string conditioning_report_json(const Model& m, const std::string& model_name)
{
    std::string layers;
    const double kappa = conditioning_walk(m, &layers);
    // (the name is the caller's: quotes and backslashes escaped, control characters dropped)
    std::string name;
    for (char c : model_name) {
        if (c == '"' || c == '\\') { name += '\\'; name += c; }
        else if ((unsigned char)c >= 0x20) name += c;
    }
    char nums[512];
    std::snprintf(nums, sizeof nums,
                  "\"scale\": %d, \"body_layers\": %d, \"features\": %d,\n \"kappa\": %.6g, \"kappa_limit\": %.3g, "
                  "\"kappa_is\": \"fp16 storage noise the weights carry to the 8-bit output, LSB rms (DESIGN.md section 3)\",\n"
                  " \"evaluation_auto_would_choose\": \"%s\",\n \"layers\": [",
                  m.scale, m.n_body, m.feat, kappa, WINOGRAD_KAPPA_LIMIT,
                  kappa < WINOGRAD_KAPPA_LIMIT ? "winograd F(2,3) along the row" : "direct");
    return "{\"model\": \"" + name + "\", " + nums + layers + "\n ]}\n";
}

int last_ncob(int scale) { return scale == 2 ? 1 : (scale == 3 ? 2 : 4); }

PackedLayer pack_body(const Model& m, int layer)
{
    PackedLayer P = pack_conv64(m.w_body[layer].data(), m.b_body[layer].data(), natural_rows(FEAT, 4));
    P.slope.resize(FEAT);
    for (int c = 0; c < FEAT; ++c) P.slope[c] = f32_to_f16(m.a_body[layer][c]);
    return P;
}

PackedLayer pack_body_wino(const Model& m, int layer)
{
    PackedLayer P = pack_body(m, layer);         // bias and slopes as for the direct kernels
    P.ncob = 2;
    P.ksteps = 3 * 4 * 2;
    const float* w = m.w_body[layer].data();
    P.wpack.assign((size_t)2 * 3 * 4 * 2 * 2 * 64 * 8, 0);
    size_t at = 0;
    for (int ch = 0; ch < 2; ++ch)
        for (int dy = 0; dy < 3; ++dy)
            for (int xi = 0; xi < 4; ++xi)
                for (int hf = 0; hf < 2; ++hf)
                    for (int mm = 0; mm < 2; ++mm)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int co = 32 * ch + 16 * mm + (lane & 15);
                                const int ci = chan_logical(32 * hf + 8 * (lane >> 4) + j);
                                const float* t = w + ((size_t)co * FEAT + ci) * 9 + dy * 3;
                                const float g0 = f16_to_f32(f32_to_f16(t[0])), g1 = f16_to_f32(f32_to_f16(t[1])), g2 = f16_to_f32(f32_to_f16(t[2]));
                                const float u = xi == 0 ? g0 : (xi == 1 ? ((g0 + g1) + g2) * 0.5f : (xi == 2 ? ((g0 - g1) + g2) * 0.5f : g2));
                                P.wpack[at++] = f32_to_f16(u);
                            }
    return P;
}

// store_order (conv_last on the body kernel's pipeline, kernels.hip): the output channels are permuted so that the four
// accumulator rows of a lane are four CONSECUTIVE output bytes of one output sub-row.  x4: row 4g+r of co-block m is byte
// 4m+r of the 12-byte run (4 sub-pixels x RGB) the LR pixel contributes to output sub-row g, i.e. channel c*16 + g*4 + j
// with (j, c) = divmod(4m + r, 3); x2 and x3 below.  store_order = false: PyTorch's natural channel order.
std::vector<int> last_rows(int scale, int co_last, bool store_order)
{
    std::vector<int> rows = natural_rows(co_last, last_ncob(scale));
    if (store_order && scale == 2) {
        // x2: the 6 bytes (2 sub-pixels x RGB) an LR pixel contributes to output sub-row i are rows of lane groups 2i (bytes
        // 0..3) and 2i + 1 (bytes 4, 5; its rows 2, 3 stay zero): byte b = sub-pixel column b / 3, colour b % 3
        std::fill(rows.begin(), rows.end(), -1);
        for (int i = 0; i < 2; ++i)
            for (int b = 0; b < 6; ++b) {
                const int g = 2 * i + (b >= 4), r = b >= 4 ? b - 4 : b, j = b / 3, c = b % 3;
                rows[4 * g + r] = c * 4 + i * 2 + j;
            }
    }
    if (store_order && scale == 3) {
        // x3: an LR pixel contributes 9 bytes (3 sub-pixels x RGB) to each of its 3 output sub-rows.  Lane groups g = 0..2 hold
        // bytes 0..3 (co-block 0) and 4..7 (co-block 1) of sub-row g; the ninth byte of sub-row i is row i of co-block 0's
        // group 3.  Byte b = sub-pixel column b / 3, colour b % 3.
        std::fill(rows.begin(), rows.end(), -1);
        for (int i = 0; i < 3; ++i) {
            for (int b = 0; b < 8; ++b) rows[16 * (b >> 2) + 4 * i + (b & 3)] = (b % 3) * 9 + i * 3 + b / 3;
            rows[4 * 3 + i] = 2 * 9 + i * 3 + 2;     // byte 8: sub-pixel column 2, colour 2
        }
    }
    if (store_order && scale == 4) {
        std::fill(rows.begin(), rows.end(), -1);
        for (int cob = 0; cob < 3; ++cob)
            for (int g = 0; g < 4; ++g)
                for (int r = 0; r < 4; ++r) {
                    const int k = 4 * cob + r, j = k / 3, c = k % 3;
                    rows[16 * cob + 4 * g + r] = c * 16 + g * 4 + j;
                }
    }
    return rows;
}

PackedLayer pack_last(const Model& m, bool store_order)
{
    return pack_conv64(m.w_last.data(), m.b_last.data(), last_rows(m.scale, m.co_last, store_order));
}

// conv_first: k = 32*s + 8*(l>>4) + j  <->  tap = k>>2, input channel = k&3 (3 = zero padding)
PackedLayer pack_first(const Model& m)
{
    PackedLayer P;
    P.ncob = 4;
    P.ksteps = 2;
    P.wpack.assign((size_t)2 * 4 * 64 * 8, 0);
    for (int s = 0; s < 2; ++s)
        for (int mb = 0; mb < 4; ++mb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * s + 8 * (lane >> 4) + j, tap = k >> 2, ci = k & 3;
                    const int co = 16 * mb + (lane & 15);
                    const float v = (tap < 9 && ci < 3) ? m.w_first[((size_t)co * 3 + ci) * 9 + tap] : 0.f;
                    P.wpack[((((size_t)s * 4 + mb) * 64) + lane) * 8 + j] = f32_to_f16(v);
                }
    P.bias.resize(FEAT);
    P.slope.resize(FEAT);
    for (int c = 0; c < FEAT; ++c) {
        P.bias[c] = f32_to_f16(m.b_first[c]);
        P.slope[c] = f32_to_f16(m.a_first[c]);
    }
    return P;
}

}  // namespace reve
