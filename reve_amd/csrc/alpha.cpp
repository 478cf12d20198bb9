#include "alpha.h"

#include <algorithm>
#include <cmath>
#include <vector>

#include "model.h"      // f32_to_f16 / f16_to_f32

namespace reve {

namespace {
// the four taps of output position d along an axis of n input samples: first tap index (may be negative: clamped when used) and weights
inline int cubic_taps(int d, int scale, float c[4])
{
    const float A = -0.75f;
    const float f = (float)(((double)d + 0.5) / (double)scale - 0.5);
    const int s = (int)std::floor(f);
    const float t = f - (float)s;
    const float t0 = t + 1.f, t1 = t, t2 = 1.f - t;
    c[0] = A * t0 * t0 * t0 - 5.f * A * t0 * t0 + 8.f * A * t0 - 4.f * A;
    c[1] = (A + 2.f) * t1 * t1 * t1 - (A + 3.f) * t1 * t1 + 1.f;
    c[2] = (A + 2.f) * t2 * t2 * t2 - (A + 3.f) * t2 * t2 + 1.f;
    c[3] = 1.f - c[0] - c[1] - c[2];
    return s - 1;
}
}  // namespace

void alpha_bicubic(const uint8_t* a, int w, int h, int scale, uint8_t* out)
{
    const int W = w * scale, H = h * scale;
    // pre-process: alpha / 255 as the binary stores it (fp16)
    std::vector<float> in((size_t)w * h);
    for (size_t i = 0; i < in.size(); ++i) in[i] = f16_to_f32(f32_to_f16((float)a[i] * (1.0f / 255.0f)));
    std::vector<int> x0(W);
    std::vector<float> cx((size_t)W * 4);
    for (int x = 0; x < W; ++x) x0[x] = cubic_taps(x, scale, &cx[(size_t)x * 4]);
    // rows interpolated along x, kept while output rows need them (four input rows per output row)
    std::vector<float> rows((size_t)h * W);
    for (int y = 0; y < h; ++y) {
        const float* r = &in[(size_t)y * w];
        float* o = &rows[(size_t)y * W];
        for (int x = 0; x < W; ++x) {
            const float* c = &cx[(size_t)x * 4];
            float v = 0.f;
            for (int k = 0; k < 4; ++k) v += r[std::min(std::max(x0[x] + k, 0), w - 1)] * c[k];
            o[x] = v;
        }
    }
    for (int y = 0; y < H; ++y) {
        float c[4];
        const int y0 = cubic_taps(y, scale, c);
        const float* r[4];
        for (int k = 0; k < 4; ++k) r[k] = &rows[(size_t)std::min(std::max(y0 + k, 0), h - 1) * W];
        uint8_t* o = out + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            const float v = f16_to_f32(f32_to_f16(r[0][x] * c[0] + r[1][x] * c[1] + r[2][x] * c[2] + r[3][x] * c[3]));
            const float q = v * 255.0f + 0.5f;
            o[x] = (uint8_t)(q <= 0.f ? 0 : (q >= 255.f ? 255 : (int)q));
        }
    }
}

}  // namespace reve
