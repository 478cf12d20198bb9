// The alpha channel of an RGBA input (reve-gui hands the binary whatever image the user picked,
// reve-gui/src-tauri/src/commands.rs:52-65; reve-cli's frames are opaque rgb24).  [UPSTREAM-RECALL] realesrgan.cpp runs the
// network on RGB only and scales the alpha plane beside it with ncnn's Interp layer, resize_type 3 (bicubic, a = -0.75,
// half-pixel centres, edge samples repeated), in the model's numeric mode: alpha / 255 stored as fp16, fp32 arithmetic, the
// result stored as fp16, then the usual clamp(v * 255 + 0.5).  Host code: one plane per image, not part of the frame path.
#pragma once
#include <cstdint>

namespace reve {
// a: w x h bytes -> out: (w * scale) x (h * scale) bytes
void alpha_bicubic(const uint8_t* a, int w, int h, int scale, uint8_t* out);
}  // namespace reve
