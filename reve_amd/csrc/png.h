// Minimal PNG codec over zlib for the directory contract at the boundary
// (reve-shared/src/lib.rs:93 `frame%08d.png` in, reve-cli/src/main.rs:297-300 `frame%08d.png` out).
// The reference binary uses stb_image / stb_image_write; only zlib exists in this image.
// Decode: plain or Adam7-interlaced, bit depth 1-16, gray / RGB / palette / gray+alpha / RGBA (+ tRNS) -> RGB8
// (png_decode_rgba8 keeps the alpha plane), inflate by fastinflate.h.  Encode: RGB8; level <= 1: Up filter + fastdeflate.h (directory mode),
// else adaptive row filters + zlib at `level`.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace reve {
std::string png_decode_rgb8(const std::vector<uint8_t>& file, std::vector<uint8_t>& rgb, int& w, int& h);
// the same with the pixels written where sink(w, h) says (called once, after the header has been checked; w * h * 3 bytes): directory
// mode decodes straight into its pinned upload buffers
std::string png_decode_rgb8_to(const std::vector<uint8_t>& file, const std::function<uint8_t*(int w, int h)>& sink, int& w, int& h);
// ... and with the file's alpha channel (colour types 4 and 6; 16-bit: its high byte) as a plane of its own: w * h bytes, empty when
// the file has none
std::string png_decode_rgba8(const std::vector<uint8_t>& file, std::vector<uint8_t>& rgb, std::vector<uint8_t>& alpha, int& w, int& h);
std::string png_encode_rgba8(const uint8_t* rgb, const uint8_t* alpha, int w, int h, std::vector<uint8_t>& file);
std::string png_encode_rgb8(const uint8_t* rgb, int w, int h, size_t stride, int level, std::vector<uint8_t>& file);
std::string read_file(const std::string& path, std::vector<uint8_t>& out);
std::string write_file(const std::string& path, const std::vector<uint8_t>& data);
}  // namespace reve
