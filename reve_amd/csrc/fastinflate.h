// An inflate (RFC 1951) / zlib (RFC 1950) DECODER for PNG scanline data, written for throughput like its counterpart
// fastdeflate.h: directory mode has to decode a 1080p frame per upscaled frame, and the frames reve hands over are decoded
// video — grain — whose streams are mostly literals: zlib's inflate spends 39 ms of one CPU on such a frame (6.2 MB of
// scanlines), a quarter of what the whole pipeline may spend per frame at 400 frames/s on 16 CPUs.  Table-driven, 64-bit bit
// buffer, several literals per refill, word-wise match copies; every index, distance and length checked (the input is
// untrusted).  One-shot: the caller knows the decoded size (a PNG header states it).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace reve {

// Inflates the zlib stream in[0, in_len) into out[0, out_len): "" on success — the stream decoded to EXACTLY out_len bytes and
// its Adler-32 matches — else what was wrong with it.  Never reads outside `in`, never writes outside `out`.
std::string fast_zlib_uncompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len);

}  // namespace reve
