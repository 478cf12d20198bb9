// A deflate (RFC 1951) / zlib (RFC 1950) ENCODER for PNG scanline data, written for throughput: directory mode has to
// encode a 25 MB 4K frame per upscaled frame, and with zlib level 1 (50-75 ms of CPU per frame) the encoders, not the
// GPU, bound the mode as soon as the host offers fewer than ~30 cores to it (the GPU boxes of this project give a
// process 16: DESIGN.md §7).  One-probe hash, greedy parse, matches extended eight bytes at a time, dynamic Huffman
// blocks of up to 32 K tokens, stored blocks where they are smaller (incompressible input goes through at memcpy
// speed), vectorised Adler-32.  The output is an ordinary zlib stream: any inflate reads it (tests: zlib, Pillow, the
// library's own decoder).  Decoding still uses zlib.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace reve {

// Writes the zlib stream of src[0, n) to the front of `out` and returns its length (0: failure, only possible for inputs of
// 4 GB and more, which go through zlib).  `out` is scratch: it is grown to the worst-case size when too small and never
// shrunk, so a per-thread vector costs no allocation or clearing per frame.  Deterministic: the same input gives the same
// bytes on every machine and thread.
size_t fast_zlib_compress(const uint8_t* src, size_t n, std::vector<uint8_t>& out);

// Adler-32 of buf[0, n) continued from `adler` (1 for a new stream), as zlib's adler32(): AVX2 when the CPU has it.
uint32_t fast_adler32(uint32_t adler, const uint8_t* buf, size_t n);

}  // namespace reve
