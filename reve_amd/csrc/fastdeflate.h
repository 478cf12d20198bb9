// A deflate (RFC 1951) / zlib (RFC 1950) ENCODER for PNG scanline data, written for throughput: directory mode has to
// encode a 25 MB 4K frame per upscaled frame, and with zlib level 1 (50-75 ms of CPU per frame) the encoders, not the
// GPU, bound the mode as soon as the host offers fewer than ~30 cores to it (the GPU boxes of this project give a
// process 16: DESIGN.md §7).  Two regimes, chosen block by block from what the match search buys:
//   * content with runs (flat shading after the Up filter): one-probe hash with the previous match's distance tried first, greedy
//     parse, matches extended eight bytes at a time and refused when short and far away, literals kept implicit (runs of source
//     bytes between matches), dynamic Huffman blocks of up to 512 KB;
//   * grain and noise (a block whose matches do not at least halve it): the following fifteen blocks carry no matches at all —
//     a sampled byte histogram, 12-bit codes, a pair table (one lookup per two literals) and two interleaved bit streams, the
//     second appended by a word-wise shift — then a 64 KB probe block is parsed again; plain noise goes out as stored blocks.
// The source can be handed over row by row (fast_zlib_compress_rows): the PNG encoder filters its scanlines straight into the
// encoder's window.  Vectorised Adler-32, carry-less-multiplication CRC-32.  One core of the build container, 4K frame: flat
// content 19 ms (zlib level 1: 36-64 ms at the same ratio), upscaled video grain 29 ms at 0.67 of the raw size (zlib: ~400 ms at
// 0.38; round 4's version of this encoder: 174 ms at 0.41), noise 28 ms (674 ms).  The output is an ordinary zlib stream: any
// inflate reads it (tests: zlib, Pillow, the library's own decoder, fastinflate.h).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

namespace reve {

// Writes the zlib stream of src[0, n) to out[offset ...) and returns its length (0: failure, only possible for inputs of
// 4 GB and more, which go through zlib).  `out` is scratch: it is grown to the worst-case size when too small and never
// shrunk, so a per-thread vector costs no allocation or clearing per frame.  Deterministic: the same input gives the same
// bytes on every machine and thread.
size_t fast_zlib_compress(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t offset = 0);

// The same encoder fed ROW BY ROW: `rows` rows of `row_bytes` bytes each, which produce(dst, first_row, n_rows) writes to dst when the
// encoder asks for them (in order, each row once, a few at a time).  The encoder keeps only a window of the stream — 32 KB of history,
// the block in hand, a few rows of lookahead — so a PNG encoder can filter its scanlines straight into it: the filtered image never
// exists in memory as a whole, and the Adler-32 is taken while the rows are in cache.  Same bytes as fast_zlib_compress over the
// concatenated rows.  Returns the stream's length (0: more than 4 GB, or no rows).
size_t fast_zlib_compress_rows(size_t rows, size_t row_bytes, const std::function<void(uint8_t* dst, size_t first_row, size_t n_rows)>& produce,
                               std::vector<uint8_t>& out, size_t offset = 0);

// CRC-32 of buf[0, n) continued from `crc` (0 for a new one), as zlib's crc32(): carry-less multiplication when the CPU has it
// (10+ GB/s instead of zlib 1.2.11's ~1 GB/s: the chunk CRC of a poorly compressible 4K frame was 14 ms of its 49).
uint32_t fast_crc32(uint32_t crc, const uint8_t* buf, size_t n);

// Adler-32 of buf[0, n) continued from `adler` (1 for a new stream), as zlib's adler32(): AVX2 when the CPU has it.
uint32_t fast_adler32(uint32_t adler, const uint8_t* buf, size_t n);

}  // namespace reve
