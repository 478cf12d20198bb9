// ncnn .param/.bin loader for SRVGGNetCompact (realesr-animevideov3-x{2,3,4}) and the repacking of
// its weights into the MFMA A-fragment order the kernels keep in registers.
//
// reve names the model on the child's command line (reve-shared/src/lib.rs:140-141 `-n
// realesr-animevideov3-x2`; reve-gui/src-tauri/src/commands.rs:58-61 `-m models -n <type>-x<f>`);
// the binary then reads <model dir>/<name>.param and .bin.  Format: SURVEY.md §2.3.3.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace reve {

struct Model {
    int scale = 0, n_body = 0, feat = 0, co_last = 0;
    std::vector<float> w_first, b_first, a_first;                 // [64][3][3][3], [64], [64]
    std::vector<std::vector<float>> w_body, b_body, a_body;       // per layer [64][64][3][3], [64], [64]
    std::vector<float> w_last, b_last;                            // [3s^2][64][3][3], [3s^2]
};

// Parses the two files' contents. Returns "" on success, else an error text.
std::string parse_ncnn(const std::string& param_text, const uint8_t* bin, size_t bin_len, Model& out);
std::string load_ncnn_files(const std::string& dir, const std::string& name, Model& out);

// Packed device images (fp16 bit patterns).
struct PackedLayer {
    std::vector<uint16_t> wpack;   // [ksteps][ncob][64 lanes][8]
    std::vector<uint16_t> bias;    // [ncob*16] logical order, zero padded
    std::vector<uint16_t> slope;   // [64] (empty for conv_last)
    int ncob = 0, ksteps = 0;
};
int last_ncob(int scale);                               // co-blocks of conv_last as launched: 1, 2, 4
PackedLayer pack_first(const Model& m);
PackedLayer pack_body(const Model& m, int layer, bool flip_rows = false);   // flip_rows: tap rows swapped (kernels.h PairArgs::up)
// The same layer for kernels_wino.hip: per tap row dy the three taps g0, g1, g2 of a (co, ci) pair become the four
// Winograd-domain weights U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2 (from the fp16-stored taps, in fp32,
// stored as fp16).  wpack = [output-channel half][tap row][xi][input-channel half][co-block of the half][64 lanes][8].
PackedLayer pack_body_wino(const Model& m, int layer);
PackedLayer pack_last(const Model& m, bool store_order);   // store_order: see model.cpp
// accumulator row (16 * co-block + 4 * lane group + r) -> logical output channel of conv_last, -1 = unused row
std::vector<int> last_rows(int scale, int co_last, bool store_order);

uint16_t f32_to_f16(float f);
float f16_to_f32(uint16_t h);

}  // namespace reve
