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

// How far the network's fp16 STORAGE roundings reach the 8-bit output, estimated from the weights alone (the rule
// behind option "winograd" = auto).  A mean-square activation is carried through the graph from an input of mean square 1/3
// (uniform [0, 1]): per convolution s2 <- ||W||_F^2 / co * s2 + mean(b^2), per PReLU s2 <- s2 * (1 + mean(a^2)) / 2.  Every
// stored blob (34 of them) is rounded to 11 bits: a relative error of 2^-11 that travels to the output with the signal's own
// gain, so the output's rounding noise is ~ 255 * g_last * sqrt(s2_16) * 2^-11 * sqrt(34) LSB rms, g_last = ||W_last||_F /
// sqrt(co_last).  Measured on the fifteen weight statistics of reve_amd/synth.py (profiles/r04/parity_sweep.txt): the thirteen
// draws on which both evaluation orders stay within 1 LSB of the oracle give 0.013 .. 0.37, the two on which every evaluation
// order (the CPU restatements included) is 6-12 LSB apart give 0.96 and 1.45.  WINOGRAD_KAPPA_LIMIT sits between.
double conditioning_kappa(const Model& m);
// the same walk as text (JSON): per layer its gain ||W||_F / sqrt(co) (rms out per unit rms in), weight and bias rms, slope range and
// the activation rms the estimate carries; kappa, the limit and the evaluation option "winograd" = auto would choose
// (reve_model_report, `realesrgan-hip --model-report`: no GPU needed)
std::string conditioning_report_json(const Model& m, const std::string& model_name);
constexpr double WINOGRAD_KAPPA_LIMIT = 0.5;

// Packed device images (fp16 bit patterns).
struct PackedLayer {
    std::vector<uint16_t> wpack;   // [ksteps][ncob][64 lanes][8]
    std::vector<uint16_t> bias;    // [ncob*16] logical order, zero padded
    std::vector<uint16_t> slope;   // [64] (empty for conv_last)
    int ncob = 0, ksteps = 0;
};
int last_ncob(int scale);                               // co-blocks of conv_last as launched: 1, 2, 4
PackedLayer pack_first(const Model& m);
PackedLayer pack_body(const Model& m, int layer);
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
