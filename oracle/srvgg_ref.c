/*
 * oracle/srvgg_ref.c — CPU ORACLE (test infrastructure, NOT a product path)
 * ==========================================================================
 * Plain-C restatement of the arithmetic that ONdraid/reve reaches by spawning
 * `realesrgan-ncnn-vulkan` (reference call sites: reve-shared/src/lib.rs:134-147,
 * reve-gui/src-tauri/src/commands.rs:52-65): the realesr-animevideov3
 * SRVGGNetCompact network (3x3 conv + PReLU stack, depth-to-space, nearest
 * residual) together with the binary's tile / pre-process / post-process loop.
 *
 * PARITY UNPINNED.  The arithmetic is NOT under /root/reference: it lives in
 * the un-vendored, un-versioned third-party executable
 * xinntao/Real-ESRGAN-ncnn-vulkan (+ Tencent/ncnn + the model files), none of
 * which exist in this image.  The reference holds no golden
 * vector, checksum or pixel assertion for this path
 * (reve-cli/tests/run_test.rs:31-34 only checks that out.mp4 exists).  This
 * file therefore restates the PUBLISHED algorithm (SRVGGNetCompact from
 * realesrgan/archs/srvgg_arch.py; ncnn layer semantics; realesrgan.cpp tiling)
 * from recall; it is cross-checked against an independent torch.nn.functional
 * restatement (tests/golden/make_golden.py) but against nothing produced by
 * the reference itself.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libreve_hip.so) never links or calls it.
 *
 * Numeric modes
 *   mode 0  "fp32"         : fp32 storage and arithmetic everywhere.
 *   mode 1  "fp16-storage" : ncnn's use_fp16_storage=1 / use_fp16_arithmetic=0
 *                            as the binary configures it: every blob and every
 *                            parameter is STORED as IEEE fp16 (round-to-nearest-
 *                            even), all arithmetic is fp32.  Rounding points:
 *                            after pre-process, after each conv(+bias), after
 *                            each PReLU, after conv_last, after the residual add.
 *                            This is the mode the HIP path is compared against.
 *   mode 2 / 3  "fp16-storage, Winograd F(2x2,3x3) / F(4x4,3x3)": what ncnn's Vulkan path MAY do for
 *                            the 3x3 stride-1 layers with >= 16 input and output channels (the 16 body
 *                            layers; conv_last of the x3 / x4 graphs) — SURVEY.md §2.3.2.  Transforms in
 *                            fp32, every blob of the pipeline stored as fp16: the transformed kernel
 *                            U = G g G^T, the transformed input tile V = B^T d B, the product sums
 *                            M = sum_ci U * V (fp32 accumulation, ci ascending), and the layer output
 *                            A^T M A + bias.  NOT a parity target: these modes exist to QUANTIFY how far
 *                            such an evaluation order moves the 8-bit output from mode 1
 *                            (tests/golden/make_winograd_report.py).
 *
 *   mode 4  "fp16-storage, Winograd F(2,3) along the row": the evaluation the HIP path's optional Winograd kernel
 *                            (reve_amd/csrc/kernels_wino.hip, reve_set_option("winograd", 1)) uses for the 16 body layers,
 *                            restated so that that kernel can be checked at the precision of a summation order instead
 *                            of through the 8-bit output only.  Per tap row dy the three taps g0, g1, g2 (fp16-stored) become
 *                            U0 = g0, U1 = ((g0 + g1) + g2) / 2, U2 = ((g0 - g1) + g2) / 2, U3 = g2 (fp32, stored fp16); a tile =
 *                            output pixels x = 2t, 2t + 1 of a row, its input pixels d0..d3 = x - 1 .. x + 2 of tap row dy give
 *                            V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3, each the CORRECTLY ROUNDED fp16 of the
 *                            exact sum (one packed fp16 add on the GPU); M_xi = sum over (dy, ci) of U_xi * V_xi in fp32
 *                            (dy, then ci ascending; M1 starts from the bias); y0 = M0 + (M1 + M2), y1 = (M1 - M2) - M3,
 *                            then the fp16 round of mode 1.  Like modes 2 / 3 it is NOT the parity target: the HIP path with
 *                            that kernel enabled is still compared with mode 1 (<= 1 LSB of the 8-bit output).
 *
 * Summation order of a conv output (the oracle's DEFINITION, both modes):
 *   acc = bias; for ky in 0..2: for kx in 0..2: for ci in 0..Cin-1:
 *       acc = fma(x[y+ky-1][x+kx-1][ci], w[co][ci][ky][kx], acc)
 *   (ncnn initialises the sum with the bias, then accumulates.)  In mode 1 the
 *   products are exact in fp32, so fma == mul+add; a different order (the GPU's
 *   MFMA order) only changes the fp32 rounding of the running sum.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SRVGG_FEAT 64

/* ---- fp16 <-> fp32, software, exact RNE (matches v_cvt_f16_f32 / F16C) ---- */
static inline uint16_t f32_to_f16_bits(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? (0x200u | ((ax >> 13) & 0x3ffu)) : 0));
    if (ax >= 0x477ff000u) /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    if (ax < 0x33000001u) /* < 2^-25 (or == 2^-25 tie -> even = 0) */
        return (uint16_t)sign;
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    if (e < -14) { /* subnormal half */
        int shift = -14 - e + 13; /* 14..24 */
        uint32_t r = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3ffu);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return (uint16_t)(sign | r);
}

static inline float f16_bits_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int s = 0; while (!(m & 0x400u)) { m <<= 1; s++; } m &= 0x3ffu; x = sign | ((uint32_t)(113 - s) << 23) | (m << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4); return f;
}

#if defined(__F16C__) && !defined(SRVGG_NO_F16C)
#include <immintrin.h>
/* hardware RNE conversion; tests check it against the software routine above */
static inline float rnd_h(float f) { return _cvtsh_ss(_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)); }
#else
static inline float rnd_h(float f) { return f16_bits_to_f32(f32_to_f16_bits(f)); }
#endif
float srvgg_round_f16(float f) { return rnd_h(f); }

/* exported for the unit tests of the conversion itself */
uint16_t srvgg_f32_to_f16(float f) { return f32_to_f16_bits(f); }
float srvgg_f16_to_f32(uint16_t h) { return f16_bits_to_f32(h); }

/* ---- weights: caller passes fp32 in PyTorch/ncnn OIHW order ---- */
typedef struct {
    int scale;          /* 2, 3, 4 */
    int n_body;         /* 16 for realesr-animevideov3 */
    const float *w_first, *b_first, *a_first;   /* [64][3][3][3], [64], [64] */
    const float *w_body, *b_body, *a_body;      /* [n][64][64][3][3], [n][64], [n][64] */
    const float *w_last, *b_last;               /* [3s^2][64][3][3], [3s^2] */
} srvgg_weights;

typedef float v8f __attribute__((vector_size(32), aligned(4)));

/* repack OIHW -> [tap][ci][co_pad] (co contiguous), optionally fp16-rounded */
static float *repack(const float *w, int co, int ci, int co_pad, int h16)
{
    float *r = (float *)calloc((size_t)9 * ci * co_pad, sizeof(float));
    for (int o = 0; o < co; o++)
        for (int i = 0; i < ci; i++)
            for (int t = 0; t < 9; t++) {
                float v = w[((size_t)o * ci + i) * 9 + t];
                r[((size_t)t * ci + i) * co_pad + o] = h16 ? rnd_h(v) : v;
            }
    return r;
}

/*
 * Register block: 6 pixels x 16 output channels = 12 ymm accumulators (+2 weight vectors, +1
 * broadcast).  Each weight vector loaded from L1/L2 feeds 6 FMAs, which keeps the 147 KB weight
 * set of a 64->64 layer from making the loop L2-bandwidth bound.  The per-output summation order
 * (bias, then ky, kx, ci ascending) is untouched by the blocking.
 */
#define NPX 6
static inline void conv_block(const float *const ip[NPX], int ci, int ws, const float *wr, int co_pad,
                              const float *bias16, v8f acc[NPX][2])
{
    const v8f b0 = ((const v8f *)bias16)[0], b1 = ((const v8f *)bias16)[1];
    v8f a00 = b0, a01 = b1, a10 = b0, a11 = b1, a20 = b0, a21 = b1;
    v8f a30 = b0, a31 = b1, a40 = b0, a41 = b1, a50 = b0, a51 = b1;
    for (int ky = 0; ky < 3; ky++)
        for (int kx = 0; kx < 3; kx++) {
            const size_t o = ((size_t)ky * ws + kx) * ci;
            const float *p0 = ip[0] + o, *p1 = ip[1] + o, *p2 = ip[2] + o;
            const float *p3 = ip[3] + o, *p4 = ip[4] + o, *p5 = ip[5] + o;
            const float *wt = wr + (size_t)(ky * 3 + kx) * ci * co_pad;
            for (int c = 0; c < ci; c++) {
                const v8f *wv = (const v8f *)(wt + (size_t)c * co_pad);
                const v8f w0 = wv[0], w1 = wv[1];
                float s;
                s = p0[c]; a00 = w0 * s + a00; a01 = w1 * s + a01;   /* contracted to FMA; products exact in mode 1 */
                s = p1[c]; a10 = w0 * s + a10; a11 = w1 * s + a11;
                s = p2[c]; a20 = w0 * s + a20; a21 = w1 * s + a21;
                s = p3[c]; a30 = w0 * s + a30; a31 = w1 * s + a31;
                s = p4[c]; a40 = w0 * s + a40; a41 = w1 * s + a41;
                s = p5[c]; a50 = w0 * s + a50; a51 = w1 * s + a51;
            }
        }
    acc[0][0] = a00; acc[0][1] = a01; acc[1][0] = a10; acc[1][1] = a11; acc[2][0] = a20; acc[2][1] = a21;
    acc[3][0] = a30; acc[3][1] = a31; acc[4][0] = a40; acc[4][1] = a41; acc[5][0] = a50; acc[5][1] = a51;
}

/*
 * 3x3 stride-1 zero-pad-1 convolution + bias on an NHWC float image that is
 * stored WITH a 1-pixel zero border: in has (h+2) x (w+2) pixels, out too
 * (border left untouched = 0).  co_pad is a multiple of 16 here (weights are repacked so).
 */
static void conv3x3(const float *in, int ci, float *out, int co, int co_pad,
                    const float *wr, const float *bias, int w, int h, int h16)
{
    const int ws = w + 2;
    float *bpad = (float *)calloc((size_t)co_pad + 16, sizeof(float));
    memcpy(bpad, bias, sizeof(float) * (size_t)co);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; y++) {
        for (int x0 = 0; x0 < w; x0 += NPX) {
            const int np = (w - x0 < NPX) ? (w - x0) : NPX;
            const float *ip[NPX];
            for (int p = 0; p < NPX; p++) ip[p] = in + ((size_t)y * ws + x0 + (p < np ? p : 0)) * ci;
            for (int cb = 0; cb < co_pad; cb += 16) {
                v8f acc[NPX][2];
                conv_block(ip, ci, ws, wr + cb, co_pad, bpad + cb, acc);
                for (int p = 0; p < np; p++) {
                    float *o = out + ((size_t)(y + 1) * ws + (x0 + p + 1)) * co;
                    for (int v = 0; v < 2; v++)
                        for (int l = 0; l < 8; l++) {
                            const int c = cb + v * 8 + l;
                            if (c < co) o[c] = h16 ? rnd_h(acc[p][v][l]) : acc[p][v][l];
                        }
                }
            }
        }
    }
    free(bpad);
}

/* ---- Winograd evaluation of a 3x3 stride-1 pad-1 layer (modes 2 and 3; see the header) ------------------
 * Lavin & Gray's minimal filtering matrices, the ones ncnn's winograd23 / winograd43 paths use. */
static const float WG2_BT[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
static const float WG2_G[4][3] = {{1, 0, 0}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0, 0, 1}};
static const float WG2_AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
static const float WG4_BT[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                   {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
static const float WG4_G[6][3] = {{1.f / 4, 0, 0}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                                  {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0, 0, 1}};
static const float WG4_AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};

/* U[(xi*n + nu)][ci][co] = (G g G^T)[xi][nu] of filter (co, ci); g = w[co][ci][3][3] (already fp16-rounded values in h16) */
static float *wino_kernel(const float *w, int co, int ci, int m, int h16)
{
    const int n = m + 2;
    float *U = (float *)malloc(sizeof(float) * (size_t)n * n * ci * co);
    for (int o = 0; o < co; o++)
        for (int i = 0; i < ci; i++) {
            float g[3][3], t[6][3];
            for (int a = 0; a < 9; a++) { float v = w[((size_t)o * ci + i) * 9 + a]; g[a / 3][a % 3] = h16 ? rnd_h(v) : v; }
            for (int a = 0; a < n; a++)
                for (int b = 0; b < 3; b++) {
                    float acc = 0.f;
                    for (int k = 0; k < 3; k++) acc += (m == 2 ? WG2_G[a][k] : WG4_G[a][k]) * g[k][b];
                    t[a][b] = acc;
                }
            for (int a = 0; a < n; a++)
                for (int b = 0; b < n; b++) {
                    float acc = 0.f;
                    for (int k = 0; k < 3; k++) acc += t[a][k] * (m == 2 ? WG2_G[b][k] : WG4_G[b][k]);
                    U[((size_t)(a * n + b) * ci + i) * co + o] = h16 ? rnd_h(acc) : acc;
                }
        }
    return U;
}

/* same image convention as conv3x3 (1-pixel zero border around both images) */
static void conv3x3_wino(const float *in, int ci, float *out, int co, const float *U, const float *bias,
                         int w, int h, int h16, int m)
{
    const int ws = w + 2, hs = h + 2, n = m + 2;
    const int tx_n = (w + m - 1) / m, ty_n = (h + m - 1) / m;
#pragma omp parallel
    {
        float *V = (float *)malloc(sizeof(float) * (size_t)n * n * ci);
        float *M = (float *)malloc(sizeof(float) * (size_t)n * n * co);
        float *d = (float *)malloc(sizeof(float) * (size_t)n * n * ci);
#pragma omp for schedule(dynamic, 1) collapse(2)
        for (int ty = 0; ty < ty_n; ty++)
            for (int tx = 0; tx < tx_n; tx++) {
                /* input tile: padded-image rows ty*m .. ty*m + n - 1 (zero beyond the padded image) */
                for (int a = 0; a < n; a++)
                    for (int b = 0; b < n; b++) {
                        const int yy = ty * m + a, xx = tx * m + b;
                        float *dp = d + (size_t)(a * n + b) * ci;
                        if (yy < hs && xx < ws) memcpy(dp, in + ((size_t)yy * ws + xx) * ci, sizeof(float) * (size_t)ci);
                        else memset(dp, 0, sizeof(float) * (size_t)ci);
                    }
                /* V = B^T d B per channel */
                for (int c = 0; c < ci; c++) {
                    float t[6][6];
                    for (int a = 0; a < n; a++)
                        for (int b = 0; b < n; b++) {
                            float acc = 0.f;
                            for (int k = 0; k < n; k++) acc += (m == 2 ? WG2_BT[a][k] : WG4_BT[a][k]) * d[(size_t)(k * n + b) * ci + c];
                            t[a][b] = acc;
                        }
                    for (int a = 0; a < n; a++)
                        for (int b = 0; b < n; b++) {
                            float acc = 0.f;
                            for (int k = 0; k < n; k++) acc += t[a][k] * (m == 2 ? WG2_BT[b][k] : WG4_BT[b][k]);
                            V[(size_t)(a * n + b) * ci + c] = h16 ? rnd_h(acc) : acc;
                        }
                }
                /* M[xi nu][co] = sum_ci U * V, fp32, ci ascending */
                for (int e = 0; e < n * n; e++) {
                    float *mp = M + (size_t)e * co;
                    for (int o = 0; o < co; o++) mp[o] = 0.f;
                    for (int c = 0; c < ci; c++) {
                        const float v = V[(size_t)e * ci + c];
                        const float *up = U + ((size_t)e * ci + c) * co;
                        for (int o = 0; o < co; o++) mp[o] += up[o] * v;
                    }
                    if (h16) for (int o = 0; o < co; o++) mp[o] = rnd_h(mp[o]);
                }
                /* Y = A^T M A + bias */
                for (int o = 0; o < co; o++) {
                    float t[4][6];
                    for (int a = 0; a < m; a++)
                        for (int b = 0; b < n; b++) {
                            float acc = 0.f;
                            for (int k = 0; k < n; k++) acc += (m == 2 ? WG2_AT[a][k] : WG4_AT[a][k]) * M[(size_t)(k * n + b) * co + o];
                            t[a][b] = acc;
                        }
                    for (int a = 0; a < m; a++)
                        for (int b = 0; b < m; b++) {
                            const int y = ty * m + a, x = tx * m + b;
                            if (y >= h || x >= w) continue;
                            float acc = 0.f;
                            for (int k = 0; k < n; k++) acc += t[a][k] * (m == 2 ? WG2_AT[b][k] : WG4_AT[b][k]);
                            acc += bias[o];
                            out[((size_t)(y + 1) * ws + (x + 1)) * co + o] = h16 ? rnd_h(acc) : acc;
                        }
                }
            }
        free(V); free(M); free(d);
    }
}

/* ---- mode 4: Winograd F(2,3) along the row, direct sum over the tap rows (see the header) ---------------- */
/* the fp16 value nearest to the exact a + b of two fp16 values (ties to even): what one fp16 add instruction returns */
static float add_h_exact(float a, float b)
{
    const double s = (double)a + (double)b;              /* exact: both are multiples of 2^-24 below 2^16 */
    if (s == 0.0 || s != s || s - s != 0.0) return (float)s;
    int e;
    (void)frexp(fabs(s), &e);                            /* 2^(e-1) <= |s| < 2^e */
    const int qe = (e - 1 < -14 ? -14 : e - 1) - 10;     /* exponent of the fp16 grid's spacing there */
    const double r = ldexp(nearbyint(ldexp(s, -qe)), qe);
    if (fabs(r) >= 65520.0) return r < 0 ? -INFINITY : INFINITY;
    return (float)r;
}

/* U[((dy*4 + xi)*ci + i)*co + o] */
static float *wino_x_kernel(const float *w, int co, int ci)
{
    float *U = (float *)malloc(sizeof(float) * (size_t)12 * ci * co);
    for (int o = 0; o < co; o++)
        for (int i = 0; i < ci; i++)
            for (int dy = 0; dy < 3; dy++) {
                const float *t = w + ((size_t)o * ci + i) * 9 + dy * 3;
                const float g0 = rnd_h(t[0]), g1 = rnd_h(t[1]), g2 = rnd_h(t[2]);
                const float u[4] = {g0, ((g0 + g1) + g2) * 0.5f, ((g0 - g1) + g2) * 0.5f, g2};
                for (int xi = 0; xi < 4; xi++) U[((size_t)(dy * 4 + xi) * ci + i) * co + o] = rnd_h(u[xi]);
            }
    return U;
}

/* same image convention as conv3x3 (1-pixel zero border around both images); fp16 storage implied */
static void conv3x3_wino_x(const float *in, int ci, float *out, int co, const float *U, const float *bias, int w, int h)
{
    const int ws = w + 2, nt = (w + 1) / 2;
#pragma omp parallel
    {
        float *V = (float *)malloc(sizeof(float) * (size_t)12 * ci);
        float *M = (float *)malloc(sizeof(float) * (size_t)4 * co);
#pragma omp for schedule(dynamic, 4) collapse(2)
        for (int y = 0; y < h; y++)
            for (int t = 0; t < nt; t++) {
                for (int dy = 0; dy < 3; dy++) {
                    const float *row = in + (size_t)(y + dy) * ws * ci;
                    for (int c = 0; c < ci; c++) {
                        float d[4];
                        for (int k = 0; k < 4; k++) d[k] = 2 * t + k < ws ? row[(size_t)(2 * t + k) * ci + c] : 0.f;
                        V[(size_t)(dy * 4 + 0) * ci + c] = add_h_exact(d[0], -d[2]);
                        V[(size_t)(dy * 4 + 1) * ci + c] = add_h_exact(d[1], d[2]);
                        V[(size_t)(dy * 4 + 2) * ci + c] = add_h_exact(d[2], -d[1]);
                        V[(size_t)(dy * 4 + 3) * ci + c] = add_h_exact(d[1], -d[3]);
                    }
                }
                for (int xi = 0; xi < 4; xi++) {
                    float *mp = M + (size_t)xi * co;
                    for (int o = 0; o < co; o++) mp[o] = xi == 1 ? bias[o] : 0.f;
                    for (int dy = 0; dy < 3; dy++)
                        for (int c = 0; c < ci; c++) {
                            const float v = V[(size_t)(dy * 4 + xi) * ci + c];
                            const float *up = U + ((size_t)(dy * 4 + xi) * ci + c) * co;
                            for (int o = 0; o < co; o++) mp[o] = fmaf(up[o], v, mp[o]);
                        }
                }
                for (int o = 0; o < co; o++) {
                    const float y0 = M[o] + (M[co + o] + M[2 * co + o]);
                    const float y1 = (M[co + o] - M[2 * co + o]) - M[3 * co + o];
                    out[((size_t)(y + 1) * ws + (2 * t + 1)) * co + o] = rnd_h(y0);
                    if (2 * t + 1 < w) out[((size_t)(y + 1) * ws + (2 * t + 2)) * co + o] = rnd_h(y1);
                }
            }
        free(V); free(M);
    }
}

static void prelu(float *a, int c, const float *slope, int w, int h, int h16)
{
    const int ws = w + 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        float *p = a + ((size_t)(y + 1) * ws + 1) * c;
        for (int x = 0; x < w; x++, p += c)
            for (int k = 0; k < c; k++) {
                float v = p[k];
                if (v < 0.f) { v = v * slope[k]; if (h16) v = rnd_h(v); }
                p[k] = v;
            }
    }
}

typedef struct {
    int h16, scale, n_body, co_last, co_last_pad;
    int wino;                                     /* 0, or the Winograd output tile size m (2 / 4) of modes 2 / 3; -1: mode 4 (row-wise F(2,3)) */
    float **u_body, *u_last;                      /* Winograd-domain kernels (modes 2 / 3) */
    float *w_first, *w_body, *w_last;            /* repacked */
    float *b_first, *a_first, *b_body, *a_body, *b_last;
} prepared;

static float *dup_round(const float *s, int n, int h16)
{
    float *d = (float *)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; i++) d[i] = h16 ? rnd_h(s[i]) : s[i];
    return d;
}

static void prepare(prepared *P, const srvgg_weights *W, int mode)
{
    const int F = SRVGG_FEAT;
    P->h16 = (mode >= 1); P->scale = W->scale; P->n_body = W->n_body;
    P->wino = mode == 2 ? 2 : (mode == 3 ? 4 : (mode == 4 ? -1 : 0));
    P->u_body = NULL; P->u_last = NULL;
    P->co_last = 3 * W->scale * W->scale;
    P->co_last_pad = (P->co_last + 15) / 16 * 16;
    P->w_first = repack(W->w_first, F, 3, F, P->h16);
    P->w_body = (float *)malloc(sizeof(float) * (size_t)W->n_body * 9 * F * F);
    for (int l = 0; l < W->n_body; l++) {
        float *r = repack(W->w_body + (size_t)l * F * F * 9, F, F, F, P->h16);
        memcpy(P->w_body + (size_t)l * 9 * F * F, r, sizeof(float) * 9 * F * F);
        free(r);
    }
    P->w_last = repack(W->w_last, P->co_last, F, P->co_last_pad, P->h16);
    P->b_first = dup_round(W->b_first, F, P->h16);
    P->a_first = dup_round(W->a_first, F, P->h16);
    P->b_body = dup_round(W->b_body, W->n_body * F, P->h16);
    P->a_body = dup_round(W->a_body, W->n_body * F, P->h16);
    P->b_last = dup_round(W->b_last, P->co_last, P->h16);
    if (P->wino > 0) {
        P->u_body = (float **)malloc(sizeof(float *) * (size_t)W->n_body);
        for (int l = 0; l < W->n_body; l++) P->u_body[l] = wino_kernel(W->w_body + (size_t)l * F * F * 9, F, F, P->wino, P->h16);
        if (P->co_last >= 16) P->u_last = wino_kernel(W->w_last, P->co_last, F, P->wino, P->h16);
    } else if (P->wino < 0) {
        P->u_body = (float **)malloc(sizeof(float *) * (size_t)W->n_body);
        for (int l = 0; l < W->n_body; l++) P->u_body[l] = wino_x_kernel(W->w_body + (size_t)l * F * F * 9, F, F);
    }
}

static void unprepare(prepared *P)
{
    free(P->w_first); free(P->w_body); free(P->w_last);
    free(P->b_first); free(P->a_first); free(P->b_body); free(P->a_body); free(P->b_last);
    if (P->u_body) { for (int l = 0; l < P->n_body; l++) free(P->u_body[l]); free(P->u_body); }
    free(P->u_last);
}

/*
 * One network evaluation on a tile.
 *   tin  : tw x th x 3 floats in [0,1] (already pre-processed / fp16-rounded)
 *   tout : (tw*s) x (th*s) x 3 floats — network `output` blob (before post-process)
 *   dump_layer >= 0: copy the activation AFTER layer `dump_layer` (0 = conv_first+PReLU,
 *                    1..n_body = body convs+PReLU) as tw*th*64 floats (logical channel order)
 *                    into dump; dump_layer == n_body+1 dumps conv_last output (tw*th*3s^2).
 */
static void net_forward(const prepared *P, const float *tin, int tw, int th, float *tout,
                        int dump_layer, float *dump)
{
    const int F = SRVGG_FEAT, ws = tw + 2, hs = th + 2, s = P->scale;
    float *x0 = (float *)calloc((size_t)ws * hs * 3, sizeof(float));
    float *A = (float *)calloc((size_t)ws * hs * F, sizeof(float));
    float *B = (float *)calloc((size_t)ws * hs * F, sizeof(float));
    float *L = (float *)calloc((size_t)ws * hs * P->co_last, sizeof(float));
    for (int y = 0; y < th; y++)
        memcpy(x0 + ((size_t)(y + 1) * ws + 1) * 3, tin + (size_t)y * tw * 3, sizeof(float) * 3 * tw);

    conv3x3(x0, 3, A, F, F, P->w_first, P->b_first, tw, th, P->h16);
    prelu(A, F, P->a_first, tw, th, P->h16);
    if (dump_layer == 0 && dump)
        for (int y = 0; y < th; y++)
            memcpy(dump + (size_t)y * tw * F, A + ((size_t)(y + 1) * ws + 1) * F, sizeof(float) * F * tw);
    for (int l = 0; l < P->n_body; l++) {
        if (P->wino < 0) conv3x3_wino_x(A, F, B, F, P->u_body[l], P->b_body + l * F, tw, th);
        else if (P->wino) conv3x3_wino(A, F, B, F, P->u_body[l], P->b_body + l * F, tw, th, P->h16, P->wino);
        else conv3x3(A, F, B, F, F, P->w_body + (size_t)l * 9 * F * F, P->b_body + l * F, tw, th, P->h16);
        prelu(B, F, P->a_body + l * F, tw, th, P->h16);
        float *t = A; A = B; B = t;
        if (dump_layer == l + 1 && dump)
            for (int y = 0; y < th; y++)
                memcpy(dump + (size_t)y * tw * F, A + ((size_t)(y + 1) * ws + 1) * F, sizeof(float) * F * tw);
    }
    if (P->u_last) conv3x3_wino(A, F, L, P->co_last, P->u_last, P->b_last, tw, th, P->h16, P->wino);
    else conv3x3(A, F, L, P->co_last, P->co_last_pad, P->w_last, P->b_last, tw, th, P->h16);
    if (dump_layer == P->n_body + 1 && dump)
        for (int y = 0; y < th; y++)
            memcpy(dump + (size_t)y * tw * P->co_last, L + ((size_t)(y + 1) * ws + 1) * P->co_last,
                   sizeof(float) * P->co_last * tw);

    /* PixelShuffle (PyTorch order: out[c][y*s+i][x*s+j] = in[c*s*s + i*s + j][y][x])
       + nearest Interp of the input blob + BinaryOp add */
    const int ow = tw * s;
    for (int y = 0; y < th; y++)
        for (int x = 0; x < tw; x++) {
            const float *lp = L + ((size_t)(y + 1) * ws + (x + 1)) * P->co_last;
            const float *ip = tin + ((size_t)y * tw + x) * 3;
            for (int c = 0; c < 3; c++)
                for (int i = 0; i < s; i++)
                    for (int j = 0; j < s; j++) {
                        float v = lp[c * s * s + i * s + j] + ip[c];
                        if (P->h16) v = rnd_h(v);
                        tout[((size_t)(y * s + i) * ow + (x * s + j)) * 3 + c] = v;
                    }
        }
    free(x0); free(A); free(B); free(L);
}

static inline float preproc(uint8_t v, int h16)
{
    float f = (float)v * (1.0f / 255.0f);          /* ncnn norm_vals = 1/255.f */
    return h16 ? rnd_h(f) : f;
}

static inline uint8_t postproc(float v)
{
    float q = v * 255.0f + 0.5f;                   /* realesrgan_postproc: v*255+0.5, clamp, truncate */
    if (!(q > 0.f)) q = 0.f;
    if (q > 255.f) q = 255.f;
    return (uint8_t)q;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/*
 * Whole pipeline of the binary for one frame.
 *   tile    : 0 = whole frame as one tile with NO apron (seam-free mode);
 *             N>0 = realesrgan.cpp tiling: ceil(w/N) x ceil(h/N) tiles, each cropped with a
 *             `prepad`-pixel apron whose samples outside the frame replicate the border
 *             (clamp-to-edge), network run per tile (zero padding at the padded-tile boundary),
 *             apron*scale cropped from the result.
 * Returns 0, or -1 on bad arguments.
 */
int srvgg_ref_upscale(const srvgg_weights *W, int mode, const uint8_t *src, int w, int h,
                      long src_stride, uint8_t *dst, long dst_stride, int tile, int prepad,
                      int nthreads)
{
    if (!W || !src || !dst || w <= 0 || h <= 0 || W->scale < 2 || W->scale > 4 || mode < 0 || mode > 4)
        return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    prepared P; prepare(&P, W, mode);
    const int s = W->scale;
    const int TX = tile > 0 ? tile : w, TY = tile > 0 ? tile : h, pad = tile > 0 ? prepad : 0;
    const int xt = (w + TX - 1) / TX, yt = (h + TY - 1) / TY;
    for (int yi = 0; yi < yt; yi++)
        for (int xi = 0; xi < xt; xi++) {
            const int x0 = xi * TX - pad, x1 = (((xi + 1) * TX < w) ? (xi + 1) * TX : w) + pad;
            const int y0 = yi * TY - pad, y1 = (((yi + 1) * TY < h) ? (yi + 1) * TY : h) + pad;
            const int tw = x1 - x0, th = y1 - y0;
            float *tin = (float *)malloc(sizeof(float) * (size_t)tw * th * 3);
            float *tout = (float *)malloc(sizeof(float) * (size_t)tw * th * 3 * s * s);
            for (int y = 0; y < th; y++)
                for (int x = 0; x < tw; x++) {
                    const uint8_t *sp = src + (size_t)clampi(y0 + y, 0, h - 1) * src_stride
                                        + (size_t)clampi(x0 + x, 0, w - 1) * 3;
                    for (int c = 0; c < 3; c++) tin[((size_t)y * tw + x) * 3 + c] = preproc(sp[c], P.h16);
                }
            net_forward(&P, tin, tw, th, tout, -1, NULL);
            const int nw = tw - 2 * pad, nh = th - 2 * pad;   /* un-padded tile */
            for (int y = 0; y < nh * s; y++) {
                uint8_t *dp = dst + (size_t)((yi * TY) * s + y) * dst_stride + (size_t)(xi * TX) * s * 3;
                const float *tp = tout + ((size_t)(y + pad * s) * (tw * s) + pad * s) * 3;
                for (int x = 0; x < nw * s * 3; x++) dp[x] = postproc(tp[x]);
            }
            free(tin); free(tout);
        }
    unprepare(&P);
    return 0;
}

/*
 * Layer probe for kernel-level parity tests: runs the network on the whole image as ONE tile
 * (no apron) and returns the activation after `layer` (see net_forward) as floats, logical
 * channel order, NHWC without border.
 */
int srvgg_ref_layer(const srvgg_weights *W, int mode, const uint8_t *src, int w, int h,
                    long src_stride, int layer, float *out)
{
    if (!W || !src || !out || w <= 0 || h <= 0) return -1;
    prepared P; prepare(&P, W, mode);
    float *tin = (float *)malloc(sizeof(float) * (size_t)w * h * 3);
    float *tout = (float *)malloc(sizeof(float) * (size_t)w * h * 3 * W->scale * W->scale);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++)
                tin[((size_t)y * w + x) * 3 + c] = preproc(src[(size_t)y * src_stride + x * 3 + c], P.h16);
    net_forward(&P, tin, w, h, tout, layer, out);
    free(tin); free(tout); unprepare(&P);
    return 0;
}

int srvgg_ref_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
