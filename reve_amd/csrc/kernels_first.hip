// conv_first for gfx950: u8 RGB frame -> fp16 activation arena (3->64 3x3 conv + bias + PReLU with the
// u8 -> fp16 pre-process fused in front; K = 27 padded to 2 k-steps of v_mfma_f32_16x16x32_f16).
// An HBM-write-bound kernel (128 B out per 3 B in).  Built with -mllvm -amdgpu-mfma-vgpr-form=1 (see the
// Makefile) so the accumulators come back in VGPRs and the epilogue has no v_accvgpr_read.
#include "kernels_dev.h"

namespace reve {

// -------------------------------------------------------------------------------------------
// conv_first: u8 RGB frame -> pre-process (x * 1/255 -> fp16) -> 3x3 conv 3->64 + bias -> fp16
// -> PReLU -> fp16 arena, 16x32 tiles; K = 9 taps x 4 (3 channels + zero) = 36 -> two 16x16x32 k-steps.
// In ncnn-compat tile mode plane pixels outside the frame replicate the frame border (clamp), pixels
// outside the PLANE are zero (the convolution's own padding).
// Persistent: a workgroup takes a run of consecutive work items and handles them in groups of FIRST_NT.
// ALL source bytes of a group are fetched before its first output store is issued: source reads queued
// behind the 128-B-per-pixel write stream take several microseconds, reads issued ahead of it do not.
// -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_first(const FirstArgs a, const PlaneDesc* __restrict__ planes,
                                               const uint32_t* __restrict__ items)
{
    constexpr int NT = FIRST_NT;
    // LDS image of a tile's 18 x 34 input pixels, 8 bytes each, ROW PITCH 49 pixels: a ds_read_b64 is served in two groups
    // of 32 lanes = two 16-pixel runs of different taps; runs in different rows must fall on the two halves of the 64
    // banks (pixel q sits on banks 2q, 2q+1 mod 64), i.e. the pitch must be 17 mod 32.  With the natural pitch of 34
    // rows were 2 pixels apart modulo the banks and 29 % of the kernel's LDS cycles were bank conflicts (r01 PMC).
    constexpr int FP = 49;
    static_assert(FP >= LDS_W && FP % 32 == 17, "k_first LDS pitch");
    __shared__ __attribute__((aligned(16))) h4 tile[NT][LDS_H * FP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;

    // the (up to) three pixels of the 18x34 input image this thread fetches; tile-invariant coordinates
    constexpr int NQ = (LDS_H * LDS_W + 255) / 256;
    int qy[NQ], qx[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = tid + 256 * i;
        qy[i] = q / LDS_W - 1;
        qx[i] = q - (qy[i] + 1) * LDS_W - 1;
    }
    // k-step 0: k = 8g + j  <->  tap 2g + (j>>2), channel j&3;  k-step 1: tap 8 lives in g == 0, j < 4
    const int t0 = 2 * g, t1 = 2 * g + 1;
    const int q0 = (t0 / 3) * FP + (t0 % 3), q1 = (t1 / 3) * FP + (t1 % 3), q8 = 2 * FP + 2;

    struct Tile { int plane, ty, tx; PlaneDesc pd; const uint8_t* src; };
    auto decode = [&](int it) {
        const Item i = decode_any(it, a, items);
        Tile t;
        t.plane = i.plane; t.ty = i.ty; t.tx = i.tx;
        t.pd = planes[t.plane];
        t.src = a.n_src ? a.src_tab[t.plane < MAX_BATCH ? t.plane : 0] : a.src;      // (several frames per launch: one plane each)
        return t;
    };
    // raw source bytes of this thread's pixels.  Branch-free: the address is clamped into the frame (plane
    // pixels outside the frame replicate its border), the load is unconditional, and pixels outside the
    // PLANE (the convolution's zero padding) are zeroed afterwards through the returned mask.
    auto fetch = [&](const Tile& t, uint32_t (&px)[NQ][3]) -> uint32_t {
        uint32_t inside = 0;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int py = t.ty * TILE_H + qy[i], pxx = t.tx * TILE_W + qx[i];
            if (py >= 0 && py < t.pd.h && pxx >= 0 && pxx < t.pd.w) inside |= 1u << i;
            int fy = t.pd.y0 + py, fx = t.pd.x0 + pxx;
            fy = fy < 0 ? 0 : (fy >= a.frame_h ? a.frame_h - 1 : fy);
            fx = fx < 0 ? 0 : (fx >= a.frame_w ? a.frame_w - 1 : fx);
            const uint8_t* sp = t.src + (uint32_t)(fy * (int)a.src_stride + fx * 3);   // < 2^31: Engine::bad_frame
            px[i][0] = sp[0];
            px[i][1] = sp[1];
            px[i][2] = sp[2];
        }
        return inside;
    };

    // XCD-aware persistent mapping as in k_body: the workgroups of one XCD take consecutive chunks of
    // the work list, so the source rows they share stay in that XCD's L2.
    const int G = gridDim.x;
    int chunk = blockIdx.x;
    if ((G & 7) == 0) chunk = (chunk & 7) * (G >> 3) + (chunk >> 3);
    int it = (int)((long long)chunk * a.n_items / G);
    const int it_end = (int)((long long)(chunk + 1) * a.n_items / G);
    if (it >= it_end) return;
    const uint32_t lane_off = (uint32_t)(pl * PIX_BYTES + 16 * g);
    const long long row_bytes = (long long)a.Wp * PIX_BYTES;
    for (int g0 = it; g0 < it_end; g0 += NT) {
        const int ng = it_end - g0 < NT ? it_end - g0 : NT;
        if (g0 != it) __syncthreads();   // the previous group's LDS images are free again
        {
            uint32_t px[NT][NQ][3], inside[NT];
            const int last = it_end - 1;
#pragma unroll
            for (int k = 0; k < NT; ++k) inside[k] = fetch(decode(g0 + k < last ? g0 + k : last), px[k]);
#pragma unroll
            for (int k = 0; k < NT; ++k) {
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const float sc = (inside[k] >> i & 1u) ? (1.0f / 255.0f) : 0.0f;
                    h4 v;
                    v[0] = (_Float16)((float)px[k][i][0] * sc);
                    v[1] = (_Float16)((float)px[k][i][1] * sc);
                    v[2] = (_Float16)((float)px[k][i][2] * sc);
                    v[3] = (_Float16)0;
                    if (i < NQ - 1 || tid + 256 * i < LDS_H * LDS_W) tile[k][(qy[i] + 1) * FP + qx[i] + 1] = v;
                }
            }
        }
        __syncthreads();
        // weights, bias and slopes are (re)loaded per group, after the fetch, so that the fetch's registers
        // and theirs are never live together (128 VGPRs = 4 workgroups per CU)
        h8 wf[2][4];
        const h8* wp = (const h8*)a.wpack;
    #pragma unroll
        for (int s = 0; s < 2; ++s)
    #pragma unroll
            for (int m = 0; m < 4; ++m) wf[s][m] = wp[(s * 4 + m) * 64 + lane];
        float bias[4][4];
    #pragma unroll
        for (int m = 0; m < 4; ++m) {
            const h4 bb = *(const h4*)(a.bias + 16 * m + 4 * g);
    #pragma unroll
            for (int r = 0; r < 4; ++r) bias[m][r] = (float)bb[r];
        }
        const h4 s0 = *(const h4*)(a.slope + 0 + 4 * g), s1 = *(const h4*)(a.slope + 16 + 4 * g);
        const h4 s2 = *(const h4*)(a.slope + 32 + 4 * g), s3 = *(const h4*)(a.slope + 48 + 4 * g);
        const h8 slope01 = __builtin_shufflevector(s0, s1, 0, 1, 2, 3, 4, 5, 6, 7);
        const h8 slope23 = __builtin_shufflevector(s2, s3, 0, 1, 2, 3, 4, 5, 6, 7);
      for (int k = 0; k < ng; ++k) {
        const Tile t = decode(g0 + k);
        const h4* buf = tile[k];
        // uniform part of the output address: this wave's first row of the tile, arena border included
        char* const wbase = a.out + t.pd.base
                            + ((long long)(t.ty * TILE_H + 4 * wave + 1) * a.Wp + (t.tx * TILE_W + 1)) * PIX_BYTES;
        const int oy0 = t.ty * TILE_H + 4 * wave, ox0 = t.tx * TILE_W + pl;
#pragma unroll
        for (int pb = 0; pb < 8; ++pb) {
            const int rr = pb >> 1, xb = pb & 1;
            const int qb = (4 * wave + rr) * FP + 16 * xb + pl;
            const h4 lo = buf[qb + q0], hi = buf[qb + q1];
            const h8 B0 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            h4 l8 = buf[qb + q8];
            if (g != 0) l8 = (h4)(_Float16)0;
            const h8 B1 = __builtin_shufflevector(l8, (h4)(_Float16)0, 0, 1, 2, 3, 4, 5, 6, 7);
            h8 o01, o23;
            f4 acc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[m] = (f4){bias[m][0], bias[m][1], bias[m][2], bias[m][3]};
                acc[m] = MFMA16(wf[0][m], B0, acc[m]);
                acc[m] = MFMA16(wf[1][m], B1, acc[m]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o01[r] = (_Float16)acc[0][r];
                o01[4 + r] = (_Float16)acc[1][r];
                o23[r] = (_Float16)acc[2][r];
                o23[4 + r] = (_Float16)acc[3][r];
            }
            o01 = prelu8(o01, slope01);
            o23 = prelu8(o23, slope23);
            if (oy0 + rr < t.pd.h && ox0 + 16 * xb < t.pd.w) {
                char* dp = wbase + rr * row_bytes + 16 * xb * PIX_BYTES + lane_off;
                *(h8*)dp = o01;            // channel half 0: co-blocks 0,1
                *(h8*)(dp + 64) = o23;     // channel half 1: co-blocks 2,3
            }
        }
      }
    }
}

int launch_first(const FirstArgs& a, int grid, void* stream)
{
    launch_prepare();
    hipLaunchKernelGGL(k_first, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, a.planes, a.items);
    return launch_status();
}

}  // namespace reve
