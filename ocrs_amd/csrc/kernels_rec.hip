// Ragged-batch kernels for the recognition conv stack.
//
// The reference pads every text line to its own width group (recognition.rs:437),
// so one request holds ~30 groups of images with different widths.  Launching each
// layer once per group gives ~30 small grids per layer that cannot fill 256 CUs.
// Here all groups live in ONE buffer (group-major, then image, y, x; NHWC) and every
// layer is ONE launch: a block owns a tile of rows (pixels) inside a single group and
// finds its group by binary search in a cumulative tile table.  Per-pixel arithmetic
// is exactly that of the per-group kernels (same fmaf chains, same order), so results
// are bit-identical.
#include <algorithm>

#include "common.hpp"
#include "kernels.hpp"
#include "split_mfma.hpp"

namespace ocrs {
namespace k {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Workgroup b runs on XCD b % 8 (observed; used for speed only).  Neighbouring tiles share
// im2col rows, so give each XCD a contiguous range of tiles: their re-reads then hit that
// XCD's L2 instead of going to HBM once per XCD.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

__device__ __forceinline__ int find_group(const int32_t* __restrict__ toff, int G, int tile) {
    int lo = 0, hi = G;  // largest g with toff[g] <= tile
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (toff[mid] <= tile) lo = mid; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------
// conv 3x3 (Cin = 1) + bias + ReLU + MaxPool 2x2, fused.  One thread = one POOLED
// pixel x 4 output channels: four conv outputs (9 fmaf each per channel, taps
// (ky,kx) ascending from acc = bias, out-of-image taps fmaf(0,w,acc)), ReLU, then
// m = v0; m = v > m ? v : m over (py,px) ascending — the spec order of the unfused ops.
// HBM: reads 4 B per input pixel (L1/L2 serve the tap re-use), writes 4*Cout B per
// pooled pixel; the full-resolution conv output is never materialised.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
conv1_relu_pool_ragged_kernel(const float* __restrict__ x, RaggedView in, const float* __restrict__ wt,
                              const float* __restrict__ bias, int cout, float* __restrict__ y, RaggedView out) {
    __shared__ float sw[9 * 64 + 64];
    for (int i = threadIdx.x; i < 9 * cout; i += 256) sw[i] = wt[i];
    for (int i = threadIdx.x; i < cout; i += 256) sw[9 * 64 + i] = bias[i];
    __syncthreads();
    const int g = find_group(out.toff256, out.G, blockIdx.x);
    const int tile = blockIdx.x - out.toff256[g];
    const int ow = out.W[g], oh = out.H, iw = in.W[g], ih = in.H;
    const int64_t rows = (int64_t)out.n[g] * oh * ow;
    const float* __restrict__ xg = x + in.poff[g];
    float* __restrict__ yg = y + out.poff[g] * cout;
    const int cq = cout >> 2;
    // (pixel, channel quad) of this thread's first item; the next item is 256 / cq pixels further on.  One 64-bit
    // division per thread up front, then the coordinates are stepped with carries (32-bit): the per-item 64-bit
    // div / mod chain cost more than the 144 FMAs of the item.
    const int step_px = 256 / cq;                       // cq divides 256 (cout = 16, 32 or 64)
    const int q = threadIdx.x % cq;
    int64_t r = (int64_t)tile * 256 + threadIdx.x / cq;
    int ox = (int)(r % ow);
    int64_t t0 = r / ow;
    int oy = (int)(t0 % oh);
    int64_t img = t0 / oh;
    // the thread's channel quad never changes: its 9 x 4 weights and 4 biases live in registers
    float wq[9][4], bq[4];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int c = 0; c < 4; c++) wq[t][c] = sw[t * cout + 4 * q + c];
#pragma unroll
    for (int c = 0; c < 4; c++) bq[c] = sw[9 * 64 + 4 * q + c];
    for (int it = 0; it < cq; it++, r += step_px) {
        if (r >= rows) break;
        if (it > 0) {
            ox += step_px;
            while (ox >= ow) { ox -= ow; if (++oy == oh) { oy = 0; img++; } }
        }
        const float* xi = xg + img * ih * iw;
        // 4x4 input patch around the 2x2 conv outputs
        float p[4][4];
        const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
        if (iy0 >= 0 && iy0 + 3 < ih && ix0 >= 0 && ix0 + 3 < iw) {
            // the whole patch is inside the image (all but the border pixels): one base pointer, constant offsets
            const float* pp = xi + (int64_t)iy0 * iw + ix0;
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) p[a][b] = pp[a * iw + b];
        } else {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int iy = iy0 + a, ix = ix0 + b;
                    p[a][b] = ((unsigned)iy < (unsigned)ih && (unsigned)ix < (unsigned)iw) ? xi[(int64_t)iy * iw + ix] : 0.0f;
                }
        }
        float m[4];
#pragma unroll
        for (int py = 0; py < 2; py++)
#pragma unroll
            for (int px = 0; px < 2; px++) {
                float acc[4];
#pragma unroll
                for (int c = 0; c < 4; c++) acc[c] = bq[c];
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        const float xv = p[py + ky][px + kx];
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[c] = fmaf(xv, wq[ky * 3 + kx][c], acc[c]);
                    }
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float v = acc[c] > 0.0f ? acc[c] : 0.0f;
                    if (py == 0 && px == 0) m[c] = v;
                    else m[c] = v > m[c] ? v : m[c];
                }
            }
        *reinterpret_cast<float4*>(yg + r * cout + 4 * q) = make_float4(m[0], m[1], m[2], m[3]);
    }
}

void conv1_relu_pool_ragged(const float* x, const RaggedView& in, const float* wt, const float* bias, int cout,
                            float* y, const RaggedView& out, hipStream_t s) {
    if (out.ntiles256 <= 0) return;
    hipLaunchKernelGGL(conv1_relu_pool_ragged_kernel, dim3(out.ntiles256), dim3(256), 0, s, x, in, wt, bias, cout, y, out);
}

// ---------------------------------------------------------------------------
// Pools on the ragged batch (kernel = stride = (kh, kw), floor).  HBM-bound.
// ---------------------------------------------------------------------------
template <bool AVG>
__global__ void __launch_bounds__(256)
pool_ragged_kernel(const float* __restrict__ x, RaggedView in, int c, int kh, int kw, float* __restrict__ y,
                   RaggedView out) {
    const int g = find_group(out.toff256, out.G, blockIdx.x);
    const int tile = blockIdx.x - out.toff256[g];
    const int ow = out.W[g], oh = out.H, iw = in.W[g], ih = in.H;
    const int64_t rows = (int64_t)out.n[g] * oh * ow;
    const float* __restrict__ xg = x + in.poff[g] * c;
    float* __restrict__ yg = y + out.poff[g] * c;
    const int cq = c >> 2;
    const float inv = 1.0f / (float)(kh * kw);
    for (int i = threadIdx.x; i < 256 * cq; i += 256) {
        const int64_t r = (int64_t)tile * 256 + i / cq;
        if (r >= rows) break;
        const int q = i % cq;
        const int ox = (int)(r % ow);
        const int oy = (int)((r / ow) % oh);
        const int64_t img = r / ((int64_t)ow * oh);
        const float* xp = xg + ((img * ih + (int64_t)oy * kh) * iw + (int64_t)ox * kw) * c + 4 * q;
        float4 acc = AVG ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(xp);
        for (int ky = 0; ky < kh; ky++)
            for (int kx = 0; kx < kw; kx++) {
                const float4 v = *reinterpret_cast<const float4*>(xp + ((int64_t)ky * iw + kx) * c);
                if (AVG) { acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w; }
                else {
                    acc.x = v.x > acc.x ? v.x : acc.x; acc.y = v.y > acc.y ? v.y : acc.y;
                    acc.z = v.z > acc.z ? v.z : acc.z; acc.w = v.w > acc.w ? v.w : acc.w;
                }
            }
        if (AVG) { acc.x = acc.x * inv; acc.y = acc.y * inv; acc.z = acc.z * inv; acc.w = acc.w * inv; }
        *reinterpret_cast<float4*>(yg + r * c + 4 * q) = acc;
    }
}

void pool_ragged(const float* x, const RaggedView& in, int c, int kh, int kw, bool avg, float* y, const RaggedView& out,
                 hipStream_t s) {
    if (out.ntiles256 <= 0) return;
    if (avg) hipLaunchKernelGGL((pool_ragged_kernel<true>), dim3(out.ntiles256), dim3(256), 0, s, x, in, c, kh, kw, y, out);
    else hipLaunchKernelGGL((pool_ragged_kernel<false>), dim3(out.ntiles256), dim3(256), 0, s, x, in, c, kh, kw, y, out);
}

// ---------------------------------------------------------------------------
// [group][n][1][T][C] features -> packed sequence rows off[t] + pos[line].
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
to_seq_packed_ragged_kernel(const float* __restrict__ x, RaggedView in, int c, const int32_t* __restrict__ pos,
                            const int32_t* __restrict__ off, float* __restrict__ y) {
    const int g = find_group(in.toff256, in.G, blockIdx.x);
    const int tile = blockIdx.x - in.toff256[g];
    const int T = in.W[g];
    const int64_t rows = (int64_t)in.n[g] * T;  // H == 1
    const float* __restrict__ xg = x + in.poff[g] * c;
    const int32_t* __restrict__ posg = pos + in.loff[g];
    const int cq = c >> 2;
    for (int i = threadIdx.x; i < 256 * cq; i += 256) {
        const int64_t r = (int64_t)tile * 256 + i / cq;
        if (r >= rows) break;
        const int q = i % cq;
        const int t = (int)(r % T);
        const int line = (int)(r / T);
        const float4 v = *reinterpret_cast<const float4*>(xg + r * c + 4 * q);
        *reinterpret_cast<float4*>(y + ((int64_t)off[t] + posg[line]) * c + 4 * q) = v;
    }
}

// The column average that ends the conv stack (AvgPool (H, 1) down to height 1) and the sequence packing in one
// pass: sum of the H rows in ascending order from 0.0f, times 1 / H — pool_ragged_kernel's arithmetic — written
// straight to the packed row of (line, time step).  `in` is the geometry BEFORE the pool (height H), `seq` after it.
__global__ void __launch_bounds__(256)
avgpool_to_seq_ragged_kernel(const float* __restrict__ x, RaggedView in, RaggedView seq, int c,
                             const int32_t* __restrict__ pos, const int32_t* __restrict__ off, float* __restrict__ y) {
    const int g = find_group(seq.toff256, seq.G, blockIdx.x);
    const int tile = blockIdx.x - seq.toff256[g];
    const int T = seq.W[g], ih = in.H;
    const int64_t rows = (int64_t)seq.n[g] * T;
    const float* __restrict__ xg = x + in.poff[g] * c;
    const int32_t* __restrict__ posg = pos + seq.loff[g];
    const int cq = c >> 2;
    const float inv = 1.0f / (float)ih;
    for (int i = threadIdx.x; i < 256 * cq; i += 256) {
        const int64_t r = (int64_t)tile * 256 + i / cq;
        if (r >= rows) break;
        const int q = i % cq;
        const int t = (int)(r % T);
        const int64_t line = r / T;
        const float* xp = xg + (line * ih * T + t) * c + 4 * q;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ky = 0; ky < ih; ky++) {
            const float4 v = *reinterpret_cast<const float4*>(xp + (int64_t)ky * T * c);
            acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
        }
        acc.x = acc.x * inv; acc.y = acc.y * inv; acc.z = acc.z * inv; acc.w = acc.w * inv;
        *reinterpret_cast<float4*>(y + ((int64_t)off[t] + posg[line]) * c + 4 * q) = acc;
    }
}

void avgpool_to_seq_ragged(const float* x, const RaggedView& in, const RaggedView& seq, int c, const int32_t* d_pos,
                           const int32_t* d_off, float* y, hipStream_t s) {
    if (seq.ntiles256 <= 0) return;
    hipLaunchKernelGGL(avgpool_to_seq_ragged_kernel, dim3(seq.ntiles256), dim3(256), 0, s, x, in, seq, c, d_pos, d_off, y);
}

void to_seq_packed_ragged(const float* x, const RaggedView& in, int c, const int32_t* d_pos, const int32_t* d_off,
                          float* y, hipStream_t s) {
    if (in.ntiles256 <= 0) return;
    hipLaunchKernelGGL(to_seq_packed_ragged_kernel, dim3(in.ntiles256), dim3(256), 0, s, x, in, c, d_pos, d_off, y);
}

// ---------------------------------------------------------------------------
// 3x3 / pad 1 convolution (+ bias + ReLU [+ MaxPool 2x1 / 2x2]) on the ragged batch as an
// implicit GEMM on the fp32 matrix cores.  Same GEMM structure as gemm_tiled_kernel
// (128 x BN x BK, LDS-staged k-major operands, double buffer, register prefetch, 2x2 waves of
// 64 x BN/2, k strictly ascending per accumulator).
//
// A block's 128 GEMM rows are a TH x TW patch (4 x 32 or 8 x 16 pixels) of ONE image:
//   * the nine taps of a patch touch (TH+2)(TW+2) = 204 / 180 distinct pixels instead of the
//     3 x 130 = 390 of a 1 x 128 row segment, so the L2 working set of the resident blocks and the
//     re-read traffic roughly halve;
//   * both pixels of a vertical pooling pair and of a horizontal one sit in the SAME lane's
//     accumulator registers (32x32 MFMA C layout: row = (r & 3) + 8 (r >> 2) + 4 half), so the
//     MaxPool that follows conv2 / conv4 / conv6 is a few v_max in the epilogue: the
//     full-resolution activation is never written nor re-read, and the pool launches disappear.
//     max(relu(a), relu(b)) over the window in (py, px) order is exactly MaxPool(ReLU(conv)).
// BK = 16 keeps LDS at 33 KB/block and the kernel at 128 VGPRs, so four blocks (4 waves/SIMD)
// share a CU and cover each other's barrier / LDS-fill phases (tools/conv_variants.sh: 113 TFLOP/s
// vs 109 with three, 102 at BK = 32 with two; s_setprio around the MFMA cluster measured -1 %).
// ---------------------------------------------------------------------------
#ifndef OCRS_CONV_BK
#define OCRS_CONV_BK 16
#endif
constexpr int RG_BM = 128, RG_BK = OCRS_CONV_BK, RG_LDA = RG_BM + 1;

#ifndef OCRS_CONV_WAVES
#define OCRS_CONV_WAVES 4
#endif
#ifndef OCRS_ABL
#define OCRS_ABL 0  // ablation builds (tools/conv_ablation.sh, tools/r6_session.sh instab; results are WRONG on purpose; 8 no epilogue, 16 half of K): 1 no barriers,
#endif              // 2 no global loads, 4 no LDS writes
// FLAT: the patches of a group tile the strip of ALL its images side by side (flat column c = img * Wp + x, Wp = W
// rounded up to PW) instead of each image on its own, so only the last patch of a GROUP is ragged, not the last
// patch of every image (group widths are multiples of 50: W / 4 = 87, 112, 137 ... wasted 7 % of the MFMA rows).
// SPLIT = 3 / 2 (the engine's relaxed / reduced numerics, BN = 128 only): the same patches, activation loads and epilogue,
// the contraction on the bf16 matrix cores with both operands cut into SPLIT bf16 terms — split_mfma.hpp has the arithmetic,
// the LDS layout and the pipeline.  Bw is then the weights' split image (split_weights).
template <int BN, int TW, int PH, int PW, bool FLAT, int SPLIT = 0>   // SPLIT: 0 exact; 3 / 2 = bf16 planes per operand (relaxed / reduced numerics)
#if defined(OCRS_PROBE_ACC_AGPR) || defined(OCRS_PROBE_ALL_AGPR)   // probe builds: room for the AGPR operands
__global__ void __launch_bounds__(256, SPLIT != 0 ? 2 : OCRS_CONV_WAVES)
#else
__global__ void __launch_bounds__(256, SPLIT == 3 ? 2 : SPLIT == 2 ? 3 : OCRS_CONV_WAVES)   // (split: 72 / 48 KB of LDS per block)
#endif
conv3x3_ragged_kernel(const float* __restrict__ X, RaggedView rv, int cin, const float* __restrict__ Bw,
                      const float* __restrict__ bias, int cout, int relu, float* __restrict__ Y,
                      const int64_t* __restrict__ out_poff) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NTW = BN / 64;
    constexpr int TH = RG_BM / TW;
    static_assert(TW == 32 || TW == 16, "patch width");
    static_assert(!SPLIT || (BN == 128 && RG_BK == 16), "split mode: 128-column blocks, 16-deep chunks");
    constexpr int NP = SPLIT ? SPLIT : 1;               // bf16 planes per operand
    // exact: As [2][BK][LDA] fp32, Bs [2][BK][BN] fp32.  split: As [2][NP planes][128 rows][16 bf16], Bs [4][NP][128 columns][16]
    constexpr int SP_PLANE = split::PLANE;              // floats per plane (128 rows x 32 bytes)
    float* As = lds;
    float* Bs = lds + (SPLIT ? 2 * NP * SP_PLANE : 2 * RG_BK * RG_LDA);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int gtile = xcd_remap(blockIdx.x, gridDim.x);
    const int32_t* __restrict__ toff = FLAT ? (PW == 2 ? rv.toff2d_flat2 : rv.toff2d_flat) : rv.toff2d;
    const int g = find_group(toff, rv.G, gtile);
    const int tile = gtile - toff[g];
    const int H = rv.H, W = rv.W[g];
    const int tiles_h = H / TH;
    // FLAT: column strip of the whole group; the tile starts in image img0 at column xf0 and may run into the next ones
    const int Wp = FLAT ? (PW == 2 ? (W + 1) & ~1 : W) : W;
    const int nimg = FLAT ? rv.n[g] : 1;
    int rb, img, x0, span = 1;
    if (FLAT) {
        rb = tile % tiles_h;
        const int c0 = (tile / tiles_h) * TW;
        img = c0 / Wp;
        x0 = c0 - img * Wp;
        span = min(nimg - img, (x0 + TW - 1) / Wp + 1);
    } else {
        const int tiles_w = (W + TW - 1) / TW;
        const int cb = tile % tiles_w;
        const int t2 = tile / tiles_w;
        rb = t2 % tiles_h;
        img = t2 / tiles_h;
        x0 = cb * TW;
    }
    const int y0 = rb * TH;
    const int n0 = blockIdx.y * BN;
    const float* __restrict__ A = X + (rv.poff[g] + (int64_t)img * H * W) * cin;
    const int K = 9 * cin;

    constexpr int AQ = RG_BK / 4;          // float4 per row per chunk
    constexpr int AROWS = 256 / AQ;        // rows covered by one pass of the block
    constexpr int AV = RG_BM / AROWS;      // passes (float4 per thread)
    const int ar = tid / AQ, akq = tid % AQ;
    // Per-thread A addressing, hoisted out of the K loop: the centre-pixel offset of each of the thread's
    // rows and a 9-bit mask of the taps that fall inside the image.  In the loop a load is then
    // base + (uniform tap delta), unconditional from a clamped address, and zeroed by a select — no
    // divergent branches and no 64-bit multiplies (the load path cost 10 % of the kernel: tools/conv_ablation.sh).
    int aoff[AV];        // ((y * W + x) * cin + akq * 4) of the centre pixel, clamped inside the image
    unsigned amask[AV];  // bit (3 * ky + kx) set <=> tap (ky, kx) of this row is inside the image
#pragma unroll
    for (int j = 0; j < AV; j++) {
        const int m = ar + AROWS * j;
        const int py = y0 + m / TW;
        int px = x0 + m % TW, ir = 0;   // FLAT: column px of image img + ir
        if (FLAT) {
#pragma unroll
            for (int q = 0; q < 3; q++) { const bool nx = px >= Wp; px -= nx ? Wp : 0; ir += nx; }  // Wp >= 11 (host)
        }
        unsigned mk = 0;
        if (px < W && ir < span) {  // a column past the image contributes nothing (its outputs are never stored)
#pragma unroll
            for (int t9 = 0; t9 < 9; t9++) {
                const int iy = py + t9 / 3 - 1, ix = px + t9 % 3 - 1;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) mk |= 1u << t9;
            }
        }
        amask[j] = mk;
        aoff[j] = ((min(ir, span - 1) * H + py) * W + min(px, W - 1)) * cin + akq * 4;
    }
    constexpr int BV = RG_BK * BN / 4 / 256;
    // A is fetched from global memory a full 128-byte line (32 channels) per pixel at a time — two
    // consecutive 16-wide K chunks — while LDS staging stays 16-wide: half the A load instructions
    // and 28 fewer VGPRs than fetching per chunk (+3.5 %).
    f32x4 pa0[AV], pa1[AV];
    // Buffer loads: the hardware range check returns 0.0f for the out-of-image taps (their lanes get an
    // offset beyond num_records), so the load is unconditional and needs neither a branch nor a select.
    // (base and size are the same in every lane — they come from blockIdx — but the compiler cannot see that through the
    // group search and kept the descriptor in vector registers, wrapping every load in a "waterfall" loop: 10 extra
    // instructions and a branch per load.  Saying so puts it in scalar registers.)
    const uint64_t a_addr = reinterpret_cast<uint64_t>(A);
    const float* a_uniform = reinterpret_cast<const float*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a_addr >> 32)) << 32) |
                                                            (uint32_t)__builtin_amdgcn_readfirstlane((int)a_addr));
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a_uniform), /*stride*/ 0,
        __builtin_amdgcn_readfirstlane((int)((int64_t)span * H * W * cin * sizeof(float))), 0x00020000);
    constexpr int OOB = 0x40000000;  // 1 GiB: past any tile's images (span * H * W * cin * 4 < 2^30 bytes, checked by the host)
    auto load_a_into = [&](int k0, f32x4 (&d0)[AV], f32x4 (&d1)[AV]) {  // chunks k0 and k0 + RG_BK (same tap: cin % (2*RG_BK) == 0)
        const int tap = k0 / cin;
        const int ci0 = k0 - tap * cin;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int delta = ((ky - 1) * W + (kx - 1)) * cin + ci0;  // wave-uniform
#pragma unroll
        for (int j = 0; j < AV; j++) {
            const int voff = ((amask[j] >> tap) & 1u) ? (aoff[j] + delta) * 4 : OOB;
            const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, voff, 0, 0);
            const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, voff + RG_BK * 4, 0, 0);
            d0[j] = __builtin_bit_cast(f32x4, v0);
            d1[j] = __builtin_bit_cast(f32x4, v1);
        }
    };
    auto load_a_pair = [&](int k0) { load_a_into(k0, pa0, pa1); };
    // B columns past cout (only when cout % BN != 0) are read from a clamped column: their products land
    // in output columns that are never stored.
    int boff[BV];
#pragma unroll
    for (int j = 0; j < BV; j++) {
        const int idx = tid + 256 * j;
        const int kk = idx / (BN / 4), nn = (idx % (BN / 4)) * 4;
        boff[j] = kk * cout + min(n0 + nn, cout - 4);
    }
    // B goes global -> LDS directly (global_load_lds_dwordx4: wave-uniform LDS base + lane * 16 B, per-lane global
    // address): the [k][BN] tile is row-major and contiguous in LDS, so chunk element idx lands at float 4 * idx.
    // No staging registers and no ds_write for B; the barrier that ends the chunk drains the copies (vmcnt(0)).
    auto load_b = [&](int k0, int buf) {
        if constexpr (SPLIT != 0) {
            // Bw = the split image: [column block][chunk][3 planes x 128 columns x 32 bytes] = 12 KB per (block, chunk)
            split::load_weights<NP>(Bw + ((int64_t)blockIdx.y * (K / RG_BK) + k0 / RG_BK) * split::image_floats,
                                    Bs + buf * NP * SP_PLANE, wave, lane);
            return;
        }
        const float* __restrict__ bk = Bw + (int64_t)k0 * cout;
        float* bdst = Bs + buf * RG_BK * BN + wave * 256;  // this wave's 64 float4 slots (wave-uniform)
#pragma unroll
        for (int j = 0; j < BV; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bk + boff[j]),
                                             (__attribute__((address_space(3))) void*)(bdst + 1024 * j), 16, 0, 0);
    };
    auto commit = [&](int buf, const f32x4 (&pa)[AV]) {
        if constexpr (SPLIT != 0) {
            char* base = reinterpret_cast<char*>(As + buf * NP * SP_PLANE);
#pragma unroll
            for (int j = 0; j < AV; j++) split::commit4<NP>(base, ar + AROWS * j, akq, pa[j][0], pa[j][1], pa[j][2], pa[j][3]);
            return;
        }
        float* a = As + buf * RG_BK * RG_LDA;
#pragma unroll
        for (int j = 0; j < AV; j++) {
            const int r = ar + AROWS * j;
            a[(akq * 4 + 0) * RG_LDA + r] = pa[j].x;
            a[(akq * 4 + 1) * RG_LDA + r] = pa[j].y;
            a[(akq * 4 + 2) * RG_LDA + r] = pa[j].z;
            a[(akq * 4 + 3) * RG_LDA + r] = pa[j].w;
        }
    };

    f32x16 acc[2][NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const int col = n0 + wn * (BN / 2) + t * 32 + l31;
        const float bv = col < cout ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[0][t][r] = bv; acc[1][t][r] = bv; }
    }
    auto compute = [&](int buf, int bbuf = -1) {
        if constexpr (SPLIT != 0) {
            split::mma_chunk<NP>(reinterpret_cast<const char*>(As + buf * NP * SP_PLANE),
                                 reinterpret_cast<const char*>(Bs + (bbuf < 0 ? buf : bbuf) * NP * SP_PLANE), wm, wn, l31, half, acc);
            return;
        }
        const float* a = As + buf * RG_BK * RG_LDA + wm * 64 + l31;
        const float* b = Bs + buf * RG_BK * BN + wn * (BN / 2) + l31;
#pragma unroll
        for (int kp = 0; kp < RG_BK / 2; kp++) {
            const int kr = 2 * kp + half;
            const float a0 = a[kr * RG_LDA], a1 = a[kr * RG_LDA + 32];
#pragma unroll
            for (int t = 0; t < NTW; t++) {
                const float bt = b[kr * BN + t * 32];
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bt, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bt, acc[1][t], 0, 0, 0);
            }
        }
    };
    const int nchunks = (OCRS_ABL & 16) ? K / RG_BK / 2 : K / RG_BK;  // even: cin % (2*RG_BK) == 0   (ablation 16: half of K)
    if constexpr (SPLIT != 0) {
        static_assert(AV == 2, "split::pipeline counts four activation loads per chunk pair");
        split::pipeline<NP>(nchunks,   // nchunks % 4 == 0 (cin % 64 == 0: the host checks)
                            [&](int k0, f32x4 (&d0)[AV], f32x4 (&d1)[AV]) { load_a_into(k0, d0, d1); },
                            [&](int k0, int ring) { load_b(k0, ring); },
                            [&](int abuf, const f32x4 (&d)[AV]) { commit(abuf, d); },
                            [&](int abuf, int ring) { compute(abuf, ring); });
    } else {
    load_a_pair(0);
    load_b(0, 0);
    commit(0, pa0);
    __syncthreads();
#define OCRS_SYNC() do { if (!(OCRS_ABL & 1)) __syncthreads(); } while (0)
    for (int c = 0; c < nchunks; c += 2) {
        // even chunk c in buffer 0; chunk c+1's A half is already in registers
        if (!(OCRS_ABL & 2)) load_b((c + 1) * RG_BK, 1);
        compute(0);
        if (!(OCRS_ABL & 4)) commit(1, pa1);
        OCRS_SYNC();
        // odd chunk c+1 in buffer 1; fetch the next pair
        const bool more = c + 2 < nchunks;
        if (more && !(OCRS_ABL & 2)) {
            load_a_pair((c + 2) * RG_BK);
            load_b((c + 2) * RG_BK, 0);
        }
        compute(1);
        if (more && !(OCRS_ABL & 4)) commit(0, pa0);
        OCRS_SYNC();
    }
#undef OCRS_SYNC
    }

    if (OCRS_ABL & 8) {   // ablation: no epilogue (one never-taken store keeps the accumulators alive)
        float sum = 0.0f;
#pragma unroll
        for (int t = 0; t < NTW; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) sum += acc[0][t][r] + acc[1][t][r];
        if (sum == 1.2345e-30f) Y[0] = sum;
        return;
    }
    // ---- epilogue: ReLU, optional in-register MaxPool, store.
    // GEMM row m = wm*64 + i*32 + q, q = (r&3) + 8*(r>>2) + 4*half, is patch pixel (m / TW, m % TW):
    //   TW = 32: ty = 2*wm + i,              tx = q            vertical partner: the other i, same r
    //   TW = 16: ty = 4*wm + 2*i + (r >> 3), tx = q & 15       vertical partner: r + 8, same i
    // horizontal partner (both): r + 1 (tx even <-> (r & 1) == 0).
    const int Ho = H / PH, Wo = W / PW;
    float* __restrict__ C = Y + (out_poff[g] + (int64_t)img * Ho * Wo) * cout;
    auto act = [&](float v) { return relu ? (v > 0.0f ? v : 0.0f) : v; };
    // Unpooled layers, full column tiles: rows as float4 through a wave-private 32 x BN/2 staging tile in the
    // idle operand buffers (gemm_tiled_kernel's epilogue, kernels_nn.hip: a lane holds 16 PIXELS of one channel, a store wants
    // 4 channels of one pixel): 16 dwordx4 stores per wave instead of 64 dword stores.  Per instance, one request at a time,
    // ABAB: 7.39-7.48 -> 6.95-7.01 ms per launch (conv3 + conv5 of the CRNN; the split kernels 4.49 -> 4.34 and 2.73 -> 2.70:
    // split::pipeline ends behind a drained barrier, the operand buffers are idle there too).  The pooled layers store half / a quarter as
    // much and get SLOWER through the tile (9.87-9.89 -> 10.23-10.28 ms, whether each half or both halves' outputs are staged
    // at once): they keep the direct form below.
#ifndef OCRS_DIRECT_EPILOGUES
#define OCRS_DIRECT_EPILOGUES 0   // ablation builds: see kernels_nn.hip
#endif
    if (!OCRS_DIRECT_EPILOGUES && PH == 1 && PW == 1 && n0 + BN <= cout && (cout & 3) == 0 && (((uintptr_t)Y) & 15) == 0) {
        constexpr int WN = BN / 2, LPR = WN / 4, RPI = 64 / LPR, NIT = 32 / RPI;
        static_assert(SPLIT != 0 || 4 * 32 * WN <= 2 * RG_BK * RG_LDA + 2 * RG_BK * BN, "staging must fit the operand tiles");   // split: 48 / 72 KB
        float* stage = lds + wave * (32 * WN);
        const int srow = lane / LPR, sc4 = (lane % LPR) * 4;
        int soff[NIT];   // element offset of (image, column, first channel) of the accumulator rows this lane stores, or -1
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int q = it * RPI + srow;
            int x = x0 + (TW == 32 ? q : (q & 15)), ir = 0;
            if (FLAT) {
#pragma unroll
                for (int k = 0; k < 3; k++) { const bool nx = x >= Wp; x -= nx ? Wp : 0; ir += nx; }
            }
            soff[it] = (x < W && (!FLAT || ir < span)) ? ((FLAT ? ir * Ho * Wo : 0) + x) * cout + n0 + wn * WN + sc4 : -1;
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
#pragma unroll
            for (int t = 0; t < NTW; t++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    stage[((r & 3) + 8 * (r >> 2) + 4 * half) * WN + t * 32 + l31] = act(acc[i][t][r]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave's LDS operations execute in order; this stops the COMPILER
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int q = it * RPI + srow;
                if (soff[it] < 0) continue;
                const int ty = TW == 32 ? 2 * wm + i : 4 * wm + 2 * i + (q >> 4);
                *reinterpret_cast<f32x4*>(&C[soff[it] + (y0 + ty) * Wo * cout]) = *reinterpret_cast<const f32x4*>(&stage[q * WN + sc4]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
    }
    // FLAT: a lane's accumulator rows cover NX distinct patch columns (tx = q & (TW - 1) below); their image, column and
    // validity are worked out once: xo_off[jx] = element offset of (image, pooled column) from C, or -1.
    constexpr int NX = TW == 16 ? 8 : 16;
    int xo_off[FLAT ? NX : 1];
    if (FLAT) {
#pragma unroll
        for (int jx = 0; jx < NX; jx++) {
            if (PW == 2 && (jx & 1)) { xo_off[jx] = -1; continue; }
            const int q = (jx & 3) + 8 * (jx >> 2) + 4 * half;
            int x = x0 + (TW == 32 ? q : (q & 15)), ir = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) { const bool nx = x >= Wp; x -= nx ? Wp : 0; ir += nx; }
            xo_off[jx] = (x + (PW - 1) < W && ir < span) ? (ir * Ho * Wo + x / PW) * cout : -1;
        }
    }
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const int col = n0 + wn * (BN / 2) + t * 32 + l31;
        if (col >= cout) continue;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (PH == 2 && TW == 32 && i == 1) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (PH == 2 && TW == 16 && (r & 8)) continue;
                if (PW == 2 && (r & 1)) continue;
                const int q = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int ty = TW == 32 ? 2 * wm + i : 4 * wm + 2 * i + (r >> 3);
                const int tx = TW == 32 ? q : (q & 15);
                const int x = x0 + tx;
                if (FLAT ? xo_off[TW == 16 ? (r & 7) : r] < 0 : x + (PW - 1) >= W) continue;
                constexpr int VI = (PH == 2 && TW == 32) ? 1 : 0;   // partner's i offset
                constexpr int VR = (PH == 2 && TW == 16) ? 8 : 0;   // partner's r offset
                float m = act(acc[i][t][r]);
                if (PW == 2) { const float v = act(acc[i][t][(r + 1) & 15]); m = v > m ? v : m; }
                if (PH == 2) {
                    { const float v = act(acc[(i + VI) & 1][t][(r + VR) & 15]); m = v > m ? v : m; }
                    if (PW == 2) { const float v = act(acc[(i + VI) & 1][t][(r + VR + 1) & 15]); m = v > m ? v : m; }
                }
                const int yo = (y0 + ty) / PH;
                if (FLAT) C[xo_off[TW == 16 ? (r & 7) : r] + yo * Wo * cout + col] = m;
                else C[((int64_t)yo * Wo + x / PW) * cout + col] = m;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// conv1 (Cin = 1) + ReLU + MaxPool 2x2 + conv2 (32 -> 64) + ReLU + MaxPool 2x2 in ONE kernel (round 3).
// conv1's pooled output (2.9 GB for a 16-page request) was written once and read back nine times, tap by tap, by
// conv2, and conv1 itself is a VALU kernel that the stream carrying the conv stacks has to wait for (1.4 ms alone,
// 2.4 ms live).  Here a block computes conv1 for the 10 x 18 positions its 8 x 16 patch of conv2 outputs needs — on
// the VALU, into an LDS tile [position][32 channels] — and conv2's MFMA loop takes its A operand straight from that
// tile (no global A loads, no register staging, no ds_write commits); only conv2's weights stream through LDS as
// before.  The conv1 work of a block (41 % more positions than the patch has: the halo) overlaps the MFMA phase of
// the three other blocks on the CU.
// Numerics: conv1 = the fmaf chain of conv1_relu_pool_ragged_kernel, position by position; conv2 = the k-ascending
// MFMA chain of conv3x3_ragged_kernel from acc = bias (k = tap * 32 + channel); zero padding of conv2 = tile entries
// of positions outside the image are 0.  The patches tile the group's strip of images with at least one empty column
// between two images (flat column = image * Wg + x, Wg = W + 1 rounded up to even), so that a position is either a
// pixel of ONE image or padding for both of its neighbours.
// ---------------------------------------------------------------------------
#ifndef OCRS_F12_ABL
#define OCRS_F12_ABL 0   // ablation builds (tools/conv12_ablation.sh; wrong results on purpose): 1 no conv1 stage, 2 no weight loads, 4 an eighth of the MFMAs
#endif
constexpr int F12_MID = 32, F12_COUT = 64, F12_TW = 16, F12_TH = 8, F12_HW = F12_TW + 2, F12_HH = F12_TH + 2;
constexpr int F12_NPOS = F12_HH * F12_HW;          // 180 halo positions
constexpr int F12_LD = F12_MID + 1;                // tile row stride (floats): consecutive positions -> consecutive banks
#ifndef OCRS_F12_BK
#define OCRS_F12_BK 16   // 32 (half the barriers, four blocks per CU instead of five) measured the same
#endif
constexpr int F12_BK = OCRS_F12_BK;                // K chunk of conv2's weight stream (one barrier per chunk)
constexpr size_t F12_LDS = (size_t)(F12_NPOS * F12_LD + 2 * F12_BK * F12_COUT) * sizeof(float);

__global__ void __launch_bounds__(256, OCRS_CONV_WAVES)
conv12_fused_kernel(const float* __restrict__ X0, RaggedView in0, RaggedView mid, const float* __restrict__ w1,
                    const float* __restrict__ b1, const float* __restrict__ Bw, const float* __restrict__ bias,
                    float* __restrict__ Y, const int64_t* __restrict__ out_poff) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BN = F12_COUT, TW = F12_TW, TH = F12_TH, LD = F12_LD;
    float* T = lds;                                  // [NPOS][LD]
    float* Bs = lds + F12_NPOS * LD;                 // [2][F12_BK][BN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int gtile = xcd_remap(blockIdx.x, gridDim.x);
    const int g = find_group(mid.toff2d_gap, mid.G, gtile);
    const int tile = gtile - mid.toff2d_gap[g];
    const int H = mid.H, W = mid.W[g];               // conv2's input geometry (after conv1's pool)
    const int H0 = in0.H, W0 = in0.W[g];             // the line images
    const int tiles_h = H / TH;
    const int Wg = (W + 2) & ~1;
    const int nimg = mid.n[g];
    const int rb = tile % tiles_h;
    const int c0 = (tile / tiles_h) * TW;
    const int img = c0 / Wg;
    const int x0 = c0 - img * Wg;
    const int span = min(nimg - img, (x0 + TW - 1) / Wg + 1);
    const int y0 = rb * TH;

    // ---- conv2 weights: first chunk on its way while conv1 runs
    constexpr int BV = F12_BK * BN / 4 / 256;        // float4 of B per thread and chunk
    const int boff = (tid / (BN / 4)) * BN + (tid % (BN / 4)) * 4;
    auto load_b = [&](int k0, int buf) {
#pragma unroll
        for (int j = 0; j < BV; j++)   // chunk element idx = tid + 256 j lands at float 4 idx (row-major [k][BN])
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Bw + (int64_t)(k0 + 16 * j) * BN + boff),
                                             (__attribute__((address_space(3))) void*)(Bs + buf * F12_BK * BN + wave * 256 + 1024 * j), 16, 0, 0);
    };
    load_b(0, 0);

    // ---- stage 1: conv1 + ReLU + pool for the halo positions -> T.  Item = (position, 4 channels); a thread keeps
    // its channel quad (256 % 8 == 0) and with it 9 x 4 weights + 4 biases in registers.
    {
        const int q = tid & 7;
        // channel pairs as 2-vectors: v_pk_fma_f32 (two IEEE fmas per instruction; the compiler packs the stand-alone
        // conv1 kernel by itself, not this loop)
        f32x2 wq[9][2], bq[2];
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int c = 0; c < 2; c++) wq[t][c] = f32x2{w1[t * F12_MID + 4 * q + 2 * c], w1[t * F12_MID + 4 * q + 2 * c + 1]};
#pragma unroll
        for (int c = 0; c < 2; c++) bq[c] = f32x2{b1[4 * q + 2 * c], b1[4 * q + 2 * c + 1]};
        const float* __restrict__ xg = X0 + in0.poff[g];
        for (int p = tid >> 3; p < ((OCRS_F12_ABL & 1) ? 0 : F12_NPOS); p += 32) {
            const int hy = p / F12_HW, hx = p - hy * F12_HW;
            const int y = y0 - 1 + hy;
            int x = x0 - 1 + hx, ir = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) { const bool nx = x >= Wg; x -= nx ? Wg : 0; ir += nx; }   // W >= 11 (host)
            f32x2 m[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
            if ((unsigned)y < (unsigned)H && x >= 0 && x < W && ir < span) {
                const float* xi = xg + (int64_t)(img + ir) * H0 * W0;
                float pt[4][4];
                const int iy0 = 2 * y - 1, ix0 = 2 * x - 1;
                if (iy0 >= 0 && iy0 + 3 < H0 && ix0 >= 0 && ix0 + 3 < W0) {
                    const float* pp = xi + (int64_t)iy0 * W0 + ix0;
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++) pt[a][b] = pp[a * W0 + b];
                } else {
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const int iy = iy0 + a, ix = ix0 + b;
                            pt[a][b] = ((unsigned)iy < (unsigned)H0 && (unsigned)ix < (unsigned)W0) ? xi[(int64_t)iy * W0 + ix] : 0.0f;
                        }
                }
#pragma unroll
                for (int py = 0; py < 2; py++)
#pragma unroll
                    for (int px = 0; px < 2; px++) {
                        f32x2 acc[2] = {bq[0], bq[1]};
#pragma unroll
                        for (int ky = 0; ky < 3; ky++)
#pragma unroll
                            for (int kx = 0; kx < 3; kx++) {
                                const float xs = pt[py + ky][px + kx];
                                const f32x2 xv = f32x2{xs, xs};
#pragma unroll
                                for (int c = 0; c < 2; c++) acc[c] = __builtin_elementwise_fma(xv, wq[ky * 3 + kx][c], acc[c]);
                            }
                        // max over the window of ReLU(v) = ReLU(max over the window of v): the largest positive value
                        // (equal values have equal bits) or +0; v_max_f32 ignores a NaN operand exactly as
                        // "v > m ? v : m" from m = ReLU(.) does; an all-NaN window ends as NaN > 0 ? . : 0 = +0.
#pragma unroll
                        for (int c = 0; c < 2; c++) m[c] = (py == 0 && px == 0) ? acc[c] : __builtin_elementwise_max(acc[c], m[c]);
                    }
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    m[c].x = m[c].x > 0.0f ? m[c].x : 0.0f;
                    m[c].y = m[c].y > 0.0f ? m[c].y : 0.0f;
                }
            }
            float* t = T + p * LD + 4 * q;
            t[0] = m[0].x; t[1] = m[0].y; t[2] = m[1].x; t[3] = m[1].y;
        }
    }

    // ---- stage 2: conv2 on the matrix cores, A from the tile
    f32x16 acc[2];
    {
        const int col = wn * 32 + l31;
        const float bv = bias[col];
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[0][r] = bv; acc[1][r] = bv; }
    }
    // GEMM row m = wm * 64 + i * 32 + l31 is patch pixel (m / 16, m % 16); its tap (ky, kx), channel c sits at
    // T[((m / 16 + ky) * 18 + m % 16 + kx) * LD + c]
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int m = wm * 64 + i * 32 + l31;
        abase[i] = ((m / TW) * F12_HW + (m % TW)) * LD + half;
    }
    __syncthreads();     // T complete, B chunk 0 landed (the barrier's vmcnt(0))
    constexpr int NCH = 9 * F12_MID / F12_BK;         // chunks (F12_BK divides 32: a chunk stays inside one tap)
    static_assert(F12_MID % F12_BK == 0 && F12_BK % 16 == 0, "chunk size");
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        if (c + 1 < NCH && !(OCRS_F12_ABL & 2)) load_b((c + 1) * F12_BK, (c + 1) & 1);
        const int tap = (c * F12_BK) / F12_MID, ch0 = (c * F12_BK) % F12_MID;
        const int toff = ((tap / 3) * F12_HW + tap % 3) * LD + ch0;
        const float* b = Bs + (c & 1) * F12_BK * BN + wn * 32 + l31;
#pragma unroll
        for (int kp = 0; kp < ((OCRS_F12_ABL & 4) ? 1 : F12_BK / 2); kp++) {
            const float a0 = T[abase[0] + toff + 2 * kp], a1 = T[abase[1] + toff + 2 * kp];
            const float bt = b[(2 * kp + half) * BN];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bt, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bt, acc[1], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: ReLU + MaxPool 2x2 in registers (accumulator row r <-> patch pixel as in conv3x3_ragged_kernel,
    // TW = 16: ty = 4 wm + 2 i + (r >> 3), tx = (r & 3) + 8 ((r >> 2) & 1) + 4 half), store the pooled pixel
    const int Ho = H / 2, Wo = W / 2;
    float* __restrict__ C = Y + (out_poff[g] + (int64_t)img * Ho * Wo) * BN;
    auto act = [&](float v) { return v > 0.0f ? v : 0.0f; };
    const int col = wn * 32 + l31;
    int xo_off[8];
#pragma unroll
    for (int jx = 0; jx < 8; jx++) {
        if (jx & 1) { xo_off[jx] = -1; continue; }
        int x = x0 + (jx & 3) + 8 * (jx >> 2) + 4 * half, ir = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) { const bool nx = x >= Wg; x -= nx ? Wg : 0; ir += nx; }
        xo_off[jx] = (x + 1 < W && ir < span) ? (ir * Ho * Wo + x / 2) * BN : -1;
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            if (xo_off[r & 7] < 0) continue;
            const float m = act(__builtin_fmaxf(__builtin_fmaxf(acc[i][r], acc[i][r + 1]), __builtin_fmaxf(acc[i][r + 8], acc[i][r + 9])));
            const int yo = (y0 + 4 * wm + 2 * i) / 2;
            C[xo_off[r & 7] + yo * Wo * BN + col] = m;
        }
}

// ---------------------------------------------------------------------------
// The same launch with conv2's contraction on the bf16 matrix cores (relaxed / reduced numerics, split_mfma.hpp): conv1 is
// computed exactly as above (fp32 VALU, the spec's chain) and CUT as it is written — the tile holds NP planes of
// [position][32 channels] bf16 instead of fp32 values — conv2's weights come pre-cut from the model (conv12_split_weights:
// per tap, per plane, [64 columns][32 k]), one tap (K = 32, two MFMA k-steps) per barrier through a ring of three LDS
// buffers.  Tile rows are 20 positions apart (18 used) and the 16-byte slot of a position's 64-byte row is XORed with
// (hx >> 2) & 3; the weight rows with (n >> 2) & 3: both operand reads (ds_read_b128, lane = pixel or column, half-wave = k
// half) are bank-conflict free for every tap.
// ---------------------------------------------------------------------------
constexpr int F12S_RS = 20;                                   // tile row stride in positions
constexpr int F12S_TILE = F12_HH * F12S_RS * 64;              // bytes per plane of the tile
constexpr int F12S_BTAP = F12_COUT * 64;                      // bytes per plane of one tap's weights
constexpr size_t f12s_lds(int np) { return (size_t)np * (F12S_TILE + 3 * F12S_BTAP); }   // 74.4 KB (NP 3) / 49.6 KB (NP 2)

template <int NP>
#if defined(OCRS_PROBE_ACC_AGPR) || defined(OCRS_PROBE_ALL_AGPR)
__global__ void __launch_bounds__(256, 2)
#else
__global__ void __launch_bounds__(256, NP == 3 ? 2 : 3)
#endif
conv12_fused_split_kernel(const float* __restrict__ X0, RaggedView in0, RaggedView mid, const float* __restrict__ w1,
                          const float* __restrict__ b1, const uint16_t* __restrict__ Bimg, const float* __restrict__ bias,
                          float* __restrict__ Y, const int64_t* __restrict__ out_poff) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BN = F12_COUT, TW = F12_TW, TH = F12_TH;
    char* T = reinterpret_cast<char*>(lds);                  // [NP][HH * RS][64 B]
    char* Bs = T + NP * F12S_TILE;                           // [3 taps][NP][64 columns][64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int gtile = xcd_remap(blockIdx.x, gridDim.x);
    const int g = find_group(mid.toff2d_gap, mid.G, gtile);
    const int tile = gtile - mid.toff2d_gap[g];
    const int H = mid.H, W = mid.W[g];
    const int H0 = in0.H, W0 = in0.W[g];
    const int tiles_h = H / TH;
    const int Wg = (W + 2) & ~1;
    const int nimg = mid.n[g];
    const int rb = tile % tiles_h;
    const int c0 = (tile / tiles_h) * TW;
    const int img = c0 / Wg;
    const int x0 = c0 - img * Wg;
    const int span = min(nimg - img, (x0 + TW - 1) / Wg + 1);
    const int y0 = rb * TH;

    // conv2 weights of tap t -> ring buffer t % 3: NP planes of 4 KB, one 16-byte piece per thread and plane
    auto load_b = [&](int tap) {
        const char* src = reinterpret_cast<const char*>(Bimg) + (size_t)tap * 3 * F12S_BTAP;     // (the image always holds 3 planes)
        char* dst = Bs + (tap % 3) * NP * F12S_BTAP;
#pragma unroll
        for (int p = 0; p < NP; p++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * F12S_BTAP + wave * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + p * F12S_BTAP + wave * 1024), 16, 0, 0);
    };
    load_b(0);
    load_b(1);

    // ---- stage 1: conv1 + ReLU + pool for the halo positions (the exact kernel's arithmetic), cut into the planes
    {
        const int q = tid & 7;
        f32x2 wq[9][2], bq[2];
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int c = 0; c < 2; c++) wq[t][c] = f32x2{w1[t * F12_MID + 4 * q + 2 * c], w1[t * F12_MID + 4 * q + 2 * c + 1]};
#pragma unroll
        for (int c = 0; c < 2; c++) bq[c] = f32x2{b1[4 * q + 2 * c], b1[4 * q + 2 * c + 1]};
        const float* __restrict__ xg = X0 + in0.poff[g];
        auto cut = [](float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2)); };
        auto lo_f = [](unsigned pk) { return __uint_as_float(pk << 16); };
        auto hi_f = [](unsigned pk) { return __uint_as_float(pk & 0xFFFF0000u); };
        for (int p = tid >> 3; p < F12_NPOS; p += 32) {
            const int hy = p / F12_HW, hx = p - hy * F12_HW;
            const int y = y0 - 1 + hy;
            int x = x0 - 1 + hx, ir = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) { const bool nx = x >= Wg; x -= nx ? Wg : 0; ir += nx; }
            f32x2 m[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
            if ((unsigned)y < (unsigned)H && x >= 0 && x < W && ir < span) {
                const float* xi = xg + (int64_t)(img + ir) * H0 * W0;
                float pt[4][4];
                const int iy0 = 2 * y - 1, ix0 = 2 * x - 1;
                if (iy0 >= 0 && iy0 + 3 < H0 && ix0 >= 0 && ix0 + 3 < W0) {
                    const float* pp = xi + (int64_t)iy0 * W0 + ix0;
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++) pt[a][b] = pp[a * W0 + b];
                } else {
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const int iy = iy0 + a, ix = ix0 + b;
                            pt[a][b] = ((unsigned)iy < (unsigned)H0 && (unsigned)ix < (unsigned)W0) ? xi[(int64_t)iy * W0 + ix] : 0.0f;
                        }
                }
#pragma unroll
                for (int py = 0; py < 2; py++)
#pragma unroll
                    for (int px = 0; px < 2; px++) {
                        f32x2 acc[2] = {bq[0], bq[1]};
#pragma unroll
                        for (int ky = 0; ky < 3; ky++)
#pragma unroll
                            for (int kx = 0; kx < 3; kx++) {
                                const float xs = pt[py + ky][px + kx];
                                const f32x2 xv = f32x2{xs, xs};
#pragma unroll
                                for (int c = 0; c < 2; c++) acc[c] = __builtin_elementwise_fma(xv, wq[ky * 3 + kx][c], acc[c]);
                            }
#pragma unroll
                        for (int c = 0; c < 2; c++) m[c] = (py == 0 && px == 0) ? acc[c] : __builtin_elementwise_max(acc[c], m[c]);
                    }
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    m[c].x = m[c].x > 0.0f ? m[c].x : 0.0f;
                    m[c].y = m[c].y > 0.0f ? m[c].y : 0.0f;
                }
            }
            // channels 4q .. 4q + 3 of position (hy, hx): 8 bytes per plane, slot (q >> 1) ^ ((hx >> 2) & 3)
            const int off = (hy * F12S_RS + hx) * 64 + ((((q >> 1) ^ (hx >> 2)) & 3) << 4) + ((q & 1) << 3);
            u32x2 ph = {cut(m[0].x, m[0].y), cut(m[1].x, m[1].y)};
            const float r0 = m[0].x - lo_f(ph[0]), r1 = m[0].y - hi_f(ph[0]), r2 = m[1].x - lo_f(ph[1]), r3 = m[1].y - hi_f(ph[1]);
            u32x2 pm = {cut(r0, r1), cut(r2, r3)};
            *reinterpret_cast<u32x2*>(T + off) = ph;
            *reinterpret_cast<u32x2*>(T + F12S_TILE + off) = pm;
            if (NP == 3) {
                u32x2 pl = {cut(r0 - lo_f(pm[0]), r1 - hi_f(pm[0])), cut(r2 - lo_f(pm[1]), r3 - hi_f(pm[1]))};
                *reinterpret_cast<u32x2*>(T + 2 * F12S_TILE + off) = pl;
            }
        }
    }

    // ---- stage 2: conv2 on the bf16 matrix cores
    f32x16 acc[2];
    {
        const float bv = bias[wn * 32 + l31];
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[0][r] = bv; acc[1][r] = bv; }
    }
    const int mx = l31 & 15;
    int pos0[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int m = wm * 64 + i * 32 + l31;
        pos0[i] = (m / TW) * F12S_RS + mx;
    }
    const int nb = wn * 32 + l31;
    const int boff = nb * 64, bsw = (nb >> 2) & 3;
    // (asm "memory" clobbers around every bare barrier: the builtin is no fence to the compiler, which otherwise hoists the next
    // tap's first operand reads above the wait + barrier that guarantee the tap's weights have landed — r6, split_mfma.hpp)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0): taps 0 and 1 landed, this thread's tile writes done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
        if (tap + 2 < 9) load_b(tap + 2);
        const int ky = tap / 3, kx = tap % 3;
        const int asw = ((mx + kx) >> 2) & 3;
        const char* bb = Bs + (tap % 3) * NP * F12S_BTAP + boff;
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            const int qa = (((2 * s2 + half) ^ asw) & 3) << 4, qb = (((2 * s2 + half) ^ bsw) & 3) << 4;
            bf16x8 af[2][3], bfr[3];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int p = 0; p < NP; p++) af[i][p] = *reinterpret_cast<const bf16x8*>(T + p * F12S_TILE + (pos0[i] + ky * F12S_RS + kx) * 64 + qa);
#pragma unroll
            for (int p = 0; p < NP; p++) bfr[p] = *reinterpret_cast<const bf16x8*>(bb + p * F12S_BTAP + qb);
#define OCRS_TERM12(PA, PB)                                                                                   \
            OCRS_SPLIT_MMA(af[0][PA], bfr[PB], acc[0]);                                                       \
            OCRS_SPLIT_MMA(af[1][PA], bfr[PB], acc[1]);
            if (NP == 3) { OCRS_TERM12(NP - 1, 0) OCRS_TERM12(0, NP - 1) OCRS_TERM12(1, 1) }
            OCRS_TERM12(1, 0) OCRS_TERM12(0, 1) OCRS_TERM12(0, 0)
#undef OCRS_TERM12
        }
        // the next tap's weights must have landed (the tap after it, just requested, may stay in flight); every wave is
        // done with this tap's buffer before it is overwritten two taps from now
        // ... and lgkmcnt(0): this wave's own operand reads of the tap's buffer have RETURNED before it crosses the barrier.  Until
        // round 6 the wait was vmcnt only; the compiler sinks the tap's last MFMAs below the barrier, so their ds_reads could
        // still be queued in the LDS unit when another wave's copy of tap + 2 (L2-hot weights: 250-400 cycles to land) overwrote
        // the buffer: one tile element wrong in ~1 000 requests under load, found by the canary.
        asm volatile("" ::: "memory");
        if (tap + 2 < 9) __builtin_amdgcn_s_waitcnt(0x0070 | NP);   // vmcnt(NP) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ---- epilogue: as conv12_fused_kernel
    const int Ho = H / 2, Wo = W / 2;
    float* __restrict__ C = Y + (out_poff[g] + (int64_t)img * Ho * Wo) * BN;
    auto act = [&](float v) { return v > 0.0f ? v : 0.0f; };
    const int col = wn * 32 + l31;
    int xo_off[8];
#pragma unroll
    for (int jx = 0; jx < 8; jx++) {
        if (jx & 1) { xo_off[jx] = -1; continue; }
        int x = x0 + (jx & 3) + 8 * (jx >> 2) + 4 * half, ir = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) { const bool nx = x >= Wg; x -= nx ? Wg : 0; ir += nx; }
        xo_off[jx] = (x + 1 < W && ir < span) ? (ir * Ho * Wo + x / 2) * BN : -1;
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            if (xo_off[r & 7] < 0) continue;
            const float m = act(__builtin_fmaxf(__builtin_fmaxf(acc[i][r], acc[i][r + 1]), __builtin_fmaxf(acc[i][r + 8], acc[i][r + 9])));
            const int yo = (y0 + 4 * wm + 2 * i) / 2;
            C[xo_off[r & 7] + yo * Wo * BN + col] = m;
        }
}

// conv2's weights [9 taps x 32 channels][64 columns] for conv12_fused_split_kernel: per tap three planes (hi, mid, lo) of
// [column][32 k] bf16, the 16-byte slots of a column's 64-byte row XORed with (column >> 2) & 3.  Host.
void conv12_split_weights(const float* w, std::vector<uint16_t>* out) {
    out->assign((size_t)9 * 3 * F12_COUT * 32, 0);
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    auto from = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    auto rne = [&](float f) { const uint32_t u = bits(f); return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16; };
    for (int tap = 0; tap < 9; tap++)
        for (int n = 0; n < F12_COUT; n++)
            for (int k = 0; k < 32; k++) {
                const float x = w[(size_t)(tap * 32 + k) * F12_COUT + n];
                const uint32_t hb = rne(x);
                const float r1 = x - from(hb << 16);
                const uint32_t mb = rne(r1);
                const uint32_t lb = rne(r1 - from(mb << 16));
                const size_t e = (size_t)n * 32 + (size_t)((((k >> 3) ^ (n >> 2)) & 3) * 8) + (k & 7);
                uint16_t* img = out->data() + (size_t)tap * 3 * F12_COUT * 32;
                img[e] = (uint16_t)hb;
                img[F12_COUT * 32 + e] = (uint16_t)mb;
                img[2 * F12_COUT * 32 + e] = (uint16_t)lb;
            }
}

bool conv12_fused_ragged(const float* x, const RaggedView& in0, const RaggedView& mid, const float* w1, const float* b1,
                         int c1, const float* w2, const float* b2, int c2, float* y, const RaggedView& out, hipStream_t s,
                         const uint16_t* w2split) {
    // x == nullptr: only asks whether the shape has this kernel
    if (option(OPT_CONV12_FUSE) == 0) return false;
    if (c1 != F12_MID || c2 != F12_COUT || mid.H % F12_TH != 0 || in0.H != 2 * mid.H || out.H * 2 != mid.H) return false;
    if (mid.min_w < 11 || mid.ntiles2d_gap <= 0) return false;
    // 32-bit element offsets inside the images one patch touches
    if (mid.max_tile_px_ * 4 * (int64_t)sizeof(float) >= (int64_t)1 << 30 ||
        mid.max_tile_px_ * F12_COUT * (int64_t)sizeof(float) >= (int64_t)1 << 30) return false;
    if (!x) return true;
    const int numerics = option(OPT_NUMERICS);
    if (numerics != 0 && w2split) {   // relaxed / reduced: conv2's contraction on the bf16 matrix cores
        static std::atomic<uint64_t> ok3{0}, ok2{0};
        if (numerics == 2) {
            allow_dynamic_lds(reinterpret_cast<const void*>(&conv12_fused_split_kernel<2>), ok2);
            hipLaunchKernelGGL((conv12_fused_split_kernel<2>), dim3(mid.ntiles2d_gap), dim3(256), f12s_lds(2), s, x, in0, mid, w1, b1, w2split, b2, y, out.poff);
        } else {
            allow_dynamic_lds(reinterpret_cast<const void*>(&conv12_fused_split_kernel<3>), ok3);
            hipLaunchKernelGGL((conv12_fused_split_kernel<3>), dim3(mid.ntiles2d_gap), dim3(256), f12s_lds(3), s, x, in0, mid, w1, b1, w2split, b2, y, out.poff);
        }
        return true;
    }
    hipLaunchKernelGGL(conv12_fused_kernel, dim3(mid.ntiles2d_gap), dim3(256), F12_LDS, s, x, in0, mid, w1, b1, w2, b2, y, out.poff);
    return true;
}

// rv carries the 2-D tiling of the INPUT geometry (toff2d / ntiles2d / tw); `out` is the geometry after the
// fused pool (== rv when ph == pw == 1).  Returns false if the shape is not supported.
// (split_mfma.hpp) the split image of a row-major [K][N] weight matrix with row pitch ldw
void split_weights(const float* w, int K, int N, int ldw, std::vector<uint16_t>* out) {
    const int nblk = N / 128, nch = K / 16;
    out->assign((size_t)nblk * nch * 3 * 128 * 16, 0);
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    auto from = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    auto rne = [&](float f) { const uint32_t u = bits(f); return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16; };   // fp32 -> bf16, round to nearest even (finite weights)
    for (int nb = 0; nb < nblk; nb++)
        for (int c = 0; c < nch; c++) {
            uint16_t* img = out->data() + ((size_t)nb * nch + c) * 3 * 128 * 16;
            for (int n = 0; n < 128; n++)
                for (int kl = 0; kl < 16; kl++) {
                    const float x = w[(size_t)(c * 16 + kl) * ldw + nb * 128 + n];
                    const uint32_t hb = rne(x);
                    const float r1 = x - from(hb << 16);          // exact
                    const uint32_t mb = rne(r1);
                    const uint32_t lb = rne(r1 - from(mb << 16));
                    const int half = kl >> 3, j = kl & 7;
                    const size_t e = (size_t)n * 16 + (size_t)(((half ^ (n >> 3)) & 1) * 8) + j;
                    img[e] = (uint16_t)hb;
                    img[128 * 16 + e] = (uint16_t)mb;
                    img[2 * 128 * 16 + e] = (uint16_t)lb;
                }
        }
}

bool conv3x3_ragged(const float* x, const RaggedView& rv, int cin, const float* wt, const float* bias, int cout, int relu,
                    int ph, int pw, float* y, const RaggedView& out, hipStream_t s, const uint16_t* wsplit) {
    if ((cin % (2 * RG_BK)) != 0 || (cout % 4) != 0 || cout < 64) return false;
    if (!((ph == 1 && pw == 1) || (ph == 2 && (pw == 1 || pw == 2)))) return false;
    if ((ph == 2 || pw == 2) && !relu) return false;  // the fused pool assumes ReLU'd (non-negative, NaN-free order) inputs
    if (rv.tw != 32 && rv.tw != 16) return false;
    if (rv.H % (RG_BM / rv.tw) != 0) return false;
    // flat tiling: 32-bit element offsets inside a tile's images, at most three image borders per patch row
    const bool flat = option(OPT_CONV_FLAT) != 0 && rv.min_w >= 11 &&
                      rv.max_tile_px_ * std::max(cin, cout) * (int64_t)sizeof(float) < (int64_t)1 << 30;
    // A is addressed through a buffer descriptor per tile (its images) with 32-bit byte offsets (OOB marker at 1 GiB)
    if (!flat && (int64_t)rv.H * rv.max_w * cin * (int64_t)sizeof(float) >= (int64_t)1 << 30) return false;
    const int ntiles = flat ? (pw == 2 ? rv.ntiles2d_flat2 : rv.ntiles2d_flat) : rv.ntiles2d;
    if (ntiles <= 0) return true;
    const bool n64 = cout <= 64;
    // relaxed numerics (ocrs_engine_params.numerics): the bf16-split contraction, where the model carries the split weights
    const int numerics = option(OPT_NUMERICS);   // 0 exact, 1 relaxed (three bf16 planes), 2 reduced (two)
    const bool split = wsplit && !n64 && (cout % 128) == 0 && (cin % 64) == 0 && numerics != 0;
    const size_t lds = split ? split::lds_bytes(numerics == 2 ? 2 : 3)   // A: 2 buffers, B: ring of 4; 4 KB per plane
                             : (size_t)(2 * RG_BK * RG_LDA + 2 * RG_BK * (n64 ? 64 : 128)) * sizeof(float);
    const dim3 grid(ntiles, (cout + (n64 ? 63 : 127)) / (n64 ? 64 : 128));
#define OCRS_LAUNCH_SPLIT(TW_, PH_, PW_, FLAT_, NP_, SLOT_)                                                              \
    if (flat == FLAT_ && (numerics == 2 ? 2 : 3) == NP_) {                                                              \
        allow_dynamic_lds(reinterpret_cast<const void*>(&conv3x3_ragged_kernel<128, TW_, PH_, PW_, FLAT_, NP_>), ok_[SLOT_]);   \
        hipLaunchKernelGGL((conv3x3_ragged_kernel<128, TW_, PH_, PW_, FLAT_, NP_>), grid, dim3(256), lds, s, x, rv, cin, ws_, bias, \
                           cout, relu, y, out.poff);                                                                    \
    }
#define OCRS_LAUNCH_CONV(BN_, TW_, PH_, PW_)                                                                            \
    do {                                                                                                                \
        if (split && BN_ == 128) {                                                                                      \
            const float* ws_ = reinterpret_cast<const float*>(wsplit);                                                  \
            static std::atomic<uint64_t> ok_[4] = {};                                                                    \
            OCRS_LAUNCH_SPLIT(TW_, PH_, PW_, true, 3, 0) OCRS_LAUNCH_SPLIT(TW_, PH_, PW_, false, 3, 1)                     \
            OCRS_LAUNCH_SPLIT(TW_, PH_, PW_, true, 2, 2) OCRS_LAUNCH_SPLIT(TW_, PH_, PW_, false, 2, 3)                     \
        } else if (flat) hipLaunchKernelGGL((conv3x3_ragged_kernel<BN_, TW_, PH_, PW_, true>), grid, dim3(256), lds, s, x, rv, \
                                     cin, wt, bias, cout, relu, y, out.poff);                                           \
        else hipLaunchKernelGGL((conv3x3_ragged_kernel<BN_, TW_, PH_, PW_, false>), grid, dim3(256), lds, s, x, rv,     \
                                cin, wt, bias, cout, relu, y, out.poff);                                                \
    } while (0)
#define OCRS_DISPATCH_POOL(BN_, TW_)                              \
    do {                                                          \
        if (ph == 1) OCRS_LAUNCH_CONV(BN_, TW_, 1, 1);            \
        else if (pw == 1) OCRS_LAUNCH_CONV(BN_, TW_, 2, 1);       \
        else OCRS_LAUNCH_CONV(BN_, TW_, 2, 2);                    \
    } while (0)
    if (n64) {
        if (rv.tw == 32) OCRS_DISPATCH_POOL(64, 32); else OCRS_DISPATCH_POOL(64, 16);
    } else {
        if (rv.tw == 32) OCRS_DISPATCH_POOL(128, 32); else OCRS_DISPATCH_POOL(128, 16);
    }
#undef OCRS_DISPATCH_POOL
#undef OCRS_LAUNCH_CONV
#undef OCRS_LAUNCH_SPLIT
    return true;
}

}  // namespace k
}  // namespace ocrs
