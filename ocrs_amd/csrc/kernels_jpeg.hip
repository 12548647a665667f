// GPU half of the JPEG hand-off (jpeg.hpp; the reference decodes on the host,
// ocrs-cli/src/main.rs:312-333): dequantisation + 8x8 inverse DCT + chroma upsampling + YCbCr -> RGB, integer
// arithmetic that restates libjpeg's reference implementation operation for operation so that the pixels are the ones
// libjpeg / libjpeg-turbo / PIL produce (jidctint.c jpeg_idct_islow, jdsample.c fancy upsampling, jdcolor.c).
//
// Both kernels are byte/integer streaming work: ~0.7 B/px of sparse coefficients in, 1-3 B/px of samples out and in
// again, 3 B/px of RGB out — HBM/L2-bound, no matrix-core shape anywhere (a DCT as a GEMM would move the same bytes).
#include "jpeg.hpp"
#include "kernels.hpp"

namespace ocrs {
namespace k {

namespace {

// jidctint.c: CONST_BITS = 13, PASS1_BITS = 2, FIX(x) = round(x * 2^13)
constexpr int kConstBits = 13, kPass1Bits = 2;
constexpr int64_t F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
                  F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069,
                  F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;

__device__ __forceinline__ int64_t descale(int64_t x, int n) { return (x + (int64_t(1) << (n - 1))) >> n; }

// The 1-D 8-point inverse DCT of jpeg_idct_islow (Loeffler-Ligtenberg-Moschytz, 12 multiplies).  in[0..7] -> out[0..7],
// result descaled by `shift`.  JLONG is 64 bits wide in the reference implementation on LP64; so is the arithmetic here.
__device__ __forceinline__ void idct8(const int64_t (&in)[8], int64_t (&out)[8], int shift) {
    // even part
    int64_t z2 = in[2], z3 = in[6];
    int64_t z1 = (z2 + z3) * F_0_541196100;
    int64_t tmp2 = z1 + z3 * (-F_1_847759065);
    int64_t tmp3 = z1 + z2 * F_0_765366865;
    z2 = in[0]; z3 = in[4];
    int64_t tmp0 = (z2 + z3) * (int64_t(1) << kConstBits);
    int64_t tmp1 = (z2 - z3) * (int64_t(1) << kConstBits);
    const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    // odd part
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int64_t z4 = tmp1 + tmp3;
    const int64_t z5 = (z3 + z4) * F_1_175875602;
    tmp0 = tmp0 * F_0_298631336; tmp1 = tmp1 * F_2_053119869; tmp2 = tmp2 * F_3_072711026; tmp3 = tmp3 * F_1_501321110;
    z1 = z1 * (-F_0_899976223); z2 = z2 * (-F_2_562915447); z3 = z3 * (-F_1_961570560); z4 = z4 * (-F_0_390180644);
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    out[0] = descale(tmp10 + tmp3, shift); out[7] = descale(tmp10 - tmp3, shift);
    out[1] = descale(tmp11 + tmp2, shift); out[6] = descale(tmp11 - tmp2, shift);
    out[2] = descale(tmp12 + tmp1, shift); out[5] = descale(tmp12 - tmp1, shift);
    out[3] = descale(tmp13 + tmp0, shift); out[4] = descale(tmp13 - tmp0, shift);
}

// The post-IDCT range-limit table of jdmaster.c (prepare_range_limit_table), as arithmetic: index = (x & 0x3FF) over a
// table whose first 128 entries are 128..255, then 384 x 255, then 384 x 0, then 0..127 — i.e. clamp(x + 128, 0, 255) for every
// value a valid stream produces, and the reference's wrap-around for the rest.
__device__ __forceinline__ uint8_t range_limit_idct(int64_t x) {
    const int i = (int)(x & 0x3FF);
    return (uint8_t)(i < 128 ? i + 128 : i < 512 ? 255 : i < 896 ? 0 : i - 896);
}

// natural position (row * 8 + col) -> zig-zag index (the sparse coefficient stream is ordered by zig-zag index: the order
// the entropy decoder produces the values in)
__constant__ uint8_t kZigOfNat[64] = {
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

struct JpegComp {
    int blocks_w, blocks_h;       // block grid
    int width, height;            // samples that matter
    int pitch;                    // bytes per sample row of the component's plane (= blocks_w * 8)
    uint32_t first_block;
    int64_t plane_off;            // byte offset of the plane in the sample buffer
    int tq;
};
struct JpegArgs {
    JpegComp comp[3];
    int ncomp, width, height, hmax, vmax, ycc;
    uint32_t nblocks;
};

// One thread = one column (pass 1) and then one row (pass 2) of an 8x8 block; 32 blocks per 256-thread workgroup, the
// workspace between the passes in LDS.
__global__ void __launch_bounds__(256)
jpeg_idct_kernel(JpegArgs a, const uint64_t* __restrict__ mask, const uint32_t* __restrict__ offset,
                 const int16_t* __restrict__ values, const uint16_t* __restrict__ quant, uint8_t* __restrict__ samples) {
    __shared__ int32_t ws[32][64];
    const int lb = threadIdx.x >> 3, c = threadIdx.x & 7;
    const uint32_t blk = blockIdx.x * 32u + (uint32_t)lb;
    const bool live = blk < a.nblocks;
    int ci = 0;
    if (live) {
        if (a.ncomp > 1 && blk >= a.comp[1].first_block) ci = 1;
        if (a.ncomp > 2 && blk >= a.comp[2].first_block) ci = 2;
    }
    const JpegComp& k = a.comp[ci];
    if (live) {
        const uint64_t m = mask[blk];
        const uint32_t off = offset[blk];
        const uint16_t* q = quant + k.tq * 64;
        int64_t in[8], out[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int p = r * 8 + c;
            const int z = kZigOfNat[p];
            int v = 0;
            if ((m >> z) & 1) v = values[off + __popcll(m & ((uint64_t(1) << z) - 1))];
            in[r] = (int64_t)v * (int64_t)q[p];   // DEQUANTIZE
        }
        idct8(in, out, kConstBits - kPass1Bits);
#pragma unroll
        for (int r = 0; r < 8; r++) ws[lb][r * 8 + c] = (int32_t)out[r];
    }
    __syncthreads();
    if (live) {
        int64_t in[8], out[8];
#pragma unroll
        for (int x = 0; x < 8; x++) in[x] = ws[lb][c * 8 + x];
        idct8(in, out, kConstBits + kPass1Bits + 3);
        const uint32_t local = blk - k.first_block;
        const int by = (int)(local / (uint32_t)k.blocks_w), bx = (int)(local % (uint32_t)k.blocks_w);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            lo |= (uint32_t)range_limit_idct(out[x]) << (8 * x);
            hi |= (uint32_t)range_limit_idct(out[4 + x]) << (8 * x);
        }
        uint2* dst = reinterpret_cast<uint2*>(samples + k.plane_off + (int64_t)(by * 8 + c) * k.pitch + bx * 8);
        *dst = make_uint2(lo, hi);
    }
}

// jdsample.c / jdcolor.c: chroma upsampling and colour conversion, one thread per output pixel.
__device__ __forceinline__ int chroma_h2v1(const uint8_t* row, int cw, int x) {
    const int i = x >> 1;
    if (cw <= 2) return row[i];   // do_fancy needs downsampled_width > 2; otherwise plain replication
    const int v = row[i] * 3;
    if (x & 1) return i == cw - 1 ? row[i] : (v + row[i + 1] + 2) >> 2;
    return i == 0 ? row[i] : (v + row[i - 1] + 1) >> 2;
}
__device__ __forceinline__ int chroma_h2v2(const uint8_t* plane, int pitch, int cw, int ch, int x, int y) {
    const int i = x >> 1, j = y >> 1;
    if (cw <= 2) return plane[(int64_t)j * pitch + i];
    // the nearer and the farther input row of this output row (above for even y, below for odd y; edges replicated)
    const int jf = (y & 1) ? (j + 1 < ch ? j + 1 : j) : (j > 0 ? j - 1 : j);
    const uint8_t* n = plane + (int64_t)j * pitch;
    const uint8_t* f = plane + (int64_t)jf * pitch;
    const int cur = n[i] * 3 + f[i];
    if (x & 1) return i == cw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + (n[i + 1] * 3 + f[i + 1]) + 7) >> 4;
    return i == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + (n[i - 1] * 3 + f[i - 1]) + 8) >> 4;
}
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

__global__ void __launch_bounds__(256)
jpeg_color_kernel(JpegArgs a, const uint8_t* __restrict__ samples, uint8_t* __restrict__ rgb) {
    const int64_t total = (int64_t)a.width * a.height;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(p / a.width), x = (int)(p % a.width);
        const int Y = samples[a.comp[0].plane_off + (int64_t)y * a.comp[0].pitch + x];
        int r = Y, g = Y, b = Y;
        if (a.ncomp == 3) {
            int c1, c2;
            const uint8_t* p1 = samples + a.comp[1].plane_off;
            const uint8_t* p2 = samples + a.comp[2].plane_off;
            if (a.hmax == 1) {
                c1 = p1[(int64_t)y * a.comp[1].pitch + x];
                c2 = p2[(int64_t)y * a.comp[2].pitch + x];
            } else if (a.vmax == 1) {
                c1 = chroma_h2v1(p1 + (int64_t)y * a.comp[1].pitch, a.comp[1].width, x);
                c2 = chroma_h2v1(p2 + (int64_t)y * a.comp[2].pitch, a.comp[2].width, x);
            } else {
                c1 = chroma_h2v2(p1, a.comp[1].pitch, a.comp[1].width, a.comp[1].height, x, y);
                c2 = chroma_h2v2(p2, a.comp[2].pitch, a.comp[2].width, a.comp[2].height, x, y);
            }
            if (a.ycc) {
                // jdcolor.c build_ycc_rgb_table: SCALEBITS 16, ONE_HALF, FIX(1.40200) = 91881, FIX(1.77200) = 116130,
                // FIX(0.71414) = 46802, FIX(0.34414) = 22554; arithmetic right shifts
                const int cb = c1 - 128, cr = c2 - 128;
                r = clamp255(Y + ((91881 * cr + 32768) >> 16));
                g = clamp255(Y + ((-22554 * cb + 32768 + -46802 * cr) >> 16));
                b = clamp255(Y + ((116130 * cb + 32768) >> 16));
            } else {
                r = Y; g = c1; b = c2;
            }
        }
        uint8_t* o = rgb + p * 3;
        o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b;
    }
}

}  // namespace

size_t jpeg_sample_bytes(const jpeg::Coefficients& c) {
    size_t n = 0;
    for (int i = 0; i < c.ncomp; i++) n += (size_t)c.comp[i].blocks_w * 8 * c.comp[i].blocks_h * 8;
    return n;
}

// d_mask / d_offset / d_values / d_quant: the coefficient arrays of `c` on the device; d_samples: jpeg_sample_bytes(c)
// bytes of scratch; d_rgb: width * height * 3 bytes, RGB u8 HWC (grey images: R = G = B, as image::into_rgb8 gives).
void jpeg_decode(const jpeg::Coefficients& c, const uint64_t* d_mask, const uint32_t* d_offset, const int16_t* d_values,
                 const uint16_t* d_quant, uint8_t* d_samples, uint8_t* d_rgb, hipStream_t s) {
    JpegArgs a{};
    a.ncomp = c.ncomp; a.width = c.width; a.height = c.height; a.hmax = c.hmax; a.vmax = c.vmax; a.ycc = c.ycc ? 1 : 0;
    a.nblocks = (uint32_t)c.nblocks();
    int64_t off = 0;
    for (int i = 0; i < c.ncomp; i++) {
        JpegComp& k = a.comp[i];
        k.blocks_w = c.comp[i].blocks_w; k.blocks_h = c.comp[i].blocks_h;
        k.width = c.comp[i].width; k.height = c.comp[i].height;
        k.pitch = k.blocks_w * 8;
        k.first_block = (uint32_t)c.comp[i].first_block;
        k.plane_off = off;
        k.tq = c.comp[i].tq;
        off += (int64_t)k.pitch * k.blocks_h * 8;
    }
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((a.nblocks + 31) / 32), dim3(256), 0, s, a, d_mask, d_offset, d_values, d_quant, d_samples);
    const int64_t px = (int64_t)c.width * c.height;
    const int grid = (int)std::min<int64_t>((px + 255) / 256, 16384);
    hipLaunchKernelGGL(jpeg_color_kernel, dim3(grid), dim3(256), 0, s, a, d_samples, d_rgb);
}

}  // namespace k
}  // namespace ocrs
