// Image-space kernels of the detection path: greyscale conversion, the two
// bilinear resizes around the CNN, and the text-map threshold.  All are
// HBM-bound streaming kernels: one pass, coalesced, 16 B/lane
// where the format allows.  Built with -ffp-contract=off: the reference's
// arithmetic has no fused multiply-adds here (preprocess.rs:229-233).
#include "common.hpp"
#include "kernels.hpp"

namespace ocrs {
namespace k {

// ---------------------------------------------------------------------------
// prepare_image — preprocess.rs:201-248.
//   out = -0.5 + sum_c px[c] * w[c]   (c ascending, one rounding per op)
// Algorithmic bytes: chans (u8) or 4*chans (f32) in, 4 out per pixel.
// ---------------------------------------------------------------------------
template <bool IS_U8, bool CHANS_LAST, int CHANS>
__global__ void __launch_bounds__(256) prepare_image_kernel(const void* __restrict__ src, float* __restrict__ out,
                                                            int64_t plane, float w0, float w1, float w2) {
    constexpr int NW = CHANS == 1 ? 1 : 3;
    const float wts[3] = {w0, w1, w2};
    const uint8_t* s8 = static_cast<const uint8_t*>(src);
    const float* sf = static_cast<const float*>(src);
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < plane;
         p += (int64_t)gridDim.x * blockDim.x) {
        float px = -0.5f;
#pragma unroll
        for (int c = 0; c < NW; c++) {
            int64_t idx = CHANS_LAST ? p * CHANS + c : (int64_t)c * plane + p;
            float v = IS_U8 ? (float)s8[idx] : sf[idx];
            px = px + v * wts[c];
        }
        out[p] = px;
    }
}

// Fast path: u8 RGB HWC, 4 pixels (12 B in, 16 B out) per thread.
__global__ void __launch_bounds__(256) prepare_image_rgb8_x4_kernel(const uint32_t* __restrict__ src,
                                                                    float4* __restrict__ out, int64_t quads,
                                                                    float w0, float w1, float w2) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads;
         q += (int64_t)gridDim.x * blockDim.x) {
        uint32_t a = src[3 * q], b = src[3 * q + 1], c = src[3 * q + 2];
        // bytes: a = R0 G0 B0 R1 | b = G1 B1 R2 G2 | c = B2 R3 G3 B3
        float r0 = (float)(a & 0xff), g0 = (float)((a >> 8) & 0xff), b0 = (float)((a >> 16) & 0xff);
        float r1 = (float)(a >> 24), g1 = (float)(b & 0xff), b1 = (float)((b >> 8) & 0xff);
        float r2 = (float)((b >> 16) & 0xff), g2 = (float)(b >> 24), b2 = (float)(c & 0xff);
        float r3 = (float)((c >> 8) & 0xff), g3 = (float)((c >> 16) & 0xff), b3 = (float)(c >> 24);
        float4 o;
        o.x = ((-0.5f + r0 * w0) + g0 * w1) + b0 * w2;
        o.y = ((-0.5f + r1 * w0) + g1 * w1) + b1 * w2;
        o.z = ((-0.5f + r2 * w0) + g2 * w1) + b2 * w2;
        o.w = ((-0.5f + r3 * w0) + g3 * w1) + b3 * w2;
        out[q] = o;
    }
}

void prepare_image(const void* d_pixels, bool is_u8, bool chans_last, int h, int w, int chans, float* d_out,
                   hipStream_t s) {
    const float itu[3] = {0.299f, 0.587f, 0.114f};
    float wt[3];
    for (int c = 0; c < 3; c++) {
        if (chans == 1) wt[c] = is_u8 ? (1.0f / 255.0f) : 1.0f;
        else wt[c] = is_u8 ? (itu[c] / 255.0f) : itu[c];
    }
    const int64_t plane = (int64_t)h * w;
    if (is_u8 && chans_last && chans == 3 && plane % 4 == 0 && ((uintptr_t)d_pixels % 4) == 0 &&
        ((uintptr_t)d_out % 16) == 0) {
        int64_t quads = plane / 4;
        int grid = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
        hipLaunchKernelGGL(prepare_image_rgb8_x4_kernel, dim3(grid), dim3(256), 0, s, (const uint32_t*)d_pixels,
                           (float4*)d_out, quads, wt[0], wt[1], wt[2]);
        return;
    }
    int grid = (int)((plane + 255) / 256 < 8192 ? (plane + 255) / 256 : 8192);
    if (grid < 1) grid = 1;
#define LAUNCH(U8, CL, CH)                                                                                  \
    hipLaunchKernelGGL((prepare_image_kernel<U8, CL, CH>), dim3(grid), dim3(256), 0, s, d_pixels, d_out, plane, \
                       wt[0], wt[1], wt[2])
#define DISPATCH_CH(U8, CL)                  \
    switch (chans) {                         \
        case 1: LAUNCH(U8, CL, 1); break;    \
        case 3: LAUNCH(U8, CL, 3); break;    \
        default: LAUNCH(U8, CL, 4); break;   \
    }
    if (is_u8) {
        if (chans_last) { DISPATCH_CH(true, true) } else { DISPATCH_CH(true, false) }
    } else {
        if (chans_last) { DISPATCH_CH(false, true) } else { DISPATCH_CH(false, false) }
    }
#undef DISPATCH_CH
#undef LAUNCH
}

// ---------------------------------------------------------------------------
// Bilinear resize, ONNX Resize linear / half_pixel (rten resize_image;
// detection.rs:168,194, recognition.rs:121).
//   c   = clamp((o + 0.5) * (in/out) - 0.5, 0, in-1);  i0 = (int)c; i1 = min(i0+1, in-1)
//   out = (1-wy) * ((1-wx)*tl + wx*tr) + wy * ((1-wx)*bl + wx*br)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void resize_axis(int o, int in_len, int out_len, int& i0, int& i1, float& wgt) {
    float scale = (float)in_len / (float)out_len;
    float c = ((float)o + 0.5f) * scale - 0.5f;
    float hi = (float)(in_len - 1);
    c = c < 0.0f ? 0.0f : c;
    c = c > hi ? hi : c;
    int a = (int)c;
    i0 = a;
    i1 = a + 1 < in_len ? a + 1 : in_len - 1;
    wgt = c - (float)a;
}

__device__ __forceinline__ float bilerp(float tl, float tr, float bl, float br, float wx, float wy) {
    float top = (1.0f - wx) * tl + wx * tr;
    float bot = (1.0f - wx) * bl + wx * br;
    return (1.0f - wy) * top + wy * bot;
}

// Virtual padded source [vh,vw]: (y<sh && x<sw) ? page : -0.5 (detection.rs:155-164).
// Algorithmic bytes per page: 4*sh*sw read (each source pixel once) + 4*dh*dw written.
__global__ void __launch_bounds__(256)
resize_pages_kernel(const float* const* __restrict__ src_ptrs, int sh, int sw, int vh, int vw,
                    float* __restrict__ dst, int dh, int dw) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int n = blockIdx.z;
    if (x >= dw) return;
    const float* __restrict__ src = src_ptrs[n];
    int y0, y1, x0, x1;
    float wy, wx;
    resize_axis(y, vh, dh, y0, y1, wy);
    resize_axis(x, vw, dw, x0, x1, wx);
    const float fill = -0.5f;
    float tl = (y0 < sh && x0 < sw) ? src[(int64_t)y0 * sw + x0] : fill;
    float tr = (y0 < sh && x1 < sw) ? src[(int64_t)y0 * sw + x1] : fill;
    float bl = (y1 < sh && x0 < sw) ? src[(int64_t)y1 * sw + x0] : fill;
    float br = (y1 < sh && x1 < sw) ? src[(int64_t)y1 * sw + x1] : fill;
    dst[((int64_t)n * dh + y) * dw + x] = bilerp(tl, tr, bl, br, wx, wy);
}

void resize_pages_to_model(const float* const* d_src_ptrs, int n, int sh, int sw, int vh, int vw, float* d_dst,
                           int dh, int dw, hipStream_t s) {
    dim3 grid((dw + 255) / 256, dh, n);
    hipLaunchKernelGGL(resize_pages_kernel, grid, dim3(256), 0, s, d_src_ptrs, sh, sw, vh, vw, d_dst, dh, dw);
}

// prob [n, mh, mw] sliced to [sh, sw], resized to [h, w], thresholded with a
// strict '>' (detection.rs:110,187-194).  4 output pixels per thread so the u8
// mask is written as one 32-bit word.
// Algorithmic bytes per page: 4*sh*sw read + h*w (mask) [+ 4*h*w map] written.
// LABELS (r6): the kernel also writes the INITIAL LABELS of the component stage (kernels_ccl.hip ccl_init4_kernel: the start of
// the horizontal run of equal mask pixels inside the wave's 256-pixel segment) — the thread holds its four mask pixels in a
// register anyway — and zeroes the page's two counters of that stage (overflow flag, contour-arena bump pointer): one launch
// and two fills fewer per request, the mask is not read back for the labels.  Needs w % 4 == 0; same bits as the two kernels.
template <bool LABELS>
__global__ void __launch_bounds__(256)
resize_threshold_kernel(const float* __restrict__ prob, int mh, int mw, int sh, int sw, float thr,
                        uint8_t* __restrict__ mask, float* __restrict__ map, int h, int w, int32_t* __restrict__ labels,
                        int32_t* __restrict__ zero_a, int32_t* __restrict__ zero_b) {
    const int xq = blockIdx.x * blockDim.x + threadIdx.x;  // group of 4 pixels
    const int y = blockIdx.y;
    const int n = blockIdx.z;
    const int x_base = xq * 4;
    const bool inb = x_base < w;
    if (!LABELS && !inb) return;
    if (LABELS && xq == 0 && y == 0) { zero_a[n] = 0; zero_b[n] = 0; }
    const float* __restrict__ src = prob + (int64_t)n * mh * mw;
    int y0, y1;
    float wy;
    resize_axis(y, sh, h, y0, y1, wy);
    uint32_t bits = 0;
    float vals[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int x = x_base + i;
        float v = 0.0f;
        if (x < w) {
            int x0, x1;
            float wx;
            resize_axis(x, sw, w, x0, x1, wx);
            float tl = src[(int64_t)y0 * mw + x0], tr = src[(int64_t)y0 * mw + x1];
            float bl = src[(int64_t)y1 * mw + x0], br = src[(int64_t)y1 * mw + x1];
            v = bilerp(tl, tr, bl, br, wx, wy);
            if (v > thr) bits |= 1u << (8 * i);
        }
        vals[i] = v;
    }
    const int64_t o = ((int64_t)n * h + y) * w + x_base;
    if (LABELS) {   // w % 4 == 0: a thread is inside or outside the row as a whole; lanes of a wave = consecutive quads of one row
        const int lane = threadIdx.x & 63;
        const uint32_t cur = inb ? bits : 0x02020202u;
        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur, 0x138, 0xF, 0xF, true);   // lane - 1's (0 at lane 0)
        const int v0 = cur & 0xFF, v1 = (cur >> 8) & 0xFF, v2 = (cur >> 16) & 0xFF, v3 = cur >> 24;
        const bool b0 = lane == 0 || v0 != (int)(prev >> 24), b1 = v1 != v0, b2 = v2 != v1, b3 = v3 != v2;
        const int top = b3 ? 3 : b2 ? 2 : b1 ? 1 : b0 ? 0 : -1;    // the lane's last boundary
        const unsigned long long any = __ballot(top >= 0);
        const unsigned long long below = any & ((1ull << lane) - 1ull);
        const int srcl = below ? 63 - __clzll(below) : 0;          // the nearest lower lane that holds a boundary
        const int src_top = __shfl(top, srcl);
        if (!inb) return;
        const int seg = y * w + x_base - 4 * lane;                 // page-local index of the wave's first pixel
        const int carried = seg + 4 * srcl + src_top;
        int4 out;
        const int s0 = b0 ? seg + 4 * lane : carried;
        const int s1 = b1 ? seg + 4 * lane + 1 : s0;
        const int s2 = b2 ? seg + 4 * lane + 2 : s1;
        const int s3 = b3 ? seg + 4 * lane + 3 : s2;
        out.x = s0; out.y = s1; out.z = s2; out.w = s3;
        *reinterpret_cast<int4*>(labels + o) = out;
        *reinterpret_cast<uint32_t*>(mask + o) = bits;
        if (map) *reinterpret_cast<float4*>(map + o) = make_float4(vals[0], vals[1], vals[2], vals[3]);
        return;
    }
    if (x_base + 3 < w && (w & 3) == 0) {
        *reinterpret_cast<uint32_t*>(mask + o) = bits;
        if (map) *reinterpret_cast<float4*>(map + o) = make_float4(vals[0], vals[1], vals[2], vals[3]);
    } else {
        for (int i = 0; i < 4 && x_base + i < w; i++) {
            mask[o + i] = (bits >> (8 * i)) & 1;
            if (map) map[o + i] = vals[i];
        }
    }
}

bool resize_threshold(const float* d_prob, int n, int mh, int mw, int sh, int sw, float thr, uint8_t* d_mask,
                      float* d_map, int h, int w, hipStream_t s, int32_t* d_labels, int32_t* d_zero_a, int32_t* d_zero_b) {
    int quads = (w + 3) / 4;
    dim3 grid((quads + 255) / 256, h, n);
    dim3 block(256);
    if (quads <= 64) block = dim3(64);
    else if (quads <= 128) block = dim3(128);
    grid.x = (quads + block.x - 1) / block.x;
    // the fused form wants what ccl_init4_kernel wants (kernels_ccl.hip ccl_label): word-aligned rows, 16-byte-aligned labels
    const bool fuse = d_labels && d_zero_a && d_zero_b && (w & 3) == 0 && (((uintptr_t)d_mask) & 3) == 0 && (((uintptr_t)d_labels) & 15) == 0 &&
                      (!d_map || (((uintptr_t)d_map) & 15) == 0) && option(OPT_CCL_QUAD) != 0;
    if (fuse)
        hipLaunchKernelGGL((resize_threshold_kernel<true>), grid, block, 0, s, d_prob, mh, mw, sh, sw, thr, d_mask, d_map, h, w, d_labels,
                           d_zero_a, d_zero_b);
    else
        hipLaunchKernelGGL((resize_threshold_kernel<false>), grid, block, 0, s, d_prob, mh, mw, sh, sw, thr, d_mask, d_map, h, w, nullptr,
                           nullptr, nullptr);
    return fuse;
}

__global__ void __launch_bounds__(256)
threshold_kernel(const float* __restrict__ p, float thr, uint8_t* __restrict__ m, int64_t count) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        m[i] = p[i] > thr ? 1 : 0;
}

void threshold_only(const float* d_prob, float thr, uint8_t* d_mask, int64_t count, hipStream_t s) {
    int grid = (int)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(threshold_kernel, dim3(grid), dim3(256), 0, s, d_prob, thr, d_mask, count);
}

}  // namespace k
}  // namespace ocrs
