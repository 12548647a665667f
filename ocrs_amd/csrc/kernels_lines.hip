// Text-line crops for recognition (recognition.rs:91-158): polygon scan-fill
// gather from the grey page, bilinear resize to [64, resized_w], right-pad with
// -0.5 into the batch tensor — fused into one pass that never materialises the
// intermediate line image.
//
// One block = one output row of one line.  The block first intersects the
// polygon with the (at most two) source scanlines that row interpolates
// between, into LDS; every output pixel then classifies its 4 source taps by
// counting crossings (even-odd rule), gathers them from the page and
// interpolates.  The LDS lists hold CROSSINGS, not edges: a line polygon of
// any number of words (4 vertices each, recognition.rs:29-55) crosses a
// scanline a handful of times; a scanline with more than MAX_LDS_EDGES
// crossings (a degenerate polygon) is handled by walking the edge list per tap.  HBM-bound: reads the line's page pixels once (L2 serves the
// 2x row re-use), writes 4*out_w bytes per row.
// (r6 tried four output rows per block — the column mapping shared, sixteen gathers in flight, a quarter of the blocks: 0.69 ms
// instead of 0.54 per 16-page request with the GPU to itself (67 VGPRs, 106 SGPRs: fewer blocks per CU); this form stays.)
#include "kernels.hpp"

namespace ocrs {
namespace k {

constexpr int MAX_LDS_EDGES = 512;

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

// x of the downward edge (ya<yb) at scanline y: xa + round_half_away((y-ya) * dx/dy)
__device__ __forceinline__ int edge_x_at(int xa, int ya, int xb, int yb, int y) {
    float t = (float)(y - ya) * ((float)(xb - xa) / (float)(yb - ya));
    return xa + (int)roundf(t);
}

__global__ void __launch_bounds__(256)
crop_lines_kernel(const float* const* __restrict__ pages, const int32_t* __restrict__ page_hw,
                  const LineDesc* __restrict__ lines, const int32_t* __restrict__ poly, int out_h,
                  float* __restrict__ batch) {
    __shared__ int xs[2][MAX_LDS_EDGES];
    __shared__ int cnt[2];
#ifdef OCRS_CROP_SETPRIO   // probe builds only (tools/build_hazard_repro.sh): the victim of the co-residency hazard at a raised wave priority
    __builtin_amdgcn_s_setprio(OCRS_CROP_SETPRIO);
#endif
    const LineDesc ln = lines[blockIdx.y];
    const int oy = blockIdx.x;
    const int out_w = ln.out_w;
    float* __restrict__ dst = batch + ln.out_off + (int64_t)oy * out_w;
    const float fill = -0.5f;
    if (ln.bh <= 0 || ln.bw <= 0) {
        for (int ox = threadIdx.x; ox < out_w; ox += blockDim.x) dst[ox] = fill;
        return;
    }
    const float* __restrict__ page = pages[ln.page];
    const int ph = page_hw[2 * ln.page], pw = page_hw[2 * ln.page + 1];
    int r0, r1;
    float wy;
    resize_axis(oy, ln.bh, out_h, r0, r1, wy);
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int32_t* pv = poly + 2 * (int64_t)ln.poly_off;
    for (int i = threadIdx.x; i < 2 * ln.poly_n; i += blockDim.x) {
        const int which = i / ln.poly_n, e = i - which * ln.poly_n;
        const int y = ln.top + (which ? r1 : r0);
        int ya = pv[2 * e], xa = pv[2 * e + 1];
        const int e2 = e + 1 == ln.poly_n ? 0 : e + 1;
        int yb = pv[2 * e2], xb = pv[2 * e2 + 1];
        if (ya == yb) continue;
        if (ya > yb) { int t = ya; ya = yb; yb = t; t = xa; xa = xb; xb = t; }
        if (y < ya || y >= yb) continue;
        int slot = atomicAdd(&cnt[which], 1);
        if (slot < MAX_LDS_EDGES) xs[which][slot] = edge_x_at(xa, ya, xb, yb, y);
    }
    __syncthreads();
    const int m0 = cnt[0], m1 = cnt[1];
    // crossings of scanline `which` at or left of x, straight from the edge list (same arithmetic as the staging loop)
    auto crossings_direct = [&](int which, int x) {
        const int y = ln.top + (which ? r1 : r0);
        int c = 0;
        for (int e = 0; e < ln.poly_n; e++) {
            int ya = pv[2 * e], xa = pv[2 * e + 1];
            const int e2 = e + 1 == ln.poly_n ? 0 : e + 1;
            int yb = pv[2 * e2], xb = pv[2 * e2 + 1];
            if (ya == yb) continue;
            if (ya > yb) { int t = ya; ya = yb; yb = t; t = xa; xa = xb; xb = t; }
            if (y < ya || y >= yb) continue;
            c += edge_x_at(xa, ya, xb, yb, y) <= x ? 1 : 0;
        }
        return c;
    };
    for (int ox = threadIdx.x; ox < out_w; ox += blockDim.x) {
        float v = fill;
        if (ox < ln.resized_w) {
            int c0, c1;
            float wx;
            resize_axis(ox, ln.bw, ln.resized_w, c0, c1, wx);
            float tap[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int which = q >> 1;
                const int r = which ? r1 : r0;
                const int c = (q & 1) ? c1 : c0;
                const int y = ln.top + r, x = ln.left + c;
                const int m = which ? m1 : m0;
                int crossings = 0;
                if (m <= MAX_LDS_EDGES)
                    for (int kk = 0; kk < m; kk++) crossings += xs[which][kk] <= x ? 1 : 0;
                else
                    crossings = crossings_direct(which, x);
                float t = fill;
                // page_index_rect.contains_point(in_p) && contains_point(out_p) (recognition.rs:100,112)
                if ((crossings & 1) && y >= 0 && y <= ph - 1 && x >= 0 && x <= pw - 1 && r <= ph - 1 && c <= pw - 1)
                    t = page[(int64_t)y * pw + x];
                tap[q] = t;
            }
            float top = (1.0f - wx) * tap[0] + wx * tap[1];
            float bot = (1.0f - wx) * tap[2] + wx * tap[3];
            v = (1.0f - wy) * top + wy * bot;
        }
        dst[ox] = v;
    }
}

void crop_lines(const float* const* d_pages, const int32_t* d_page_hw, const LineDesc* d_lines, const int32_t* d_poly,
                int n_lines, int out_h, float* d_out, hipStream_t s) {
    if (n_lines <= 0) return;
    hipLaunchKernelGGL(crop_lines_kernel, dim3(out_h, n_lines), dim3(256), 0, s, d_pages, d_page_hw, d_lines, d_poly,
                       out_h, d_out);
}

}  // namespace k
}  // namespace ocrs
