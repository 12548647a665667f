// Text-line crops for recognition (recognition.rs:91-158): polygon scan-fill
// gather from the grey page, bilinear resize to [64, resized_w], right-pad with
// -0.5 into the batch tensor — fused into one pass that never materialises the
// intermediate line image.
//
// One block = four consecutive output rows of one line.  The block first intersects the
// polygon with the (at most two per row) source scanlines those rows interpolate
// between, into LDS; every output pixel then classifies its 4 source taps by
// counting crossings (even-odd rule), gathers them from the page and
// interpolates.  The LDS lists hold CROSSINGS, not edges: a line polygon of
// any number of words (4 vertices each, recognition.rs:29-55) crosses a
// scanline a handful of times; a scanline with more than CROP_EDGES
// crossings (a degenerate polygon) is handled by walking the edge list per tap.  HBM-bound: reads the line's page pixels once (L2 serves the
// 2x row re-use), writes 4*out_w bytes per row.
#include "kernels.hpp"

namespace ocrs {
namespace k {


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

// r6: a block covers CROP_ROWS = 4 consecutive output rows of a line (the x-axis mapping of a column — c0, c1, wx — is worked
// out once per thread and serves the four rows; its sixteen page gathers are in flight together; a quarter of the blocks):
// the kernel is bound by its dependent chain (crossing lists -> gather addresses -> L2 round trip -> store), not by bytes.
constexpr int CROP_ROWS = 4;
constexpr int CROP_EDGES = 256;   // crossings per scanline kept in LDS (a line polygon crosses a scanline a handful of times)

__global__ void __launch_bounds__(256)
crop_lines_kernel(const float* const* __restrict__ pages, const int32_t* __restrict__ page_hw,
                  const LineDesc* __restrict__ lines, const int32_t* __restrict__ poly, int out_h,
                  float* __restrict__ batch) {
    __shared__ int xs[2 * CROP_ROWS][CROP_EDGES];
    __shared__ int cnt[2 * CROP_ROWS];
#ifdef OCRS_CROP_SETPRIO   // probe builds only (tools/build_hazard_repro.sh): the victim of the co-residency hazard at a raised wave priority
    __builtin_amdgcn_s_setprio(OCRS_CROP_SETPRIO);
#endif
    const LineDesc ln = lines[blockIdx.y];
    const int oy0 = blockIdx.x * CROP_ROWS;
    const int nrows = min(CROP_ROWS, out_h - oy0);
    const int out_w = ln.out_w;
    float* __restrict__ dst = batch + ln.out_off + (int64_t)oy0 * out_w;
    const float fill = -0.5f;
    if (ln.bh <= 0 || ln.bw <= 0) {
        for (int i = threadIdx.x; i < nrows * out_w; i += blockDim.x) dst[i] = fill;
        return;
    }
    const float* __restrict__ page = pages[ln.page];
    const int ph = page_hw[2 * ln.page], pw = page_hw[2 * ln.page + 1];
    int rr[2 * CROP_ROWS];      // source rows: [2 j] = r0, [2 j + 1] = r1 of output row oy0 + j
    float wy[CROP_ROWS];
#pragma unroll
    for (int j = 0; j < CROP_ROWS; j++) resize_axis(min(oy0 + j, out_h - 1), ln.bh, out_h, rr[2 * j], rr[2 * j + 1], wy[j]);
    if (threadIdx.x < 2 * CROP_ROWS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int32_t* pv = poly + 2 * (int64_t)ln.poly_off;
    for (int i = threadIdx.x; i < 2 * CROP_ROWS * ln.poly_n; i += blockDim.x) {
        const int which = i / ln.poly_n, e = i - which * ln.poly_n;
        int r = rr[0];
#pragma unroll
        for (int q = 1; q < 2 * CROP_ROWS; q++) r = which == q ? rr[q] : r;
        const int y = ln.top + r;
        int ya = pv[2 * e], xa = pv[2 * e + 1];
        const int e2 = e + 1 == ln.poly_n ? 0 : e + 1;
        int yb = pv[2 * e2], xb = pv[2 * e2 + 1];
        if (ya == yb) continue;
        if (ya > yb) { int t = ya; ya = yb; yb = t; t = xa; xa = xb; xb = t; }
        if (y < ya || y >= yb) continue;
        int slot = atomicAdd(&cnt[which], 1);
        if (slot < CROP_EDGES) xs[which][slot] = edge_x_at(xa, ya, xb, yb, y);
    }
    __syncthreads();
    // crossings of source row r at or left of x, straight from the edge list (same arithmetic as the staging loop)
    auto crossings_direct = [&](int r, int x) {
        const int y = ln.top + r;
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
        if (ox >= ln.resized_w) {
#pragma unroll
            for (int j = 0; j < CROP_ROWS; j++)
                if (j < nrows) dst[(int64_t)j * out_w + ox] = fill;
            continue;
        }
        int c0, c1;
        float wx;
        resize_axis(ox, ln.bw, ln.resized_w, c0, c1, wx);
        float tap[CROP_ROWS][4];
#pragma unroll
        for (int j = 0; j < CROP_ROWS; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int which = 2 * j + (q >> 1);
                const int r = rr[which];
                const int c = (q & 1) ? c1 : c0;
                const int y = ln.top + r, x = ln.left + c;
                const int m = cnt[which];
                int crossings = 0;
                if (m <= CROP_EDGES)
                    for (int kk = 0; kk < m; kk++) crossings += xs[which][kk] <= x ? 1 : 0;
                else
                    crossings = crossings_direct(r, x);
                float t = fill;
                // page_index_rect.contains_point(in_p) && contains_point(out_p) (recognition.rs:100,112)
                if (j < nrows && (crossings & 1) && y >= 0 && y <= ph - 1 && x >= 0 && x <= pw - 1 && r <= ph - 1 && c <= pw - 1)
                    t = page[(int64_t)y * pw + x];
                tap[j][q] = t;
            }
#pragma unroll
        for (int j = 0; j < CROP_ROWS; j++) {
            const float top = (1.0f - wx) * tap[j][0] + wx * tap[j][1];
            const float bot = (1.0f - wx) * tap[j][2] + wx * tap[j][3];
            if (j < nrows) dst[(int64_t)j * out_w + ox] = (1.0f - wy[j]) * top + wy[j] * bot;
        }
    }
}

void crop_lines(const float* const* d_pages, const int32_t* d_page_hw, const LineDesc* d_lines, const int32_t* d_poly,
                int n_lines, int out_h, float* d_out, hipStream_t s) {
    if (n_lines <= 0) return;
    hipLaunchKernelGGL(crop_lines_kernel, dim3((out_h + CROP_ROWS - 1) / CROP_ROWS, n_lines), dim3(256), 0, s, d_pages, d_page_hw, d_lines,
                       d_poly, out_h, d_out);
}

}  // namespace k
}  // namespace ocrs
