// Text-mask post-processing on the GPU (detection.rs:41-62):
//   find_contours(mask, External) -> simplify_polygon(eps=2) -> min_area_rect
//   -> resize(w + 2*expand, h + 2*expand) -> keep area >= min_area.
//
// Design:
//  1. Connected-component labelling by union-find over BOTH pixel classes in
//     one pass: foreground with 8-connectivity, background with 4-connectivity,
//     plus a virtual frame node (-1).  Union is by minimum linear index, so
//       * the root of a foreground component is its raster-first pixel — exactly
//         the pixel at which Suzuki-Abe's raster scan starts that component's
//         outer border, and roots in index order are the reference's contour
//         discovery order;
//       * a background component touching the image frame flattens to -1, so a
//         component is an OUTERMOST one (RetrievalMode::External) iff the
//         background pixel left of its root is frame-connected (or x == 0).
//  2. Ordered compaction of the external roots (row counts -> write; a row's
//     block sums the counts of the rows above itself).
//  3. Border following from each root.  The walk only tests pixels for
//     non-zero, so it needs the binary mask, not Suzuki's marks.  One
//     wavefront per component: count walk, atomic bump of the page's contour
//     arena, write walk + simplify + rectangle.
// Everything here is integer/byte work or latency-bound geometry on a 1 MiB
// mask that lives in L2: there is no MFMA-shaped computation in this stage.
#include "common.hpp"
#include "kernels.hpp"

namespace ocrs {
namespace k {

// barrier + LDS / memory visibility inside a one-wavefront workgroup
#define WAVE_SYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
        __builtin_amdgcn_s_barrier();                            \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   \
    } while (0)

// ---------------------------------------------------------------------------
// Union-find helpers (labels are page-local linear indices; -1 = frame).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int uf_find(const int32_t* L, int x) {
    while (x >= 0) {
        int p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        x = p;
    }
    return -1;
}

__device__ __forceinline__ void uf_union(int32_t* L, int a, int b) {
    for (;;) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }  // a > b, a >= 0
        int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

// Initial labels: start of the horizontal run of equal pixels inside the
// lane's 64-pixel segment (one ballot + clz instead of 63 unions).
__global__ void __launch_bounds__(256)
ccl_init_kernel(const uint8_t* __restrict__ mask, int32_t* __restrict__ labels, int h, int w) {
    const int n = blockIdx.z;
    const int y = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint8_t* m = mask + (int64_t)n * h * w;
    int32_t* L = labels + (int64_t)n * h * w;
    const int lane = threadIdx.x & 63;
    const bool inb = x < w;
    const int v = inb ? m[y * w + x] : 2;
    const int vl = (inb && x > 0) ? m[y * w + x - 1] : 3;
    const bool boundary = (lane == 0) || (v != vl);
    const unsigned long long bal = __ballot(boundary);
    if (!inb) return;
    const unsigned long long upto = bal & (~0ull >> (63 - lane));
    const int start_lane = 63 - __clzll(upto);
    L[y * w + x] = y * w + (x - (lane - start_lane));
}

__global__ void __launch_bounds__(256)
ccl_merge_kernel(const uint8_t* __restrict__ mask, int32_t* __restrict__ labels, int h, int w) {
    const int n = blockIdx.z;
    const int y = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    const uint8_t* m = mask + (int64_t)n * h * w;
    int32_t* L = labels + (int64_t)n * h * w;
    const int p = y * w + x;
    const int v = m[p];
    const bool hasW = x > 0, hasN = y > 0, hasE = x + 1 < w;
    const int vW = hasW ? m[p - 1] : -1;
    const int vN = hasN ? m[p - w] : -1;
    const int vNW = (hasW && hasN) ? m[p - w - 1] : -1;
    const bool run_start = ((threadIdx.x & 63) == 0) || vW != v;
    if (run_start && vW == v) uf_union(L, p, p - 1);
    if (v) {
        // 8-connectivity.  p~N unless already implied through W and NW.
        if (vN == 1) {
            if (!(vW == 1 && vNW == 1)) uf_union(L, p, p - w);
        } else {
            if (vNW == 1 && vW != 1) uf_union(L, p, p - w - 1);
            if (hasN && hasE && m[p - w + 1] == 1 && m[p + 1] != 1) uf_union(L, p, p - w + 1);
        }
    } else {
        // 4-connectivity + virtual frame node.
        if (vN == 0 && !(vW == 0 && vNW == 0)) uf_union(L, p, p - w);
        const bool on_frame = (y == 0 || y == h - 1) ? run_start : false;
        if (on_frame || x == 0 || x == w - 1) uf_union(L, p, -1);
    }
}

__global__ void __launch_bounds__(256)
ccl_flatten_kernel(int32_t* __restrict__ labels, int64_t total_per_page, int n_pages) {
    const int64_t total = total_per_page * n_pages;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int32_t* L = labels + (i / total_per_page) * total_per_page;
        int p = (int)(i % total_per_page);
        L[p] = uf_find(L, p);
    }
}

// A root needs no flattened labels: L[p] == p holds for roots of the forest as it is after the merge pass, and only
// for the (few thousand) root pixels is the left neighbour's component looked up — so the separate pass that
// compressed every path of every pixel (46 us per 8 pages) is gone.
__device__ __forceinline__ bool is_external_root(const uint8_t* m, const int32_t* L, int w, int p, int x) {
    return m[p] && L[p] == p && (x == 0 || uf_find(L, p - 1) < 0);
}

// One block per (row, page).
__global__ void __launch_bounds__(256)
count_roots_kernel(const uint8_t* __restrict__ mask, const int32_t* __restrict__ labels, int h, int w,
                   int32_t* __restrict__ row_counts) {
    const int y = blockIdx.x, n = blockIdx.y;
    const uint8_t* m = mask + (int64_t)n * h * w;
    const int32_t* L = labels + (int64_t)n * h * w;
    int cnt = 0;
    for (int x = threadIdx.x; x < w; x += blockDim.x) cnt += is_external_root(m, L, w, y * w + x, x) ? 1 : 0;
    __shared__ int red[256];
    red[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) row_counts[n * h + y] = red[0];
}

// One block per (row, page).  The row's first slot = number of roots in the rows above (summed here from the row
// counts: no separate scan launch); the block of the last row also publishes the page's total.
__global__ void __launch_bounds__(256)
write_roots_kernel(const uint8_t* __restrict__ mask, const int32_t* __restrict__ labels, int h, int w,
                   const int32_t* __restrict__ row_counts, int32_t* __restrict__ roots, int32_t* __restrict__ n_roots,
                   int32_t* __restrict__ overflow, int max_comp) {
    const int y = blockIdx.x, n = blockIdx.y;
    const uint8_t* m = mask + (int64_t)n * h * w;
    const int32_t* L = labels + (int64_t)n * h * w;
    __shared__ int wave_cnt[4];
    __shared__ int red[256];
    __shared__ int base;
    int above = 0;
    for (int i = threadIdx.x; i < y; i += blockDim.x) above += row_counts[n * h + i];
    red[threadIdx.x] = above;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        base = red[0];
        if (y == h - 1) {
            const int total = red[0] + row_counts[n * h + y];
            n_roots[n] = total;
            if (total > max_comp) overflow[n] = 1;
        }
    }
    __syncthreads();
    if (row_counts[n * h + y] == 0) return;   // (uniform) most rows of a page start no component
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int x0 = 0; x0 < w; x0 += 256) {
        int x = x0 + threadIdx.x;
        bool r = x < w && is_external_root(m, L, w, y * w + x, x);
        unsigned long long bal = __ballot(r);
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int i = 0; i < wv; i++) off += wave_cnt[i];
        if (r) {
            int idx = off + __popcll(bal & ((1ull << lane) - 1));
            if (idx < max_comp) roots[(int64_t)n * max_comp + idx] = y * w + x;
        }
        __syncthreads();
        if (threadIdx.x == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// r4: the same four passes with FOUR pixels per thread (w % 4 == 0: the masks resize_threshold writes).  The byte
// kernels above issue 4-6 one-byte loads per 64 pixels and are bound by load instructions, not by bytes (8 pages:
// 17 + 101 + 19 + 16 us for 8 MiB of mask and 32 MiB of labels); here a thread loads its row's mask word and the word
// above, takes the neighbouring pixels from the neighbouring lanes' registers (DPP wave shifts; the two lanes at a
// wave's ends fetch the missing bytes themselves) and stores its four labels as one 16-byte word.  A wave's run
// segment is 256 pixels instead of 64.  The unions made are a different set with the same transitive closure, and a
// tree's root is the minimum index of its nodes whatever the order of the unions, so roots, their order and every
// later stage are unchanged (test_component_rects_*: both kernel sets against the oracle).
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_prev_u32(uint32_t v) {   // lane - 1's value (0 at lane 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t lane_next_u32(uint32_t v) {   // lane + 1's value (0 at lane 63)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);
}

__global__ void __launch_bounds__(256)
ccl_init4_kernel(const uint8_t* __restrict__ mask, int32_t* __restrict__ labels, int h, int w) {
    const int n = blockIdx.z, y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int lane = threadIdx.x & 63;
    const bool inb = x0 < w;                                   // w % 4 == 0: a thread is inside or outside as a whole
    const int64_t row = ((int64_t)n * h + y) * w;
    const uint32_t cur = inb ? *reinterpret_cast<const uint32_t*>(mask + row + x0) : 0x02020202u;
    const uint32_t prev = lane_prev_u32(cur);
    const int v0 = cur & 0xFF, v1 = (cur >> 8) & 0xFF, v2 = (cur >> 16) & 0xFF, v3 = cur >> 24;
    const bool b0 = lane == 0 || v0 != (int)(prev >> 24), b1 = v1 != v0, b2 = v2 != v1, b3 = v3 != v2;
    const int top = b3 ? 3 : b2 ? 2 : b1 ? 1 : b0 ? 0 : -1;    // the lane's last boundary
    const unsigned long long any = __ballot(top >= 0);
    const unsigned long long below = any & ((1ull << lane) - 1ull);
    const int src = below ? 63 - __clzll(below) : 0;           // the nearest lower lane that holds a boundary
    const int src_top = __shfl(top, src);
    if (!inb) return;
    const int seg = (int)(row - (int64_t)n * h * w) + x0 - 4 * lane;   // page-local index of the wave's first pixel
    const int carried = seg + 4 * src + src_top;               // run start inherited from the lanes below (lane 0 has b0)
    int4 out;
    const int s0 = b0 ? seg + 4 * lane : carried;
    const int s1 = b1 ? seg + 4 * lane + 1 : s0;
    const int s2 = b2 ? seg + 4 * lane + 2 : s1;
    const int s3 = b3 ? seg + 4 * lane + 3 : s2;
    out.x = s0; out.y = s1; out.z = s2; out.w = s3;
    *reinterpret_cast<int4*>(labels + row + x0) = out;
}

__global__ void __launch_bounds__(256)
ccl_merge4_kernel(const uint8_t* __restrict__ mask, int32_t* __restrict__ labels, int h, int w) {
    const int n = blockIdx.z, y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int lane = threadIdx.x & 63;
    const bool inb = x0 < w;
    const uint8_t* m = mask + (int64_t)n * h * w;
    int32_t* L = labels + (int64_t)n * h * w;
    const int p0 = y * w + x0;
    const bool hasN = y > 0;
    // 0xFF = "no such pixel": differs from both classes, like the -1 of the byte kernel
    const uint32_t cur = inb ? *reinterpret_cast<const uint32_t*>(m + p0) : 0xFFFFFFFFu;
    const uint32_t up = (inb && hasN) ? *reinterpret_cast<const uint32_t*>(m + p0 - w) : 0xFFFFFFFFu;
    uint32_t cur_w = lane_prev_u32(cur) >> 24, up_w = lane_prev_u32(up) >> 24;     // pixel x0 - 1
    uint32_t cur_e = lane_next_u32(cur) & 0xFF, up_e = lane_next_u32(up) & 0xFF;   // pixel x0 + 4
    if (lane == 0) {
        cur_w = (inb && x0 > 0) ? m[p0 - 1] : 0xFF;
        up_w = (inb && x0 > 0 && hasN) ? m[p0 - w - 1] : 0xFF;
    }
    if (lane == 63) {
        cur_e = (inb && x0 + 4 < w) ? m[p0 + 4] : 0xFF;
        up_e = (inb && x0 + 4 < w && hasN) ? m[p0 - w + 4] : 0xFF;
    }
    if (!inb) return;
    const uint32_t c[6] = {cur_w, cur & 0xFF, (cur >> 8) & 0xFF, (cur >> 16) & 0xFF, cur >> 24, cur_e};   // x0 - 1 .. x0 + 4
    const uint32_t u[6] = {up_w, up & 0xFF, (up >> 8) & 0xFF, (up >> 16) & 0xFF, up >> 24, up_e};
    // The unions this thread has to make, as a bit set: bit 5 j + t = pixel j with its neighbour of kind t
    // (0: W, 1: N, 2: NW, 3: NE, 4: the frame node).  They are then made one per loop iteration: a wave pays one union
    // latency (a few dependent L2 round trips) per iteration, and the number of iterations is the largest number of unions
    // any of its lanes has (1-3 inside text) — as straight-line code every one of the 20 union sites that ANY lane needs
    // costs the wave that latency.
    uint32_t todo = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = x0 + j;
        const uint32_t v = c[j + 1], vW = c[j], vN = u[j + 1], vNW = u[j], vNE = u[j + 2], vE = c[j + 2];
        const bool run_start = (lane == 0 && j == 0) || vW != v;
        uint32_t t = 0;
        if (run_start && vW == v) t |= 1u;
        if (v) {
            // 8-connectivity.  p~N unless already implied through W and NW.
            if (vN == 1) {
                if (!(vW == 1 && vNW == 1)) t |= 2u;
            } else {
                if (vNW == 1 && vW != 1) t |= 4u;
                if (vNE == 1 && vE != 1) t |= 8u;     // (vNE is 0xFF without a row above or a column to the right)
            }
        } else {
            // 4-connectivity + virtual frame node.
            if (vN == 0 && !(vW == 0 && vNW == 0)) t |= 2u;
            const bool on_frame = (y == 0 || y == h - 1) ? run_start : false;
            if (on_frame || x == 0 || x == w - 1) t |= 16u;
        }
        todo |= t << (5 * j);
    }
    while (todo) {
        const int k = __ffs(todo) - 1;
        todo &= todo - 1;
        const int j = k / 5, t = k - 5 * j;
        const int p = p0 + j;
        const int q = t == 0 ? p - 1 : t == 1 ? p - w : t == 2 ? p - w - 1 : t == 3 ? p - w + 1 : -1;
        uf_union(L, p, q);
    }
}

// is_external_root for four pixels: bit j of the result
__device__ __forceinline__ uint32_t external_roots4(const uint8_t* m, const int32_t* L, int p0, int x0) {
    const uint32_t mv = *reinterpret_cast<const uint32_t*>(m + p0);
    if (mv == 0) return 0;
    const int4 lv = *reinterpret_cast<const int4*>(L + p0);
    uint32_t r = 0;
    if ((mv & 0xFF) && lv.x == p0 && (x0 == 0 || uf_find(L, p0 - 1) < 0)) r |= 1;
    if (((mv >> 8) & 0xFF) && lv.y == p0 + 1 && uf_find(L, p0) < 0) r |= 2;
    if (((mv >> 16) & 0xFF) && lv.z == p0 + 2 && uf_find(L, p0 + 1) < 0) r |= 4;
    if ((mv >> 24) && lv.w == p0 + 3 && uf_find(L, p0 + 2) < 0) r |= 8;
    return r;
}

// One block per (row, page): counts the row's external roots, four pixels per thread.
__global__ void __launch_bounds__(256)
count_roots4_kernel(const uint8_t* __restrict__ mask, const int32_t* __restrict__ labels, int h, int w,
                    int32_t* __restrict__ row_counts) {
    const int y = blockIdx.x, n = blockIdx.y;
    const uint8_t* m = mask + (int64_t)n * h * w;
    const int32_t* L = labels + (int64_t)n * h * w;
    int cnt = 0;
    for (int x0 = threadIdx.x * 4; x0 < w; x0 += 1024) cnt += __popc(external_roots4(m, L, y * w + x0, x0));
    __shared__ int red[256];
    red[threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) row_counts[n * h + y] = red[0];
}

__global__ void __launch_bounds__(256)
write_roots4_kernel(const uint8_t* __restrict__ mask, const int32_t* __restrict__ labels, int h, int w,
                    const int32_t* __restrict__ row_counts, int32_t* __restrict__ roots, int32_t* __restrict__ n_roots,
                    int32_t* __restrict__ overflow, int max_comp) {
    const int y = blockIdx.x, n = blockIdx.y;
    const uint8_t* m = mask + (int64_t)n * h * w;
    const int32_t* L = labels + (int64_t)n * h * w;
    __shared__ int wave_cnt[4];
    __shared__ int red[256];
    __shared__ int base;
    int above = 0;
    for (int i = threadIdx.x; i < y; i += blockDim.x) above += row_counts[n * h + i];
    red[threadIdx.x] = above;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        base = red[0];
        if (y == h - 1) {
            const int total = red[0] + row_counts[n * h + y];
            n_roots[n] = total;
            if (total > max_comp) overflow[n] = 1;
        }
    }
    __syncthreads();
    if (row_counts[n * h + y] == 0) return;   // (uniform) most rows of a page start no component
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int xb = 0; xb < w; xb += 1024) {
        const int x0 = xb + threadIdx.x * 4;
        const uint32_t r = x0 < w ? external_roots4(m, L, y * w + x0, x0) : 0;
        const int mine = __popc(r);
        // exclusive prefix of `mine` over the wave's lanes (roots are numbered in x order)
        int pre = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(pre, d);
            if (lane >= d) pre += t;
        }
        if (lane == 63) wave_cnt[wv] = pre;
        __syncthreads();
        int off = base + pre - mine;
        for (int i = 0; i < wv; i++) off += wave_cnt[i];
        for (int j = 0; j < 4; j++)
            if (r & (1u << j)) {
                if (off < max_comp) roots[(int64_t)n * max_comp + off] = y * w + x0 + j;
                off++;
            }
        __syncthreads();
        if (threadIdx.x == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
}

void ccl_label(const uint8_t* d_mask, int n, int h, int w, const CclBuffers& b, int max_comp, hipStream_t s, bool prepared) {
    // option "ccl_quad" (default 1): four pixels per thread where the rows are word-aligned
    if (option(OPT_CCL_QUAD) && (w & 3) == 0 && (((uintptr_t)d_mask) & 3) == 0 && (((uintptr_t)b.labels) & 15) == 0) {
        dim3 grid4((w + 1023) / 1024, h, n);
        // (prepared: the initial labels came out of the threshold kernel, kernels_image.hip resize_threshold_kernel<true>)
        if (!prepared) hipLaunchKernelGGL(ccl_init4_kernel, grid4, dim3(256), 0, s, d_mask, b.labels, h, w);
        hipLaunchKernelGGL(ccl_merge4_kernel, grid4, dim3(256), 0, s, d_mask, b.labels, h, w);
        hipLaunchKernelGGL(count_roots4_kernel, dim3(h, n), dim3(256), 0, s, d_mask, b.labels, h, w, b.row_counts);
        hipLaunchKernelGGL(write_roots4_kernel, dim3(h, n), dim3(256), 0, s, d_mask, b.labels, h, w, b.row_counts, b.roots,
                           b.n_roots, b.overflow, max_comp);
        return;
    }
    dim3 grid((w + 255) / 256, h, n);
    if (prepared) fail(OCRS_ERR_DEVICE, "internal: labels prepared for the four-pixel component kernels, which do not apply");
    hipLaunchKernelGGL(ccl_init_kernel, grid, dim3(256), 0, s, d_mask, b.labels, h, w);
    hipLaunchKernelGGL(ccl_merge_kernel, grid, dim3(256), 0, s, d_mask, b.labels, h, w);
    hipLaunchKernelGGL(count_roots_kernel, dim3(h, n), dim3(256), 0, s, d_mask, b.labels, h, w, b.row_counts);
    hipLaunchKernelGGL(write_roots_kernel, dim3(h, n), dim3(256), 0, s, d_mask, b.labels, h, w, b.row_counts, b.roots,
                       b.n_roots, b.overflow, max_comp);
}

// ---------------------------------------------------------------------------
// Border following (Suzuki-Abe steps 3.1-3.5) from the raster-first pixel.
// Directions are indexed clockwise on screen: W NW N NE E SE S SW.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int dir_dy(int d) { return (int)((0xA901u >> (2 * d)) & 3u) - 1; }  // {0,-1,-1,-1,0,1,1,1}+1
__device__ __forceinline__ int dir_dx(int d) { return (int)((0x1A90u >> (2 * d)) & 3u) - 1; }  // {-1,-1,0,1,1,1,0,-1}+1

// The walk by a whole wavefront: lane d < 8 fetches neighbour d, a ballot gives the mask, the walk state is
// wave-uniform (scalar registers).  One memory latency and a handful of scalar instructions per border pixel —
// a single lane running trace_border pays the issue latency of every instruction of the step.
// The walk reads the mask through a WINDOW of it in LDS (kWinH rows x kWinW bytes, loaded by the whole wave with
// coalesced 16-byte loads, zeros outside the image): one LDS latency per border pixel instead of one L2 round trip,
// and a reload — one global latency — only when the walk leaves the window (a word-sized blob fits one; a 600-pixel
// text line takes ~10).  The window is re-centred on the current pixel when that happens.
constexpr int kWinW = 128, kWinH = 32;

__device__ __forceinline__ void load_window(const uint8_t* __restrict__ m, int h, int w, int wy0, int wx0, uint8_t* win, int lane) {
    // lane -> (row = lane / 2, 64-byte half of the row)
    const int r = lane >> 1, y = wy0 + r, x0 = wx0 + 64 * (lane & 1);
    uint8_t* dst = win + r * kWinW + 64 * (lane & 1);
    if ((unsigned)y < (unsigned)h && x0 >= 0 && x0 + 64 <= w) {
        const uint8_t* src = m + (int64_t)y * w + x0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint4 v;
            __builtin_memcpy(&v, src + 16 * q, 16);   // (the address is byte-aligned only: the compiler picks the load width)
            *reinterpret_cast<uint4*>(dst + 16 * q) = v;
        }
    } else {
        for (int i = 0; i < 64; i++) {
            const int x = x0 + i;
            dst[i] = ((unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w) ? m[(int64_t)y * w + x] : (uint8_t)0;
        }
    }
}

// WRITE: points go to out[0 .. cap) (LDS or global); the return value is the full length either way.
// win: kWinW * kWinH bytes of LDS.
template <bool WRITE>
__device__ int trace_border_wave(const uint8_t* __restrict__ m, int h, int w, int sy, int sx, uint32_t* out, int cap,
                                 uint8_t* win, int lane) {
    const int ldy = dir_dy(lane & 7), ldx = dir_dx(lane & 7);
    // the component lies below and to both sides of its raster-first pixel
    int wy0 = sy - 1, wx0 = sx - kWinW / 2;
    load_window(m, h, w, wy0, wx0, win, lane);
    WAVE_SYNC();
    auto neighbours = [&](int y, int x) -> unsigned {
        if (y - 1 < wy0 || y + 1 >= wy0 + kWinH || x - 1 < wx0 || x + 1 >= wx0 + kWinW) {   // (wave-uniform)
            WAVE_SYNC();
            wy0 = y - kWinH / 2;
            wx0 = x - kWinW / 2;
            load_window(m, h, w, wy0, wx0, win, lane);
            WAVE_SYNC();
        }
        const int yy = y + ldy, xx = x + ldx;
        const uint8_t v = win[(yy - wy0) * kWinW + (xx - wx0)];    // zeros outside the image
        return (unsigned)(__ballot(lane < 8 && v != 0) & 0xffull);
    };
    const unsigned nb0 = neighbours(sy, sx);
    if (nb0 == 0) {
        if (WRITE && lane == 0 && cap > 0) out[0] = ((uint32_t)sy << 16) | (uint32_t)sx;
        return 1;
    }
    const int first = __ffs((int)nb0) - 1;
    const int i1 = sy + dir_dy(first), j1 = sx + dir_dx(first);
    int i3 = sy, j3 = sx, d0 = first, n = 0;
    unsigned nb = nb0;
    for (;;) {
        // 3.3: the first set direction counter-clockwise from d0 - 1: rotate the mask so that direction d0 - 1 is bit 7,
        // d0 - 2 bit 6, ...; the highest set bit is the answer
        const unsigned rot = ((nb | (nb << 8)) >> (d0 & 7)) & 0xffu;      // bit k = direction (d0 + k) & 7; k = 0 is d0 itself
        // search order s = 1..8 -> directions d0-1, ..., d0-8 (= d0): bits 7, 6, ..., 1, then 0
        int dn = d0, i4 = i3, j4 = j3;
        const unsigned hi = rot & 0xfeu;
        if (hi) {
            const int k = 31 - __clz((int)hi);
            dn = (d0 + k) & 7;
        } else if (rot & 1u) {
            dn = d0;
        }
        if (rot) { i4 = i3 + dir_dy(dn); j4 = j3 + dir_dx(dn); }
        if (WRITE && lane == 0 && n < cap) out[n] = ((uint32_t)i3 << 16) | (uint32_t)j3;
        n++;
        if (i4 == sy && j4 == sx && i3 == i1 && j3 == j1) break;
        i3 = i4; j3 = j4;
        d0 = (dn + 4) & 7;
        nb = neighbours(i3, j3);
    }
    return n;
}

// ---- geometry restated from rten-imageproc ------------
struct P2 { float x, y; };
__device__ __forceinline__ P2 unpack_pt(uint32_t v) { return P2{(float)(v & 0xffffu), (float)(v >> 16)}; }

__device__ __forceinline__ float seg_distance(P2 a, P2 b, P2 p) {
    float abx = b.x - a.x, aby = b.y - a.y;
    float apx = p.x - a.x, apy = p.y - a.y;
    float len2 = abx * abx + aby * aby;
    if (len2 == 0.0f) return sqrtf(apx * apx + apy * apy);
    float t = (apx * abx + apy * aby) / len2;
    t = t < 0.0f ? 0.0f : t;
    t = t > 1.0f ? 1.0f : t;
    float qx = a.x + t * abx, qy = a.y + t * aby;
    float dx = p.x - qx, dy = p.y - qy;
    return sqrtf(dx * dx + dy * dy);
}

__device__ __forceinline__ float cross3(P2 o, P2 a, P2 b) {
    return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x);
}


// Ramer-Douglas-Peucker on the closed polyline p[0..n] (p[n] = p[0]): marks the kept points, returns their number.  Stack-free: each round walks the current kept points in order and splits every not-yet-final
// segment once.  keep: 0 = dropped, 1 = kept, 2 = kept and the segment starting here is final.  The surviving set does
// not depend on the order in which segments are split.  Index n (the duplicated start) is implicit: always kept.
// (forceinline: called once with LDS and once with global pointers; each copy gets its own address space.)
__device__ __forceinline__ int rdp_mark(const uint32_t* pts, uint8_t* keep, int n, float eps, int lane) {
    for (int k = lane; k < n; k += 64) keep[k] = 0;
    WAVE_SYNC();
    if (lane == 0) keep[0] = 1;
    WAVE_SYNC();
    bool changed = true;
    while (changed) {
        changed = false;
        int lo = 0;
        while (lo < n) {
            // next kept index after lo (or n)
            int hi = n;
            for (int base = lo + 1; base < n; base += 64) {
                int k = base + lane;
                unsigned long long bal = __ballot(k < n && keep[k] != 0);
                if (bal) { hi = base + __ffsll((long long)bal) - 1; break; }
            }
            const bool final_seg = keep[lo] == 2;
            if (!final_seg) {
                const P2 a = unpack_pt(pts[lo]);
                const P2 b = unpack_pt(pts[hi == n ? 0 : hi]);
                float maxd = 0.0f;
                int maxi = -1;
                for (int k = lo + 1 + lane; k < hi; k += 64) {
                    float d = seg_distance(a, b, unpack_pt(pts[k]));
                    if (d > maxd) { maxd = d; maxi = k; }
                }
                // wave arg-max, ties -> smallest index (first maximum)
                for (int o = 32; o > 0; o >>= 1) {
                    float od = __shfl_xor(maxd, o);
                    int oi = __shfl_xor(maxi, o);
                    bool take = (oi >= 0) && (maxi < 0 || od > maxd || (od == maxd && oi < maxi));
                    if (take) { maxd = od; maxi = oi; }
                }
                if (maxi >= 0 && maxd > eps) {
                    if (lane == 0) keep[maxi] = 1;
                    changed = true;
                } else {
                    if (lane == 0) keep[lo] = 2;
                }
            }
            lo = hi;
        }
        WAVE_SYNC();
    }
    int mcount = 0;   // how many points survive
    for (int base = 0; base < n; base += 64) {
        int k = base + lane;
        mcount += __popcll(__ballot(k < n && keep[k] != 0));
    }
    return mcount;
}

// ordered gather of the kept points
__device__ __forceinline__ void gather_kept(const uint32_t* pts, const uint8_t* keep, int n, uint32_t* simp, int lane) {
    int mcount = 0;
    for (int base = 0; base < n; base += 64) {
        int k = base + lane;
        bool kp = k < n && keep[k] != 0;
        unsigned long long bal = __ballot(kp);
        if (kp) simp[mcount + __popcll(bal & ((1ull << lane) - 1))] = pts[k];
        mcount += __popcll(bal);
    }
    WAVE_SYNC();
}

// Steps 4-6 for one component: rank sort of the simplified polygon, convex hull, minimum-area rectangle (+ expand,
// area test).  simp / sorted / hull: scratch of m, m and 2 m words (LDS for small polygons, the arena otherwise);
// forceinline so that each call site keeps its pointers' address space.
__device__ __forceinline__ void hull_rect(const uint32_t* simp, uint32_t* sorted, uint32_t* hull, int mcount, int* s_hn,
                                          float expand, float min_area, float* rr, uint8_t* valid_out, int lane) {
    // ---- 4. rank sort by (x, y) (index breaks ties) for the monotone chain
    for (int i = lane; i < mcount; i += 64) {
        uint32_t pi = simp[i];
        uint32_t key_i = ((pi & 0xffffu) << 16) | (pi >> 16);
        int rank = 0;
        for (int j = 0; j < mcount; j++) {
            uint32_t pj = simp[j];
            uint32_t key_j = ((pj & 0xffffu) << 16) | (pj >> 16);
            rank += (key_j < key_i || (key_j == key_i && j < i)) ? 1 : 0;
        }
        sorted[rank] = pi;
    }
    WAVE_SYNC();

    // ---- 5. convex hull (Andrew monotone chain, duplicates and collinear points dropped)
    if (lane == 0) {
        int un = 0;  // dedupe in place
        for (int i = 0; i < mcount; i++)
            if (un == 0 || sorted[i] != sorted[un - 1]) sorted[un++] = sorted[i];
        int kk = 0;
        if (un <= 2) {
            for (int i = 0; i < un; i++) hull[kk++] = sorted[i];
        } else {
            for (int i = 0; i < un; i++) {
                P2 c = unpack_pt(sorted[i]);
                while (kk >= 2 && cross3(unpack_pt(hull[kk - 2]), unpack_pt(hull[kk - 1]), c) <= 0.0f) kk--;
                hull[kk++] = sorted[i];
            }
            int lower = kk + 1;
            for (int i = un - 2; i >= 0; i--) {
                P2 c = unpack_pt(sorted[i]);
                while (kk >= lower && cross3(unpack_pt(hull[kk - 2]), unpack_pt(hull[kk - 1]), c) <= 0.0f) kk--;
                hull[kk++] = sorted[i];
            }
            kk -= 1;
        }
        *s_hn = kk;
    }
    WAVE_SYNC();
    const int hn = *s_hn;

    // ---- 6. minimum-area rectangle: exhaustive search over hull edges
    float best_area = 3.40282347e+38f;
    int best_e = -1;
    for (int e = lane; e < hn; e += 64) {
        P2 a = unpack_pt(hull[e]), b = unpack_pt(hull[e + 1 == hn ? 0 : e + 1]);
        float ex = b.x - a.x, ey = b.y - a.y;
        float len = sqrtf(ex * ex + ey * ey);
        float parx = ex / len, pary = ey / len;
        float perx = -pary, pery = parx;
        float min_par = 3.40282347e+38f, max_par = -3.40282347e+38f, max_perp = -3.40282347e+38f;
        for (int q = 0; q < hn; q++) {
            P2 c = unpack_pt(hull[q]);
            float dx = c.x - a.x, dy = c.y - a.y;
            float pp = parx * dx + pary * dy;
            float qq = perx * dx + pery * dy;
            min_par = pp < min_par ? pp : min_par;
            max_par = pp > max_par ? pp : max_par;
            max_perp = qq > max_perp ? qq : max_perp;
        }
        float area = max_perp * (max_par - min_par);
        if (area < best_area) { best_area = area; best_e = e; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        float oa = __shfl_xor(best_area, o);
        int oe = __shfl_xor(best_e, o);
        bool take = (oe >= 0) && (best_e < 0 || oa < best_area || (oa == best_area && oe < best_e));
        if (take) { best_area = oa; best_e = oe; }
    }
    if (lane == 0) {
        uint8_t ok = 0;
        if (best_e >= 0) {
            const int e = best_e;
            P2 a = unpack_pt(hull[e]), b = unpack_pt(hull[e + 1 == hn ? 0 : e + 1]);
            float ex = b.x - a.x, ey = b.y - a.y;
            float len = sqrtf(ex * ex + ey * ey);
            float parx = ex / len, pary = ey / len;
            float perx = -pary, pery = parx;
            float min_par = 3.40282347e+38f, max_par = -3.40282347e+38f, max_perp = -3.40282347e+38f;
            for (int q = 0; q < hn; q++) {
                P2 c = unpack_pt(hull[q]);
                float dx = c.x - a.x, dy = c.y - a.y;
                float pp = parx * dx + pary * dy;
                float qq = perx * dx + pery * dy;
                min_par = pp < min_par ? pp : min_par;
                max_par = pp > max_par ? pp : max_par;
                max_perp = qq > max_perp ? qq : max_perp;
            }
            float height = max_perp;
            float width = max_par - min_par;
            float along = min_par + width / 2.0f;
            float half_h = height / 2.0f;
            float ul = sqrtf(perx * perx + pery * pery);
            rr[0] = a.x + along * parx + half_h * perx;
            rr[1] = a.y + along * pary + half_h * pery;
            rr[2] = perx / ul;
            rr[3] = pery / ul;
            float ew = width + 2.0f * expand, eh = height + 2.0f * expand;
            rr[4] = ew;
            rr[5] = eh;
            ok = (ew * eh >= min_area) ? 1 : 0;
        }
        *valid_out = ok;
    }
}

constexpr int kSmallPoly = 64;   // simplified polygons up to this size keep their sort / hull scratch in LDS
constexpr int kWalkBuf = 1024;   // border points buffered in LDS by the contour kernel: LDS per 64-thread block stays near 10 KB, i.e. ~15 components in flight per CU — the stage is bound by one wave's instruction latencies, so concurrency matters more than where the data sits.  (r4: the kernel took 196 VGPRs, which allowed only 8 per CU whatever the LDS said; __launch_bounds__(64, 4) caps it at 128 — 64 registers of the rectangle geometry spill to scratch — and the 15 are real: 151 -> 138 us per 8 pages)

// One wavefront (= one 64-thread block) per component.
__global__ void __launch_bounds__(64, 4)
contour_rect_kernel(const uint8_t* __restrict__ mask, int h, int w, const int32_t* __restrict__ n_roots,
                    const int32_t* __restrict__ roots, int32_t* __restrict__ arena_top, int32_t* __restrict__ overflow,
                    uint32_t* __restrict__ pts_all, uint32_t* __restrict__ tmp_all, uint8_t* __restrict__ keep_all,
                    float* __restrict__ rects, uint8_t* __restrict__ valid, int max_comp, int64_t arena,
                    float expand, float min_area, float eps) {
    const int page = blockIdx.y;
    if (n_roots[page] > max_comp) return;    // the page's component stage is re-run with larger buffers (engine.cpp)
    const int cnt = n_roots[page];
    const uint8_t* m = mask + (int64_t)page * h * w;
    const int lane = threadIdx.x;
    for (int ci = blockIdx.x; ci < cnt; ci += gridDim.x) {
        const int64_t slot = (int64_t)page * max_comp + ci;
        const int root = roots[slot];
        // ---- 0/1. border following (serial by nature; wave-cooperative: one memory latency per border pixel).  ONE walk:
        // the points go to an LDS buffer, the length is known at the end, the component then takes its piece of the
        // page's arena with an atomic bump (placement order is irrelevant — results are indexed by slot) and the buffer
        // is copied there with coalesced stores.  Only a border longer than the buffer is walked a second time, straight
        // into the arena.  (r2: a separate count kernel, one lane per component, + a scan launch + the write walk.)
        __shared__ uint32_t walk[kWalkBuf];
        __shared__ __attribute__((aligned(16))) uint8_t win[kWinW * kWinH];
        const int n = trace_border_wave<true>(m, h, w, root / w, root % w, walk, kWalkBuf, win, lane);
        int off32 = 0;
        if (lane == 0) off32 = atomicAdd(&arena_top[page], n);
        off32 = __builtin_amdgcn_readfirstlane(off32);
        // the bump counter keeps growing after the arena is full and is 32 bits wide: on a very large noisy page it can
        // wrap, so a negative offset is an overflow too (nothing has been written for this component yet)
        if (off32 < 0 || (int64_t)off32 + n > arena) {
            if (lane == 0) {
                overflow[page] = 1;
                valid[slot] = 0;
            }
            WAVE_SYNC();
            continue;
        }
        const int64_t off = (int64_t)page * arena + off32;
        uint32_t* pts = pts_all + off;
        uint8_t* keep = keep_all + off;
        // three scratch regions per component: simplified / sorted / hull (<= 2m + 2)
        uint32_t* simp = tmp_all + (int64_t)page * arena * 4 + (int64_t)off32 * 4;
        uint32_t* sorted = simp + n;
        uint32_t* hull = sorted + n;  // 2n words
        WAVE_SYNC();   // lane 0's LDS writes visible to the wave
        // ---- 2/3. simplify (RDP) and gather the kept points.  The border stays in LDS when it fits the walk buffer
        // (one LDS latency per access in the chains of dependent reads below instead of one L2 round trip); a longer
        // one is walked again, straight into the arena, and simplified from there.
        __shared__ uint8_t keep_l[kWalkBuf];
        int mcount;
        if (n <= kWalkBuf) {
            mcount = rdp_mark(walk, keep_l, n, eps, lane);
        } else {
            trace_border_wave<true>(m, h, w, root / w, root % w, pts, n, win, lane);
            WAVE_SYNC();
            mcount = rdp_mark(pts, keep, n, eps, lane);
        }

        // ---- 4-6. hull and rectangle; polygons of up to kSmallPoly points (all but pathological ones) in LDS scratch
        __shared__ int s_hn;
        if (mcount <= kSmallPoly) {
            __shared__ uint32_t simp_l[kSmallPoly], sorted_l[kSmallPoly], hull_l[2 * kSmallPoly + 2];
            if (n <= kWalkBuf) gather_kept(walk, keep_l, n, simp_l, lane);
            else gather_kept(pts, keep, n, simp_l, lane);
            hull_rect(simp_l, sorted_l, hull_l, mcount, &s_hn, expand, min_area, rects + slot * 6, valid + slot, lane);
        } else {
            if (n <= kWalkBuf) gather_kept(walk, keep_l, n, simp, lane);
            else gather_kept(pts, keep, n, simp, lane);
            hull_rect(simp, sorted, hull, mcount, &s_hn, expand, min_area, rects + slot * 6, valid + slot, lane);
        }
        WAVE_SYNC();
    }
}

void contour_rects(const uint8_t* d_mask, int n, int h, int w, const CclBuffers& b, int max_comp, int64_t arena,
                   float expand, float min_area, float eps, hipStream_t s, bool prepared) {
    // b.offsets[0..n) serves as the per-page arena bump counter (zeroed here, or by the threshold kernel); b.lengths is unused since r3
    if (!prepared) (void)hipMemsetAsync(b.offsets, 0, (size_t)n * sizeof(int32_t), s);
    hipLaunchKernelGGL(contour_rect_kernel, dim3(2048, n), dim3(64), 0, s, d_mask, h, w, b.n_roots, b.roots, b.offsets,
                       b.overflow, b.pts, b.tmp, b.keep, b.rects, b.valid, max_comp, arena, expand, min_area, eps);
}

}  // namespace k
}  // namespace ocrs
