// The deep levels of the detection U-Net in ONE launch (Model::run of the detection model, ocrs/src/detection.rs:184;
// round 4, verdict item "fewer, fatter detection launches").
//
// Below 50 x 38 pixels a level's tensors are < 0.5 MB per page and every operator of the levels with >= 64 channels —
// MaxPool 2x2, depthwise 3x3, pointwise 1x1, ConvTranspose 2x2/s2, the [skip, pad(up)] concatenation read in place — is a
// launch of 5-20 us whose work is a fraction of that: 24 launches per request (10 small GEMMs, 2 ConvTransposes, 8 + 2
// depthwise convs, 2 pools), 0.24 ms alone, and beside other requests' conv stacks 24 separate waits for CU slots.
// Here the whole run is one persistent kernel: a few workgroups per page walk the operator list ("phases") together; a phase
// splits its output elements evenly over the page's workgroups, and a page-wide barrier (one counter word per page in global
// memory: release fence, arrive, spin, acquire fence) separates consecutive phases.  Pages never wait for each other.
//
// Arithmetic per output element is that of the per-operator kernels and of the oracle:
//   pool      m = first; m = v > m ? v : m in (ky, kx) order
//   depthwise acc = bias; acc = fmaf(x_tap, w_tap, acc) for (ky, kx) ascending, out-of-image taps as fmaf(0, w, acc)
//   pointwise / ConvTranspose   acc = bias; acc = fmaf(x[k], W[k][co], acc) for k ascending; ReLU
// — the k-ascending fmaf chain is bit for bit what v_mfma_f32_32x32x2_f32 computes in the per-operator GEMMs (tested), so
// the contraction may run on either pipe; at these sizes (108 to 1 850 rows per page) it is a few microseconds of VALU
// work per phase and runs as plain fmaf chains, one thread per (row, 4 output channels).
//
// Co-residency: the workgroups of one page wait for each other, so they must all become resident.  They are 256-thread
// workgroups without LDS and with few registers (a CU takes eight of them beside anything else), dispatched in order, and a
// page's workgroups lie within one window of 8 x kWgPerPage consecutive blocks: other kernels' blocks retire and the
// recurrence kernels never wait for anything of ours, so a partly resident window always completes.  A wait that exceeds
// ~10 s traps (the host then sees a device error instead of a hang).
#include "kernels.hpp"

namespace ocrs {
namespace k {

namespace {

constexpr int kWgPerPage = 8;

__device__ __forceinline__ void page_barrier(uint32_t* word, uint32_t target) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // this workgroup's stores of the phase -> visible device-wide
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 27)) __builtin_trap();
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the other workgroups' stores -> visible to this one's loads
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// One phase for one page: `part` of `parts` workgroups.
__device__ void run_phase(const TailPhase& p, int page, int part, int parts) {
    const int tid = threadIdx.x;
    const int stride = parts * 256;
    const int first = part * 256 + tid;
    switch (p.type) {
        case TAIL_POOL: {   // src [h2, w2, c] -> dst [h, w, c], 2x2 windows
            const int cq = p.cout >> 2;
            const int items = p.h * p.w * cq;
            const float* src = p.src + (int64_t)page * p.h2 * p.w2 * p.cout;
            float* dst = p.dst + (int64_t)page * p.h * p.w * p.cout;
            for (int i = first; i < items; i += stride) {
                const int g = i % cq, pix = i / cq;
                const int ox = pix % p.w, oy = pix / p.w;
                const float* xp = src + ((int64_t)(2 * oy) * p.w2 + 2 * ox) * p.cout + 4 * g;
                float4 m = ld4(xp);
#pragma unroll
                for (int ky = 0; ky < 2; ky++)
#pragma unroll
                    for (int kx = 0; kx < 2; kx++) {
                        const float4 v = ld4(xp + ((int64_t)ky * p.w2 + kx) * p.cout);
                        m.x = v.x > m.x ? v.x : m.x; m.y = v.y > m.y ? v.y : m.y;
                        m.z = v.z > m.z ? v.z : m.z; m.w = v.w > m.w ? v.w : m.w;
                    }
                *reinterpret_cast<float4*>(dst + (int64_t)pix * p.cout + 4 * g) = m;
            }
            break;
        }
        case TAIL_DW: {     // depthwise 3x3 over src [h, w, c]; with src2: over [src (cin channels), pad(src2 [h2, w2, c2])]
            const int c = p.cout, cs = p.src2 ? p.cin : c;
            const int cq = c >> 2;
            const int items = p.h * p.w * cq;
            const float* skip = p.src + (int64_t)page * p.h * p.w * cs;
            const float* up = p.src2 ? p.src2 + (int64_t)page * p.h2 * p.w2 * p.c2 : nullptr;
            float* dst = p.dst + (int64_t)page * p.h * p.w * c;
            const int py = (p.h - p.h2) / 2, px = (p.w - p.w2) / 2;
            for (int i = first; i < items; i += stride) {
                const int g = i % cq, pix = i / cq;
                const int ox = pix % p.w, oy = pix / p.w;
                const bool from_skip = 4 * g < cs;
                float4 acc = ld4(p.bias + 4 * g);
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        const int iy = oy + ky - 1, ix = ox + kx - 1;
                        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (from_skip) {
                            if ((unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w) xv = ld4(skip + ((int64_t)iy * p.w + ix) * cs + 4 * g);
                        } else {
                            const int uy = iy - py, ux = ix - px;
                            if ((unsigned)uy < (unsigned)p.h2 && (unsigned)ux < (unsigned)p.w2) xv = ld4(up + ((int64_t)uy * p.w2 + ux) * p.c2 + (4 * g - cs));
                        }
                        const float4 wv = ld4(p.wt + (ky * 3 + kx) * c + 4 * g);
                        acc.x = fmaf(xv.x, wv.x, acc.x); acc.y = fmaf(xv.y, wv.y, acc.y);
                        acc.z = fmaf(xv.z, wv.z, acc.z); acc.w = fmaf(xv.w, wv.w, acc.w);
                    }
                if (p.relu) {
                    acc.x = acc.x > 0.f ? acc.x : 0.f; acc.y = acc.y > 0.f ? acc.y : 0.f;
                    acc.z = acc.z > 0.f ? acc.z : 0.f; acc.w = acc.w > 0.f ? acc.w : 0.f;
                }
                *reinterpret_cast<float4*>(dst + (int64_t)pix * c + 4 * g) = acc;
            }
            break;
        }
        case TAIL_PW:       // src [rows, cin] x W [cin, ncol] -> dst [rows, ncol]            (rows = h * w)
        case TAIL_CONVT: {  // the same contraction with ncol = 4 * cout, scattered to [2h, 2w, cout] by parity (dy, dx)
            const int ncol = p.type == TAIL_CONVT ? 4 * p.cout : p.cout;
            const int nq = ncol >> 2;
            const int rows = p.h * p.w;
            const int items = rows * nq;
            const float* src = p.src + (int64_t)page * rows * p.cin;
            float* dst = p.dst + (int64_t)page * rows * ncol;   // (ConvT: 2h * 2w * cout = rows * 4 * cout)
            for (int i = first; i < items; i += stride) {
                const int q = i % nq, row = i / nq;
                const float* xr = src + (int64_t)row * p.cin;
                const float* wc = p.wt + 4 * q;
                float4 acc = ld4(p.bias + 4 * q);
                for (int k = 0; k < p.cin; k += 4) {
                    const float4 xv = ld4(xr + k);
                    const float4 w0 = ld4(wc + (int64_t)k * ncol), w1 = ld4(wc + (int64_t)(k + 1) * ncol);
                    const float4 w2 = ld4(wc + (int64_t)(k + 2) * ncol), w3 = ld4(wc + (int64_t)(k + 3) * ncol);
                    acc.x = fmaf(xv.x, w0.x, acc.x); acc.y = fmaf(xv.x, w0.y, acc.y); acc.z = fmaf(xv.x, w0.z, acc.z); acc.w = fmaf(xv.x, w0.w, acc.w);
                    acc.x = fmaf(xv.y, w1.x, acc.x); acc.y = fmaf(xv.y, w1.y, acc.y); acc.z = fmaf(xv.y, w1.z, acc.z); acc.w = fmaf(xv.y, w1.w, acc.w);
                    acc.x = fmaf(xv.z, w2.x, acc.x); acc.y = fmaf(xv.z, w2.y, acc.y); acc.z = fmaf(xv.z, w2.z, acc.z); acc.w = fmaf(xv.z, w2.w, acc.w);
                    acc.x = fmaf(xv.w, w3.x, acc.x); acc.y = fmaf(xv.w, w3.y, acc.y); acc.z = fmaf(xv.w, w3.z, acc.z); acc.w = fmaf(xv.w, w3.w, acc.w);
                }
                if (p.relu) {
                    acc.x = acc.x > 0.f ? acc.x : 0.f; acc.y = acc.y > 0.f ? acc.y : 0.f;
                    acc.z = acc.z > 0.f ? acc.z : 0.f; acc.w = acc.w > 0.f ? acc.w : 0.f;
                }
                if (p.type == TAIL_PW) {
                    *reinterpret_cast<float4*>(dst + (int64_t)row * ncol + 4 * q) = acc;
                } else {
                    const int col = 4 * q;                      // 4 consecutive columns share their parity (cout % 4 == 0)
                    const int par = col / p.cout, co = col - par * p.cout;
                    const int y = row / p.w, x = row - y * p.w;
                    const int dy = par >> 1, dx = par & 1;
                    *reinterpret_cast<float4*>(dst + (((int64_t)(2 * y + dy)) * (2 * p.w) + 2 * x + dx) * p.cout + co) = acc;
                }
            }
            break;
        }
        default: break;
    }
}

__global__ void __launch_bounds__(256)
det_tail_kernel(const TailArgs args, int n_phases, int n_pages, uint32_t* __restrict__ bar) {
    // block b -> page (b % 8) + 8 * (b / (8 * kWgPerPage)), part (b / 8) % kWgPerPage: the workgroups of a page share b % 8,
    // i.e. (as the dispatcher deals blocks to XCDs round-robin) one XCD and its L2 — speed only, the fences are device-wide
    const int b = blockIdx.x;
    const int page = (b & 7) + 8 * (b / (8 * kWgPerPage));
    const int part = (b >> 3) % kWgPerPage;
    if (page >= n_pages) return;
    for (int i = 0; i < n_phases; i++) {
        run_phase(args.ph[i], page, part, kWgPerPage);   // (kernel arguments: scalar loads, uniform across the workgroup)
        if (i + 1 < n_phases) page_barrier(bar + page, (uint32_t)kWgPerPage * (uint32_t)(i + 1));
    }
}

}  // namespace

// d_bar: n_pages words, ZEROED by the caller on the same stream.
void det_tail(const TailArgs& phases, int n_phases, int n_pages, uint32_t* d_bar, hipStream_t s) {
    if (n_phases <= 0 || n_pages <= 0) return;
    const int groups = (n_pages + 7) / 8;
    hipLaunchKernelGGL(det_tail_kernel, dim3(groups * 8 * kWgPerPage), dim3(256), 0, s, phases, n_phases, n_pages, d_bar);
}

}  // namespace k
}  // namespace ocrs
