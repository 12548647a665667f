// Fused "DoubleConv" blocks of the detection U-Net (TextDetector's Model::run, ocrs/src/detection.rs:184; ops of
// ocrs/src/wasm_api.rs:35-56 after BN folding), one launch per block instead of 4-6:
//
//   encoder:  x -> dw3x3 -> pw1x1 -> dw3x3 -> pw1x1 -> y   [-> MaxPool 2x2 -> ypool]
//   decoder:  cat(skip, pad(ConvT2x2/s2(x1))) -> dw3x3 -> pw1x1 -> dw3x3 -> pw1x1 -> y   [-> conv1x1(->1) -> sigmoid]
//
// The stack is depthwise-separable with 8..64 channels: HBM-bound, not a dense contraction, so
// the design goal is bytes: every input element is read from HBM once per tile (+ halo) and only the block's
// outputs are written.  A workgroup owns a TH x TW output tile of one image:
//   stage 0   input region (tile + 2-pixel halo) -> LDS, coalesced 16-byte loads, zeros outside the image;
//             decoder: the low-resolution x1 region -> LDS, then ConvT2x2/s2 evaluated straight into the `up`
//             channels of the LDS tile (the concatenation and the upsampled tensor never exist in HBM);
//   stage 1   dw1 + pw1 on the tile + 1-pixel halo -> LDS (zeros outside the image: that IS the next
//             depthwise conv's zero padding);
//   stage 2   dw2 + pw2 on the tile -> registers -> HBM, [final 1x1 conv + sigmoid], [2x2 max-pool via LDS].
// One thread = one pixel, all channels in registers; weights are wave-uniform and travel through the scalar
// cache (s_load), the per-pixel data through conflict-free ds_read_b128 (pixel stride = C + 4 floats).
//
// NUMERIC SPEC, identical to the unfused kernels in kernels_nn.hip and to the oracle:
//   dw:    acc = bias; for (ky,kx) ascending: acc = fmaf(x, w, acc), out-of-image taps contribute fmaf(0, w, acc)
//   pw:    acc = bias; for ci ascending: acc = fmaf(x[ci], W[ci][co], acc)
//   convT: acc = bias; for ci ascending: acc = fmaf(x1[ci], W[dy][dx][ci][co], acc)
//   relu v > 0 ? v : 0;  max-pool m = v > m ? v : m in (ky,kx) order;  sigmoid = spec_sigmoidf.
#include "common.hpp"
#include "kernels.hpp"
#include "spec_math.hpp"

namespace ocrs {
namespace k {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CS_, int CX_, int CMID_, int COUT_, int TH_, int TW_, bool POOL_, bool FINAL_>
struct DcCfg {
    static constexpr int CS = CS_, CX = CX_, CMID = CMID_, COUT = COUT_, TH = TH_, TW = TW_;
    static constexpr bool POOL = POOL_, FINAL = FINAL_, DEC = CX_ > 0;
    static constexpr int CU = DEC ? CS_ : 0;            // ConvT output channels = skip channels in this U-Net
    static constexpr int CIN = CS + CU;
    static constexpr bool VEC = (CIN % 4) == 0;         // CIN == 1 (first encoder block) takes the scalar path
    static constexpr int SA = VEC ? CIN + 4 : CIN;      // LDS pixel strides (floats)
    static constexpr int SC = CMID + 4, SE = COUT + 4, SX = CX + 4;
    static constexpr int R0H = TH + 4, R0W = TW + 4, R1H = TH + 2, R1W = TW + 2;
    static constexpr int LH = (TH + 4) / 2 + 1, LW = (TW + 4) / 2 + 1;   // low-res region of the decoder
    static constexpr int A_FLOATS = R0H * R0W * SA;
    static constexpr int E_FLOATS = POOL ? TH * TW * SE : 0;
    static constexpr int AE_FLOATS = A_FLOATS > E_FLOATS ? A_FLOATS : E_FLOATS;   // sE reuses sA's space
    static constexpr int C_FLOATS = R1H * R1W * SC;
    static constexpr int X_FLOATS = DEC ? LH * LW * SX : 0;
    static constexpr size_t LDS_BYTES = (size_t)(AE_FLOATS + C_FLOATS + X_FLOATS) * sizeof(float);
};

__device__ __forceinline__ int floor_div2(int v) { return v >> 1; }  // arithmetic shift: floor for negatives too

template <class Cfg>
__global__ void __launch_bounds__(256)
double_conv_kernel(DoubleConvArgs a) {
    constexpr int CS = Cfg::CS, CX = Cfg::CX, CU = Cfg::CU, CIN = Cfg::CIN, CMID = Cfg::CMID, COUT = Cfg::COUT;
    constexpr int TH = Cfg::TH, TW = Cfg::TW, SA = Cfg::SA, SC = Cfg::SC, SE = Cfg::SE, SX = Cfg::SX;
    constexpr int R0H = Cfg::R0H, R0W = Cfg::R0W, R1H = Cfg::R1H, R1W = Cfg::R1W, LH = Cfg::LH, LW = Cfg::LW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sA = lds;                       // [R0H*R0W][SA]   input region (skip | up)
    float* sC = lds + Cfg::AE_FLOATS;      // [R1H*R1W][SC]   after dw1+pw1
    float* sX = sC + Cfg::C_FLOATS;        // [LH*LW][SX]     decoder: low-res x1 region
    float* sE = lds;                       // [TH*TW][SE]     block output for the pool (reuses sA)
    const int tid = threadIdx.x;
    // XCD-aware tile order: consecutive tiles of an image go to the same XCD (block b runs on XCD b % 8), so
    // that the halo rows two neighbouring tiles share are fetched into one L2
    const int nblk = gridDim.x;
    const int per_xcd = (nblk + 7) / 8;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= a.n * a.tiles_y * a.tiles_x) return;
    const int img = lin / (a.tiles_y * a.tiles_x);
    const int trem = lin - img * (a.tiles_y * a.tiles_x);
    const int Y0 = (trem / a.tiles_x) * TH, X0 = (trem % a.tiles_x) * TW;
    const int h = a.h, w = a.w;

    // ---------------- stage 0: input region -> LDS
    const float* __restrict__ skip = a.skip + (int64_t)img * h * w * CS;
    if constexpr (Cfg::VEC) {
        constexpr int Q = CS / 4;
        for (int i = tid; i < R0H * R0W * Q; i += 256) {
            const int p = i / Q, c4 = i - p * Q;
            const int gy = Y0 - 2 + p / R0W, gx = X0 - 2 + p % R0W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w)
                v = *reinterpret_cast<const f32x4*>(skip + ((int64_t)gy * w + gx) * CS + c4 * 4);
            *reinterpret_cast<f32x4*>(&sA[p * SA + c4 * 4]) = v;
        }
    } else {
        for (int p = tid; p < R0H * R0W; p += 256) {
            const int gy = Y0 - 2 + p / R0W, gx = X0 - 2 + p % R0W;
            float v = 0.f;
            if ((unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w) v = skip[(int64_t)gy * w + gx];
            sA[p] = v;
        }
    }
    if constexpr (Cfg::DEC) {
        // up = zero-pad(ConvT(x1)) centred in the skip's frame (padcat: before = d/2)
        const int pyo = (h - 2 * a.h1) / 2, pxo = (w - 2 * a.w1) / 2;
        const int ly0 = floor_div2(Y0 - 2 - pyo), lx0 = floor_div2(X0 - 2 - pxo);
        const float* __restrict__ x1 = a.x1 + (int64_t)img * a.h1 * a.w1 * CX;
        constexpr int QX = CX / 4;
        for (int i = tid; i < LH * LW * QX; i += 256) {
            const int p = i / QX, c4 = i - p * QX;
            const int ly = ly0 + p / LW, lx = lx0 + p % LW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)ly < (unsigned)a.h1 && (unsigned)lx < (unsigned)a.w1)
                v = *reinterpret_cast<const f32x4*>(x1 + ((int64_t)ly * a.w1 + lx) * CX + c4 * 4);
            *reinterpret_cast<f32x4*>(&sX[p * SX + c4 * 4]) = v;
        }
        // the `up` channels default to zero (outside the image, and inside it where the padding is)
        constexpr int QU = CU / 4;
        for (int i = tid; i < R0H * R0W * QU; i += 256) {
            const int p = i / QU, c4 = i - p * QU;
            *reinterpret_cast<f32x4*>(&sA[p * SA + CS + c4 * 4]) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        // ConvTranspose 2x2 / stride 2: one thread per low-res pixel, the four output parities in turn (the
        // weights of a parity are wave-uniform)
        for (int p = tid; p < LH * LW; p += 256) {
            const int ly = ly0 + p / LW, lx = lx0 + p % LW;
            if ((unsigned)ly >= (unsigned)a.h1 || (unsigned)lx >= (unsigned)a.w1) continue;
            float xin[CX];
#pragma unroll
            for (int c4 = 0; c4 < CX / 4; c4++) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&sX[p * SX + c4 * 4]);
                xin[c4 * 4] = v[0]; xin[c4 * 4 + 1] = v[1]; xin[c4 * 4 + 2] = v[2]; xin[c4 * 4 + 3] = v[3];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int gy = 2 * ly + (q >> 1) + pyo, gx = 2 * lx + (q & 1) + pxo;
                const int ry = gy - (Y0 - 2), rx = gx - (X0 - 2);
                if ((unsigned)ry >= (unsigned)R0H || (unsigned)rx >= (unsigned)R0W) continue;
                if ((unsigned)gy >= (unsigned)h || (unsigned)gx >= (unsigned)w) continue;
                const float* __restrict__ wq = a.wt + (size_t)q * CX * CU;
#pragma unroll
                for (int co4 = 0; co4 < CU / 4; co4++) {
                    float o0 = a.bt[co4 * 4], o1 = a.bt[co4 * 4 + 1], o2 = a.bt[co4 * 4 + 2], o3 = a.bt[co4 * 4 + 3];
#pragma unroll
                    for (int ci = 0; ci < CX; ci++) {
                        o0 = fmaf(xin[ci], wq[ci * CU + co4 * 4], o0);
                        o1 = fmaf(xin[ci], wq[ci * CU + co4 * 4 + 1], o1);
                        o2 = fmaf(xin[ci], wq[ci * CU + co4 * 4 + 2], o2);
                        o3 = fmaf(xin[ci], wq[ci * CU + co4 * 4 + 3], o3);
                    }
                    *reinterpret_cast<f32x4*>(&sA[(ry * R0W + rx) * SA + CS + co4 * 4]) = f32x4{o0, o1, o2, o3};
                }
            }
        }
    }
    __syncthreads();

    // ---------------- stage 1: dw1 + pw1 on the tile + 1-pixel halo -> sC
    for (int p = tid; p < R1H * R1W; p += 256) {
        const int ry = p / R1W, rx = p - ry * R1W;
        const int gy = Y0 - 1 + ry, gx = X0 - 1 + rx;
        float* dst = &sC[p * SC];
        if ((unsigned)gy >= (unsigned)h || (unsigned)gx >= (unsigned)w) {
#pragma unroll
            for (int c4 = 0; c4 < CMID / 4; c4++) *reinterpret_cast<f32x4*>(dst + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
            continue;
        }
        float d[CIN];
        if constexpr (Cfg::VEC) {
#pragma unroll
            for (int c4 = 0; c4 < CIN / 4; c4++) {
                float a0 = a.bd1[c4 * 4], a1 = a.bd1[c4 * 4 + 1], a2 = a.bd1[c4 * 4 + 2], a3 = a.bd1[c4 * 4 + 3];
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&sA[((ry + t / 3) * R0W + rx + t % 3) * SA + c4 * 4]);
                    const float* __restrict__ wv = a.wd1 + t * CIN + c4 * 4;
                    a0 = fmaf(v[0], wv[0], a0); a1 = fmaf(v[1], wv[1], a1);
                    a2 = fmaf(v[2], wv[2], a2); a3 = fmaf(v[3], wv[3], a3);
                }
                d[c4 * 4] = a0; d[c4 * 4 + 1] = a1; d[c4 * 4 + 2] = a2; d[c4 * 4 + 3] = a3;
            }
        } else {
#pragma unroll
            for (int c = 0; c < CIN; c++) {
                float acc = a.bd1[c];
#pragma unroll
                for (int t = 0; t < 9; t++) acc = fmaf(sA[((ry + t / 3) * R0W + rx + t % 3) * SA + c], a.wd1[t * CIN + c], acc);
                d[c] = acc;
            }
        }
        if (a.relu_d1) {
#pragma unroll
            for (int c = 0; c < CIN; c++) d[c] = d[c] > 0.f ? d[c] : 0.f;
        }
#pragma unroll
        for (int co4 = 0; co4 < CMID / 4; co4++) {
            float o0 = a.bp1[co4 * 4], o1 = a.bp1[co4 * 4 + 1], o2 = a.bp1[co4 * 4 + 2], o3 = a.bp1[co4 * 4 + 3];
#pragma unroll
            for (int ci = 0; ci < CIN; ci++) {
                const float* __restrict__ wv = a.wp1 + ci * CMID + co4 * 4;
                o0 = fmaf(d[ci], wv[0], o0); o1 = fmaf(d[ci], wv[1], o1);
                o2 = fmaf(d[ci], wv[2], o2); o3 = fmaf(d[ci], wv[3], o3);
            }
            if (a.relu_p1) {
                o0 = o0 > 0.f ? o0 : 0.f; o1 = o1 > 0.f ? o1 : 0.f; o2 = o2 > 0.f ? o2 : 0.f; o3 = o3 > 0.f ? o3 : 0.f;
            }
            *reinterpret_cast<f32x4*>(dst + co4 * 4) = f32x4{o0, o1, o2, o3};
        }
    }
    __syncthreads();

    // ---------------- stage 2: dw2 + pw2 on the tile -> HBM (+ final conv / pool staging)
    float* __restrict__ yimg = a.y + (int64_t)img * h * w * (Cfg::FINAL ? 1 : COUT);
    for (int p = tid; p < TH * TW; p += 256) {
        const int ty = p / TW, tx = p - ty * TW;
        const int gy = Y0 + ty, gx = X0 + tx;
        const bool inside = gy < h && gx < w;
        float o[COUT];
        if (inside) {
            float d[CMID];
#pragma unroll
            for (int c4 = 0; c4 < CMID / 4; c4++) {
                float a0 = a.bd2[c4 * 4], a1 = a.bd2[c4 * 4 + 1], a2 = a.bd2[c4 * 4 + 2], a3 = a.bd2[c4 * 4 + 3];
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&sC[((ty + t / 3) * R1W + tx + t % 3) * SC + c4 * 4]);
                    const float* __restrict__ wv = a.wd2 + t * CMID + c4 * 4;
                    a0 = fmaf(v[0], wv[0], a0); a1 = fmaf(v[1], wv[1], a1);
                    a2 = fmaf(v[2], wv[2], a2); a3 = fmaf(v[3], wv[3], a3);
                }
                d[c4 * 4] = a0; d[c4 * 4 + 1] = a1; d[c4 * 4 + 2] = a2; d[c4 * 4 + 3] = a3;
            }
            if (a.relu_d2) {
#pragma unroll
                for (int c = 0; c < CMID; c++) d[c] = d[c] > 0.f ? d[c] : 0.f;
            }
#pragma unroll
            for (int co4 = 0; co4 < COUT / 4; co4++) {
                float o0 = a.bp2[co4 * 4], o1 = a.bp2[co4 * 4 + 1], o2 = a.bp2[co4 * 4 + 2], o3 = a.bp2[co4 * 4 + 3];
#pragma unroll
                for (int ci = 0; ci < CMID; ci++) {
                    const float* __restrict__ wv = a.wp2 + ci * COUT + co4 * 4;
                    o0 = fmaf(d[ci], wv[0], o0); o1 = fmaf(d[ci], wv[1], o1);
                    o2 = fmaf(d[ci], wv[2], o2); o3 = fmaf(d[ci], wv[3], o3);
                }
                if (a.relu_p2) {
                    o0 = o0 > 0.f ? o0 : 0.f; o1 = o1 > 0.f ? o1 : 0.f; o2 = o2 > 0.f ? o2 : 0.f; o3 = o3 > 0.f ? o3 : 0.f;
                }
                o[co4 * 4] = o0; o[co4 * 4 + 1] = o1; o[co4 * 4 + 2] = o2; o[co4 * 4 + 3] = o3;
            }
            if constexpr (Cfg::FINAL) {
                float f = a.bf[0];
#pragma unroll
                for (int c = 0; c < COUT; c++) f = fmaf(o[c], a.wf[c], f);
                yimg[(int64_t)gy * w + gx] = a.sigmoid ? spec_sigmoidf(f) : f;
            } else {
                float* yp = yimg + ((int64_t)gy * w + gx) * COUT;
#pragma unroll
                for (int co4 = 0; co4 < COUT / 4; co4++)
                    *reinterpret_cast<f32x4*>(yp + co4 * 4) = f32x4{o[co4 * 4], o[co4 * 4 + 1], o[co4 * 4 + 2], o[co4 * 4 + 3]};
            }
        }
        if constexpr (Cfg::POOL) {   // sA is dead (all of stage 1 is behind the barrier): its space holds the tile
            if (inside) {
#pragma unroll
                for (int co4 = 0; co4 < COUT / 4; co4++)
                    *reinterpret_cast<f32x4*>(&sE[p * SE + co4 * 4]) = f32x4{o[co4 * 4], o[co4 * 4 + 1], o[co4 * 4 + 2], o[co4 * 4 + 3]};
            }
        }
    }
    if constexpr (Cfg::POOL) {
        __syncthreads();
        const int ph = h / 2, pw = w / 2;
        float* __restrict__ pimg = a.ypool + (int64_t)img * ph * pw * COUT;
        constexpr int Q = COUT / 4;
        for (int i = tid; i < (TH / 2) * (TW / 2) * Q; i += 256) {
            const int pp = i / Q, c4 = i - pp * Q;
            const int py = pp / (TW / 2), px = pp - py * (TW / 2);
            const int gy = Y0 / 2 + py, gx = X0 / 2 + px;
            if (gy >= ph || gx >= pw) continue;
            const float* e = &sE[((2 * py) * TW + 2 * px) * SE + c4 * 4];
            f32x4 m = *reinterpret_cast<const f32x4*>(e);
#pragma unroll
            for (int t = 0; t < 4; t++) {   // (ky,kx) order, the first tap again is a no-op of v > m
                const f32x4 v = *reinterpret_cast<const f32x4*>(e + ((t >> 1) * TW + (t & 1)) * SE);
#pragma unroll
                for (int c = 0; c < 4; c++) m[c] = v[c] > m[c] ? v[c] : m[c];
            }
            *reinterpret_cast<f32x4*>(pimg + ((int64_t)gy * pw + gx) * COUT + c4 * 4) = m;
        }
    }
}

template <class Cfg>
void launch_dc(const DoubleConvArgs& a0, hipStream_t s) {
    DoubleConvArgs a = a0;
    a.tiles_y = (a.h + Cfg::TH - 1) / Cfg::TH;
    a.tiles_x = (a.w + Cfg::TW - 1) / Cfg::TW;
    const int tiles = a.n * a.tiles_y * a.tiles_x;
    const int grid = ((tiles + 7) / 8) * 8;
    static std::atomic<uint64_t> lds_ok{0};
    if (Cfg::LDS_BYTES > 64 * 1024) allow_dynamic_lds(reinterpret_cast<const void*>(&double_conv_kernel<Cfg>), lds_ok);
    hipLaunchKernelGGL((double_conv_kernel<Cfg>), dim3(grid), dim3(256), Cfg::LDS_BYTES, s, a);
}


// =====================================================================================================================
// r3: the same blocks with the POINTWISE convolutions on the matrix cores.
//
// Where the time of the thread-per-pixel kernel above went (rocprofv3, 8 pages): 14 000 clocks per tile per CU for
// ~2 000 clocks of LDS traffic, ~1 700 of VALU work — chains of dependent FMAs per thread at two waves per SIMD, every
// depthwise tap a 16-byte LDS read per 4 FMAs.  This kernel splits each conv pair by what it is:
//   * depthwise 3x3 (no contraction: 9 taps per channel) stays on the VALU, register-tiled: a thread owns a strip of P
//     horizontal pixels x 4 channels, loads each input row once for all three horizontal taps (3 (P + 2) LDS reads
//     per 36 P FMAs instead of 9 P) and keeps 4 P independent accumulation chains in flight;
//   * pointwise 1x1 (a dense contraction over channels) runs as v_mfma_f32_16x16x4_f32 over groups of 16 pixels:
//     A = W^T (16 output channels x 4 k, preloaded into registers once per workgroup), B = the depthwise output of
//     16 pixels straight from LDS, D = 16 output channels x 16 pixels, the accumulator initialised with the bias.  The
//     instruction is bit for bit the chain acc = fmaf(x[k], W[k][co], acc), k ascending (test_mfma_chain_is_bitwise_
//     fmaf_chain), i.e. the numeric spec of the VALU kernels.  With Cout = 8 half of the 16 rows are idle: fp32 MFMA
//     runs at the VALU's FLOP rate on gfx950, so the point is not a higher peak but a second pipe — the depthwise
//     FMAs of one workgroup and the contractions of another overlap on a CU.
// LDS holds every activation in a PERMUTED channel order, position(c) = (c % 4) * (K / 4) + c / 4 for a K-channel
// tensor: the B operand of k-step s wants lane (pixel, kq) to hold channel 4 s + kq, so lane group kq reads ONE
// contiguous run of K / 4 floats; a depthwise thread reads / writes one 16-byte block = 4 channels (which 4 is
// irrelevant to a depthwise conv); and the MFMA's rows are assigned to output channels so that the 4 rows a lane
// ends up with are again one 16-byte block of the next tensor's permuted order (natural order for the block output).
// =====================================================================================================================
template <int CS_, int CX_, int CMID_, int COUT_, int TH_, int TW_, bool POOL_, bool FINAL_>
struct McCfg {
    static constexpr int NT = 512;                        // threads per workgroup: one tile's phases are short, more waves hide their latencies
    static constexpr int CS = CS_, CX = CX_, CMID = CMID_, COUT = COUT_, TH = TH_, TW = TW_;
    static constexpr bool POOL = POOL_, FINAL = FINAL_, DEC = CX_ > 0;
    static constexpr int CU = DEC ? CS_ : 0;
    static constexpr int CIN = CS + CU;
    static constexpr bool ONE = CIN == 1;                 // first encoder block: 1 input channel, no contraction in pw1
    static constexpr int P = TW_ >= 32 ? 4 : 2;           // depthwise strip length
    static constexpr int SA = ONE ? 1 : CIN + 4, SD1 = ONE ? 1 : CIN + 4, SC = CMID + 4, SD2 = CMID + 4, SE = COUT + 4, SX = CX + 4;
    static constexpr int R0H = TH + 4, R0W = TW + 4, R1H = TH + 2, R1W = TW + 2;
    static constexpr int LH = (TH + 4) / 2 + 1, LW = (TW + 4) / 2 + 1;
    static constexpr int NPIX0 = R0H * R0W, NPIX1 = R1H * R1W, NPIX2 = TH * TW;
    // regions (floats).  CIN > 1:  U = sA -> sC -> sE,  V = sX -> sD1 -> sD2.   CIN == 1:  U = sA -> sD2,  V = sC -> sE.
    static constexpr int A_FLOATS = NPIX0 * SA, C_FLOATS = NPIX1 * SC;
    static constexpr int X_FLOATS = DEC ? LH * LW * SX : 0, D1_FLOATS = ONE ? 0 : NPIX1 * SD1, D2_FLOATS = NPIX2 * SD2;
    static constexpr int E_FLOATS = POOL ? NPIX2 * SE : 0;
    static constexpr int max3(int x, int y, int z) { return (x > y ? x : y) > z ? (x > y ? x : y) : z; }
    static constexpr int U_RAW = ONE ? max3(A_FLOATS, D2_FLOATS, 0) : max3(A_FLOATS, C_FLOATS, E_FLOATS);
    static constexpr int V_RAW = ONE ? max3(C_FLOATS, E_FLOATS, 0) : max3(X_FLOATS, D1_FLOATS, D2_FLOATS);
    static constexpr int U_FLOATS = (U_RAW + 3) / 4 * 4 + 64, V_FLOATS = (V_RAW + 3) / 4 * 4 + 64;
    static constexpr int W_FLOATS = 10 * CIN + 10 * CMID;          // depthwise weights + biases, permuted
    static constexpr size_t LDS_BYTES = (size_t)(U_FLOATS + V_FLOATS + W_FLOATS) * sizeof(float);
};

// permuted position of channel c in a K-channel tensor, and its inverse
template <int K> __device__ __forceinline__ constexpr int ch_pos(int c) { return (c & 3) * (K / 4) + (c >> 2); }
template <int K> __device__ __forceinline__ constexpr int ch_of(int n) { return (n % (K / 4)) * 4 + n / (K / 4); }

// Depthwise 3x3 over a region: src [ (rows + 2) x SRCW pixels ][S_IN] (permuted channels, zeros where the conv pads),
// dst [rows x cols pixels][S_OUT]; weights sw[tap][pos], bias sw[9][pos].  Thread = strip of P pixels x one 16-byte
// channel block.
template <int K, int P, int S_IN, int S_OUT>
__device__ __forceinline__ void dw_stage(const float* __restrict__ src, int srcw, float* __restrict__ dst, int rows, int cols,
                                         const float* __restrict__ sw, bool relu, int tid, int nt) {
    constexpr int NB = K / 4;
    const int ns = (cols + P - 1) / P;
    const int items = rows * ns * NB;
    for (int it = tid; it < items; it += nt) {
        const int blk = it % NB;
        const int t2 = it / NB;
        const int strip = t2 % ns, r = t2 / ns;
        const int x0 = strip * P;
        f32x4 acc[P];
        const f32x4 bias = *reinterpret_cast<const f32x4*>(sw + 9 * K + 4 * blk);
#pragma unroll
        for (int j = 0; j < P; j++) acc[j] = bias;
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            f32x4 row[P + 2];
#pragma unroll
            for (int j = 0; j < P + 2; j++) {
                const int xx = x0 + j < srcw ? x0 + j : srcw - 1;   // past the region's edge: feeds only outputs that are not stored
                row[j] = *reinterpret_cast<const f32x4*>(src + ((r + dy) * srcw + xx) * S_IN + 4 * blk);
            }
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(sw + (dy * 3 + dx) * K + 4 * blk);
#pragma unroll
                for (int j = 0; j < P; j++) {
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[j][e] = fmaf(row[j + dx][e], wv[e], acc[j][e]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < P; j++) {
            if (x0 + j < cols) {
                f32x4 v = acc[j];
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                *reinterpret_cast<f32x4*>(dst + (r * cols + x0 + j) * S_OUT + 4 * blk) = v;
            }
        }
    }
}

// B operand of one 16-pixel group: KS = K / 4 floats, the run [kq * KS, kq * KS + KS) of the pixel's permuted channels
template <int KS>
__device__ __forceinline__ void load_b(const float* __restrict__ p, float (&b)[KS]) {
    if constexpr (KS == 1) {
        b[0] = p[0];
    } else if constexpr (KS == 2) {
        const float2 v = *reinterpret_cast<const float2*>(p);
        b[0] = v.x; b[1] = v.y;
    } else {
#pragma unroll
        for (int q = 0; q < KS / 4; q++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * q);
            b[4 * q] = v[0]; b[4 * q + 1] = v[1]; b[4 * q + 2] = v[2]; b[4 * q + 3] = v[3];
        }
    }
}

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT)
double_conv_mfma_kernel(DoubleConvArgs a) {
    constexpr int NT = Cfg::NT, NW = Cfg::NT / 64;
    constexpr int CS = Cfg::CS, CX = Cfg::CX, CU = Cfg::CU, CIN = Cfg::CIN, CMID = Cfg::CMID, COUT = Cfg::COUT;
    constexpr int TH = Cfg::TH, TW = Cfg::TW, SA = Cfg::SA, SD1 = Cfg::SD1, SC = Cfg::SC, SD2 = Cfg::SD2, SE = Cfg::SE, SX = Cfg::SX;
    constexpr int R0H = Cfg::R0H, R0W = Cfg::R0W, R1H = Cfg::R1H, R1W = Cfg::R1W, LH = Cfg::LH, LW = Cfg::LW;
    constexpr int NPIX0 = Cfg::NPIX0, NPIX1 = Cfg::NPIX1, NPIX2 = Cfg::NPIX2, P = Cfg::P;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sU = lds;                         // sA [NPIX0][SA]  ->  sC [NPIX1][SC]
    float* sV = lds + Cfg::U_FLOATS;         // sX [LH*LW][SX]  ->  sD1 [NPIX1][SD1]  ->  sD2 [NPIX2][SD2]  ->  sE [NPIX2][SE]
    float* sW1 = sV + Cfg::V_FLOATS;         // depthwise 1: [9 taps + bias][CIN] permuted
    float* sW2 = sW1 + 10 * CIN;             // depthwise 2: [9 taps + bias][CMID] permuted
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int nblk = gridDim.x;
    const int per_xcd = (nblk + 7) / 8;
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (lin >= a.n * a.tiles_y * a.tiles_x) return;
    const int img = lin / (a.tiles_y * a.tiles_x);
    const int trem = lin - img * (a.tiles_y * a.tiles_x);
    const int Y0 = (trem / a.tiles_x) * TH, X0 = (trem % a.tiles_x) * TW;
    const int h = a.h, w = a.w;

    // ---------------- depthwise weights -> LDS (permuted)
    for (int i = tid; i < 10 * CIN; i += NT) {
        const int t = i / CIN, n = i - t * CIN;
        const int c = Cfg::ONE ? 0 : ch_of<Cfg::ONE ? 4 : CIN>(n);
        sW1[i] = t < 9 ? a.wd1[t * CIN + c] : a.bd1[c];
    }
    for (int i = tid; i < 10 * CMID; i += NT) {
        const int t = i / CMID, n = i - t * CMID;
        const int c = ch_of<CMID>(n);
        sW2[i] = t < 9 ? a.wd2[t * CMID + c] : a.bd2[c];
    }

    // ---------------- stage 0: input region -> sA (zeros outside the image)
    float* sA = sU;
    const float* __restrict__ skip = a.skip + (int64_t)img * h * w * CS;
    if constexpr (Cfg::ONE) {
        for (int p = tid; p < NPIX0; p += NT) {
            const int gy = Y0 - 2 + p / R0W, gx = X0 - 2 + p % R0W;
            float v = 0.f;
            if ((unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w) v = skip[(int64_t)gy * w + gx];
            sA[p] = v;
        }
    } else {
        constexpr int Q = CS / 4;
        for (int i = tid; i < NPIX0 * Q; i += NT) {
            const int p = i / Q, c4 = i - p * Q;
            const int gy = Y0 - 2 + p / R0W, gx = X0 - 2 + p % R0W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w)
                v = *reinterpret_cast<const f32x4*>(skip + ((int64_t)gy * w + gx) * CS + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; e++) sA[p * SA + ch_pos<CIN>(c4 * 4 + e)] = v[e];
        }
    }
    if constexpr (Cfg::DEC) {
        float* sX = sV;
        const int pyo = (h - 2 * a.h1) / 2, pxo = (w - 2 * a.w1) / 2;
        const int ly0 = floor_div2(Y0 - 2 - pyo), lx0 = floor_div2(X0 - 2 - pxo);
        const float* __restrict__ x1 = a.x1 + (int64_t)img * a.h1 * a.w1 * CX;
        constexpr int QX = CX / 4, NLOW = LH * LW;
        for (int i = tid; i < NLOW * QX; i += NT) {               // low-res region -> sX, permuted for the B operand
            const int p = i / QX, c4 = i - p * QX;
            const int ly = ly0 + p / LW, lx = lx0 + p % LW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)ly < (unsigned)a.h1 && (unsigned)lx < (unsigned)a.w1)
                v = *reinterpret_cast<const f32x4*>(x1 + ((int64_t)ly * a.w1 + lx) * CX + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; e++) sX[p * SX + ch_pos<CX>(c4 * 4 + e)] = v[e];
        }
        for (int i = tid; i < NPIX0 * CU; i += NT) {             // the `up` channels default to zero
            const int p = i / CU, c = i - p * CU;
            sA[p * SA + ch_pos<CIN>(CS + c)] = 0.f;
        }
        // ConvTranspose 2x2 / stride 2 on the matrix cores: per group of 16 low-res pixels, D rows = (parity q, output
        // channel co), n = q * CU + co — with CU = 8 two parities share one 16-row tile, so no row is idle.
        // acc = bias; acc = fmaf(x1[ci], Wt[q][ci][co], acc) for ci ascending: the spec's chain.
        constexpr int KSX = CX / 4, NGT = (4 * CU) / 16;
        float at[NGT][KSX];
        f32x4 biast[NGT];
#pragma unroll
        for (int g = 0; g < NGT; g++) {
            const int n = 16 * g + i16, q = n / CU, co = n % CU;
#pragma unroll
            for (int s4 = 0; s4 < KSX; s4++) at[g][s4] = a.wt[((size_t)q * CX + 4 * s4 + kq) * CU + co];
#pragma unroll
            for (int r = 0; r < 4; r++) biast[g][r] = a.bt[(16 * g + 4 * kq + r) % CU];
        }
        __syncthreads();
        for (int pg = wave; pg * 16 < NLOW; pg += NW) {
            const int p = pg * 16 + i16;
            const int pc = p < NLOW ? p : NLOW - 1;
            float b[KSX];
            load_b<KSX>(sX + pc * SX + kq * KSX, b);
            const int ly = ly0 + pc / LW, lx = lx0 + pc % LW;
            const bool src_ok = p < NLOW && (unsigned)ly < (unsigned)a.h1 && (unsigned)lx < (unsigned)a.w1;
#pragma unroll
            for (int g = 0; g < NGT; g++) {
                f32x4 acc = biast[g];
#pragma unroll
                for (int s4 = 0; s4 < KSX; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(at[g][s4], b[s4], acc, 0, 0, 0);
                const int n0 = 16 * g + 4 * kq;                 // this lane's 4 rows: one parity, 4 consecutive channels
                const int q = n0 / CU, co0 = n0 % CU;
                const int gy = 2 * ly + (q >> 1) + pyo, gx = 2 * lx + (q & 1) + pxo;
                const int ry = gy - (Y0 - 2), rx = gx - (X0 - 2);
                if (src_ok && (unsigned)ry < (unsigned)R0H && (unsigned)rx < (unsigned)R0W && (unsigned)gy < (unsigned)h &&
                    (unsigned)gx < (unsigned)w) {
#pragma unroll
                    for (int r = 0; r < 4; r++) sA[(ry * R0W + rx) * SA + ch_pos<CIN>(CS + co0 + r)] = acc[r];
                }
            }
        }
    }
    __syncthreads();

    // ---------------- stage 1: dw1 (VALU) -> sD1, pw1 (MFMA) -> sC   (tile + 1-pixel halo; zeros outside the image)
    float* sC = sU;
    if constexpr (Cfg::ONE) {
        // one input channel: no contraction — thread per pixel, outputs straight into sC's permuted order
        float* sCt = sV;   // (sA is still being read: build sC in V, copy is not needed — stage 2 reads from V)
        for (int p = tid; p < NPIX1; p += NT) {
            const int ry = p / R1W, rx = p - ry * R1W;
            const int gy = Y0 - 1 + ry, gx = X0 - 1 + rx;
            float* dst = &sCt[p * SC];
            if ((unsigned)gy >= (unsigned)h || (unsigned)gx >= (unsigned)w) {
#pragma unroll
                for (int c = 0; c < CMID; c++) dst[c] = 0.f;
                continue;
            }
            float d = sW1[9];
#pragma unroll
            for (int t = 0; t < 9; t++) d = fmaf(sA[(ry + t / 3) * R0W + rx + t % 3], sW1[t], d);
            if (a.relu_d1) d = d > 0.f ? d : 0.f;
#pragma unroll
            for (int c = 0; c < CMID; c++) {
                float o = fmaf(d, a.wp1[c], a.bp1[c]);
                if (a.relu_p1) o = o > 0.f ? o : 0.f;
                dst[ch_pos<CMID>(c)] = o;
            }
        }
        __syncthreads();
        sC = sCt;
    } else {
        float* sD1 = sV;
        dw_stage<CIN, P, SA, SD1>(sA, R0W, sD1, R1H, R1W, sW1, a.relu_d1 != 0, tid, NT);
        // A operand (W1^T) and bias of this lane's rows, for all k-steps: rows are output channels in sC's permuted order
        constexpr int KS = CIN / 4, NG = (CMID + 15) / 16;
        float aw[NG][KS];
        f32x4 bias1[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int n = 16 * g + i16;
            const int co = n < CMID ? ch_of<CMID>(n) : 0;
#pragma unroll
            for (int s4 = 0; s4 < KS; s4++) aw[g][s4] = n < CMID ? a.wp1[(4 * s4 + kq) * CMID + co] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nr = 16 * g + 4 * kq + r;
                bias1[g][r] = nr < CMID ? a.bp1[ch_of<CMID>(nr)] : 0.f;
            }
        }
        __syncthreads();   // sD1 complete, sA dead
        for (int pg = wave; pg * 16 < NPIX1; pg += NW) {
            const int p = pg * 16 + i16;
            const int pc = p < NPIX1 ? p : NPIX1 - 1;
            float b[KS];
            load_b<KS>(sD1 + pc * SD1 + kq * KS, b);
            const int ry = pc / R1W, rx = pc - ry * R1W;
            const int gy = Y0 - 1 + ry, gx = X0 - 1 + rx;
            const bool inside = (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                f32x4 acc = bias1[g];
#pragma unroll
                for (int s4 = 0; s4 < KS; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[g][s4], b[s4], acc, 0, 0, 0);
                if (a.relu_p1) {
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[r] = acc[r] > 0.f ? acc[r] : 0.f;
                }
                if (!inside) acc = f32x4{0.f, 0.f, 0.f, 0.f};      // the next depthwise conv's zero padding
                if (p < NPIX1 && 16 * g + 4 * kq < CMID) *reinterpret_cast<f32x4*>(sC + p * SC + 16 * g + 4 * kq) = acc;
            }
        }
        __syncthreads();   // sC complete, sD1 dead
    }

    // ---------------- stage 2: dw2 (VALU) -> sD2, pw2 (MFMA) -> registers -> HBM (+ final conv / pool staging)
    float* sD2 = Cfg::ONE ? sU : sV;    // (ONE: sC lives in V, sA in U is dead)
    dw_stage<CMID, P, SC, SD2>(sC, R1W, sD2, TH, TW, sW2, a.relu_d2 != 0, tid, NT);
    constexpr int KS2 = CMID / 4, NG2 = (COUT + 15) / 16;
    float aw2[NG2][KS2];
    f32x4 bias2[NG2];
#pragma unroll
    for (int g = 0; g < NG2; g++) {
        const int co = 16 * g + i16;                    // natural channel order for the block's output
#pragma unroll
        for (int s4 = 0; s4 < KS2; s4++) aw2[g][s4] = co < COUT ? a.wp2[(4 * s4 + kq) * COUT + co] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++) bias2[g][r] = 16 * g + 4 * kq + r < COUT ? a.bp2[16 * g + 4 * kq + r] : 0.f;
    }
    __syncthreads();       // sD2 complete, sC dead
    float* __restrict__ yimg = a.y + (int64_t)img * h * w * (Cfg::FINAL ? 1 : COUT);
    float* sE = Cfg::ONE ? sV : sU;     // pool staging: a region that is dead by now (ONE: sC in V is dead after dw2; else sC in U)
    for (int pg = wave; pg * 16 < NPIX2; pg += NW) {
        const int p = pg * 16 + i16;    // NPIX2 is a multiple of 16
        float b[KS2];
        load_b<KS2>(sD2 + p * SD2 + kq * KS2, b);
        const int ty = p / TW, tx = p - ty * TW;
        const int gy = Y0 + ty, gx = X0 + tx;
        const bool inside = gy < h && gx < w;
        f32x4 o[NG2];
#pragma unroll
        for (int g = 0; g < NG2; g++) {
            f32x4 acc = bias2[g];
#pragma unroll
            for (int s4 = 0; s4 < KS2; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw2[g][s4], b[s4], acc, 0, 0, 0);
            if (a.relu_p2) {
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = acc[r] > 0.f ? acc[r] : 0.f;
            }
            o[g] = acc;
        }
        if constexpr (Cfg::FINAL) {
            // f = bf; f = fmaf(o[c], wf[c], f) for c ascending: the chain runs through the lane groups that hold the
            // channels (kq = 0: 0..3, kq = 1: 4..7, ...), handed on by a lane broadcast
            float f = a.bf[0];
#pragma unroll
            for (int g = 0; g < NG2; g++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (16 * g + 4 * q < COUT) {
                        float fq = f;
#pragma unroll
                        for (int r = 0; r < 4; r++) fq = fmaf(o[g][r], a.wf[16 * g + 4 * q + r], fq);
                        f = __shfl(fq, i16 + 16 * q);     // the value computed by lane group q (own o[g] there)
                    }
                }
            }
            if (inside && kq == 0) yimg[(int64_t)gy * w + gx] = a.sigmoid ? spec_sigmoidf(f) : f;
        } else {
#pragma unroll
            for (int g = 0; g < NG2; g++) {
                if (16 * g + 4 * kq < COUT) {
                    if (inside) *reinterpret_cast<f32x4*>(yimg + ((int64_t)gy * w + gx) * COUT + 16 * g + 4 * kq) = o[g];
                    if constexpr (Cfg::POOL) {
                        if (inside) *reinterpret_cast<f32x4*>(sE + p * SE + 16 * g + 4 * kq) = o[g];
                    }
                }
            }
        }
    }
    if constexpr (Cfg::POOL) {
        __syncthreads();
        const int ph = h / 2, pw = w / 2;
        float* __restrict__ pimg = a.ypool + (int64_t)img * ph * pw * COUT;
        constexpr int Q = COUT / 4;
        for (int i = tid; i < (TH / 2) * (TW / 2) * Q; i += NT) {
            const int pp = i / Q, c4 = i - pp * Q;
            const int py = pp / (TW / 2), px = pp - py * (TW / 2);
            const int gy = Y0 / 2 + py, gx = X0 / 2 + px;
            if (gy >= ph || gx >= pw) continue;
            const float* e = &sE[((2 * py) * TW + 2 * px) * SE + c4 * 4];
            f32x4 m = *reinterpret_cast<const f32x4*>(e);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(e + ((t >> 1) * TW + (t & 1)) * SE);
#pragma unroll
                for (int c = 0; c < 4; c++) m[c] = v[c] > m[c] ? v[c] : m[c];
            }
            *reinterpret_cast<f32x4*>(pimg + ((int64_t)gy * pw + gx) * COUT + c4 * 4) = m;
        }
    }
}

template <class Cfg>
void launch_mc(const DoubleConvArgs& a0, hipStream_t s) {
    DoubleConvArgs a = a0;
    a.tiles_y = (a.h + Cfg::TH - 1) / Cfg::TH;
    a.tiles_x = (a.w + Cfg::TW - 1) / Cfg::TW;
    const int tiles = a.n * a.tiles_y * a.tiles_x;
    const int grid = ((tiles + 7) / 8) * 8;
    static std::atomic<uint64_t> lds_ok{0};
    if (Cfg::LDS_BYTES > 64 * 1024) allow_dynamic_lds(reinterpret_cast<const void*>(&double_conv_mfma_kernel<Cfg>), lds_ok);
    hipLaunchKernelGGL((double_conv_mfma_kernel<Cfg>), dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, s, a);
}

}  // namespace

// Shapes with a fused kernel: (skip channels, ConvT input channels or 0, mid, out, pool, final).
bool double_conv_fused(const DoubleConvArgs& a, int cs, int cx, int cmid, int cout, bool pool, bool final_conv, int fuse_level,
                       bool launch, hipStream_t s, bool* on_mfma, int* path) {
    if (path) *path = 0;
    // option "det_mfma": 1 (default) = the pointwise convs and ConvTransposes of every block shape listed below run on
    // the matrix cores (double_conv_mfma_kernel, 512 threads per tile) — including the 32-channel levels, which as
    // thread-per-pixel blocks lost to the per-op kernels and as MFMA blocks win (detection-only +4 %); 0 = the round-2
    // thread-per-pixel kernels for the shapes `fuse_level` selects.  Same bits in every mode.  Per 8 pages, MFMA vs VALU
    // kernel: encoder levels 0-2 96 / 47 / 35 vs 109 / 54 / 46 us, decoder level 1 140 vs 211, level 0 256 vs 263.
    const int mode = option(OPT_DET_MFMA);
    // option "det_rows" (default 1): the row-streaming workgroup kernels of kernels_det_rows.hip where the shape has one and
    // the request is one they take (up to 8 pages under option value 1); then option "det_stream" (default 1): the wave
    // kernels of kernels_det_stream.hip; then the LDS-tiled blocks below
    if (option(OPT_DET_ROWS) >= 1 && fuse_level >= 1 && ((!launch && a.n == 0) || a.rtape) && double_conv_rows_takes(a, cx) &&
        double_conv_rows(a, cs, cx, cmid, cout, pool, final_conv, launch, s)) {
        if (on_mfma) *on_mfma = true;
        if (path) *path = 2;
        return true;
    }
    if (option(OPT_DET_STREAM) >= 1 && fuse_level >= 1 && ((!launch && a.n == 0) || a.tape) && double_conv_stream(a, cs, cx, cmid, cout, pool, final_conv, launch, s)) {
        if (on_mfma) *on_mfma = false;
        if (path) *path = 1;
        return true;
    }
#define OCRS_DC(CS, CX, CM, CO, TH, TW, P, F)                                                   \
    if (cs == CS && cx == CX && cmid == CM && cout == CO && pool == P && final_conv == F) {      \
        if (on_mfma) *on_mfma = mode >= 1;                                                       \
        if (launch) {                                                                            \
            if (mode >= 1) launch_mc<McCfg<CS, CX, CM, CO, TH, TW, P, F>>(a, s);                  \
            else launch_dc<DcCfg<CS, CX, CM, CO, TH, TW, P, F>>(a, s);                            \
        }                                                                                        \
        return true;                                                                             \
    }
    // Shapes where one fused launch beats the per-op kernels (rocprofv3, 8 pages of 800x600, profiles/r2_det_*):
    // encoder level 0 107 us vs 202, level 1 54 vs 93, level 2 46 vs 48; decoder level 0 268 vs 531, level 1 210 vs 208.
    // The 32-channel levels (<= 200x150 pixels) lost as thread-per-pixel VALU blocks (decoder level 2: 322 us vs ~120,
    // level 3: 200 vs ~90; encoder level 3: 40 vs 31) and win as MFMA blocks (round 3): with det_mfma they are fused at
    // fuse level 1 as well (the last three shapes below); the 64-256-channel levels stay per-op.
    // encoder blocks (input -> skip [+ pooled])
    OCRS_DC(1, 0, 8, 8, 16, 32, true, false)
    OCRS_DC(8, 0, 16, 16, 8, 32, true, false)
    OCRS_DC(16, 0, 32, 32, 8, 16, true, false)
    // decoder blocks (skip + ConvT(x1) -> out [-> final conv + sigmoid])
    OCRS_DC(8, 16, 8, 8, 8, 32, false, true)
    OCRS_DC(8, 16, 8, 8, 8, 32, false, false)
    OCRS_DC(16, 32, 16, 16, 8, 16, false, false)
    if (fuse_level >= 2 || (fuse_level >= 1 && mode >= 1)) {   // every shape that has a kernel
        OCRS_DC(32, 0, 32, 32, 8, 16, true, false)
        OCRS_DC(32, 32, 32, 32, 8, 16, false, false)
        OCRS_DC(32, 64, 32, 32, 8, 16, false, false)
    }
#undef OCRS_DC
    return false;
}

}  // namespace k
}  // namespace ocrs
