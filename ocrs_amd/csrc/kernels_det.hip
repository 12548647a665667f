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
    static bool attr_set = [] {
        if (Cfg::LDS_BYTES > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&double_conv_kernel<Cfg>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
        return true;
    }();
    (void)attr_set;
    hipLaunchKernelGGL((double_conv_kernel<Cfg>), dim3(grid), dim3(256), Cfg::LDS_BYTES, s, a);
}

}  // namespace

// Shapes with a fused kernel: (skip channels, ConvT input channels or 0, mid, out, pool, final).
bool double_conv_fused(const DoubleConvArgs& a, int cs, int cx, int cmid, int cout, bool pool, bool final_conv, int fuse_level,
                       bool launch, hipStream_t s) {
#define OCRS_DC(CS, CX, CM, CO, TH, TW, P, F)                                                   \
    if (cs == CS && cx == CX && cmid == CM && cout == CO && pool == P && final_conv == F) {      \
        if (launch) launch_dc<DcCfg<CS, CX, CM, CO, TH, TW, P, F>>(a, s);                         \
        return true;                                                                             \
    }
    // Shapes where one fused launch beats the per-op kernels (rocprofv3, 8 pages of 800x600, profiles/r2_det_*):
    // encoder level 0 107 us vs 202, level 1 54 vs 93, level 2 46 vs 48; decoder level 0 268 vs 531, level 1 210 vs 208.
    // Deeper levels (C >= 32 on <= 200x150 pixels) are thread-per-pixel VALU chains on few tiles and lose
    // (decoder level 2: 322 us vs ~120, level 3: 200 vs ~90; encoder level 3: 40 vs 31), so they stay per-op.
    // encoder blocks (input -> skip [+ pooled])
    OCRS_DC(1, 0, 8, 8, 16, 32, true, false)
    OCRS_DC(8, 0, 16, 16, 8, 32, true, false)
    OCRS_DC(16, 0, 32, 32, 8, 16, true, false)
    // decoder blocks (skip + ConvT(x1) -> out [-> final conv + sigmoid])
    OCRS_DC(8, 16, 8, 8, 8, 32, false, true)
    OCRS_DC(8, 16, 8, 8, 8, 32, false, false)
    OCRS_DC(16, 32, 16, 16, 8, 16, false, false)
    if (fuse_level >= 2) {   // every shape that has a kernel (tests / experiments)
        OCRS_DC(32, 0, 32, 32, 8, 16, true, false)
        OCRS_DC(32, 32, 32, 32, 8, 16, false, false)
        OCRS_DC(32, 64, 32, 32, 8, 16, false, false)
    }
#undef OCRS_DC
    return false;
}

}  // namespace k
}  // namespace ocrs
