// Round 4: the DoubleConv blocks of the detection U-Net's 16-64-channel levels as row-streaming WORKGROUP kernels
// (TextDetector's Model::run, ocrs/src/detection.rs:184; block structure as in kernels_det.hip).
//
// kernels_det_stream.hip keeps a pixel's channels in ONE lane: that ends at 16 channels (registers).  The LDS-tiled blocks
// of kernels_det.hip that serve the wider levels pay 1.9 x halo recomputation on 8 x 16 tiles, seven barriers per tile on
// phases that fill a fraction of the threads, and 60-114 KB of LDS per workgroup (one or two workgroups per CU):
// 10 700 clocks per tile per CU for ~3 500 clocks of arithmetic.  Here a WORKGROUP of four waves owns a strip of 64 image
// columns (60 produce output) and walks down S + 4 rows of it, and the two kinds of convolution are split by what they are:
//   * depthwise 3x3 — no contraction — stays in registers exactly as in the wave kernel (a lane = a column, DPP lane
//     shifts for the horizontal taps, three rotating accumulator sets for the vertical ones, weights as an SGPR tape), with
//     the CHANNELS split over the four waves: wave v convolves channels [v C/4, (v+1) C/4) of all 64 columns;
//   * pointwise 1x1 and ConvTranspose 2x2/s2 — dense contractions — run as v_mfma_f32_16x16x4_f32 with the PIXELS split
//     over the waves: wave v contracts the 16 pixels [16 v, 16 v + 16) for all output channels.  The depthwise output row
//     goes through LDS once (64 pixels x C floats, written channel-slice-wise, read pixel-group-wise as the B operand), the
//     pointwise output row goes back the same way; A = weights^T from registers (ConvT: from LDS), accumulator initialised
//     with the bias: bit for bit the chain acc = fmaf(x[k], W[k][co], acc), k ascending.
// LDS holds ROWS, not tiles with halos: 30-60 KB per workgroup, four barriers per row, no vertical recomputation inside a
// segment (S + 4 rows in for S rows out), 6.7 % horizontal.
//
// NUMERIC SPEC: identical to kernels_det.hip / kernels_det_stream.hip / kernels_nn.hip / the oracle.
#include <vector>

#include "common.hpp"
#include "det_stream.hpp"
#include "kernels.hpp"

namespace ocrs {
namespace k {

namespace {

using namespace dstream;

template <int CS_, int CX_, int CMID_, int COUT_, bool POOL_, int S_>
struct RwCfg {
    static constexpr int CS = CS_, CX = CX_, CMID = CMID_, COUT = COUT_, S = S_;
    static constexpr bool POOL = POOL_, DEC = CX_ > 0;
    static constexpr int CU = DEC ? CS_ : 0;
    static constexpr int CIN = CS + CU;
    static constexpr int CW1 = CIN / 4, CW2 = CMID / 4;           // channels per wave in the two depthwise convs
    static constexpr int NG1 = CMID / 16, NG2 = COUT / 16;        // 16-row groups of the pointwise outputs
    static constexpr int K1 = CIN / 4, K2 = CMID / 4;             // k-steps
    static constexpr int NGT = DEC ? 2 * CU / 16 : 0;             // ConvT: rows = (column parity, channel)
    static constexpr int KT = DEC ? CX / 4 : 0;
    static constexpr int TPW = DEC ? NGT * 2 / 4 : 0;             // ConvT tasks (row group x pixel group) per wave
    static constexpr int SX = CX + 4, SU = CU + 4, SD1 = CIN + 4, SM = CMID + 4, SD2 = CMID + 4;   // LDS pixel strides
    static constexpr int X_FLOATS = DEC ? 32 * SX : 0, UP_FLOATS = DEC ? 64 * SU : 0;
    static constexpr int D1_FLOATS = 64 * SD1, M_FLOATS = 64 * SM, D2_FLOATS = 64 * SD2;
    static constexpr int WT_FLOATS = DEC ? 2 * NGT * KT * 64 : 0;  // ConvT weights as A operands, both row parities
    static constexpr size_t LDS_BYTES = (size_t)(X_FLOATS + UP_FLOATS + D1_FLOATS + M_FLOATS + D2_FLOATS + WT_FLOATS) * sizeof(float);
    // per-wave tape: dw1 [CW1 / 2 pairs][bias, 9 taps], dw2 [CW2 / 2 pairs][bias, 9 taps]
    static constexpr int P_DW1 = 0, N_DW1 = 10 * CW1, P_DW2 = N_DW1, N_DW2 = 10 * CW2;
    static constexpr int USED = N_DW1 + N_DW2;
    static constexpr int STG = kStage;                             // tape stage (floats); 32 measured: 75 SGPRs spilled, decoder level 1 79 vs 71 us
    static constexpr int NST = (USED + STG - 1) / STG;
    static constexpr int LEN = NST * STG;                          // floats per wave
    static_assert((S + 4) % 3 == 0 && S % 2 == 0, "");
    static_assert(CW1 % 2 == 0 && CW2 % 2 == 0 && CMID % 16 == 0 && COUT % 16 == 0 && CIN % 4 == 0, "");
    static_assert(!DEC || (CS % CW1 == 0 && (2 * CU) % 16 == 0 && (NGT * 2) % 4 == 0 && CX % 4 == 0), "");
};

template <class Cfg>
struct RGeo {
    int img, c0, col, Y0, h, w, lane, wave, i16, kq;
    bool col_ok;
    int pyo, pxo, h1, w1, lx0;
    __amdgpu_buffer_rsrc_t skip_rs, x1_rs;
    int skip_off;                                   // byte offset of (the lane's column, the wave's channel slice) inside a row
    bool slice_is_skip;
};

// the wave's slice of a skip row: CW1 floats of the lane's pixel
template <class Cfg>
__device__ __forceinline__ void load_skip_slice(const RGeo<Cfg>& g, int i, float (&out)[Cfg::CW1]) {
    const int voff = ((unsigned)i < (unsigned)g.h && g.col_ok) ? i * (g.w * Cfg::CS * 4) + g.skip_off : kOobOffset;
    if constexpr (Cfg::CW1 == 2) {   // (two dword loads: the 64-bit raw buffer load builtin returned other data here)
        out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g.skip_rs, voff, 0, 0));
        out[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g.skip_rs, voff + 4, 0, 0));
    } else {
#pragma unroll
        for (int q = 0; q < Cfg::CW1 / 4; q++) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g.skip_rs, voff + 16 * q, 0, 0));
            out[4 * q] = v[0]; out[4 * q + 1] = v[1]; out[4 * q + 2] = v[2]; out[4 * q + 3] = v[3];
        }
    }
}

// this thread's part of the low-resolution row that feeds input row i: X_PER float4 of [32 pixels][CX]
template <class Cfg>
__device__ __forceinline__ void load_x1_part(const RGeo<Cfg>& g, int i, f32x4 (&part)[(Cfg::DEC ? 32 * Cfg::CX / 4 : 256) / 256], int tid) {
    constexpr int QX = Cfg::CX / 4, PER = 32 * QX / 256;
    const int uy = i - g.pyo;
    const bool row_ok = (unsigned)uy < (unsigned)(2 * g.h1);
#pragma unroll
    for (int e = 0; e < PER; e++) {
        const int idx = tid + 256 * e;
        const int j = idx / QX, c4 = idx - j * QX;
        const int lx = g.lx0 + j;
        const int voff = (row_ok && (unsigned)lx < (unsigned)g.w1) ? ((uy >> 1) * g.w1 + lx) * (Cfg::CX * 4) + 16 * c4 : kOobOffset;
        part[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g.x1_rs, voff, 0, 0));
    }
}
template <class Cfg>
__device__ __forceinline__ void store_x1_part(float* sX, const f32x4 (&part)[(Cfg::DEC ? 32 * Cfg::CX / 4 : 256) / 256], int tid) {
    constexpr int QX = Cfg::CX / 4, PER = 32 * QX / 256;
#pragma unroll
    for (int e = 0; e < PER; e++) {
        const int idx = tid + 256 * e;
        const int j = idx / QX, c4 = idx - j * QX;
        *reinterpret_cast<f32x4*>(sX + j * Cfg::SX + 4 * c4) = part[e];
    }
}

template <class Cfg>
struct RState {
    float acc1[3][Cfg::CW1];
    float acc2[3][Cfg::CW2];
    float nxt[Cfg::CW1];                               // the wave's slice of the next skip row
    f32x4 xpart[(Cfg::DEC ? 32 * Cfg::CX / 4 : 256) / 256];   // this thread's part of the next low-resolution row
    float aw1[Cfg::NG1][Cfg::K1], aw2[Cfg::NG2][Cfg::K2];      // pointwise weights as A operands
    f32x4 bias1[Cfg::NG1], bias2[Cfg::NG2], biast[Cfg::DEC ? Cfg::TPW : 1];
    f32x4 prev[Cfg::POOL ? Cfg::NG2 : 1];              // the previous output row of this lane's pixel, for the pool
};

template <class Cfg, int PH>
__device__ __forceinline__ void rows_step(const DoubleConvArgs& a, const RGeo<Cfg>& g, RState<Cfg>& st, float* lds, int t) {
    constexpr int CS = Cfg::CS, CU = Cfg::CU, CW1 = Cfg::CW1, CW2 = Cfg::CW2, S = Cfg::S;
    float* sX = lds;
    float* sUP = sX + Cfg::X_FLOATS;
    float* sD1 = sUP + Cfg::UP_FLOATS;
    float* sM = sD1 + Cfg::D1_FLOATS;
    float* sD2 = sM + Cfg::M_FLOATS;
    float* sWT = sD2 + Cfg::D2_FLOATS;
    const int tid = threadIdx.x;
    const int i = g.Y0 - 2 + t;                        // the arriving input row
    Tape<Cfg::NST, Cfg::STG> tape;
    tape.base = (cfp)(uintptr_t)(a.rtape + (size_t)g.wave * Cfg::LEN);
    tape.template issue<0>();

    // ---- the wave's slice of the input row: skip channels from HBM (fetched a step ahead), or ConvTranspose channels
    float in[CW1];
#pragma unroll
    for (int c = 0; c < CW1; c++) in[c] = st.nxt[c];
    if (g.slice_is_skip && t + 1 < S + 4) load_skip_slice<Cfg>(g, i + 1, st.nxt);
    if constexpr (Cfg::DEC) {
        // ConvTranspose of the low-resolution row in sX on the matrix cores: D rows = (column parity, channel) for this
        // row's parity, columns = 16 low-resolution pixels; a lane ends up with 4 consecutive channels of one output pixel
        const int uy = i - g.pyo;
        const int ypar = uy & 1;
        const bool row_up_ok = (unsigned)uy < (unsigned)(2 * g.h1);
#pragma unroll
        for (int q = 0; q < Cfg::TPW; q++) {
            const int id = g.wave + 4 * q, gi = id >> 1, pg = id & 1;
            f32x4 acc = st.biast[q];
            const float* __restrict__ ap = sWT + ((ypar * Cfg::NGT + gi) * Cfg::KT) * 64 + g.lane;
            const float* __restrict__ bp = sX + (16 * pg + g.i16) * Cfg::SX + g.kq;
#pragma unroll
            for (int s4 = 0; s4 < Cfg::KT; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[64 * s4], bp[4 * s4], acc, 0, 0, 0);
            const int n0 = 16 * gi + 4 * g.kq, xpar = n0 / CU, co0 = n0 - xpar * CU;
            const int cc = 2 * (16 * pg + g.i16) + xpar;             // strip column
            const int col = g.c0 + cc;
            const bool ok = row_up_ok && (unsigned)col < (unsigned)g.w && (unsigned)(col - g.pxo) < (unsigned)(2 * g.w1);
            *reinterpret_cast<f32x4*>(sUP + cc * Cfg::SU + co0) = ok ? acc : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();                                // B1: the `up` row is complete, sX is free
        if (t + 1 < S + 4) {
            store_x1_part<Cfg>(sX, st.xpart, tid);      // the low-resolution row of the next step (visible after B2)
            if (t + 2 < S + 4) load_x1_part<Cfg>(g, i + 2, st.xpart, tid);
        }
        if (!g.slice_is_skip) {
            const float* __restrict__ up = sUP + g.lane * Cfg::SU + (g.wave * CW1 - CS);
#pragma unroll
            for (int q = 0; q < CW1 / 4; q++) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(up + 4 * q);
                in[4 * q] = v[0]; in[4 * q + 1] = v[1]; in[4 * q + 2] = v[2]; in[4 * q + 3] = v[3];
            }
        }
    }

    // ---- conv pair 1: depthwise on the wave's channel slice (completes row i - 1) -> sD1
    constexpr int NEW = (PH + 1) % 3, MID = PH, OLD = (PH + 2) % 3;
    dw_row<CW1, NEW, MID, OLD, Cfg::P_DW1>(in, st.acc1, tape);
    {
        float d1[CW1];
#pragma unroll
        for (int c = 0; c < CW1; c++) d1[c] = st.acc1[OLD][c];
        if (a.relu_d1) {
#pragma unroll
            for (int c = 0; c < CW1; c++) d1[c] = relu1(d1[c]);
        }
        float* dst = sD1 + g.lane * Cfg::SD1 + g.wave * CW1;
        if constexpr (CW1 == 2) {
            *reinterpret_cast<f32x2*>(dst) = f32x2{d1[0], d1[1]};
        } else {
#pragma unroll
            for (int q = 0; q < CW1 / 4; q++) *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{d1[4 * q], d1[4 * q + 1], d1[4 * q + 2], d1[4 * q + 3]};
        }
    }
    __syncthreads();                                    // B2: the depthwise row is complete
    // pointwise 1 on the wave's 16 pixels -> sM (zeros outside the image: the next depthwise conv's padding)
    {
        const int r1 = i - 1;
        const int col = g.c0 + 16 * g.wave + g.i16;
        const bool mid_ok = t >= 2 && (unsigned)r1 < (unsigned)g.h && (unsigned)col < (unsigned)g.w;
        const float* __restrict__ bp = sD1 + (16 * g.wave + g.i16) * Cfg::SD1 + g.kq;
#pragma unroll
        for (int gi = 0; gi < Cfg::NG1; gi++) {
            f32x4 acc = st.bias1[gi];
#pragma unroll
            for (int s4 = 0; s4 < Cfg::K1; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(st.aw1[gi][s4], bp[4 * s4], acc, 0, 0, 0);
            if (a.relu_p1) {
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = relu1(acc[r]);
            }
            *reinterpret_cast<f32x4*>(sM + (16 * g.wave + g.i16) * Cfg::SM + 16 * gi + 4 * g.kq) = mid_ok ? acc : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();                                    // B3: the pointwise row is complete

    // ---- conv pair 2: depthwise on the wave's slice of that row (completes output row i - 2) -> sD2
    constexpr int PH2 = (PH + 1) % 3;
    constexpr int NEW2 = (PH2 + 1) % 3, MID2 = PH2, OLD2 = (PH2 + 2) % 3;
    {
        float mid[CW2];
        const float* __restrict__ mp = sM + g.lane * Cfg::SM + g.wave * CW2;
#pragma unroll
        for (int q = 0; q < CW2 / 4; q++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(mp + 4 * q);
            mid[4 * q] = v[0]; mid[4 * q + 1] = v[1]; mid[4 * q + 2] = v[2]; mid[4 * q + 3] = v[3];
        }
        dw_row<CW2, NEW2, MID2, OLD2, Cfg::P_DW2>(mid, st.acc2, tape);
        float d2[CW2];
#pragma unroll
        for (int c = 0; c < CW2; c++) d2[c] = st.acc2[OLD2][c];
        if (a.relu_d2) {
#pragma unroll
            for (int c = 0; c < CW2; c++) d2[c] = relu1(d2[c]);
        }
        float* dst = sD2 + g.lane * Cfg::SD2 + g.wave * CW2;
#pragma unroll
        for (int q = 0; q < CW2 / 4; q++) *reinterpret_cast<f32x4*>(dst + 4 * q) = f32x4{d2[4 * q], d2[4 * q + 1], d2[4 * q + 2], d2[4 * q + 3]};
    }
    __syncthreads();                                    // B4: the second depthwise row is complete
    // pointwise 2 on the wave's 16 pixels -> HBM (+ 2x2 max-pool with the previous row)
    {
        const int r2 = i - 2;                           // = Y0 + t - 4
        const int cc = 16 * g.wave + g.i16, col = g.c0 + cc;
        const bool store_ok = t >= 4 && r2 < g.h && cc >= 2 && cc < 2 + kValid && (unsigned)col < (unsigned)g.w;
        const float* __restrict__ bp = sD2 + cc * Cfg::SD2 + g.kq;
        f32x4 o[Cfg::NG2];
#pragma unroll
        for (int gi = 0; gi < Cfg::NG2; gi++) {
            f32x4 acc = st.bias2[gi];
#pragma unroll
            for (int s4 = 0; s4 < Cfg::K2; s4++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(st.aw2[gi][s4], bp[4 * s4], acc, 0, 0, 0);
            if (a.relu_p2) {
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = relu1(acc[r]);
            }
            o[gi] = acc;
            if (store_ok) *reinterpret_cast<f32x4*>(a.y + (((int64_t)g.img * g.h + r2) * g.w + col) * Cfg::COUT + 16 * gi + 4 * g.kq) = acc;
        }
        if constexpr (Cfg::POOL) {
            if ((t & 1) == 0) {                         // Y0 is even: an even output row, the upper half of a pooling pair
#pragma unroll
                for (int gi = 0; gi < Cfg::NG2; gi++) st.prev[gi] = o[gi];
            } else {
                const int ph = g.h / 2, pw = g.w / 2;
                const int py = r2 >> 1, px = col >> 1;
#pragma unroll
                for (int gi = 0; gi < Cfg::NG2; gi++) {
                    f32x4 m;
#pragma unroll
                    for (int r = 0; r < 4; r++) {       // (ky,kx) order; the lane holds kx = 0 if its column is even
                        float v = st.prev[gi][r];
                        const float b = lane_pair(st.prev[gi][r]);
                        v = b > v ? b : v;
                        v = o[gi][r] > v ? o[gi][r] : v;
                        const float d = lane_pair(o[gi][r]);
                        v = d > v ? d : v;
                        m[r] = v;
                    }
                    if (store_ok && (cc & 1) == 0 && py < ph && px < pw)
                        *reinterpret_cast<f32x4*>(a.ypool + (((int64_t)g.img * ph + py) * pw + px) * Cfg::COUT + 16 * gi + 4 * g.kq) = m;
                }
            }
        }
    }
}

template <class Cfg>
__global__ void __launch_bounds__(256) rows_block_kernel(DoubleConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    RGeo<Cfg> g;
    const int tid = threadIdx.x;
    g.lane = tid & 63;
    g.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    g.i16 = g.lane & 15; g.kq = g.lane >> 4;
    const int strips = a.tiles_x, segs = a.tiles_y;
    const int wid = blockIdx.x;
    g.img = wid / (strips * segs);
    const int rem = wid - g.img * (strips * segs);
    const int seg = rem / strips, strip = rem - seg * strips;
    g.c0 = strip * kValid - 2;
    g.col = g.c0 + g.lane;
    g.Y0 = seg * Cfg::S;
    g.h = a.h; g.w = a.w;
    g.col_ok = (unsigned)g.col < (unsigned)a.w;
    g.h1 = a.h1; g.w1 = a.w1;
    g.pyo = Cfg::DEC ? (a.h - 2 * a.h1) / 2 : 0;
    g.pxo = Cfg::DEC ? (a.w - 2 * a.w1) / 2 : 0;
    g.lx0 = (g.c0 - g.pxo) >> 1;                       // (c0 - pxo is even: checked by the launcher)
    g.slice_is_skip = g.wave * Cfg::CW1 < Cfg::CS;
    g.skip_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.skip) + (int64_t)g.img * a.h * a.w * Cfg::CS, 0, a.h * a.w * Cfg::CS * 4, 0x00020000);
    g.skip_off = (g.col * Cfg::CS + g.wave * Cfg::CW1) * 4;
    if constexpr (Cfg::DEC)
        g.x1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x1) + (int64_t)g.img * a.h1 * a.w1 * Cfg::CX, 0, a.h1 * a.w1 * Cfg::CX * 4, 0x00020000);
    else
        g.x1_rs = g.skip_rs;

    RState<Cfg> st;
#pragma unroll
    for (int p = 0; p < 3; p++) {
#pragma unroll
        for (int c = 0; c < Cfg::CW1; c++) st.acc1[p][c] = 0.f;
#pragma unroll
        for (int c = 0; c < Cfg::CW2; c++) st.acc2[p][c] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < (Cfg::POOL ? Cfg::NG2 : 1); q++) st.prev[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // pointwise weights as A operands (row n = output channel 16 g + i16, k = 4 s + kq) and the biases of the lane's rows
#pragma unroll
    for (int gi = 0; gi < Cfg::NG1; gi++) {
#pragma unroll
        for (int s4 = 0; s4 < Cfg::K1; s4++) st.aw1[gi][s4] = a.wp1[(4 * s4 + g.kq) * Cfg::CMID + 16 * gi + g.i16];
#pragma unroll
        for (int r = 0; r < 4; r++) st.bias1[gi][r] = a.bp1[16 * gi + 4 * g.kq + r];
    }
#pragma unroll
    for (int gi = 0; gi < Cfg::NG2; gi++) {
#pragma unroll
        for (int s4 = 0; s4 < Cfg::K2; s4++) st.aw2[gi][s4] = a.wp2[(4 * s4 + g.kq) * Cfg::COUT + 16 * gi + g.i16];
#pragma unroll
        for (int r = 0; r < 4; r++) st.bias2[gi][r] = a.bp2[16 * gi + 4 * g.kq + r];
    }
    if constexpr (Cfg::DEC) {
        float* sWT = lds + Cfg::X_FLOATS + Cfg::UP_FLOATS + Cfg::D1_FLOATS + Cfg::M_FLOATS + Cfg::D2_FLOATS;
        // ConvT weights as A operands for both row parities: [ypar][row group][k-step][lane]
        for (int idx = tid; idx < Cfg::WT_FLOATS; idx += 256) {
            const int ln = idx & 63, rest = idx >> 6;
            const int s4 = rest % Cfg::KT, gi = (rest / Cfg::KT) % Cfg::NGT, ypar = rest / (Cfg::KT * Cfg::NGT);
            const int n = 16 * gi + (ln & 15), xpar = n / Cfg::CU, co = n - xpar * Cfg::CU;
            sWT[idx] = a.wt[((size_t)(ypar * 2 + xpar) * Cfg::CX + 4 * s4 + (ln >> 4)) * Cfg::CU + co];
        }
#pragma unroll
        for (int q = 0; q < Cfg::TPW; q++) {
            const int gi = (g.wave + 4 * q) >> 1;
#pragma unroll
            for (int r = 0; r < 4; r++) st.biast[q][r] = a.bt[(16 * gi + 4 * g.kq + r) % Cfg::CU];
        }
        // the low-resolution row of step 0, and this thread's part of step 1's
        load_x1_part<Cfg>(g, g.Y0 - 2, st.xpart, tid);
        store_x1_part<Cfg>(lds, st.xpart, tid);
        load_x1_part<Cfg>(g, g.Y0 - 1, st.xpart, tid);
    }
#pragma unroll
    for (int c = 0; c < Cfg::CW1; c++) st.nxt[c] = 0.f;
    if (g.slice_is_skip) load_skip_slice<Cfg>(g, g.Y0 - 2, st.nxt);
    __syncthreads();
    for (int t = 0; t < Cfg::S + 4; t += 3) {
        if (g.Y0 + t - 4 >= g.h) break;               // nothing below the image (uniform over the workgroup)
        rows_step<Cfg, 0>(a, g, st, lds, t);
        rows_step<Cfg, 1>(a, g, st, lds, t + 1);
        rows_step<Cfg, 2>(a, g, st, lds, t + 2);
    }
}

template <class Cfg>
void launch_rows(const DoubleConvArgs& a0, hipStream_t s) {
    DoubleConvArgs a = a0;
    a.tiles_x = (a.w + kValid - 1) / kValid;
    a.tiles_y = (a.h + Cfg::S - 1) / Cfg::S;
    static std::atomic<uint64_t> lds_ok{0};
    if (Cfg::LDS_BYTES > 64 * 1024) allow_dynamic_lds(reinterpret_cast<const void*>(&rows_block_kernel<Cfg>), lds_ok);
    hipLaunchKernelGGL((rows_block_kernel<Cfg>), dim3(a.n * a.tiles_x * a.tiles_y), dim3(256), Cfg::LDS_BYTES, s, a);
}

// The four per-wave tapes of a block: wave v's depthwise weights (channel slices of dw1 and dw2) in the order its row step
// reads them.
template <class Cfg>
std::vector<float> build_rows_tape(const StreamWeights& w) {
    std::vector<float> tape((size_t)4 * Cfg::LEN, 0.f);
    for (int v = 0; v < 4; v++) {
        float* t = tape.data() + (size_t)v * Cfg::LEN;
        auto dw = [&](int P0, int C, int CW, const float* wd, const float* bd) {
            for (int q = 0; q < CW / 2; q++)
                for (int e = 0; e < 2; e++) {
                    const int c = v * CW + 2 * q + e;
                    t[P0 + 20 * q + e] = bd[c];
                    for (int tap = 0; tap < 9; tap++) t[P0 + 20 * q + 2 * (1 + tap) + e] = wd[tap * C + c];
                }
        };
        dw(Cfg::P_DW1, Cfg::CIN, Cfg::CW1, w.wd1, w.bd1);
        dw(Cfg::P_DW2, Cfg::CMID, Cfg::CW2, w.wd2, w.bd2);
    }
    return tape;
}

}  // namespace

// The launch-time conditions.  Under option value 1 requests of more than 8 pages keep the other kernels: such requests are
// the 16-page batches of the full pipeline, whose detection kernels run beside other requests' recognition conv stacks — and
// there short-lived workgroups (~10 us each for the tiled blocks) co-schedule better than these (a workgroup lives 30-70 us
// and keeps its CU slot): default bench 271.5 / 271.6 / 272.2 / 270.6 pages/s without these against 267.2 / 270.0 / 269.7 with
// them (ABAB, one box), while a detection request on its own is 8-10 % faster with these.
bool double_conv_rows_takes(const DoubleConvArgs& a, int cx) {
    const int opt = option(OPT_DET_ROWS);
    if (opt < 1) return false;
    if (opt == 1 && a.n > 8) return false;
    if (cx > 0 && (((a.w - 2 * a.w1) / 2) & 1)) return false;   // the strip's first column must map to the first half of a low-resolution pixel
    return true;
}

// Shapes with a workgroup row-streaming kernel (option "det_rows"); same contract as double_conv_stream.
// Rows per workgroup: a workgroup's run time is (S + 4) x the latency of one row step (tape stages, four barriers, LDS
// round trips, dependent MFMA chains: ~2 us), and what hides it is other workgroups on the same CU — so the launch wants
// as many workgroups as can be RESIDENT at once (OCC per CU from registers / LDS, x 256 CUs) and no more: the smallest S of
// 8 / 14 / 20 / 32 whose workgroup count fits one round.
bool double_conv_rows(const DoubleConvArgs& a, int cs, int cx, int cmid, int cout, bool pool, bool final_conv, bool launch, hipStream_t s,
                      const StreamWeights* hw, std::vector<float>* tape_out, int* tape_len) {
    if (final_conv) return false;
    auto groups = [&](int S) { return (int64_t)a.n * ((a.w + kValid - 1) / kValid) * ((a.h + S - 1) / S); };
    const int opt = option(OPT_DET_ROWS);     // 1: by the rule above; 8 / 14 / 20 / 32: that segment height (tests, A/B)
    auto pick = [&](int occ) {
        if (opt == 8 || opt == 14 || opt == 20 || opt == 32) return opt;
        const int64_t cap = (int64_t)occ * (launch ? ctx().cu_count() : 256);
        return groups(8) <= cap ? 8 : groups(14) <= cap ? 14 : groups(20) <= cap ? 20 : 32;
    };
    // the request-dependent conditions also apply to a query made with the request's arguments (a.n > 0): the caller may
    // have no other fused kernel for the shape and must know before it commits to the fused path
    const bool real = launch || a.n > 0;
    if (real && !double_conv_rows_takes(a, cx)) return false;
    if (real && ((int64_t)a.h * a.w * cs * 4 >= kOobOffset || (int64_t)a.h1 * a.w1 * cx * 4 >= kOobOffset)) return false;
#define OCRS_RW(CS, CX, CM, CO, P, OCC)                                                           \
    if (cs == CS && cx == CX && cmid == CM && cout == CO && pool == P) {                            \
        typedef RwCfg<CS, CX, CM, CO, P, 32> Cfg;                                                  \
        if (tape_out) *tape_out = build_rows_tape<Cfg>(*hw);                                       \
        if (tape_len) *tape_len = 4 * Cfg::LEN;                                                    \
        if (launch) {                                                                              \
            if (!a.rtape || a.rtape_len != 4 * Cfg::LEN) fail(OCRS_ERR_RUN_FAILED, "row-streaming DoubleConv block without its weight tape"); \
            const int S = pick(OCC);                                                               \
            if (S == 32) launch_rows<Cfg>(a, s);                                                   \
            else if (S == 20) launch_rows<RwCfg<CS, CX, CM, CO, P, 20>>(a, s);                      \
            else if (S == 14) launch_rows<RwCfg<CS, CX, CM, CO, P, 14>>(a, s);                      \
            else launch_rows<RwCfg<CS, CX, CM, CO, P, 8>>(a, s);                                    \
        }                                                                                          \
        return true;                                                                               \
    }
    // (shape, workgroups resident per CU: 58 / 109 / 199 / 205 VGPRs, 13 / 37 / 66 / 87 KB of LDS)
    OCRS_RW(8, 0, 16, 16, true, 8)
    // (the 32-channel encoder blocks were measured too: 38 / 28 us against 35 / 22 for the tiled blocks — they stay tiled;
    // so was the 8-channel decoder block at full resolution: 140 us against 117 for the wave kernel of kernels_det_stream.hip —
    // with 2-4 channels per wave a row step is all fixed cost: barriers, tape, masks, LDS round trips)
    OCRS_RW(16, 32, 16, 16, false, 4)
    OCRS_RW(32, 32, 32, 32, false, 2)
    OCRS_RW(32, 64, 32, 32, false, 1)
#undef OCRS_RW
    return false;
}

}  // namespace k
}  // namespace ocrs
