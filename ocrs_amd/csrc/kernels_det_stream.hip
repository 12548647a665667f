// Round 4: the DoubleConv blocks of the detection U-Net's full-resolution levels as ROW-STREAMING kernels — no LDS, no
// barriers (TextDetector's Model::run, ocrs/src/detection.rs:184; block structure as in kernels_det.hip).
//
// Why: the LDS-tiled blocks of kernels_det.hip spend ~10 000 clocks per 8x32 tile for ~2 000 clocks of arithmetic — six
// workgroup barriers per tile, phases that occupy a quarter to two thirds of the threads, 69 % halo recomputation, every
// operand through LDS.  Here a WAVE owns a strip of 64 image columns (60 of them produce output; two
// columns of halo on each side are recomputed) and walks down S + 4 input rows of it; a LANE owns one column and keeps every
// channel of its pixel in registers:
//   * depthwise 3x3: the left / right neighbours of a value are the neighbouring lanes' registers (DPP wave_shr:1 /
//     wave_shl:1, one VALU op each, no LDS); vertically the three output rows a freshly arrived input row contributes to
//     are three accumulator sets that rotate — the arriving row is tap row ky = 0 of output row i + 1 (accumulator
//     initialised with the bias first), ky = 1 of row i, ky = 2 of row i - 1, which is then complete.  Per output the
//     operations are bias, then the nine taps in (ky, kx) order: the numeric spec's chain, bit for bit.
//   * pointwise 1x1, ConvTranspose 2x2/s2, final 1x1: per-pixel chains over the lane's registers with wave-uniform
//     weights as SGPR operands, pairs of output channels as v_pk_fma_f32 (measured on gfx950: plain v_fma_f32 69 TFLOP/s,
//     v_pk_fma_f32 119 — tools/micro/valu_probe.hip).  The ConvTranspose's weights depend on the column's parity, which
//     alternates across lanes: both parities are computed and the lane selects (the price of keeping one column per
//     lane: 128 of the decoder block's 672 FMAs per pixel).
//   * THE WEIGHT TAPE.  A row step uses every weight of the block exactly once (~700 floats for the level-0 decoder
//     block), far more than the ~100 SGPRs of a wave: the weights have to stream through the SGPRs once per row.  Left to
//     the compiler this fails (it gathers the s_loads at the top of the step, or hoists them out of the row loop, and
//     spills 400-600 SGPRs into VGPR lanes — one v_readlane per weight use).  So the host lays the block's weights out as a
//     TAPE in exactly the order the row step consumes them (double_conv_stream_tape below; one tape per row parity for a
//     decoder block), and the kernel streams it through two 32-float SGPR buffers with explicit s_load_dwordx16 pairs:
//     entering stage k waits for its buffer (lgkmcnt(0): scalar loads return out of order) and then issues stage k + 1
//     into the buffer stage k - 1 has just finished with, so one stage's loads fly under the previous stage's FMAs.
//   * the second conv pair streams the same way on the rows of the first pair's output (two rows later), so one input row
//     in gives one output row out, four rows behind; 2x2 max-pool: the previous output row stays in registers, the
//     horizontal partner is lane ^ 1 (DPP quad_perm).
// Every input element is read once per strip (+ 6.7 % columns, + 4 rows per S), nothing intermediate leaves the
// registers, stores are whole pixels (COUT * 4 bytes per lane, consecutive lanes consecutive pixels).
//
// NUMERIC SPEC, identical to kernels_det.hip / kernels_nn.hip / the oracle:
//   dw:    acc = bias; for (ky,kx) ascending: acc = fmaf(x, w, acc), out-of-image taps contribute fmaf(0, w, acc)
//   pw:    acc = bias; for ci ascending: acc = fmaf(x[ci], W[ci][co], acc)
//   convT: acc = bias; for ci ascending: acc = fmaf(x1[ci], W[dy][dx][ci][co], acc)
//   relu v > 0 ? v : 0;  max-pool m = v > m ? v : m in (ky,kx) order;  sigmoid = spec_sigmoidf.
// (v_pk_fma_f32 is two independent IEEE fused multiply-adds: the same bits as two fmaf.)
#include <type_traits>
#include <vector>

#include "common.hpp"
#include "det_stream.hpp"
#include "kernels.hpp"
#include "spec_math.hpp"

namespace ocrs {
namespace k {

namespace {

using namespace dstream;

// ---- the block's shape and the layout of its weight tape (pairs of floats; positions in floats, always even)
template <int CS_, int CX_, int CMID_, int COUT_, bool POOL_, bool FINAL_, int S_>
struct StCfg {
    static constexpr int CS = CS_, CX = CX_, CMID = CMID_, COUT = COUT_, S = S_;
    static constexpr bool POOL = POOL_, FINAL = FINAL_, DEC = CX_ > 0;
    static constexpr int CU = DEC ? CS_ : 0;
    static constexpr int CIN = CS + CU;
    static constexpr int CSV = CS >= 4 ? CS / 4 : 1, CXV = DEC ? CX / 4 : 1;
    // tape sections (floats): ConvT [bias pairs][ci][q][parity-0 pair, parity-1 pair]; dw [channel pair][bias, 9 taps];
    // pw [bias pairs][ci][co pairs]; final [bf, 0][wf pairs].  A lone channel (CIN == 1) is a pair with a zero partner.
    static constexpr int P_CT = 0;
    static constexpr int N_CT = DEC ? 2 * (CU / 2 + CX * CU) : 0;
    static constexpr int P_DW1 = P_CT + N_CT;
    static constexpr int N_DW1 = 2 * 10 * ((CIN + 1) / 2);
    static constexpr int P_PW1 = P_DW1 + N_DW1;
    static constexpr int N_PW1 = 2 * (CMID / 2 + CIN * (CMID / 2));
    static constexpr int P_DW2 = P_PW1 + N_PW1;
    static constexpr int N_DW2 = 2 * 10 * (CMID / 2);
    static constexpr int P_PW2 = P_DW2 + N_DW2;
    static constexpr int N_PW2 = 2 * (COUT / 2 + CMID * (COUT / 2));
    static constexpr int P_FIN = P_PW2 + N_PW2;
    static constexpr int N_FIN = FINAL ? 2 * (1 + COUT / 2) : 0;
    static constexpr int USED = P_FIN + N_FIN;
    static constexpr int NST = (USED + kStage - 1) / kStage;
    static constexpr int LEN = NST * kStage;            // floats per tape (per row parity)
    static_assert((S + 4) % 3 == 0, "the row loop is unrolled by the three accumulator phases");
    static_assert(S % 2 == 0, "pooling pairs must not straddle two segments");
    static_assert(CS == 1 || CS % 4 == 0, "");
    static_assert(CMID % 2 == 0 && COUT % 2 == 0 && (CIN == 1 || CIN % 2 == 0), "");
};

// One input row as it comes from HBM.
template <class Cfg>
struct RawRow {
    f32x4 s[Cfg::CSV];
    f32x4 x[Cfg::CXV];
};

// Pointwise 1x1: out[co] = bias[co] + sum_ci x[ci] W[ci][co], ci ascending (all CO / 2 pair accumulators side by side).
template <int CI, int CO, int P0, class T>
__device__ __forceinline__ void pw_row(const float (&x)[CI], float (&out)[CO], T& tape, bool relu) {
    f32x2 o[CO / 2];
    static_for<0, CO / 2>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        o[q] = tape.template get2<P0 + 2 * q>();
    });
    static_for<0, CI>([&](auto cc) {
        constexpr int ci = decltype(cc)::value;
        const f32x2 xv = {x[ci], x[ci]};
        static_for<0, CO / 2>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            o[q] = fma2(xv, tape.template get2<P0 + CO + ci * CO + 2 * q>(), o[q]);
        });
    });
#pragma unroll
    for (int q = 0; q < CO / 2; q++) {
        pin(o[q]);
        out[2 * q] = o[q][0];
        out[2 * q + 1] = o[q][1];
    }
    if (relu) {          // (a uniform branch around the selects: as one select of selects every channel's mask sits in its own SGPR pair)
#pragma unroll
        for (int c = 0; c < CO; c++) out[c] = relu1(out[c]);
    }
}


template <class Cfg>
struct Geo {
    int img, col, Y0, h, w, lane;
    bool col_ok;
    int pyo, pxo, ux, h1, w1;    // decoder
    bool up_col_ok;
    // this image's planes as buffer resources (hardware range check: out-of-image taps load 0.0f, no branch, no select)
    __amdgpu_buffer_rsrc_t skip_rs, x1_rs;
    int skip_col_off, x1_col_off;   // byte offset of the lane's column inside a row
};

template <class Cfg>
__device__ __forceinline__ RawRow<Cfg> load_row(const DoubleConvArgs& a, const Geo<Cfg>& g, int i) {
    RawRow<Cfg> r;
    {
        const int voff = ((unsigned)i < (unsigned)g.h && g.col_ok) ? i * (g.w * Cfg::CS * 4) + g.skip_col_off : kOobOffset;
        if constexpr (Cfg::CS == 1) {
            r.s[0] = f32x4{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g.skip_rs, voff, 0, 0)), 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int q = 0; q < Cfg::CSV; q++) r.s[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g.skip_rs, voff + 16 * q, 0, 0));
        }
    }
    if constexpr (Cfg::DEC) {
        const int uy = i - g.pyo;
        const int voff = ((unsigned)uy < (unsigned)(2 * g.h1) && g.up_col_ok) ? (uy >> 1) * (g.w1 * Cfg::CX * 4) + g.x1_col_off : kOobOffset;
#pragma unroll
        for (int q = 0; q < Cfg::CXV; q++) r.x[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g.x1_rs, voff + 16 * q, 0, 0));
    } else {
        r.x[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return r;
}

template <class Cfg>
struct State {
    float acc1[3][Cfg::CIN];
    float acc2[3][Cfg::CMID];
    float prev[Cfg::POOL ? Cfg::COUT : 1];   // the previous output row, for the pool
    RawRow<Cfg> nxt;
};

// One input row through the block.  PH = t % 3 selects the roles of the accumulator sets.
template <class Cfg, int PH>
__device__ __forceinline__ void row_step(const DoubleConvArgs& a, const Geo<Cfg>& g, State<Cfg>& st, int t) {
    constexpr int CS = Cfg::CS, CX = Cfg::CX, CU = Cfg::CU, CIN = Cfg::CIN, CMID = Cfg::CMID, COUT = Cfg::COUT, S = Cfg::S;
    const int i = g.Y0 - 2 + t;                       // the arriving input row
    const RawRow<Cfg> cur = st.nxt;
    if (t + 1 < S + 4) st.nxt = load_row<Cfg>(a, g, i + 1);
    Tape<Cfg::NST> tape;
    {
        const int par = Cfg::DEC ? ((i - g.pyo) & 1) : 0;                       // the ConvTranspose's row parity picks the tape
        tape.base = (cfp)(uintptr_t)(a.tape + (size_t)par * Cfg::LEN);
    }
    tape.template issue<0>();

    // ---- the block's input pixel: skip channels | ConvTranspose(x1) channels
    float in[CIN];
    if constexpr (CS == 1) {
        in[0] = cur.s[0][0];
    } else {
#pragma unroll
        for (int c = 0; c < CS; c++) in[c] = cur.s[c / 4][c % 4];
    }
    if constexpr (Cfg::DEC) {
        const int uy = i - g.pyo;
        const bool up_ok = (unsigned)uy < (unsigned)(2 * g.h1) && g.up_col_ok;
        float x1[CX];
#pragma unroll
        for (int c = 0; c < CX; c++) x1[c] = cur.x[c / 4][c % 4];
        const bool odd = (g.ux & 1) != 0;
        f32x2 e[CU / 2], o[CU / 2];
        static_for<0, CU / 2>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            e[q] = o[q] = tape.template get2<Cfg::P_CT + 2 * q>();
        });
        static_for<0, CX>([&](auto cc) {
            constexpr int ci = decltype(cc)::value;
            const f32x2 xv = {x1[ci], x1[ci]};
            static_for<0, CU / 2>([&](auto qc) {
                constexpr int q = decltype(qc)::value, P = Cfg::P_CT + CU + (ci * (CU / 2) + q) * 4;
                e[q] = fma2(xv, tape.template get2<P>(), e[q]);
                o[q] = fma2(xv, tape.template get2<P + 2>(), o[q]);
            });
        });
#pragma unroll
        for (int q = 0; q < CU / 2; q++) {
            pin(e[q]); pin(o[q]);
            const f32x2 v = odd ? o[q] : e[q];
            in[CS + 2 * q] = up_ok ? v[0] : 0.f;
            in[CS + 2 * q + 1] = up_ok ? v[1] : 0.f;
        }
    }

    // ---- conv pair 1: the arriving row completes depthwise row i - 1
    constexpr int NEW = (PH + 1) % 3, MID = PH, OLD = (PH + 2) % 3;
    dw_row<CIN, NEW, MID, OLD, Cfg::P_DW1>(in, st.acc1, tape);
    const int r1 = i - 1;
    float mid[CMID];
    {
        float d1[CIN];
#pragma unroll
        for (int c = 0; c < CIN; c++) d1[c] = st.acc1[OLD][c];
        if (a.relu_d1) {
#pragma unroll
            for (int c = 0; c < CIN; c++) d1[c] = relu1(d1[c]);
        }
        pw_row<CIN, CMID, Cfg::P_PW1>(d1, mid, tape, a.relu_p1 != 0);
        const bool mid_ok = t >= 2 && (unsigned)r1 < (unsigned)g.h && g.col_ok;    // outside the image: the next depthwise conv's zero padding
#pragma unroll
        for (int c = 0; c < CMID; c++) mid[c] = mid_ok ? mid[c] : 0.f;
    }

    // ---- conv pair 2 on the rows of `mid` (phase of row r1 = phase of t - 2 = (PH + 1) % 3): completes output row r1 - 1
    constexpr int PH2 = (PH + 1) % 3;
    constexpr int NEW2 = (PH2 + 1) % 3, MID2 = PH2, OLD2 = (PH2 + 2) % 3;
    dw_row<CMID, NEW2, MID2, OLD2, Cfg::P_DW2>(mid, st.acc2, tape);
    const int r2 = i - 2;                              // = Y0 + t - 4
    float out[COUT];
    {
        float d2[CMID];
#pragma unroll
        for (int c = 0; c < CMID; c++) d2[c] = st.acc2[OLD2][c];
        if (a.relu_d2) {
#pragma unroll
            for (int c = 0; c < CMID; c++) d2[c] = relu1(d2[c]);
        }
        pw_row<CMID, COUT, Cfg::P_PW2>(d2, out, tape, a.relu_p2 != 0);
    }
    const bool store_ok = t >= 4 && r2 < g.h && g.col_ok && g.lane >= 2 && g.lane < 2 + kValid;
    if constexpr (Cfg::FINAL) {
        float f = tape.template get2<Cfg::P_FIN>()[0];
        static_for<0, COUT / 2>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const f32x2 wv = tape.template get2<Cfg::P_FIN + 2 + 2 * q>();
            f = fmaf(out[2 * q], wv[0], f);
            f = fmaf(out[2 * q + 1], wv[1], f);
        });
        pin(f);
        if (store_ok) a.y[((int64_t)g.img * g.h + r2) * g.w + g.col] = a.sigmoid ? spec_sigmoidf(f) : f;
    } else {
        if (store_ok) {
            float* __restrict__ yp = a.y + (((int64_t)g.img * g.h + r2) * g.w + g.col) * COUT;
#pragma unroll
            for (int q = 0; q < COUT / 4; q++) *reinterpret_cast<f32x4*>(yp + 4 * q) = f32x4{out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]};
        }
        if constexpr (Cfg::POOL) {
            if ((t & 1) == 0) {                        // Y0 is even: an even output row, the upper half of a pooling pair
#pragma unroll
                for (int c = 0; c < COUT; c++) st.prev[c] = out[c];
            } else {
                const int ph = g.h / 2, pw = g.w / 2;
                const int py = r2 >> 1, px = g.col >> 1;
                float m[COUT];
#pragma unroll
                for (int c = 0; c < COUT; c++) {       // (ky,kx) order; the lane holds kx = 0 if its column is even
                    float v = st.prev[c];
                    const float b = lane_pair(st.prev[c]);
                    v = b > v ? b : v;
                    v = out[c] > v ? out[c] : v;
                    const float d = lane_pair(out[c]);
                    v = d > v ? d : v;
                    m[c] = v;
                }
                if (store_ok && (g.lane & 1) == 0 && py < ph && px < pw) {
                    float* __restrict__ pp = a.ypool + (((int64_t)g.img * ph + py) * pw + px) * COUT;
#pragma unroll
                    for (int q = 0; q < COUT / 4; q++) *reinterpret_cast<f32x4*>(pp + 4 * q) = f32x4{m[4 * q], m[4 * q + 1], m[4 * q + 2], m[4 * q + 3]};
                }
            }
        }
    }
}

template <class Cfg, int WPB>
__global__ void __launch_bounds__(64 * WPB) stream_block_kernel(DoubleConvArgs a) {
    Geo<Cfg> g;
    g.lane = threadIdx.x & 63;
    const int wid = blockIdx.x * WPB + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: rows, segments and tape addresses are scalars
    const int strips = a.tiles_x, segs = a.tiles_y;
    if (wid >= a.n * strips * segs) return;
    g.img = wid / (strips * segs);
    const int rem = wid - g.img * (strips * segs);
    const int seg = rem / strips, strip = rem - seg * strips;
    g.col = strip * kValid + g.lane - 2;
    g.Y0 = seg * Cfg::S;
    g.h = a.h; g.w = a.w;
    g.col_ok = (unsigned)g.col < (unsigned)a.w;
    g.h1 = a.h1; g.w1 = a.w1;
    g.pyo = Cfg::DEC ? (a.h - 2 * a.h1) / 2 : 0;
    g.pxo = Cfg::DEC ? (a.w - 2 * a.w1) / 2 : 0;
    g.ux = g.col - g.pxo;
    g.up_col_ok = Cfg::DEC && g.col_ok && (unsigned)g.ux < (unsigned)(2 * a.w1);
    g.skip_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.skip) + (int64_t)g.img * a.h * a.w * Cfg::CS, 0, a.h * a.w * Cfg::CS * 4, 0x00020000);
    g.skip_col_off = g.col * Cfg::CS * 4;
    if constexpr (Cfg::DEC) {
        g.x1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x1) + (int64_t)g.img * a.h1 * a.w1 * Cfg::CX, 0, a.h1 * a.w1 * Cfg::CX * 4, 0x00020000);
        g.x1_col_off = (g.ux >> 1) * Cfg::CX * 4;
    } else {
        g.x1_rs = g.skip_rs;
        g.x1_col_off = 0;
    }

    State<Cfg> st;
#pragma unroll
    for (int p = 0; p < 3; p++) {
#pragma unroll
        for (int c = 0; c < Cfg::CIN; c++) st.acc1[p][c] = 0.f;
#pragma unroll
        for (int c = 0; c < Cfg::CMID; c++) st.acc2[p][c] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < (Cfg::POOL ? Cfg::COUT : 1); c++) st.prev[c] = 0.f;
    st.nxt = load_row<Cfg>(a, g, g.Y0 - 2);
    for (int t = 0; t < Cfg::S + 4; t += 3) {
        if (g.Y0 + t - 4 >= g.h) break;               // nothing below the image
        row_step<Cfg, 0>(a, g, st, t);
        row_step<Cfg, 1>(a, g, st, t + 1);
        row_step<Cfg, 2>(a, g, st, t + 2);
    }
}

template <class Cfg>
void launch_stream(const DoubleConvArgs& a0, hipStream_t s) {
    constexpr int WPB = 4;
    DoubleConvArgs a = a0;
    if (!a.tape || a.tape_len != Cfg::LEN) fail(OCRS_ERR_RUN_FAILED, "streaming DoubleConv block without its weight tape");
    if ((int64_t)a.h * a.w * Cfg::CS * 4 >= kOobOffset || (int64_t)a.h1 * a.w1 * Cfg::CX * 4 >= kOobOffset) fail(OCRS_ERR_CAPACITY, "detection input too large for the streaming blocks");
    a.tiles_x = (a.w + kValid - 1) / kValid;          // strips
    a.tiles_y = (a.h + Cfg::S - 1) / Cfg::S;          // row segments
    const int waves = a.n * a.tiles_x * a.tiles_y;
    hipLaunchKernelGGL((stream_block_kernel<Cfg, WPB>), dim3((waves + WPB - 1) / WPB), dim3(64 * WPB), 0, s, a);
}

// The tape of one block: the weights in the order row_step reads them (host pointers in, floats out).
template <class Cfg>
std::vector<float> build_tape(const StreamWeights& w) {
    constexpr int CX = Cfg::CX, CU = Cfg::CU, CIN = Cfg::CIN, CMID = Cfg::CMID, COUT = Cfg::COUT;
    const int n_par = Cfg::DEC ? 2 : 1;
    std::vector<float> tape((size_t)n_par * Cfg::LEN, 0.f);
    for (int par = 0; par < n_par; par++) {
        float* t = tape.data() + (size_t)par * Cfg::LEN;
        if constexpr (Cfg::DEC) {
            for (int co = 0; co < CU; co++) t[Cfg::P_CT + co] = w.bt[co];
            for (int ci = 0; ci < CX; ci++)
                for (int q = 0; q < CU / 2; q++)
                    for (int xp = 0; xp < 2; xp++)
                        for (int e = 0; e < 2; e++)
                            t[Cfg::P_CT + CU + (ci * (CU / 2) + q) * 4 + xp * 2 + e] = w.wt[((size_t)(par * 2 + xp) * CX + ci) * CU + 2 * q + e];
        }
        auto dw = [&](int P0, int C, const float* wd, const float* bd) {
            for (int q = 0; q < (C + 1) / 2; q++)
                for (int e = 0; e < 2; e++) {
                    const int c = 2 * q + e;
                    if (c >= C) continue;
                    t[P0 + 20 * q + e] = bd[c];
                    for (int tap = 0; tap < 9; tap++) t[P0 + 20 * q + 2 * (1 + tap) + e] = wd[tap * C + c];
                }
        };
        auto pw = [&](int P0, int CI, int CO, const float* wp, const float* bp) {
            for (int co = 0; co < CO; co++) t[P0 + co] = bp[co];
            for (int ci = 0; ci < CI; ci++)
                for (int co = 0; co < CO; co++) t[P0 + CO + ci * CO + co] = wp[ci * CO + co];
        };
        dw(Cfg::P_DW1, CIN, w.wd1, w.bd1);
        pw(Cfg::P_PW1, CIN, CMID, w.wp1, w.bp1);
        dw(Cfg::P_DW2, CMID, w.wd2, w.bd2);
        pw(Cfg::P_PW2, CMID, COUT, w.wp2, w.bp2);
        if constexpr (Cfg::FINAL) {
            t[Cfg::P_FIN] = w.bf[0];
            for (int c = 0; c < COUT; c++) t[Cfg::P_FIN + 2 + c] = w.wf[c];
        }
    }
    return tape;
}

}  // namespace

// Shapes with a streaming kernel (option "det_stream"); same contract as double_conv_fused.  With `tape_out` the block's
// weight tape is built from the host weights in `hw` (the caller uploads it and passes it in DoubleConvArgs::tape).
//
// Segment height S: a wave's run time is proportional to S + 4 rows, the redundant work to (S + 4) / S, and the launch
// wants at least a wave per SIMD (1 024 on the MI355X) and at most one round of resident waves: measured per 8 pages of
// 800 x 600 at S = 8 / 14 / 20 / 26 / 32 / 44: decoder block (147 VGPRs, 3 waves per SIMD) 129 / 128 / 133 / 127 / 116 /
// 142 us, encoder block 44 / 44 / 48 / 53 / 48 / 60 us.  So: decoder 32 rows where that still gives 1 024 waves (8 pages:
// 2 000), else 14, else 8 (one page at S = 8: 1 000 waves of 12 rows instead of 250 of 36); encoder 14, else 8.  The tape
// does not depend on S.
bool double_conv_stream(const DoubleConvArgs& a, int cs, int cx, int cmid, int cout, bool pool, bool final_conv, bool launch, hipStream_t s,
                        const StreamWeights* hw, std::vector<float>* tape_out, int* tape_len) {
    auto waves = [&](int S) { return (int64_t)a.n * ((a.w + kValid - 1) / kValid) * ((a.h + S - 1) / S); };
    const int opt = option(OPT_DET_STREAM);   // 1: by the rule above; 8 / 14 / 32: that segment height (tests, A/B)
    // (the encoder block is light — 66 VGPRs, seven waves per SIMD — and runs 4 640 waves of 18 rows in one round: 14 rows)
    const int64_t simds = launch ? 4 * (int64_t)ctx().cu_count() : 1024;   // (1 024 on the MI355X)
    const int S = (opt == 8 || opt == 14 || opt == 32) ? opt : (cx > 0 && waves(32) >= simds) ? 32 : waves(14) >= simds ? 14 : 8;
#define OCRS_ST(CS, CX, CM, CO, P, F)                                                             \
    if (cs == CS && cx == CX && cmid == CM && cout == CO && pool == P && final_conv == F) {        \
        typedef StCfg<CS, CX, CM, CO, P, F, 32> Cfg;                                               \
        if (tape_out) *tape_out = build_tape<Cfg>(*hw);                                            \
        if (tape_len) *tape_len = Cfg::LEN;                                                        \
        if (launch) {                                                                              \
            if (S == 32) launch_stream<Cfg>(a, s);                                                 \
            else if (S == 14) launch_stream<StCfg<CS, CX, CM, CO, P, F, 14>>(a, s);                 \
            else launch_stream<StCfg<CS, CX, CM, CO, P, F, 8>>(a, s);                               \
        }                                                                                          \
        return true;                                                                               \
    }
    // the full-resolution level of the U-Net (8 channels): depthwise-heavy blocks, where one lane per pixel pays.  The
    // 16-channel encoder block was measured too (53-64 us per 8 pages against 47 for the LDS-tiled MFMA block: its
    // pointwise convs dominate and the 400 x 300 level yields too few waves) and stays with kernels_det.hip.
    OCRS_ST(1, 0, 8, 8, true, false)
    OCRS_ST(8, 16, 8, 8, false, true)
    OCRS_ST(8, 16, 8, 8, false, false)
#undef OCRS_ST
    return false;
}

}  // namespace k
}  // namespace ocrs
