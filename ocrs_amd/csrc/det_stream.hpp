// Shared pieces of the row-streaming detection kernels (kernels_det_stream.hip: a wave per strip, everything in registers;
// kernels_det_rows.hip: a workgroup per strip, channels split over its waves, pointwise convs on the matrix cores):
// DPP lane shifts, packed FMAs, the SGPR weight tape and the rotating-accumulator depthwise row step.  See
// kernels_det_stream.hip for the design notes.
#pragma once
#include <type_traits>

#include "common.hpp"

namespace ocrs {
namespace k {
namespace dstream {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kValid = 60;    // output columns per 64-lane strip
constexpr int kOobOffset = 0x7f000000;   // a byte offset past any image: the buffer load's range check returns 0.0f for it
constexpr int kStage = 16;    // floats per tape stage (16: one s_load_dwordx16, 32: two — measured equal, and 32 spills SGPRs)

typedef const float __attribute__((address_space(4)))* cfp;

__device__ __forceinline__ float lane_left(float v) {    // the value of lane - 1 (0 at lane 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_right(float v) {   // the value of lane + 1 (0 at lane 63)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_pair(float v) {    // the value of lane ^ 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ f32x2 fma2(f32x2 x, f32x2 w, f32x2 acc) { return __builtin_elementwise_fma(x, w, acc); }
// relu: v > 0 ? v : 0.  v_max_f32(v, +0) is that for every input except a signalling NaN (the instruction orders -0 < +0
// and returns the non-NaN operand); written as asm because the compiler's own lowering of the select puts a canonicalising
// v_max in front (it cannot see that the input is an FMA result).
__device__ __forceinline__ float relu1(float v) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}
// Pins a value's computation to this point of the program.  The accumulators a row step leaves for the NEXT step have no
// reader in this one, and the compiler otherwise sinks their FMAs towards that reader — past the tape's stage changes,
// which keeps every stage's 16 SGPRs alive (spilled to VGPR lanes, one v_readlane per use) until the FMAs finally run.
__device__ __forceinline__ void pin(f32x2& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// The tape reader: two SGPR buffers of kStage floats (16-float halves, one s_load_dwordx16 each); stage k lives in buffer k % 2.
template <int NST, int STG = kStage>
struct Tape {
    static constexpr int H = STG / 16;      // halves per buffer
    cfp base;
    f32x16 a[H], b[H];
    template <int ST>
    __device__ __forceinline__ void issue() {
        static_for<0, H>([&](auto hc) {
            constexpr int hh = decltype(hc)::value;
            if constexpr (ST % 2 == 0) asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(a[hh]) : "s"(base), "i"((ST * STG + 16 * hh) * 4));
            else asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(b[hh]) : "s"(base), "i"((ST * STG + 16 * hh) * 4));
        });
    }
    // first touch of stage ST: its loads (and every other outstanding scalar load) land, then the next stage takes off
    template <int ST>
    __device__ __forceinline__ void enter() {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (H == 1) {
            if constexpr (ST % 2 == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a[0]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b[0]));
        } else {
            if constexpr (ST % 2 == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a[0]), "+s"(a[1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(b[0]), "+s"(b[1]));
        }
        if constexpr (ST + 1 < NST) issue<ST + 1>();
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int POS>
    __device__ __forceinline__ f32x2 get2() {
        static_assert(POS % 2 == 0, "");
        constexpr int ST = POS / STG, E = POS % STG;
        if constexpr (E == 0) enter<ST>();
        if constexpr (ST % 2 == 0) return f32x2{a[E / 16][E % 16], a[E / 16][E % 16 + 1]};
        else return f32x2{b[E / 16][E % 16], b[E / 16][E % 16 + 1]};
    }
};

// Depthwise 3x3, one arriving row: `in` [C] -> the three accumulator sets; afterwards acc[OLD] holds the finished row.
template <int C, int NEW, int MID, int OLD, int P0, class T>
__device__ __forceinline__ void dw_row(const float (&in)[C], float (&acc)[3][C], T& tape) {
    if constexpr (C == 1) {
        const float l = lane_left(in[0]), r = lane_right(in[0]);
        float n = tape.template get2<P0>()[0];
        n = fmaf(l, tape.template get2<P0 + 2>()[0], n); n = fmaf(in[0], tape.template get2<P0 + 4>()[0], n); n = fmaf(r, tape.template get2<P0 + 6>()[0], n);
        pin(n);
        acc[NEW][0] = n;
        float m = acc[MID][0];
        m = fmaf(l, tape.template get2<P0 + 8>()[0], m); m = fmaf(in[0], tape.template get2<P0 + 10>()[0], m); m = fmaf(r, tape.template get2<P0 + 12>()[0], m);
        pin(m);
        acc[MID][0] = m;
        float o = acc[OLD][0];
        o = fmaf(l, tape.template get2<P0 + 14>()[0], o); o = fmaf(in[0], tape.template get2<P0 + 16>()[0], o); o = fmaf(r, tape.template get2<P0 + 18>()[0], o);
        pin(o);
        acc[OLD][0] = o;
    } else {
        static_for<0, C / 2>([&](auto qc) {
            constexpr int q = decltype(qc)::value, c = 2 * q, P = P0 + 20 * q;
            const f32x2 v = {in[c], in[c + 1]};
            const f32x2 l = {lane_left(in[c]), lane_left(in[c + 1])};
            const f32x2 r = {lane_right(in[c]), lane_right(in[c + 1])};
            f32x2 n = tape.template get2<P>();
            n = fma2(l, tape.template get2<P + 2>(), n);
            n = fma2(v, tape.template get2<P + 4>(), n);
            n = fma2(r, tape.template get2<P + 6>(), n);
            pin(n);
            acc[NEW][c] = n[0]; acc[NEW][c + 1] = n[1];
            f32x2 m = {acc[MID][c], acc[MID][c + 1]};
            m = fma2(l, tape.template get2<P + 8>(), m);
            m = fma2(v, tape.template get2<P + 10>(), m);
            m = fma2(r, tape.template get2<P + 12>(), m);
            pin(m);
            acc[MID][c] = m[0]; acc[MID][c + 1] = m[1];
            f32x2 o = {acc[OLD][c], acc[OLD][c + 1]};
            o = fma2(l, tape.template get2<P + 14>(), o);
            o = fma2(v, tape.template get2<P + 16>(), o);
            o = fma2(r, tape.template get2<P + 18>(), o);
            pin(o);
            acc[OLD][c] = o[0]; acc[OLD][c + 1] = o[1];
        });
    }
}

}  // namespace dstream
}  // namespace k
}  // namespace ocrs
