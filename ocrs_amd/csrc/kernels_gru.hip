// Persistent bidirectional-GRU recurrence on the packed ("ragged") sequence batch
// (the sequential part of TextRecognizer::run, ocrs/src/recognition.rs:341-360).
//
// ONE launch per GRU layer replaces the chain of Tmax dependent launches of gru_step_fused
// (kernels_nn.hip).  Numerics are unchanged: per output
//     gh = bh;  gh = fmaf(h[k], Wh[k][j], gh)  for k ascending  (v_mfma_f32_16x16x4_f32 chain)
//     r = sigma(gx_r + gh_r), z = sigma(gx_z + gh_z), n = tanh(fmaf(r, gh_n, gx_n)), h' = fmaf(z, h - n, n).
//
// Decomposition.  Rows (text lines, sorted by sequence length descending) are cut into 16-row
// tiles; the H hidden units into UB = H/16 slices.  A workgroup = 4 waves that share ONE 16-unit
// slice of Wh (H x 48 floats, staged into LDS once for all T steps) and serve up to 4 row tiles per wave (which
// ones: gru_assign_tiles, a longest-first deal that balances the waves' run times).  The UB workgroups that own the
// slices of the same rows form a "cluster"; the only data they exchange is the new hidden state of their rows.
//
// Exchange = the layer's own output, and the data is its own flag.  y[row(t, m)][dir*H + unit] has to be
// written anyway; the host pre-fills y with the word 0xFFFFFFFF (a NaN no result can be: the epilogue maps that one
// bit pattern to the canonical NaN).  A wave writes its 16 rows x 16 units with 16-byte WRITE-THROUGH stores and
// moves on — no drain, no counter.  The UB waves that need the tile's full state for the next step read the y rows
// of the previous step with cache-bypassing loads and look at every 32-bit word: any 0xFFFFFFFF left means "not yet
// written", and the wave re-reads (MI355X_MICROARCH.md "inter-workgroup visibility", form R2 — the payload is the
// flag — checked per 32-bit word, so no assumption about the atomicity of wider accesses is made).  The loads of
// the next item are issued before the current item is computed whenever it belongs to another tile, so that by the
// time they are checked the round trip is long over.  No grid barrier: tiles never wait for each other, and nothing
// depends on workgroup placement or order (blocks of a cluster are merely steered to one XCD for speed).  Every
// wait is bounded: on a time-out the kernel raises the error word and returns, the host reports OCRS_ERR_DEVICE.
//
// MFMA roles.  D = A.B with A = Wh^T (16 units x 4 k, from LDS) and B = h^T (4 k x 16 rows, from
// registers), so a lane ends up with 4 CONSECUTIVE units of one row: the epilogue's gx reads and
// the y store are one 16-byte access per gate / per lane in the natural layouts.  The B operand
// wants lane (row, kq) to hold h[row][4*s + kq]; rows are fetched as 16-byte pieces and turned by a
// 4x4 transpose across the four 16-lane groups (v_permlane16_swap / v_permlane32_swap).
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"
#include "spec_math.hpp"

namespace ocrs {
namespace k {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kMaxSlots = 128;  // waves per direction (4 per cluster)

struct GruParams {
    const float* gx;     // [2][R][3H] input projections (+ bi), natural column order (r | z | n)
    const float* wh;     // [2][H][3H]
    const float* bh;     // [2][3H]
    float* y;            // [R][2H]
    const int32_t* Tm;   // [M] sequence length of line m (descending)
    const int32_t* off;  // [Tmax + 1] first packed row of time t
    uint32_t* sync;      // [1] error word; zeroed before the launch
    uint32_t* place;     // [grid] XCD id + 1 of every workgroup, written by the kernel; zeroed before the launch
    int64_t R;
    int M, ncl, Tmax;
    int prio;            // s_setprio level of the waves (0..3)
    int allow_local;     // 0: write-through hand-offs whatever the placement (OCRS_GRU_LOCAL=0)
    int scatter;         // 1: clusters deliberately spread over the XCDs (OCRS_GRU_SCATTER=1; tests the census)
    int16_t tiles[kMaxSlots * 4];  // row tiles of wave slot (cluster-in-direction * 4 + wave), longest first; -1 = none
    uint32_t spin_limit;
    int lazy;            // background kernel: wait this many x 1024 clocks before every re-read (its waits are whole rounds long)
};

// Hand-off accesses to y: 16-byte raw-buffer loads/stores with the sc1 (agent-scope) cache bit — the store is
// written through to memory, the load bypasses the CU's L1 (MI355X_MICROARCH.md: "`sc1` loads may replace the
// acquire when the producer stored `sc1`").  Buffer intrinsics rather than `volatile` accesses: the compiler
// follows a volatile access with s_waitcnt vmcnt(0), which would serialise the 17 loads of an item.  The buffer
// resource spans y (< 4 GiB, checked by gru_plan); offsets are bytes.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kAuxSc1 = 16;
__device__ __forceinline__ f32x4 load_bypass(__amdgpu_buffer_rsrc_t y, uint32_t byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(y, (int)byte_off, 0, kAuxSc1));
}
__device__ __forceinline__ void store_through(__amdgpu_buffer_rsrc_t y, uint32_t byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y, (int)byte_off, 0, kAuxSc1);
}
// Plain store: the line stays in the XCD's L2, where the bypassing loads of the SAME XCD find it (an L2 hit instead
// of a trip over the fabric: 1.0-1.2 us less per step).  Invisible to other XCDs until the kernel ends, so only a
// cluster whose workgroups have all found themselves on one XCD uses it (cluster_is_local below).
__device__ __forceinline__ void store_local(__amdgpu_buffer_rsrc_t y, uint32_t byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y, (int)byte_off, 0, 0);
}
typedef __attribute__((address_space(1))) uint32_t gu32;

// 4x4 transpose between (register e, 16-lane group g):  out[a] in group g  =  in[g] of group a
__device__ __forceinline__ void transpose4(const f32x4& in, float* out) {
    const unsigned v0 = __float_as_uint(in[0]), v1 = __float_as_uint(in[1]);
    const unsigned v2 = __float_as_uint(in[2]), v3 = __float_as_uint(in[3]);
    // permlane16_swap(a, b): a's odd 16-lane rows <-> b's even rows
    const u32x2 p01 = __builtin_amdgcn_permlane16_swap(v0, v1, false, false);
    const u32x2 p23 = __builtin_amdgcn_permlane16_swap(v2, v3, false, false);
    // permlane32_swap(a, b): a's rows 2,3 <-> b's rows 0,1
    const u32x2 q02 = __builtin_amdgcn_permlane32_swap(p01[0], p23[0], false, false);
    const u32x2 q13 = __builtin_amdgcn_permlane32_swap(p01[1], p23[1], false, false);
    out[0] = __uint_as_float(q02[0]);
    out[1] = __uint_as_float(q13[0]);
    out[2] = __uint_as_float(q02[1]);
    out[3] = __uint_as_float(q13[1]);
}

constexpr unsigned kUnwritten = 0xFFFFFFFFu;  // what gru_persistent() fills y with

template <int H>
struct Loaded {             // everything one (tile, step) item reads from memory
    f32x4 h[H / 16];        // lane (row, kq): pieces q = 4j + kq of the row's previous state
    f32x4 hp;               // previous state of this lane's own 4 units
    f32x4 gr, gz, gn;       // gx of this lane's 4 units
    uint32_t out_off;       // byte offset in y of this lane's 4 output units
    uint32_t prev_off;      // byte offset in y of the row's previous-step output (this direction's half)
    bool has_prev;          // false: h = 0 (first step / idle lane)
    bool active;
};

// row bookkeeping + the gx operands of the epilogue (independent of the recurrence).  tm = length of the lane's
// row (0: no such row); off_l = the packed-row table off[] in LDS.  No global load here other than gx: a wait on
// one would also wait for the write-through store of the previous item (vmcnt retires in order).
template <int H>
__device__ __forceinline__ void issue_meta(const GruParams& p, int dir, int ub, int tile, int tm, const int* off_l, int s,
                                           int i16, int kq, Loaded<H>& L) {
    const int m = tile * 16 + i16;
    L.active = tm > s;
    const int t = dir ? tm - 1 - s : s;
    const int64_t row = L.active ? (int64_t)off_l[t] + m : 0;
    L.out_off = (uint32_t)((row * 2 * H + dir * H + ub * 16 + kq * 4) * sizeof(float));
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    L.gr = L.gz = L.gn = zero;   // (the state registers h / hp are issue_state's: zeroing them here would keep 68 registers
                                 // of zeros alive across the MFMA chain when the state loads are issued after it)
    L.has_prev = L.active && s > 0;
    L.prev_off = 0;
    if (L.active) {
        const float* g = p.gx + ((int64_t)dir * p.R + row) * 3 * H + ub * 16 + kq * 4;
        L.gr = *reinterpret_cast<const f32x4*>(g);
        L.gz = *reinterpret_cast<const f32x4*>(g + H);
        L.gn = *reinterpret_cast<const f32x4*>(g + 2 * H);
        if (s > 0) L.prev_off = (uint32_t)((((int64_t)off_l[dir ? tm - s : s - 1] + m) * 2 * H + dir * H) * sizeof(float));
    }
}

// the previous state of the lane's row (no wait: the loads are checked by state_ready())
template <int H>
__device__ __forceinline__ void issue_state(__amdgpu_buffer_rsrc_t y, int ub, int kq, Loaded<H>& L) {
    if (L.has_prev) {
#pragma unroll
        for (int j = 0; j < H / 16; j++) L.h[j] = load_bypass(y, L.prev_off + (16 * j + 4 * kq) * 4);
        L.hp = load_bypass(y, L.prev_off + (ub * 16 + kq * 4) * 4);
    } else {   // first step of the row / idle lane: h = 0
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < H / 16; j++) L.h[j] = zero;
        L.hp = zero;
    }
}

// true when no lane of the wave still sees an unwritten word
template <int H>
__device__ __forceinline__ bool state_ready(const Loaded<H>& L) {
    unsigned mx = 0;
#pragma unroll
    for (int j = 0; j < H / 16; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) mx = max(mx, __float_as_uint(L.h[j][e]));
#pragma unroll
    for (int e = 0; e < 4; e++) mx = max(mx, __float_as_uint(L.hp[e]));
    return !__any(mx == kUnwritten);
}

// wait until the previous state of every row of the tile is complete, re-reading as needed
template <int H>
__device__ __forceinline__ bool await_state(const GruParams& p, __amdgpu_buffer_rsrc_t y, int ub, int kq, Loaded<H>& L) {
    for (uint32_t spins = 0; !state_ready<H>(L); spins++) {
        __builtin_amdgcn_s_sleep(2);
        for (int z = 0; z < p.lazy; z++) __builtin_amdgcn_s_sleep(16);
        // Back off when the wait is a long one (a peer workgroup not resident yet, or held up): a re-read costs 17
        // line fetches per lane group, and a CU whose pollers re-issue them back to back can keep its own memory
        // pipeline so full that the store everybody is waiting for does not get through (seen with two workgroups
        // of gru_gates_kernel on one CU: six pollers, waits of seconds).  The first re-reads stay immediate.
        if (spins >= 4u) {
            const uint32_t n = spins < 36u ? (spins >> 2) : 9u;   // 1 .. 9 x 1024 clocks
            for (uint32_t z = 0; z < n; z++) __builtin_amdgcn_s_sleep(16);
        }
        if ((spins & 255u) == 255u) {
            gu32* err = (gu32*)p.sync;
            const uint32_t e = __builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (e != 0 || spins >= p.spin_limit) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        issue_state<H>(y, ub, kq, L);
    }
    return true;
}

// The three gate accumulators of one item (D layout 16x16: lane holds D[i = 4*kq + r][j = i16] = unit 16ub + 4kq + r
// of row i16).
struct GateAcc { f32x4 r, z, n; };

template <int H>
__device__ __forceinline__ GateAcc mfma_chain(int lane, const float (&w)[H / 4], const float* lds_w, const f32x4& br,
                                              const f32x4& bz, const f32x4& bn) {
    f32x4 acc_r = br, acc_z = bz, acc_n = bn;
    // A operand: lane (unit c = i16, kq) feeds Wh[4*s4 + kq][g*H + 16ub + c].  LDS holds, per (gate, block of 4
    // steps), one 16-byte piece per lane: a conflict-free ds_read_b128 fetches 4 steps of one gate.
    // Reads run one block ahead of the MFMAs that consume them.
    const f32x4* ap = reinterpret_cast<const f32x4*>(lds_w) + lane;
    f32x4 ar = ap[0], az = ap[(H / 16) * 64], an = ap[2 * (H / 16) * 64];
#pragma unroll
    for (int blk = 0; blk < H / 16; blk++) {
        f32x4 nr = ar, nz = az, nn = an;
        if (blk + 1 < H / 16) {
            nr = ap[(blk + 1) * 64];
            nz = ap[((H / 16) + blk + 1) * 64];
            nn = ap[(2 * (H / 16) + blk + 1) * 64];
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float b = w[4 * blk + e];
            acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[e], b, acc_r, 0, 0, 0);
            acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(az[e], b, acc_z, 0, 0, 0);
            acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(an[e], b, acc_n, 0, 0, 0);
        }
        ar = nr; az = nz; an = nn;
    }
    return GateAcc{acc_r, acc_z, acc_n};
}

// spec_expf / spec_sigmoidf / spec_tanhf (spec_math.hpp) without the early return for NaN — the same value through
// a final select — so that the gate arithmetic is straight-line code the scheduler can interleave with MFMAs.
__device__ __forceinline__ float expf_sl(float x0) {
    float x = x0 > 88.0f ? 88.0f : x0;
    x = x < -87.0f ? -87.0f : x;
    const float kf = rintf(x * 1.44269504088896341f);
    float r = fmaf(kf, -0.693145751953125f, x);
    r = fmaf(kf, -1.42860682030941723212e-6f, r);
    float q = 1.98412698412698413e-4f;
    q = fmaf(q, r, 1.38888888888888894e-3f);
    q = fmaf(q, r, 8.33333333333333322e-3f);
    q = fmaf(q, r, 4.16666666666666644e-2f);
    q = fmaf(q, r, 1.66666666666666657e-1f);
    q = fmaf(q, r, 0.5f);
    q = fmaf(q, r, 1.0f);
    q = fmaf(q, r, 1.0f);
    const float v = __int_as_float(__float_as_int(q) + (((int)kf) << 23));
    return x0 != x0 ? x0 : v;
}
#ifndef OCRS_GATE_MATH
#define OCRS_GATE_MATH 0   // timing ablations (tools/ab_gate_math.sh; results are WRONG on purpose for 1..3)
#endif
#if OCRS_GATE_MATH == 0
__device__ __forceinline__ float sigmoidf_sl(float x) { return 1.0f / (1.0f + expf_sl(-x)); }
__device__ __forceinline__ float tanhf_sl(float x) {
    const float t = expf_sl(2.0f * x);
    return (t - 1.0f) / (t + 1.0f);
}
#elif OCRS_GATE_MATH == 1   // hardware v_exp_f32 / v_rcp_f32
__device__ __forceinline__ float sigmoidf_sl(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f)); }
__device__ __forceinline__ float tanhf_sl(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 2.88539008177792682f);
    return (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f);
}
#elif OCRS_GATE_MATH == 2   // candidate re-spec: degree-6 exp, division-free reciprocal (magic seed + two cubic steps)
__device__ __forceinline__ float expf_v2(float x0) {
    const float x = __builtin_amdgcn_fmed3f(x0, -87.0f, 88.0f);
    const float kf = rintf(x * 1.44269504088896341f);
    float r = fmaf(kf, -0.693145751953125f, x);
    r = fmaf(kf, -1.42860682030941723212e-6f, r);
    float q = 1.38888888888888894e-3f;
    q = fmaf(q, r, 8.33333333333333322e-3f);
    q = fmaf(q, r, 4.16666666666666644e-2f);
    q = fmaf(q, r, 1.66666666666666657e-1f);
    q = fmaf(q, r, 0.5f);
    q = fmaf(q, r, 1.0f);
    q = fmaf(q, r, 1.0f);
    return __int_as_float(__float_as_int(q) + (((int)kf) << 23));
}
__device__ __forceinline__ float rcp_v2(float d) {
    float r = __int_as_float(0x7EF311C7 - __float_as_int(d));
    float e = fmaf(-d, r, 1.0f);
    float t = fmaf(e, e, e);
    r = fmaf(r, t, r);
    e = fmaf(-d, r, 1.0f);
    t = fmaf(e, e, e);
    return fmaf(r, t, r);
}
__device__ __forceinline__ float sigmoidf_sl(float x) { return rcp_v2(1.0f + expf_v2(-x)); }
__device__ __forceinline__ float tanhf_sl(float x) {
    const float t = expf_v2(2.0f * x);
    return (t - 1.0f) * rcp_v2(t + 1.0f);
}
#else                       // 3: no transcendental at all (the floor of the item's other work)
__device__ __forceinline__ float sigmoidf_sl(float x) { return x; }
__device__ __forceinline__ float tanhf_sl(float x) { return x; }
#endif

// gates + new state of the lane's 4 units; `store` = this lane's row is live at this step
template <int H>
__device__ __forceinline__ void epilogue_store(__amdgpu_buffer_rsrc_t y, const GateAcc& a, const f32x4& gr, const f32x4& gz,
                                               const f32x4& gn, const f32x4& hp, uint32_t out_off, bool store, bool local) {
    f32x4 hn;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float rg = sigmoidf_sl(gr[r] + a.r[r]);
        const float zg = sigmoidf_sl(gz[r] + a.z[r]);
        const float ng = tanhf_sl(fmaf(rg, a.n[r], gn[r]));
        const float hv = fmaf(zg, hp[r] - ng, ng);
        hn[r] = __float_as_uint(hv) == kUnwritten ? __uint_as_float(0x7FC00000u) : hv;  // keep the flag word free
    }
    // fire and forget (the data is the flag).  Unconditional instruction: a lane with nothing to store aims past
    // the buffer and the hardware range check drops it — a branch here would let the compiler sink the whole gate
    // arithmetic under it, out of the MFMA shadow.
    const uint32_t o = store ? out_off : 0xFFFFFFF0u;
    if (local) store_local(y, o, hn);  // wave-uniform
    else store_through(y, o, hn);
}

// The same arithmetic one unit at a time (the straight-line form above keeps twelve transcendental chains in flight —
// good in the shadow of a wave's own MFMAs, 60 registers too many for the loader wave of gru_teams_kernel).
template <int H>
__device__ __forceinline__ void epilogue_store_lean(__amdgpu_buffer_rsrc_t y, const GateAcc& a, const f32x4& gr, const f32x4& gz,
                                                    const f32x4& gn, const f32x4& hp, uint32_t out_off, bool store, bool local) {
    f32x4 hn;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float rg = sigmoidf_sl(gr[r] + a.r[r]);
        const float zg = sigmoidf_sl(gz[r] + a.z[r]);
        const float ng = tanhf_sl(fmaf(rg, a.n[r], gn[r]));
        const float hv = fmaf(zg, hp[r] - ng, ng);
        hn[r] = __float_as_uint(hv) == kUnwritten ? __uint_as_float(0x7FC00000u) : hv;
        __builtin_amdgcn_sched_barrier(0);
    }
    const uint32_t o = store ? out_off : 0xFFFFFFF0u;
    if (local) store_local(y, o, hn);
    else store_through(y, o, hn);
}

template <int H>
__global__ void __launch_bounds__(256)
gru_persistent_kernel(GruParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds_w[];  // H*48 floats, layout below
    constexpr int UB = H / 16;
    constexpr int WV = 4;
    constexpr bool EARLY = true;
    // blocks of one cluster share blockIdx % 8 (observed: block b runs on XCD b % 8 — speed only)
    const int b = blockIdx.x;
    const int q = b >> 3;
    // (p.scatter, a test knob, deals consecutive blocks to a cluster instead: every cluster then spans all XCDs)
    const int ub = p.scatter ? b % UB : q % UB;
    const int cid = p.scatter ? b / UB : (q / UB) * 8 + (b & 7);
    if (cid >= 2 * p.ncl) return;
    const int dir = cid & 1, cl = cid >> 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    // Placement census, part 1: publish which XCD this workgroup runs on (+1: the host zeroed the table).
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if (tid == 0) __hip_atomic_store((gu32*)p.place + cid * UB + ub, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float* __restrict__ whd = p.wh + (int64_t)dir * H * 3 * H;
    const float* __restrict__ bhd = p.bh + (int64_t)dir * 3 * H;
    const int j0 = ub * 16;
    // the Wh slice [:, g*H + j0 .. +16) for g = r,z,n -> LDS, once: element (k = 16*blk + 4*e + kq, gate g, unit c)
    // sits at float ((g*(H/16) + blk)*64 + kq*16 + c)*4 + e
    for (int i = tid; i < H * 12; i += 64 * WV) {
        const int k = i / 12, qq = i - k * 12;
        const int g = qq >> 2, c4 = (qq & 3) * 4;
        const float4 v = *reinterpret_cast<const float4*>(whd + (int64_t)k * 3 * H + g * H + j0 + c4);
        const int blk = k >> 4, e = (k >> 2) & 3, kk = k & 3;
        float* dst = &lds_w[(((g * (H / 16) + blk) * 64) + kk * 16 + c4) * 4 + e];
        dst[0] = v.x; dst[4] = v.y; dst[8] = v.z; dst[12] = v.w;
    }
    int* off_l = reinterpret_cast<int*>(lds_w + H * 48);  // off[0 .. Tmax]
    for (int i = tid; i <= p.Tmax; i += 64 * WV) off_l[i] = p.off[i];
    const f32x4 br = *reinterpret_cast<const f32x4*>(bhd + j0 + kq * 4);
    const f32x4 bz = *reinterpret_cast<const f32x4*>(bhd + H + j0 + kq * 4);
    const f32x4 bn = *reinterpret_cast<const f32x4*>(bhd + 2 * H + j0 + kq * 4);
    // this wave's tiles: tile_of(0), tile_of(1), ... (at most 4 of them); per tile the length of the lane's row and of the
    // tile's first (= longest) row, both kept in registers for the whole run
    // Which tiles: decided by the host (gru_assign_tiles) from the tiles' lengths, the same for both directions.
    const int slot = cl * WV + wave;
    const int t0 = p.tiles[slot * 4 + 0], t1 = p.tiles[slot * 4 + 1], t2 = p.tiles[slot * 4 + 2], t3 = p.tiles[slot * 4 + 3];
    auto tile_of = [&](int i) { return i == 0 ? t0 : i == 1 ? t1 : i == 2 ? t2 : t3; };
    int tmr[4], tT[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = tile_of(i) * 16 + i16;
        tmr[i] = (tile_of(i) >= 0 && m < p.M) ? p.Tm[m] : 0;
        tT[i] = __builtin_amdgcn_readfirstlane(tmr[i]);
    }
    auto sel = [](const int (&a)[4], int i) { return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : i == 3 ? a[3] : 0; };
    __syncthreads();
    if (p.prio >= 3) __builtin_amdgcn_s_setprio(3);  // a short dependent chain: outrank co-resident throughput kernels at issue
    else if (p.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
    if (tT[0] <= 0) return;
    // Placement census, part 2: the UB workgroups of the cluster read each other's entries; all of them see the same
    // UB values, so all of them take the same decision.  One XCD for the whole cluster (what the grid layout aims
    // for and the dispatcher has always delivered) -> hand-offs through that XCD's L2; anything else -> write-through
    // stores.  Placement therefore changes speed only.
    bool local;
    {
        const gu32* pl = (const gu32*)p.place + cid * UB;
        uint32_t v = xcc + 1u;
        for (uint32_t spins = 0;; spins++) {
            if (lane < UB) v = __hip_atomic_load(pl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!__any(v == 0u)) break;
            __builtin_amdgcn_s_sleep(8);
            if (spins >= p.spin_limit) {
                __hip_atomic_store((gu32*)p.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        local = !__any(v != xcc + 1u) && p.allow_local;
    }
    // items in (step, tile) order; the state loads of the NEXT item are issued before the current one is
    // computed whenever it belongs to another tile (its inputs cannot depend on the current item)
    const __amdgpu_buffer_rsrc_t yb = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(uint32_t)(p.R * 2 * H * sizeof(float)), 0x00020000);
    // Two register sets used alternately (A holds the current item while B receives the next one's loads, then
    // the roles swap): copying a set would make the wave wait for loads that are still in flight.
    int s = 0, i = 0;
    Loaded<H> bufA, bufB;
    issue_meta<H>(p, dir, ub, tile_of(0), tmr[0], off_l, 0, i16, kq, bufA);
    issue_state<H>(yb, ub, kq, bufA);   // step 0: no previous state, h = 0
    // one item: returns 0 = done, 1 = go on, -1 = timed out
    auto item = [&](Loaded<H>& cur, Loaded<H>& nxt) -> int {
        int ns = s, ni = i + 1;
        if (sel(tT, ni) <= s) { ns = s + 1; ni = 0; }
        const bool have_next = sel(tT, ni) > ns;
        const bool early = EARLY && have_next && ni != i;
        // B operand: lane (row, kq) feeds h[row][4*s4 + kq].  Turned BEFORE the next item's loads are issued so
        // that those can land in the registers the pieces leave behind.
        float w[H / 4];
#pragma unroll
        for (int j = 0; j < H / 16; j++) transpose4(cur.h[j], &w[4 * j]);
        if (have_next) issue_meta<H>(p, dir, ub, tile_of(ni), sel(tmr, ni), off_l, ns, i16, kq, nxt);
        if (early) issue_state<H>(yb, ub, kq, nxt);
        const GateAcc acc = mfma_chain<H>(lane, w, lds_w, br, bz, bn);
        epilogue_store<H>(yb, acc, cur.gr, cur.gz, cur.gn, cur.hp, cur.out_off, cur.active, local);
        if (!have_next) return 0;
        if (!early) issue_state<H>(yb, ub, kq, nxt);
        if (!await_state<H>(p, yb, ub, kq, nxt)) return -1;
        s = ns;
        i = ni;
        return 1;
    };
    for (;;) {
        if (item(bufA, bufB) <= 0) break;
        if (item(bufB, bufA) <= 0) break;
    }
}



// ---------------------------------------------------------------------------------------------------------------
// Gate-per-wave variant for requests with so few row tiles that every tile gets a cluster of its own (<= 8 tiles per
// direction at H = 256: a single page).  There the recurrence is a pure latency chain — T dependent steps, nothing to
// interleave — and the longest link of a step is the wave's 192 MFMAs.  Here the three gates of a step run on three
// waves (three SIMDs) at once, 64 MFMAs each, and the fourth wave does the gate arithmetic and the store:
//   waves 0..2 (gate r, z, n):  previous state of the tile (polled as above) -> 4x4 transposes -> 64-MFMA chain of
//                               their gate -> accumulators to LDS -> workgroup barrier
//   wave 3:                     gx of the step (prefetched one step ahead) and the previous state of its own 4 units
//                               (kept in registers: it wrote them) -> barrier -> gates from LDS -> sigma / tanh -> store
// Same arithmetic per output (each gate's chain is the k-ascending fmaf chain), so the bits equal the other paths'.
// The exchange buffer needs no second barrier: a gate wave can only write step s + 1's accumulators after it has seen
// the tile's state of step s, which includes what this workgroup's wave 3 stored after reading step s's accumulators.
// ---------------------------------------------------------------------------------------------------------------
template <int H>
__device__ __forceinline__ f32x4 gate_chain(int lane, const float (&w)[H / 4], const float* lds_w, int g, const f32x4& bias) {
    f32x4 acc = bias;
    const f32x4* ap = reinterpret_cast<const f32x4*>(lds_w) + g * (H / 16) * 64 + lane;
    f32x4 a = ap[0];
#pragma unroll
    for (int blk = 0; blk < H / 16; blk++) {
        f32x4 na = a;
        if (blk + 1 < H / 16) na = ap[(blk + 1) * 64];
#pragma unroll
        for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], w[4 * blk + e], acc, 0, 0, 0);
        a = na;
    }
    return acc;
}

template <int H, bool PACK2>
__global__ void __launch_bounds__(256)
gru_gates_kernel(GruParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds_w[];  // H*48 floats | off[Tmax+1] | exchange | abort
    // ONE workgroup of this kernel per CU, enforced through the register file: the clobber pushes the allocation past
    // 256 registers per lane, i.e. one wave per SIMD.  At its natural 117 registers the hardware puts two workgroups
    // on a CU as soon as other requests' kernels occupy part of the chip; their six polling waves then re-issued the 17
    // state loads back to back and kept that CU's memory pipeline so full that the awaited stores did not get through
    // (waits of seconds in every run with 2+ requests in flight).  await_state now backs off, with which the shared
    // placement works too; one workgroup per CU — what the general kernel's 299 registers impose anyway — stays
    // because it measured 5 % faster under load.  PACK2 (option "gru_gates_pack" = 2) drops the clobber: two workgroups
    // per CU, i.e. room for twice the row tiles (requests of up to ~3 pages) at this kernel's shorter step.
    if (!PACK2) asm volatile("v_accvgpr_write_b32 a200, 0" ::: "a200");
    constexpr int UB = H / 16;
    const int b = blockIdx.x;
    const int q = b >> 3;
    const int ub = p.scatter ? b % UB : q % UB;
    const int cid = p.scatter ? b / UB : (q / UB) * 8 + (b & 7);
    if (cid >= 2 * p.ncl) return;
    const int dir = cid & 1, tile = cid >> 1;    // one tile per cluster
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if (tid == 0) __hip_atomic_store((gu32*)p.place + cid * UB + ub, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float* __restrict__ whd = p.wh + (int64_t)dir * H * 3 * H;
    const float* __restrict__ bhd = p.bh + (int64_t)dir * 3 * H;
    const int j0 = ub * 16;
    for (int i = tid; i < H * 12; i += 256) {   // Wh slice -> LDS, layout as in gru_persistent_kernel
        const int k = i / 12, qq = i - k * 12;
        const int g = qq >> 2, c4 = (qq & 3) * 4;
        const float4 v = *reinterpret_cast<const float4*>(whd + (int64_t)k * 3 * H + g * H + j0 + c4);
        const int blk = k >> 4, e = (k >> 2) & 3, kk = k & 3;
        float* dst = &lds_w[(((g * (H / 16) + blk) * 64) + kk * 16 + c4) * 4 + e];
        dst[0] = v.x; dst[4] = v.y; dst[8] = v.z; dst[12] = v.w;
    }
    int* off_l = reinterpret_cast<int*>(lds_w + H * 48);
    for (int i = tid; i <= p.Tmax; i += 256) off_l[i] = p.off[i];
    // exchange area behind the off table, 16-byte aligned: 3 gates x 64 lanes x 16 bytes, then the abort word
    f32x4* xch = reinterpret_cast<f32x4*>(lds_w + H * 48 + (((p.Tmax + 1) + 3) & ~3));
    int* abort_w = reinterpret_cast<int*>(xch + 3 * 64);
    if (tid == 0) *abort_w = 0;
    const int m = tile * 16 + i16;
    const int tm = m < p.M ? p.Tm[m] : 0;
    const int T = __builtin_amdgcn_readfirstlane(tm);   // lane 0 = the tile's first = longest row
    __syncthreads();
    if (p.prio >= 3) __builtin_amdgcn_s_setprio(3);
    else if (p.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
    if (T <= 0) return;
    bool local;
    {   // placement census, part 2 (see gru_persistent_kernel)
        const gu32* pl = (const gu32*)p.place + cid * UB;
        uint32_t v = xcc + 1u;
        for (uint32_t spins = 0;; spins++) {
            if (lane < UB) v = __hip_atomic_load(pl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!__any(v == 0u)) break;
            __builtin_amdgcn_s_sleep(8);
            if (spins >= p.spin_limit) {
                __hip_atomic_store((gu32*)p.sync, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *abort_w = 1;   // the other waves leave at their next barrier
                break;
            }
        }
        local = !__any(v != xcc + 1u) && p.allow_local;
    }
    const __amdgpu_buffer_rsrc_t yb = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(uint32_t)(p.R * 2 * H * sizeof(float)), 0x00020000);
    if (wave < 3) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(bhd + wave * H + j0 + kq * 4);
        Loaded<H> L;
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        L.gr = L.gz = L.gn = zero;
        L.out_off = 0;
        for (int s = 0; s < T; s++) {
            L.active = tm > s;
            L.has_prev = L.active && s > 0;
            L.hp = zero;
#pragma unroll
            for (int j = 0; j < H / 16; j++) L.h[j] = zero;
            L.prev_off = L.has_prev ? (uint32_t)((((int64_t)off_l[dir ? tm - s : s - 1] + m) * 2 * H + dir * H) * sizeof(float)) : 0u;
            issue_state<H>(yb, ub, kq, L);
            if (!await_state<H>(p, yb, ub, kq, L)) *abort_w = 1;
            float w[H / 4];
#pragma unroll
            for (int j = 0; j < H / 16; j++) transpose4(L.h[j], &w[4 * j]);
            xch[wave * 64 + lane] = gate_chain<H>(lane, w, lds_w, wave, bias);
            __syncthreads();
            if (*abort_w) return;
        }
    } else {
        f32x4 hp = {0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 gr, gz, gn, ngr, ngz, ngn;
        uint32_t out_off, nout_off;
        bool active, nactive;
        auto fetch = [&](int s, f32x4& r_, f32x4& z_, f32x4& n_, uint32_t& o_, bool& a_) {
            a_ = tm > s;
            const int t = dir ? tm - 1 - s : s;
            const int64_t row = a_ ? (int64_t)off_l[t] + m : 0;
            o_ = (uint32_t)((row * 2 * H + dir * H + ub * 16 + kq * 4) * sizeof(float));
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            r_ = z_ = n_ = zero;
            if (a_) {
                const float* g = p.gx + ((int64_t)dir * p.R + row) * 3 * H + ub * 16 + kq * 4;
                r_ = *reinterpret_cast<const f32x4*>(g);
                z_ = *reinterpret_cast<const f32x4*>(g + H);
                n_ = *reinterpret_cast<const f32x4*>(g + 2 * H);
            }
        };
        fetch(0, gr, gz, gn, out_off, active);
        for (int s = 0; s < T; s++) {
            if (s + 1 < T) fetch(s + 1, ngr, ngz, ngn, nout_off, nactive);
            __syncthreads();
            if (*abort_w) return;
            GateAcc a;
            a.r = xch[lane]; a.z = xch[64 + lane]; a.n = xch[128 + lane];
            f32x4 hn;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float rg = sigmoidf_sl(gr[r] + a.r[r]);
                const float zg = sigmoidf_sl(gz[r] + a.z[r]);
                const float ng = tanhf_sl(fmaf(rg, a.n[r], gn[r]));
                const float hv = fmaf(zg, hp[r] - ng, ng);
                hn[r] = __float_as_uint(hv) == kUnwritten ? __uint_as_float(0x7FC00000u) : hv;
            }
            const uint32_t o = active ? out_off : 0xFFFFFFF0u;
            if (local) store_local(yb, o, hn);
            else store_through(yb, o, hn);
            if (active) hp = hn;    // what the next step would read back as this row's previous state
            gr = ngr; gz = ngz; gn = ngn; out_off = nout_off; active = nactive;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// r3: the gate-per-wave layout for LARGE requests, as a background kernel.
//
// The general kernel above is built to finish a layer as fast as possible: three interleaved MFMA chains per wave,
// the next item's 17 loads in flight under them, 299 registers — one wave per SIMD, and while it is resident a CU has
// room for ONE block of another request's conv stack instead of four (conv3x3 0.79 of the MFMA peak alone, 0.63-0.67
// live).  The recurrence is 9 % of a step's arithmetic; the conv stacks are 79 %.  This kernel trades the
// recurrence's own speed for its footprint: the gate-per-wave roles of gru_gates_kernel (117 registers: it fits in
// the register space of ONE conv wave per SIMD, so a CU keeps three conv blocks beside it), no second register set —
// a wave's load and hand-off latencies are filled by the conv waves it shares the SIMD with — and SEVERAL row tiles
// per cluster, stepped round-robin (tile j's state of step s - 1 has a whole round of the other tiles to arrive).
//   waves 0..2 (gate r, z, n) per item (tile j, step s): previous state of the tile (polled) -> transposes ->
//              64-MFMA chain of their gate -> accumulators to LDS (two exchange buffers, used alternately) -> barrier
//   wave 3:    gx of the item (prefetched one item ahead), previous state of its own 4 units (read back from y) ->
//              barrier -> gates from LDS -> sigma / tanh -> store
// One barrier per item: the gate waves reach item i + 2 (same exchange buffer as item i) only through the barrier of
// item i + 1, which wave 3 enters after it has read item i's accumulators.
// Same arithmetic per output as every other path (tests compare the bits).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMultiTiles = 16;   // row tiles per cluster

template <int H>
__global__ void __launch_bounds__(256)
gru_gates_multi_kernel(GruParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds_w[];  // H*48 floats | off[Tmax+1] | 2 x exchange | tile tables | abort
    constexpr int UB = H / 16;
    const int b = blockIdx.x;
    const int q = b >> 3;
    const int ub = p.scatter ? b % UB : q % UB;
    const int cid = p.scatter ? b / UB : (q / UB) * 8 + (b & 7);
    if (cid >= 2 * p.ncl) return;
    const int dir = cid & 1, cl = cid >> 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if (tid == 0) __hip_atomic_store((gu32*)p.place + cid * UB + ub, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float* __restrict__ whd = p.wh + (int64_t)dir * H * 3 * H;
    const float* __restrict__ bhd = p.bh + (int64_t)dir * 3 * H;
    const int j0 = ub * 16;
    for (int i = tid; i < H * 12; i += 256) {   // Wh slice -> LDS, layout as in gru_persistent_kernel
        const int k = i / 12, qq = i - k * 12;
        const int g = qq >> 2, c4 = (qq & 3) * 4;
        const float4 v = *reinterpret_cast<const float4*>(whd + (int64_t)k * 3 * H + g * H + j0 + c4);
        const int blk = k >> 4, e = (k >> 2) & 3, kk = k & 3;
        float* dst = &lds_w[(((g * (H / 16) + blk) * 64) + kk * 16 + c4) * 4 + e];
        dst[0] = v.x; dst[4] = v.y; dst[8] = v.z; dst[12] = v.w;
    }
    int* off_l = reinterpret_cast<int*>(lds_w + H * 48);
    for (int i = tid; i <= p.Tmax; i += 256) off_l[i] = p.off[i];
    f32x4* xch = reinterpret_cast<f32x4*>(lds_w + H * 48 + (((p.Tmax + 1) + 3) & ~3));   // [2][3][64]
    int* tm_l = reinterpret_cast<int*>(xch + 2 * 3 * 64);      // [kMultiTiles][16] lengths of the tiles' rows
    int* tile_l = tm_l + kMultiTiles * 16;                      // [kMultiTiles] tile index (-1: none)
    int* abort_w = tile_l + kMultiTiles;
    if (tid == 0) *abort_w = 0;
    for (int i = tid; i < kMultiTiles * 16; i += 256) {
        const int t = p.tiles[cl * kMultiTiles + (i >> 4)];
        const int m = t * 16 + (i & 15);
        tm_l[i] = (t >= 0 && m < p.M) ? p.Tm[m] : 0;
        if ((i & 15) == 0) tile_l[i >> 4] = t;
    }
    __syncthreads();
    if (p.prio >= 3) __builtin_amdgcn_s_setprio(3);
    else if (p.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
    if (tm_l[0] <= 0) return;                     // (lists are sorted by length: an empty first tile = an empty cluster)
    bool local;
    {   // placement census, part 2 (see gru_persistent_kernel)
        const gu32* pl = (const gu32*)p.place + cid * UB;
        uint32_t v = xcc + 1u;
        for (uint32_t spins = 0;; spins++) {
            if (lane < UB) v = __hip_atomic_load(pl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!__any(v == 0u)) break;
            __builtin_amdgcn_s_sleep(8);
            if (spins >= p.spin_limit) {
                __hip_atomic_store((gu32*)p.sync, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *abort_w = 1;
                break;
            }
        }
        local = !__any(v != xcc + 1u) && p.allow_local;
    }
    const __amdgpu_buffer_rsrc_t yb = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(uint32_t)(p.R * 2 * H * sizeof(float)), 0x00020000);
    // item iterator, identical in all four waves: (step s, list position j); tile j is live at step s iff its longest
    // row (row 0 of the tile) is longer than s; lists are sorted by length, so the live tiles of a step are a prefix
    auto tile_T = [&](int j) { return j < kMultiTiles ? tm_l[j * 16] : 0; };
    auto advance = [&](int& s, int& j) -> bool {   // false: no further item
        if (tile_T(j + 1) > s) { j++; return true; }
        if (tile_T(0) > s + 1) { s++; j = 0; return true; }
        return false;
    };
    int s = 0, j = 0, it = 0;
    if (wave < 3) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(bhd + wave * H + j0 + kq * 4);
        Loaded<H> L;
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        L.gr = L.gz = L.gn = zero;
        L.out_off = 0;
        for (;;) {
            const int tm = tm_l[j * 16 + i16];
            const int m = tile_l[j] * 16 + i16;
            L.active = tm > s;
            L.has_prev = L.active && s > 0;
            L.hp = zero;
#pragma unroll
            for (int jj = 0; jj < H / 16; jj++) L.h[jj] = zero;
            L.prev_off = L.has_prev ? (uint32_t)((((int64_t)off_l[dir ? tm - s : s - 1] + m) * 2 * H + dir * H) * sizeof(float)) : 0u;
            issue_state<H>(yb, ub, kq, L);
            if (!await_state<H>(p, yb, ub, kq, L)) *abort_w = 1;
            float w[H / 4];
#pragma unroll
            for (int jj = 0; jj < H / 16; jj++) transpose4(L.h[jj], &w[4 * jj]);
            xch[(it & 1) * 192 + wave * 64 + lane] = gate_chain<H>(lane, w, lds_w, wave, bias);
            __syncthreads();
            if (*abort_w) return;
            it++;
            if (!advance(s, j)) return;
        }
    } else {
        struct Gx { f32x4 gr, gz, gn, hp; uint32_t out_off; bool active; };
        // gx of an item + the previous state of this lane's own 4 units (its own store of one round earlier, read back
        // with the hand-off's bypassing load; polled like any other state word, although it has long landed)
        auto fetch = [&](int ss, int jj, Gx& g) {
            const int tm = tm_l[jj * 16 + i16];
            const int m = tile_l[jj] * 16 + i16;
            g.active = tm > ss;
            const int t = dir ? tm - 1 - ss : ss;
            const int64_t row = g.active ? (int64_t)off_l[t] + m : 0;
            g.out_off = (uint32_t)((row * 2 * H + dir * H + ub * 16 + kq * 4) * sizeof(float));
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            g.gr = g.gz = g.gn = g.hp = zero;
            if (g.active) {
                const float* gp = p.gx + ((int64_t)dir * p.R + row) * 3 * H + ub * 16 + kq * 4;
                g.gr = *reinterpret_cast<const f32x4*>(gp);
                g.gz = *reinterpret_cast<const f32x4*>(gp + H);
                g.gn = *reinterpret_cast<const f32x4*>(gp + 2 * H);
                if (ss > 0) {
                    const uint32_t po = (uint32_t)((((int64_t)off_l[dir ? tm - ss : ss - 1] + m) * 2 * H + dir * H + ub * 16 + kq * 4) * sizeof(float));
                    g.hp = load_bypass(yb, po);
                    // re-read while any word is still unwritten (bounded; cannot really happen: see above)
                    for (uint32_t spins = 0; spins < 4096u; spins++) {
                        unsigned mx = 0;
#pragma unroll
                        for (int e = 0; e < 4; e++) mx = max(mx, __float_as_uint(g.hp[e]));
                        if (mx != kUnwritten) break;
                        __builtin_amdgcn_s_sleep(4);
                        g.hp = load_bypass(yb, po);
                    }
                }
            }
        };
        auto finish = [&](const Gx& g) {   // after the item's barrier: gates from the exchange buffer, store
            GateAcc a;
            const f32x4* x = xch + (it & 1) * 192;
            a.r = x[lane]; a.z = x[64 + lane]; a.n = x[128 + lane];
            f32x4 hn;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float rg = sigmoidf_sl(g.gr[r] + a.r[r]);
                const float zg = sigmoidf_sl(g.gz[r] + a.z[r]);
                const float ng = tanhf_sl(fmaf(rg, a.n[r], g.gn[r]));
                const float hv = fmaf(zg, g.hp[r] - ng, ng);
                hn[r] = __float_as_uint(hv) == kUnwritten ? __uint_as_float(0x7FC00000u) : hv;
            }
            const uint32_t o = g.active ? g.out_off : 0xFFFFFFF0u;
            if (local) store_local(yb, o, hn);
            else store_through(yb, o, hn);
        };
        Gx ga, gb;
        fetch(0, 0, ga);
        // two register sets used alternately, as in the general kernel (copying one would wait for loads in flight)
        auto step = [&](Gx& cur, Gx& nxt) -> bool {   // false: done / aborted
            int ns = s, nj = j;
            const bool have_next = advance(ns, nj);
            // the next item's gx may be fetched now; its hp only if it belongs to ANOTHER tile (this item's own store is
            // the previous state of the same tile's next step) — with one tile in the list it is fetched after the store
            const bool same_tile = have_next && nj == j;
            if (have_next && !same_tile) fetch(ns, nj, nxt);
            __syncthreads();
            if (*abort_w) return false;
            finish(cur);
            it++;
            if (!have_next) return false;
            if (same_tile) fetch(ns, nj, nxt);
            s = ns; j = nj;
            return true;
        };
        for (;;) {
            if (!step(ga, gb)) return;
            if (!step(gb, ga)) return;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Round 4: gate-per-wave TEAMS for large requests (gru_teams_kernel, option "gru_waves" = 16; NOT the default — measured
// 6.2 ms per layer against the general kernel's 5.5 in the same serial run, see the end of this comment).
//
// What bounds the general kernel (4.58 ms per layer on the bench's 1 232 lines, 0.41 of the fp32 MFMA peak) is not
// the matrix pipe (1.9 ms of MFMAs per SIMD) but the chain of dependent steps of the LONGEST lines: 600 steps x 6.6 us —
// store, L2 round trip, 17 loads, check, 64 lane swaps, 192 MFMAs (2.6 us), gate arithmetic, all on one wave.  The
// gate-per-wave kernel above cuts that step to 4.55 us (three gates on three SIMDs at once, 64 MFMAs each; the gate
// arithmetic on a fourth wave) but only had one such team per workgroup, which for a 16-page request ran at a third of
// the general kernel's throughput (r2: 7.2 ms, r3's lean variant 10.1 ms per layer).
// Here a workgroup holds FOUR teams of four waves (1 024 threads, ~120 registers: four waves per SIMD), all on the same
// 16-unit slice of Wh in LDS, each team stepping its own list of row tiles round-robin.  A SIMD then carries three gate
// waves and one arithmetic wave of different teams (roles are rotated by team), so one team's load / poll / swap /
// exchange latencies lie under the other teams' MFMA chains, and the longest lines advance at the short step.
// Teams synchronise among their four waves through LDS words (produced / consumed item counters) — no workgroup barrier
// in the loop, teams never wait for each other.  Arithmetic per output is that of every other path (k-ascending chain
// per gate, the same gate formulas): the tests compare the bits.
// Three builds, all bit-identical, per layer on the bench's 1 232 lines (serial run, same box; general kernel 5.4-5.6 ms):
//   v1  every gate wave loads and polls the tile's state itself: 10.6 ms — twelve pollers per CU re-issuing 17 line fetches
//       each keep the CU's memory pipeline so full that the awaited stores crawl (the effect r2 saw with six);
//   v2  one poller per team, the other two gate waves wait on an LDS word and then load: 7.7 ms;
//   v3  (this code) the fourth wave of a team is loader + arithmetic: state global -> LDS directly (global_load_lds, no
//       staging registers: 83 VGPRs), lane transposes in place, the gate waves stream BOTH operands from LDS and touch no
//       global memory: 6.2 ms.  What is left: a team's items pass one after the other through ONE 16 KB operand buffer
//       (two per team do not fit the LDS beside the Wh slice), and the compiler's vmcnt(0) at the loop's back edge makes the
//       loader wait for its own write-through store every item.
// Also built and dropped on the way: the general kernel with eight waves per workgroup, two per SIMD, one register set, no
// early issue (174 VGPRs): 5.8 ms — the second wave does not make up for the lost prefetch.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kTeams = 4;        // teams per workgroup
constexpr int kTeamTiles = 8;    // row tiles per team
// LDS words (floats) of one team: B-operand buffer | 2 x exchange | tile tables | flags
template <int H> constexpr int team_words() { return (H / 16) * 64 * 4 + 2 * 3 * 64 * 4 + kTeamTiles * 16 + kTeamTiles + 8; }

__device__ __forceinline__ int lds_flag_load(const int* f) {
    return __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_flag_store(int* f, int v) {
    __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One gate's chain with BOTH operands streamed from LDS a block ahead: A = the Wh slice, B = the team's state buffer
// (tb[blk * 64 + lane] = h[row][16 blk + 4 e + kq], e = 0..3)
template <int H>
__device__ __forceinline__ f32x4 gate_chain_lds(int lane, const f32x4* tb, const float* lds_w, int g, const f32x4& bias) {
    f32x4 acc = bias;
    const f32x4* ap = reinterpret_cast<const f32x4*>(lds_w) + g * (H / 16) * 64 + lane;
    const f32x4* bp = tb + lane;
    f32x4 a = ap[0], bq = bp[0];
#pragma unroll
    for (int blk = 0; blk < H / 16; blk++) {
        f32x4 na = a, nb = bq;
        if (blk + 1 < H / 16) { na = ap[(blk + 1) * 64]; nb = bp[(blk + 1) * 64]; }
#pragma unroll
        for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], bq[e], acc, 0, 0, 0);
        a = na; bq = nb;
    }
    return acc;
}

template <int H>
__global__ void __launch_bounds__(64 * 4 * kTeams)
gru_teams_kernel(GruParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds_w[];  // H*48 floats | off[Tmax+1] | per team {B operand, 2 x exchange, tile tables, flags} | abort
    constexpr int UB = H / 16;
    constexpr int NT = 64 * 4 * kTeams;
    constexpr int kTeamWords = team_words<H>();
    const int b = blockIdx.x;
    const int q = b >> 3;
    const int ub = p.scatter ? b % UB : q % UB;
    const int cid = p.scatter ? b / UB : (q / UB) * 8 + (b & 7);
    if (cid >= 2 * p.ncl) return;
    const int dir = cid & 1, cl = cid >> 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wave >> 2;
    const int role = ((wave & 3) + team) & 3;   // 0..2: gate r / z / n; 3: state loader + gate arithmetic.  Rotated by team:
                                                // wave w runs on SIMD w mod 4, so every SIMD carries three gate waves and one loader
    const int i16 = lane & 15, kq = lane >> 4;
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if (tid == 0) __hip_atomic_store((gu32*)p.place + cid * UB + ub, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float* __restrict__ whd = p.wh + (int64_t)dir * H * 3 * H;
    const float* __restrict__ bhd = p.bh + (int64_t)dir * 3 * H;
    const int j0 = ub * 16;
    for (int i = tid; i < H * 12; i += NT) {   // Wh slice -> LDS, layout as in gru_persistent_kernel
        const int k = i / 12, qq = i - k * 12;
        const int g = qq >> 2, c4 = (qq & 3) * 4;
        const float4 v = *reinterpret_cast<const float4*>(whd + (int64_t)k * 3 * H + g * H + j0 + c4);
        const int blk = k >> 4, e = (k >> 2) & 3, kk = k & 3;
        float* dst = &lds_w[(((g * (H / 16) + blk) * 64) + kk * 16 + c4) * 4 + e];
        dst[0] = v.x; dst[4] = v.y; dst[8] = v.z; dst[12] = v.w;
    }
    int* off_l = reinterpret_cast<int*>(lds_w + H * 48);
    for (int i = tid; i <= p.Tmax; i += NT) off_l[i] = p.off[i];
    float* team_base = lds_w + H * 48 + (((p.Tmax + 1) + 3) & ~3) + team * kTeamWords;
    f32x4* tbuf = reinterpret_cast<f32x4*>(team_base);                       // [H/16][64]: the item's B operand
    f32x4* xch = tbuf + (H / 16) * 64;                                       // [2][3][64]: gate accumulators
    int* tm_l = reinterpret_cast<int*>(xch + 2 * 3 * 64);                    // [kTeamTiles][16] lengths of the tiles' rows
    int* tile_l = tm_l + kTeamTiles * 16;                                    // [kTeamTiles] tile index (-1: none)
    int* flags = tile_l + kTeamTiles;   // 0..2 accumulators of gate g written (items; the gate is then also done with tbuf),
                                        // 3 accumulators read by the loader (items), 4 B operand of item it in tbuf (it + 1)
    int* abort_w = reinterpret_cast<int*>(lds_w + H * 48 + (((p.Tmax + 1) + 3) & ~3) + kTeams * kTeamWords);
    if (tid == 0) *abort_w = 0;
    {
        const int tt = tid & 255;   // the team's own 256 threads fill the team's tables
        if (tt < 8) flags[tt] = 0;
        if (tt < kTeamTiles * 16) {
            const int t = p.tiles[(cl * kTeams + team) * kTeamTiles + (tt >> 4)];
            const int m = t * 16 + (tt & 15);
            tm_l[tt] = (t >= 0 && m < p.M) ? p.Tm[m] : 0;
            if ((tt & 15) == 0) tile_l[tt >> 4] = t;
        }
    }
    __syncthreads();
    if (p.prio >= 3) __builtin_amdgcn_s_setprio(3);
    else if (p.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
    if (tm_l[0] <= 0) return;                     // (lists are sorted by length: an empty first tile = a team without work)
    // item iterator, identical in the team's four waves: (step s, list position j); the live tiles of a step are a prefix
    auto tile_T = [&](int j) { return j < kTeamTiles ? tm_l[j * 16] : 0; };
    auto advance = [&](int& s, int& j) -> bool {   // false: no further item
        if (tile_T(j + 1) > s) { j++; return true; }
        if (tile_T(0) > s + 1) { s++; j = 0; return true; }
        return false;
    };
    // bounded wait on one of the team's LDS counters; false: aborted / timed out
    auto wait_flag = [&](const int* f, int want) -> bool {
        for (uint32_t spins = 0; lds_flag_load(f) < want; spins++) {
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 63u) == 63u) {
                if (*abort_w) return false;
                if (spins >= p.spin_limit) {
                    __hip_atomic_store((gu32*)p.sync, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *abort_w = 1;
                    return false;
                }
            }
        }
        return true;
    };
    int s = 0, j = 0, it = 0;
    if (role < 3) {
        // ---- gate wave: no global memory at all.  B operand from the team's buffer, A from the Wh slice, 64 MFMAs.
        const f32x4 bias = *reinterpret_cast<const f32x4*>(bhd + role * H + j0 + kq * 4);
        for (;;) {
            if (!wait_flag(flags + 4, it + 1)) return;
            const f32x4 acc = gate_chain_lds<H>(lane, tbuf, lds_w, role, bias);
            // exchange buffer (it & 1) was item it - 2's: the loader must have read that one
            if (it >= 2 && !wait_flag(flags + 3, it - 1)) return;
            xch[(it & 1) * 192 + role * 64 + lane] = acc;
            if (lane == 0) lds_flag_store(flags + role, it + 1);       // (release: after this wave's exchange writes)
            it++;
            if (!advance(s, j)) return;
        }
    } else {
        // ---- loader + arithmetic wave: the team's only contact with global memory
        bool local;
        {   // placement census, part 2 (see gru_persistent_kernel)
            const gu32* pl = (const gu32*)p.place + cid * UB;
            uint32_t v = xcc + 1u;
            for (uint32_t spins = 0;; spins++) {
                if (lane < UB) v = __hip_atomic_load(pl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!__any(v == 0u)) break;
                __builtin_amdgcn_s_sleep(8);
                if (spins >= p.spin_limit) {
                    __hip_atomic_store((gu32*)p.sync, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *abort_w = 1;
                    return;
                }
            }
            local = !__any(v != xcc + 1u) && p.allow_local;
        }
        const __amdgpu_buffer_rsrc_t yb = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(uint32_t)(p.R * 2 * H * sizeof(float)), 0x00020000);
        struct Gx { f32x4 gr, gz, gn; uint32_t out_off; bool active; };
        auto fetch_gx = [&](int ss, int jj, Gx& g) {
            const int tm = tm_l[jj * 16 + i16];
            const int m = tile_l[jj] * 16 + i16;
            g.active = tm > ss;
            const int t = dir ? tm - 1 - ss : ss;
            const int64_t row = g.active ? (int64_t)off_l[t] + m : 0;
            g.out_off = (uint32_t)((row * 2 * H + dir * H + ub * 16 + kq * 4) * sizeof(float));
            // unconditional (idle lanes read row 0 and their result is never stored): exactly three load instructions per
            // call, which settle()'s partial vmcnt wait counts on
            const float* gp = p.gx + ((int64_t)dir * p.R + row) * 3 * H + ub * 16 + kq * 4;
            g.gr = *reinterpret_cast<const f32x4*>(gp);
            g.gz = *reinterpret_cast<const f32x4*>(gp + H);
            g.gn = *reinterpret_cast<const f32x4*>(gp + 2 * H);
        };
        // The tile's previous state goes global -> LDS directly (16 x global_load_lds_dwordx4 per lane group: lane (row, kq)
        // fetches the row's pieces 16 j + 4 kq .. + 3, which land at tbuf[j * 64 + lane]), so the 64 registers a staged copy
        // would need do not exist; the 17th piece — this lane's own 4 units, for the gate arithmetic — comes to a register.
        // Rows without a previous state (first step / idle lanes) get zeros.
        bool has_prev = false;
        uint32_t prev_off = 0;
        f32x4 hpn = {0.0f, 0.0f, 0.0f, 0.0f};
        auto state_meta = [&](int ss, int jj) {
            const int tm = tm_l[jj * 16 + i16];
            const int m = tile_l[jj] * 16 + i16;
            has_prev = tm > ss && ss > 0;
            prev_off = has_prev ? (uint32_t)((((int64_t)off_l[dir ? tm - ss : ss - 1] + m) * 2 * H + dir * H) * sizeof(float)) : 0u;
        };
        auto issue_to_lds = [&]() {
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            if (has_prev) {
                const char* src = reinterpret_cast<const char*>(p.y) + prev_off + (4 * kq) * 4;
#pragma unroll
                for (int jj = 0; jj < H / 16; jj++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + jj * 64),
                                                     (__attribute__((address_space(3))) void*)(tbuf + jj * 64), 16, 0, kAuxSc1);
                hpn = load_bypass(yb, prev_off + (ub * 16 + kq * 4) * 4);
            } else {
#pragma unroll
                for (int jj = 0; jj < H / 16; jj++) tbuf[jj * 64 + lane] = zero;
                hpn = zero;
            }
        };
        // one pass over the landed pieces: unwritten words? 4x4 lane transposes in place (tbuf then holds the B operand)
        // `younger` = vector-memory instructions issued after the state loads (vmcnt retires in order): 4 when the loads went
        // out before this item's store (another tile's state: store + 3 gx loads are younger — waiting for vmcnt(0) would
        // also wait for the write-through store's acknowledgement, a trip to memory and back, every item), 3 when they followed
        // the store (same tile), 0 after a re-issue.
        auto settle = [&](int younger) -> bool {
            if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (younger == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned mx = 0;
#pragma unroll
            for (int jj = 0; jj < H / 16; jj++) {
                const f32x4 v = tbuf[jj * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; e++) mx = max(mx, __float_as_uint(v[e]));
                float t4[4];
                transpose4(v, t4);
                const f32x4 o = {t4[0], t4[1], t4[2], t4[3]};
                tbuf[jj * 64 + lane] = o;
            }
#pragma unroll
            for (int e = 0; e < 4; e++) mx = max(mx, __float_as_uint(hpn[e]));
            return !__any(mx == kUnwritten);
        };
        Gx cur;
        fetch_gx(0, 0, cur);
        state_meta(0, 0);
        issue_to_lds();
        int younger = 0;
        for (;;) {
            // (1) the item's state complete?  (polled here and only here: one poller per team)
            for (uint32_t spins = 0; !settle(spins == 0 ? younger : 0); spins++) {
                __builtin_amdgcn_s_sleep(2);
                if (spins >= 4u) {
                    const uint32_t n = spins < 36u ? (spins >> 2) : 9u;   // back off: 1 .. 9 x 1024 clocks
                    for (uint32_t z = 0; z < n; z++) __builtin_amdgcn_s_sleep(16);
                }
                if ((spins & 255u) == 255u) {
                    gu32* err = (gu32*)p.sync;
                    const uint32_t e = __builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    if (e != 0 || spins >= p.spin_limit || *abort_w) {
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        *abort_w = 1;
                        return;
                    }
                }
                issue_to_lds();
            }
            const f32x4 hp = hpn;
            if (lane == 0) lds_flag_store(flags + 4, it + 1);   // (release: after the buffer writes) -> the gate waves start
            // (2) the three accumulators; with them the gate waves are also done with the state buffer
            if (!wait_flag(flags + 0, it + 1) || !wait_flag(flags + 1, it + 1) || !wait_flag(flags + 2, it + 1)) return;
            GateAcc a;
            {
                const f32x4* x = xch + (it & 1) * 192;
                a.r = x[lane]; a.z = x[64 + lane]; a.n = x[128 + lane];
            }
            if (lane == 0) lds_flag_store(flags + 3, it + 1);   // (release: after this wave's exchange reads)
            // (3) the next item's state may be fetched under this item's gate arithmetic if it belongs to another tile
            int ns = s, nj = j;
            const bool have_next = advance(ns, nj);
            const bool same_tile = have_next && nj == j;
            if (have_next && !same_tile) { state_meta(ns, nj); issue_to_lds(); }
            epilogue_store_lean<H>(yb, a, cur.gr, cur.gz, cur.gn, hp, cur.out_off, cur.active, local);
            it++;
            if (!have_next) return;
            if (same_tile) { state_meta(ns, nj); issue_to_lds(); }   // after this item's store: it is that tile's previous state
            s = ns; j = nj;
            fetch_gx(s, j, cur);   // HBM loads; used after the state poll and the gates' chains
            younger = same_tile ? 3 : 4;
        }
    }
}

}  // namespace

constexpr int kMaxGrid = 4096;
// kernel for requests beyond one row tile per cluster (option "gru_waves"): 4 = the general kernel (one wave per SIMD, three
// interleaved chains per wave; default), 16 = four gate-per-wave teams per workgroup (gru_teams_kernel)
static int gru_waves() { return option(OPT_GRU_WAVES) == 16 ? 16 : 4; }
template <int H>
static void gru_teams_allow_lds() {   // once per device would do; the call is cheap
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_teams_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
static size_t gru_teams_lds_bytes(int H, int Tmax) {
    const size_t team = (size_t)(H / 16) * 64 * 4 + 2 * 3 * 64 * 4 + kTeamTiles * 16 + kTeamTiles + 8;   // team_words<H>()
    return (size_t)H * 48 * sizeof(float) + (size_t)(((Tmax + 1) + 3) & ~3) * sizeof(int) + (size_t)kTeams * team * sizeof(float) + 16;
}
static size_t gru_gates_lds_bytes(int H, int Tmax) {   // Wh slice | off table (padded to 16 bytes) | 3 x 64 x 16 B | abort word
    return (size_t)H * 48 * sizeof(float) + (size_t)(((Tmax + 1) + 3) & ~3) * sizeof(int) + 3 * 64 * 16 + 16;
}
size_t gru_persistent_sync_words(int) { return kMaxGrid + 1; }  // placement table + error word (last)

// y -> all words "unwritten"; any stream that is ordered before the recurrence (it does not depend on gx)
hipError_t gru_persistent_prepare(float* y, int64_t R, int H, hipStream_t s) {
    return R > 0 ? hipMemsetAsync(y, 0xFF, (size_t)R * 2 * H * sizeof(float), s) : hipSuccess;
}

// How many workgroups of the recurrence kernels the CURRENT device keeps resident at once: CUs x workgroups per CU
// (hipOccupancyMaxActiveBlocksPerMultiprocessor for the kernel's registers and LDS), capped at one per CU — the
// placement both kernels are built for.  Their workgroups wait for each other, so a launch is only safe when every
// workgroup of the earliest unfinished group of clusters can be resident together.  With in-order dispatch that
// holds iff the device holds one whole group (8 clusters x UB workgroups): the resident set is always the earliest
// unfinished workgroups, and while the earliest unfinished group is not fully dispatched fewer than 8 * UB of its
// workgroups hold slots, so the dispatcher still has room.  Devices / partitions with fewer slots get no plan and
// the caller runs the per-step kernels instead.  Cached per device.
static int gru_resident_capacity(int H, bool gates, size_t lds) {
    static std::mutex mu;
    static std::map<int, int> cache;   // key: device * 8 + (gates ? 4 : 0) + log-ish(H)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const int key = dev * 16 + (gates ? 4 : 0) + (!gates && gru_waves() == 16 ? 8 : 0) + (H == 256 ? 2 : H == 128 ? 1 : 0);
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    int per_cu = 0;
    hipError_t e;
    if (gates) {
        e = H == 256 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_gates_kernel<256, false>, 256, lds)
          : H == 128 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_gates_kernel<128, false>, 256, lds)
                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_gates_kernel<64, false>, 256, lds);
    } else {
        if (gru_waves() == 16) { gru_teams_allow_lds<256>(); gru_teams_allow_lds<128>(); gru_teams_allow_lds<64>(); }
        if (gru_waves() == 16)
            e = H == 256 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_teams_kernel<256>, 1024, lds)
              : H == 128 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_teams_kernel<128>, 1024, lds)
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_teams_kernel<64>, 1024, lds);
        else
            e = H == 256 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_persistent_kernel<256>, 256, lds)
              : H == 128 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_persistent_kernel<128>, 256, lds)
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_persistent_kernel<64>, 256, lds);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
    int cap = prop.multiProcessorCount * (per_cu > 0 ? 1 : 0);   // one workgroup per CU by design
    if (const char* env = getenv("OCRS_GRU_BLOCKS")) { const int v = atoi(env); if (v > 0) cap = v < cap ? v : cap; }
    std::lock_guard<std::mutex> g(mu);
    cache[key] = cap;
    return cap;
}

// Grid geometry: ncl clusters per direction (UB workgroups each); false if the shape is not supported.
// Every workgroup of a cluster must be resident at once, so the first wave of the grid stays within `max_blocks`
// resident workgroups (the device's capacity, at most one per CU: 256 on MI355X); requests of more than 4 tiles per
// wave at that size (> 2048 lines at H = 256) get twice the clusters, whose second half is dispatched as the first
// finishes (see gru_resident_capacity for why that cannot deadlock).
// wv = 4: the general kernel (4 wave slots per cluster, up to 4 tiles each); 16: the teams kernel (kTeams team slots per
// cluster, up to kTeamTiles tiles each).
static bool gru_plan(int M, int Tmax, int H, int* ncl, int max_blocks = 256, int wv = 4) {
    if (H != 256 && H != 128 && H != 64) return false;
    const int ntiles = (M + 15) / 16;
    const int UB = H / 16;
    max_blocks = max_blocks > 256 ? 256 : max_blocks;
    max_blocks -= max_blocks % (8 * UB);                 // whole groups of 8 clusters
    if (max_blocks < 8 * UB) return false;               // the device cannot hold one group of clusters
    int max_ncl = max_blocks / UB / 2;
    const int per_cluster = wv == 16 ? kTeams * kTeamTiles : 16;   // tiles a cluster can hold
    const int slots = wv == 16 ? kTeams : 4;
    if (ntiles > per_cluster * max_ncl) max_ncl *= 2;
    if (ntiles > per_cluster * max_ncl || slots * max_ncl > kMaxSlots) return false;
    *ncl = (ntiles + slots - 1) / slots < max_ncl ? (ntiles + slots - 1) / slots : max_ncl;
    // (the teams kernel raises its dynamic LDS limit: 160 KB per workgroup on gfx950; the general kernel stays within the default 64 KB)
    return wv == 16 ? gru_teams_lds_bytes(H, Tmax) <= 150 * 1024
                    : (size_t)H * 48 * sizeof(float) + ((size_t)Tmax + 1) * sizeof(int) <= 64 * 1024;
}

// Deal the row tiles (tile k = lines 16k .. 16k+15 of the length-sorted batch, so len[k] >= len[k + 1]) to the
// 4 * ncl waves of a direction, at most 4 per wave.  A wave steps its tiles round-robin; a round of n live tiles
// costs max(n * c, L): c = the wave's own work per item (MFMA chain + epilogue), L = the store -> visible -> re-read
// latency of the state exchange that a single tile cannot hide.  Its time is the sum over rounds, and the kernel
// ends with the slowest wave: longest-tile-first greedy on that cost (the longest tiles end up alone or with one
// short partner, the mid-length ones in twos and threes).  Dealing consecutive tiles to a cluster took 8.4 ms per
// layer on 1 232 lines of 100..600 steps, a snake deal 6.5 ms, this 4.7 ms.
static void gru_assign_tiles(const int32_t* h_Tm, int M, int ncl, int16_t* tiles, int wv = 4) {
    const int per = wv == 16 ? kTeamTiles : 4;               // tiles per slot
    const int ntiles = (M + 15) / 16, nslots = (wv == 16 ? kTeams : 4) * ncl;
    // units of 0.1 us.  General kernel (slot = wave): 4.7 us per item, 6.6 us per lone step (measured).  Teams kernel (slot =
    // team of four waves): ~3 us per item when the team has several tiles to interleave, 4.6 us per lone step.
    int64_t c = wv == 16 ? 30 : 47, L = wv == 16 ? 46 : 66;
    if (const char* e = getenv("OCRS_GRU_COST_C")) c = atoi(e);
    if (const char* e = getenv("OCRS_GRU_COST_L")) L = atoi(e);
    for (int i = 0; i < nslots * per; i++) tiles[i] = -1;
    std::vector<int> cnt(nslots, 0);
    auto cost = [&](int slot, int extra_len) {   // lengths are descending within a slot, extra_len <= all of them
        int64_t tot = 0;
        int len[kTeamTiles + 1], n = cnt[slot];
        for (int i = 0; i < n; i++) len[i] = h_Tm[tiles[slot * per + i] * 16];
        if (extra_len > 0) len[n++] = extra_len;
        for (int i = n - 1, below = 0; i >= 0; i--) {   // rounds in which exactly i + 1 tiles are live
            const int64_t per = (i + 1) * c > L ? (i + 1) * c : L;
            tot += (int64_t)(len[i] - below) * per;
            below = len[i];
        }
        return tot;
    };
    for (int k = 0; k < ntiles; k++) {
        const int len = h_Tm[k * 16];
        int best = -1;
        int64_t best_cost = 0;
        for (int sl = 0; sl < nslots; sl++) {
            if (cnt[sl] >= per) continue;
            const int64_t cs = cost(sl, len);
            if (best < 0 || cs < best_cost) { best = sl; best_cost = cs; }
        }
        tiles[best * per + cnt[best]++] = (int16_t)k;
    }
}

// The deal gru_persistent would use for these (descending) line lengths: clusters per direction and, per wave slot
// (cluster * 4 + wave), up to 4 row-tile indices, -1 = none.  Host only (tests).
bool gru_tile_plan(const int32_t* h_Tm, int M, int H, int* ncl, int* waves, int16_t* tiles /* [kMaxSlots * 4] */) {
    for (int i = 0; i < kMaxSlots * 4; i++) tiles[i] = -1;
    *waves = gru_waves();   // the whole buffer: slots beyond 4 * ncl stay "none"
    if (M <= 0 || !gru_plan(M, h_Tm[0], H, ncl, 256, gru_waves())) return false;   // (host only: planned for a 256-CU device)
    gru_assign_tiles(h_Tm, M, *ncl, tiles, gru_waves());
    return true;
}

static size_t gru_general_lds_bytes(int H, int Tmax) { return (size_t)H * 48 * sizeof(float) + ((size_t)Tmax + 1) * sizeof(int); }

bool gru_persistent_supported(int M, int Tmax, int64_t R, int H) {
    int ncl;
    if (H != 256 && H != 128 && H != 64) return false;
    // y is addressed through one buffer resource: < 4 GiB
    return M > 0 && (uint64_t)R * 2 * H * sizeof(float) < (uint64_t(1) << 32) &&
           gru_plan(M, Tmax, H, &ncl, gru_resident_capacity(H, false, gru_waves() == 16 ? gru_teams_lds_bytes(H, Tmax) : gru_general_lds_bytes(H, Tmax)), gru_waves());
}

// gate-per-wave kernel: every tile has a cluster of its own.  *pack = workgroups per CU the launch relies on.
static bool gru_gates_plan(int M, int Tmax, int H, int* ncl, int* pack) {
    if (H != 256 && H != 128 && H != 64) return false;
    if (gru_gates_lds_bytes(H, Tmax) > 64 * 1024) return false;
    const int ntiles = (M + 15) / 16, UB = H / 16;
    int cap = gru_resident_capacity(H, true, gru_gates_lds_bytes(H, Tmax));
    cap = cap > 256 ? 256 : cap;
    cap -= cap % (8 * UB);
    if (cap < 8 * UB) return false;
    // One workgroup per CU, as for the general kernel: cap / UB / 2 clusters per direction (8 at H = 256 on MI355X:
    // one page).  With option "gru_gates_pack" = 2 requests of up to twice that run two workgroups per CU (117
    // registers, two fit): 2-3 page requests then keep this kernel's 4.6 us step instead of the general kernel's
    // 6.6 us lone-tile step.  (Packed placement once made waits time out under concurrency — pollers saturating the
    // CU's memory pipeline — which await_state's back-off cured; see gru_gates_kernel.  A multi-tile variant of this
    // kernel for the 16-page request was also built: 7.2 vs 4.6 ms per layer, the general kernel's three interleaved
    // chains per wave use the matrix cores better.)
    const int max_ncl = cap / UB / 2;
    *pack = 1;
    if (ntiles > max_ncl) {
        if (option(OPT_GRU_GATES_PACK) < 2 || ntiles > 2 * max_ncl) return false;
        *pack = 2;
    }
    *ncl = ntiles;
    return true;
}


// ---- background (lean) recurrence for large requests: gru_gates_multi_kernel --------------------------------------
static size_t gru_multi_lds_bytes(int H, int Tmax) {
    return (size_t)H * 48 * sizeof(float) + (size_t)(((Tmax + 1) + 3) & ~3) * sizeof(int) + 2 * 3 * 64 * 16 +
           (size_t)kMultiTiles * 16 * sizeof(int) + (size_t)kMultiTiles * sizeof(int) + 16;
}

// ncl clusters per direction, up to kMultiTiles row tiles each.
static bool gru_multi_plan(int M, int Tmax, int H, int* ncl) {
    if (H != 256 && H != 128 && H != 64) return false;
    if (gru_multi_lds_bytes(H, Tmax) > 64 * 1024) return false;
    const int ntiles = (M + 15) / 16, UB = H / 16;
    int cap = gru_resident_capacity(H, true, gru_multi_lds_bytes(H, Tmax));
    cap = cap > 256 ? 256 : cap;
    cap -= cap % (8 * UB);
    if (cap < 8 * UB) return false;
    const int max_ncl = cap / UB / 2;
    if (ntiles > kMultiTiles * max_ncl || max_ncl * kMultiTiles > kMaxSlots * 4) return false;
    *ncl = ntiles < max_ncl ? ntiles : max_ncl;
    return true;
}

// Tiles (longest first) to the cluster with the least work so far (work = sum of the tiles' lengths: a cluster
// steps its tiles one after the other); every list ends up sorted by length.
static void gru_assign_multi(const int32_t* h_Tm, int M, int ncl, int16_t* tiles) {
    const int ntiles = (M + 15) / 16;
    for (int i = 0; i < kMaxSlots * 4; i++) tiles[i] = -1;
    std::vector<int64_t> load(ncl, 0);
    std::vector<int> cnt(ncl, 0);
    for (int k = 0; k < ntiles; k++) {
        int best = -1;
        for (int c = 0; c < ncl; c++)
            if (cnt[c] < kMultiTiles && (best < 0 || load[c] < load[best])) best = c;
        tiles[best * kMultiTiles + cnt[best]++] = (int16_t)k;
        load[best] += h_Tm[k * 16];
    }
}

bool gru_persistent(const float* gx, const float* wh, const float* bh, float* y, const int32_t* d_Tm, const int32_t* d_off,
                    const int32_t* h_Tm, int64_t R, int M, int Tmax, int H, uint32_t* d_sync, hipStream_t s) {
    if (M <= 0) return true;
    if (option(OPT_GRU_GATES)) {
        GruParams g{};
        int pack = 1;
        if (gru_gates_plan(M, Tmax, H, &g.ncl, &pack)) {
            g.gx = gx; g.wh = wh; g.bh = bh; g.y = y; g.Tm = d_Tm; g.off = d_off;
            g.place = d_sync;
            g.sync = d_sync + kMaxGrid;
            g.R = R; g.M = M; g.Tmax = Tmax;
            g.prio = 3;
            g.spin_limit = 1u << 21;
            g.allow_local = option(OPT_GRU_LOCAL) != 0;
            g.scatter = option(OPT_GRU_SCATTER) != 0;
            const int UBg = H / 16;
            const dim3 grid(8 * UBg * ((2 * g.ncl + 7) / 8));
            const size_t lds = gru_gates_lds_bytes(H, Tmax);
            OCRS_HIP(hipMemsetAsync(d_sync, 0, gru_persistent_sync_words(M) * sizeof(uint32_t), s));
            static const bool lean = getenv("OCRS_GRU_GATES_LEAN") != nullptr;   // experiment: 117 registers also for pack 1
            if (pack == 2 || lean) {
                if (H == 256) hipLaunchKernelGGL((gru_gates_kernel<256, true>), grid, dim3(256), lds, s, g);
                else if (H == 128) hipLaunchKernelGGL((gru_gates_kernel<128, true>), grid, dim3(256), lds, s, g);
                else hipLaunchKernelGGL((gru_gates_kernel<64, true>), grid, dim3(256), lds, s, g);
            } else {
                if (H == 256) hipLaunchKernelGGL((gru_gates_kernel<256, false>), grid, dim3(256), lds, s, g);
                else if (H == 128) hipLaunchKernelGGL((gru_gates_kernel<128, false>), grid, dim3(256), lds, s, g);
                else hipLaunchKernelGGL((gru_gates_kernel<64, false>), grid, dim3(256), lds, s, g);
            }
            return true;
        }
    }
    if (option(OPT_GRU_BACKGROUND)) {
        GruParams g{};
        if (gru_multi_plan(M, Tmax, H, &g.ncl)) {
            g.gx = gx; g.wh = wh; g.bh = bh; g.y = y; g.Tm = d_Tm; g.off = d_off;
            g.place = d_sync;
            g.sync = d_sync + kMaxGrid;
            g.R = R; g.M = M; g.Tmax = Tmax;
            g.prio = 3;
            if (const char* e = getenv("OCRS_GRU_BG_PRIO")) g.prio = atoi(e);
            if (const char* e = getenv("OCRS_GRU_BG_LAZY")) g.lazy = atoi(e);
            g.spin_limit = 1u << 21;
            g.allow_local = option(OPT_GRU_LOCAL) != 0;
            g.scatter = option(OPT_GRU_SCATTER) != 0;
            gru_assign_multi(h_Tm, M, g.ncl, g.tiles);
            const int UBg = H / 16;
            const dim3 grid(8 * UBg * ((2 * g.ncl + 7) / 8));
            const size_t lds = gru_multi_lds_bytes(H, Tmax);
            OCRS_HIP(hipMemsetAsync(d_sync, 0, gru_persistent_sync_words(M) * sizeof(uint32_t), s));
            if (H == 256) hipLaunchKernelGGL((gru_gates_multi_kernel<256>), grid, dim3(256), lds, s, g);
            else if (H == 128) hipLaunchKernelGGL((gru_gates_multi_kernel<128>), grid, dim3(256), lds, s, g);
            else hipLaunchKernelGGL((gru_gates_multi_kernel<64>), grid, dim3(256), lds, s, g);
            return true;
        }
    }
    GruParams p{};
    p.gx = gx; p.wh = wh; p.bh = bh; p.y = y; p.Tm = d_Tm; p.off = d_off;
    p.place = d_sync;
    p.sync = d_sync + kMaxGrid;
    p.R = R; p.M = M; p.Tmax = Tmax;
    if (H != 256 && H != 128 && H != 64) return false;
    const int UB = H / 16;
    const int wv = gru_waves();
    if (!gru_plan(M, Tmax, H, &p.ncl, gru_resident_capacity(H, false, wv == 16 ? gru_teams_lds_bytes(H, Tmax) : gru_general_lds_bytes(H, Tmax)), wv)) return false;
    gru_assign_tiles(h_Tm, M, p.ncl, p.tiles, wv);
    p.prio = 3;
    if (const char* e = getenv("OCRS_GRU_PRIO")) p.prio = atoi(e);
    p.spin_limit = 1u << 21;  // re-reads of >= ~1 us each: seconds, far beyond any legitimate wait
    const int groups = (2 * p.ncl + 7) / 8;
    const dim3 grid(8 * UB * groups);
    if (grid.x > (unsigned)kMaxGrid) return false;
    p.allow_local = option(OPT_GRU_LOCAL) != 0;
    p.scatter = option(OPT_GRU_SCATTER) != 0;
    const size_t lds = wv == 16 ? gru_teams_lds_bytes(H, Tmax) : gru_general_lds_bytes(H, Tmax);
    OCRS_HIP(hipMemsetAsync(d_sync, 0, gru_persistent_sync_words(M) * sizeof(uint32_t), s));  // (y: gru_persistent_prepare)
    if (wv == 16) {
        if (H == 256) gru_teams_allow_lds<256>(); else if (H == 128) gru_teams_allow_lds<128>(); else gru_teams_allow_lds<64>();
        if (H == 256) hipLaunchKernelGGL((gru_teams_kernel<256>), grid, dim3(1024), lds, s, p);
        else if (H == 128) hipLaunchKernelGGL((gru_teams_kernel<128>), grid, dim3(1024), lds, s, p);
        else hipLaunchKernelGGL((gru_teams_kernel<64>), grid, dim3(1024), lds, s, p);
    } else {
        if (H == 256) hipLaunchKernelGGL((gru_persistent_kernel<256>), grid, dim3(256), lds, s, p);
        else if (H == 128) hipLaunchKernelGGL((gru_persistent_kernel<128>), grid, dim3(256), lds, s, p);
        else hipLaunchKernelGGL((gru_persistent_kernel<64>), grid, dim3(256), lds, s, p);
    }
    return true;
}

}  // namespace k
}  // namespace ocrs
