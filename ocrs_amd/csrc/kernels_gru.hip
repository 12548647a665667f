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
// Exchange: the data is its own flag.  The host pre-fills the hand-off buffer with the word 0xFFFFFFFF (a NaN no result can
// be: the epilogue maps that one bit pattern to the canonical NaN).  A wave writes its 16 rows x 16 units with 16-byte
// WRITE-THROUGH stores and moves on — no drain, no counter.  The UB waves that need the tile's full state for the next step
// read the previous step with cache-bypassing loads and look at every 32-bit word: any 0xFFFFFFFF left means "not yet
// written", and the wave re-reads (MI355X_MICROARCH.md "inter-workgroup visibility", form R2 — the payload is the
// flag — checked per 32-bit word, so no assumption about the atomicity of wider accesses is made).  The loads of
// the next item are issued before the current item is computed whenever it belongs to another tile, so that by the
// time they are checked the round trip is long over.  No grid barrier: tiles never wait for each other, and nothing
// depends on workgroup placement or order (blocks of a cluster are merely steered to one XCD for speed).  Every
// wait is bounded: on a time-out the kernel raises the error word and returns, the host reports OCRS_ERR_DEVICE.
// The hand-off buffer is NOT the layer output (round 5; it was until then): hx[direction][block][ub]
// [16 rows][16 units], one 16 KB block (at H = 256) per (row tile, step), tile k's blocks starting at tbase[k] = the sum of
// the longer tiles' lengths.  A wave's store instruction writes 1 KB contiguous = eight WHOLE 128-byte lines, and a load
// instruction of a consumer reads 1 KB contiguous.  In the row-major y a line is pieced together from two workgroups'
// partial writes — the L2 has to fetch the rest before it can serve a read — and a load instruction touches 16 lines half
// each; a CU moves bypassing loads at ~10 bytes per clock whatever they hit, so the wasted halves were time
// (tools/gru_round_probe.py: 7.6 -> 6.3 us per step with one tile per wave, 24.7 -> 20.0 with four).  y gets a second, plain store.
//
// MFMA roles.  D = A.B with A = Wh^T (16 units x 4 k, from LDS) and B = h^T (4 k x 16 rows, from
// registers), so a lane ends up with 4 CONSECUTIVE units of one row: the epilogue's gx reads and
// the y store are one 16-byte access per gate / per lane in the natural layouts.  The B operand
// wants lane (row, kq) to hold h[row][4*s + kq]; rows are fetched as 16-byte pieces and turned by a
// 4x4 transpose across the four 16-lane groups (v_permlane16_swap / v_permlane32_swap).
#include <algorithm>
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
    float* hx;           // the hand-off buffer [2][TB][H / 16][16][16], pre-filled with 0xFFFFFFFF words
    const int32_t* Tm;   // [M] sequence length of line m (descending)
    const int32_t* off;  // [Tmax + 1] first packed row of time t
    uint32_t* sync;      // [1] error word; zeroed before the launch
    uint32_t* place;     // [grid] XCD id + 1 of every workgroup, written by the kernel; zeroed before the launch
    int64_t R;
    int M, ncl, Tmax;
    int prio;            // s_setprio level of the waves (0..3)
    int allow_local;     // 0: write-through hand-offs whatever the placement (option gru_local = 0)
    int16_t tiles[kMaxSlots * 4];  // row tiles of wave slot (cluster-in-direction * 4 + wave), longest first; -1 = none
    int32_t tbase[kMaxSlots * 4];  // first hand-off block of tiles[..] (gate-per-wave kernel: of tile [..])
    int TB;                        // hand-off blocks per direction: the sum of the tiles' lengths
    uint32_t spin_limit;
};

// Hand-off accesses: 16-byte raw-buffer loads/stores with the sc1 (agent-scope) cache bit — the store is
// written through to memory, the load bypasses the CU's L1 (MI355X_MICROARCH.md: "`sc1` loads may replace the
// acquire when the producer stored `sc1`").  Buffer intrinsics rather than `volatile` accesses: the compiler
// follows a volatile access with s_waitcnt vmcnt(0), which would serialise the 17 loads of an item.  The buffer
// resources span y and the hand-off buffer (each < 4 GiB, checked on the host); offsets are bytes.
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
    uint32_t hx_off;        // byte offset in hx of the same 4 units
    uint32_t prev_off;      // byte offset in hx of the row's previous state (its piece of ub = 0)
    bool has_prev;          // false: h = 0 (first step / idle lane)
    bool active;
};

// row bookkeeping + the gx operands of the epilogue (independent of the recurrence).  tm = length of the lane's
// row (0: no such row); off_l = the packed-row table off[] in LDS.  No global load here other than gx: a wait on
// one would also wait for the write-through store of the previous item (vmcnt retires in order).
// tbase = first hand-off block of the tile.
template <int H>
__device__ __forceinline__ void issue_meta(const GruParams& p, int dir, int ub, int tile, int tbase, int tm, const int* off_l, int s,
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
    const uint32_t blk = (uint32_t)(dir * p.TB + tbase + s);   // (tile, step) block of this direction
    L.hx_off = (blk * (H / 16) + ub) * 1024u + i16 * 64u + kq * 16u;
    if (L.active) {
        const float* g = p.gx + ((int64_t)dir * p.R + row) * 3 * H + ub * 16 + kq * 4;
        L.gr = *reinterpret_cast<const f32x4*>(g);
        L.gz = *reinterpret_cast<const f32x4*>(g + H);
        L.gn = *reinterpret_cast<const f32x4*>(g + 2 * H);
        if (s > 0) L.prev_off = (blk - 1u) * (H / 16) * 1024u + i16 * 64u;
    }
}

// the previous state of the lane's row (no wait: the loads are checked by state_ready())
template <int H>
__device__ __forceinline__ void issue_state(__amdgpu_buffer_rsrc_t y, int ub, int kq, Loaded<H>& L) {
    constexpr uint32_t kPiece = 1024u;   // bytes between the row's pieces of consecutive 16-unit slices (workgroups)
    if (L.has_prev) {
#pragma unroll
        for (int j = 0; j < H / 16; j++) L.h[j] = load_bypass(y, L.prev_off + j * kPiece + kq * 16);
        L.hp = load_bypass(y, L.prev_off + ub * kPiece + kq * 16);
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
// FAST = the engine's relaxed numerics (ocrs_engine_params.numerics): hardware v_exp_f32 / v_rcp_f32 (1 ulp each, not
// reproducible on a CPU) instead of the spec's fmaf polynomial + IEEE divide.  Measured (profiles/r5_gate_math_ablation.txt):
// 8.9 % of the general kernel's time alone; a cheaper CPU-reproducible spec (degree-6 exp, division-free reciprocal) would
// have bought 2.5 %, so the spec stays as it is.
template <bool FAST>
__device__ __forceinline__ float sigmoid_g(float x) {
    if (FAST) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
    return 1.0f / (1.0f + expf_sl(-x));
}
template <bool FAST>
__device__ __forceinline__ float tanh_g(float x) {
    if (FAST) {
        const float t = __builtin_amdgcn_exp2f((x > 40.0f ? 40.0f : x) * 2.88539008177792682f);   // (clamped: inf * 0 below otherwise)
        return (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f);
    }
    const float t = expf_sl(2.0f * x);
    return (t - 1.0f) / (t + 1.0f);
}

// gates + new state of the lane's 4 units; `store` = this lane's row is live at this step
template <int H, bool FAST>
__device__ __forceinline__ f32x4 epilogue_store(__amdgpu_buffer_rsrc_t y, const GateAcc& a, const f32x4& gr, const f32x4& gz,
                                                const f32x4& gn, const f32x4& hp, uint32_t out_off, bool store, bool local) {
    f32x4 hn;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float rg = sigmoid_g<FAST>(gr[r] + a.r[r]);
        const float zg = sigmoid_g<FAST>(gz[r] + a.z[r]);
        const float ng = tanh_g<FAST>(fmaf(rg, a.n[r], gn[r]));
        const float hv = fmaf(zg, hp[r] - ng, ng);
        hn[r] = __float_as_uint(hv) == kUnwritten ? __uint_as_float(0x7FC00000u) : hv;  // keep the flag word free
    }
    // fire and forget (the data is the flag).  Unconditional instruction: a lane with nothing to store aims past
    // the buffer and the hardware range check drops it — a branch here would let the compiler sink the whole gate
    // arithmetic under it, out of the MFMA shadow.
    const uint32_t o = store ? out_off : 0xFFFFFFF0u;
    if (local) store_local(y, o, hn);  // wave-uniform
    else store_through(y, o, hn);
    return hn;
}

template <int H, bool FAST>
__global__ void __launch_bounds__(256)
gru_persistent_kernel(GruParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds_w[];  // H*48 floats, layout below
    constexpr int UB = H / 16;
    constexpr int WV = 4;
    constexpr bool EARLY = true;
    // blocks of one cluster share blockIdx % 8 (observed: block b runs on XCD b % 8 — speed only)
    const int b = blockIdx.x;
    const int q = b >> 3;
    const int ub = q % UB;
    const int cid = (q / UB) * 8 + (b & 7);
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
    const __amdgpu_buffer_rsrc_t hxb = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)((uint32_t)p.TB * 2u * (H / 16) * 1024u), 0x00020000);
    const int tb0 = p.tbase[slot * 4 + 0], tb1 = p.tbase[slot * 4 + 1], tb2 = p.tbase[slot * 4 + 2], tb3 = p.tbase[slot * 4 + 3];
    auto tbase_of = [&](int i) { return i == 0 ? tb0 : i == 1 ? tb1 : i == 2 ? tb2 : tb3; };
    // Two register sets used alternately (A holds the current item while B receives the next one's loads, then
    // the roles swap): copying a set would make the wave wait for loads that are still in flight.
    int s = 0, i = 0;
#ifdef OCRS_GRU_PROBE   // where an item's cycles go (tools/gru_round_probe.py --once)
    uint64_t pr_tr = 0, pr_mma = 0, pr_gate = 0, pr_wait = 0, pr_items = 0, pr_t0 = __builtin_readcyclecounter();
#endif
    Loaded<H> bufA, bufB;
    issue_meta<H>(p, dir, ub, tile_of(0), tbase_of(0), tmr[0], off_l, 0, i16, kq, bufA);
    issue_state<H>(hxb, ub, kq, bufA);   // step 0: no previous state, h = 0
    // one item: returns 0 = done, 1 = go on, -1 = timed out
    auto item = [&](Loaded<H>& cur, Loaded<H>& nxt) -> int {
        int ns = s, ni = i + 1;
        if (sel(tT, ni) <= s) { ns = s + 1; ni = 0; }
        const bool have_next = sel(tT, ni) > ns;
        const bool early = EARLY && have_next && ni != i;
        // B operand: lane (row, kq) feeds h[row][4*s4 + kq].  Turned BEFORE the next item's loads are issued so
        // that those can land in the registers the pieces leave behind.
#ifdef OCRS_GRU_PROBE
        const uint64_t pt0 = __builtin_readcyclecounter();
#endif
        float w[H / 4];
#pragma unroll
        for (int j = 0; j < H / 16; j++) transpose4(cur.h[j], &w[4 * j]);
        if (have_next) issue_meta<H>(p, dir, ub, tile_of(ni), tbase_of(ni), sel(tmr, ni), off_l, ns, i16, kq, nxt);
        if (early) issue_state<H>(hxb, ub, kq, nxt);
#ifdef OCRS_GRU_PROBE
        asm volatile("" :: "v"(w[0]), "v"(w[H / 4 - 1]));
        const uint64_t pt1 = __builtin_readcyclecounter();
#endif
        const GateAcc acc = mfma_chain<H>(lane, w, lds_w, br, bz, bn);
#ifdef OCRS_GRU_PROBE
        asm volatile("" :: "v"(acc.r), "v"(acc.z), "v"(acc.n));
        const uint64_t pt2 = __builtin_readcyclecounter();
#endif
        // the hand-off store first (what the cluster waits for), then the layer output: a plain store, read by the next launch
        const f32x4 hn = epilogue_store<H, FAST>(hxb, acc, cur.gr, cur.gz, cur.gn, cur.hp, cur.hx_off, cur.active, local);
        store_local(yb, cur.active ? cur.out_off : 0xFFFFFFF0u, hn);
#ifdef OCRS_GRU_PROBE
        const uint64_t pt3 = __builtin_readcyclecounter();
        pr_tr += pt1 - pt0; pr_mma += pt2 - pt1; pr_gate += pt3 - pt2; pr_items++;
#endif
        if (!have_next) return 0;
        if (!early) issue_state<H>(hxb, ub, kq, nxt);
        if (!await_state<H>(p, hxb, ub, kq, nxt)) return -1;
#ifdef OCRS_GRU_PROBE
        pr_wait += __builtin_readcyclecounter() - pt3;
#endif
        s = ns;
        i = ni;
        return 1;
    };
    for (;;) {
        if (item(bufA, bufB) <= 0) break;
        if (item(bufB, bufA) <= 0) break;
    }
#ifdef OCRS_GRU_PROBE
    if (cid == 0 && ub == 0 && lane == 0)
        printf("probe fp32 wave %d local %d tiles %d: items %llu, cycles per item %llu = transposes + issue %llu | chain %llu | gates + stores %llu | wait %llu\n", wave, (int)local,
               (t0 >= 0) + (t1 >= 0) + (t2 >= 0) + (t3 >= 0), (unsigned long long)pr_items, (unsigned long long)((__builtin_readcyclecounter() - pr_t0) / pr_items),
               (unsigned long long)(pr_tr / pr_items), (unsigned long long)(pr_mma / pr_items), (unsigned long long)(pr_gate / pr_items), (unsigned long long)(pr_wait / pr_items));
#endif
}



// ---------------------------------------------------------------------------------------------------------------
// Gate-per-wave variant for requests with so few row tiles that every tile gets a cluster of its own (<= 8 tiles per
// direction at H = 256: a single page).  There the recurrence is a pure latency chain — T dependent steps, nothing to
// interleave — and the longest link of a step is the wave's 192 MFMAs.  Here the three gates of a step run on three
// waves (three SIMDs) at once, 64 MFMAs each, and the fourth wave does the gate arithmetic and the store:
//   waves 0..2 (gate r, z, n):  previous state of the tile (polled as above) -> 64-MFMA chain of their gate (no transposes:
//                               wave 3 stores the hand-off piece already turned, lane (row, kq) holding units 16 ub + 4 j + kq,
//                               its consumers' operands of the k-steps 4 ub + j; one transpose per step instead of 3 x H / 16) ->
//                               accumulators to LDS -> workgroup barrier
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

template <int H, bool FAST>
__global__ void __launch_bounds__(256)
gru_gates_kernel(GruParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds_w[];  // H*48 floats | off[Tmax+1] | exchange | abort
    // ONE workgroup of this kernel per CU, enforced through the register file: the clobber pushes the allocation past
    // 256 registers per lane, i.e. one wave per SIMD.  At its natural 117 registers the hardware puts two workgroups
    // on a CU as soon as other requests' kernels occupy part of the chip; their six polling waves then re-issued the 17
    // state loads back to back and kept that CU's memory pipeline so full that the awaited stores did not get through
    // (waits of seconds in every run with 2+ requests in flight).  await_state now backs off, with which the shared
    // placement works too; one workgroup per CU — what the general kernel's 299 registers impose anyway — stays
    // because it measured 5 % faster under load.
    asm volatile("v_accvgpr_write_b32 a200, 0" ::: "a200");
    constexpr int UB = H / 16;
    const int b = blockIdx.x;
    const int q = b >> 3;
    const int ub = q % UB;
    const int cid = (q / UB) * 8 + (b & 7);
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
    const __amdgpu_buffer_rsrc_t hxb = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)((uint32_t)p.TB * 2u * (H / 16) * 1024u), 0x00020000);
    const uint32_t blk0 = (uint32_t)(dir * p.TB + p.tbase[tile]);   // the tile's hand-off block of step 0 (tbase by tile here)
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
            L.prev_off = L.has_prev ? (blk0 + s - 1u) * (H / 16) * 1024u + i16 * 64u : 0u;
            issue_state<H>(hxb, ub, kq, L);
            if (!await_state<H>(p, hxb, ub, kq, L)) *abort_w = 1;
            float w[H / 4];
#pragma unroll
            for (int j = 0; j < H / 16; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) w[4 * j + e] = L.h[j][e];
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
                const float rg = sigmoid_g<FAST>(gr[r] + a.r[r]);
                const float zg = sigmoid_g<FAST>(gz[r] + a.z[r]);
                const float ng = tanh_g<FAST>(fmaf(rg, a.n[r], gn[r]));
                const float hv = fmaf(zg, hp[r] - ng, ng);
                hn[r] = __float_as_uint(hv) == kUnwritten ? __uint_as_float(0x7FC00000u) : hv;
            }
            const uint32_t o = active ? ((blk0 + s) * (H / 16) + ub) * 1024u + i16 * 64u + kq * 16u : 0xFFFFFFF0u;
            float t[4];
            transpose4(hn, t);   // the hand-off piece in the gate waves' operand order (the 4 lanes of a row are live or idle together)
            const f32x4 ht = {t[0], t[1], t[2], t[3]};
            if (local) store_local(hxb, o, ht);
            else store_through(hxb, o, ht);
            store_local(yb, active ? out_off : 0xFFFFFFF0u, hn);   // the layer output: read by the next launch only
            if (active) hp = hn;    // what the next step would read back as this row's previous state
            gr = ngr; gz = ngz; gn = ngn; out_off = nout_off; active = nactive;
        }
    }
}


}  // namespace

constexpr int kMaxGrid = 4096;
static size_t gru_gates_lds_bytes(int H, int Tmax) {   // Wh slice | off table (padded to 16 bytes) | 3 x 64 x 16 B | abort word
    return (size_t)H * 48 * sizeof(float) + (size_t)(((Tmax + 1) + 3) & ~3) * sizeof(int) + 3 * 64 * 16 + 16;
}
static size_t gru_general_lds_bytes(int H, int Tmax) { return (size_t)H * 48 * sizeof(float) + ((size_t)Tmax + 1) * sizeof(int); }
size_t gru_persistent_sync_words(int) { return kMaxGrid + 1; }  // placement table + error word (last)

// hand-off blocks per direction: every row tile holds one per step of its longest (= first) line
static int64_t gru_tile_blocks(const int32_t* h_Tm, int M) {
    int64_t tb = 0;
    for (int m = 0; m < M; m += 16) tb += h_Tm[m];
    return tb;
}
size_t gru_persistent_exchange_bytes(const int32_t* h_Tm, int M, int H) { return (size_t)2 * (size_t)gru_tile_blocks(h_Tm, M) * 16 * H * sizeof(float); }

// hand-off buffer -> all words "unwritten"; any stream that is ordered before the recurrence (it does not depend on gx)
hipError_t gru_persistent_prepare(float* hx, const int32_t* h_Tm, int M, int H, hipStream_t s) {
    return M > 0 ? hipMemsetAsync(hx, 0xFF, gru_persistent_exchange_bytes(h_Tm, M, H), s) : hipSuccess;
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
    static std::map<int, int> cache;   // key: device, kernel, H
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const int key = dev * 8 + (gates ? 4 : 0) + (H == 256 ? 2 : H == 128 ? 1 : 0);
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
        e = H == 256 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_persistent_kernel<256, false>, 256, lds)
          : H == 128 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_persistent_kernel<128, false>, 256, lds)
                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_persistent_kernel<64, false>, 256, lds);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
    // one workgroup per CU by design (two per CU of the general kernel, 216 registers without the look-ahead loads: the same
    // 18.7 vs 20.0 us per round of 128 equal tiles, 265 vs 269 pages/s in the bench — profiles/r5_gru_kernel_experiments.txt)
    const int cap = prop.multiProcessorCount * (per_cu > 0 ? 1 : 0);
    std::lock_guard<std::mutex> g(mu);
    cache[key] = cap;
    return cap;
}

// Grid geometry: ncl clusters per direction (UB workgroups each); false if the shape is not supported.
// Every workgroup of a cluster must be resident at once, so the first wave of the grid stays within `max_blocks`
// resident workgroups (the device's capacity, at most one per CU: 256 on MI355X); requests of more than 4 tiles per
// wave at that size (> 2048 lines at H = 256) get twice the clusters, whose second half is dispatched as the first
// finishes (see gru_resident_capacity for why that cannot deadlock).
static bool gru_plan(int M, int Tmax, int H, int* ncl, int max_blocks = 256) {
    if (H != 256 && H != 128 && H != 64) return false;
    const int ntiles = (M + 15) / 16;
    const int UB = H / 16;
    max_blocks = max_blocks > 256 ? 256 : max_blocks;
    max_blocks -= max_blocks % (8 * UB);                 // whole groups of 8 clusters
    if (max_blocks < 8 * UB) return false;               // the device cannot hold one group of clusters
    int max_ncl = max_blocks / UB / 2;
    if (ntiles > 16 * max_ncl) max_ncl *= 2;             // a cluster holds 4 waves x 4 tiles
    if (ntiles > 16 * max_ncl || 4 * max_ncl > kMaxSlots) return false;
    *ncl = (ntiles + 3) / 4 < max_ncl ? (ntiles + 3) / 4 : max_ncl;
    return gru_general_lds_bytes(H, Tmax) <= 64 * 1024;
}

// Deal the row tiles (tile k = lines 16k .. 16k+15 of the length-sorted batch, so len[k] >= len[k + 1]) to the
// 4 * ncl waves of a direction, at most 4 per wave.  A wave steps its tiles round-robin; a round of n live tiles
// costs round[n]: measured per kernel with equal-length requests (tools/gru_round_probe.py, profiles/r5_gru_round_probe.json;
// units of 0.1 us, all four waves of the workgroups busy) — one tile cannot hide the store -> visible -> re-read latency of
// the state exchange, two hide most of it, from there every further tile adds a whole item.  A wave's time is the sum over
// its rounds, and the kernel ends with the slowest wave: longest-tile-first greedy on that cost (the longest tiles end up
// alone or with one short partner, the mid-length ones in twos and threes).  Dealing consecutive tiles to a cluster took
// 8.4 ms per layer on 1 232 lines of 100..600 steps, a snake deal 6.5 ms, this 4.7 ms (round 3's kernel).
// kernel: 0 = fp32 (this file), 1 / 2 = kernels_gru_split.hip with 3 / 2 planes.
static const int kRoundCost[3][5] = {{0, 63, 101, 150, 200}, {0, 54, 91, 128, 170}, {0, 44, 68, 99, 135}};
static void gru_assign_tiles(const int32_t* h_Tm, int M, int ncl, int16_t* tiles, int kernel = 0) {
    const int ntiles = (M + 15) / 16, nslots = 4 * ncl;
    const int* round = kRoundCost[kernel >= 0 && kernel < 3 ? kernel : 0];
    for (int i = 0; i < nslots * 4; i++) tiles[i] = -1;
    std::vector<int> cnt(nslots, 0);
    auto cost = [&](int slot, int extra_len) {   // lengths are descending within a slot, extra_len <= all of them
        int64_t tot = 0;
        int len[5], n = cnt[slot];
        for (int i = 0; i < n; i++) len[i] = h_Tm[tiles[slot * 4 + i] * 16];
        if (extra_len > 0) len[n++] = extra_len;
        for (int i = n - 1, below = 0; i >= 0; i--) {   // rounds in which exactly i + 1 tiles are live
            tot += (int64_t)(len[i] - below) * round[i + 1];
            below = len[i];
        }
        return tot;
    };
    for (int k = 0; k < ntiles; k++) {
        const int len = h_Tm[k * 16];
        int best = -1;
        int64_t best_cost = 0;
        for (int sl = 0; sl < nslots; sl++) {
            if (cnt[sl] >= 4) continue;
            const int64_t cs = cost(sl, len);
            if (best < 0 || cs < best_cost) { best = sl; best_cost = cs; }
        }
        tiles[best * 4 + cnt[best]++] = (int16_t)k;
    }
}

// The deal gru_persistent would use for these (descending) line lengths: clusters per direction and, per wave slot
// (cluster * 4 + wave), up to 4 row-tile indices, -1 = none.  Host only (tests).
bool gru_tile_plan(const int32_t* h_Tm, int M, int H, int* ncl, int* waves, int16_t* tiles /* [kMaxSlots * 4] */) {
    for (int i = 0; i < kMaxSlots * 4; i++) tiles[i] = -1;   // the whole buffer: slots beyond 4 * ncl stay "none"
    *waves = 4;
    if (M <= 0 || !gru_plan(M, h_Tm[0], H, ncl, 256)) return false;   // (host only: planned for a 256-CU device)
    gru_assign_tiles(h_Tm, M, *ncl, tiles);
    return true;
}

// The same geometry and deal for a kernel with `cap` resident workgroups (kernels_gru_split.hip shares the decomposition).
bool gru_general_tile_plan(const int32_t* h_Tm, int M, int Tmax, int H, int cap, int* ncl, int16_t* tiles, int kernel) {
    if (M <= 0 || !gru_plan(M, Tmax, H, ncl, cap)) return false;
    if (tiles) gru_assign_tiles(h_Tm, M, *ncl, tiles, kernel);
    return true;
}

bool gru_persistent_supported(const int32_t* h_Tm, int M, int Tmax, int64_t R, int H) {
    int ncl;
    if (H != 256 && H != 128 && H != 64) return false;
    // y and the hand-off buffer are each addressed through one buffer resource: < 4 GiB
    return M > 0 && (uint64_t)R * 2 * H * sizeof(float) < (uint64_t(1) << 32) && (uint64_t)gru_persistent_exchange_bytes(h_Tm, M, H) < (uint64_t(1) << 32) &&
           gru_plan(M, Tmax, H, &ncl, gru_resident_capacity(H, false, gru_general_lds_bytes(H, Tmax)));
}

// gate-per-wave kernel: every tile has a cluster of its own, one workgroup per CU: cap / UB / 2 clusters per direction
// (8 at H = 256 on MI355X: one page).  (Built and dropped: two workgroups per CU for 2-3 page requests; a multi-tile
// variant for the 16-page request, 7.2 vs 4.6 ms per layer — the general kernel's three interleaved chains per wave use
// the matrix cores better; round 4's four-team workgroups, 12 % slower.  profiles/r4_gru_kernel_experiments.txt.)
static bool gru_gates_plan(int M, int Tmax, int H, int* ncl) {
    if (H != 256 && H != 128 && H != 64) return false;
    if (gru_gates_lds_bytes(H, Tmax) > 64 * 1024) return false;
    const int ntiles = (M + 15) / 16, UB = H / 16;
    int cap = gru_resident_capacity(H, true, gru_gates_lds_bytes(H, Tmax));
    cap = cap > 256 ? 256 : cap;
    cap -= cap % (8 * UB);
    if (cap < 8 * UB || ntiles > cap / UB / 2) return false;
    *ncl = ntiles;
    return true;
}

template <int H>
static void launch_gates(bool fast, dim3 grid, size_t lds, hipStream_t s, const GruParams& g) {
    if (fast) hipLaunchKernelGGL((gru_gates_kernel<H, true>), grid, dim3(256), lds, s, g);
    else hipLaunchKernelGGL((gru_gates_kernel<H, false>), grid, dim3(256), lds, s, g);
}
template <int H>
static void launch_general(bool fast, dim3 grid, size_t lds, hipStream_t s, const GruParams& g) {
    if (fast) hipLaunchKernelGGL((gru_persistent_kernel<H, true>), grid, dim3(256), lds, s, g);
    else hipLaunchKernelGGL((gru_persistent_kernel<H, false>), grid, dim3(256), lds, s, g);
}

bool gru_persistent(const float* gx, const float* wh, const float* bh, float* y, float* hx, const int32_t* d_Tm, const int32_t* d_off,
                    const int32_t* h_Tm, int64_t R, int M, int Tmax, int H, uint32_t* d_sync, hipStream_t s) {
    if (M <= 0) return true;
    if (!hx || !gru_persistent_supported(h_Tm, M, Tmax, R, H)) return false;
    const bool fast = option(OPT_NUMERICS) != 0;
    GruParams p{};
    p.gx = gx; p.wh = wh; p.bh = bh; p.y = y; p.hx = hx; p.Tm = d_Tm; p.off = d_off;
    std::vector<int32_t> base((M + 15) / 16 + 1, 0);
    for (size_t k = 0; k + 1 < base.size(); k++) base[k + 1] = base[k] + h_Tm[k * 16];
    p.TB = base.back();
    p.place = d_sync;
    p.sync = d_sync + kMaxGrid;
    p.R = R; p.M = M; p.Tmax = Tmax;
    p.prio = 3;
    p.spin_limit = 1u << 21;  // re-reads of >= ~1 us each: seconds, far beyond any legitimate wait
    p.allow_local = option(OPT_GRU_LOCAL) != 0;
    const int UB = H / 16;
    if (option(OPT_GRU_GATES) && gru_gates_plan(M, Tmax, H, &p.ncl)) {
        for (int k = 0; k < p.ncl; k++) p.tbase[k] = base[k];   // one tile per cluster
        const dim3 grid(8 * UB * ((2 * p.ncl + 7) / 8));
        const size_t lds = gru_gates_lds_bytes(H, Tmax);
        OCRS_HIP(hipMemsetAsync(d_sync, 0, gru_persistent_sync_words(M) * sizeof(uint32_t), s));
        if (H == 256) launch_gates<256>(fast, grid, lds, s, p);
        else if (H == 128) launch_gates<128>(fast, grid, lds, s, p);
        else launch_gates<64>(fast, grid, lds, s, p);
        return true;
    }
    if (!gru_plan(M, Tmax, H, &p.ncl, gru_resident_capacity(H, false, gru_general_lds_bytes(H, Tmax)))) return false;
    gru_assign_tiles(h_Tm, M, p.ncl, p.tiles);
    for (int i = 0; i < 16 * p.ncl; i++) p.tbase[i] = p.tiles[i] >= 0 ? base[p.tiles[i]] : 0;
    const dim3 grid(8 * UB * ((2 * p.ncl + 7) / 8));
    if (grid.x > (unsigned)kMaxGrid) return false;
    const size_t lds = gru_general_lds_bytes(H, Tmax);
    OCRS_HIP(hipMemsetAsync(d_sync, 0, gru_persistent_sync_words(M) * sizeof(uint32_t), s));  // (hx: gru_persistent_prepare)
    if (H == 256) launch_general<256>(fast, grid, lds, s, p);
    else if (H == 128) launch_general<128>(fast, grid, lds, s, p);
    else launch_general<64>(fast, grid, lds, s, p);
    return true;
}

}  // namespace k
}  // namespace ocrs
