// The BiGRU recurrence of the relaxed / reduced numerics (ocrs_engine_params.numerics): the persistent
// general kernel of kernels_gru.hip — same decomposition (16-row tiles, 16-unit slices, a workgroup of four waves per slice,
// H/16 workgroups per cluster, the host's longest-first deal of tiles to waves), same placement census, same "the data is its
// own flag" hand-off — with the hidden-state contraction on the bf16 matrix cores:
//
//   * the state travels CUT: a wave stores the new state of its 16 rows x 16 units twice — fp32 into the layer's output y
//     (what the next layer reads; no longer the hand-off) and as NP bf16 planes (hi, mid[, lo]: split_mfma.hpp's cut) into an
//     exchange buffer that the host pre-fills with 0xFFFFFFFF words.  A consumer's B operand of v_mfma_f32_16x16x32_bf16 —
//     lane (row, kq) feeds k = 32 s + 8 kq .. + 7 — is then ONE 16-byte load per k-step and plane: no cutting on the
//     consumer's side and none of the 64 lane swaps the fp32 kernel needs for the 16x16x4 layout;
//   * the exchange buffer is laid out for the memory system, not for a reader of rows: hx[plane][direction][block][ub][16 rows]
//     [16 units], one 8 KB block per (row tile, step) — tile k's blocks start at tbase[k] = the sum of the longer tiles'
//     lengths.  A wave's store instruction then writes 512 contiguous bytes = four WHOLE 128-byte lines per plane (in the
//     row-major layout of y a line is pieced together from four workgroups' partial writes, and the L2 has to fetch the rest
//     of it before it can serve a read), and a consumer's load instruction reads 1 KB contiguous (two workgroups' pieces);
//   * Wh is cut once per launch by the workgroup itself while it stages its slice into LDS (NP planes of 3 gates x H/32
//     k-steps x 64 lanes x 16 bytes: 48 KB at NP = 2, H = 256 — what the fp32 slice takes);
//   * per (tile, step) item 3 gates x H/32 k-steps x {3, 6} terms = 72 / 144 MFMAs of 16 pipe cycles instead of 192 of 32;
//   * a lane keeps the previous state of its own four units in registers (it wrote them), in fp32: h' = n + z (h - n) is
//     evaluated on the uncut values; gates with hardware exp / rcp.
//
// A pair of bf16 values equal to 0xFFFFFFFF cannot be data: NaN states are stored as the canonical 0x7FC0 | 0.
#include <algorithm>
#include <atomic>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace ocrs {
namespace k {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) uint32_t gu32;

constexpr int kMaxSlots = 128;
constexpr int kMaxGrid = 4096;
constexpr unsigned kUnwritten = 0xFFFFFFFFu;
constexpr int kAuxSc1 = 16;

struct SplitParams {
    const float* gx;     // [2][R][3H]
    const float* wh;     // [2][H][3H]
    const float* bh;     // [2][3H]
    float* y;            // [R][2H] fp32 layer output
    uint16_t* hx;        // [NP][2][TB][H / 16][16][16] bf16 planes: the hand-off; pre-filled with 0xFFFFFFFF words
    const int32_t* Tm;
    const int32_t* off;
    uint32_t* sync;
    uint32_t* place;
    int64_t R;
    int M, ncl, Tmax;
    int allow_local;
    int TB;              // blocks per direction and plane: the sum of the tiles' lengths
    int16_t tiles[kMaxSlots * 4];
    int32_t tbase[kMaxSlots * 4];   // first block of tiles[..]
    uint32_t spin_limit;
};

template <int H, int NP>
struct Loaded {
    u32x4 hq[H / 32][NP];   // lane (row, kq): 8 consecutive units 32 s + 8 kq .. of the row's previous state, per plane
    f32x4 gr, gz, gn;
    uint32_t y_off;         // byte offset in y of this lane's 4 output units
    uint32_t hx_off;        // byte offset inside a plane of this lane's 4 units (8 bytes)
    uint32_t prev_off;      // byte offset inside a plane of the row's piece of the previous step's block (ub = 0, units 0..7)
    bool has_prev, active;
};

template <int H, int NP>
__device__ __forceinline__ void issue_meta(const SplitParams& p, int dir, int ub, int tile, int tbase, int tm, const int* off_l, int s, int i16,
                                           int kq, Loaded<H, NP>& L) {
    const int m = tile * 16 + i16;
    L.active = tm > s;
    const int t = dir ? tm - 1 - s : s;
    const int64_t row = L.active ? (int64_t)off_l[t] + m : 0;
    const int64_t col = dir * H + ub * 16 + kq * 4;
    L.y_off = (uint32_t)((row * 2 * H + col) * sizeof(float));
    const uint32_t blk = (uint32_t)(dir * p.TB + tbase + s);          // (tile, step) block of this direction
    L.hx_off = (blk * (H / 16) + ub) * 512u + i16 * 32u + kq * 8u;
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    L.gr = L.gz = L.gn = zero;
    L.has_prev = L.active && s > 0;
    L.prev_off = 0;
    if (L.active) {
        const float* g = p.gx + ((int64_t)dir * p.R + row) * 3 * H + ub * 16 + kq * 4;
        L.gr = *reinterpret_cast<const f32x4*>(g);
        L.gz = *reinterpret_cast<const f32x4*>(g + H);
        L.gn = *reinterpret_cast<const f32x4*>(g + 2 * H);
        if (s > 0) L.prev_off = (blk - 1u) * (H / 16) * 512u + i16 * 32u;
    }
}

template <int H, int NP>
__device__ __forceinline__ void issue_state(__amdgpu_buffer_rsrc_t hx, uint32_t plane_bytes, int kq, Loaded<H, NP>& L) {
    if (L.has_prev) {
#pragma unroll
        for (int s = 0; s < H / 32; s++)
#pragma unroll
            for (int pl = 0; pl < NP; pl++)
                L.hq[s][pl] = __builtin_amdgcn_raw_buffer_load_b128(hx, (int)(pl * plane_bytes + L.prev_off + (2 * s + (kq >> 1)) * 512 + (kq & 1) * 16), 0, kAuxSc1);
    } else {
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int s = 0; s < H / 32; s++)
#pragma unroll
            for (int pl = 0; pl < NP; pl++) L.hq[s][pl] = zero;
    }
}

template <int H, int NP>
__device__ __forceinline__ bool state_ready(const Loaded<H, NP>& L) {
    unsigned mx = 0;
#pragma unroll
    for (int s = 0; s < H / 32; s++)
#pragma unroll
        for (int pl = 0; pl < NP; pl++)
#pragma unroll
            for (int e = 0; e < 4; e++) mx = max(mx, L.hq[s][pl][e]);
    return !__any(mx == kUnwritten);
}

template <int H, int NP>
__device__ __forceinline__ bool await_state(const SplitParams& p, __amdgpu_buffer_rsrc_t hx, uint32_t plane_bytes, int kq, Loaded<H, NP>& L) {
    for (uint32_t spins = 0; !state_ready<H, NP>(L); spins++) {
        __builtin_amdgcn_s_sleep(2);
        if (spins >= 4u) {   // back off on long waits (see kernels_gru.hip await_state)
            const uint32_t n = spins < 36u ? (spins >> 2) : 9u;
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
        issue_state<H, NP>(hx, plane_bytes, kq, L);
    }
    return true;
}

__device__ __forceinline__ unsigned cut2(float a, float b) {   // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float hi_f(unsigned pk) { return __uint_as_float(pk & 0xFFFF0000u); }

template <int H, int NP>
__global__ void __launch_bounds__(256)
gru_split_kernel(SplitParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds_w[];   // [3 gates][H/32][NP][64 lanes] x 16 bytes | off[Tmax + 1]
    constexpr int UB = H / 16, S = H / 32, WV = 4;
    constexpr int W_FLOATS = 3 * S * NP * 64 * 4;
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
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if (tid == 0) __hip_atomic_store((gu32*)p.place + cid * UB + ub, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float* __restrict__ whd = p.wh + (int64_t)dir * H * 3 * H;
    const float* __restrict__ bhd = p.bh + (int64_t)dir * 3 * H;
    const int j0 = ub * 16;
    // Wh slice -> LDS, cut: operand piece (gate g, k-step s, lane (unit c, kq')) = Wh[32 s + 8 kq' + 0..7][g H + j0 + c]
    for (int i = tid; i < 3 * S * 64; i += 64 * WV) {
        const int g = i / (S * 64), r0 = i - g * (S * 64), s = r0 >> 6, ln = r0 & 63;
        const int c = ln & 15, kk = ln >> 4;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = whd[(int64_t)(32 * s + 8 * kk + j) * 3 * H + g * H + j0 + c];
        u32x4 ph, pm, pl3;
        float res[8];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            ph[j] = cut2(v[2 * j], v[2 * j + 1]);
            res[2 * j] = v[2 * j] - lo_f(ph[j]);
            res[2 * j + 1] = v[2 * j + 1] - hi_f(ph[j]);
            pm[j] = cut2(res[2 * j], res[2 * j + 1]);
            pl3[j] = cut2(res[2 * j] - lo_f(pm[j]), res[2 * j + 1] - hi_f(pm[j]));
        }
        u32x4* dst = reinterpret_cast<u32x4*>(lds_w) + ((g * S + s) * NP) * 64 + ln;
        dst[0] = ph;
        dst[64] = pm;
        if (NP == 3) dst[128] = pl3;
    }
    int* off_l = reinterpret_cast<int*>(lds_w + W_FLOATS);
    for (int i = tid; i <= p.Tmax; i += 64 * WV) off_l[i] = p.off[i];
    const f32x4 br = *reinterpret_cast<const f32x4*>(bhd + j0 + kq * 4);
    const f32x4 bz = *reinterpret_cast<const f32x4*>(bhd + H + j0 + kq * 4);
    const f32x4 bn = *reinterpret_cast<const f32x4*>(bhd + 2 * H + j0 + kq * 4);
    const int slot = cl * WV + wave;
    const int t0 = p.tiles[slot * 4 + 0], t1 = p.tiles[slot * 4 + 1], t2 = p.tiles[slot * 4 + 2], t3 = p.tiles[slot * 4 + 3];
    auto tile_of = [&](int i) { return i == 0 ? t0 : i == 1 ? t1 : i == 2 ? t2 : t3; };
    int tmr[4], tT[4], tB[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = tile_of(i) * 16 + i16;
        tmr[i] = (tile_of(i) >= 0 && m < p.M) ? p.Tm[m] : 0;
        tT[i] = __builtin_amdgcn_readfirstlane(tmr[i]);
        tB[i] = p.tbase[slot * 4 + i];
    }
    auto sel = [](const int (&a)[4], int i) { return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : i == 3 ? a[3] : 0; };
    __syncthreads();
    __builtin_amdgcn_s_setprio(3);
    if (tT[0] <= 0) return;
    bool local;
    {   // placement census, part 2 (kernels_gru.hip)
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
    const uint32_t plane_bytes = (uint32_t)p.TB * 2u * (H / 16) * 512u;
    const __amdgpu_buffer_rsrc_t hxb = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)(uint32_t)((uint64_t)NP * plane_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t yb = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(uint32_t)(p.R * 2 * H * sizeof(float)), 0x00020000);
    f32x4 hp0 = {0.f, 0.f, 0.f, 0.f}, hp1 = hp0, hp2 = hp0, hp3 = hp0;   // previous state of the lane's own units, per tile slot
    int s = 0, i = 0;
#ifdef OCRS_GRU_PROBE
    uint64_t pr_wait = 0, pr_mma = 0, pr_gate = 0, pr_store = 0, pr_items = 0, pr_polls = 0, pr_t0 = __builtin_readcyclecounter();
#define PROBE_T() __builtin_readcyclecounter()
#endif
    Loaded<H, NP> bufA, bufB;
    issue_meta<H, NP>(p, dir, ub, tile_of(0), tB[0], tmr[0], off_l, 0, i16, kq, bufA);
    issue_state<H, NP>(hxb, plane_bytes, kq, bufA);
    auto item = [&](Loaded<H, NP>& cur, Loaded<H, NP>& nxt) -> int {
        int ns = s, ni = i + 1;
        if (sel(tT, ni) <= s) { ns = s + 1; ni = 0; }
        const bool have_next = sel(tT, ni) > ns;
        const bool early = have_next && ni != i;
        if (have_next) issue_meta<H, NP>(p, dir, ub, tile_of(ni), sel(tB, ni), sel(tmr, ni), off_l, ns, i16, kq, nxt);
        if (early) issue_state<H, NP>(hxb, plane_bytes, kq, nxt);
#ifdef OCRS_GRU_PROBE
        const uint64_t pt0 = PROBE_T();
#endif
        // three interleaved chains (r, z, n); A from LDS one k-step ahead
        f32x4 acc_r = br, acc_z = bz, acc_n = bn;
        const u32x4* ap = reinterpret_cast<const u32x4*>(lds_w) + lane;
#pragma unroll
        for (int ks = 0; ks < S; ks++) {
            bf16x8 a[3][NP], bb[NP];
#pragma unroll
            for (int g = 0; g < 3; g++)
#pragma unroll
                for (int pl = 0; pl < NP; pl++) a[g][pl] = __builtin_bit_cast(bf16x8, ap[((g * S + ks) * NP + pl) * 64]);
#pragma unroll
            for (int pl = 0; pl < NP; pl++) bb[pl] = __builtin_bit_cast(bf16x8, cur.hq[ks][pl]);
#define OCRS_GTERM(PA, PB)                                                                                 \
            acc_r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][PA], bb[PB], acc_r, 0, 0, 0);             \
            acc_z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][PA], bb[PB], acc_z, 0, 0, 0);             \
            acc_n = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][PA], bb[PB], acc_n, 0, 0, 0);
            if (NP == 3) { OCRS_GTERM(NP - 1, 0) OCRS_GTERM(0, NP - 1) OCRS_GTERM(1, 1) }
            OCRS_GTERM(1, 0) OCRS_GTERM(0, 1) OCRS_GTERM(0, 0)
#undef OCRS_GTERM
        }
#ifdef OCRS_GRU_PROBE
        asm volatile("" :: "v"(acc_r), "v"(acc_z), "v"(acc_n));
        const uint64_t pt1 = PROBE_T();
#endif
        // gates (hardware exp / rcp), new state from the lane's own uncut previous state
        const f32x4 hp = i == 0 ? hp0 : i == 1 ? hp1 : i == 2 ? hp2 : hp3;
        f32x4 hn;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((cur.gr[r] + acc_r[r]) * -1.44269504088896341f));
            const float zg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((cur.gz[r] + acc_z[r]) * -1.44269504088896341f));
            const float xn = fmaf(rg, acc_n[r], cur.gn[r]);
            const float tn = __builtin_amdgcn_exp2f((xn > 40.0f ? 40.0f : xn) * 2.88539008177792682f);
            const float ng = (tn - 1.0f) * __builtin_amdgcn_rcpf(tn + 1.0f);
            const float hv = fmaf(zg, hp[r] - ng, ng);
            hn[r] = hv != hv ? __uint_as_float(0x7FC00000u) : hv;   // canonical NaN: its bf16 planes are 0x7FC0 | 0, never the flag
        }
#ifdef OCRS_GRU_PROBE
        asm volatile("" :: "v"(hn));
        const uint64_t pt2 = PROBE_T();
#endif
        const bool st = cur.active;
        if (st) {   // (wave-uniform per 16-lane row group is not guaranteed: per-lane predication)
            if (i == 0) hp0 = hn; else if (i == 1) hp1 = hn; else if (i == 2) hp2 = hn; else hp3 = hn;
        }
        // fp32 into the layer output (read by the next launch only), the cut planes into the hand-off buffer
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hn), yb, (int)(st ? cur.y_off : 0xFFFFFFF0u), 0, 0);
        u32x2 ph = {cut2(hn[0], hn[1]), cut2(hn[2], hn[3])};
        const float r0 = hn[0] - lo_f(ph[0]), r1 = hn[1] - hi_f(ph[0]), r2 = hn[2] - lo_f(ph[1]), r3 = hn[3] - hi_f(ph[1]);
        u32x2 pm = {cut2(r0, r1), cut2(r2, r3)};
        if (hn[0] != hn[0] || hn[1] != hn[1]) pm[0] = 0u;   // NaN - NaN: keep the mid plane clean
        if (hn[2] != hn[2] || hn[3] != hn[3]) pm[1] = 0u;
        const uint32_t o = st ? cur.hx_off : 0xFFFFFFF0u;
        if (local) {
            __builtin_amdgcn_raw_buffer_store_b64(ph, hxb, (int)o, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(pm, hxb, (int)(st ? plane_bytes + cur.hx_off : 0xFFFFFFF0u), 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b64(ph, hxb, (int)o, 0, kAuxSc1);
            __builtin_amdgcn_raw_buffer_store_b64(pm, hxb, (int)(st ? plane_bytes + cur.hx_off : 0xFFFFFFF0u), 0, kAuxSc1);
        }
        if (NP == 3) {
            u32x2 pl3 = {cut2(r0 - lo_f(pm[0]), r1 - hi_f(pm[0])), cut2(r2 - lo_f(pm[1]), r3 - hi_f(pm[1]))};
            if (hn[0] != hn[0] || hn[1] != hn[1]) pl3[0] = 0u;
            if (hn[2] != hn[2] || hn[3] != hn[3]) pl3[1] = 0u;
            const uint32_t o3 = st ? 2u * plane_bytes + cur.hx_off : 0xFFFFFFF0u;
            if (local) __builtin_amdgcn_raw_buffer_store_b64(pl3, hxb, (int)o3, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b64(pl3, hxb, (int)o3, 0, kAuxSc1);
        }
#ifdef OCRS_GRU_PROBE
        const uint64_t pt3 = PROBE_T();
        pr_mma += pt1 - pt0; pr_gate += pt2 - pt1; pr_store += pt3 - pt2; pr_items++;
#endif
        if (!have_next) return 0;
        if (!early) issue_state<H, NP>(hxb, plane_bytes, kq, nxt);
#ifdef OCRS_GRU_PROBE
        if (!state_ready<H, NP>(nxt)) pr_polls++;
#endif
        if (!await_state<H, NP>(p, hxb, plane_bytes, kq, nxt)) return -1;
#ifdef OCRS_GRU_PROBE
        pr_wait += PROBE_T() - pt3;
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
    if (cid < 2 && ub == 0 && lane == 0)
        printf("probe cid %d wave %d local %d tiles %d: items %llu total %llu | mma %llu gate %llu store %llu wait %llu (first-miss %llu) cycles per item\n", cid, wave, (int)local,
               (t0 >= 0) + (t1 >= 0) + (t2 >= 0) + (t3 >= 0), (unsigned long long)pr_items, (unsigned long long)((PROBE_T() - pr_t0) / (pr_items ? pr_items : 1)),
               (unsigned long long)(pr_mma / pr_items), (unsigned long long)(pr_gate / pr_items), (unsigned long long)(pr_store / pr_items),
               (unsigned long long)(pr_wait / pr_items), (unsigned long long)pr_polls);
#endif
}

template <int H, int NP>
int resident_capacity(size_t lds) {   // see gru_resident_capacity (kernels_gru.hip): one workgroup per CU, or no plan
    static std::atomic<uint64_t> ok{0};
    allow_dynamic_lds(reinterpret_cast<const void*>(&gru_split_kernel<H, NP>), ok);
    static std::atomic<int> cache[64];   // per device ordinal; 0 = not yet known, -1 = none
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int cap = cache[dev].load(std::memory_order_relaxed);
    if (cap == 0) {
        hipDeviceProp_t prop;
        int per_cu = 0;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_split_kernel<H, NP>, 256, lds) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        cap = per_cu > 0 ? prop.multiProcessorCount : -1;
        cache[dev].store(cap, std::memory_order_relaxed);
    }
    return cap > 0 ? cap : 0;
}

size_t split_lds_bytes(int H, int np, int Tmax) { return (size_t)3 * (H / 32) * np * 1024 + ((size_t)Tmax + 1) * sizeof(int); }

int capacity(int H, int np, size_t lds) {
    if (np == 2) return H == 256 ? resident_capacity<256, 2>(lds) : H == 128 ? resident_capacity<128, 2>(lds) : resident_capacity<64, 2>(lds);
    return H == 256 ? resident_capacity<256, 3>(lds) : H == 128 ? resident_capacity<128, 3>(lds) : resident_capacity<64, 3>(lds);
}

template <int H, int NP>
void launch(const SplitParams& p, size_t lds, hipStream_t s) {
    const dim3 grid(8 * (H / 16) * ((2 * p.ncl + 7) / 8));
    hipLaunchKernelGGL((gru_split_kernel<H, NP>), grid, dim3(256), lds, s, p);
}

}  // namespace

// blocks per direction: every row tile holds one per step of its longest (= first) line
static int64_t tile_blocks(const int32_t* h_Tm, int M) {
    int64_t tb = 0;
    for (int m = 0; m < M; m += 16) tb += h_Tm[m];
    return tb;
}
size_t gru_split_exchange_bytes(const int32_t* h_Tm, int M, int H, int np) { return (size_t)np * 2 * (size_t)tile_blocks(h_Tm, M) * 16 * H * 2; }

bool gru_split_supported(const int32_t* h_Tm, int M, int Tmax, int64_t R, int H, int np) {
    if (M <= 0 || (H != 256 && H != 128 && H != 64) || (np != 2 && np != 3)) return false;
    // y and the exchange buffer are each addressed through one buffer resource: < 4 GiB
    if ((uint64_t)gru_split_exchange_bytes(h_Tm, M, H, np) >= (uint64_t(1) << 32) || (uint64_t)R * 2 * H * sizeof(float) >= (uint64_t(1) << 32)) return false;
    const size_t lds = split_lds_bytes(H, np, Tmax);
    if (lds > 150 * 1024) return false;
    int ncl = 0;
    if (!gru_general_tile_plan(nullptr, M, Tmax, H, capacity(H, np, lds), &ncl, nullptr, 0)) return false;
    return 8 * (H / 16) * ((2 * ncl + 7) / 8) <= kMaxGrid;
}

// every word of the exchange buffer "unwritten"; any stream ordered before the recurrence
hipError_t gru_split_prepare(uint16_t* hx, const int32_t* h_Tm, int M, int H, int np, hipStream_t s) {
    return M > 0 ? hipMemsetAsync(hx, 0xFF, gru_split_exchange_bytes(h_Tm, M, H, np), s) : hipSuccess;
}

// false: no plan for this shape on this device, nothing launched (gru_split_supported says so beforehand)
bool gru_persistent_split(const float* gx, const float* wh, const float* bh, float* y, uint16_t* hx, const int32_t* d_Tm, const int32_t* d_off,
                          const int32_t* h_Tm, int64_t R, int M, int Tmax, int H, int np, uint32_t* d_sync, hipStream_t s) {
    if (M <= 0) return true;
    if (!hx || !gru_split_supported(h_Tm, M, Tmax, R, H, np)) return false;
    SplitParams p{};
    p.gx = gx; p.wh = wh; p.bh = bh; p.y = y; p.hx = hx; p.Tm = d_Tm; p.off = d_off;
    p.place = d_sync;
    p.sync = d_sync + kMaxGrid;
    p.R = R; p.M = M; p.Tmax = Tmax;
    p.allow_local = option(OPT_GRU_LOCAL) != 0;
    p.spin_limit = 1u << 21;
    const size_t lds = split_lds_bytes(H, np, Tmax);
    if (!gru_general_tile_plan(h_Tm, M, Tmax, H, capacity(H, np, lds), &p.ncl, p.tiles, np == 3 ? 1 : 2)) return false;
    {
        std::vector<int32_t> base((M + 15) / 16 + 1, 0);
        for (size_t k = 0; k + 1 < base.size(); k++) base[k + 1] = base[k] + h_Tm[k * 16];
        p.TB = base.back();
        for (int i = 0; i < kMaxSlots * 4; i++) p.tbase[i] = i < 16 * p.ncl && p.tiles[i] >= 0 ? base[p.tiles[i]] : 0;
    }
    OCRS_HIP(hipMemsetAsync(d_sync, 0, ((size_t)kMaxGrid + 1) * sizeof(uint32_t), s));
    if (np == 2) { if (H == 256) launch<256, 2>(p, lds, s); else if (H == 128) launch<128, 2>(p, lds, s); else launch<64, 2>(p, lds, s); }
    else { if (H == 256) launch<256, 3>(p, lds, s); else if (H == 128) launch<128, 3>(p, lds, s); else launch<64, 3>(p, lds, s); }
    return true;
}

}  // namespace k
}  // namespace ocrs
