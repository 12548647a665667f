// The bf16-split contraction of the relaxed / reduced numerics (ocrs_engine_params.numerics), shared by the recognition
// convs (kernels_rec.hip) and the GRU input projections (kernels_nn.hip).
//
// An fp32 value is cut into bf16 terms by round-to-nearest (v_cvt_pk_bf16_f32) of the running residual:
// x = hi + mid + lo with |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|, residual <= 2^-24 |x|.  Every bf16 x bf16 product is exact in
// the fp32 accumulator of v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate); the sum runs in the matrix core's own order,
// NOT in the numeric spec's k-ascending fmaf chain.
//   NP = 3 (relaxed): a.b ~ ah.bh + ah.bm + am.bh + am.bm + ah.bl + al.bh, dropped terms <= 2^-23 |a.b| — fp32-class products;
//                     six bf16 MFMAs of K = 16 replace eight fp32 MFMAs of K = 2 (192 vs 512 matrix-pipe cycles);
//   NP = 2 (reduced): a.b ~ ah.bh + ah.bm + am.bh, dropped terms <= 1.5 x 2^-15 |a.b| (a 16-bit significand, between fp16's 11
//                     and fp32's 24 bits); three MFMAs; the lo planes are neither built nor loaded.
//
// Block tile 128 rows x 128 columns, K in chunks of 16, four waves as 2 x 2 of 64 x 64.  Operands in LDS as planes
// [row or column][16 k] of bf16, 32 bytes per row, the two 16-byte halves of a row swapped on rows with bit 3 set: the
// 16-byte operand reads (lane = row, half of the wave = k half) are bank-conflict free (SQ_LDS_BANK_CONFLICT = 0).
//   A (activations): cut by the thread that stages them (once per block, not once per consuming wave);
//   B (weights): cut once when the model is loaded (split_weights), stored in global memory as the exact LDS image of every
//                (column block, chunk): global_load_lds copies it without touching a register.
//
// Pipeline (split_pipeline).  A chunk's matrix work is 2.7x (5.3x) shorter than in the exact kernels while the memory
// latencies are what they were, so a one-chunk look-ahead no longer covers them.  The weights travel three chunks ahead
// through a ring of four LDS buffers, the activations two to three chunks ahead in two register sets (one per chunk pair,
// used alternately); a half-step ends with the bare barrier instruction behind COUNTED waits — only the copies of the chunk
// that is consumed next must have landed, the younger ones stay in flight (vmcnt retires in order: "at most N outstanding" =
// everything older than the N youngest has arrived).
//
// Measured (profiles/r5_*): the relaxed conv kernels sustain 1.23 PFLOP/s of bf16 work — what the matrix pipe delivers under
// the power budget on real data (MI355X_MICROARCH.md "DVFS give-back": 1.25 PFLOP/s for a tuned 8192^3 GEMM).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace ocrs {
namespace k {
namespace split {

typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int PLANE = BM * 8;                 // floats per plane: 128 rows x 32 bytes
constexpr int RING = 4;                       // weight buffers
constexpr size_t lds_bytes(int np) { return (size_t)(2 + RING) * np * PLANE * sizeof(float); }   // 72 KB (NP 3) / 48 KB (NP 2)
constexpr int image_floats = 3 * PLANE;       // one (column block, chunk) of the weight image: always three planes

// byte offset of (row r, k = 8 * half .. 8 * half + 7) inside a plane
__device__ __forceinline__ int operand_off(int r, int half) { return r * 32 + (((half ^ (r >> 3)) & 1) << 4); }

// four consecutive k (k = 4 * kq .. 4 * kq + 3) of row r -> 8 bytes in each plane of the A buffer at `base`
template <int NP>
__device__ __forceinline__ void commit4(char* base, int r, int kq, float x0, float x1, float x2, float x3) {
    auto cut = [](float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2s{a, b}, bf16x2s)); };   // v_cvt_pk_bf16_f32 (RNE)
    auto lo_f = [](unsigned pk) { return __uint_as_float(pk << 16); };
    auto hi_f = [](unsigned pk) { return __uint_as_float(pk & 0xFFFF0000u); };
    const int off = r * 32 + ((((kq >> 1) ^ (r >> 3)) & 1) << 4) + ((kq & 1) << 3);
    u32x2s ph = {cut(x0, x1), cut(x2, x3)};
    const float r0 = x0 - lo_f(ph[0]), r1 = x1 - hi_f(ph[0]), r2 = x2 - lo_f(ph[1]), r3 = x3 - hi_f(ph[1]);   // exact
    u32x2s pm = {cut(r0, r1), cut(r2, r3)};
    *reinterpret_cast<u32x2s*>(base + off) = ph;
    *reinterpret_cast<u32x2s*>(base + PLANE * 4 + off) = pm;
    if (NP == 3) {
        u32x2s pl = {cut(r0 - lo_f(pm[0]), r1 - hi_f(pm[0])), cut(r2 - lo_f(pm[1]), r3 - hi_f(pm[1]))};
        *reinterpret_cast<u32x2s*>(base + 2 * PLANE * 4 + off) = pl;
    }
}

// this wave's share (a quarter) of one chunk of the weight image -> ring buffer `bdst` (the chunk's base in LDS).
// MUBUF-form LDS-DMA (buffer_load_dwordx4 ... lds), NOT global_load_lds: the FLAT-encoded form "accesses VMEM and LDS" in the
// compiler's wait-count model ("pending flat"), which then answers EVERY later vector-memory dependency of the wave — the
// activation registers a commit reads — with s_waitcnt vmcnt(0): the drain also waits for the weight copy issued a moment
// earlier and the counted waits of split::pipeline never get to matter (round 5's build: tools/counted_waits.py found 105 of its
// 118 counted waits behind such a drain).  The MUBUF form is an ordinary load to that model: every wait counts as written
// (relaxed 334-337 -> 344, reduced 474 -> 482-493 pages/s, ABAB on one box).
template <int NP>
__device__ __forceinline__ void load_weights(const float* img, float* bdst, int wave, int lane) {
    const uint64_t a = reinterpret_cast<uint64_t>(img);   // wave-uniform by construction (blockIdx, chunk): say so, or every load gets a waterfall loop
    const void* u = reinterpret_cast<const void*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)a));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(u), /*stride*/ 0, NP * 4096, 0x00020000);
#pragma unroll
    for (int j = 0; j < NP; j++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(bdst + 1024 * j + wave * 256), 16,
                                                 (1024 * j + wave * 256 + lane * 4) * 4, 0, 0, 0);
}

// C += A . B on the bf16 matrix cores (one 32 x 32 x 16 step); the probe forms exist for tools/hazard_repro.hip and the variant
// library of tools/r6_session.sh only
#if defined(OCRS_PROBE_ALL_AGPR)       // probe builds only: accumulators AND both operands in AGPRs (the MFMA stream touches no VGPR)
#define OCRS_SPLIT_MMA(A, B, C) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(C) : "a"(A), "a"(B))
#elif defined(OCRS_PROBE_ACC_AGPR)      // probe builds only (tools/build_hazard_repro.sh): the accumulators in the AGPR half of the register file
#define OCRS_SPLIT_MMA(A, B, C) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(C) : "v"(A), "v"(B))
#elif defined(OCRS_PROBE_MFMA16)      // probe builds only: the same registers driven through v_mfma_f32_16x16x32_bf16 (NOT the same arithmetic)
#define OCRS_SPLIT_MMA(A, B, C)                                                                                       \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; q_++) {                                                          \
        f32x4s c4_ = {C[4 * q_], C[4 * q_ + 1], C[4 * q_ + 2], C[4 * q_ + 3]};                                  \
        c4_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, c4_, 0, 0, 0);                                      \
        C[4 * q_] = c4_[0]; C[4 * q_ + 1] = c4_[1]; C[4 * q_ + 2] = c4_[2]; C[4 * q_ + 3] = c4_[3];             \
    }
#else
#define OCRS_SPLIT_MMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#endif

// one chunk: the wave's 64 x 64 tile (rows wm * 64 .., columns wn * 64 ..) += A(abase) . B(bbase)
template <int NP>
__device__ __forceinline__ void mma_chunk(const char* abase, const char* bbase, int wm, int wn, int l31, int half, f32x16s (&acc)[2][2]) {
    bf16x8s af[2][3], bfr[2][3];   // (planes NP.. unused)
#if defined(OCRS_PROBE_ALL_AGPR)
    // operand reads straight into AGPRs (ds_read_b128 a[..]); the compiler does not see their latency: one explicit wait
#define OCRS_LDS_A(DST, PTR) asm volatile("ds_read_b128 %0, %1" : "=a"(DST) : "v"((unsigned)(uintptr_t)(PTR)))
#else
#define OCRS_LDS_A(DST, PTR) DST = *reinterpret_cast<const bf16x8s*>(PTR)
#endif
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int off = operand_off(wm * 64 + i * 32 + l31, half);
#pragma unroll
        for (int pl = 0; pl < NP; pl++) OCRS_LDS_A(af[i][pl], abase + pl * PLANE * 4 + off);
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int off = operand_off(wn * 64 + t * 32 + l31, half);
#pragma unroll
        for (int pl = 0; pl < NP; pl++) OCRS_LDS_A(bfr[t][pl], bbase + pl * PLANE * 4 + off);
    }
#if defined(OCRS_PROBE_ALL_AGPR)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#undef OCRS_LDS_A
    // smallest terms first; consecutive MFMAs go to different accumulators (no back-to-back dependency)
#define OCRS_TERM(PA, PB)                                                                                       \
    _Pragma("unroll") for (int t = 0; t < 2; t++) {                                                             \
        OCRS_SPLIT_MMA(af[0][PA], bfr[t][PB], acc[0][t]);                                                             \
        OCRS_SPLIT_MMA(af[1][PA], bfr[t][PB], acc[1][t]);                                                             \
    }
    if (NP == 3) { OCRS_TERM(NP - 1, 0) OCRS_TERM(0, NP - 1) OCRS_TERM(1, 1) }
    OCRS_TERM(1, 0) OCRS_TERM(0, 1) OCRS_TERM(0, 0)
#undef OCRS_TERM
}

// The K loop.  nchunks % 4 == 0.  load_a(k0, d0, d1): the thread's activation values of chunks k0 and k0 + 16 (two float4 row
// passes each) into registers — exactly FOUR vector-memory instructions; load_b(k0, ring): load_weights of chunk k0 into ring
// buffer `ring` — exactly NP vector-memory instructions; commit(abuf, d): cut and store one chunk's values into A buffer
// `abuf`; compute(abuf, ring).  The counted waits below rely on those instruction counts AND on their program order; the
// steady-state body therefore has no conditional loads (the last four chunks are peeled off and drain with full waits), so
// that it compiles to straight-line code whose vector-memory instructions tests/test_counted_waits.py counts in the
// disassembly of the shipped code object (a compiler that splits, merges or reorders one of these loads fails that test,
// not a tolerance).
template <int NP, class LoadA, class LoadB, class Commit, class Compute>
__device__ __forceinline__ void pipeline(int nchunks, LoadA&& load_a, LoadB&& load_b, Commit&& commit, Compute&& compute) {
    f32x4s sa[2][2][2];                     // [set = pair parity][chunk of the pair][row pass]
    // s_waitcnt immediates (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt = bits 6:4, lgkmcnt = bits 11:8)
    // The bare s_barrier builtin is no memory fence to the compiler (IntrNoMem): without the two empty asm statements with a
    // "memory" clobber it hoists the NEXT chunk's first operand reads (ds_read of the weight ring) above the wait and the
    // barrier that guarantee the copy has landed — found in round 6 in conv12_fused_split_kernel's tap loop, which had the
    // same construction: one line's log-probs in ~1 000 requests differed from run to run.
#define OCRS_END_HALF(VMCNT_IMM)                                                                                  \
    do {                                                                                                          \
        asm volatile("" ::: "memory");                                                                            \
        __builtin_amdgcn_s_waitcnt(VMCNT_IMM); __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */              \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
    } while (0)
#define OCRS_DRAIN()                                                                                              \
    do {                                                                                                          \
        asm volatile("" ::: "memory");                                                                            \
        __builtin_amdgcn_s_waitcnt(0x0070); /* vmcnt(0) lgkmcnt(0) */ __builtin_amdgcn_s_barrier();               \
        asm volatile("" ::: "memory");                                                                            \
    } while (0)
    // B(c+1) landed — younger: B(c+2) NP, A(p+1) 4, B(c+3) NP;  B(c+2) landed — younger: A(p+1) 4, B(c+3) NP, B(c+4) NP, A(p+2) 4
    constexpr int kWaitB1 = 0x0F70 | (NP == 3 ? 10 : 8);    // vmcnt(10) / vmcnt(8)
    constexpr int kWaitB2 = 0x0F70 | (NP == 3 ? 14 : 12);   // vmcnt(14) / vmcnt(12)
    // prologue, issued in the order of the steady state so that the counted waits hold from the first half-step on
    load_a(0, sa[0][0], sa[0][1]);
    load_b(0, 0);
    commit(0, sa[0][0]);
    OCRS_DRAIN();
    load_b(1 * BK, 1);
    load_b(2 * BK, 2);
    load_a(2 * BK, sa[1][0], sa[1][1]);
    int c = 0;
    for (; c + 4 < nchunks; c += 4) {   // chunks c .. c+3 = pairs p (register set 0) and p+1 (set 1); every load below exists
        // even chunk c: ring 0 -> fetch ring 3
        load_b((c + 3) * BK, 3);
        compute(0, 0);
        commit(1, sa[0][1]);
        OCRS_END_HALF(kWaitB1);
        // odd chunk c+1: ring 1 -> fetch ring 0; commit chunk c+2 (pair p+1, set 1); set 0 is free: fetch pair p+2 into it
        load_b((c + 4) * BK, 0);
        compute(1, 1);
        commit(0, sa[1][0]);
        load_a((c + 4) * BK, sa[0][0], sa[0][1]);
        OCRS_END_HALF(kWaitB2);
        // even chunk c+2: ring 2 -> fetch ring 1
        load_b((c + 5) * BK, 1);
        compute(0, 2);
        commit(1, sa[1][1]);
        OCRS_END_HALF(kWaitB1);
        // odd chunk c+3: ring 3 -> fetch ring 2; commit chunk c+4 (pair p+2, set 0); fetch pair p+3 into set 1
        load_b((c + 6) * BK, 2);
        compute(1, 3);
        commit(0, sa[0][0]);
        load_a((c + 6) * BK, sa[1][0], sa[1][1]);
        OCRS_END_HALF(kWaitB2);
    }
    // the last four chunks: one weight fetch is left, counted waits would be too lax — drain instead
    load_b((c + 3) * BK, 3);
    compute(0, 0);
    commit(1, sa[0][1]);
    OCRS_DRAIN();
    compute(1, 1);
    commit(0, sa[1][0]);
    OCRS_DRAIN();
    compute(0, 2);
    commit(1, sa[1][1]);
    OCRS_DRAIN();
    compute(1, 3);
    OCRS_DRAIN();
#undef OCRS_END_HALF
#undef OCRS_DRAIN
}

}  // namespace split

// The split image of a row-major [K][N] weight matrix (N % 128 == 0, K % 16 == 0): per (128-column block, 16-row chunk) three
// planes (hi, mid, lo) of [column][16 k] bf16 with the swizzle of the LDS layout.  Host.
void split_weights(const float* w, int K, int N, int ldw, std::vector<uint16_t>* out);

}  // namespace k
}  // namespace ocrs
