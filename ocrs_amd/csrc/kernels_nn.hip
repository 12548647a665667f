// Neural-network kernels of the fixed-graph executor (the `Model::run` side of
// ocrs/src/model.rs:33-40).  Activations are NHWC fp32.
//
// NUMERIC SPEC.  The reference computes in fp32; so does this
// file, and in one canonical order, so that results are bit-identical to the
// CPU oracle:
//   * every contraction is  acc = bias;  acc = fmaf(a_k, b_k, acc)  for k
//     ascending (conv: k = (ky, kx, ci)).  gfx950's v_mfma_f32_32x32x2_f32 is
//     bit-for-bit that fmaf chain (no wider internal accumulation), so the
//     dense contractions run on the matrix cores without changing a bit;
//   * exp / log / sigmoid / tanh are the fixed polynomials below (only fmaf,
//     rint, IEEE divide and exponent-field arithmetic), never libm/ocml.
// Built with -ffp-contract=off: an fma happens only where fmaf() is written.
#include "common.hpp"
#include "kernels.hpp"
#include "split_mfma.hpp"
#include "spec_math.hpp"

namespace ocrs {
namespace k {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------
// GEMM on the fp32 matrix cores:  C = act(A . B + bias), k ascending.
//
// Work decomposition: a block is 4 wavefronts; each wavefront owns 32 rows of
// A and NT 32-column tiles of C (NT accumulators of 16 VGPRs).  A is read
// straight from global/L2 by the lane that needs it (lane l feeds row l&31;
// both half-waves read the same float4 and pick k or k+1 — v_mfma_f32_32x32x2
// wants A[i=l&31][k=l>>5]); B (the weights: small, shared by every row) is
// staged through LDS in k-major chunks so the B operand read is one
// conflict-free ds_read_b32 per MFMA.
//
// Loader variants: dense rows, or the im2col view of a 3x3/pad-1 NHWC
// convolution (K = 9*Cin, zero outside the image).  Epilogue variants: plain
// row-major store, or the ConvTranspose2x2/s2 scatter.
// Roofline: dense 3x3 convs / GRU / linear -> fp32 MFMA (157 TFLOP/s peak);
// the pointwise convs of the detection U-Net have K <= 64 and are HBM-bound
// (A is streamed exactly once).
// ---------------------------------------------------------------------------
constexpr int GEMM_BK = 32;

template <int NT, bool IM2COL, bool CONVT>
__global__ void __launch_bounds__(256) gemm_mfma_kernel(GemmDesc d) {
    extern __shared__ __attribute__((aligned(16))) float lds_b[];  // [GEMM_BK][32*NT]
    constexpr int BN = 32 * NT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int z = blockIdx.z;
    const float* __restrict__ A = d.A + (int64_t)z * d.strideA;
    const float* __restrict__ B = d.B + (int64_t)z * d.strideB;
    const float* __restrict__ bias = d.bias ? d.bias + (int64_t)z * d.strideBias : nullptr;
    float* __restrict__ C = d.C + (int64_t)z * d.strideC;

    const int n0 = blockIdx.y * BN;
    const int64_t row = (int64_t)blockIdx.x * 128 + wave * 32 + l31;
    const bool row_ok = row < d.M;
    const int64_t rowc = row_ok ? row : (int64_t)d.M - 1;

    // im2col row decomposition
    int py = 0, px = 0;
    int64_t img_base = 0;
    if (IM2COL) {
        int64_t hw = (int64_t)d.H * d.W;
        int64_t img = rowc / hw;
        int rem = (int)(rowc - img * hw);
        py = rem / d.W;
        px = rem - py * d.W;
        img_base = img * hw * d.Cin;
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        int col = n0 + t * 32 + l31;
        float bv = (bias && col < d.N) ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = bv;
    }

    for (int k0 = 0; k0 < d.K; k0 += GEMM_BK) {
        const int kc = min(GEMM_BK, d.K - k0);
        // ---- stage B[k0:k0+kc][n0:n0+BN] into LDS (zero-filled outside N / K)
        __syncthreads();
        for (int i = tid; i < GEMM_BK * BN / 4; i += 256) {
            int kk = (i * 4) / BN;
            int nn = (i * 4) % BN;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk < kc) {
                const float* src = B + (int64_t)(k0 + kk) * d.ldb + n0 + nn;
                if (n0 + nn + 3 < d.N && ((d.ldb & 3) == 0) && (((uintptr_t)B & 15) == 0)) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (n0 + nn + 0 < d.N) v.x = src[0];
                    if (n0 + nn + 1 < d.N) v.y = src[1];
                    if (n0 + nn + 2 < d.N) v.z = src[2];
                    if (n0 + nn + 3 < d.N) v.w = src[3];
                }
            }
            *reinterpret_cast<float4*>(&lds_b[kk * BN + nn]) = v;
        }
        __syncthreads();

        // ---- A pointer for this chunk
        const float* arow;
        bool a_ok = true;
        if (IM2COL) {
            int tap = k0 / d.Cin;  // chunks never straddle taps (Cin % GEMM_BK == 0 or Cin % kc == 0)
            int ci0 = k0 - tap * d.Cin;
            int ky = tap / 3, kx = tap - ky * 3;
            int iy = py + ky - 1, ix = px + kx - 1;
            a_ok = (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
            arow = A + img_base + ((int64_t)(a_ok ? iy : 0) * d.W + (a_ok ? ix : 0)) * d.Cin + ci0;
        } else {
            arow = A + rowc * d.lda + k0;
        }

        for (int kk = 0; kk < kc; kk += 4) {
            float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_ok) av = *reinterpret_cast<const float4*>(arow + kk);
            const float a0 = half ? av.y : av.x;
            const float a1 = half ? av.w : av.z;
            const float* b0p = &lds_b[(kk + half) * BN + l31];
            const float* b1p = &lds_b[(kk + 2 + half) * BN + l31];
#pragma unroll
            for (int t = 0; t < NT; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0p[t * 32], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1p[t * 32], acc[t], 0, 0, 0);
        }
    }

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int64_t row_base = (int64_t)blockIdx.x * 128 + wave * 32;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int col = n0 + t * 32 + l31;
        if (col >= d.N) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int64_t rr = row_base + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (rr >= d.M) continue;
            float v = acc[t][r];
            if (d.relu) v = v > 0.0f ? v : 0.0f;
            if (CONVT) {
                // row = input pixel (img, y, x); col = (dy, dx, co)
                int64_t hw = (int64_t)d.H * d.W;
                int64_t img = rr / hw;
                int rem = (int)(rr - img * hw);
                int y = rem / d.W, x = rem - y * d.W;
                int q = col / d.Cout, co = col - q * d.Cout;
                int dy = q >> 1, dx = q & 1;
                C[((img * 2 * d.H + 2 * y + dy) * (2 * (int64_t)d.W) + 2 * x + dx) * d.Cout + co] = v;
            } else {
                C[rr * d.ldc + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Tiled variant for the compute-bound shapes (3x3 convs of the CRNN, GRU input
// projection, Linear): 128 x BN block tile, BK = 32, both operands staged in LDS
// (k-major, so every MFMA operand is one conflict-free ds_read_b32), 4 waves as
// 2 x 2, each owning 64 x BN/2 (2 x BN/64 tiles of 32x32).  The next chunk's
// global loads are issued into registers before the current chunk's MFMAs and
// written to the other LDS buffer afterwards: one barrier per chunk, global
// latency hidden behind 64-128 MFMAs per wave.  K order per accumulator is
// still strictly ascending, so results stay bit-identical.
// Requires K % 32 == 0 (im2col: Cin % 32 == 0).
// ---------------------------------------------------------------------------
constexpr int TG_BM = 128, TG_BK = 16, TG_LDA = TG_BM + 1;

typedef float f32x4v __attribute__((ext_vector_type(4)));

#ifndef OCRS_DIRECT_EPILOGUES
#define OCRS_DIRECT_EPILOGUES 0   // ablation builds (tools/r6_session.sh epiab): 1 = the dword epilogues of rounds 1-5 here and in kernels_rec.hip
#endif
// The row-wise epilogue of gemm_tiled_kernel / gemm_split_kernel (see the comment at its first use): the wave's 64 x 32 NTW
// sub-tile, 32 rows at a time, through `stage` (32 x 32 NTW floats owned by this wave) to C (already at the sub-tile's first
// column) as float4 per lane.  The caller guarantees full columns, ldc % 4 == 0, C 16-byte aligned, and that no wave of the
// block still reads what `stage` overlays.
template <int NTW, class Acc>
__device__ __forceinline__ void store_tile_rows(const Acc (&acc)[2][NTW], float* stage, int lane, bool relu, float* __restrict__ C,
                                                int64_t row0, int64_t M, int ldc) {
    constexpr int WN = 32 * NTW, LPR = WN / 4, RPI = 64 / LPR;
    const int half = lane >> 5, l31 = lane & 31;
    const int srow = lane / LPR, sc4 = (lane % LPR) * 4;
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int t = 0; t < NTW; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float v = acc[i][t][r];
                if (relu) v = v > 0.0f ? v : 0.0f;
                stage[((r & 3) + 8 * (r >> 2) + 4 * half) * WN + t * 32 + l31] = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 32 / RPI; it++) {
            const int row = it * RPI + srow;
            const f32x4v v = *reinterpret_cast<const f32x4v*>(&stage[row * WN + sc4]);
            const int64_t rr = row0 + i * 32 + row;
            if (rr < M) *reinterpret_cast<f32x4v*>(&C[rr * ldc + sc4]) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int BN, bool IM2COL, bool PAIR>
__global__ void __launch_bounds__(256, 4) gemm_tiled_kernel(GemmDesc d) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NTW = BN / 64;  // 32-col tiles per wave
    float* As = lds;                        // [2][TG_BK][TG_LDA]
    float* Bs = lds + 2 * TG_BK * TG_LDA;   // [2][TG_BK][BN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z;
    const float* __restrict__ A = d.A + (int64_t)z * d.strideA;
    const float* __restrict__ B = d.B + (int64_t)z * d.strideB;
    const float* __restrict__ bias = d.bias ? d.bias + (int64_t)z * d.strideBias : nullptr;
    float* __restrict__ C = d.C + (int64_t)z * d.strideC;
    // XCD-aware row-block order (see kernels_rec.hip): consecutive row blocks share im2col rows / A panels
    int mblk, nblk;
    if (d.nfast) {
        // dense A with several column tiles: the ny column tiles of a row tile are consecutive blocks of ONE XCD
        // (block b runs on XCD b % 8), so the A tile is fetched from HBM once and served ny - 1 times by that L2
        // (with the column tile on blockIdx.y every row tile was re-read from memory once per column tile).
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        nblk = q % d.ny;
        mblk = (q / d.ny) * 8 + xcd;
        if (mblk >= d.nx) return;
    } else {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, xq = nwg >> 3, xr = nwg & 7;
        mblk = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
        nblk = blockIdx.y;
    }
    const int64_t m0 = (int64_t)mblk * TG_BM;
    const int n0 = nblk * BN;

    // ---- per-thread load assignments
    // A: 128 rows x TG_BK/4 float4 per chunk; thread handles rows ar + AROWS*j, float4 column akq
    constexpr int AQ = TG_BK / 4, AROWS = 256 / AQ, AV = TG_BM / AROWS;
    const int ar = tid / AQ, akq = tid % AQ;
    const float* arow_ptr[AV];
    int apy[AV], apx[AV];
#pragma unroll
    for (int j = 0; j < AV; j++) {
        int64_t row = m0 + ar + AROWS * j;
        if (row >= d.M) row = d.M - 1;
        if (IM2COL) {
            const int64_t hw = (int64_t)d.H * d.W;
            const int64_t img = row / hw;
            const int rem = (int)(row - img * hw);
            apy[j] = rem / d.W;
            apx[j] = rem - apy[j] * d.W;
            arow_ptr[j] = A + img * hw * d.Cin;
        } else {
            apy[j] = apx[j] = 0;
            arow_ptr[j] = A + row * d.lda;
        }
    }
    // B: TG_BK x BN floats per chunk
    constexpr int BV = TG_BK * BN / 4 / 256;
    const bool b_vec = ((d.ldb & 3) == 0) && (((uintptr_t)B & 15) == 0);

    // dense B with 16-byte rows and N % 4 == 0: unconditional float4 loads from a clamped column (columns past N
    // only feed output columns that are never stored) — no divergent branches in the K loop
    int boff[BV];
#pragma unroll
    for (int j = 0; j < BV; j++) {
        const int idx = tid + 256 * j;
        const int kk = idx / (BN / 4), nn = (idx % (BN / 4)) * 4;
        boff[j] = kk * d.ldb + min(n0 + nn, d.N - 4);
    }
    f32x4v pa[AV], pa2[AV], pb[BV];
    auto prefetch_b_fast = [&](int k0) {
        const float* __restrict__ bk = B + (int64_t)k0 * d.ldb;
#pragma unroll
        for (int j = 0; j < BV; j++) pb[j] = *reinterpret_cast<const f32x4v*>(bk + boff[j]);
    };
    auto prefetch_a_pair = [&](int k0) {  // dense A, chunks k0 and k0 + TG_BK: one full 128-byte line per row
#pragma unroll
        for (int j = 0; j < AV; j++) {
            const float* src = arow_ptr[j] + k0 + akq * 4;
            pa[j] = *reinterpret_cast<const f32x4v*>(src);
            pa2[j] = *reinterpret_cast<const f32x4v*>(src + TG_BK);
        }
    };
    auto prefetch = [&](int k0) {
        if (IM2COL) {
            const int tap = k0 / d.Cin;
            const int ci0 = k0 - tap * d.Cin;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int j = 0; j < AV; j++) {
                const int iy = apy[j] + ky - 1, ix = apx[j] + kx - 1;
                const bool ok = (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                pa[j] = ok ? *reinterpret_cast<const f32x4v*>(arow_ptr[j] + ((int64_t)iy * d.W + ix) * d.Cin + ci0 + akq * 4)
                           : f32x4v{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int j = 0; j < AV; j++) pa[j] = *reinterpret_cast<const f32x4v*>(arow_ptr[j] + k0 + akq * 4);
        }
#pragma unroll
        for (int j = 0; j < BV; j++) {
            const int idx = tid + 256 * j;
            const int kk = idx / (BN / 4), nn = (idx % (BN / 4)) * 4;
            const float* src = B + (int64_t)(k0 + kk) * d.ldb + n0 + nn;
            f32x4v v = {0.f, 0.f, 0.f, 0.f};
            if (b_vec && n0 + nn + 3 < d.N) {
                v = *reinterpret_cast<const f32x4v*>(src);
            } else {
                if (n0 + nn + 0 < d.N) v.x = src[0];
                if (n0 + nn + 1 < d.N) v.y = src[1];
                if (n0 + nn + 2 < d.N) v.z = src[2];
                if (n0 + nn + 3 < d.N) v.w = src[3];
            }
            pb[j] = v;
        }
    };
    auto commit_from = [&](int buf, const f32x4v (&src)[AV]) {
        float* a = As + buf * TG_BK * TG_LDA;
#pragma unroll
        for (int j = 0; j < AV; j++) {
            const int r = ar + AROWS * j;
            a[(akq * 4 + 0) * TG_LDA + r] = src[j].x;
            a[(akq * 4 + 1) * TG_LDA + r] = src[j].y;
            a[(akq * 4 + 2) * TG_LDA + r] = src[j].z;
            a[(akq * 4 + 3) * TG_LDA + r] = src[j].w;
        }
        float* b = Bs + buf * TG_BK * BN;
#pragma unroll
        for (int j = 0; j < BV; j++) {
            const int idx = tid + 256 * j;
            const int kk = idx / (BN / 4), nn = (idx % (BN / 4)) * 4;
            *reinterpret_cast<f32x4v*>(&b[kk * BN + nn]) = pb[j];
        }
    };
    auto commit = [&](int buf) { commit_from(buf, pa); };

    f32x16 acc[2][NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const int col = n0 + wn * (BN / 2) + t * 32 + l31;
        const float bv = (bias && col < d.N) ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[0][t][r] = bv; acc[1][t][r] = bv; }
    }

    const int nchunks = d.K / TG_BK;
    auto compute = [&](int buf) {
        const float* a = As + buf * TG_BK * TG_LDA + wm * 64 + l31;
        const float* b = Bs + buf * TG_BK * BN + wn * (BN / 2) + l31;
#pragma unroll
        for (int kp = 0; kp < TG_BK / 2; kp++) {
            const int kr = 2 * kp + half;
            const float a0 = a[kr * TG_LDA], a1 = a[kr * TG_LDA + 32];
#pragma unroll
            for (int t = 0; t < NTW; t++) {
                const float bt = b[kr * BN + t * 32];
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bt, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bt, acc[1][t], 0, 0, 0);
            }
        }
    };
    if (PAIR) {
        // dense A, K % (2 * TG_BK) == 0, fast B: the chunk-pair loop of conv3x3_ragged_kernel
        prefetch_a_pair(0);
        prefetch_b_fast(0);
        commit_from(0, pa);
        __syncthreads();
        for (int c = 0; c < nchunks; c += 2) {
            prefetch_b_fast((c + 1) * TG_BK);
            compute(0);
            commit_from(1, pa2);
            __syncthreads();
            const bool more = c + 2 < nchunks;
            if (more) {
                prefetch_a_pair((c + 2) * TG_BK);
                prefetch_b_fast((c + 2) * TG_BK);
            }
            compute(1);
            if (more) commit_from(0, pa);
            __syncthreads();
        }
    } else {
        prefetch(0);
        commit(0);
        __syncthreads();
        for (int c = 0; c < nchunks; c++) {
            const int buf = c & 1;
            if (c + 1 < nchunks) prefetch((c + 1) * TG_BK);
            compute(buf);
            if (c + 1 < nchunks) commit(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue (32x32 C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    // Full column tiles with 16-byte rows: a lane's 16 accumulator registers are 16 ROWS of one column, so the direct form
    // below is 16 x NTW dword stores per 32-row half (a row of the wave's sub-tile = two 128-byte runs from two
    // instructions).  Instead the wave turns each 32 x BN/2 half through its own 32 x BN/2 floats of the (now idle) operand
    // tiles — written by column as the MFMA left them, read back by row as float4 — and stores whole rows: 8 dwordx4 stores
    // of 4 x 256 contiguous bytes instead of 32 dword stores (BN = 128).  Wave-private staging: no workgroup barrier; LDS
    // operations of one wave execute in order, the asm statements only keep the COMPILER from moving the reads over the
    // writes (per thread they never alias).  Pure data movement: the bits are the direct form's.  GRU input projection alone
    // (K = 256 / 512, N = 1536: 6 KB stored per 1-2 KB read): 3.20 -> 3.04 ms per launch, 0.73 -> 0.775 of the fp32 MFMA peak (ABAB).
    if (!OCRS_DIRECT_EPILOGUES && n0 + BN <= d.N && (d.ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0) {
        static_assert(4 * 32 * (BN / 2) <= 2 * TG_BK * TG_LDA + 2 * TG_BK * BN, "staging must fit the operand tiles");
        store_tile_rows<NTW>(acc, lds + wave * (32 * (BN / 2)), lane, d.relu != 0, C + n0 + wn * (BN / 2), m0 + wm * 64, d.M, d.ldc);
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const int col = n0 + wn * (BN / 2) + t * 32 + l31;
            if (col >= d.N) continue;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t rr = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (rr >= d.M) continue;
                float v = acc[i][t][r];
                if (d.relu) v = v > 0.0f ? v : 0.0f;
                C[rr * d.ldc + col] = v;
            }
        }
}

template <int BN>
static void launch_gemm_tiled(const GemmDesc& d, hipStream_t s) {
    dim3 grid((unsigned)((d.M + TG_BM - 1) / TG_BM), (unsigned)((d.N + BN - 1) / BN), (unsigned)(d.batch > 0 ? d.batch : 1));
    size_t lds = (size_t)(2 * TG_BK * TG_LDA + 2 * TG_BK * BN) * sizeof(float);
    const bool pair = !d.im2col && (d.K % (2 * TG_BK)) == 0 && (d.ldb & 3) == 0 && (((uintptr_t)d.B) & 15) == 0 &&
                      (d.strideB & 3) == 0 && (d.N & 3) == 0 && d.N >= 4;
    if (d.im2col) hipLaunchKernelGGL((gemm_tiled_kernel<BN, true, false>), grid, dim3(256), lds, s, d);
    else if (pair && grid.y > 1) {   // column tiles of a row tile side by side on one XCD
        GemmDesc e = d;
        e.nfast = 1; e.nx = (int)grid.x; e.ny = (int)grid.y;
        const dim3 g1((unsigned)((grid.x + 7) / 8 * 8 * grid.y), 1, grid.z);
        hipLaunchKernelGGL((gemm_tiled_kernel<BN, false, true>), g1, dim3(256), lds, s, e);
    } else if (pair) hipLaunchKernelGGL((gemm_tiled_kernel<BN, false, true>), grid, dim3(256), lds, s, d);
    else hipLaunchKernelGGL((gemm_tiled_kernel<BN, false, false>), grid, dim3(256), lds, s, d);
}

// ---------------------------------------------------------------------------
// Relaxed / reduced numerics (ocrs_engine_params.numerics) of the dense GEMM C = A . B + bias with pre-cut weights: the
// GRU input projections.  Tiles, operand layout, arithmetic and pipeline: split_mfma.hpp.  A [M][lda] fp32 row-major,
// d.Bsplit the weights' split image per batch, K % 64 == 0, N % 128 == 0.  1-D grid, the column tiles of a row tile side by
// side on one XCD (as gemm_tiled_kernel's nfast form).
// ---------------------------------------------------------------------------
template <int NP>
#if defined(OCRS_PROBE_ACC_AGPR) || defined(OCRS_PROBE_ALL_AGPR)   // probe builds: room for the AGPR copies (three waves per SIMD spill them)
__global__ void __launch_bounds__(256, 2) gemm_split_kernel(GemmDesc d) {
#else
__global__ void __launch_bounds__(256, NP == 3 ? 2 : 3) gemm_split_kernel(GemmDesc d) {
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PL = split::PLANE;
    float* As = lds;                     // [2][NP][PLANE]
    float* Bs = lds + 2 * NP * PL;       // [4][NP][PLANE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int z = blockIdx.z;
    const float* __restrict__ A = d.A + (int64_t)z * d.strideA;
    const float* __restrict__ bias = d.bias ? d.bias + (int64_t)z * d.strideBias : nullptr;
    float* __restrict__ C = d.C + (int64_t)z * d.strideC;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int nblk = q % d.ny;
    const int mblk = (q / d.ny) * 8 + xcd;
    if (mblk >= d.nx) return;
    const int64_t m0 = (int64_t)mblk * split::BM;
    const int n0 = nblk * split::BN;
    const int nchunks = d.K / split::BK;
    const float* __restrict__ img = reinterpret_cast<const float*>(d.Bsplit + (int64_t)z * d.strideBsplit) +
                                    (int64_t)nblk * nchunks * split::image_floats;
    const int ar = tid >> 2, akq = tid & 3;
    const float* arow[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        int64_t row = m0 + ar + 64 * j;
        if (row >= d.M) row = d.M - 1;
        arow[j] = A + row * d.lda + akq * 4;
    }
    split::f32x16s acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int col = n0 + wn * 64 + t * 32 + l31;
        const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[0][t][r] = bv; acc[1][t][r] = bv; }
    }
    split::pipeline<NP>(nchunks,
        [&](int k0, split::f32x4s (&d0)[2], split::f32x4s (&d1)[2]) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                d0[j] = *reinterpret_cast<const split::f32x4s*>(arow[j] + k0);
                d1[j] = *reinterpret_cast<const split::f32x4s*>(arow[j] + k0 + split::BK);
            }
        },
        [&](int k0, int ring) { split::load_weights<NP>(img + (int64_t)(k0 / split::BK) * split::image_floats, Bs + ring * NP * PL, wave, lane); },
        [&](int abuf, const split::f32x4s (&v)[2]) {
            char* base = reinterpret_cast<char*>(As + abuf * NP * PL);
#pragma unroll
            for (int j = 0; j < 2; j++) split::commit4<NP>(base, ar + 64 * j, akq, v[j][0], v[j][1], v[j][2], v[j][3]);
        },
        [&](int abuf, int ring) {
            split::mma_chunk<NP>(reinterpret_cast<const char*>(As + abuf * NP * PL), reinterpret_cast<const char*>(Bs + ring * NP * PL),
                                 wm, wn, l31, half, acc);
        });
    // epilogue: rows as float4 through the operand buffers (gemm_tiled_kernel's; N % 128 == 0 here), else the direct form
    // (32x32 C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    if (!OCRS_DIRECT_EPILOGUES && (d.ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0) {
        // (split::pipeline ends behind a drained barrier: no wave still reads the operand buffers)
        store_tile_rows<2>(acc, lds + wave * (32 * 64), lane, d.relu != 0, C + n0 + wn * 64, m0 + wm * 64, d.M, d.ldc);
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int col = n0 + wn * 64 + t * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t rr = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (rr >= d.M) continue;
                float v = acc[i][t][r];
                if (d.relu) v = v > 0.0f ? v : 0.0f;
                C[rr * d.ldc + col] = v;
            }
        }
}

// false: the shape has no split form (the caller runs the exact kernel)
static bool launch_gemm_split(const GemmDesc& d, int numerics, hipStream_t s) {
    if (!d.Bsplit || d.im2col || d.convt || (d.K % 64) != 0 || (d.N % 128) != 0 || (d.lda & 3) != 0 || (((uintptr_t)d.A) & 15) != 0 ||
        (d.strideA & 3) != 0 || d.M < 1) return false;
    GemmDesc e = d;
    e.nx = (int)((d.M + split::BM - 1) / split::BM);
    e.ny = d.N / split::BN;
    const dim3 grid((unsigned)((e.nx + 7) / 8 * 8 * e.ny), 1, (unsigned)(d.batch > 0 ? d.batch : 1));
    static std::atomic<uint64_t> ok3{0}, ok2{0};
    if (numerics == 2) {
        allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_split_kernel<2>), ok2);
        hipLaunchKernelGGL((gemm_split_kernel<2>), grid, dim3(256), split::lds_bytes(2), s, e);
    } else {
        allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_split_kernel<3>), ok3);
        hipLaunchKernelGGL((gemm_split_kernel<3>), grid, dim3(256), split::lds_bytes(3), s, e);
    }
    return true;
}

template <int NT>
static void launch_gemm(const GemmDesc& d, hipStream_t s) {
    dim3 grid((unsigned)((d.M + 127) / 128), (unsigned)((d.N + 32 * NT - 1) / (32 * NT)), (unsigned)(d.batch > 0 ? d.batch : 1));
    size_t lds = (size_t)GEMM_BK * 32 * NT * sizeof(float);
    if (d.im2col)
        hipLaunchKernelGGL((gemm_mfma_kernel<NT, true, false>), grid, dim3(256), lds, s, d);
    else if (d.convt)
        hipLaunchKernelGGL((gemm_mfma_kernel<NT, false, true>), grid, dim3(256), lds, s, d);
    else
        hipLaunchKernelGGL((gemm_mfma_kernel<NT, false, false>), grid, dim3(256), lds, s, d);
}

// ---------------------------------------------------------------------------
// Small-M variant (the pointwise convs and ConvTransposes of the deep U-Net levels: 864 .. 15 000 rows,
// K = 64 .. 512, N up to 512).  The two kernels above walk K in chunks with a barrier and a fresh
// round of A loads per chunk; with a handful of workgroups that chain of latencies IS the run time
// (30-85 us for 0.05-0.4 GFLOP).  Here a wave owns one 32 x 32 output tile: the whole B panel
// [K][32] is staged in LDS once, the wave's A fragment is fetched 128 k at a time, one chunk
// ahead of the MFMAs that consume it, and the grid is (M/128) x (N/32) workgroups.
// Same chain as everywhere: acc = bias; k ascending (v_mfma_f32_32x32x2_f32).
// ---------------------------------------------------------------------------
constexpr int GS_KC = 128;  // A chunk (k values) held in registers: GS_KC / 2 per lane

template <bool CONVT>
__global__ void __launch_bounds__(256) gemm_small_kernel(GemmDesc d) {
    extern __shared__ __attribute__((aligned(16))) float lds_b[];  // [K][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const float* __restrict__ A = d.A;
    const float* __restrict__ B = d.B;
    float* __restrict__ C = d.C;
    const int n0 = blockIdx.y * 32;
    const int64_t row_base = ((int64_t)blockIdx.x * 4 + wave) * 32;
    const int64_t rowc = row_base + l31 < d.M ? row_base + l31 : (int64_t)d.M - 1;
    const float* __restrict__ arow = A + rowc * d.lda;
    // ---- B panel -> LDS (zero beyond N)
    for (int i = tid; i < d.K * 8; i += 256) {
        const int kk = i >> 3, c4 = (i & 7) * 4;
        const float* src = B + (int64_t)kk * d.ldb + n0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + c4 + 3 < d.N && (d.ldb & 3) == 0 && (((uintptr_t)B) & 15) == 0) {
            v = *reinterpret_cast<const float4*>(src);
        } else {
            if (n0 + c4 + 0 < d.N) v.x = src[0];
            if (n0 + c4 + 1 < d.N) v.y = src[1];
            if (n0 + c4 + 2 < d.N) v.z = src[2];
            if (n0 + c4 + 3 < d.N) v.w = src[3];
        }
        *reinterpret_cast<float4*>(&lds_b[kk * 32 + c4]) = v;
    }
    // ---- A fragment of the first chunk: lane (row l31, half) feeds A[row][2*kk + half]
    float a_cur[GS_KC / 2], a_nxt[GS_KC / 2];
    auto load_a = [&](int k0, float (&a)[GS_KC / 2]) {
#pragma unroll
        for (int j = 0; j < GS_KC / 4; j++) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 + 4 * j < d.K) v = *reinterpret_cast<const float4*>(arow + k0 + 4 * j);
            a[2 * j] = half ? v.y : v.x;
            a[2 * j + 1] = half ? v.w : v.z;
        }
    };
    load_a(0, a_cur);
    f32x16 acc;
    {
        const int col = n0 + l31;
        const float bv = (d.bias && col < d.N) ? d.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = bv;
    }
    __syncthreads();
    for (int k0 = 0; k0 < d.K; k0 += GS_KC) {
        const bool more = k0 + GS_KC < d.K;
        if (more) load_a(k0 + GS_KC, a_nxt);
        const float* bp = &lds_b[(k0 + half) * 32 + l31];
        if (k0 + GS_KC <= d.K) {
#pragma unroll
            for (int kk = 0; kk < GS_KC / 2; kk++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[kk], bp[kk * 64], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < GS_KC / 2; kk++)
                if (k0 + 2 * kk < d.K) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[kk], bp[kk * 64], acc, 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int kk = 0; kk < GS_KC / 2; kk++) a_cur[kk] = a_nxt[kk];
        }
    }
    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col = n0 + l31;
    if (col >= d.N) return;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int64_t rr = row_base + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (rr >= d.M) continue;
        float v = acc[r];
        if (d.relu) v = v > 0.0f ? v : 0.0f;
        if (CONVT) {
            const int64_t hw = (int64_t)d.H * d.W;
            const int64_t img = rr / hw;
            const int rem = (int)(rr - img * hw);
            const int y = rem / d.W, x = rem - y * d.W;
            const int q = col / d.Cout, co = col - q * d.Cout;
            const int dy = q >> 1, dx = q & 1;
            C[((img * 2 * d.H + 2 * y + dy) * (2 * (int64_t)d.W) + 2 * x + dx) * d.Cout + co] = v;
        } else {
            C[rr * d.ldc + col] = v;
        }
    }
}

static bool gemm_small_ok(const GemmDesc& d) {
    return !d.im2col && d.batch <= 1 && d.M <= 16384 && d.K >= 16 && d.K <= 512 && (d.K & 3) == 0 && (d.lda & 3) == 0 &&
           (((uintptr_t)d.A) & 15) == 0;
}

void gemm(const GemmDesc& d, hipStream_t s) {
    if (d.M <= 0 || d.N <= 0) return;
    if (d.Bsplit) {   // relaxed / reduced numerics of the calling engine — for every M: what a line's result is must not depend on
                      // how many rows share its launch (small requests alone vs merged with others: tools/soak_varied.py --numerics)
        const int numerics = option(OPT_NUMERICS);
        if (numerics != 0 && launch_gemm_split(d, numerics, s)) return;
    }
    if (gemm_small_ok(d)) {
        const dim3 grid((unsigned)((d.M + 127) / 128), (unsigned)((d.N + 31) / 32));
        const size_t lds = (size_t)d.K * 32 * sizeof(float);
        if (d.convt) hipLaunchKernelGGL((gemm_small_kernel<true>), grid, dim3(256), lds, s, d);
        else hipLaunchKernelGGL((gemm_small_kernel<false>), grid, dim3(256), lds, s, d);
        return;
    }
    const bool tiled_ok = !d.convt && d.N >= 64 && (d.K % TG_BK) == 0 && d.M >= 256 &&
                          (d.im2col ? (d.Cin % TG_BK) == 0 : ((d.lda & 3) == 0 && ((uintptr_t)d.A & 15) == 0));
    if (tiled_ok) {
        if (d.N <= 64) launch_gemm_tiled<64>(d, s);
        else launch_gemm_tiled<128>(d, s);
        return;
    }
    if (d.N <= 32) launch_gemm<1>(d, s);
    else if (d.N <= 64) launch_gemm<2>(d, s);
    else launch_gemm<4>(d, s);
}

// ---------------------------------------------------------------------------
// Direct convolution for tiny contractions (Cin == 1: first conv of the CRNN,
// first pointwise of the U-Net) and the fallback for shapes the MFMA GEMM does
// not take (Cout % 4 == 0 required).  One thread = one output pixel x 4 output channels; the 8-32
// output channels of a pixel are written by adjacent lanes (coalesced).
// acc = bias; taps (ky, kx) ascending; out-of-image taps contribute fmaf(0,w,acc).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
conv_direct_kernel(const float* __restrict__ x, int n, int h, int w, int cin, const float* __restrict__ wt,
                   const float* __restrict__ bias, int kh, int kw, int cout, int relu, float* __restrict__ y) {
    const int cq = cout >> 2;  // float4 groups per pixel
    const int64_t total = (int64_t)n * h * w * cq;
    const int ph = kh / 2, pw = kw / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cq);
        const int64_t pix = i / cq;
        const int ox = (int)(pix % w);
        const int oy = (int)((pix / w) % h);
        const int64_t img = pix / ((int64_t)w * h);
        float4 acc = *reinterpret_cast<const float4*>(bias + 4 * g);
        for (int ky = 0; ky < kh; ky++)
            for (int kx = 0; kx < kw; kx++) {
                int iy = oy + ky - ph, ix = ox + kx - pw;
                const bool inb = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
                const float* xp = x + ((img * h + (inb ? iy : 0)) * w + (inb ? ix : 0)) * cin;
                const float* wp = wt + (int64_t)(ky * kw + kx) * cin * cout + 4 * g;
                for (int ci = 0; ci < cin; ci++) {
                    float xv = inb ? xp[ci] : 0.0f;
                    float4 wv = *reinterpret_cast<const float4*>(wp + (int64_t)ci * cout);
                    acc.x = fmaf(xv, wv.x, acc.x);
                    acc.y = fmaf(xv, wv.y, acc.y);
                    acc.z = fmaf(xv, wv.z, acc.z);
                    acc.w = fmaf(xv, wv.w, acc.w);
                }
            }
        if (relu) {
            acc.x = acc.x > 0.f ? acc.x : 0.f; acc.y = acc.y > 0.f ? acc.y : 0.f;
            acc.z = acc.z > 0.f ? acc.z : 0.f; acc.w = acc.w > 0.f ? acc.w : 0.f;
        }
        *reinterpret_cast<float4*>(y + pix * cout + 4 * g) = acc;
    }
}

void conv_direct(const float* x, int n, int h, int w, int cin, const float* wt, const float* bias, int kh, int kw,
                 int cout, int relu, float* y, hipStream_t s) {
    int64_t total = (int64_t)n * h * w * (cout / 4);
    int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(conv_direct_kernel, dim3(grid), dim3(256), 0, s, x, n, h, w, cin, wt, bias, kh, kw, cout, relu, y);
}

// ---------------------------------------------------------------------------
// Depthwise 3x3, pad 1, NHWC.  HBM-bound: 4*C bytes in + 4*C bytes out per
// pixel (the 9-tap re-reads are served by L1/L2).  acc = bias; (ky,kx) ascending.
// ---------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256)
dwconv3x3_kernel(const float* __restrict__ x, int n, int h, int w, int c, const float* __restrict__ wt,
                 const float* __restrict__ bias, int relu, float* __restrict__ y) {
    const int cq = c / VEC;
    const int64_t total = (int64_t)n * h * w * cq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cq);
        const int64_t pix = i / cq;
        const int ox = (int)(pix % w);
        const int oy = (int)((pix / w) % h);
        const int64_t img = pix / ((int64_t)w * h);
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; v++) acc[v] = bias[g * VEC + v];
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                int iy = oy + ky - 1, ix = ox + kx - 1;
                bool inb = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
                const float* xp = x + ((img * h + (inb ? iy : 0)) * w + (inb ? ix : 0)) * c + g * VEC;
                const float* wp = wt + (ky * 3 + kx) * c + g * VEC;
                if (VEC == 4) {
                    float4 xv = inb ? *reinterpret_cast<const float4*>(xp) : make_float4(0.f, 0.f, 0.f, 0.f);
                    float4 wv = *reinterpret_cast<const float4*>(wp);
                    acc[0] = fmaf(xv.x, wv.x, acc[0]);
                    acc[1 % VEC] = fmaf(xv.y, wv.y, acc[1 % VEC]);
                    acc[2 % VEC] = fmaf(xv.z, wv.z, acc[2 % VEC]);
                    acc[3 % VEC] = fmaf(xv.w, wv.w, acc[3 % VEC]);
                } else {
                    acc[0] = fmaf(inb ? xp[0] : 0.0f, wp[0], acc[0]);
                }
            }
        float* yp = y + pix * c + g * VEC;
#pragma unroll
        for (int v = 0; v < VEC; v++) {
            float r = acc[v];
            if (relu) r = r > 0.f ? r : 0.f;
            acc[v] = r;
        }
        if (VEC == 4) *reinterpret_cast<float4*>(yp) = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
        else yp[0] = acc[0];
    }
}

// Depthwise 3x3 over the channel concatenation [skip, pad(up)] WITHOUT materialising it (the U-Net's
// Up block: PADCAT feeding the first depthwise conv of its DoubleConv).  Channel group g reads from `skip`
// ([n,h,w,cs]) when g*4 < cs, otherwise from `up` ([n,uh,uw,cu]) shifted by the centred pad; positions outside
// `up` and outside the image contribute fmaf(0, w, acc) exactly as the padded, concatenated tensor would.
__global__ void __launch_bounds__(256)
dwconv3x3_cat_kernel(const float* __restrict__ skip, int n, int h, int w, int cs, const float* __restrict__ up, int uh,
                     int uw, int cu, const float* __restrict__ wt, const float* __restrict__ bias, int relu,
                     float* __restrict__ y) {
    const int c = cs + cu;
    const int cq = c / 4;
    const int py = (h - uh) / 2, px = (w - uw) / 2;
    const int64_t total = (int64_t)n * h * w * cq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cq);
        const int64_t pix = i / cq;
        const int ox = (int)(pix % w);
        const int oy = (int)((pix / w) % h);
        const int64_t img = pix / ((int64_t)w * h);
        const bool from_skip = g * 4 < cs;
        float acc[4];
#pragma unroll
        for (int v = 0; v < 4; v++) acc[v] = bias[g * 4 + v];
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const int iy = oy + ky - 1, ix = ox + kx - 1;
                float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (from_skip) {
                    if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w)
                        xv = *reinterpret_cast<const float4*>(skip + ((img * h + iy) * w + ix) * cs + g * 4);
                } else {
                    const int uy = iy - py, ux = ix - px;
                    if ((unsigned)uy < (unsigned)uh && (unsigned)ux < (unsigned)uw)
                        xv = *reinterpret_cast<const float4*>(up + ((img * uh + uy) * uw + ux) * cu + (g * 4 - cs));
                }
                const float4 wv = *reinterpret_cast<const float4*>(wt + (ky * 3 + kx) * c + g * 4);
                acc[0] = fmaf(xv.x, wv.x, acc[0]);
                acc[1] = fmaf(xv.y, wv.y, acc[1]);
                acc[2] = fmaf(xv.z, wv.z, acc[2]);
                acc[3] = fmaf(xv.w, wv.w, acc[3]);
            }
#pragma unroll
        for (int v = 0; v < 4; v++)
            if (relu) acc[v] = acc[v] > 0.f ? acc[v] : 0.f;
        *reinterpret_cast<float4*>(y + pix * c + g * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

bool dwconv3x3_cat(const float* skip, int n, int h, int w, int cs, const float* up, int uh, int uw, int cu, const float* wt,
                   const float* bias, int relu, float* y, hipStream_t s) {
    if ((cs % 4) != 0 || (cu % 4) != 0 || uh > h || uw > w) return false;
    int64_t total = (int64_t)n * h * w * ((cs + cu) / 4);
    int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(dwconv3x3_cat_kernel, dim3(grid), dim3(256), 0, s, skip, n, h, w, cs, up, uh, uw, cu, wt, bias, relu, y);
    return true;
}

// ---------------------------------------------------------------------------
// Depthwise 3x3 + pointwise 1x1 in one pass (the DoubleConv halves of the detection U-Net at the levels
// where C <= 32).  Same thread layout as dwconv3x3_kernel — one thread = one pixel x 4 channels, so the
// loads stay coalesced — then the T = CIN/4 lanes of a pixel exchange their depthwise outputs with wave
// shuffles and each computes COUT/T of the pointwise outputs.  The intermediate tensor never exists.
// Arithmetic is exactly that of the two kernels: dw  acc = bias; (ky,kx) ascending fmaf, zero taps
// included; [relu]; pw  acc = bias; ci ascending fmaf(dw[ci], W[ci][co], acc); [relu].
// ---------------------------------------------------------------------------
template <int CIN, int COUT>
__global__ void __launch_bounds__(256)
dwpw_fused_kernel(const float* __restrict__ x, int n, int h, int w, const float* __restrict__ wdw,
                  const float* __restrict__ bdw, int relu_dw, const float* __restrict__ wpw,
                  const float* __restrict__ bpw, int relu_pw, float* __restrict__ y) {
    constexpr int T = CIN / 4;      // lanes per pixel
    constexpr int OPT = COUT / T;   // pointwise outputs per lane
    static_assert(CIN % 4 == 0 && COUT % T == 0 && (OPT == 2 || OPT % 4 == 0), "shape");
    __shared__ float s_w[CIN * COUT];
    for (int i = threadIdx.x; i < CIN * COUT; i += 256) s_w[i] = wpw[i];
    __syncthreads();
    const int64_t total = (int64_t)n * h * w * T;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;   // a multiple of 64, hence of T
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int g = (int)(i % T);
        const int64_t pix = i / T;
        const int ox = (int)(pix % w);
        const int oy = (int)((pix / w) % h);
        const int64_t img = pix / ((int64_t)w * h);
        float acc[4];
#pragma unroll
        for (int v = 0; v < 4; v++) acc[v] = bdw[g * 4 + v];
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const int iy = oy + ky - 1, ix = ox + kx - 1;
                const bool inb = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
                const float* xp = x + ((img * h + (inb ? iy : 0)) * w + (inb ? ix : 0)) * CIN + g * 4;
                const float4 xv = inb ? *reinterpret_cast<const float4*>(xp) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 wv = *reinterpret_cast<const float4*>(wdw + (ky * 3 + kx) * CIN + g * 4);
                acc[0] = fmaf(xv.x, wv.x, acc[0]);
                acc[1] = fmaf(xv.y, wv.y, acc[1]);
                acc[2] = fmaf(xv.z, wv.z, acc[2]);
                acc[3] = fmaf(xv.w, wv.w, acc[3]);
            }
        if (relu_dw) {
#pragma unroll
            for (int v = 0; v < 4; v++) acc[v] = acc[v] > 0.f ? acc[v] : 0.f;
        }
        // all CIN depthwise outputs of this pixel, from the T neighbouring lanes
        float d[CIN];
        const int base = lane & ~(T - 1);
#pragma unroll
        for (int gg = 0; gg < T; gg++)
#pragma unroll
            for (int v = 0; v < 4; v++) d[gg * 4 + v] = __shfl(acc[v], base + gg, 64);
        float o[OPT];
#pragma unroll
        for (int q = 0; q < OPT; q++) o[q] = bpw[g * OPT + q];
#pragma unroll
        for (int ci = 0; ci < CIN; ci++)
#pragma unroll
            for (int q = 0; q < OPT; q++) o[q] = fmaf(d[ci], s_w[ci * COUT + g * OPT + q], o[q]);
        if (relu_pw) {
#pragma unroll
            for (int q = 0; q < OPT; q++) o[q] = o[q] > 0.f ? o[q] : 0.f;
        }
        float* yp = y + pix * COUT + g * OPT;
        if (OPT == 2) {
            *reinterpret_cast<float2*>(yp) = make_float2(o[0], o[1 % OPT]);
        } else {
#pragma unroll
            for (int q = 0; q < OPT; q += 4)
                *reinterpret_cast<float4*>(yp + q) = make_float4(o[q], o[(q + 1) % OPT], o[(q + 2) % OPT], o[(q + 3) % OPT]);
        }
    }
}

bool dwpw_fused_supported(int cin, int cout) {
    return (cin == 8 || cin == 16 || cin == 32) && (cout == 8 || cout == 16 || cout == 32);
}

void dwpw_fused(const float* x, int n, int h, int w, int cin, const float* wdw, const float* bdw, int relu_dw, int cout,
                const float* wpw, const float* bpw, int relu_pw, float* y, hipStream_t s) {
    const int64_t total = (int64_t)n * h * w * (cin / 4);
    int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (grid < 1) grid = 1;
#define OCRS_DWPW(CI, CO)                                                                                              \
    if (cin == CI && cout == CO) {                                                                                     \
        hipLaunchKernelGGL((dwpw_fused_kernel<CI, CO>), dim3(grid), dim3(256), 0, s, x, n, h, w, wdw, bdw, relu_dw, wpw, \
                           bpw, relu_pw, y);                                                                           \
        return;                                                                                                        \
    }
    OCRS_DWPW(8, 8) OCRS_DWPW(8, 16) OCRS_DWPW(8, 32) OCRS_DWPW(16, 8) OCRS_DWPW(16, 16) OCRS_DWPW(16, 32)
    OCRS_DWPW(32, 16) OCRS_DWPW(32, 32)
#undef OCRS_DWPW
}

void dwconv3x3(const float* x, int n, int h, int w, int c, const float* wt, const float* bias, int relu, float* y,
               hipStream_t s) {
    const bool v4 = (c % 4) == 0;
    int64_t total = (int64_t)n * h * w * (v4 ? c / 4 : c);
    int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (grid < 1) grid = 1;
    if (v4) hipLaunchKernelGGL((dwconv3x3_kernel<4>), dim3(grid), dim3(256), 0, s, x, n, h, w, c, wt, bias, relu, y);
    else hipLaunchKernelGGL((dwconv3x3_kernel<1>), dim3(grid), dim3(256), 0, s, x, n, h, w, c, wt, bias, relu, y);
}

// ---------------------------------------------------------------------------
// Pools.  m = v0; m = v > m ? v : m  /  s = ((v0+v1)+...)*(1/k).
// ---------------------------------------------------------------------------
template <bool AVG>
__global__ void __launch_bounds__(256)
pool_kernel(const float* __restrict__ x, int n, int h, int w, int c, int kh, int kw, float* __restrict__ y) {
    const int oh = h / kh, ow = w / kw;
    const int64_t total = (int64_t)n * oh * ow * c;
    const float inv = 1.0f / (float)(kh * kw);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        const int64_t pix = i / c;
        const int ox = (int)(pix % ow);
        const int oy = (int)((pix / ow) % oh);
        const int64_t img = pix / ((int64_t)ow * oh);
        const float* xp = x + ((img * h + (int64_t)oy * kh) * w + (int64_t)ox * kw) * c + ch;
        float acc = AVG ? 0.0f : xp[0];
        for (int ky = 0; ky < kh; ky++)
            for (int kx = 0; kx < kw; kx++) {
                float v = xp[((int64_t)ky * w + kx) * c];
                if (AVG) acc = acc + v;
                else acc = v > acc ? v : acc;
            }
        y[i] = AVG ? acc * inv : acc;
    }
}

void maxpool(const float* x, int n, int h, int w, int c, int kh, int kw, float* y, hipStream_t s) {
    int64_t total = (int64_t)n * (h / kh) * (w / kw) * c;
    int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((pool_kernel<false>), dim3(grid), dim3(256), 0, s, x, n, h, w, c, kh, kw, y);
}

void avgpool(const float* x, int n, int h, int w, int c, int kh, int kw, float* y, hipStream_t s) {
    int64_t total = (int64_t)n * (h / kh) * (w / kw) * c;
    int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((pool_kernel<true>), dim3(grid), dim3(256), 0, s, x, n, h, w, c, kh, kw, y);
}

// ---------------------------------------------------------------------------
// Zero-pad x to skip's spatial size (before = d/2, after = d - d/2) and
// concatenate channels [skip, x].
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
padcat_kernel(const float* __restrict__ skip, int n, int sh, int sw, int cs, const float* __restrict__ x, int h,
              int w, int cx, float* __restrict__ y) {
    const int ct = cs + cx;
    const int py = (sh - h) / 2, px = (sw - w) / 2;
    const int64_t total = (int64_t)n * sh * sw * ct;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % ct);
        const int64_t pix = i / ct;
        float v;
        if (ch < cs) {
            v = skip[pix * cs + ch];
        } else {
            const int ox = (int)(pix % sw);
            const int oy = (int)((pix / sw) % sh);
            const int64_t img = pix / ((int64_t)sw * sh);
            const int iy = oy - py, ix = ox - px;
            v = ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) ? x[((img * h + iy) * w + ix) * cx + (ch - cs)] : 0.0f;
        }
        y[i] = v;
    }
}

void padcat(const float* skip, int n, int sh, int sw, int cs, const float* x, int h, int w, int cx, float* y,
            hipStream_t s) {
    int64_t total = (int64_t)n * sh * sw * (cs + cx);
    int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(padcat_kernel, dim3(grid), dim3(256), 0, s, skip, n, sh, sw, cs, x, h, w, cx, y);
}

__global__ void __launch_bounds__(256) sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t count) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = spec_sigmoidf(x[i]);
}

void sigmoid(const float* x, float* y, int64_t count, hipStream_t s) {
    int grid = (int)((count + 255) / 256 < 16384 ? (count + 255) / 256 : 16384);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(sigmoid_kernel, dim3(grid), dim3(256), 0, s, x, y, count);
}

// Pointwise conv with a single output channel (+ fused sigmoid): the U-Net's
// output layer.  y[p] = act(b + chain_c fmaf(x[p,c], w[c])).
__global__ void __launch_bounds__(256)
conv1x1_cout1_kernel(const float* __restrict__ x, int64_t pixels, int cin, const float* __restrict__ wt,
                     const float* __restrict__ bias, int do_sigmoid, float* __restrict__ y) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += (int64_t)gridDim.x * blockDim.x) {
        float acc = bias[0];
        const float* xp = x + p * cin;
        if ((cin & 3) == 0) {
            for (int c = 0; c < cin; c += 4) {
                float4 v = *reinterpret_cast<const float4*>(xp + c);
                acc = fmaf(v.x, wt[c], acc);
                acc = fmaf(v.y, wt[c + 1], acc);
                acc = fmaf(v.z, wt[c + 2], acc);
                acc = fmaf(v.w, wt[c + 3], acc);
            }
        } else {
            for (int c = 0; c < cin; c++) acc = fmaf(xp[c], wt[c], acc);
        }
        y[p] = do_sigmoid ? spec_sigmoidf(acc) : acc;
    }
}

void conv1x1_cout1(const float* x, int64_t pixels, int cin, const float* wt, const float* bias, int do_sigmoid,
                   float* y, hipStream_t s) {
    int grid = (int)((pixels + 255) / 256 < 16384 ? (pixels + 255) / 256 : 16384);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(conv1x1_cout1_kernel, dim3(grid), dim3(256), 0, s, x, pixels, cin, wt, bias, do_sigmoid, y);
}

// [N,1,W,C] -> [W,N,C]
__global__ void __launch_bounds__(256) to_seq_kernel(const float* __restrict__ x, int n, int w, int c, float* __restrict__ y) {
    const int64_t total = (int64_t)n * w * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        const int64_t r = i / c;       // output row = t * n + b
        const int b = (int)(r % n);
        const int t = (int)(r / n);
        y[i] = x[((int64_t)b * w + t) * c + ch];
    }
}

void to_seq(const float* x, int n, int w, int c, float* y, hipStream_t s) {
    int64_t total = (int64_t)n * w * c;
    int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(to_seq_kernel, dim3(grid), dim3(256), 0, s, x, n, w, c, y);
}

// ---------------------------------------------------------------------------
// GRU gates for one time step (PyTorch / ONNX linear_before_reset=1):
//   r = sig(gx_r + gh_r); z = sig(gx_z + gh_z); n = tanh(fmaf(r, gh_n, gx_n));
//   h' = fmaf(z, h - n, n)
// dir = blockIdx.y; forward uses t = step, reverse t = T-1-step.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
gru_gates_kernel(const float* __restrict__ gx, const float* __restrict__ gh, float* __restrict__ h,
                 float* __restrict__ y, int T, int N, int H, int step) {
    const int dir = blockIdx.y;
    const int t = dir ? T - 1 - step : step;
    const int64_t total = (int64_t)N * H;
    const float* gxp = gx + (int64_t)dir * T * N * 3 * H + (int64_t)t * N * 3 * H;
    const float* ghp = gh + (int64_t)dir * N * 3 * H;
    float* hp = h + (int64_t)dir * N * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % H);
        const int64_t b = i / H;
        const float* gxr = gxp + b * 3 * H;
        const float* ghr = ghp + b * 3 * H;
        float r = spec_sigmoidf(gxr[j] + ghr[j]);
        float z = spec_sigmoidf(gxr[H + j] + ghr[H + j]);
        float nn = spec_tanhf(fmaf(r, ghr[2 * H + j], gxr[2 * H + j]));
        float hprev = hp[i];
        float hn = fmaf(z, hprev - nn, nn);
        hp[i] = hn;
        y[((int64_t)t * N + b) * 2 * H + (int64_t)dir * H + j] = hn;
    }
}

void gru_gates(const float* gx, const float* gh, float* h, float* y, int T, int N, int H, int step, hipStream_t s) {
    int64_t total = (int64_t)N * H;
    int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(gru_gates_kernel, dim3(grid, 2), dim3(256), 0, s, gx, gh, h, y, T, N, H, step);
}

// ---------------------------------------------------------------------------
// LogSoftmax over C, then -inf masking (recognition.rs:547-561) + arg-max (first
// maximum, rten decode_greedy).  Spec order: m = max; s = sum_c exp(v_c - m)
// with c ascending from s = 0; out = v - (m + log(s)).
// Rows are staged through LDS so global traffic stays coalesced while each
// lane walks its own row in the canonical order.
// ---------------------------------------------------------------------------
constexpr int LSM_ROWS = 64;

__global__ void __launch_bounds__(LSM_ROWS)
log_softmax_argmax_kernel(const float* __restrict__ logits, int64_t rows, int c, const uint8_t* __restrict__ excl,
                          float* __restrict__ logp, int32_t* __restrict__ labels) {
    extern __shared__ float tile[];  // [LSM_ROWS][c] (c odd or padded -> conflict-free row walks)
    const int cp = (c & 1) ? c : c + 1;
    const int64_t r0 = (int64_t)blockIdx.x * LSM_ROWS;
    const int nrows = (int)min((int64_t)LSM_ROWS, rows - r0);
    for (int i = threadIdx.x; i < nrows * c; i += LSM_ROWS) {
        int rr = i / c, cc = i - rr * c;
        tile[rr * cp + cc] = logits[r0 * c + i];
    }
    __syncthreads();
    if ((int)threadIdx.x < nrows) {
        float* row = tile + threadIdx.x * cp;
        float m = row[0];
        for (int j = 1; j < c; j++) m = row[j] > m ? row[j] : m;
        float ssum = 0.0f;
        for (int j = 0; j < c; j++) ssum = ssum + spec_expf(row[j] - m);
        const float lse = m + spec_logf(ssum);
        // the reference masks excluded labels on the model OUTPUT, then arg-maxes
        const float ninf = -__builtin_huge_valf();
        int best = 0;
        float bv = row[0] - lse;
        row[0] = bv;
        if (excl && excl[0]) bv = ninf;
        for (int j = 1; j < c; j++) {
            float v = row[j] - lse;
            row[j] = v;
            if (excl && excl[j]) v = ninf;
            if (v > bv) { bv = v; best = j; }
        }
        if (labels) labels[r0 + threadIdx.x] = best;
    }
    __syncthreads();
    if (logp)
        for (int i = threadIdx.x; i < nrows * c; i += LSM_ROWS) {
            int rr = i / c, cc = i - rr * c;
            logp[r0 * c + i] = tile[rr * cp + cc];
        }
}

bool log_softmax_argmax(const float* logits, int64_t rows, int c, const uint8_t* d_excluded, float* logp,
                        int32_t* labels, hipStream_t s) {
    if (rows <= 0) return true;
    int cp = (c & 1) ? c : c + 1;
    size_t lds = (size_t)LSM_ROWS * cp * sizeof(float);
    // 64 rows x C floats are staged in LDS: beyond 64 KB the launch needs the opt-in, beyond the CU's 160 KB
    // (an alphabet of more than ~630 classes) this kernel cannot run — the caller reports it as a capacity error
    // instead of a generic launch failure
    if (lds > 160 * 1024) return false;
    if (lds > 64 * 1024) {
        static std::atomic<uint64_t> lds_ok{0};
        allow_dynamic_lds(reinterpret_cast<const void*>(&log_softmax_argmax_kernel), lds_ok);
    }
    int grid = (int)((rows + LSM_ROWS - 1) / LSM_ROWS);
    hipLaunchKernelGGL(log_softmax_argmax_kernel, dim3(grid), dim3(LSM_ROWS), lds, s, logits, rows, c, d_excluded, logp,
                       labels);
    return true;
}

// ---------------------------------------------------------------------------
// Ragged ("packed") sequence batch.  Lines of every width group are sorted by
// sequence length T descending; at time t the active lines are the prefix
// m < active(t), stored at rows off[t] + m.  No padding rows exist, so GEMMs,
// the recurrence and the head touch exactly sum_m T_m rows.
// ---------------------------------------------------------------------------
// features [n, 1, T, C] of one width group -> packed rows
__global__ void __launch_bounds__(256)
to_seq_packed_kernel(const float* __restrict__ x, int n, int T, int c, const int32_t* __restrict__ pos,
                     const int32_t* __restrict__ off, float* __restrict__ y) {
    const int64_t total = (int64_t)n * T * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        const int64_t r = i / c;  // = line * T + t
        const int t = (int)(r % T);
        const int line = (int)(r / T);
        y[((int64_t)off[t] + pos[line]) * c + ch] = x[i];
    }
}

void to_seq_packed(const float* x, int n, int T, int c, const int32_t* d_pos, const int32_t* d_off, float* y,
                   hipStream_t s) {
    int64_t total = (int64_t)n * T * c;
    if (total <= 0) return;
    int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(to_seq_packed_kernel, dim3(grid), dim3(256), 0, s, x, n, T, c, d_pos, d_off, y);
}

// GRU gates on the packed layout, both directions (blockIdx.y).  At step s the
// forward direction is at t = s, the reverse one at t = T_m - 1 - s (per line).
// gx: [2][R][3H]; gh: [2][Mcap][3H]; h: [2][Mcap][H]; y: [R][2H].
__global__ void __launch_bounds__(256)
gru_gates_packed_kernel(const float* __restrict__ gx, const float* __restrict__ gh, float* __restrict__ h,
                        float* __restrict__ y, const int32_t* __restrict__ Tm, const int32_t* __restrict__ off,
                        int64_t R, int Mcap, int active, int H, int step) {
    const int dir = blockIdx.y;
    const int64_t total = (int64_t)active * H;
    const float* gxd = gx + (int64_t)dir * R * 3 * H;
    const float* ghd = gh + (int64_t)dir * Mcap * 3 * H;
    float* hd = h + (int64_t)dir * Mcap * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % H);
        const int m = (int)(i / H);
        const int t = dir ? Tm[m] - 1 - step : step;
        const int64_t row = (int64_t)off[t] + m;
        const float* gxr = gxd + row * 3 * H;
        const float* ghr = ghd + (int64_t)m * 3 * H;
        float r = spec_sigmoidf(gxr[j] + ghr[j]);
        float z = spec_sigmoidf(gxr[H + j] + ghr[H + j]);
        float nn = spec_tanhf(fmaf(r, ghr[2 * H + j], gxr[2 * H + j]));
        float hprev = hd[i];
        float hn = fmaf(z, hprev - nn, nn);
        hd[i] = hn;
        y[row * 2 * H + (int64_t)dir * H + j] = hn;
    }
}

// ---------------------------------------------------------------------------
// Fused GRU step on the packed layout:  gh = hT . Wh + bh  (fp32 MFMA, k ascending),
// then the gates, for BOTH directions, in one launch per time step.
//
// Latency is what matters here (T up to 600 dependent steps), so the work is cut
// into many short MFMA chains: one wave = 16 batch rows x 16 hidden units x 3
// gates on v_mfma_f32_16x16x4_f32 (192 MFMAs of 32 cycles, three independent
// accumulators so the 40-cycle dependent latency is covered).  A block = 4 waves
// (64 rows) sharing one 16-unit slice of Wh (H x 48 floats) staged in LDS.
// The state is kept TRANSPOSED, hT[k][m], so that the A operand (lane l needs
// h[row l&15][k = 4s + (l>>4)]) is a coalesced 64-byte read per 16 lanes, and
// the epilogue (each lane owns 4 consecutive rows of one unit) writes the new
// state as one float4.  State is ping-ponged: other blocks still read step s's
// state while this one writes step s+1's.
// ---------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int H>
__global__ void __launch_bounds__(256)
gru_step_fused_kernel(const float* __restrict__ gx, const float* __restrict__ wh, const float* __restrict__ bh,
                      const float* __restrict__ hT_in, float* __restrict__ hT_out, float* __restrict__ y,
                      const int32_t* __restrict__ Tm, const int32_t* __restrict__ off, int64_t R, int Mcap, int active,
                      int step) {
    extern __shared__ __attribute__((aligned(16))) float lds_w[];  // [H][48]: k-major, (gate, unit) columns
    const int dir = blockIdx.z;
    const int j0 = blockIdx.y * 16;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * 64 + wave * 16;
    const float* __restrict__ whd = wh + (int64_t)dir * H * 3 * H;
    const float* __restrict__ bhd = bh + (int64_t)dir * 3 * H;
    const float* __restrict__ hin = hT_in + (int64_t)dir * H * Mcap;
    float* __restrict__ hout = hT_out + (int64_t)dir * H * Mcap;
    const float* __restrict__ gxd = gx + (int64_t)dir * R * 3 * H;

    const int i16 = lane & 15, kq = lane >> 4;
    const bool wave_active = row0 < active;  // wave-uniform

    // ---- issue every global load this wave needs up front (one memory latency, not 64):
    // (1) the A fragment: lane (i16, kq) feeds h[row0 + i16][k = 4*s + kq] for s = 0..H/4
    float areg[H / 4];
    const int arow = min(row0 + i16, Mcap - 1);
    if (wave_active) {
        const float* ap = hin + (int64_t)kq * Mcap + arow;
#pragma unroll
        for (int s4 = 0; s4 < H / 4; s4++) areg[s4] = ap[(int64_t)s4 * 4 * Mcap];
    }
    // (2) the epilogue operands of this lane's 4 rows of hidden unit j
    const int j = j0 + i16;
    const int mbase = row0 + kq * 4;
    float gxr_[4], gxz_[4], gxn_[4], hprev_[4];
    int64_t yrow[4];
    bool rok[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int m = mbase + r;
        rok[r] = wave_active && m < active;
        gxr_[r] = gxz_[r] = gxn_[r] = hprev_[r] = 0.0f;
        yrow[r] = 0;
        if (rok[r]) {
            const int t = dir ? Tm[m] - 1 - step : step;
            const int64_t row = (int64_t)off[t] + m;
            const float* g = gxd + row * 3 * H;
            gxr_[r] = g[j];
            gxz_[r] = g[H + j];
            gxn_[r] = g[2 * H + j];
            hprev_[r] = hin[(int64_t)j * Mcap + m];
            yrow[r] = row;
        }
    }
    // (3) the Wh slice [:, g*H + j0 .. +16) for g = r,z,n -> LDS
    for (int i = tid; i < H * 12; i += 256) {  // 12 float4 per k-row
        const int k = i / 12, q = i - k * 12;
        const int g = q >> 2, c4 = (q & 3) * 4;
        const float4 v = *reinterpret_cast<const float4*>(whd + (int64_t)k * 3 * H + g * H + j0 + c4);
        *reinterpret_cast<float4*>(&lds_w[k * 48 + g * 16 + c4]) = v;
    }
    __syncthreads();
    if (!wave_active) return;

    f32x4 acc_r, acc_z, acc_n;
    {
        const float br = bhd[j0 + i16], bz = bhd[H + j0 + i16], bn = bhd[2 * H + j0 + i16];
        for (int r = 0; r < 4; r++) { acc_r[r] = br; acc_z[r] = bz; acc_n[r] = bn; }
    }
    const float* bp = &lds_w[kq * 48 + i16];
#pragma unroll
    for (int s4 = 0; s4 < H / 4; s4++) {
        const float a = areg[s4];
        const float* b = bp + s4 * 4 * 48;
        acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[0], acc_r, 0, 0, 0);
        acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[16], acc_z, 0, 0, 0);
        acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[32], acc_n, 0, 0, 0);
    }
    // C/D layout 16x16: col = lane & 15 (hidden unit), row = (lane >> 4) * 4 + reg (batch row)
    float hnew[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float hn = 0.0f;
        if (rok[r]) {
            const float rg = spec_sigmoidf(gxr_[r] + acc_r[r]);
            const float zg = spec_sigmoidf(gxz_[r] + acc_z[r]);
            const float ng = spec_tanhf(fmaf(rg, acc_n[r], gxn_[r]));
            hn = fmaf(zg, hprev_[r] - ng, ng);
            y[yrow[r] * 2 * H + (int64_t)dir * H + j] = hn;
        }
        hnew[r] = hn;
    }
    float* hp = hout + (int64_t)j * Mcap + mbase;
    if (mbase + 3 < Mcap) {
        *reinterpret_cast<float4*>(hp) = make_float4(hnew[0], hnew[1], hnew[2], hnew[3]);
    } else {
        for (int r = 0; r < 4 && mbase + r < Mcap; r++) hp[r] = hnew[r];
    }
}

// hT buffers: [2 dirs][H][Mcap] with Mcap a multiple of 4.  Returns false if H is unsupported.
bool gru_step_fused(const float* gx, const float* wh, const float* bh, const float* hT_in, float* hT_out, float* y,
                    const int32_t* d_Tm, const int32_t* d_off, int64_t R, int Mcap, int active, int H, int step,
                    hipStream_t s) {
    if (active <= 0) return true;
    dim3 grid((active + 63) / 64, H / 16, 2);
    size_t lds = (size_t)H * 48 * sizeof(float);
    if (H == 256)
        hipLaunchKernelGGL((gru_step_fused_kernel<256>), grid, dim3(256), lds, s, gx, wh, bh, hT_in, hT_out, y, d_Tm, d_off,
                           R, Mcap, active, step);
    else if (H == 128)
        hipLaunchKernelGGL((gru_step_fused_kernel<128>), grid, dim3(256), lds, s, gx, wh, bh, hT_in, hT_out, y, d_Tm, d_off,
                           R, Mcap, active, step);
    else if (H == 64)
        hipLaunchKernelGGL((gru_step_fused_kernel<64>), grid, dim3(256), lds, s, gx, wh, bh, hT_in, hT_out, y, d_Tm, d_off, R,
                           Mcap, active, step);
    else
        return false;
    return true;
}

void gru_gates_packed(const float* gx, const float* gh, float* h, float* y, const int32_t* d_Tm, const int32_t* d_off,
                      int64_t R, int Mcap, int active, int H, int step, hipStream_t s) {
    int64_t total = (int64_t)active * H;
    if (total <= 0) return;
    int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(gru_gates_packed_kernel, dim3(grid, 2), dim3(256), 0, s, gx, gh, h, y, d_Tm, d_off, R, Mcap, active,
                       H, step);
}

// Greedy CTC collapse over the packed layout: one lane per line.
__global__ void __launch_bounds__(64)
ctc_collapse_packed_kernel(const int32_t* __restrict__ labels, const int32_t* __restrict__ Tm,
                           const int32_t* __restrict__ off, int M, int Tmax, uint32_t* __restrict__ out_labels,
                           uint32_t* __restrict__ out_pos, int32_t* __restrict__ out_count) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int T = Tm[m];
    int last = 0, cnt = 0;
    for (int t = 0; t < T; t++) {
        int l = labels[(int64_t)off[t] + m];
        if (l == last) continue;
        last = l;
        if (l > 0) {
            out_labels[(int64_t)m * Tmax + cnt] = (uint32_t)l;
            out_pos[(int64_t)m * Tmax + cnt] = (uint32_t)t;
            cnt++;
        }
    }
    out_count[m] = cnt;
}

void ctc_collapse_packed(const int32_t* labels, const int32_t* d_Tm, const int32_t* d_off, int M, int Tmax,
                         uint32_t* out_labels, uint32_t* out_pos, int32_t* out_count, hipStream_t s) {
    if (M <= 0) return;
    hipLaunchKernelGGL(ctc_collapse_packed_kernel, dim3((M + 63) / 64), dim3(64), 0, s, labels, d_Tm, d_off, M, Tmax,
                       out_labels, out_pos, out_count);
}

// Arg-max only (for caller-implemented models whose output is already
// [T,N,C]); excluded labels read as -inf (recognition.rs:547-561).  First maximum.
__global__ void __launch_bounds__(256)
argmax_rows_kernel(const float* __restrict__ x, int64_t rows, int c, const uint8_t* __restrict__ excl,
                   int32_t* __restrict__ labels) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        const float* row = x + r * c;
        const float ninf = -__builtin_huge_valf();
        int best = 0;
        float bv = (excl && excl[0]) ? ninf : row[0];
        for (int j = 1; j < c; j++) {
            float v = (excl && excl[j]) ? ninf : row[j];
            if (v > bv) { bv = v; best = j; }
        }
        labels[r] = best;
    }
}

void argmax_rows(const float* x, int64_t rows, int c, const uint8_t* d_excluded, int32_t* labels, hipStream_t s) {
    if (rows <= 0) return;
    int grid = (int)((rows + 255) / 256 < 4096 ? (rows + 255) / 256 : 4096);
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(grid), dim3(256), 0, s, x, rows, c, d_excluded, labels);
}

// ---------------------------------------------------------------------------
// Greedy CTC collapse: one lane per line.  labels: [T][N].
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
ctc_collapse_kernel(const int32_t* __restrict__ labels, int T, int N, uint32_t* __restrict__ out_labels,
                    uint32_t* __restrict__ out_pos, int32_t* __restrict__ out_count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= N) return;
    int last = 0, cnt = 0;
    for (int t = 0; t < T; t++) {
        int l = labels[(int64_t)t * N + b];
        if (l == last) continue;
        last = l;
        if (l > 0) {
            out_labels[(int64_t)b * T + cnt] = (uint32_t)l;
            out_pos[(int64_t)b * T + cnt] = (uint32_t)t;
            cnt++;
        }
    }
    out_count[b] = cnt;
}

void ctc_collapse(const int32_t* labels, int T, int N, uint32_t* out_labels, uint32_t* out_pos, int32_t* out_count,
                  hipStream_t s) {
    if (N <= 0) return;
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3((N + 63) / 64), dim3(64), 0, s, labels, T, N, out_labels, out_pos,
                       out_count);
}

}  // namespace k
}  // namespace ocrs
