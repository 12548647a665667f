// CTC prefix beam search on the GPU — rten::ctc::CtcDecoder::decode_beam as called at
// ocrs/src/recognition.rs:512-514 (DecodeMethod::BeamSearch; width 100 from ocrs-cli/src/main.rs:403-404),
// on the packed log-probabilities the recognition head leaves in HBM.  One workgroup per text line.
//
// Same function as the host's ctc_beam_search (ctc_beam.cpp), same float64 log-sum-exp (beam_math.hpp), same
// tie rules, so the steps (label, position) are identical.  Per time step, with W beams and C classes:
//   * the W*C candidate slots live in LDS as order-preserving 64-bit keys; slot index = the candidate's
//     insertion key: slot i*C + c is the extension of beam i by label c, slot i*C the entry of beam i's own
//     prefix — unless beam i's parent prefix is an EARLIER beam p, then the parent's extension inserted that
//     entry first and it lives in slot p*C + label (and takes the positions of that insertion);
//   * the `keep = min(W, candidates)` best are found with an 8-bit radix select over the keys (histograms are
//     wave-aggregated; once the undecided bucket is small the passes run over a compacted list), ties at the
//     threshold are taken in slot order, and the survivors are ordered by (score desc, slot asc) — exactly the
//     stable sort of the textbook formulation;
//   * new prefixes get nodes (parent, label) / (parent, time) in per-line arenas in HBM; the answer is read back
//     by walking the best beam's chain.
#include "beam_math.hpp"
#include "common.hpp"
#include "kernels.hpp"

namespace ocrs {
namespace k {

namespace {

using beam::kNegInf;
using beam::lse;

constexpr int BEAM_MAX_W = 128, BEAM_LIST = 1024;
constexpr int BEAM_NT = 1024;   // threads per line: the per-step loops are chains of LDS latencies, 16 waves hide them

struct BeamArgs {
    const float* logp;        // packed [R][C]
    const int32_t* Tm;        // [M]
    const int32_t* off;       // [Tmax + 1]
    const uint8_t* excluded;  // [C] or null
    int2* nodes;              // [M][cap]  (parent, label)
    int2* posn;               // [M][cap]  (parent, time)
    uint32_t* out_labels;     // [M][Tmax]
    uint32_t* out_pos;        // [M][Tmax]
    int32_t* out_count;       // [M]
    int M, C, W, Tmax, cap;
};

__device__ __forceinline__ uint64_t sortable(double x) {   // order-preserving map double -> u64; never 0 (= absent)
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// hist[digit] += 1 for every active lane.  High bytes of the keys are concentrated on a few values (the scores
// share their exponent), low bytes are spread: two rounds of "leader adds the population count of its digit"
// take care of the former without contention, plain atomics of the latter without a 64-round loop.
__device__ __forceinline__ void hist_add(uint32_t* hist, unsigned digit, bool active) {
    uint64_t todo = __ballot(active);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (!todo) break;
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned d = __shfl(digit, leader);
        const uint64_t same = __ballot(active && digit == d) & todo;
        if (lane == leader) atomicAdd(&hist[d], (uint32_t)__popcll(same));
        todo &= ~same;
    }
    if ((todo >> lane) & 1) atomicAdd(&hist[digit], 1u);
}

// append `value` to list[] for every lane with `take`, one atomic on the counter per wave
__device__ __forceinline__ void list_append(uint16_t* list, int* counter, bool take, uint16_t value) {
    const uint64_t mask = __ballot(take);
    if (!mask) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(counter, (int)__popcll(mask));
    base = __shfl(base, __ffsll((long long)mask) - 1);
    if (take) list[base + __popcll(mask & ((1ull << lane) - 1))] = value;
}

__global__ void __launch_bounds__(BEAM_NT)
ctc_beam_kernel(BeamArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = a.C, W = a.W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = blockIdx.x;
    const int T = a.Tm[m];
    // ---- LDS carve-up
    uint64_t* key = reinterpret_cast<uint64_t*>(smem);                 // [W*C] candidate keys (0 = absent)
    double* row = reinterpret_cast<double*>(key + (size_t)W * C);      // [C]
    double* pb = row + C;                                              // beams: [W] each
    double* pnb = pb + W;
    double* tot = pnb + W;
    double* spb = tot + W;                                             // the beam's own-prefix entry of this step
    double* spnb = spb + W;
    double* npb = spnb + W;                                            // next beams
    double* npnb = npb + W;
    uint64_t* sel_key = reinterpret_cast<uint64_t*>(npnb + W);         // [W] survivors (unordered)
    int* node = reinterpret_cast<int*>(sel_key + W);                   // [W] each
    int* par = node + W;
    int* pos = par + W;
    int* lab = pos + W;
    int* pidx = lab + W;
    int* nnode = pidx + W;
    int* npar = nnode + W;
    int* npos = npar + W;
    int* nlab = npos + W;
    int* sel_idx = nlab + W;
    int* sorted = sel_idx + W;
    uint32_t* hist = reinterpret_cast<uint32_t*>(sorted + W);          // [256]
    uint32_t* cnt256 = hist + 256;                                     // [BEAM_NT] slot (w*64 + 63): ties in wave w (ordered gather)
    int* ctl = reinterpret_cast<int*>(cnt256 + BEAM_NT);                 // [16] control words
    uint16_t* list = reinterpret_cast<uint16_t*>(ctl + 16);            // [BEAM_LIST] compacted slots
    uint8_t* child = reinterpret_cast<uint8_t*>(list + BEAM_LIST);     // [W*C] which beam IS this extension (255 none)
    enum { NB = 0, NODE_CNT, POS_CNT, N_PRESENT, DIGIT, KLEFT, BUCKET, N_LIST, N_SEL, N_GT_TOTAL };

    int2* nodes = a.nodes + (size_t)m * a.cap;
    int2* posn = a.posn + (size_t)m * a.cap;
    if (tid == 0) {
        node[0] = 0; par[0] = -1; pos[0] = 0; lab[0] = -1; pb[0] = 0.0; pnb[0] = kNegInf;
        ctl[NB] = 1; ctl[NODE_CNT] = 1; ctl[POS_CNT] = 1;
        nodes[0] = make_int2(-1, 0);
        posn[0] = make_int2(-1, 0);
    }
    __syncthreads();

    for (int t = 0; t < T; t++) {
        const int nb = ctl[NB];
        const int N = nb * C;
        // ---- 1. this step's log-probabilities (masked as recognition.rs:547-561), beam totals, parents
        if (tid < C) {
            float v = a.logp[((size_t)a.off[t] + m) * C + tid];
            if (a.excluded && a.excluded[tid]) v = -__builtin_huge_valf();
            row[tid] = (double)v;
        }
        if (tid < nb) {
            tot[tid] = lse(pb[tid], pnb[tid]);
            int p = -1;
            const int pr = par[tid];
            if (pr >= 0)
                for (int j = 0; j < nb; j++)
                    if (node[j] == pr) { p = j; break; }
            pidx[tid] = p;
        }
        for (int i = tid; i < N; i += BEAM_NT) { key[i] = 0; child[i] = 255; }
        if (tid == 0) ctl[N_PRESENT] = 0;
        __syncthreads();
        if (tid < nb && pidx[tid] >= 0) child[pidx[tid] * C + lab[tid]] = (uint8_t)tid;
        __syncthreads();
        // ---- 2. candidates
        int present = 0;
        if (tid < nb) {
            const int i = tid, last = lab[i];
            const double b = tot[i] + row[0];
            double nbv = kNegInf;
            int slot = i * C;
            if (last >= 1 && row[last] != kNegInf) {
                nbv = lse(nbv, pnb[i] + row[last]);
                const int p = pidx[i];
                if (p >= 0) {
                    nbv = lse(nbv, (last == lab[p] ? pb[p] : tot[p]) + row[last]);
                    if (p < i) slot = p * C + last;
                }
            }
            spb[i] = b; spnb[i] = nbv;
            key[slot] = sortable(lse(b, nbv));
            present++;
        }
        for (int idx = tid; idx < N; idx += BEAM_NT) {
            const int i = idx / C, c = idx - i * C;
            if (c == 0) continue;
            const double lp = row[c];
            if (lp == kNegInf || child[idx] != 255) continue;
            key[idx] = sortable((c == lab[i] ? pb[i] : tot[i]) + lp);
            present++;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) present += __shfl_down(present, d);
        if (lane == 0) atomicAdd(&ctl[N_PRESENT], present);
        __syncthreads();
        const int keep = min(W, ctl[N_PRESENT]);
        // ---- 3. radix select: thr = the keep-th largest key
        uint64_t prefix = 0;
        int kleft = keep;
        bool use_list = false;
        for (int pass = 0; pass < 8; pass++) {
            const int shift = 56 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const int n_it = use_list ? ctl[N_LIST] : N;
            for (int base = 0; base < n_it; base += BEAM_NT) {
                const int q = base + tid;
                bool act = q < n_it;
                unsigned digit = 0;
                if (act) {
                    const uint64_t kv = key[use_list ? (int)list[q] : q];
                    act = pass == 0 || (kv >> (shift + 8)) == (prefix >> (shift + 8));
                    digit = (unsigned)(kv >> shift) & 255u;
                }
                hist_add(hist, digit, act);
            }
            __syncthreads();
            if (wave == 0) {   // from the top bin down: the bin where the cumulative count reaches kleft
                const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                uint32_t suf = h0 + h1 + h2 + h3;   // inclusive suffix sum over lanes (lane 63 = top bins)
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_down(suf, d);
                    if (lane + d < 64) suf += o;
                }
                const uint32_t above = suf - (h0 + h1 + h2 + h3);   // elements in higher lanes' bins
                if (suf >= (uint32_t)kleft && above < (uint32_t)kleft) {
                    uint32_t cum = above;
                    int dsel = 0;
                    uint32_t hh[4] = {h0, h1, h2, h3};
                    for (int j = 3; j >= 0; j--) {
                        if (cum + hh[j] >= (uint32_t)kleft) { dsel = 4 * lane + j; break; }
                        cum += hh[j];
                    }
                    ctl[DIGIT] = dsel;
                    ctl[KLEFT] = kleft - (int)cum;
                    ctl[BUCKET] = (int)hist[dsel];
                }
            }
            __syncthreads();
            prefix |= (uint64_t)(unsigned)ctl[DIGIT] << shift;
            kleft = ctl[KLEFT];
            const int bucket = ctl[BUCKET];
            if (!use_list && pass < 7 && bucket <= BEAM_LIST) {   // the undecided bucket is small: compact it
                if (tid == 0) ctl[N_LIST] = 0;
                __syncthreads();
                for (int base = 0; base < N; base += BEAM_NT) {
                    const int q = base + tid;
                    list_append(list, &ctl[N_LIST], q < N && (key[q] >> shift) == (prefix >> shift), (uint16_t)q);
                }
                use_list = true;
            }
            __syncthreads();
        }
        const uint64_t thr = prefix;   // kleft of the keys equal to thr survive, lowest slots first
        // ---- 4. gather the survivors: key > thr, and the first kleft (in slot order) of key == thr
        {
            const int chunk = (N + BEAM_NT - 1) / BEAM_NT;
            const int lo = tid * chunk, hi = min(N, lo + chunk);
            int eq = 0;
            for (int q = lo; q < hi; q++) eq += key[q] == thr;
            {   // cnt256[w*64 + 63] = number of ties in wave w
                int tot_w = eq;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) tot_w += __shfl_down(tot_w, d);
                tot_w = __shfl(tot_w, 0);
                if (lane == 63) cnt256[tid] = (uint32_t)tot_w;
            }
            if (tid == 0) ctl[N_SEL] = 0;
            __syncthreads();
            // exclusive prefix of the per-thread tie counts (threads own consecutive slot ranges)
            int before = 0;
            {
                int incl = eq;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(incl, d);
                    if (lane >= d) incl += o;
                }
                before = incl - eq;
                for (int wv = 0; wv < wave; wv++) before += (int)cnt256[wv * 64 + 63];   // totals of the lower waves
            }
            for (int q = lo; q < hi; q++) {
                const uint64_t kv = key[q];
                bool take = kv > thr;
                if (kv == thr) { take = before < kleft; before++; }
                if (take) {
                    const int s = atomicAdd(&ctl[N_SEL], 1);
                    sel_idx[s] = q;
                    sel_key[s] = kv;
                }
            }
            __syncthreads();
        }
        // ---- 5. order them: (score desc, slot asc)
        if (tid < keep) {
            const uint64_t kv = sel_key[tid];
            const int q = sel_idx[tid];
            int rank = 0;
            for (int j = 0; j < keep; j++) {
                const uint64_t kj = sel_key[j];
                rank += (kj > kv) || (kj == kv && sel_idx[j] < q);
            }
            sorted[rank] = tid;
        }
        __syncthreads();
        // ---- 6. the next beams
        if (tid < keep) {
            const int s = sorted[tid];
            const int idx = sel_idx[s];
            const int i = idx / C, c = idx - i * C;
            const int owner = c == 0 ? i : (int)child[idx];
            if (c == 0) {                 // beam i's own prefix, inserted by itself
                nnode[tid] = node[i]; npar[tid] = par[i]; npos[tid] = pos[i]; nlab[tid] = lab[i];
                npb[tid] = spb[i]; npnb[tid] = spnb[i];
            } else if (owner != 255) {    // beam `owner`'s prefix, first inserted as the extension (i, c) of its parent i
                const int pn = atomicAdd(&ctl[POS_CNT], 1);
                posn[pn] = make_int2(pos[i], t);
                nnode[tid] = node[owner]; npar[tid] = par[owner]; npos[tid] = pn; nlab[tid] = lab[owner];
                npb[tid] = spb[owner]; npnb[tid] = spnb[owner];
            } else {                      // a new prefix
                const int nn = atomicAdd(&ctl[NODE_CNT], 1);
                const int pn = atomicAdd(&ctl[POS_CNT], 1);
                nodes[nn] = make_int2(node[i], c);
                posn[pn] = make_int2(pos[i], t);
                nnode[tid] = nn; npar[tid] = node[i]; npos[tid] = pn; nlab[tid] = c;
                npb[tid] = kNegInf;
                npnb[tid] = (c == lab[i] ? pb[i] : tot[i]) + row[c];
            }
        }
        __syncthreads();
        if (tid < keep) {
            node[tid] = nnode[tid]; par[tid] = npar[tid]; pos[tid] = npos[tid]; lab[tid] = nlab[tid];
            pb[tid] = npb[tid]; pnb[tid] = npnb[tid];
        }
        if (tid == 0) ctl[NB] = keep;
        __syncthreads();
    }
    // ---- the first beam with the maximal total score; its labels and positions, oldest first
    if (tid == 0) {
        const int nb = ctl[NB];
        int best = 0;
        double bs = lse(pb[0], pnb[0]);
        for (int i = 1; i < nb; i++) {
            const double sc = lse(pb[i], pnb[i]);
            if (sc > bs) { best = i; bs = sc; }
        }
        int n = 0;
        for (int ln = node[best]; ln > 0; ln = nodes[ln].x) n++;
        a.out_count[m] = n;
        int ln = node[best], pn = pos[best];
        for (int q = n - 1; q >= 0; q--) {
            a.out_labels[(size_t)m * a.Tmax + q] = (uint32_t)nodes[ln].y;
            a.out_pos[(size_t)m * a.Tmax + q] = (uint32_t)posn[pn].y;
            ln = nodes[ln].x;
            pn = posn[pn].x;
        }
    }
}

size_t beam_lds_bytes(int W, int C) {
    size_t b = (size_t)W * C * 8;          // key
    b += (size_t)C * 8;                    // row
    b += (size_t)W * 8 * 7;                // pb pnb tot spb spnb npb npnb
    b += (size_t)W * 8;                    // sel_key
    b += (size_t)W * 4 * 11;               // node par pos lab pidx nnode npar npos nlab sel_idx sorted
    b += 256 * 4 + BEAM_NT * 4 + 16 * 4;   // hist cnt256 ctl
    b += BEAM_LIST * 2;                    // list
    b += (size_t)W * C;                    // child
    return (b + 15) & ~size_t(15);
}

}  // namespace

bool ctc_beam_supported(int C, int width) {
    return C >= 2 && C <= 128 && width >= 1 && width <= BEAM_MAX_W && (size_t)width * C <= 65535 &&
           beam_lds_bytes(width, C) <= 160 * 1024;
}

size_t ctc_beam_arena_entries(int Tmax, int width) { return (size_t)Tmax * width + 1; }

bool ctc_beam_packed(const float* logp, const int32_t* d_Tm, const int32_t* d_off, int M, int Tmax, int C, int width,
                     const uint8_t* d_excluded, int2* d_nodes, int2* d_posn, uint32_t* out_labels, uint32_t* out_pos,
                     int32_t* out_count, hipStream_t s) {
    if (M <= 0) return true;
    if (!ctc_beam_supported(C, width)) return false;
    BeamArgs a{};
    a.logp = logp; a.Tm = d_Tm; a.off = d_off; a.excluded = d_excluded; a.nodes = d_nodes; a.posn = d_posn;
    a.out_labels = out_labels; a.out_pos = out_pos; a.out_count = out_count;
    a.M = M; a.C = C; a.W = width; a.Tmax = Tmax; a.cap = (int)ctc_beam_arena_entries(Tmax, width);
    const size_t lds = beam_lds_bytes(width, C);
    static std::atomic<uint64_t> lds_ok{0};
    if (lds > 64 * 1024) allow_dynamic_lds(reinterpret_cast<const void*>(&ctc_beam_kernel), lds_ok);
    hipLaunchKernelGGL(ctc_beam_kernel, dim3(M), dim3(BEAM_NT), lds, s, a);
    return true;
}

}  // namespace k
}  // namespace ocrs
