// CTC prefix beam search — rten::ctc::CtcDecoder::decode_beam as called at
// ocrs/src/recognition.rs:512-514 (DecodeMethod::BeamSearch, width 100 from the CLI,
// ocrs-cli/src/main.rs:403-404).  Host implementations; the HIP one is kernels_beam.hip.
//
// rten is not vendored in the reference tree, so this is the published algorithm
// (Hannun et al. 2014, Alg. 1) with every tie rule fixed — the same rules as
// oracle/pipeline.py::ctc_beam_search, which it must match exactly (float64 scores with the fixed
// log-sum-exp of beam_math.hpp, first-insertion candidate order, blank first then labels ascending,
// stable pruning, first maximum wins).  Parity with rten itself is unpinned.
//
// Two implementations of the same function:
//   * ctc_beam_search_reference — the algorithm as written down: a label trie, a per-step candidate
//     map keyed by prefix, candidates appended in insertion order, stable sort.  1.6 s per 600-step
//     line at width 100.
//   * ctc_beam_search — the same candidates without the maps.  A step's candidates are the W current
//     beams ("stay") and the W x (C-1) one-label extensions; an extension can only coincide with
//     another candidate if it IS one of the current beams (prefix of beam j = prefix of beam p +
//     label), so merging reduces to "does beam j's parent prefix belong to a current beam p", and the
//     insertion order is a closed-form key: (beam index, label) for an extension, (own index, 0) for a
//     stay — unless the parent p comes earlier in the beam order, then the parent's extension inserted
//     the entry first, at (p, label), and the entry carries the positions of that insertion.
//     Selection is nth_element + sort on (score desc, key asc).  ~100x faster, identical output
//     (tests/test_host_cpu.py compares the two on random matrices for widths 1..100).
#include <algorithm>
#include <cmath>
#include <limits>
#include <unordered_map>
#include <vector>

#include "beam_math.hpp"
#include "engine.hpp"

namespace ocrs {

namespace {
const double NEG = beam::kNegInf;
using beam::lse;

struct LabelNode { int parent; int label; std::unordered_map<int, int> children; };
struct PosNode { int parent; uint32_t pos; };
struct Beam { int labels; int positions; double pb, pnb; };
}  // namespace

std::vector<CtcStep> ctc_beam_search_reference(const float* logp, int T, int C, int row_stride, uint32_t width) {
    std::vector<LabelNode> trie(1);
    trie[0].parent = -1;
    trie[0].label = 0;
    std::vector<PosNode> pos_nodes(1, PosNode{-1, 0});
    auto child = [&](int node, int c) {
        auto it = trie[node].children.find(c);
        if (it != trie[node].children.end()) return it->second;
        const int id = (int)trie.size();
        trie.push_back(LabelNode{node, c, {}});
        trie[node].children.emplace(c, id);
        return id;
    };
    std::vector<Beam> beams{Beam{0, 0, 0.0, NEG}};
    std::vector<Beam> cand;
    std::unordered_map<int, int> cand_index;  // label node -> index in cand
    std::vector<double> row(C);
    for (int t = 0; t < T; t++) {
        for (int c = 0; c < C; c++) row[c] = (double)logp[(size_t)t * row_stride + c];
        cand.clear();
        cand_index.clear();
        auto add = [&](int labels, int positions_parent, bool append_pos, double pb, double pnb) {
            auto it = cand_index.find(labels);
            if (it == cand_index.end()) {
                int pn = positions_parent;
                if (append_pos) {
                    pn = (int)pos_nodes.size();
                    pos_nodes.push_back(PosNode{positions_parent, (uint32_t)t});
                }
                cand_index.emplace(labels, (int)cand.size());
                cand.push_back(Beam{labels, pn, pb, pnb});
            } else {
                Beam& e = cand[it->second];
                e.pb = lse(e.pb, pb);
                e.pnb = lse(e.pnb, pnb);
            }
        };
        for (const Beam& b : beams) {
            const double total = lse(b.pb, b.pnb);
            add(b.labels, b.positions, false, total + row[0], NEG);
            const int last = b.labels == 0 ? -1 : trie[b.labels].label;
            for (int c = 1; c < C; c++) {
                const double lp = row[c];
                if (lp == NEG) continue;
                if (c == last) {
                    add(b.labels, b.positions, false, NEG, b.pnb + lp);
                    add(child(b.labels, c), b.positions, true, NEG, b.pb + lp);
                } else {
                    add(child(b.labels, c), b.positions, true, NEG, total + lp);
                }
            }
        }
        std::vector<std::pair<double, int>> scored(cand.size());
        for (size_t i = 0; i < cand.size(); i++) scored[i] = {lse(cand[i].pb, cand[i].pnb), (int)i};
        std::stable_sort(scored.begin(), scored.end(),
                         [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; });
        const size_t keep = std::min<size_t>(width, scored.size());
        beams.clear();
        for (size_t i = 0; i < keep; i++) beams.push_back(cand[scored[i].second]);
    }
    size_t best = 0;
    double best_score = lse(beams[0].pb, beams[0].pnb);
    for (size_t i = 1; i < beams.size(); i++) {
        const double sc = lse(beams[i].pb, beams[i].pnb);
        if (sc > best_score) { best = i; best_score = sc; }
    }
    std::vector<CtcStep> out;
    int ln = beams[best].labels, pn = beams[best].positions;
    while (ln > 0) {
        out.push_back(CtcStep{(uint32_t)trie[ln].label, pos_nodes[pn].pos});
        ln = trie[ln].parent;
        pn = pos_nodes[pn].parent;
    }
    std::reverse(out.begin(), out.end());
    return out;
}

std::vector<CtcStep> ctc_beam_search(const float* logp, int T, int C, int row_stride, uint32_t width) {
    struct Node { int parent; int label; };
    struct FBeam { int node, pos; double pb, pnb; };
    struct Cand { double score; int key; };
    if (width == 0) width = 1;
    std::vector<Node> nodes(1, Node{-1, 0});
    std::vector<PosNode> pos_nodes(1, PosNode{-1, 0});
    std::vector<FBeam> beams{FBeam{0, 0, 0.0, NEG}}, next;
    std::vector<int> node2beam(1, -1);      // node id -> index among the current beams
    std::vector<double> row(C), total, stay_pb, stay_pnb;
    std::vector<int> pidx, stay_key;
    std::vector<uint8_t> is_beam_child;      // [beam][label]: that extension is one of the current beams
    std::vector<Cand> cand;
    auto before = [](const Cand& a, const Cand& b) { return a.score > b.score || (a.score == b.score && a.key < b.key); };
    for (int t = 0; t < T; t++) {
        for (int c = 0; c < C; c++) row[c] = (double)logp[(size_t)t * row_stride + c];
        const int nb = (int)beams.size();
        total.resize(nb); stay_pb.resize(nb); stay_pnb.resize(nb); pidx.resize(nb); stay_key.resize(nb);
        is_beam_child.assign((size_t)nb * C, 0);
        for (int i = 0; i < nb; i++) node2beam[beams[i].node] = i;
        for (int i = 0; i < nb; i++) {
            total[i] = lse(beams[i].pb, beams[i].pnb);
            const int par = nodes[beams[i].node].parent;
            pidx[i] = par >= 0 ? node2beam[par] : -1;
            if (pidx[i] >= 0) is_beam_child[(size_t)pidx[i] * C + nodes[beams[i].node].label] = 1;
        }
        cand.clear();
        for (int i = 0; i < nb; i++) {
            const FBeam& b = beams[i];
            const int last = b.node == 0 ? -1 : nodes[b.node].label;
            // the entry of this beam's own prefix: blank, repeat of its last label, and the extension of its parent
            double pb = total[i] + row[0], pnb = NEG;
            int key = i * C;
            if (last >= 1 && row[last] != NEG) {
                pnb = lse(pnb, b.pnb + row[last]);
                const int p = pidx[i];
                if (p >= 0) {
                    const int plast = beams[p].node == 0 ? -1 : nodes[beams[p].node].label;
                    pnb = lse(pnb, (last == plast ? beams[p].pb : total[p]) + row[last]);
                    if (p < i) key = p * C + last;   // the parent's extension inserted this entry first
                }
            }
            stay_pb[i] = pb; stay_pnb[i] = pnb; stay_key[i] = key;
            cand.push_back(Cand{lse(pb, pnb), key});
            // one-label extensions that are not current beams themselves
            const uint8_t* ibc = &is_beam_child[(size_t)i * C];
            for (int c = 1; c < C; c++) {
                const double lp = row[c];
                if (lp == NEG || ibc[c]) continue;
                cand.push_back(Cand{(c == last ? b.pb : total[i]) + lp, i * C + c});
            }
        }
        for (int i = 0; i < nb; i++) node2beam[beams[i].node] = -1;
        const size_t keep = std::min<size_t>(width, cand.size());
        if (keep < cand.size()) std::nth_element(cand.begin(), cand.begin() + keep, cand.end(), before);
        std::sort(cand.begin(), cand.begin() + keep, before);
        // a stay entry's key is either i*C (its own) or p*C + label (the parent's extension): map keys back
        next.clear();
        std::unordered_map<int, int> early;  // key of an early-inserted stay entry -> beam index (at most nb entries)
        for (int i = 0; i < nb; i++)
            if (stay_key[i] != i * C) early.emplace(stay_key[i], i);
        for (size_t q = 0; q < keep; q++) {
            const int key = cand[q].key;
            const int i = key / C, c = key - i * C;
            auto it = c == 0 ? early.end() : early.find(key);
            if (c == 0) {   // beam i's own prefix, inserted by itself
                next.push_back(FBeam{beams[i].node, beams[i].pos, stay_pb[i], stay_pnb[i]});
            } else if (it != early.end()) {   // beam j's prefix, first inserted as the extension (i, c) of its parent i
                const int j = it->second;
                pos_nodes.push_back(PosNode{beams[i].pos, (uint32_t)t});
                next.push_back(FBeam{beams[j].node, (int)pos_nodes.size() - 1, stay_pb[j], stay_pnb[j]});
            } else {        // a new prefix
                const int last = beams[i].node == 0 ? -1 : nodes[beams[i].node].label;
                nodes.push_back(Node{beams[i].node, c});
                node2beam.push_back(-1);
                pos_nodes.push_back(PosNode{beams[i].pos, (uint32_t)t});
                next.push_back(FBeam{(int)nodes.size() - 1, (int)pos_nodes.size() - 1, NEG,
                                     (c == last ? beams[i].pb : total[i]) + row[c]});
            }
        }
        beams.swap(next);
    }
    size_t best = 0;
    double best_score = lse(beams[0].pb, beams[0].pnb);
    for (size_t i = 1; i < beams.size(); i++) {
        const double sc = lse(beams[i].pb, beams[i].pnb);
        if (sc > best_score) { best = i; best_score = sc; }
    }
    std::vector<CtcStep> out;
    int ln = beams[best].node, pn = beams[best].pos;
    while (ln > 0) {
        out.push_back(CtcStep{(uint32_t)nodes[ln].label, pos_nodes[pn].pos});
        ln = nodes[ln].parent;
        pn = pos_nodes[pn].parent;
    }
    std::reverse(out.begin(), out.end());
    return out;
}

}  // namespace ocrs
