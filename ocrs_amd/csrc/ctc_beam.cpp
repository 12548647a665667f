// CTC prefix beam search — rten::ctc::CtcDecoder::decode_beam as called at
// ocrs/src/recognition.rs:512-514 (DecodeMethod::BeamSearch, width 100 from the CLI,
// ocrs-cli/src/main.rs:403-404).  Host side.
//
// rten is not vendored in the reference tree, so this is the published algorithm
// (Hannun et al. 2014, Alg. 1) with every tie rule fixed — the same rules as
// oracle/pipeline.py::ctc_beam_search, which it must match exactly (float64 scores,
// first-insertion candidate order, blank first then labels ascending, stable pruning,
// first maximum wins).  Parity with rten itself is unpinned.
#include <algorithm>
#include <cmath>
#include <limits>
#include <unordered_map>
#include <vector>

#include "engine.hpp"

namespace ocrs {

namespace {
const double NEG = -std::numeric_limits<double>::infinity();

inline double lse(double a, double b) {
    if (a == NEG) return b;
    if (b == NEG) return a;
    const double m = a > b ? a : b;
    return m + std::log(std::exp(a - m) + std::exp(b - m));
}

struct LabelNode { int parent; int label; std::unordered_map<int, int> children; };
struct PosNode { int parent; uint32_t pos; };
struct Beam { int labels; int positions; double pb, pnb; };
}  // namespace

std::vector<CtcStep> ctc_beam_search(const float* logp, int T, int C, int row_stride, uint32_t width) {
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

}  // namespace ocrs
