// Host half of the JPEG hand-off (jpeg.hpp): marker parsing and Huffman entropy decoding of baseline / extended
// sequential (SOF0, SOF1) and progressive (SOF2) streams into quantised DCT coefficients, ITU-T T.81 Annex F / G.
// Written against the standard; the procedures mirror libjpeg's jdhuff.c / jdphuff.c only in that both implement the
// same normative decoding procedures (DECODE, RECEIVE, EXTEND; the progressive EOB-run / correction-bit rules of G.1.2).
// No HIP here: tests/sanitize builds this file under ASan / UBSan.
#include <cstring>

#include "common.hpp"
#include "jpeg.hpp"

namespace ocrs {
namespace jpeg {
namespace {

const uint8_t kNatural[64 + 16] = {   // zig-zag index -> natural (row * 8 + col) position; 16 extra entries catch corrupt runs
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

}  // namespace

const uint8_t Coefficients::kZigzagOfNatural[64] = {
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

namespace {

[[noreturn]] void bad(const char* what) { fail(OCRS_ERR_IMAGE_SOURCE, "JPEG: %s", what); }

struct Huff {
    bool present = false;
    uint8_t bits[17] = {};
    uint8_t vals[256] = {};
    int32_t mincode[17] = {}, maxcode[18] = {}, valptr[17] = {};
    static constexpr int kLook = 9;
    uint8_t look_n[1 << kLook] = {}, look_v[1 << kLook] = {};   // codes of up to 9 bits by their left-aligned 9-bit prefix
    // AC fast path: when code + magnitude bits fit the 9-bit window and the value fits 8 bits, the whole coefficient comes
    // from one table look-up: (value << 8) | (run << 4) | total bits; 0 = take the general path
    int16_t fast_ac[1 << kLook] = {};
    void build() {
        int code = 0, k = 0;
        memset(look_n, 0, sizeof look_n);
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k;
            mincode[l] = code;
            if (bits[l]) {
                if (l <= kLook) {
                    for (int i = 0; i < bits[l]; i++) {
                        const int first = (code + i) << (kLook - l);
                        for (int f = 0; f < (1 << (kLook - l)); f++) {
                            if (first + f >= (1 << kLook)) bad("bad Huffman table");
                            look_n[first + f] = (uint8_t)l;
                            look_v[first + f] = vals[k + i];
                        }
                    }
                }
                code += bits[l];
                k += bits[l];
                maxcode[l] = code - 1;
            } else {
                maxcode[l] = -1;
            }
            if (code > (1 << l)) bad("bad Huffman table");
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        for (int i = 0; i < (1 << kLook); i++) {
            fast_ac[i] = 0;
            const int len = look_n[i];
            if (!len) continue;
            const int rs = look_v[i], run = rs >> 4, mag = rs & 15;
            if (mag == 0 || len + mag > kLook) continue;
            int v = ((i << len) & ((1 << kLook) - 1)) >> (kLook - mag);
            if (v < (1 << (mag - 1))) v += (int)((~0u) << mag) + 1;   // EXTEND
            if (v >= -128 && v <= 127) fast_ac[i] = (int16_t)(v * 256 + run * 16 + (len + mag));
        }
        present = true;
    }
};

// Entropy-coded segment reader: byte stuffing (FF 00), markers end the data (zero bits are supplied past them, as
// every decoder does for truncated scans).
struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int n = 0;
    int marker = 0;   // the marker that stopped the segment (0: none yet)
    Bits(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}
    void fill() {
        while (n <= 56) {
            uint32_t c = 0;
            if (!marker && p < end) {
                c = *p;
                if (c == 0xFF) {
                    const uint8_t* q = p + 1;
                    while (q < end && *q == 0xFF) q++;   // fill bytes
                    if (q >= end) { marker = 0xD9; c = 0; }
                    else if (*q == 0) { p = q + 1; }
                    else { marker = *q; p = q + 1; c = 0; }
                } else {
                    p++;
                }
            } else if (!marker) {
                marker = 0xD9;   // ran off the end: as if EOI
            }
            acc = (acc << 8) | c;
            n += 8;
        }
    }
    uint32_t peek(int k) { if (n < k) fill(); return (uint32_t)(acc >> (n - k)) & ((1u << k) - 1); }
    void skip(int k) { n -= k; }
    uint32_t get(int k) { if (k == 0) return 0; const uint32_t v = peek(k); skip(k); return v; }
    int decode(const Huff& h) {
        const uint32_t p9 = peek(Huff::kLook);
        if (h.look_n[p9]) { skip(h.look_n[p9]); return h.look_v[p9]; }
        int l = Huff::kLook + 1;
        int32_t code = (int32_t)peek(l);
        while (l <= 16 && code > h.maxcode[l]) { l++; if (l <= 16) code = (int32_t)peek(l); }
        if (l > 16) { skip(16); return 0; }   // corrupt data: the garbage-in rule of every decoder, no error
        skip(l);
        return h.vals[(h.valptr[l] + code - h.mincode[l]) & 255];
    }
    // byte-align and consume the restart marker that must follow
    void restart() {
        n = 0; acc = 0;
        if (!marker) {   // the marker has not been read yet: skip to it
            while (p + 1 < end && !(p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF)) p++;
            if (p + 1 < end) { marker = p[1]; p += 2; }
        }
        if (marker >= 0xD0 && marker <= 0xD7) marker = 0;   // resume after RSTn; any other marker ends the scan (zeros follow)
    }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Decoder {
    Coefficients c;
    Huff dc[4], ac[4];
    int restart_interval = 0;
    std::vector<int16_t> coef;   // dense [nblocks][64], natural order
    bool saw_sof = false, adobe = false, jfif = false;
    int n_scans = 0;                 // scans decoded so far (bounded: every scan re-walks every block it names)
    bool seq_done[3] = {false, false, false};   // sequential streams: components a scan has already decoded
    int adobe_transform = -1;

    size_t block_index(const Component& k, int by, int bx) const { return k.first_block + (size_t)by * k.blocks_w + bx; }
    int16_t* block(const Component& k, int by, int bx) { return coef.data() + block_index(k, by, bx) * 64; }

    void parse_sof(const uint8_t* s, size_t len, int marker) {
        if (saw_sof) bad("more than one frame");
        if (len < 6) bad("short SOF");
        if (s[0] != 8) bad("only 8-bit samples are supported");
        c.height = (s[1] << 8) | s[2];
        c.width = (s[3] << 8) | s[4];
        c.ncomp = s[5];
        c.progressive = marker == 0xC2;
        if (c.width <= 0 || c.height <= 0) bad("empty image");
        if (c.ncomp != 1 && c.ncomp != 3) bad("only 1- and 3-component images are supported");
        // Budget BEFORE any allocation: a 200-byte file may claim 65535 x 65535.  512 MiB of decoded samples is the default
        // allocation limit of the `image` crate the reference decodes with (ocrs-cli/src/main.rs:312-323).
        if ((uint64_t)c.width * (uint64_t)c.height * (uint64_t)c.ncomp > (uint64_t(512) << 20))
            bad("image too large (more than 512 MiB of decoded samples)");
        if (len < 6 + 3 * (size_t)c.ncomp) bad("short SOF");
        for (int i = 0; i < c.ncomp; i++) {
            Component& k = c.comp[i];
            k.id = s[6 + 3 * i];
            k.h = s[7 + 3 * i] >> 4;
            k.v = s[7 + 3 * i] & 15;
            k.tq = s[8 + 3 * i];
            if (k.h < 1 || k.h > 2 || k.v < 1 || k.v > 2 || k.tq > 3) bad("unsupported sampling factors");
            c.hmax = std::max(c.hmax, k.h);
            c.vmax = std::max(c.vmax, k.v);
        }
        if (c.ncomp == 1) { c.comp[0].h = c.comp[0].v = 1; c.hmax = c.vmax = 1; }   // a single component is never subsampled
        if (c.ncomp == 3) {
            if (c.comp[1].h != 1 || c.comp[1].v != 1 || c.comp[2].h != 1 || c.comp[2].v != 1) bad("unsupported chroma sampling");
            if (c.comp[0].h == 1 && c.comp[0].v == 2) bad("4:4:0 sampling is not supported");
        }
        const int mcux = (c.width + 8 * c.hmax - 1) / (8 * c.hmax), mcuy = (c.height + 8 * c.vmax - 1) / (8 * c.vmax);
        size_t first = 0;
        for (int i = 0; i < c.ncomp; i++) {
            Component& k = c.comp[i];
            k.width = (c.width * k.h + c.hmax - 1) / c.hmax;
            k.height = (c.height * k.v + c.vmax - 1) / c.vmax;
            k.blocks_w = mcux * k.h;
            k.blocks_h = mcuy * k.v;
            k.first_block = first;
            first += (size_t)k.blocks_w * k.blocks_h;
        }
        if (first > ((size_t)1 << 26)) bad("image too large");
        // progressive scans refine coefficients in place: dense accumulation, made sparse at the end.  Sequential streams
        // decode every block exactly once: the sparse form is written directly (no 2 bytes-per-sample buffer to clear and scan)
        if (c.progressive) coef.assign(first * 64, 0);
        c.mask.assign(first, 0);
        c.offset.assign(first + 1, 0);
        c.values.clear();
        c.values.reserve(std::min<size_t>(first * 12, (size_t)1 << 24));   // (grows with what the stream really holds)
        saw_sof = true;
    }

    void parse_dqt(const uint8_t* s, size_t len) {
        while (len > 0) {
            const int pq = s[0] >> 4, tq = s[0] & 15;
            if (tq > 3 || pq > 1) bad("bad DQT");
            const size_t need = 1 + 64 * (size_t)(pq + 1);
            if (len < need) bad("short DQT");
            for (int i = 0; i < 64; i++) {
                const int v = pq ? (s[1 + 2 * i] << 8) | s[2 + 2 * i] : s[1 + i];
                c.quant[tq][kNatural[i]] = (uint16_t)v;
            }
            s += need;
            len -= need;
        }
    }

    void parse_dht(const uint8_t* s, size_t len) {
        while (len > 0) {
            if (len < 17) bad("short DHT");
            const int tc = s[0] >> 4, th = s[0] & 15;
            if (tc > 1 || th > 3) bad("bad DHT");
            Huff& h = tc ? ac[th] : dc[th];
            int total = 0;
            h.bits[0] = 0;
            for (int i = 1; i <= 16; i++) { h.bits[i] = s[i]; total += s[i]; }
            if (total > 256 || len < 17 + (size_t)total) bad("bad DHT");
            memset(h.vals, 0, sizeof h.vals);
            memcpy(h.vals, s + 17, total);
            h.build();
            s += 17 + total;
            len -= 17 + total;
        }
    }

    // One scan.  Returns the position after its entropy-coded data (at the marker that ended it).
    const uint8_t* scan(const uint8_t* s, size_t len, const uint8_t* data, const uint8_t* end) {
        if (!saw_sof) bad("SOS before SOF");
        if (len < 1) bad("short SOS");
        if (++n_scans > 500) bad("too many scans");   // (libjpeg-turbo's scan limit for untrusted input; real progressive files have ~10)
        const int ns = s[0];
        if (ns < 1 || ns > c.ncomp || len < 1 + 2 * (size_t)ns + 3) bad("bad SOS");
        int ci[3], td[3], ta[3];
        for (int i = 0; i < ns; i++) {
            int found = -1;
            for (int k = 0; k < c.ncomp; k++)
                if (c.comp[k].id == s[1 + 2 * i]) found = k;
            if (found < 0) bad("SOS names an unknown component");
            for (int q = 0; q < i; q++)
                if (ci[q] == found) bad("SOS names a component twice");
            ci[i] = found;
            td[i] = s[2 + 2 * i] >> 4;
            ta[i] = s[2 + 2 * i] & 15;
            if (td[i] > 3 || ta[i] > 3) bad("bad table selector");
        }
        int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
        if (c.progressive) {
            if (Ss > Se || Se > 63 || Al > 13 || Ah > 13) bad("bad progressive scan parameters");
            if (Ss == 0 && Se != 0) bad("bad progressive scan parameters");   // DC scans carry no AC
            if (Ss > 0 && ns != 1) bad("an AC scan must have one component");
            if (Ah != 0 && Ah != Al + 1) bad("bad successive approximation");
        } else {
            Ss = 0; Se = 63; Ah = 0; Al = 0;
            for (int i = 0; i < ns; i++) {   // a sequential scan decodes its blocks once and for all: a second one would append to `values`
                if (seq_done[ci[i]]) bad("a component is coded by two scans");
                seq_done[ci[i]] = true;
            }
        }
        for (int i = 0; i < ns; i++) {
            if ((Ss == 0 && Ah == 0) && !dc[td[i]].present) bad("missing DC Huffman table");
            if ((!c.progressive || Ss > 0) && !ac[ta[i]].present) bad("missing AC Huffman table");
        }
        Bits br(data, end);
        int pred[3] = {0, 0, 0};
        uint32_t eobrun = 0;
        // geometry: interleaved scans walk MCUs (h x v blocks per component), single-component scans walk that
        // component's own blocks (only those that cover the image)
        const bool inter = ns > 1;
        const Component& k0 = c.comp[ci[0]];
        const int mcux = inter ? (c.width + 8 * c.hmax - 1) / (8 * c.hmax) : (k0.width + 7) / 8;
        const int mcuy = inter ? (c.height + 8 * c.vmax - 1) / (8 * c.vmax) : (k0.height + 7) / 8;
        int until_restart = restart_interval;
        for (int my = 0; my < mcuy; my++) {
            for (int mx = 0; mx < mcux; mx++) {
                if (restart_interval && until_restart == 0) {
                    br.restart();
                    pred[0] = pred[1] = pred[2] = 0;
                    eobrun = 0;
                    until_restart = restart_interval;
                }
                for (int i = 0; i < ns; i++) {
                    const Component& k = c.comp[ci[i]];
                    const int bh = inter ? k.h : 1, bv = inter ? k.v : 1;
                    for (int v = 0; v < bv; v++)
                        for (int h = 0; h < bh; h++) {
                            if (!c.progressive) {
                                sequential_block(br, block_index(k, my * bv + v, mx * bh + h), dc[td[i]], ac[ta[i]], pred[i]);
                                continue;
                            }
                            int16_t* b = block(k, my * bv + v, mx * bh + h);
                            if (Ss == 0) { if (Ah == 0) dc_first(br, b, dc[td[i]], pred[i], Al); else dc_refine(br, b, Al); }
                            else if (Ah == 0) ac_first(br, b, ac[ta[i]], Ss, Se, Al, eobrun);
                            else ac_refine(br, b, ac[ta[i]], Ss, Se, Al, eobrun);
                        }
                }
                until_restart--;
            }
        }
        // position of the marker that ends the scan
        if (br.marker) return br.p - 2 >= data ? br.p - 2 : data;
        const uint8_t* p = br.p;
        while (p + 1 < end && !(p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF && !(p[1] >= 0xD0 && p[1] <= 0xD7))) p++;
        return p;
    }

    // One block of a sequential scan, straight into the sparse form: mask bit z = zig-zag index z is non-zero, values in
    // ascending zig-zag index = the order they are decoded in.  (A block named by two scans of a corrupt stream keeps the last.)
    void sequential_block(Bits& br, size_t blk, const Huff& hd, const Huff& ha, int& pred) {
        int s = br.decode(hd);
        if (s) { if (s > 15) s = 15; const int r = (int)br.get(s); s = extend(r, s); }
        pred = (int16_t)(pred + s);   // (wraps like the 16-bit coefficient it becomes: no signed overflow on corrupt streams)
        uint64_t m = 0;
        c.offset[blk] = (uint32_t)c.values.size();
        if ((int16_t)pred != 0) { m |= 1; c.values.push_back((int16_t)pred); }
        for (int k = 1; k < 64; k++) {
            const int fa = ha.fast_ac[br.peek(Huff::kLook)];
            if (fa) {
                k += (fa >> 4) & 15;
                br.skip(fa & 15);
                if (k > 63) break;   // corrupt run
                m |= uint64_t(1) << k;
                c.values.push_back((int16_t)(fa >> 8));
                continue;
            }
            const int rs = br.decode(ha);
            const int r = rs >> 4, sz = rs & 15;
            if (sz) {
                k += r;
                const int v = extend((int)br.get(sz), sz);
                if (k > 63) break;
                if ((int16_t)v != 0) { m |= uint64_t(1) << k; c.values.push_back((int16_t)v); }
            } else {
                if (r != 15) break;
                k += 15;
            }
        }
        c.mask[blk] = m;
    }
    static void dc_first(Bits& br, int16_t* b, const Huff& hd, int& pred, int Al) {
        int s = br.decode(hd);
        if (s) { if (s > 15) s = 15; const int r = (int)br.get(s); s = extend(r, s); }
        pred = (int16_t)(pred + s);
        b[0] = (int16_t)((unsigned)pred << Al);
    }
    static void dc_refine(Bits& br, int16_t* b, int Al) {
        if (br.get(1)) b[0] = (int16_t)(b[0] | (1 << Al));
    }
    static void ac_first(Bits& br, int16_t* b, const Huff& ha, int Ss, int Se, int Al, uint32_t& eobrun) {
        if (eobrun > 0) { eobrun--; return; }
        for (int k = Ss; k <= Se; k++) {
            const int rs = br.decode(ha);
            const int r = rs >> 4, sz = rs & 15;
            if (sz) {
                k += r;
                const int v = extend((int)br.get(sz), sz);
                b[kNatural[k]] = (int16_t)(v * (1 << Al));
            } else if (r == 15) {
                k += 15;
            } else {
                eobrun = 1u << r;
                if (r) eobrun += br.get(r);
                eobrun--;
                break;
            }
        }
    }
    static void ac_refine(Bits& br, int16_t* b, const Huff& ha, int Ss, int Se, int Al, uint32_t& eobrun) {
        const int p1 = 1 << Al, m1 = -(1 << Al);
        int k = Ss;
        auto correct = [&](int16_t* t) {   // a correction bit for a coefficient with history (G.1.2.3)
            if (br.get(1) && (*t & p1) == 0) *t = (int16_t)(*t >= 0 ? *t + p1 : *t + m1);
        };
        if (eobrun == 0) {
            for (; k <= Se; k++) {
                const int rs = br.decode(ha);
                int r = rs >> 4, s = rs & 15;
                if (s) {
                    s = br.get(1) ? p1 : m1;   // a newly non-zero coefficient (its size must be 1)
                } else if (r != 15) {
                    eobrun = 1u << r;
                    if (r) eobrun += br.get(r);
                    break;                    // the rest of the band belongs to the EOB run
                }
                // skip r zero-history coefficients, correcting the non-zero ones passed on the way
                do {
                    int16_t* t = b + kNatural[k];
                    if (*t != 0) correct(t);
                    else if (--r < 0) break;
                    k++;
                } while (k <= Se);
                if (s && k <= Se) b[kNatural[k]] = (int16_t)s;
            }
        }
        if (eobrun > 0) {
            for (; k <= Se; k++) {
                int16_t* t = b + kNatural[k];
                if (*t != 0) correct(t);
            }
            eobrun--;
        }
    }

    void sparsify() {   // progressive: dense -> sparse, values in ascending zig-zag index
        const size_t nb = c.mask.size();
        if (!c.progressive) { c.offset[nb] = (uint32_t)c.values.size(); return; }
        c.values.clear();
        for (size_t i = 0; i < nb; i++) {
            const int16_t* b = coef.data() + i * 64;
            uint64_t m = 0;
            c.offset[i] = (uint32_t)c.values.size();
            for (int z = 0; z < 64; z++) {
                const int16_t v = b[kNatural[z]];
                if (v) { m |= uint64_t(1) << z; c.values.push_back(v); }
            }
            c.mask[i] = m;
        }
        c.offset[nb] = (uint32_t)c.values.size();
    }
};

}  // namespace

Coefficients decode_coefficients(const uint8_t* data, size_t len) {
    if (!data || len < 4 || data[0] != 0xFF || data[1] != 0xD8) bad("not a JPEG stream (no SOI)");
    Decoder d;
    const uint8_t* p = data + 2;
    const uint8_t* end = data + len;
    bool scanned = false;
    for (;;) {
        while (p < end && *p != 0xFF) p++;        // (garbage between segments is skipped)
        while (p < end && *p == 0xFF) p++;
        if (p >= end) break;
        const int m = *p++;
        if (m == 0xD9) break;                       // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0) continue;
        if (p + 2 > end) bad("truncated marker segment");
        const size_t seg = ((size_t)p[0] << 8) | p[1];
        if (seg < 2 || p + seg > end) bad("truncated marker segment");
        const uint8_t* s = p + 2;
        const size_t n = seg - 2;
        p += seg;
        switch (m) {
            case 0xC0: case 0xC1: case 0xC2: d.parse_sof(s, n, m); break;
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                bad("lossless / hierarchical JPEG is not supported");
            case 0xC9: case 0xCA: bad("arithmetic-coded JPEG is not supported");
            case 0xC4: d.parse_dht(s, n); break;
            case 0xCC: bad("arithmetic-coded JPEG is not supported");
            case 0xDB: d.parse_dqt(s, n); break;
            case 0xDD: if (n < 2) bad("short DRI"); d.restart_interval = (s[0] << 8) | s[1]; break;
            case 0xE0: if (n >= 5 && memcmp(s, "JFIF", 5) == 0) d.jfif = true; break;
            case 0xEE: if (n >= 12 && memcmp(s, "Adobe", 5) == 0) { d.adobe = true; d.adobe_transform = s[11]; } break;
            case 0xDA: p = d.scan(s, n, p, end); scanned = true; break;
            default: break;   // APPn, COM, DNL ...: skipped
        }
    }
    if (!d.saw_sof || !scanned) bad("no image data");
    for (int i = 0; i < d.c.ncomp; i++) {
        bool any = false;
        for (int q = 0; q < 64; q++) any |= d.c.quant[d.c.comp[i].tq][q] != 0;
        if (!any) bad("missing quantisation table");
    }
    // colour space as libjpeg decides it (jdapimin.c default_decompress_parms): JFIF -> YCbCr; Adobe -> by its transform
    // flag; neither -> RGB only if the component ids spell it, else YCbCr
    if (d.c.ncomp == 3) {
        if (d.jfif) d.c.ycc = true;
        else if (d.adobe) d.c.ycc = d.adobe_transform != 0;
        else d.c.ycc = !(d.c.comp[0].id == 'R' && d.c.comp[1].id == 'G' && d.c.comp[2].id == 'B');
    }
    d.sparsify();
    return std::move(d.c);
}

std::string describe(const Coefficients& c) {
    char buf[128];
    const char* ss = c.ncomp == 1 ? "grey" : (c.hmax == 1 && c.vmax == 1) ? "4:4:4" : (c.hmax == 2 && c.vmax == 1) ? "4:2:2" : "4:2:0";
    snprintf(buf, sizeof buf, "%dx%d %s %s%s", c.width, c.height, c.progressive ? "progressive" : "sequential", ss,
             c.ncomp == 3 && !c.ycc ? " RGB" : "");
    return buf;
}

}  // namespace jpeg
}  // namespace ocrs
