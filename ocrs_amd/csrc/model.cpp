#include "model.hpp"

#include <cstdio>
#include <functional>
#include <map>
#include <mutex>

#include "kernels.hpp"

namespace ocrs {

namespace {
#pragma pack(push, 1)
struct FileHeader {
    char magic[8];
    uint32_t version, kind;
    int64_t input_shape[4];
    uint32_t n_ops, n_slots, out_slot, reserved;
    uint64_t blob_floats;
};
struct FileOp {
    uint32_t type;
    int32_t in0, in1, out, relu, kh, kw, cin, cout, hidden;
    uint32_t n_w, reserved;
    struct { uint64_t off, cnt; } w[8];
};
#pragma pack(pop)
static_assert(sizeof(FileHeader) == 72, "header layout");
static_assert(sizeof(FileOp) == 176, "op layout");

const char* const kOpNames[OP_COUNT] = {"conv", "dwconv3", "maxpool", "avgpool", "convt2", "padcat",
                                        "sigmoid", "toseq", "gru", "linear", "logsoftmax"};
}  // namespace

void CallbackModel::run(const float* input, const int64_t in_shape[4], std::vector<float>& out,
                        int64_t out_shape[4], int* out_ndim) const {
    float* o = nullptr;
    int64_t os[4] = {0, 0, 0, 0};
    int nd = 0;
    int rc = fn(user, input, in_shape, &o, os, &nd);
    if (rc != 0 || !o || nd < 1 || nd > 4) {
        if (o) free(o);
        fail(OCRS_ERR_RUN_FAILED, "model run failed: callback returned %d", rc);
    }
    int64_t cnt = 1;
    for (int i = 0; i < nd; i++) cnt *= os[i];
    out.assign(o, o + cnt);
    free(o);
    for (int i = 0; i < 4; i++) out_shape[i] = i < nd ? os[i] : 1;
    *out_ndim = nd;
}

std::unique_ptr<HipModel> HipModel::load(const void* data, size_t len, int device) {
    if (len < sizeof(FileHeader)) fail(OCRS_ERR_IO, "model file too short");
    FileHeader hd;
    memcpy(&hd, data, sizeof hd);
    if (memcmp(hd.magic, "OCRSMDL1", 8) != 0 || hd.version != 1) fail(OCRS_ERR_IO, "not an OCRSMDL1 model file");
    if (hd.n_ops > (1u << 20) || hd.n_slots == 0 || hd.n_slots > (1u << 20) || hd.out_slot >= hd.n_slots)
        fail(OCRS_ERR_IO, "model file header out of range (%u ops, %u slots, output slot %u)", hd.n_ops, hd.n_slots, hd.out_slot);
    const size_t table = sizeof(FileHeader) + (size_t)hd.n_ops * sizeof(FileOp);
    if (len < table || hd.blob_floats > (len - table) / sizeof(float)) fail(OCRS_ERR_IO, "model file truncated");
    const float* blob = reinterpret_cast<const float*>(static_cast<const char*>(data) + table);

    auto m = std::make_unique<HipModel>();
    m->kind = hd.kind;
    for (int i = 0; i < 4; i++) m->input_shape[i] = hd.input_shape[i];
    m->n_slots = hd.n_slots;
    m->out_slot = hd.out_slot;

    // Host image of the device slab: the file blob followed by derived tensors.
    std::vector<float> slab(blob, blob + hd.blob_floats);
    auto pad16 = [&]() { while (slab.size() % 4) slab.push_back(0.f); };
    pad16();
    struct Fix { size_t op; int which; size_t off; };
    std::vector<Fix> fixes;
    std::vector<FileOp> fops(hd.n_ops);
    for (uint32_t i = 0; i < hd.n_ops; i++) {
        memcpy(&fops[i], static_cast<const char*>(data) + sizeof(FileHeader) + (size_t)i * sizeof(FileOp), sizeof(FileOp));
        const FileOp& f = fops[i];
        if (f.type >= OP_COUNT || f.n_w > 8) fail(OCRS_ERR_IO, "bad op record %u", i);
        for (uint32_t j = 0; j < f.n_w; j++)
            if (f.w[j].off > hd.blob_floats || f.w[j].cnt > hd.blob_floats - f.w[j].off)
                fail(OCRS_ERR_IO, "weight reference out of range in op %u", i);
        // A malformed or stale file must fail here, not index out of bounds in infer()/run_device() or on the device:
        // slot numbers within the table, weight tensors of exactly the sizes the op's shape implies.
        {
            const int64_t ns = hd.n_slots;
            const bool two_in = f.type == OP_PADCAT;
            if (f.in0 < 0 || f.in0 >= ns || f.out <= 0 || f.out >= ns || (two_in ? (f.in1 < 0 || f.in1 >= ns) : false) ||
                (!two_in && f.in1 >= ns))
                fail(OCRS_ERR_IO, "slot index out of range in op %u", i);
            auto want = [&](uint32_t j, int64_t cnt) {
                if (j >= f.n_w || cnt < 0 || (int64_t)f.w[j].cnt != cnt)
                    fail(OCRS_ERR_IO, "op %u (%s): weight tensor %u has %llu floats, the op's shape needs %lld", i,
                         kOpNames[f.type], j, j < f.n_w ? (unsigned long long)f.w[j].cnt : 0ull, (long long)cnt);
            };
            auto pos = [&](int64_t v, const char* what) {
                if (v <= 0 || v > (1 << 20)) fail(OCRS_ERR_IO, "op %u (%s): bad %s %lld", i, kOpNames[f.type], what, (long long)v);
            };
            switch (f.type) {
                case OP_CONV: pos(f.kh, "kh"); pos(f.kw, "kw"); pos(f.cin, "cin"); pos(f.cout, "cout");
                    want(0, (int64_t)f.kh * f.kw * f.cin * f.cout); want(1, f.cout); break;
                case OP_DWCONV3: pos(f.cin, "channels"); want(0, 9LL * f.cin); want(1, f.cin); break;
                case OP_MAXPOOL: case OP_AVGPOOL: pos(f.kh, "kh"); pos(f.kw, "kw"); break;
                case OP_CONVT2: pos(f.cin, "cin"); pos(f.cout, "cout"); want(0, 4LL * f.cin * f.cout); want(1, f.cout); break;
                case OP_GRU: pos(f.cin, "input size"); pos(f.hidden, "hidden size");
                    if (f.n_w != 8) fail(OCRS_ERR_IO, "GRU op %u needs 8 weight tensors", i);
                    for (uint32_t d = 0; d < 2; d++) {
                        want(4 * d + 0, 3LL * f.cin * f.hidden); want(4 * d + 1, 3LL * f.hidden);
                        want(4 * d + 2, 3LL * f.hidden * f.hidden); want(4 * d + 3, 3LL * f.hidden);
                    }
                    break;
                case OP_LINEAR: pos(f.cin, "cin"); pos(f.cout, "cout"); want(0, (int64_t)f.cin * f.cout); want(1, f.cout); break;
                default: break;
            }
        }
        GraphOp op{};
        op.type = f.type; op.in0 = f.in0; op.in1 = f.in1; op.out = f.out;
        op.relu = f.relu; op.kh = f.kh; op.kw = f.kw; op.cin = f.cin; op.cout = f.cout; op.hidden = f.hidden;
        for (uint32_t j = 0; j < f.n_w; j++) op.wcount[j] = f.w[j].cnt;
        if (f.type == OP_CONVT2) {
            // file: [2][2][Cin][Cout] -> GEMM B [Cin][(dy,dx,co)], bias repeated per (dy,dx)
            const float* w = blob + f.w[0].off;
            const float* b = blob + f.w[1].off;
            fixes.push_back({i, 0, slab.size()});
            for (int ci = 0; ci < f.cin; ci++)
                for (int q = 0; q < 4; q++)
                    for (int co = 0; co < f.cout; co++) slab.push_back(w[((size_t)q * f.cin + ci) * f.cout + co]);
            pad16();
            fixes.push_back({i, 1, slab.size()});
            for (int q = 0; q < 4; q++)
                for (int co = 0; co < f.cout; co++) slab.push_back(b[co]);
            pad16();
        } else if (f.type == OP_GRU) {
            if (f.n_w != 8) fail(OCRS_ERR_IO, "GRU op %u needs 8 weight tensors", i);
            // pack both directions contiguously: Wi [2][I][3H], bi [2][3H], Wh [2][H][3H], bh [2][3H]
            for (int part = 0; part < 4; part++) {
                fixes.push_back({i, part, slab.size()});
                for (int d = 0; d < 2; d++) {
                    const float* src = blob + f.w[4 * d + part].off;
                    slab.insert(slab.end(), src, src + f.w[4 * d + part].cnt);
                }
                pad16();
            }
        }
        m->ops.push_back(op);
    }
    // everything above is validation on the host (a malformed file fails before any device work); from here on
    // the thread is bound to the device that will hold the weights
    DeviceScope bind(device);
    m->device = ctx().device;
    m->weights = DevBuf(slab.size() * sizeof(float));
    OCRS_HIP(hipMemcpy(m->weights.p, slab.data(), slab.size() * sizeof(float), hipMemcpyHostToDevice));
    const float* base = m->weights.as<float>();
    for (uint32_t i = 0; i < hd.n_ops; i++)
        for (uint32_t j = 0; j < fops[i].n_w; j++) m->ops[i].w[j] = base + fops[i].w[j].off;
    for (const Fix& fx : fixes) {
        GraphOp& op = m->ops[fx.op];
        const float* p = base + fx.off;
        if (fx.which == 0) op.aux0 = p;
        else if (fx.which == 1) op.aux1 = p;
        else if (fx.which == 2) op.aux2 = p;
        else op.aux3 = p;
    }
    // Relaxed numerics (ocrs_engine_params.numerics): the 3x3 convs of a recognition stack whose contraction can run on the
    // bf16 matrix cores carry their weights a second time, cut into three bf16 terms in the kernel's LDS layout
    // (split_mfma.hpp split_weights; 1.5x the fp32 bytes, a few MB per model).
    for (uint32_t i = 0; i < hd.n_ops; i++) {
        GraphOp& op = m->ops[i];
        if (m->kind == 1 && op.type == OP_CONV && op.kh == 3 && op.kw == 3 && (op.cin % 32) == 0 && (op.cout % 128) == 0) {
            std::vector<uint16_t> img;
            k::split_weights(slab.data() + fops[i].w[0].off, 9 * op.cin, op.cout, op.cout, &img);
            m->tapes.emplace_back(img.size() * sizeof(uint16_t));
            OCRS_HIP(hipMemcpy(m->tapes.back().p, img.data(), img.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            op.wsplit = m->tapes.back().as<uint16_t>();
        }
        // conv2 of the fused conv1 + conv2 launch (32 -> 64 channels): its own image (one tap per barrier)
        if (m->kind == 1 && op.type == OP_CONV && op.kh == 3 && op.kw == 3 && op.cin == 32 && op.cout == 64) {
            std::vector<uint16_t> img;
            k::conv12_split_weights(slab.data() + fops[i].w[0].off, &img);
            m->tapes.emplace_back(img.size() * sizeof(uint16_t));
            OCRS_HIP(hipMemcpy(m->tapes.back().p, img.data(), img.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            op.wsplit = m->tapes.back().as<uint16_t>();
        }
    }
    // the same for the GRU input projections ([I][3H] per direction; derived tensor aux0 = [2][I][3H] in the slab)
    for (const Fix& fx : fixes) {
        GraphOp& op = m->ops[fx.op];
        if (fx.which != 0 || op.type != OP_GRU || (op.cin % 64) != 0 || ((3 * op.hidden) % 128) != 0) continue;
        std::vector<uint16_t> both;
        for (int dir = 0; dir < 2; dir++) {
            std::vector<uint16_t> img;
            k::split_weights(slab.data() + fx.off + (size_t)dir * op.cin * 3 * op.hidden, op.cin, 3 * op.hidden, 3 * op.hidden, &img);
            both.insert(both.end(), img.begin(), img.end());
        }
        m->tapes.emplace_back(both.size() * sizeof(uint16_t));
        OCRS_HIP(hipMemcpy(m->tapes.back().p, both.data(), both.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        op.wsplit = m->tapes.back().as<uint16_t>();
    }
    // Fold SIGMOID into a directly preceding Cout==1 pointwise conv.
    for (size_t i = 0; i + 1 < m->ops.size(); i++) {
        GraphOp& a = m->ops[i];
        GraphOp& b = m->ops[i + 1];
        if (a.type == OP_CONV && a.cout == 1 && a.kh == 1 && a.kw == 1 && b.type == OP_SIGMOID && b.in0 == a.out) {
            bool other_use = false;
            for (size_t j = i + 2; j < m->ops.size(); j++)
                if (m->ops[j].in0 == a.out || m->ops[j].in1 == a.out) other_use = true;
            if (!other_use && (uint32_t)a.out != m->out_slot) b.fused_into_prev = true;
        }
    }
    // Depthwise 3x3 followed by its pointwise 1x1 (both 8 <= C <= 32) run as one kernel.
    for (size_t i = 0; i + 1 < m->ops.size(); i++) {
        GraphOp& a = m->ops[i];
        GraphOp& b = m->ops[i + 1];
        if (a.type == OP_DWCONV3 && b.type == OP_CONV && b.kh == 1 && b.kw == 1 && b.in0 == a.out && b.cin == a.cin &&
            k::dwpw_fused_supported(a.cin, b.cout) && (uint32_t)a.out != m->out_slot) {
            bool other_use = false;
            for (size_t j = i + 2; j < m->ops.size(); j++)
                if (m->ops[j].in0 == a.out || m->ops[j].in1 == a.out) other_use = true;
            if (!other_use) { a.fuse_next_pw = true; b.done_by_prev = true; }
        }
    }
    // PADCAT consumed only by the depthwise conv that follows: that conv reads skip / up directly.
    for (size_t i = 0; i + 1 < m->ops.size(); i++) {
        GraphOp& a = m->ops[i];
        GraphOp& b = m->ops[i + 1];
        if (a.type == OP_PADCAT && b.type == OP_DWCONV3 && b.in0 == a.out && (uint32_t)a.out != m->out_slot) {
            bool other_use = false;
            for (size_t j = i + 2; j < m->ops.size(); j++)
                if (m->ops[j].in0 == a.out || m->ops[j].in1 == a.out) other_use = true;
            if (!other_use) { a.cat_into_next = true; b.reads_cat = true; }
        }
    }
    // Whole DoubleConv blocks of the detection U-Net as one fused launch each (kernels_det.hip).
    {
        auto& ops = m->ops;
        const int n = (int)ops.size();
        auto uses = [&](int slot) {
            int u = 0;
            for (const GraphOp& o : ops) u += (o.in0 == slot) + (o.in1 == slot);
            return u + ((uint32_t)slot == m->out_slot ? 1 : 0);
        };
        auto pw11 = [](const GraphOp& o) { return o.type == OP_CONV && o.kh == 1 && o.kw == 1; };
        for (int i = 0; i + 3 < n; i++) {
            DcBlock b;
            int j = i;
            if (ops[j].type == OP_CONVT2 && j + 1 < n && ops[j + 1].type == OP_PADCAT && ops[j + 1].in1 == ops[j].out &&
                uses(ops[j].out) == 1 && uses(ops[j + 1].out) == 1) {
                b.convt = j; b.cat = j + 1;
                j += 2;
            }
            if (j + 3 >= n) continue;
            const GraphOp &d1 = ops[j], &p1 = ops[j + 1], &d2 = ops[j + 2], &p2 = ops[j + 3];
            if (d1.type != OP_DWCONV3 || !pw11(p1) || d2.type != OP_DWCONV3 || !pw11(p2)) continue;
            if (b.cat >= 0 && d1.in0 != ops[b.cat].out) continue;
            if (p1.in0 != d1.out || d2.in0 != p1.out || p2.in0 != d2.out) continue;
            if (uses(d1.out) != 1 || uses(p1.out) != 1 || uses(d2.out) != 1) continue;
            if (p1.cin != d1.cin || d2.cin != p1.cout || p2.cin != d2.cin) continue;
            b.dw1 = j; b.pw1 = j + 1; b.dw2 = j + 2; b.pw2 = j + 3;
            b.first = i; b.last = j + 3;
            b.cmid = p1.cout; b.cout = p2.cout;
            if (b.convt >= 0) {
                b.cx = ops[b.convt].cin;
                b.cs = d1.cin - ops[b.convt].cout;
                if (b.cs != ops[b.convt].cout) continue;   // the kernels assume ConvT cout == skip channels
            } else {
                b.cs = d1.cin;
            }
            const int k = j + 4;
            if (k < n && ops[k].type == OP_MAXPOOL && ops[k].kh == 2 && ops[k].kw == 2 && ops[k].in0 == p2.out && b.convt < 0) {
                b.pool = k; b.last = k;
            } else if (k < n && pw11(ops[k]) && ops[k].cout == 1 && ops[k].in0 == p2.out && uses(p2.out) == 1 && b.convt >= 0) {
                b.fin = k; b.last = k;
                if (k + 1 < n && ops[k + 1].type == OP_SIGMOID && ops[k + 1].fused_into_prev) { b.sig = k + 1; b.last = k + 1; }
                else if (uses(ops[k].out) == 0) continue;
            }
            k::DoubleConvArgs none{};
            if (!k::double_conv_fused(none, b.cs, b.cx, b.cmid, b.cout, b.pool >= 0, b.fin >= 0, 2, false, nullptr)) continue;
            {   // a row-streaming kernel for the shape: its weights laid out as the tape a row step reads (kernels_det_stream.hip)
                auto host = [&](int op, int j) -> const float* { return op >= 0 ? slab.data() + fops[op].w[j].off : nullptr; };
                k::StreamWeights hw{host(b.convt, 0), host(b.convt, 1), host(b.dw1, 0), host(b.dw1, 1), host(b.pw1, 0), host(b.pw1, 1),
                                    host(b.dw2, 0), host(b.dw2, 1), host(b.pw2, 0), host(b.pw2, 1), host(b.fin, 0), host(b.fin, 1)};
                std::vector<float> tape;
                int tape_len = 0;
                if (k::double_conv_stream(none, b.cs, b.cx, b.cmid, b.cout, b.pool >= 0, b.fin >= 0, false, nullptr, &hw, &tape, &tape_len)) {
                    m->tapes.emplace_back(tape.size() * sizeof(float));
                    OCRS_HIP(hipMemcpy(m->tapes.back().p, tape.data(), tape.size() * sizeof(float), hipMemcpyHostToDevice));
                    b.tape = m->tapes.back().as<float>();
                    b.tape_len = tape_len;
                }
                if (k::double_conv_rows(none, b.cs, b.cx, b.cmid, b.cout, b.pool >= 0, b.fin >= 0, false, nullptr, &hw, &tape, &tape_len)) {
                    m->tapes.emplace_back(tape.size() * sizeof(float));
                    OCRS_HIP(hipMemcpy(m->tapes.back().p, tape.data(), tape.size() * sizeof(float), hipMemcpyHostToDevice));
                    b.rtape = m->tapes.back().as<float>();
                    b.rtape_len = tape_len;
                }
            }
            ops[i].dc_block = (int)m->dc_blocks.size();
            m->dc_blocks.push_back(b);
            i = b.last;   // blocks do not overlap
        }
    }
    return m;
}

TensorShape HipModel::infer(int n, int h, int w, std::vector<TensorShape>* slots_out) const {
    std::vector<TensorShape> s(n_slots);
    s[0] = TensorShape{n, h, w, 1, false};
    for (const GraphOp& op : ops) {
        const TensorShape a = s[op.in0];
        TensorShape o = a;
        switch (op.type) {
            case OP_CONV: o.c = op.cout; break;
            case OP_DWCONV3: break;
            case OP_MAXPOOL:
            case OP_AVGPOOL: o.h = a.h / op.kh; o.w = a.w / op.kw; break;
            case OP_CONVT2: o.h = 2 * a.h; o.w = 2 * a.w; o.c = op.cout; break;
            case OP_PADCAT: o.c = a.c + s[op.in1].c; break;
            case OP_SIGMOID:
            case OP_LOGSOFTMAX: break;
            case OP_TOSEQ: o = TensorShape{a.w, a.n, 1, a.c, true}; break;  // [T,N,C]
            case OP_GRU: o.c = 2 * op.hidden; break;
            case OP_LINEAR: o.c = op.cout; break;
            default: fail(OCRS_ERR_IO, "bad op type %u", op.type);
        }
        s[op.out] = o;
    }
    TensorShape out = s[out_slot];
    if (slots_out) *slots_out = std::move(s);
    return out;
}

double HipModel::flops(int n, int h, int w) const {
    std::vector<TensorShape> s;
    infer(n, h, w, &s);
    double total = 0;
    for (const GraphOp& op : ops) {
        const TensorShape a = s[op.in0];
        const double px = (double)a.n * a.h * a.w;
        switch (op.type) {
            case OP_CONV: total += 2.0 * px * op.cout * op.cin * op.kh * op.kw; break;
            case OP_DWCONV3: total += 2.0 * px * a.c * 9; break;
            case OP_CONVT2: total += 2.0 * px * op.cin * op.cout * 4; break;
            case OP_GRU: total += 2.0 * 2.0 * px * 3 * op.hidden * (op.cin + op.hidden); break;
            case OP_LINEAR: total += 2.0 * px * op.cin * op.cout; break;
            default: break;
        }
    }
    return total;
}

float* HipModel::run_device(Workspace& ws, const float* d_in, int n, int h, int w, TensorShape* out_shape,
                            StageTimers* timers, const uint8_t* d_excluded, int32_t* d_labels, bool want_logp,
                            bool print_timing, int stop_before, hipStream_t exec) const {
    std::vector<TensorShape> shp;
    TensorShape out = infer(n, h, w, &shp);
    const size_t n_run = stop_before >= 0 ? (size_t)stop_before : ops.size();
    const uint32_t ret_slot = stop_before >= 0 ? (uint32_t)ops[stop_before].in0 : out_slot;
    if (stop_before >= 0) out = shp[ret_slot];
    if (out_shape) *out_shape = out;
    hipStream_t st = exec ? exec : ws.s();

    // liveness: last op index that reads each slot
    std::vector<int> last_use(n_slots, -1);
    for (size_t i = 0; i < ops.size(); i++) {
        last_use[ops[i].in0] = (int)i;
        if (ops[i].in1 >= 0) last_use[ops[i].in1] = (int)i;
    }
    last_use[out_slot] = (int)ops.size() + 1;
    last_use[ret_slot] = (int)ops.size() + 1;
    auto cat_fused = [&](size_t i) {  // PADCAT i is skipped and op i+1 reads its inputs (only inside this run's range)
        return ops[i].cat_into_next && i + 1 < n_run && (int)ret_slot != ops[i].out;
    };
    for (size_t i = 0; i + 1 < ops.size(); i++)
        if (cat_fused(i)) {
            last_use[ops[i].in0] = std::max(last_use[ops[i].in0], (int)i + 1);
            last_use[ops[i].in1] = std::max(last_use[ops[i].in1], (int)i + 1);
        }
    std::vector<float*> ptr(n_slots, nullptr);
    std::vector<size_t> cap(n_slots, 0);
    std::multimap<size_t, float*> free_local;
    auto get = [&](size_t floats) -> std::pair<float*, size_t> {
        size_t bytes = floats * sizeof(float);
        auto it = free_local.lower_bound(bytes);
        if (it != free_local.end() && it->first <= bytes * 2 + 4096) {
            auto r = std::make_pair(it->second, it->first);
            free_local.erase(it);
            return r;
        }
        size_t rounded = (bytes + 255) & ~size_t(255);
        return {static_cast<float*>(ws.alloc(rounded)), rounded};
    };
    ptr[0] = const_cast<float*>(d_in);

    std::vector<hipEvent_t> ev;
    if (print_timing) {
        ev.resize(ops.size() + 1);
        for (auto& e : ev) OCRS_HIP(hipEventCreate(&e));
        OCRS_HIP(hipEventRecord(ev[0], st));
    }

    int stage_token = -1;
    int cur_stage = -1;
    auto enter_stage = [&](int stage) {
        if (!timers || stage == cur_stage) return;
        if (cur_stage >= 0) timers->end(stage_token, st);
        stage_token = timers->begin(stage, st, 0);
        cur_stage = stage;
    };
    bool seen_seq = false;
    // per-launch kernel timing (only when the engine asked for it)
    auto timed = [&](int cls, double flops, double bytes, auto&& launch, double mfma_flops = -1.0) {
        int tok = timers ? timers->kbegin(cls, st, flops, bytes, mfma_flops) : -1;
        launch();
        if (tok >= 0) timers->end(tok, st);
    };
    auto wbytes = [](const GraphOp& o, int j) { return (double)o.wcount[j] * 4.0; };

    std::vector<char> covered(ops.size(), 0);   // ops already done by a fused DoubleConv launch
    const int det_fuse = option(OPT_DET_FUSE);
    for (size_t i = 0; i < n_run; i++) {
        const GraphOp& op = ops[i];
        const TensorShape a = shp[op.in0];
        const TensorShape o = shp[op.out];
        // one stage per op, chosen once: detection graphs are one stage; recognition graphs switch at TOSEQ
        if (kind == 0) enter_stage(ST_DET_CNN);
        else if (op.type == OP_GRU) enter_stage(ST_REC_GRU);
        else if (op.type == OP_LINEAR || op.type == OP_LOGSOFTMAX) enter_stage(seen_seq ? ST_REC_HEAD : ST_REC_CONV);
        else enter_stage(seen_seq ? ST_REC_GRU : ST_REC_CONV);
        if (op.type == OP_TOSEQ) seen_seq = true;
        if (covered[i]) {
            for (int sl = 1; sl < (int)n_slots; sl++)   // any slot whose last reader this op was (incl. a skipped PADCAT's inputs)
                if (last_use[sl] == (int)i && ptr[sl] && cap[sl]) {
                    free_local.emplace(cap[sl], ptr[sl]);
                    ptr[sl] = nullptr;
                    cap[sl] = 0;
                }
            if (print_timing) OCRS_HIP(hipEventRecord(ev[i + 1], st));
            continue;
        }
        if (det_fuse && op.dc_block >= 0) {
            const DcBlock& b = dc_blocks[op.dc_block];
            bool ok = (size_t)b.last < n_run;
            for (int q = b.first; ok && q < b.last; q++)   // no intermediate may be what this run returns
                if ((uint32_t)ops[q].out == ret_slot && q != b.pw2) ok = false;
            if (ok && b.fin >= 0 && (uint32_t)ops[b.pw2].out == ret_slot) ok = false;
            if (ok) {
                const GraphOp &d1 = ops[b.dw1], &p1 = ops[b.pw1], &d2 = ops[b.dw2], &p2 = ops[b.pw2];
                const int skip_slot = b.cat >= 0 ? ops[b.cat].in0 : d1.in0;
                const TensorShape sk = shp[skip_slot];
                k::DoubleConvArgs da{};
                da.skip = ptr[skip_slot];
                da.n = sk.n; da.h = sk.h; da.w = sk.w;
                double px1 = 0;
                if (b.convt >= 0) {
                    const GraphOp& ct = ops[b.convt];
                    const TensorShape x1 = shp[ct.in0];
                    da.x1 = ptr[ct.in0]; da.h1 = x1.h; da.w1 = x1.w;
                    da.wt = ct.w[0]; da.bt = ct.w[1];
                    px1 = (double)x1.n * x1.h * x1.w;
                    if (2 * x1.h > sk.h || 2 * x1.w > sk.w) ok = false;
                }
                da.wd1 = d1.w[0]; da.bd1 = d1.w[1]; da.wp1 = p1.w[0]; da.bp1 = p1.w[1];
                da.wd2 = d2.w[0]; da.bd2 = d2.w[1]; da.wp2 = p2.w[0]; da.bp2 = p2.w[1];
                da.relu_d1 = d1.relu; da.relu_p1 = p1.relu; da.relu_d2 = d2.relu; da.relu_p2 = p2.relu;
                da.tape = b.tape; da.tape_len = b.tape_len; da.rtape = b.rtape; da.rtape_len = b.rtape_len;
                // Is there a fused kernel for THIS request (its page count and sizes, not just the block's channel counts)?  A
                // query with the real arguments: the row-streaming kernels decline some requests (more than 8 pages, an odd pad
                // offset) and two block shapes have no other fused kernel when option det_mfma is 0 — those fall through to the
                // per-operator kernels below.
                bool on_mfma = false;
                int path = 0;   // which kernel family will take it (nothing is launched by the query)
                if (ok) ok = k::double_conv_fused(da, b.cs, b.cx, b.cmid, b.cout, b.pool >= 0, b.fin >= 0, det_fuse, false, nullptr, &on_mfma, &path);
                const double px = (double)sk.n * sk.h * sk.w;
                double out_floats = 0;
                if (ok) {
                    if (b.fin >= 0) {
                        const int fo = b.sig >= 0 ? ops[b.sig].out : ops[b.fin].out;
                        auto r = get((size_t)shp[fo].count());
                        ptr[fo] = r.first; cap[fo] = r.second;
                        da.y = r.first; da.wf = ops[b.fin].w[0]; da.bf = ops[b.fin].w[1]; da.sigmoid = b.sig >= 0;
                        out_floats = (double)shp[fo].count();
                    } else {
                        auto r = get((size_t)shp[p2.out].count());
                        ptr[p2.out] = r.first; cap[p2.out] = r.second;
                        da.y = r.first;
                        out_floats = (double)shp[p2.out].count();
                        if (b.pool >= 0) {
                            auto rp = get((size_t)shp[ops[b.pool].out].count());
                            ptr[ops[b.pool].out] = rp.first; cap[ops[b.pool].out] = rp.second;
                            da.ypool = rp.first;
                            out_floats += (double)shp[ops[b.pool].out].count();
                        }
                    }
                    const int cin = d1.cin;
                    const double fl = 2.0 * px * (9.0 * cin + (double)cin * b.cmid + 9.0 * b.cmid + (double)b.cmid * b.cout +
                                                  (b.fin >= 0 ? b.cout : 0)) + 2.0 * px * (b.convt >= 0 ? (double)b.cx * b.cs : 0.0);
                    // of which dense contractions (pointwise convs, ConvTranspose) — on the matrix cores if the block's MFMA variant runs
                    const double fl_dense = 2.0 * px * ((double)cin * b.cmid + (double)b.cmid * b.cout) +
                                            2.0 * px * (b.convt >= 0 ? (double)b.cx * b.cs : 0.0);
                    bool launched = false;
                    timed(path == 1 ? KC_DET_STREAM_WAVE : path == 2 ? KC_DET_STREAM_ROWS : KC_DET_BLOCK, fl, 4.0 * (px * b.cs + px1 * b.cx + out_floats), [&] {
                        launched = k::double_conv_fused(da, b.cs, b.cx, b.cmid, b.cout, b.pool >= 0, b.fin >= 0, det_fuse, true, st);
                    }, on_mfma ? fl_dense : 0.0);
                    if (!launched) fail(OCRS_ERR_RUN_FAILED, "model run failed: internal (no fused kernel took the DoubleConv block its query accepted)");
                    for (int q = b.first + 1; q <= b.last; q++) covered[q] = 1;
                    for (int sl = 1; sl < (int)n_slots; sl++)
                        if (last_use[sl] == (int)i && ptr[sl] && cap[sl]) {
                            free_local.emplace(cap[sl], ptr[sl]);
                            ptr[sl] = nullptr;
                            cap[sl] = 0;
                        }
                    if (print_timing) OCRS_HIP(hipEventRecord(ev[i + 1], st));
                    continue;
                }
            }
        }

        const float* x = ptr[op.in0];
        float* y = nullptr;
        const bool is_final_logsoftmax = op.type == OP_LOGSOFTMAX && (uint32_t)op.out == out_slot;
        if (op.fused_into_prev) {
            ptr[op.out] = ptr[op.in0];  // already holds sigmoid(conv)
            cap[op.out] = cap[op.in0];
            cap[op.in0] = 0;
            ptr[op.in0] = nullptr;
        } else if (op.type == OP_PADCAT && cat_fused(i)) {
            // nothing to do: the depthwise conv that follows reads skip / up directly
        } else if (op.done_by_prev && ptr[op.out]) {
            // computed together with the preceding depthwise conv
        } else if (op.fuse_next_pw && !op.reads_cat && i + 1 < n_run && (int)ret_slot != op.out) {
            const GraphOp& pw = ops[i + 1];
            const TensorShape po = shp[pw.out];
            auto r = get((size_t)po.count());
            ptr[pw.out] = r.first;
            cap[pw.out] = r.second;
            const double px = (double)a.n * a.h * a.w;
            timed(KC_DWCONV3X3, 2.0 * px * (9.0 * op.cin + (double)op.cin * pw.cout), 4.0 * px * (op.cin + pw.cout), [&] {
                k::dwpw_fused(x, a.n, a.h, a.w, op.cin, op.w[0], op.w[1], op.relu, pw.cout, pw.w[0], pw.w[1], pw.relu, r.first, st);
            });
        } else {
        if (!(is_final_logsoftmax && !want_logp)) {
            auto r = get((size_t)o.count());
            y = r.first;
            ptr[op.out] = y;
            cap[op.out] = r.second;
        }
        switch (op.type) {
            case OP_CONV: {
                const int64_t px = (int64_t)a.n * a.h * a.w;
                const bool next_sigmoid = i + 1 < ops.size() && ops[i + 1].fused_into_prev;
                const double cflops = 2.0 * px * op.cout * op.cin * op.kh * op.kw;
                const double cbytes = 4.0 * (double)px * (op.cin + op.cout) + wbytes(op, 0);
                if (op.kh == 1 && op.kw == 1 && op.cout == 1) {
                    timed(KC_CONV1X1_SIGMOID, cflops, cbytes,
                          [&] { k::conv1x1_cout1(x, px, op.cin, op.w[0], op.w[1], next_sigmoid ? 1 : 0, y, st); });
                } else if (op.kh == 1 && op.kw == 1 && (op.cin % 4) == 0) {
                    k::GemmDesc d{};
                    d.A = x; d.lda = op.cin; d.B = op.w[0]; d.ldb = op.cout; d.bias = op.w[1];
                    d.C = y; d.ldc = op.cout; d.M = (int)px; d.N = op.cout; d.K = op.cin; d.relu = op.relu;
                    timed(KC_GEMM_POINTWISE, cflops, cbytes, [&] { k::gemm(d, st); });
                } else if (op.kh == 3 && op.kw == 3 && (op.cin % 32) == 0) {
                    k::GemmDesc d{};
                    d.A = x; d.B = op.w[0]; d.ldb = op.cout; d.bias = op.w[1];
                    d.C = y; d.ldc = op.cout; d.M = (int)px; d.N = op.cout; d.K = 9 * op.cin; d.relu = op.relu;
                    d.im2col = 1; d.H = a.h; d.W = a.w; d.Cin = op.cin;
                    timed(KC_GEMM_CONV3X3, cflops, cbytes, [&] { k::gemm(d, st); });
                } else if ((op.cout % 4) == 0) {
                    timed(KC_CONV_DIRECT, cflops, cbytes, [&] {
                        k::conv_direct(x, a.n, a.h, a.w, op.cin, op.w[0], op.w[1], op.kh, op.kw, op.cout, op.relu, y, st);
                    });
                } else {
                    fail(OCRS_ERR_RUN_FAILED, "model run failed: unsupported conv shape %dx%d %d->%d", op.kh, op.kw,
                         op.cin, op.cout);
                }
                break;
            }
            case OP_DWCONV3:
                if (op.reads_cat && i > 0 && cat_fused(i - 1)) {
                    const GraphOp& cat = ops[i - 1];
                    const TensorShape sk = shp[cat.in0], up = shp[cat.in1];
                    if ((sk.c % 4) == 0 && (up.c % 4) == 0) {
                        timed(KC_DWCONV3X3, 18.0 * a.count(), 8.0 * a.count(), [&] {
                            k::dwconv3x3_cat(ptr[cat.in0], sk.n, sk.h, sk.w, sk.c, ptr[cat.in1], up.h, up.w, up.c, op.w[0],
                                             op.w[1], op.relu, y, st);
                        });
                    } else {  // channel counts the vectorised kernel does not take: build the concatenation after all
                        auto tmp = get((size_t)a.count());
                        timed(KC_PADCAT, 0, 8.0 * a.count(), [&] {
                            k::padcat(ptr[cat.in0], sk.n, sk.h, sk.w, sk.c, ptr[cat.in1], up.h, up.w, up.c, tmp.first, st);
                        });
                        timed(KC_DWCONV3X3, 18.0 * a.count(), 8.0 * a.count(),
                              [&] { k::dwconv3x3(tmp.first, a.n, a.h, a.w, a.c, op.w[0], op.w[1], op.relu, y, st); });
                        free_local.emplace(tmp.second, tmp.first);
                    }
                    for (int sl : {cat.in0, cat.in1})  // the skipped PADCAT's inputs: this was their last reader
                        if (sl > 0 && last_use[sl] == (int)i && ptr[sl] && cap[sl]) {
                            free_local.emplace(cap[sl], ptr[sl]);
                            ptr[sl] = nullptr;
                            cap[sl] = 0;
                        }
                    break;
                }
                timed(KC_DWCONV3X3, 18.0 * a.count(), 8.0 * a.count(),
                      [&] { k::dwconv3x3(x, a.n, a.h, a.w, a.c, op.w[0], op.w[1], op.relu, y, st); });
                break;
            case OP_MAXPOOL:
                timed(KC_POOL, 0, 4.0 * (a.count() + o.count()), [&] { k::maxpool(x, a.n, a.h, a.w, a.c, op.kh, op.kw, y, st); });
                break;
            case OP_AVGPOOL:
                timed(KC_POOL, 0, 4.0 * (a.count() + o.count()), [&] { k::avgpool(x, a.n, a.h, a.w, a.c, op.kh, op.kw, y, st); });
                break;
            case OP_CONVT2: {
                k::GemmDesc d{};
                d.A = x; d.lda = op.cin; d.B = op.aux0; d.ldb = 4 * op.cout; d.bias = op.aux1;
                d.C = y; d.M = (int)((int64_t)a.n * a.h * a.w); d.N = 4 * op.cout; d.K = op.cin;
                d.convt = 1; d.H = a.h; d.W = a.w; d.Cout = op.cout;
                timed(KC_GEMM_CONVT, 2.0 * d.M * op.cin * op.cout * 4, 4.0 * (a.count() + o.count()) + wbytes(op, 0),
                      [&] { k::gemm(d, st); });
                break;
            }
            case OP_PADCAT: {
                const TensorShape b = shp[op.in1];
                timed(KC_PADCAT, 0, 8.0 * o.count(),
                      [&] { k::padcat(x, a.n, a.h, a.w, a.c, ptr[op.in1], b.h, b.w, b.c, y, st); });
                break;
            }
            case OP_SIGMOID: timed(KC_OTHER, 0, 8.0 * a.count(), [&] { k::sigmoid(x, y, a.count(), st); }); break;
            case OP_TOSEQ:
                if (a.h != 1) fail(OCRS_ERR_RUN_FAILED, "model run failed: TOSEQ expects height 1, got %d", a.h);
                timed(KC_OTHER, 0, 8.0 * a.count(), [&] { k::to_seq(x, a.n, a.w, a.c, y, st); });
                break;
            case OP_GRU: {
                const int T = a.n, N = a.h, I = a.c, H = op.hidden;
                auto gx = get((size_t)2 * T * N * 3 * H);
                auto gh = get((size_t)2 * N * 3 * H);
                auto hs = get((size_t)2 * N * H);
                OCRS_HIP(hipMemsetAsync(hs.first, 0, (size_t)2 * N * H * sizeof(float), st));
                k::GemmDesc d{};
                d.A = x; d.lda = I; d.B = op.aux0; d.ldb = 3 * H; d.bias = op.aux1; d.C = gx.first; d.ldc = 3 * H;
                d.M = T * N; d.N = 3 * H; d.K = I; d.batch = 2;
                d.strideA = 0; d.strideB = (int64_t)I * 3 * H; d.strideBias = 3 * H; d.strideC = (int64_t)T * N * 3 * H;
                timed(KC_GEMM_GRU_INPUT, 2.0 * 2 * d.M * (double)d.N * d.K,
                      4.0 * ((double)a.count() + 2.0 * d.M * d.N + 2.0 * d.K * d.N), [&] { k::gemm(d, st); });
                k::GemmDesc r{};
                r.A = hs.first; r.lda = H; r.B = op.aux2; r.ldb = 3 * H; r.bias = op.aux3; r.C = gh.first; r.ldc = 3 * H;
                r.M = N; r.N = 3 * H; r.K = H; r.batch = 2;
                r.strideA = (int64_t)N * H; r.strideB = (int64_t)H * 3 * H; r.strideBias = 3 * H; r.strideC = (int64_t)N * 3 * H;
                for (int step = 0; step < T; step++) {
                    timed(KC_GEMM_GRU_HIDDEN, 2.0 * 2 * r.M * (double)r.N * r.K,
                          4.0 * 2 * ((double)r.M * r.K + (double)r.M * r.N + (double)r.K * r.N), [&] { k::gemm(r, st); });
                    timed(KC_GRU_GATES, 0, 4.0 * 2 * N * (double)H * 9,
                          [&] { k::gru_gates(gx.first, gh.first, hs.first, y, T, N, H, step, st); });
                }
                free_local.emplace(gx.second, gx.first);
                free_local.emplace(gh.second, gh.first);
                free_local.emplace(hs.second, hs.first);
                break;
            }
            case OP_LINEAR: {
                k::GemmDesc d{};
                d.A = x; d.lda = op.cin; d.B = op.w[0]; d.ldb = op.cout; d.bias = op.w[1];
                d.C = y; d.ldc = op.cout; d.M = (int)((int64_t)a.n * a.h * a.w); d.N = op.cout; d.K = op.cin;
                d.relu = op.relu;
                timed(KC_GEMM_LINEAR, 2.0 * d.M * (double)d.N * d.K, 4.0 * ((double)a.count() + o.count()) + wbytes(op, 0),
                      [&] { k::gemm(d, st); });
                break;
            }
            case OP_LOGSOFTMAX:
                timed(KC_LOGSOFTMAX_ARGMAX, 0, 4.0 * a.count() * (y ? 2.0 : 1.0), [&] {
                    if (!k::log_softmax_argmax(x, (int64_t)a.n * a.h * a.w, a.c, is_final_logsoftmax ? d_excluded : nullptr, y,
                                               is_final_logsoftmax ? d_labels : nullptr, st))
                        fail(OCRS_ERR_CAPACITY, "LogSoftmax over %d classes exceeds the kernel's LDS staging (max ~630)", a.c);
                });
                break;
            default: fail(OCRS_ERR_RUN_FAILED, "model run failed: bad op");
        }
        }
        if (print_timing) OCRS_HIP(hipEventRecord(ev[i + 1], st));
        // release inputs whose last reader this was (slot 0 is the caller's)
        for (int sl : {op.in0, op.in1}) {
            if (sl > 0 && last_use[sl] == (int)i && ptr[sl] && cap[sl]) {
                free_local.emplace(cap[sl], ptr[sl]);
                ptr[sl] = nullptr;
                cap[sl] = 0;
            }
        }
    }
    if (timers && cur_stage >= 0) timers->end(stage_token, st);
    OCRS_HIP(hipGetLastError());

    if (print_timing) {
        OCRS_HIP(hipStreamSynchronize(st));
        float total = 0.f;
        for (size_t i = 0; i < n_run; i++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            total += ms;
            const TensorShape o = shp[ops[i].out];
            printf("%-10s [%d,%d,%d,%d] %.3fms\n", kOpNames[ops[i].type], o.n, o.h, o.w, o.c, ms);
        }
        printf("total %.3fms\n", total);
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    return ptr[ret_slot];
}

// ---------------------------------------------------------------------------
// Ragged recognition batch
// ---------------------------------------------------------------------------
float* HipModel::run_prefix_ragged(Workspace& ws, hipStream_t exec, const std::vector<PackedGroup>& groups,
                                   const PackedPlan& plan, int h, int ts, StageTimers* timers, int* feat_c,
                                   const std::function<void()>& before_launch) const {
    // supported stack: CONV 3x3 (Cin == 1 directly followed by MAXPOOL 2x2, or Cin % 32 == 0), MAXPOOL, AVGPOOL,
    // each consuming the previous op's output
    const int G = (int)groups.size();
    if (G == 0 || ts <= 0) return nullptr;
    for (int i = 0; i < ts; i++) {
        const GraphOp& op = ops[i];
        if (op.in0 != (i == 0 ? 0 : ops[i - 1].out)) return nullptr;
        if (op.type == OP_CONV) {
            if (op.kh != 3 || op.kw != 3 || !op.relu) return nullptr;
            if (op.cin == 1) {
                if (i + 1 >= ts || ops[i + 1].type != OP_MAXPOOL || ops[i + 1].kh != 2 || ops[i + 1].kw != 2) return nullptr;
                if (op.cout > 64 || op.cout < 4 || (op.cout & (op.cout - 1)) != 0) return nullptr;  // 4, 8, 16, 32, 64
            } else if ((op.cin % 32) != 0 || op.cout < 64 || (op.cout % 4) != 0) {
                return nullptr;
            }
        } else if (op.type != OP_MAXPOOL && op.type != OP_AVGPOOL) {
            return nullptr;
        }
    }
    {  // the patch-tiled conv needs H % 4 == 0 at every Cin > 1 conv
        int hh = h;
        for (int i = 0; i < ts; i++) {
            if (ops[i].type == OP_CONV && ops[i].cin != 1 && (hh % 4) != 0) return nullptr;
            if (ops[i].type == OP_MAXPOOL || ops[i].type == OP_AVGPOOL) hh /= ops[i].kh;
        }
    }
    for (int g = 0; g + 1 < G; g++)  // groups must be contiguous in memory
        if (groups[g].d_batch + (size_t)groups[g].n * h * groups[g].w != groups[g + 1].d_batch) return nullptr;

    hipStream_t st = exec;
    auto timed = [&](int cls, double flops, double bytes, auto&& launch) {
        int tok = timers ? timers->kbegin(cls, st, flops, bytes) : -1;
        launch();
        if (tok >= 0) timers->end(tok, st);
    };
    // ---- per-layer geometry on the host, one metadata upload
    struct Geo { int H; std::vector<int32_t> W; };
    std::vector<Geo> geos;  // geos[0] = input; geos[i+1] = output of op i
    geos.push_back(Geo{h, {}});
    for (const PackedGroup& g : groups) geos[0].W.push_back(g.w);
    for (int i = 0; i < ts; i++) {
        Geo o = geos.back();
        if (ops[i].type == OP_MAXPOOL || ops[i].type == OP_AVGPOOL) {
            o.H /= ops[i].kh;
            for (auto& w : o.W) w /= ops[i].kw;
        }
        geos.push_back(std::move(o));
    }
    std::vector<int32_t> nvec;
    for (const PackedGroup& g : groups) nvec.push_back(g.n);
    // layout of the metadata blob per geometry: W[G] | toff128[G+1] | toff256[G+1] | poff[G+1] (int64)
    std::vector<int32_t> meta32;
    std::vector<int64_t> meta64;
    struct GeoOff { size_t w, t128, t256, t32, t16, p, f32[2], f16[2], gap; int nt128, nt256, nt32, nt16, nf32[2], nf16[2], ngap; int64_t pixels; };
    std::vector<GeoOff> goff;
    const size_t n_at = 0;
    meta32.insert(meta32.end(), nvec.begin(), nvec.end());
    const size_t loff_at = meta32.size();
    {
        int32_t acc = 0;
        meta32.push_back(0);
        for (int g = 0; g < G; g++) { acc += nvec[g]; meta32.push_back(acc); }
    }
    for (const Geo& ge : geos) {
        GeoOff o{};
        o.w = meta32.size();
        meta32.insert(meta32.end(), ge.W.begin(), ge.W.end());
        o.t128 = meta32.size();
        int32_t a128 = 0, a256 = 0, a32 = 0, a16 = 0;
        int64_t px = 0;
        std::vector<int32_t> t256{0}, t32{0}, t16{0};  // t32 / t16: 4 x 32 and 8 x 16 pixel patches (conv3x3_ragged)
        // the same patches over the group's whole strip of images; [1]: image widths rounded up to even
        std::vector<int32_t> f32[2] = {{0}, {0}}, f16[2] = {{0}, {0}};
        int32_t af32[2] = {0, 0}, af16[2] = {0, 0};
        std::vector<int32_t> tgap{0};   // 8 x 16 patches, images one empty column (at least) apart (conv12_fused_ragged)
        int32_t agap = 0;
        std::vector<int64_t> pv{0};
        meta32.push_back(0);
        for (int g = 0; g < G; g++) {
            const int64_t rows = (int64_t)nvec[g] * ge.H * ge.W[g];
            a128 += (int32_t)((rows + 127) / 128);
            a256 += (int32_t)((rows + 255) / 256);
            a32 += nvec[g] * (ge.H / 4) * ((ge.W[g] + 31) / 32);
            a16 += nvec[g] * (ge.H / 8) * ((ge.W[g] + 15) / 16);
            agap += (int32_t)((ge.H / 8) * (((int64_t)nvec[g] * ((ge.W[g] + 2) & ~1) + 15) / 16));
            tgap.push_back(agap);
            for (int e = 0; e < 2; e++) {
                const int64_t cols = (int64_t)nvec[g] * (e ? (ge.W[g] + 1) & ~1 : ge.W[g]);
                af32[e] += (int32_t)((ge.H / 4) * ((cols + 31) / 32));
                af16[e] += (int32_t)((ge.H / 8) * ((cols + 15) / 16));
                f32[e].push_back(af32[e]);
                f16[e].push_back(af16[e]);
            }
            px += rows;
            meta32.push_back(a128);
            t256.push_back(a256);
            t32.push_back(a32);
            t16.push_back(a16);
            pv.push_back(px);
        }
        o.t256 = meta32.size();
        meta32.insert(meta32.end(), t256.begin(), t256.end());
        o.t32 = meta32.size();
        meta32.insert(meta32.end(), t32.begin(), t32.end());
        o.t16 = meta32.size();
        meta32.insert(meta32.end(), t16.begin(), t16.end());
        o.nt32 = a32; o.nt16 = a16;
        o.gap = meta32.size();
        meta32.insert(meta32.end(), tgap.begin(), tgap.end());
        o.ngap = agap;
        for (int e = 0; e < 2; e++) {
            o.f32[e] = meta32.size();
            meta32.insert(meta32.end(), f32[e].begin(), f32[e].end());
            o.f16[e] = meta32.size();
            meta32.insert(meta32.end(), f16[e].begin(), f16[e].end());
            o.nf32[e] = af32[e]; o.nf16[e] = af16[e];
        }
        o.p = meta64.size();
        meta64.insert(meta64.end(), pv.begin(), pv.end());
        o.nt128 = a128; o.nt256 = a256; o.pixels = px;
        goff.push_back(o);
    }
    int64_t* d64 = ws.alloc_n<int64_t>(meta64.size());
    int32_t* d32 = ws.alloc_n<int32_t>(meta32.size());
    ws.upload(d64, meta64.data(), meta64.size() * sizeof(int64_t));
    ws.upload(d32, meta32.data(), meta32.size() * sizeof(int32_t));
    if (before_launch) before_launch();   // everything above is host work and uploads on the request's own stream
    if (exec != ws.s()) {  // inputs (crops, plan, metadata) were produced on the request's stream
        hipEvent_t ready = ws.make_event();
        OCRS_HIP(hipEventRecord(ready, ws.s()));
        OCRS_HIP(hipStreamWaitEvent(exec, ready, 0));
    }
    auto view = [&](size_t gi) {
        k::RaggedView v{};
        v.G = G; v.H = geos[gi].H;
        v.W = d32 + goff[gi].w; v.n = d32 + n_at; v.poff = d64 + goff[gi].p;
        v.toff128 = d32 + goff[gi].t128; v.toff256 = d32 + goff[gi].t256; v.loff = d32 + loff_at;
        v.ntiles128 = goff[gi].nt128; v.ntiles256 = goff[gi].nt256; v.pixels = goff[gi].pixels;
        v.max_w = 0;
        for (int32_t w : geos[gi].W) v.max_w = std::max(v.max_w, (int)w);
        // patch shape for the 3x3 convs: the one that wastes fewer lanes on ragged right edges
        const bool use16 = geos[gi].H % 8 == 0 && (geos[gi].H % 4 != 0 || goff[gi].nt16 < goff[gi].nt32);
        v.tw = use16 ? 16 : 32;
        v.toff2d = d32 + (use16 ? goff[gi].t16 : goff[gi].t32);
        v.ntiles2d = use16 ? goff[gi].nt16 : goff[gi].nt32;
        v.toff2d_flat = d32 + (use16 ? goff[gi].f16[0] : goff[gi].f32[0]);
        v.toff2d_flat2 = d32 + (use16 ? goff[gi].f16[1] : goff[gi].f32[1]);
        v.ntiles2d_flat = use16 ? goff[gi].nf16[0] : goff[gi].nf32[0];
        v.ntiles2d_flat2 = use16 ? goff[gi].nf16[1] : goff[gi].nf32[1];
        v.toff2d_gap = d32 + goff[gi].gap;
        v.ntiles2d_gap = geos[gi].H % 8 == 0 ? goff[gi].ngap : 0;
        v.max_tile_px_ = 0;
        v.min_w = INT32_MAX;
        for (int g = 0; g < G; g++) {
            const int64_t w = geos[gi].W[g], span = std::min<int64_t>(nvec[g], (v.tw - 1 + w - 1) / w + 1);
            v.max_tile_px_ = std::max(v.max_tile_px_, span * geos[gi].H * w);
            v.min_w = std::min(v.min_w, (int)w);
        }
        return v;
    };

    int tok = timers ? timers->begin(ST_REC_CONV, st, 0) : -1;
    const float* cur = groups[0].d_batch;
    int curC = 1;
    float* prev_buf = nullptr;  // buffer holding `cur` (nullptr: the caller's input)
    std::vector<std::pair<float*, size_t>> spare;  // released buffers, reused by later layers
    // On the device's shared conv-stack stream the intermediate activations come from the DEVICE's arena instead of the
    // request's workspace: they are touched only by kernels of that one stream, which run in the order they were enqueued
    // (under heavy_phase, held here since before_launch), so every request's stack can use the same few buffers — with
    // six 16-page requests in flight that is ~5 GB once instead of ~5 GB per request.  Buffers are never handed back while
    // the process runs (an earlier request's kernels may still be using them); the arena grows to the largest request seen.
    // (a request of a device that runs one kernel at a time — StreamLease::serial() — has exec == its own stream: same rule)
    const bool shared_arena = exec != ws.s() || (ws.stream.serial() && before_launch);
    std::vector<char> arena_taken;
    auto get = [&](size_t floats) -> float* {
        const size_t bytes = floats * sizeof(float);
        for (size_t i = 0; i < spare.size(); i++)
            if (spare[i].second >= bytes) { float* p = spare[i].first; spare.erase(spare.begin() + i); return p; }
        if (!shared_arena) return static_cast<float*>(ws.alloc(bytes));
        auto& arena = ctx().heavy_arena;
        arena_taken.resize(arena.size(), 0);
        size_t best = arena.size();
        for (size_t i = 0; i < arena.size(); i++)
            if (!arena_taken[i] && arena[i].bytes >= bytes && (best == arena.size() || arena[i].bytes < arena[best].bytes)) best = i;
        if (best == arena.size()) {
            arena.emplace_back(bytes + bytes / 8);   // a little head-room: requests of a stream differ by a few lines
            arena_taken.push_back(0);
        }
        arena_taken[best] = 1;
        return arena[best].as<float>();
    };
    size_t prev_bytes = 0;
    for (int i = 0; i < ts; i++) {
        const GraphOp& op = ops[i];
        const k::RaggedView vin = view(i);
        float* y = nullptr;
        size_t ybytes = 0;
        // conv1 + pool + conv2 + pool in one kernel where the shapes allow (kernels_rec.hip: conv12_fused_ragged)
        if (op.type == OP_CONV && op.cin == 1 && i + 3 < ts && ops[i + 2].type == OP_CONV && ops[i + 2].cin == op.cout &&
            ops[i + 2].relu && ops[i + 3].type == OP_MAXPOOL && ops[i + 3].kh == 2 && ops[i + 3].kw == 2) {
            const GraphOp& op2 = ops[i + 2];
            const k::RaggedView vmid = view(i + 2), vout = view(i + 4);
            bool ok = k::conv12_fused_ragged(nullptr, vin, vmid, nullptr, nullptr, op.cout, nullptr, nullptr, op2.cout, nullptr, vout, st);
            float* y2 = ok ? get((size_t)vout.pixels * op2.cout) : nullptr;
            if (ok) {
                // counted as conv2's launch with conv2's FLOPs (the matrix-core work); conv1's VALU work rides along
                timed(KC_GEMM_CONV3X3, 2.0 * vmid.pixels * 9.0 * op2.cin * op2.cout,
                      4.0 * (vin.pixels + vout.pixels * op2.cout) + 4.0 * op2.wcount[0],
                      [&] { k::conv12_fused_ragged(cur, vin, vmid, op.w[0], op.w[1], op.cout, op2.w[0], op2.w[1], op2.cout, y2, vout, st, op2.wsplit); });
                y = y2;
                ybytes = (size_t)vout.pixels * op2.cout * sizeof(float);
                curC = op2.cout;
                i += 3;
                if (prev_buf) spare.emplace_back(prev_buf, prev_bytes);
                prev_buf = y;
                prev_bytes = ybytes;
                cur = y;
                continue;
            }
        }
        if (op.type == OP_CONV && op.cin == 1) {
            const k::RaggedView vout = view(i + 2);  // after the fused MaxPool 2x2
            ybytes = (size_t)vout.pixels * op.cout * sizeof(float);
            y = get((size_t)vout.pixels * op.cout);
            timed(KC_CONV_DIRECT, 2.0 * vin.pixels * 9 * op.cout, 4.0 * vin.pixels + 4.0 * vout.pixels * op.cout,
                  [&] { k::conv1_relu_pool_ragged(cur, vin, op.w[0], op.w[1], op.cout, y, vout, st); });
            curC = op.cout;
            i += 1;  // the pool is fused
        } else if (op.type == OP_CONV) {
            // a MaxPool 2x1 / 2x2 that consumes this conv is folded into its epilogue
            const bool fuse = i + 1 < ts && ops[i + 1].type == OP_MAXPOOL && ops[i + 1].kh == 2 &&
                              (ops[i + 1].kw == 1 || ops[i + 1].kw == 2) && vin.H % 2 == 0;
            const int ph = fuse ? 2 : 1, pw = fuse ? ops[i + 1].kw : 1;
            const k::RaggedView vout = view(fuse ? i + 2 : i + 1);
            ybytes = (size_t)vout.pixels * op.cout * sizeof(float);
            y = get((size_t)vout.pixels * op.cout);
            bool ok = false;
            timed(KC_GEMM_CONV3X3, 2.0 * vin.pixels * 9.0 * op.cin * op.cout,
                  4.0 * (vin.pixels * op.cin + vout.pixels * op.cout) + 4.0 * op.wcount[0],
                  [&] { ok = k::conv3x3_ragged(cur, vin, op.cin, op.w[0], op.w[1], op.cout, op.relu, ph, pw, y, vout, st, op.wsplit); });
            if (!ok) fail(OCRS_ERR_RUN_FAILED, "model run failed: ragged conv %d->%d at height %d not supported", op.cin, op.cout, vin.H);
            curC = op.cout;
            if (fuse) i += 1;
        } else if (op.type == OP_AVGPOOL && i + 1 == ts && op.kw == 1 && op.kh == vin.H && (curC & 3) == 0) {
            break;   // the column average down to height 1 that ends the stack: done together with the sequence packing below
        } else {
            const k::RaggedView vout = view(i + 1);
            ybytes = (size_t)vout.pixels * curC * sizeof(float);
            y = get((size_t)vout.pixels * curC);
            timed(KC_POOL, 0, 4.0 * curC * (vin.pixels + vout.pixels),
                  [&] { k::pool_ragged(cur, vin, curC, op.kh, op.kw, op.type == OP_AVGPOOL, y, vout, st); });
        }
        if (prev_buf) spare.emplace_back(prev_buf, prev_bytes);
        prev_buf = y;
        prev_bytes = ybytes;
        cur = y;
    }
    const k::RaggedView vf = view(ts);
    if (vf.H != 1) fail(OCRS_ERR_RUN_FAILED, "model run failed: TOSEQ expects height 1, got %d", vf.H);
    float* X = ws.alloc_n<float>((size_t)plan.R * curC);
    const bool pool_here = ts >= 1 && ops[ts - 1].type == OP_AVGPOOL && ops[ts - 1].kw == 1 && (curC & 3) == 0 &&
                           ops[ts - 1].kh == view(ts - 1).H;
    if (pool_here) {
        const k::RaggedView vp = view(ts - 1);
        timed(KC_POOL, 0, 4.0 * curC * (vp.pixels + vf.pixels),
              [&] { k::avgpool_to_seq_ragged(cur, vp, vf, curC, groups[0].d_pos, plan.d_off, X, st); });
    } else {
        timed(KC_OTHER, 0, 8.0 * vf.pixels * curC,
              [&] { k::to_seq_packed_ragged(cur, vf, curC, groups[0].d_pos, plan.d_off, X, st); });
    }
    if (tok >= 0) timers->end(tok, st);
    if (exec != ws.s()) {  // the request's stream continues once the conv stack has drained
        hipEvent_t done = ws.make_event();
        OCRS_HIP(hipEventRecord(done, exec));
        OCRS_HIP(hipStreamWaitEvent(ws.s(), done, 0));
    }
    *feat_c = curC;
    return X;
}

int HipModel::packed_split() const {
    int ts = -1;
    for (size_t i = 0; i < ops.size(); i++)
        if (ops[i].type == OP_TOSEQ) { ts = (int)i; break; }
    if (ts < 0 || (size_t)ts + 3 > ops.size()) return -1;
    int prev_out = ops[ts].out;
    for (size_t i = ts + 1; i < ops.size(); i++) {
        const GraphOp& op = ops[i];
        const bool last = i + 1 == ops.size(), second_last = i + 2 == ops.size();
        if (op.in0 != prev_out) return -1;
        if (last) { if (op.type != OP_LOGSOFTMAX || (uint32_t)op.out != out_slot) return -1; }
        else if (second_last) { if (op.type != OP_LINEAR) return -1; }
        else if (op.type != OP_GRU) return -1;
        prev_out = op.out;
    }
    for (int i = 0; i < ts; i++)
        if (ops[i].type == OP_GRU || ops[i].type == OP_LINEAR || ops[i].type == OP_LOGSOFTMAX || ops[i].type == OP_TOSEQ)
            return -1;
    return ts;
}

int HipModel::run_recognition_packed(Workspace& ws, const std::vector<PackedGroup>& groups, const PackedPlan& plan, int h,
                                     StageTimers* timers, const uint8_t* d_excluded, int32_t* d_labels,
                                     float** d_logp) const {
    const int ts = packed_split();
    if (ts < 0) fail(OCRS_ERR_RUN_FAILED, "model run failed: graph is not <conv stack> TOSEQ GRU* LINEAR LOGSOFTMAX");
    hipStream_t st = ws.s();
    auto timed = [&](int cls, double flops, double bytes, auto&& launch) {
        int tok = timers ? timers->kbegin(cls, st, flops, bytes) : -1;
        launch();
        if (tok >= 0) timers->end(tok, st);
    };
    const int64_t R = plan.R;
    const int M = plan.M;

    // ---- conv stack -> packed feature rows: one launch per layer over all groups (ragged),
    // or group by group if the stack has an op the ragged kernels do not cover.
    // The conv stack saturates the GPU; the recurrence that follows is a chain of small,
    // latency-bound launches.  When several requests are in flight (host threads / streams),
    // two conv stacks at once gain nothing, but a conv stack next to other requests' GRU chains
    // does: so every request's conv stack is enqueued on one shared stream (FIFO on the GPU, linked
    // to the request's own stream by events) and everything after it overlaps freely.
    float* X = nullptr;
    int C0 = 0;
    {
        // all conv stacks of a device go through ONE stream, in request order.  The lock is taken by the hook, i.e. after
        // the request's geometry has been worked out and its metadata uploads are queued.
        std::unique_lock<std::mutex> heavy(ctx().heavy_phase, std::defer_lock);
        const hipStream_t conv = ws.stream.conv_stream();   // the device's conv-stack stream
        try {
            X = run_prefix_ragged(ws, conv, groups, plan, h, ts, timers, &C0, [&] { heavy.lock(); });
        } catch (...) {
            // Kernels of this request may already be queued on the shared stream, reading and writing scratch
            // that ~Workspace hands back to the pool after draining only the request's OWN stream: make that
            // stream wait for them first (best effort, no throw on the error path).
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess) {
                ws.events.push_back(e);
                if (hipEventRecord(e, conv) != hipSuccess || hipStreamWaitEvent(ws.s(), e, 0) != hipSuccess)
                    (void)hipStreamSynchronize(conv);
            } else {
                (void)hipStreamSynchronize(conv);
            }
            throw;
        }
    }
    if (!X)
    for (const PackedGroup& g : groups) {
        TensorShape fs;
        float* feat = run_device(ws, g.d_batch, g.n, h, g.w, &fs, timers, nullptr, nullptr, true, false, ts);
        if (fs.h != 1) fail(OCRS_ERR_RUN_FAILED, "model run failed: TOSEQ expects height 1, got %d", fs.h);
        if (!X) {
            C0 = fs.c;
            X = ws.alloc_n<float>((size_t)R * C0);
        }
        int tok = timers ? timers->begin(ST_REC_CONV, st, 0) : -1;
        timed(KC_OTHER, 0, 8.0 * fs.count(), [&] { k::to_seq_packed(feat, g.n, fs.w, fs.c, g.d_pos, plan.d_off, X, st); });
        if (tok >= 0) timers->end(tok, st);
    }
    if (!X) return 0;

    const float* cur = X;
    int curC = C0;
    int classes = 0;
    int gru_layer = 0;
    float* gx_buf = nullptr;
    size_t gx_cap = 0;
    void* hx_buf = nullptr;   // the recurrences' hand-off buffer: likewise one for all layers (a layer's marks are written after the
    size_t hx_cap = 0;        // previous layer's recurrence has finished: this stream waits for it)
    for (size_t i = ts + 1; i < ops.size(); i++) {
        const GraphOp& op = ops[i];
        if (op.type == OP_GRU) {
            const int H = op.hidden, I = op.cin;
            if (I != curC) fail(OCRS_ERR_RUN_FAILED, "model run failed: GRU input size %d != %d", I, curC);
            int tok = timers ? timers->begin(ST_REC_GRU, st, 0) : -1;
            // the input projections of a layer are dead once its recurrence has run, and the next layer's projection is
            // ordered after that recurrence (it reads its output): one buffer serves every layer (2.3 GB per layer at 16 pages)
            const size_t gx_floats = (size_t)2 * R * 3 * H;
            if (gx_floats > gx_cap) { gx_buf = ws.alloc_n<float>(gx_floats); gx_cap = gx_floats; }
            float* gx = gx_buf;
            float* y = ws.alloc_n<float>((size_t)R * 2 * H);
            const bool fused = (H == 256 || H == 128 || H == 64);
            const bool persistent = fused && gru_mode() == GRU_PERSISTENT && k::gru_persistent_supported(plan.h_Tm.data(), M, plan.Tmax, R, H);
            // numerics != exact: the recurrence on the bf16 matrix cores (kernels_gru_split.hip), state cut into 3 / 2 planes
            const int np = option(OPT_NUMERICS) == 1 ? 3 : option(OPT_NUMERICS) == 2 ? 2 : 0;
            const bool split = persistent && np != 0 && k::gru_split_supported(plan.h_Tm.data(), M, plan.Tmax, R, H, np);
            // the hand-off buffer of the launch, every word marked "unwritten" ahead of the input GEMM
            const size_t hx_bytes = split ? k::gru_split_exchange_bytes(plan.h_Tm.data(), M, H, np) : persistent ? k::gru_persistent_exchange_bytes(plan.h_Tm.data(), M, H) : 0;
            if (hx_bytes > hx_cap) { hx_buf = ws.alloc(hx_bytes); hx_cap = hx_bytes; }
            uint16_t* hx = split ? static_cast<uint16_t*>(hx_buf) : nullptr;
            float* hxf = persistent && !split ? static_cast<float*>(hx_buf) : nullptr;
            if (split) OCRS_HIP(k::gru_split_prepare(hx, plan.h_Tm.data(), M, H, np, st));
            else if (persistent) OCRS_HIP(k::gru_persistent_prepare(hxf, plan.h_Tm.data(), M, H, st));
            k::GemmDesc d{};
            d.A = cur; d.lda = I; d.B = op.aux0; d.ldb = 3 * H; d.bias = op.aux1; d.C = gx; d.ldc = 3 * H;
            d.M = (int)R; d.N = 3 * H; d.K = I; d.batch = 2;
            d.strideA = 0; d.strideB = (int64_t)I * 3 * H; d.strideBias = 3 * H; d.strideC = R * 3 * H;
            d.Bsplit = op.wsplit; d.strideBsplit = (int64_t)(3 * H / 128) * (I / 16) * 3 * 128 * 16;   // (uint16 units per direction)
            const double gx_flops = 2.0 * 2 * R * (double)d.N * d.K, gx_bytes = 4.0 * ((double)R * I + 2.0 * R * d.N + 2.0 * d.K * d.N);
            // (round 3's option gx_heavy queued these projections on the conv-stack stream: every MFMA class then ran at its
            // alone speed at the same or slightly lower pages/s — a zero-sum trade, removed in round 5)
            timed(KC_GEMM_GRU_INPUT, gx_flops, gx_bytes, [&] { k::gemm(d, st); });
            bool ran_persistent = false;
            if (persistent) {
                // ONE launch for all Tmax steps of both directions (kernels_gru.hip).  Its workgroups wait on
                // each other, so two such kernels must never be half-resident at the same time: every request's
                // recurrences go through one stream per device (FIFO on the GPU, linked by events, no host wait).
                uint32_t* d_sync = ws.alloc_n<uint32_t>(k::gru_persistent_sync_words(M));
                double fl = 0.0;
                for (int step = 0; step < plan.Tmax; step++) fl += 2.0 * 2 * plan.active[step] * 3.0 * H * H;
                {
                    DeviceContext& dc = ctx();
                    std::lock_guard<std::mutex> g(dc.rec_phase);
                    hipStream_t rs = ws.stream.recurrent_stream();
                    hipEvent_t ready = ws.make_event(), done = ws.make_event();
                    OCRS_HIP(hipEventRecord(ready, st));
                    OCRS_HIP(hipStreamWaitEvent(rs, ready, 0));
                    int ktok = timers ? timers->kbegin(KC_GEMM_GRU_HIDDEN, rs, fl, 4.0 * ((double)R * (2.0 * 3 * H + 2.0 * 2 * H) + 2.0 * 3 * H * H)) : -1;
                    ran_persistent = split ? k::gru_persistent_split(gx, op.aux2, op.aux3, y, hx, plan.d_Tm, plan.d_off, plan.h_Tm.data(), R, M, plan.Tmax, H, np, d_sync, rs)
                                           : k::gru_persistent(gx, op.aux2, op.aux3, y, hxf, plan.d_Tm, plan.d_off, plan.h_Tm.data(), R, M, plan.Tmax, H, d_sync, rs);
                    if (ktok >= 0) timers->end(ktok, rs);
                    if (ran_persistent && plan.h_status && gru_layer < 8)
                        ws.download(plan.h_status + gru_layer, d_sync + k::gru_persistent_sync_words(M) - 1, sizeof(uint32_t), rs);
                    OCRS_HIP(hipEventRecord(done, rs));
                    OCRS_HIP(hipStreamWaitEvent(st, done, 0));
                }
            }
            if (ran_persistent) {
                // done
            } else if (fused) {
                // One launch per time step (option gru_mode = 1, and shapes the persistent kernel has no plan for):
                // hidden GEMM + gates in one kernel, transposed ping-ponged state hT[2 dirs][H][Mcap].
                const int Mcap = (M + 3) & ~3;
                float* hT0 = ws.alloc_n<float>((size_t)2 * H * Mcap);
                float* hT1 = ws.alloc_n<float>((size_t)2 * H * Mcap);
                OCRS_HIP(hipMemsetAsync(hT0, 0, (size_t)2 * H * Mcap * sizeof(float), st));
                for (int step = 0; step < plan.Tmax; step++) {
                    const int act = plan.active[step];
                    if (act <= 0) break;
                    const float* hin = (step & 1) ? hT1 : hT0;
                    float* hout = (step & 1) ? hT0 : hT1;
                    timed(KC_GEMM_GRU_HIDDEN, 2.0 * 2 * act * 3.0 * H * H,
                          4.0 * 2 * ((double)act * H * 2 + (double)act * 3 * H + 3.0 * H * H + (double)act * H), [&] {
                              if (!k::gru_step_fused(gx, op.aux2, op.aux3, hin, hout, y, plan.d_Tm, plan.d_off, R, Mcap, act, H, step, st))
                                  fail(OCRS_ERR_RUN_FAILED, "model run failed: no fused GRU step kernel for hidden size %d", H);
                          });
                }
            } else {
                // any other hidden size: hidden GEMM and gate kernel as two launches per time step
                float* gh = ws.alloc_n<float>((size_t)2 * M * 3 * H);
                float* hs = ws.alloc_n<float>((size_t)2 * M * H);
                OCRS_HIP(hipMemsetAsync(hs, 0, (size_t)2 * M * H * sizeof(float), st));
                k::GemmDesc r{};
                r.A = hs; r.lda = H; r.B = op.aux2; r.ldb = 3 * H; r.bias = op.aux3; r.C = gh; r.ldc = 3 * H;
                r.N = 3 * H; r.K = H; r.batch = 2;
                r.strideA = (int64_t)M * H; r.strideB = (int64_t)H * 3 * H; r.strideBias = 3 * H; r.strideC = (int64_t)M * 3 * H;
                for (int step = 0; step < plan.Tmax; step++) {
                    const int act = plan.active[step];
                    if (act <= 0) break;
                    r.M = act;
                    timed(KC_GEMM_GRU_HIDDEN, 2.0 * 2 * act * (double)r.N * r.K,
                          4.0 * 2 * ((double)act * r.K + (double)act * r.N + (double)r.K * r.N), [&] { k::gemm(r, st); });
                    timed(KC_GRU_GATES, 0, 4.0 * 2 * act * (double)H * 9, [&] {
                        k::gru_gates_packed(gx, gh, hs, y, plan.d_Tm, plan.d_off, R, M, act, H, step, st);
                    });
                }
            }
            if (tok >= 0) timers->end(tok, st);
            gru_layer++;
            cur = y;
            curC = 2 * H;
        } else if (op.type == OP_LINEAR) {
            if (op.cin != curC) fail(OCRS_ERR_RUN_FAILED, "model run failed: Linear input size %d != %d", op.cin, curC);
            int tok = timers ? timers->begin(ST_REC_HEAD, st, 0) : -1;
            float* y = ws.alloc_n<float>((size_t)R * op.cout);
            k::GemmDesc d{};
            d.A = cur; d.lda = op.cin; d.B = op.w[0]; d.ldb = op.cout; d.bias = op.w[1];
            d.C = y; d.ldc = op.cout; d.M = (int)R; d.N = op.cout; d.K = op.cin; d.relu = op.relu;
            timed(KC_GEMM_LINEAR, 2.0 * R * (double)d.N * d.K, 4.0 * ((double)R * (op.cin + op.cout)) + 4.0 * op.wcount[0],
                  [&] { k::gemm(d, st); });
            if (tok >= 0) timers->end(tok, st);
            cur = y;
            curC = op.cout;
        } else {  // LOGSOFTMAX (+ arg-max)
            int tok = timers ? timers->begin(ST_REC_HEAD, st, 0) : -1;
            float* lp = nullptr;
            if (d_logp) {
                lp = ws.alloc_n<float>((size_t)R * curC);
                *d_logp = lp;
            }
            timed(KC_LOGSOFTMAX_ARGMAX, 0, 4.0 * R * curC * (lp ? 2.0 : 1.0),
                  [&] {
                      if (!k::log_softmax_argmax(cur, R, curC, d_excluded, lp, d_labels, st))
                          fail(OCRS_ERR_CAPACITY, "LogSoftmax over %d classes exceeds the kernel's LDS staging (max ~630)", curC);
                  });
            if (tok >= 0) timers->end(tok, st);
            classes = curC;
        }
    }
    OCRS_HIP(hipGetLastError());
    return classes;
}

}  // namespace ocrs
