// Fixed-graph HIP executor: the MI355X replacement for `rten::Model`
// behind `trait Model` (ocrs/src/model.rs:6-41).
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"

namespace ocrs {

enum OpType : uint32_t {
    OP_CONV = 0, OP_DWCONV3, OP_MAXPOOL, OP_AVGPOOL, OP_CONVT2, OP_PADCAT, OP_SIGMOID, OP_TOSEQ, OP_GRU, OP_LINEAR,
    OP_LOGSOFTMAX, OP_COUNT
};

struct GraphOp {
    uint32_t type;
    int32_t in0, in1, out;
    int32_t relu, kh, kw, cin, cout, hidden;
    // device weights (owned by the model's weight slab)
    const float* w[8] = {nullptr};
    size_t wcount[8] = {0};
    // derived device tensors built at load time
    const float* aux0 = nullptr;  // convT: B re-laid out as [Cin][4*Cout]; GRU: Wi of both dirs [2][I][3H]
    const float* aux1 = nullptr;  // convT: bias x4; GRU: bi [2][3H]
    const float* aux2 = nullptr;  // GRU: Wh [2][H][3H]
    const float* aux3 = nullptr;  // GRU: bh [2][3H]
    const uint16_t* wsplit = nullptr;   // 3x3 conv / GRU input weights cut into bf16 terms for the relaxed-numerics kernels (or null)
    bool fused_into_prev = false; // e.g. SIGMOID folded into the preceding Cout==1 conv
    bool fuse_next_pw = false;    // DWCONV3 whose only consumer is the next op, a 1x1 CONV (8 <= C <= 32): one fused launch
    bool done_by_prev = false;    // that 1x1 CONV
    bool cat_into_next = false;   // PADCAT whose only reader is the next op, a DWCONV3: the concatenation is never built
    bool reads_cat = false;       // that DWCONV3 (reads the two PADCAT inputs directly)
    int dc_block = -1;            // index into HipModel::dc_blocks if a fused DoubleConv block starts at this op
};

// A DoubleConv block of the detection U-Net that one fused launch covers (kernels_det.hip); op indices, -1 = absent.
struct DcBlock {
    int first = -1, last = -1;
    int convt = -1, cat = -1, dw1 = -1, pw1 = -1, dw2 = -1, pw2 = -1, pool = -1, fin = -1, sig = -1;
    int cs = 0, cx = 0, cmid = 0, cout = 0;
    const float* tape = nullptr;  // row-streaming wave kernel (kernels_det_stream.hip): the block's weight tape on the device
    int tape_len = 0;
    const float* rtape = nullptr; // row-streaming workgroup kernel (kernels_det_rows.hip): its four per-wave tapes
    int rtape_len = 0;
};

struct TensorShape {  // NHWC activations, or [T,N,C] sequences (n=T, h=N, w=1, c=C)
    int n = 0, h = 0, w = 0, c = 0;
    bool seq = false;
    int64_t count() const { return (int64_t)n * h * w * c; }
};

// What a caller-implemented model looks like to the engine (`trait Model`).
struct ModelBase {
    virtual ~ModelBase() = default;
    int device = -1;                            // HIP device that holds the weights (-1: none, a callback model)
    int64_t input_shape[4] = {-1, -1, -1, -1};  // NCHW, -1 = symbolic
    virtual bool is_callback() const = 0;
};

struct CallbackModel : ModelBase {
    ocrs_model_run_fn fn = nullptr;
    void* user = nullptr;
    bool is_callback() const override { return true; }
    // host NCHW in -> host out (malloc'ed by callee, adopted here)
    void run(const float* input, const int64_t in_shape[4], std::vector<float>& out, int64_t out_shape[4],
             int* out_ndim) const;
};

struct HipModel : ModelBase {
    uint32_t kind = 0;  // 0 detection, 1 recognition
    std::vector<GraphOp> ops;
    std::vector<DcBlock> dc_blocks;
    uint32_t n_slots = 0, out_slot = 0;
    DevBuf weights;      // one slab: file blob + derived tensors
    std::vector<DevBuf> tapes;   // weight tapes of the row-streaming DoubleConv blocks (DcBlock::tape)
    bool is_callback() const override { return false; }

    // device < 0: the process default (ocrs_set_device)
    static std::unique_ptr<HipModel> load(const void* data, size_t len, int device = -1);

    // Shape inference for an input of n x h x w (C = 1); returns the output shape.
    TensorShape infer(int n, int h, int w, std::vector<TensorShape>* slots = nullptr) const;
    double flops(int n, int h, int w) const;

    // Run on device.  d_in: [n,h,w,1] fp32.  Returns the output tensor, allocated from
    // `ws` (detection: [n,H,W,1] probabilities; recognition: [T,n,C] log-probs or, when
    // want_logp is false, nothing but the arg-max labels).  Recognition extras:
    //   d_excluded: [C] bytes or null; d_labels: [T*n] arg-max labels or null.
    // stop_before >= 0: execute only ops [0, stop_before) and return that op's input tensor.
    float* run_device(Workspace& ws, const float* d_in, int n, int h, int w, TensorShape* out_shape,
                      StageTimers* timers, const uint8_t* d_excluded = nullptr, int32_t* d_labels = nullptr,
                      bool want_logp = true, bool print_timing = false, int stop_before = -1,
                      hipStream_t exec = nullptr /* launch here instead of on ws's stream (the caller links the two) */) const;

    // ---- ragged recognition batch (all width groups of a request at once) ----
    // The graph must be: <conv stack> TOSEQ GRU* LINEAR LOGSOFTMAX.  Returns the index
    // of the TOSEQ op, or -1 if the graph has another shape.
    int packed_split() const;
    struct PackedGroup {      // one width group: batch [n, h, w] and, per line, its row slot m
        const float* d_batch;  // groups are contiguous in memory, in this order (ragged buffer)
        int n, w;
        const int32_t* d_pos;  // [n] device; the pos arrays of consecutive groups are contiguous too
    };
    struct PackedPlan {       // lines sorted by T descending; rows off[t] + m
        int M = 0, Tmax = 0;
        int64_t R = 0;
        const int32_t* d_Tm = nullptr;   // [M]
        const int32_t* d_off = nullptr;  // [Tmax + 1]
        std::vector<int> active;         // [Tmax] host
        std::vector<int32_t> h_Tm;       // [M] host copy of d_Tm
        uint32_t* h_status = nullptr;    // host [8]: slot i receives the time-out word of the i-th GRU layer's
                                         // persistent kernel after the workspace's next sync (0 = fine)
    };
    // Writes arg-max labels of every packed row to d_labels [R]; returns class count.
    // d_logp (optional): receives the packed log-probabilities [R][classes] (model output, unmasked).
    int run_recognition_packed(Workspace& ws, const std::vector<PackedGroup>& groups, const PackedPlan& plan, int h,
                               StageTimers* timers, const uint8_t* d_excluded, int32_t* d_labels,
                               float** d_logp = nullptr) const;
    // Conv stack (ops [0, ts)) over all groups at once; writes packed feature rows.  Returns
    // nullptr if the stack has an op the ragged kernels do not cover.
    // Kernels are launched on `exec` (which may differ from ws.s(); the caller links the two with events).
    // before_launch: called once after the host-side planning and the metadata uploads, before the first launch on `exec`
    // (the caller takes the shared stream's lock there).
    float* run_prefix_ragged(Workspace& ws, hipStream_t exec, const std::vector<PackedGroup>& groups,
                             const PackedPlan& plan, int h, int ts, StageTimers* timers, int* feat_c,
                             const std::function<void()>& before_launch = nullptr) const;
};

}  // namespace ocrs

// The opaque C handle.
struct ocrs_model {
    std::unique_ptr<ocrs::ModelBase> impl;
};
