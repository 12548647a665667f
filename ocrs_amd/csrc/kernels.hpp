// Launch wrappers for every HIP kernel in libocrs_amd (gfx950).  All take the
// stream to launch on; none synchronises.  Activations are NHWC fp32.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ocrs {
namespace jpeg { struct Coefficients; }
namespace k {

// ---- kernels_image.hip ----------------------------------------------------
// prepare_image (preprocess.rs:149-248): pixels -> grey f32 [H,W] in [-0.5,0.5].
void prepare_image(const void* d_pixels, bool is_u8, bool chans_last, int h, int w, int chans, float* d_out,
                   hipStream_t s);
// pad(-0.5) + bilinear resize of N equally sized pages to the model input
// (detection.rs:155-171).  src_ptrs: device array of N page pointers [sh,sw].
void resize_pages_to_model(const float* const* d_src_ptrs, int n, int sh, int sw, int vh, int vw, float* d_dst,
                           int dh, int dw, hipStream_t s);
// slice + bilinear resize back + strict threshold (detection.rs:187-194,110).
// prob: [n, mh, mw] (only the top-left [sh, sw] is read); mask: [n, h, w] u8;
// d_map (optional, may be null): [n, h, w] f32 probability map.
// d_labels / d_zero_a / d_zero_b (optional, r6): the component stage's label array [n, h, w] and its two per-page counters
// (CclBuffers::overflow, ::offsets).  When all three are given and the shapes allow (w % 4 == 0, aligned buffers, option
// ccl_quad) the launch also writes the initial labels and zeroes the counters, and returns true: ccl_label and contour_rects
// are then told to skip their first step (`prepared`).
bool resize_threshold(const float* d_prob, int n, int mh, int mw, int sh, int sw, float thr, uint8_t* d_mask,
                      float* d_map, int h, int w, hipStream_t s, int32_t* d_labels = nullptr, int32_t* d_zero_a = nullptr,
                      int32_t* d_zero_b = nullptr);
void threshold_only(const float* d_prob, float thr, uint8_t* d_mask, int64_t count, hipStream_t s);

// ---- kernels_ccl.hip ------------------------------------------------------
struct CclBuffers {
    int32_t* labels;      // [n, h, w]
    int32_t* row_counts;  // [n, h]
    int32_t* row_offsets; // [n, h]
    int32_t* n_roots;     // [n]  external components per page
    int32_t* roots;       // [n, max_comp]  raster-ordered start pixels
    int32_t* lengths;     // [n, max_comp]  (unused since r3: lengths are found by the contour kernel itself)
    int32_t* offsets;     // [n]            per-page bump counter of the contour arena
    int32_t* overflow;    // [n]  set if a capacity was exceeded
    uint32_t* pts;        // [n, arena]  packed (y << 16 | x) contour points
    uint32_t* tmp;        // [n, 4 * arena]  per component: simplified | sorted | hull (2x) scratch
    uint8_t* keep;        // [n, arena]
    float* rects;         // [n, max_comp, 6]
    uint8_t* valid;       // [n, max_comp]
};
// mask [n,h,w] -> per page: raster-ordered external components -> rects
// (find_contours(External) -> simplify_polygon(2) -> min_area_rect -> resize(+2*expand)
//  -> area >= min_area; detection.rs:41-62).
// prepared: resize_threshold already wrote the initial labels / zeroed the counters (it returned true)
void ccl_label(const uint8_t* d_mask, int n, int h, int w, const CclBuffers& b, int max_comp, hipStream_t s, bool prepared = false);
void contour_rects(const uint8_t* d_mask, int n, int h, int w, const CclBuffers& b, int max_comp, int64_t arena,
                   float expand, float min_area, float eps, hipStream_t s, bool prepared = false);

// ---- kernels_nn.hip -------------------------------------------------------
// C[M,N] = act(A[M,K] . B[K,N] + bias[N]) as an exact fp32 MFMA chain, k ascending.
struct GemmDesc {
    const float* A; int lda;      // dense: row-major [M][lda]
    const float* B; int ldb;      // row-major [K][ldb]
    const float* bias;            // [N] (may be null -> 0)
    float* C; int ldc;
    int M, N, K;
    int relu;
    // im2col (3x3, pad 1) view of an NHWC tensor [n, H, W, Cin]: M = n*H*W, K = 9*Cin
    int im2col; int H, W, Cin;
    // ConvTranspose2x2/s2 scatter epilogue: rows are input pixels [n,H,W], columns (dy,dx,co)
    int convt; int Cout;
    // batched (blockIdx.z) strides, elements
    int batch; int64_t strideA, strideB, strideBias, strideC;
    // set by the launcher (gemm_tiled, dense A): 1-D grid, the column tiles of a row tile run side by side on one XCD
    int nfast = 0, nx = 0, ny = 0;
    // relaxed / reduced numerics: B cut into bf16 terms (split_mfma.hpp split_weights), per batch strideBsplit uint16 apart; or null
    const uint16_t* Bsplit = nullptr; int64_t strideBsplit = 0;
};
void gemm(const GemmDesc& d, hipStream_t s);
// Direct conv (k x k, same padding, Cout % 4 == 0) for tiny contractions / odd shapes.
void conv_direct(const float* x, int n, int h, int w, int cin, const float* wt, const float* bias, int kh, int kw,
                 int cout, int relu, float* y, hipStream_t s);
void dwconv3x3(const float* x, int n, int h, int w, int c, const float* wt, const float* bias, int relu, float* y,
               hipStream_t s);
// dw 3x3 + pw 1x1 in one pass (8 <= C <= 32); same arithmetic as the two separate kernels
bool dwpw_fused_supported(int cin, int cout);
void dwpw_fused(const float* x, int n, int h, int w, int cin, const float* wdw, const float* bdw, int relu_dw, int cout,
                const float* wpw, const float* bpw, int relu_pw, float* y, hipStream_t s);
// depthwise 3x3 over concat([skip, centred-pad(up)]) without building the concatenation; false if unsupported
bool dwconv3x3_cat(const float* skip, int n, int h, int w, int cs, const float* up, int uh, int uw, int cu, const float* wt,
                   const float* bias, int relu, float* y, hipStream_t s);
// ---- kernels_det.hip: a whole DoubleConv block of the detection U-Net in one launch (LDS halo tiles):
//   [cat(skip, pad(ConvT2x2/s2(x1)))] -> dw3x3 -> pw1x1 -> dw3x3 -> pw1x1 -> y  [-> maxpool 2x2 | -> conv1x1(->1) -> sigmoid]
struct DoubleConvArgs {
    const float* skip;            // [n,h,w,CS]; encoder: the block input
    const float* x1;              // decoder: [n,h1,w1,CX] input of the ConvTranspose (else null)
    const float *wt, *bt;         // ConvT [2][2][CX][CS], [CS]
    const float *wd1, *bd1;       // dw1 [3][3][CIN], [CIN]   (CIN = CS, or 2*CS for a decoder block)
    const float *wp1, *bp1;       // pw1 [CIN][CMID], [CMID]
    const float *wd2, *bd2;       // dw2 [3][3][CMID], [CMID]
    const float *wp2, *bp2;       // pw2 [CMID][COUT], [COUT]
    const float *wf, *bf;         // final 1x1 conv [COUT], [1]
    float* y;                     // [n,h,w,COUT], or [n,h,w,1] with the final conv
    float* ypool;                 // [n,h/2,w/2,COUT] or null
    int n, h, w, h1, w1;
    int relu_d1, relu_p1, relu_d2, relu_p2, sigmoid;
    int tiles_x, tiles_y;         // filled by the launcher
    const float* tape = nullptr;  // wave streaming kernels (kernels_det_stream.hip): the block's weight tape(s) on the device
    int tape_len = 0;             // floats per tape
    const float* rtape = nullptr; // workgroup streaming kernels (kernels_det_rows.hip): the four waves' tapes
    int rtape_len = 0;
};
// host pointers to a block's weights, for building its tape
struct StreamWeights {
    const float *wt, *bt, *wd1, *bd1, *wp1, *bp1, *wd2, *bd2, *wp2, *bp2, *wf, *bf;
};
// true if a fused kernel exists for the shape (cs skip channels, cx ConvT input channels or 0, ...) at this
// fuse level (option "det_fuse": 1 = the shapes where fusion wins, 2 = every shape that has a kernel);
// launches it when `launch` is set.  *on_mfma: the block's pointwise convs / ConvTranspose run on the matrix cores
// (option "det_mfma").
// *path: which kernel family takes the shape with the current options and this request (a.n, a.h, a.w): 0 = LDS-tiled
// block, 1 = row-streaming wave kernel, 2 = row-streaming workgroup kernel.
bool double_conv_fused(const DoubleConvArgs& a, int cs, int cx, int cmid, int cout, bool pool, bool final_conv, int fuse_level,
                       bool launch, hipStream_t s, bool* on_mfma = nullptr, int* path = nullptr);
// ---- kernels_det_stream.hip (r4): the same blocks as row-streaming register kernels for the full-resolution levels;
// true if the shape has one (launches it when `launch` is set).  Same bits as the tiled blocks and the per-op kernels.
// With `hw` / `tape_out` the block's weight tape is built (the weights in the order a row step consumes them; the caller
// uploads it and passes it in DoubleConvArgs::tape); *tape_len = floats per tape.
bool double_conv_stream(const DoubleConvArgs& a, int cs, int cx, int cmid, int cout, bool pool, bool final_conv, bool launch, hipStream_t s,
                        const StreamWeights* hw = nullptr, std::vector<float>* tape_out = nullptr, int* tape_len = nullptr);
// ---- kernels_det_rows.hip (r4): the blocks of the 16-64-channel levels as row-streaming workgroup kernels; same contract
// (the tape holds the four waves' depthwise weights).  At launch time false also when the geometry does not fit (then the
// tiled block runs).
bool double_conv_rows(const DoubleConvArgs& a, int cs, int cx, int cmid, int cout, bool pool, bool final_conv, bool launch, hipStream_t s,
                      const StreamWeights* hw = nullptr, std::vector<float>* tape_out = nullptr, int* tape_len = nullptr);
// the launch-time conditions of double_conv_rows (request size under option value 1, geometry): true if it will run
bool double_conv_rows_takes(const DoubleConvArgs& a, int cx);
void maxpool(const float* x, int n, int h, int w, int c, int kh, int kw, float* y, hipStream_t s);
void avgpool(const float* x, int n, int h, int w, int c, int kh, int kw, float* y, hipStream_t s);
void padcat(const float* skip, int n, int sh, int sw, int cs, const float* x, int h, int w, int cx, float* y,
            hipStream_t s);
void sigmoid(const float* x, float* y, int64_t count, hipStream_t s);
// pointwise conv with Cout == 1 (+ optional sigmoid): y[p] = act(b + sum_c x[p,c] w[c])
void conv1x1_cout1(const float* x, int64_t pixels, int cin, const float* wt, const float* bias, int do_sigmoid,
                   float* y, hipStream_t s);
// [N,1,W,C] -> [W,N,C]
void to_seq(const float* x, int n, int w, int c, float* y, hipStream_t s);
// GRU gates for one time step, both directions (grid.z = dir).
// gx: [2][T*N][3H] (dir-major), gh: [2][N][3H], h: [2][N][H], y: [T][N][2H].
void gru_gates(const float* gx, const float* gh, float* h, float* y, int T, int N, int H, int step, hipStream_t s);
// log_softmax over C (+ optional -inf masking of excluded labels) + argmax.
// logits/logp: [rows][C]; labels: [rows] (first max).  logp may be null.  Returns false (nothing launched)
// if C is too large for the kernel's LDS staging (more than ~630 classes).
bool log_softmax_argmax(const float* logits, int64_t rows, int c, const uint8_t* d_excluded /*[C] or null*/,
                        float* logp, int32_t* labels, hipStream_t s);
// Ragged sequence batch (lines sorted by length, rows off[t] + m); see kernels_nn.hip.
void to_seq_packed(const float* x, int n, int T, int c, const int32_t* d_pos, const int32_t* d_off, float* y,
                   hipStream_t s);
void gru_gates_packed(const float* gx, const float* gh, float* h, float* y, const int32_t* d_Tm, const int32_t* d_off,
                      int64_t R, int Mcap, int active, int H, int step, hipStream_t s);
// Fused recurrent step (hidden GEMM on 16x16x4 fp32 MFMA + gates), both directions.
// hT_in/hT_out: [2][H][Mcap] transposed state (Mcap % 4 == 0), ping-ponged by the caller.
bool gru_step_fused(const float* gx, const float* wh, const float* bh, const float* hT_in, float* hT_out, float* y,
                    const int32_t* d_Tm, const int32_t* d_off, int64_t R, int Mcap, int active, int H, int step,
                    hipStream_t s);
// ---- kernels_gru.hip: all time steps of one bidirectional GRU layer in ONE persistent launch.
// d_sync: gru_persistent_sync_words(M) words of scratch (zeroed by the call); its last word is non-zero
// afterwards if a wait inside the kernel timed out.  Returns false (nothing launched) if the shape is not
// supported (H, more than 4096 lines, Tmax beyond the LDS table, a device that cannot keep a whole group of clusters
// resident): the caller then runs gru_step_fused per step.
// hx: gru_persistent_exchange_bytes() of scratch, the hand-off buffer between the workgroups; gru_persistent_prepare marks
// every word of it "unwritten" (the data is its own flag); call it on a stream ordered before gru_persistent.
size_t gru_persistent_sync_words(int M);
size_t gru_persistent_exchange_bytes(const int32_t* h_Tm, int M, int H);
bool gru_persistent_supported(const int32_t* h_Tm, int M, int Tmax, int64_t R, int H);
bool gru_tile_plan(const int32_t* h_Tm, int M, int H, int* ncl, int* waves, int16_t* tiles /* [512] */);  // host only: the deal of row tiles to waves
hipError_t gru_persistent_prepare(float* hx, const int32_t* h_Tm, int M, int H, hipStream_t s);
// h_Tm: the same lengths as d_Tm on the host (descending) — the deal of row tiles to waves is computed from them.
bool gru_persistent(const float* gx, const float* wh, const float* bh, float* y, float* hx, const int32_t* d_Tm, const int32_t* d_off,
                    const int32_t* h_Tm, int64_t R, int M, int Tmax, int H, uint32_t* d_sync, hipStream_t s);
bool gru_general_tile_plan(const int32_t* h_Tm, int M, int Tmax, int H, int cap, int* ncl, int16_t* tiles /* [512] or null */, int kernel /* 0 fp32, 1 / 2: split with 3 / 2 planes */);
// ---- kernels_gru_split.hip: the same launch for numerics != exact: hidden contraction on the bf16 matrix cores with the state
// cut into np (2: reduced, 3: relaxed) bf16 planes.  hx: gru_split_exchange_bytes() of scratch, marked by gru_split_prepare on a
// stream ordered before; y needs no marks.  d_sync as above.
size_t gru_split_exchange_bytes(const int32_t* h_Tm, int M, int H, int np);
bool gru_split_supported(const int32_t* h_Tm, int M, int Tmax, int64_t R, int H, int np);
hipError_t gru_split_prepare(uint16_t* hx, const int32_t* h_Tm, int M, int H, int np, hipStream_t s);
bool gru_persistent_split(const float* gx, const float* wh, const float* bh, float* y, uint16_t* hx, const int32_t* d_Tm, const int32_t* d_off,
                          const int32_t* h_Tm, int64_t R, int M, int Tmax, int H, int np, uint32_t* d_sync, hipStream_t s);
void ctc_collapse_packed(const int32_t* labels, const int32_t* d_Tm, const int32_t* d_off, int M, int Tmax,
                         uint32_t* out_labels, uint32_t* out_pos, int32_t* out_count, hipStream_t s);
void argmax_rows(const float* x, int64_t rows, int c, const uint8_t* d_excluded, int32_t* labels, hipStream_t s);
// ---- kernels_beam.hip: CTC prefix beam search (rten decode_beam) on the packed log-probabilities, one workgroup
// per line; same results as the host's ctc_beam_search.  d_nodes / d_posn: M * ctc_beam_arena_entries(Tmax, width)
// int2 each (scratch).  Outputs in the layout of ctc_collapse_packed.  false if (C, width) is not supported.
bool ctc_beam_supported(int C, int width);
size_t ctc_beam_arena_entries(int Tmax, int width);
bool ctc_beam_packed(const float* logp, const int32_t* d_Tm, const int32_t* d_off, int M, int Tmax, int C, int width,
                     const uint8_t* d_excluded, int2* d_nodes, int2* d_posn, uint32_t* out_labels, uint32_t* out_pos,
                     int32_t* out_count, hipStream_t s);
// Greedy CTC collapse (rten decode_greedy): labels [T][N] -> per line (label,pos) lists.
void ctc_collapse(const int32_t* labels, int T, int N, uint32_t* out_labels, uint32_t* out_pos, int32_t* out_count,
                  hipStream_t s);

// ---- kernels_rec.hip: ragged batch of the recognition conv stack -----------
// All width groups of one request in one NHWC buffer: group-major, then image, y, x.
struct RaggedView {          // device pointers live in one metadata upload; passed by value
    int G;                   // groups
    int H;                   // image height at this layer (same for every group)
    const int32_t* W;        // [G] image width at this layer
    const int32_t* n;        // [G] images per group
    const int64_t* poff;     // [G+1] pixel offset of each group in the buffer
    const int32_t* toff128;  // [G+1] cumulative 128-pixel tile counts
    const int32_t* toff256;  // [G+1] cumulative 256-pixel tile counts
    const int32_t* loff;     // [G+1] cumulative image (line) counts
    int ntiles128, ntiles256;
    const int32_t* toff2d;   // [G+1] cumulative TH x TW patch counts (conv3x3_ragged), TH = 128 / tw
    int ntiles2d, tw;
    // the same with the patches tiling the whole group's strip of images (flat column = img * Wp + x; Wp = W, or W
    // rounded up to even in the *_flat2 table of a layer whose epilogue pools horizontally)
    const int32_t* toff2d_flat;
    const int32_t* toff2d_flat2;
    int ntiles2d_flat, ntiles2d_flat2;
    const int32_t* toff2d_gap;   // 8 x 16 patches over the strip with >= 1 empty column between images (conv12_fused_ragged)
    int ntiles2d_gap;
    int64_t max_tile_px_;    // most pixels in the images one flat tile touches (host)
    int min_w;               // narrowest image at this layer (host)
    int64_t pixels;          // total pixels (host)
    int max_w;               // widest image at this layer (host)
};
void conv1_relu_pool_ragged(const float* x, const RaggedView& in, const float* wt, const float* bias, int cout,
                            float* y, const RaggedView& out, hipStream_t s);
void pool_ragged(const float* x, const RaggedView& in, int c, int kh, int kw, bool avg, float* y, const RaggedView& out,
                 hipStream_t s);
void to_seq_packed_ragged(const float* x, const RaggedView& in, int c, const int32_t* d_pos, const int32_t* d_off,
                          float* y, hipStream_t s);
// AvgPool (in.H, 1) + the sequence packing in one pass (`in`: geometry before the pool, `seq`: after it, height 1)
void avgpool_to_seq_ragged(const float* x, const RaggedView& in, const RaggedView& seq, int c, const int32_t* d_pos,
                           const int32_t* d_off, float* y, hipStream_t s);
// conv1 (Cin = 1) + ReLU + pool 2x2 + conv2 + ReLU + pool 2x2 in one launch; false = not this shape (run the two ops)
// w2split: conv2's weights cut into bf16 terms (conv12_split_weights) or null; used when the calling engine's numerics are not exact
bool conv12_fused_ragged(const float* x, const RaggedView& in0, const RaggedView& mid, const float* w1, const float* b1,
                         int c1, const float* w2, const float* b2, int c2, float* y, const RaggedView& out, hipStream_t s,
                         const uint16_t* w2split = nullptr);
void conv12_split_weights(const float* w2_host /* [288][64] */, std::vector<uint16_t>* out);   // host
// returns false if the shape is not supported (caller falls back to the per-group path)
// wsplit: the weights cut into bf16 terms (split_mfma.hpp split_weights) or null; used when the calling engine's numerics are not exact
bool conv3x3_ragged(const float* x, const RaggedView& rv, int cin, const float* wt, const float* bias, int cout, int relu,
                    int ph, int pw, float* y, const RaggedView& out, hipStream_t s, const uint16_t* wsplit = nullptr);
void split_weights(const float* w_host, int K, int N, int ldw, std::vector<uint16_t>* out);   // host; N % 128 == 0, K % 16 == 0 (split_mfma.hpp)

// ---- kernels_lines.hip ----------------------------------------------------
struct LineDesc {      // one text line to crop (recognition.rs:91-126)
    int32_t page;      // index into page pointer table
    int32_t poly_off;  // first vertex in the packed polygon array
    int32_t poly_n;    // vertex count
    int32_t top, left, bh, bw;  // polygon bounding rect (exclusive bottom/right)
    int32_t resized_w;
    int32_t out_w;     // padded width of this line's batch (its width group)
    int64_t out_off;   // float offset of this line's [out_h, out_w] image in the output buffer
};
// Fill + gather + bilinear resize + right-pad with -0.5; every line writes its own
// [out_h, out_w] image at d_out + out_off (lines of different width groups in one launch).
void crop_lines(const float* const* d_pages, const int32_t* d_page_hw /*[pages][2]*/, const LineDesc* d_lines,
                const int32_t* d_poly /*(y,x) pairs*/, int n_lines, int out_h, float* d_out, hipStream_t s);

// kernels_peaks.hip
void measure_peaks(double* mfma_tflops, double* copy_gbps);


// kernels_jpeg.hip — the GPU half of the JPEG hand-off (jpeg.hpp)
size_t jpeg_sample_bytes(const jpeg::Coefficients& c);
void jpeg_decode(const jpeg::Coefficients& c, const uint64_t* d_mask, const uint32_t* d_offset, const int16_t* d_values,
                 const uint16_t* d_quant, uint8_t* d_samples, uint8_t* d_rgb, hipStream_t s);

}  // namespace k
}  // namespace ocrs
