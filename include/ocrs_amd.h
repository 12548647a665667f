/*
 * ocrs_amd.h — C ABI of the MI355X-native OCR engine (libocrs_amd.so).
 *
 * This is the drop-in boundary for the hot path of robertknight/ocrs
 * (prepare_input -> detect_words -> find_text_lines -> recognize_text).  Every
 * entry point names the reference interface it replaces (paths relative to
 * the reference tree).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every function returns an ocrs_status; on failure ocrs_last_error()
 *     (thread-local) holds the message the reference would have put into its
 *     anyhow::Error / ModelRunError (ocrs/src/errors.rs:6-25).  Nothing aborts.
 *   - all handles are safe to use from several host threads at once (the
 *     reference calls Model::run concurrently from a rayon pool,
 *     ocrs/src/recognition.rs:465-485; models are `Send + Sync`,
 *     detection.rs:67, recognition.rs:316).
 *   - buffers returned through `T**` are allocated by the library and released
 *     with ocrs_buffer_free().
 *   - a RotatedRect crosses the boundary as 6 floats:
 *     center.x, center.y, up.x, up.y, width, height
 *     (rten_imageproc::RotatedRect::new(center, up_axis, width, height),
 *     ctor order shown at recognition.rs:582).
 */
#ifndef OCRS_AMD_H
#define OCRS_AMD_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define OCRS_API __attribute__((visibility("default")))
#else
#define OCRS_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ocrs_status {
    OCRS_OK = 0,
    OCRS_ERR_INVALID_ARGUMENT = 1,
    OCRS_ERR_MODEL_NOT_LOADED = 2, /* lib.rs:197,211,254,274 */
    OCRS_ERR_MODEL_DIMS = 3,       /* detection.rs:141-144 "failed to get model dims" */
    OCRS_ERR_RUN_FAILED = 4,       /* ModelRunError::RunFailed, errors.rs:8 */
    OCRS_ERR_WRONG_OUTPUT = 5,     /* ModelRunError::WrongOutput, errors.rs:11 */
    OCRS_ERR_IMAGE_SOURCE = 6,     /* ImageSourceError, preprocess.rs:38-46 */
    OCRS_ERR_DEVICE = 7,           /* HIP runtime failure / no GPU */
    OCRS_ERR_IO = 8,
    OCRS_ERR_CAPACITY = 9
} ocrs_status;

/* Message for the last failure on the calling thread ("" if none). */
OCRS_API const char* ocrs_last_error(void);

/* Layout version of the parameter structs below (ocrs_engine_params, ocrs_group_params) and of the argument lists: a binding
 * compares ocrs_abi_version() with the OCRS_ABI_VERSION it was built against when it loads the library and refuses a
 * mismatch, instead of passing a struct the library reads past the end of.  Bumped whenever a struct grows or an argument
 * list changes (6 = round 6; rounds 1-5 had no version: their structs were shorter).  Names of options removed since are
 * still accepted by ocrs_set_option and ignored. */
#define OCRS_ABI_VERSION 6u
OCRS_API uint32_t ocrs_abi_version(void);

/* Release any buffer handed out through a `T**` out-parameter. */
OCRS_API void ocrs_buffer_free(void* p);

/* Devices.  Every handle (model, engine, page) belongs to ONE HIP device, and every entry point binds the calling
 * host thread to its handle's device for the duration of the call: one process may hold engines on several GPUs
 * (and use them from any thread), or one process per GPU may each use its own.  The reference engine is immutable
 * `&self` (ocrs/src/lib.rs:183-256), which is what makes both shapes natural.
 *   ocrs_device_count  HIP devices visible
 *   ocrs_set_device    the DEFAULT device: where handles created without an explicit device live (the
 *                      one-process-per-GPU deployment calls it once, with LOCAL_RANK; initially 0)
 *   ocrs_get_device    the current default */
OCRS_API ocrs_status ocrs_device_count(int* n);
OCRS_API ocrs_status ocrs_set_device(int device);
OCRS_API ocrs_status ocrs_get_device(int* device);

/* rten::ctc::CtcDecoder::decode_beam (as called at ocrs/src/recognition.rs:512-514) on a host matrix of
 * log-probabilities [T][C] (blank = class 0): the steps (label, position) of the best prefix.  `impl`: 0 = the
 * host implementation the engine uses, 1 = the same function written as the algorithm is usually stated (slow;
 * the test oracle of 0), 2 = the HIP kernel the engine uses for DecodeMethod::BeamSearch (uploads the matrix).
 * Outputs are malloc'ed (ocrs_buffer_free). */
OCRS_API ocrs_status ocrs_ctc_beam_search(const float* logp, int t, int c, uint32_t width, int impl, uint32_t** labels,
                                          uint32_t** positions, size_t* n);

/* Test hook (host only, no GPU work): how the persistent GRU kernel would deal the 16-line row tiles of a request
 * to its waves.  lengths_desc = sequence lengths of the lines, descending (tile k = lines 16k .. 16k+15).
 * *n_clusters = clusters per direction, *waves = 4 = wave slots per cluster; tiles[4s .. 4s+3] for slot
 * s = cluster * 4 + wave.  Tile indices longest first, -1 = none.  OCRS_ERR_CAPACITY if the shape has no persistent plan. */
OCRS_API ocrs_status ocrs_gru_tile_plan(const int32_t* lengths_desc, size_t n_lines, int hidden, int32_t* n_clusters, int32_t* waves,
                                        int16_t* tiles);

/* Test hook (host only, no GPU work) for the request coalescer behind the one-page entry points (ocrs_engine_params.coalesce):
 * n_threads callers submit requests_per_thread requests each (of two incompatible kinds, weights 1..2 pages); request
 * ids divisible by fail_every (> 0) fail.  out = {batches run, requests carried, errors delivered to their callers,
 * requests that were run twice / not at all / got a wrong result or shared a batch with the other kind, largest batch
 * in pages}. */
OCRS_API ocrs_status ocrs_coalescer_selftest(int n_threads, int requests_per_thread, int max_active, int max_pages,
                                             long window_us, int fail_every, uint64_t out[5]);

/* Integer tuning options (no reference counterpart: RTen's equivalents are compile-time).  They select between kernels
 * that compute the SAME bits — results never depend on an option; the tests run the alternatives against each other.
 *
 * Scope.  The process holds the defaults: initial value from the environment variable OCRS_<NAME IN CAPITALS>, read once
 * when the library first needs an option; ocrs_set_option changes a default.  An engine COPIES the defaults when it is
 * created (ocrs_engine_new / ocrs_engine_group_new) and from then on owns its copy: ocrs_engine_set_option changes one
 * engine only, two engines in one process can differ, and changing a default never affects an engine that exists.
 * ocrs_model_run on a bare model uses the defaults.  Set an engine's options before it is in use by other threads.
 *
 *   "gru_mode"        0 = one persistent kernel per GRU layer (default), 1 = one launch per time step
 *   "gru_gates"       1 = requests small enough that every 16-line row tile gets its own cluster of workgroups run the
 *                     gate-per-wave recurrence kernel (default), 0 = always the general persistent kernel (exact numerics
 *                     only: the relaxed / reduced modes have one persistent recurrence kernel for every request size)
 *   "gru_local"       persistent GRU kernels: 1 = a cluster of workgroups that finds itself on one XCD hands its state
 *                     over through that XCD's L2 (default), 0 = always through write-through stores
 *   "det_fuse"        1 = fused DoubleConv blocks of the detection U-Net where they win (default), 2 = for every block
 *                     shape that has a fused kernel, 0 = per-operator kernels only
 *   "det_mfma"        1 = the LDS-tiled DoubleConv blocks run their pointwise convolutions and ConvTranspose on the matrix
 *                     cores (default), 0 = thread-per-pixel VALU kernels
 *   "det_stream"      1 (default) = the DoubleConv blocks of the U-Net's full-resolution level run as row-streaming wave
 *                     kernels (a wave walks down a 64-column strip, a lane keeps its pixel's channels in registers,
 *                     horizontal taps through DPP lane shifts), rows per wave chosen from the request's size; 8 / 14 / 32 =
 *                     the same with that many rows per wave; 0 = the LDS-tiled blocks
 *   "det_rows"        1 (default) = the DoubleConv blocks of the 16-64-channel levels run as row-streaming workgroup
 *                     kernels for requests of up to 8 pages; 8 / 14 / 20 / 32 = those kernels for every request, with that
 *                     many rows per workgroup; 0 = the LDS-tiled blocks always
 *   "ccl_quad"        1 (default) = the component-labelling and root-compaction kernels of detect_words handle four mask
 *                     pixels per thread when the page width is a multiple of 4; 0 = one pixel per thread
 *   "conv12_fuse"     1 (default) = the first two recognition convs and their 2x2 pools run as one kernel (conv1 into an
 *                     LDS tile per patch, conv2's matrix-core loop reads its operand there); 0 = two kernels
 *   "conv_flat"       recognition 3x3 convs: the 128-pixel patches tile a width group's strip of images side by side
 *                     (1, default: only a group's last patch is ragged) or every image on its own (0)
 *   "beam_gpu"        1 = DecodeMethod::BeamSearch runs on the GPU (default), 0 = on the host (threaded over lines)
 *
 * OCRS_ERR_INVALID_ARGUMENT for an unknown name.  (Rounds 2-4 carried 28 process-wide options, many of them switches for
 * experiments that had lost their A/B; round 5 removed those kernels and moved what is configuration — numerics, request
 * coalescing, host threads, the sub-request size, the group's dealing blocks — into ocrs_engine_params / ocrs_group_params.) */
OCRS_API ocrs_status ocrs_set_option(const char* name, long value);
/* Name of option `index` (0 .. count - 1); *name = NULL past the last one. */
OCRS_API ocrs_status ocrs_option_name(int index, const char** name);

/* Memory the library holds on a device: out = {device bytes in use, device bytes cached (free, reusable), cap on the
 * cached device bytes, peak of bytes in use, hipMalloc calls, blocks returned to the driver, and the same six numbers for
 * the pinned host staging pool of that device's context}.  device < 0: the default device. */
OCRS_API ocrs_status ocrs_device_pool_stats(int device, uint64_t out[12]);
/* Caps on the CACHED bytes (0 = leave as is).  Defaults: a quarter of the device's memory (environment variable
 * OCRS_POOL_CAP_GB, read once per device, overrides) and 1 GiB of pinned host memory.  Blocks above a cap are returned
 * to the driver by a background thread, never on a request's thread (hipFree waits for the device). */
OCRS_API ocrs_status ocrs_device_pool_configure(int device, uint64_t device_cached_cap_bytes, uint64_t pinned_cached_cap_bytes);
/* Gives back what the library holds for re-use on `device`: every cached block and the activation arena the recognition conv
 * stacks of all requests share (kept at the size of the largest request seen).  Bytes in use afterwards = weights, pages the
 * caller still holds, requests in flight.  Waits for the conv stacks in flight; for idle moments, not for the request path. */
OCRS_API ocrs_status ocrs_device_pool_trim(int device);

/* Isolation of the bf16-MFMA kernels (engines with numerics != exact), per device.  Round 5 observed OTHER requests' line crops
 * change while those kernels ran beside them; round 6 reproduced it from the two real kernels alone (tools/hazard_repro.hip:
 * every twin launch of the crop kernel differs while a dense bf16-MFMA kernel whose matrix instructions read VGPR operands shares its compute units,
 * none when the two are confined to disjoint units; DESIGN.md §4.4 "Concurrency").  The library therefore never lets kernels of
 * different requests overlap on a device that has such an engine.
 *   OCRS_ISOLATION_AUTO (default)   every call on the device enqueues on ONE stream while such an engine exists;
 *   OCRS_ISOLATION_NONE             nothing (diagnostics: tools/hazard_canary.py reproduces the hazard with it — never in
 *                                   production).
 * Exact-only devices are untouched by any of this (and covered by a standing canary: tests/test_gpu_r6.py).  A request is told
 * its regime once, when it starts, and keeps it; a change of regime — this call, the first engine with numerics != exact
 * created on the device, the last one destroyed — BLOCKS until the requests in flight on the device have finished (new ones
 * wait meanwhile) and the device is idle, so requests of two regimes never run side by side.  Do not call from inside a
 * request.  device < 0: the default device. */
typedef enum { OCRS_ISOLATION_AUTO = 0, OCRS_ISOLATION_NONE = 1 } ocrs_isolation;
OCRS_API ocrs_status ocrs_device_set_isolation(int device, ocrs_isolation policy);
/* out = {what a request starting now is told: 0 free (own stream per call), 1 serial (one stream);
 *        engines with numerics != exact alive on the device; compute units of the device}. */
OCRS_API ocrs_status ocrs_device_isolation(int device, int out[3]);

/* ------------------------------------------------------------------------
 * L2 seam: `trait Model` (ocrs/src/model.rs:6-17) and its rten impl
 * (model.rs:19-41).  A Rust `impl Model for HipModel` binds these four calls
 * (INTEGRATION.md §1).
 * ---------------------------------------------------------------------- */
typedef struct ocrs_model ocrs_model;

/* rten::Model::load_file (ocrs-cli/src/models.rs:100-107): loads an `.ocrsm`
 * fixed-graph file and uploads the weights to HBM. */
OCRS_API ocrs_status ocrs_model_load_file(const char* path, ocrs_model** out);
OCRS_API ocrs_status ocrs_model_load_bytes(const void* data, size_t len, ocrs_model** out);
/* The same with the weights on an explicit device (device < 0: the default device).  The file is validated on the
 * host first; a malformed file fails with OCRS_ERR_IO before any device is touched. */
OCRS_API ocrs_status ocrs_model_load_file_on_device(const char* path, int device, ocrs_model** out);
OCRS_API ocrs_status ocrs_model_load_bytes_on_device(const void* data, size_t len, int device, ocrs_model** out);
/* Device that holds the model's weights (-1 for a callback model). */
OCRS_API ocrs_status ocrs_model_device(const ocrs_model* m, int* device);

/* A model implemented by the caller — the counterpart of implementing
 * `trait Model` in Rust (the reference's tests inject FakeDetectionModel /
 * FakeRecognitionModel this way, lib.rs:339-422).  `run` receives a contiguous
 * NCHW f32 input, must malloc() the output, fill out_shape/out_ndim (<= 4) and
 * return 0; non-zero means failure (-> OCRS_ERR_RUN_FAILED). */
typedef int (*ocrs_model_run_fn)(void* user, const float* input, const int64_t in_shape[4], float** output,
                                 int64_t out_shape[4], int* out_ndim);
OCRS_API ocrs_status ocrs_model_from_callback(const int64_t input_shape[4], /* -1 = symbolic */
                                     ocrs_model_run_fn run, void* user, ocrs_model** out);

/* Model::input_shape (model.rs:20-31): NCHW; dims[i] = -1 and is_fixed[i] = 0
 * for Dimension::Symbolic. */
OCRS_API ocrs_status ocrs_model_input_shape(const ocrs_model* m, int64_t dims[4], uint8_t is_fixed[4]);

typedef struct ocrs_run_options {
    int timing; /* RunOptions.timing (detection.rs:178-182): print per-op times */
} ocrs_run_options;

/* Model::run (model.rs:33-40).  `input` is host memory, contiguous NCHW f32.
 * Detection graphs return [N,1,H,W] probabilities; recognition graphs return
 * [T,N,C] log-probabilities (recognition.rs:399-401).  *output is host memory
 * owned by the caller (ocrs_buffer_free). */
OCRS_API ocrs_status ocrs_model_run(const ocrs_model* m, const float* input, const int64_t in_shape[4],
                           const ocrs_run_options* opts, float** output, int64_t out_shape[4], int* out_ndim);

/* Algorithmic FLOPs of one forward at the given input shape. */
OCRS_API ocrs_status ocrs_model_flops(const ocrs_model* m, const int64_t in_shape[4], double* flops);

OCRS_API void ocrs_model_free(ocrs_model* m);

/* ------------------------------------------------------------------------
 * L4 API: OcrEngine (ocrs/src/lib.rs:111-301) with the grey page resident in
 * HBM between calls (OcrInput, lib.rs:125-128).
 * ---------------------------------------------------------------------- */
typedef struct ocrs_engine ocrs_engine;
typedef struct ocrs_page ocrs_page; /* OcrInput */

typedef enum ocrs_numerics { OCRS_NUMERICS_EXACT = 0, OCRS_NUMERICS_RELAXED = 1, OCRS_NUMERICS_REDUCED = 2 } ocrs_numerics;

typedef enum ocrs_decode_method { /* DecodeMethod, recognition.rs:198-205 */
    OCRS_DECODE_GREEDY = 0,
    OCRS_DECODE_BEAM_SEARCH = 1
} ocrs_decode_method;

/* OcrEngineParams (lib.rs:38-71).  Models are borrowed: they must outlive the
 * engine.  NULL strings select the defaults (DEFAULT_ALPHABET, lib.rs:34). */
typedef struct ocrs_engine_params {
    const ocrs_model* detection_model;   /* may be NULL */
    const ocrs_model* recognition_model; /* may be NULL */
    int debug;
    ocrs_decode_method decode_method;
    uint32_t beam_width;
    const char* alphabet;      /* UTF-8 */
    const char* allowed_chars; /* UTF-8 */
    /* --- no reference counterpart (RTen is fp32 on the CPU, one page per call) --- */
    ocrs_numerics numerics;    /* OCRS_NUMERICS_EXACT (0, default): every kernel follows the numeric spec, results are
                                * bit-identical to the CPU oracle.  The other two are EXPERIMENTAL.  OCRS_NUMERICS_RELAXED: fp32-class arithmetic that is not
                                * reproducible on a CPU — hardware exp / rcp in the recurrence's gates; the operands of the
                                * recognition convs, the GRU input projections and the recurrence's own contraction cut
                                * into three bf16 terms on the bf16 matrix cores (products good to 2^-23), the matrix core's
                                * accumulation order.  OCRS_NUMERICS_REDUCED: the same with two bf16 terms per
                                * operand (products good to 2^-15: a 16-bit significand).  Boxes and tokens are expected, not
                                * guaranteed, to match the exact mode: DESIGN.md "what exactness costs" has the measured flips.
                                * While an engine of these modes exists, every call on its device (of any engine of the
                                * process) runs its kernels one at a time on one stream (ocrs_device_set_isolation below says why and what a
                                * change of regime waits for): create it before serving traffic */
    int coalesce;              /* one-page calls that wait at the same time are merged into one ragged request per stage
                                * (lines are independent: nobody's bits change).  Merged batches in flight per stage:
                                * 0 = default (2), negative = every call runs on its own */
    int coalesce_pages;        /* pages per merged batch (0 = default 16); requests of half that size or more are never merged */
    int coalesce_window_us;    /* while other batches are in flight, how long the next one lets its queue fill
                                * (0 = default 300, negative = no wait) */
    int layout_threads;        /* host threads ocrs_engine_find_text_lines_batch may use (0 = one per page up to the host's cores) */
    int64_t rec_max_pixels;    /* input pixels (padded line batch) one recognition sub-request may hold; larger requests run
                                * as consecutive sub-requests (0 = 2e9, sized for the activation memory of one GPU) */
} ocrs_engine_params;

/* OcrEngine::new (lib.rs:132-180).  The engine lives on its models' device (both models must be on the same one;
 * an engine with callback models only lives on the default device); pages it prepares live there too, and a page
 * can only be given to an engine of its own device (OCRS_ERR_INVALID_ARGUMENT otherwise). */
OCRS_API ocrs_status ocrs_engine_new(const ocrs_engine_params* params, ocrs_engine** out);
OCRS_API void ocrs_engine_free(ocrs_engine* e);
OCRS_API ocrs_status ocrs_engine_device(const ocrs_engine* e, int* device);
/* This engine's copy of a tuning option (see ocrs_set_option for the list and the scoping rule).  Also accepted, by
 * field name: "coalesce", "coalesce_pages", "coalesce_window_us", "layout_threads", "rec_max_pixels" (the values as
 * stored: coalesce 0 = off); "numerics" can be read but is fixed when the engine is created. */
OCRS_API ocrs_status ocrs_engine_set_option(ocrs_engine* e, const char* name, long value);
OCRS_API ocrs_status ocrs_engine_get_option(const ocrs_engine* e, const char* name, long* value);

typedef enum ocrs_dim_order { OCRS_HWC = 0, OCRS_CHW = 1 } ocrs_dim_order; /* DimOrder, preprocess.rs:50-57 */
typedef enum ocrs_pixel_type { OCRS_U8 = 0, OCRS_F32 = 1 } ocrs_pixel_type; /* ImagePixels, preprocess.rs:9-14 */

/* ImageSource::from_bytes (preprocess.rs:81-101): validates only. */
OCRS_API ocrs_status ocrs_image_source_check_bytes(size_t len, uint32_t width, uint32_t height, uint32_t* channels);

/* ImageSource::from_tensor + OcrEngine::prepare_input (preprocess.rs:105-123,
 * lib.rs:183-187): converts to greyscale f32 [1,H,W] in [-0.5,0.5] on the GPU.
 * `pixels` is host memory. */
OCRS_API ocrs_status ocrs_engine_prepare_input(const ocrs_engine* e, const void* pixels, ocrs_pixel_type type,
                                      ocrs_dim_order order, int height, int width, int channels,
                                      ocrs_page** out);
/* Several equally sized host images in one call: all uploads and conversions are queued on one stream and waited
 * for once (OcrEngine::prepare_input, lib.rs:183-187, per image).  out[n] receives the pages.  The form bench.py's
 * headline times (host pixels in page-locked buffers, the upload inside the timed region). */
OCRS_API ocrs_status ocrs_engine_prepare_input_batch(const ocrs_engine* e, const void* const* pixels, size_t n,
                                                     ocrs_pixel_type type, ocrs_dim_order order, int height, int width,
                                                     int channels, ocrs_page** out);
/* Same, with `pixels` already resident in HBM (device pointer): what a GPU image decoder hands over (bench.py
 * --resident and its `value_resident` extra time this form). */
OCRS_API ocrs_status ocrs_engine_prepare_input_device(const ocrs_engine* e, const void* d_pixels, ocrs_pixel_type type,
                                             ocrs_dim_order order, int height, int width, int channels,
                                             ocrs_page** out);
OCRS_API void ocrs_page_free(ocrs_page* p);
OCRS_API ocrs_status ocrs_page_dims(const ocrs_page* p, int* height, int* width);
/* Copy the prepared grey page [H,W] f32 to host (OcrInput.image, lib.rs:127). */
OCRS_API ocrs_status ocrs_page_image(const ocrs_page* p, float* out_hw);

/* OcrEngine::detect_words (lib.rs:193-199 -> detection.rs:104-122).
 * *rects receives n x 6 floats in contour discovery order. */
OCRS_API ocrs_status ocrs_engine_detect_words(const ocrs_engine* e, const ocrs_page* page, float** rects, size_t* n);
/* Batched form: pages of ANY sizes (the reference takes any image per call, detection.rs:131-171; the model runs once over
 * the whole batch at its own fixed size, the size-dependent kernels once per distinct page size — r6; rounds 1-5 wanted one
 * size per batch); rects of page i are (*rects)[6*offsets[i] .. 6*offsets[i+1]); offsets has n_pages+1 entries.  Concurrent
 * one-page calls are merged the same way whatever their sizes. */
OCRS_API ocrs_status ocrs_engine_detect_words_batch(const ocrs_engine* e, const ocrs_page* const* pages, size_t n_pages,
                                           float** rects, size_t* offsets);

/* OcrEngine::detect_text_pixels (lib.rs:207-213 -> detection.rs:131-200):
 * writes the [H,W] probability map. */
OCRS_API ocrs_status ocrs_engine_detect_text_pixels(const ocrs_engine* e, const ocrs_page* page, float* out_hw);

/* OcrEngine::detection_threshold (lib.rs:282-287). */
OCRS_API float ocrs_engine_detection_threshold(const ocrs_engine* e);

/* OcrEngine::find_text_lines (lib.rs:222-228 -> layout_analysis.rs:158-233).
 * Host-side.  *line_rects receives the same n_words rects permuted into
 * reading order; line i owns rects [line_offsets[i], line_offsets[i+1]). */
OCRS_API ocrs_status ocrs_engine_find_text_lines(const ocrs_engine* e, const ocrs_page* page, const float* word_rects,
                                        size_t n_words, float** line_rects, size_t** line_offsets,
                                        size_t* n_lines);

/* The same for several pages at once (one host thread per page).  Words of page p
 * are word_rects[6*word_offsets[p] .. 6*word_offsets[p+1]).  *line_rects receives all
 * words, permuted into reading order page by page; line i (numbered across pages) owns
 * rects [line_offsets[i], line_offsets[i+1]); page p owns lines
 * [page_line_offsets[p], page_line_offsets[p+1]). */
OCRS_API ocrs_status ocrs_engine_find_text_lines_batch(const ocrs_engine* e, size_t n_pages, const float* word_rects,
                                                       const size_t* word_offsets, float** line_rects,
                                                       size_t** line_offsets, size_t** page_line_offsets);

/* One recognised character: TextChar (text_items.rs:48-54). */
typedef struct ocrs_text_char {
    uint32_t ch;                      /* Unicode scalar value */
    int32_t top, left, bottom, right; /* rten_imageproc::Rect */
} ocrs_text_char;

/* OcrEngine::recognize_text (lib.rs:237-256 -> recognition.rs:404-540).
 * Lines are given as in ocrs_engine_find_text_lines' output.  chars of line i
 * are (*chars)[char_offsets[i] .. char_offsets[i+1]); an empty range is the
 * reference's `None`.  char_offsets has n_lines+1 entries. */
OCRS_API ocrs_status ocrs_engine_recognize_text(const ocrs_engine* e, const ocrs_page* page, const float* line_rects,
                                       const size_t* line_offsets, size_t n_lines, ocrs_text_char** chars,
                                       size_t** char_offsets);
/* Batched form over several pages (lines of all pages share recognition
 * batches; padded widths stay those of recognition.rs:437).  page_line_offsets
 * has n_pages+1 entries indexing into line_offsets' line numbering. */
OCRS_API ocrs_status ocrs_engine_recognize_text_batch(const ocrs_engine* e, const ocrs_page* const* pages, size_t n_pages,
                                             const size_t* page_line_offsets, const float* line_rects,
                                             const size_t* line_offsets, size_t n_lines,
                                             ocrs_text_char** chars, size_t** char_offsets);

/* Raw CTC output for the same lines (labels and time steps of
 * CtcHypothesis::steps(), recognition.rs:257-289), for token-level parity. */
OCRS_API ocrs_status ocrs_engine_recognize_tokens(const ocrs_engine* e, const ocrs_page* page, const float* line_rects,
                                         const size_t* line_offsets, size_t n_lines, uint32_t** labels,
                                         uint32_t** positions, size_t** token_offsets);

/* The recognition model's output for the same lines, before masking and decoding (TextRecognizer::run,
 * recognition.rs:341-360): line i owns rows [row_offsets[i], row_offsets[i+1]) of the [rows][classes] matrix *logp — its
 * T_i time steps.  One request, never coalesced.  For tolerance checks between numerics modes. */
OCRS_API ocrs_status ocrs_engine_recognize_logits(const ocrs_engine* e, const ocrs_page* page, const float* line_rects,
                                         const size_t* line_offsets, size_t n_lines, float** logp, size_t** row_offsets,
                                         int* classes);

/* TextItem::rotated_rect (text_items.rs:18-30) for a TextLine / TextWord given its characters'
 * rects (n x {top,left,bottom,right}): minimum-area rectangle of the box corners, oriented
 * towards "up" (y = -1).  out6 = (center.x, center.y, up.x, up.y, width, height).  Host side. */
OCRS_API ocrs_status ocrs_text_item_rotated_rect(const int32_t* rects_tlbr, size_t n_chars, float out6[6]);
/* RotatedRect::corners (order pinned by text_items.rs:156-166): out8 = 4 x (x, y). */
OCRS_API ocrs_status ocrs_rotated_rect_corners(const float rect6[6], float out8[8]);

/* OcrEngine::prepare_recognition_input (lib.rs:268-278 -> recognition.rs:366-392):
 * *out receives a [height, width] f32 line image. */
OCRS_API ocrs_status ocrs_engine_prepare_recognition_input(const ocrs_engine* e, const ocrs_page* page, const float* line,
                                                  size_t n_words, float** out, int* height, int* width);

/* OcrEngine::get_text (lib.rs:290-300): UTF-8, lines joined by '\n'. */
OCRS_API ocrs_status ocrs_engine_get_text(const ocrs_engine* e, const ocrs_page* page, char** text);

/* ------------------------------------------------------------------------
 * JPEG hand-off.  The reference decodes image files on the host with the `image` crate before
 * prepare_input (ocrs-cli/src/main.rs:312-333: image::open(path).into_rgb8()).  Here the host does only what is
 * inherently serial — marker parsing and Huffman entropy decoding, baseline / extended sequential and progressive —
 * and the GPU does dequantisation, the 8x8 inverse DCT, chroma upsampling and YCbCr -> RGB with libjpeg's integer
 * arithmetic (jidctint.c islow, jdsample.c fancy upsampling, jdcolor.c), so the pixels equal libjpeg-turbo's / PIL's
 * bit for bit; what crosses PCIe is the sparse coefficient stream (*coef_bytes, ~0.5-1 byte per pixel) instead of
 * 3 bytes per pixel.  Unsupported flavours (arithmetic coding, lossless, 12-bit, CMYK, 4:4:0 or exotic sampling) return
 * OCRS_ERR_IMAGE_SOURCE: decode those on the host as the reference does and call ocrs_engine_prepare_input.
 *   ocrs_engine_prepare_input_jpeg  OcrEngine::prepare_input(ImageSource::from_bytes(into_rgb8(decode(file)))) in one call
 *   ocrs_jpeg_decode_rgb            the decoded RGB8 HWC pixels on the host (tests, debugging); *rgb: ocrs_buffer_free
 *   ocrs_jpeg_info                  host only: dimensions, component count, progressive?, non-zero coefficients
 * coef_bytes and the out-parameters of ocrs_jpeg_info may be NULL.
 * ---------------------------------------------------------------------- */
OCRS_API ocrs_status ocrs_engine_prepare_input_jpeg(const ocrs_engine* e, const void* jpeg, size_t len, ocrs_page** out,
                                                    size_t* coef_bytes);
OCRS_API ocrs_status ocrs_jpeg_decode_rgb(int device, const void* jpeg, size_t len, uint8_t** rgb, int* height, int* width,
                                          size_t* coef_bytes);
OCRS_API ocrs_status ocrs_jpeg_info(const void* jpeg, size_t len, int* height, int* width, int* components, int* progressive,
                                    size_t* nonzero);
/* Test hook (host only): what the host half hands to the GPU, dense.  geom = {width, height, components, hmax, vmax,
 * progressive, ycc} + per component {h, v, tq, width, height, blocks_w, blocks_h}; quant = 4 tables x 64, natural order;
 * *coef = n_blocks x 64 quantised coefficients in natural order, blocks in component order, row-major (ocrs_buffer_free). */
OCRS_API ocrs_status ocrs_jpeg_coefficients(const void* jpeg, size_t len, int32_t geom[28], uint16_t quant[256], int16_t** coef,
                                            size_t* n_blocks);

/* ------------------------------------------------------------------------
 * Several GPUs in one process: an engine group.  One engine (and one replica of the weights) per member device;
 * every page of a call is processed by a member of the device the page lives on, every member runs its share on a
 * worker thread and streams of its own, results come back in page order.  Pages are independent (OcrEngine is
 * immutable `&self`, ocrs/src/lib.rs:183-256), so there is no collective on the compute path; the only exchange is
 * the gather of the packed results, by one of two transports that deliver the same bytes:
 *   OCRS_GATHER_HOST  every member hands its results over through its own pinned staging / PCIe link and the calling
 *                     thread concatenates them
 *   OCRS_GATHER_RCCL  the packed results ({rect f32 x 6} per word, {char, box} per character), prefixed by their
 *                     length, are all-gathered device to device over xGMI (ncclCommInitAll communicator, one grouped
 *                     ncclAllGather) and read back from the root member
 *   OCRS_GATHER_AUTO  for ocrs_group_params.gather (the per-request gathers inside ocrs_group_detect_words_batch /
 *                     ocrs_group_recognize_text_batch): host — inside one process the results are on the host
 *                     already.  For ocrs_group_final_gather (the end-of-stream gather; north_star: "RCCL over xGMI
 *                     only for the final result gather"): RCCL when the group has two or more members and a
 *                     communicator can be had, else host.
 * librccl is loaded at run time (dlopen: `librccl.so.1`, or what OCRS_RCCL_LIB names) by the first gather that asks for
 * it; libocrs_amd.so itself does not depend on it.  RCCL not loadable, a communicator refused (RCCL does not accept the
 * same device twice: a group such as [0, 0] on a one-GPU box) — every such case falls back to the host transport and
 * ocrs_group_last_gather reports it; none is an error.
 *
 * Dealing.  Pages the caller made resident (ocrs_group_prepare_input_device_batch; ocrs_page handles) stay where they
 * are.  Pages the group places itself (ocrs_group_prepare_input_batch) go to the devices in contiguous blocks of
 * ceil(n / devices) pages, but at least `group_min_block` (option, default 8): a call of 16 pages on 8 devices uses
 * two of them, and successive calls start at successive devices — a device handed two pages runs its batch kernels
 * at a fraction of their efficiency.  Members that share a device split its pages the same way in blocks of at
 * least `group_shared_block` (16).  ocrs_group_deal is the block rule on its own, starting at member 0.
 * ---------------------------------------------------------------------- */
typedef struct ocrs_engine_group ocrs_engine_group;
typedef enum ocrs_gather_mode { OCRS_GATHER_AUTO = 0, OCRS_GATHER_HOST = 1, OCRS_GATHER_RCCL = 2 } ocrs_gather_mode;

typedef struct ocrs_group_params { /* OcrEngineParams (lib.rs:38-71) + the member devices */
    const void* detection_model;   /* `.ocrsm` image (as for ocrs_model_load_bytes) or NULL */
    size_t detection_model_len;
    const void* recognition_model;
    size_t recognition_model_len;
    const int* devices;            /* member m runs on devices[m] */
    size_t n_devices;
    int debug;
    ocrs_decode_method decode_method;
    uint32_t beam_width;
    const char* alphabet;
    const char* allowed_chars;
    ocrs_gather_mode gather;
    ocrs_numerics numerics;        /* as in ocrs_engine_params, for every member */
    int coalesce, coalesce_pages, coalesce_window_us, layout_threads;
    int64_t rec_max_pixels;
    int min_block;                 /* pages the group places itself go to a device in contiguous blocks of at least this
                                    * many (0 = default 8; a small call then uses fewer devices, successive calls rotate) */
    int shared_block;              /* the same between members that share one device (0 = default 16) */
} ocrs_group_params;

OCRS_API ocrs_status ocrs_engine_group_new(const ocrs_group_params* params, ocrs_engine_group** out);
OCRS_API void ocrs_engine_group_free(ocrs_engine_group* g);
OCRS_API ocrs_status ocrs_engine_group_size(const ocrs_engine_group* g, size_t* n_members);
/* Member i's engine (borrowed; usable with every ocrs_engine_* call) and device. */
OCRS_API ocrs_status ocrs_engine_group_member(const ocrs_engine_group* g, size_t i, const ocrs_engine** engine, int* device);
/* The dealing rule on its own (host only): block = max(ceil(n_pages / n_members), min(min_block, n_pages)) with
 * min_block = 0 meaning the default 8, member_of_page[i] = (i / block) mod n_members; pages_per_member may be NULL. */
OCRS_API ocrs_status ocrs_group_deal(size_t n_pages, size_t n_members, size_t min_block, size_t* member_of_page, size_t* pages_per_member);

/* OcrEngine::prepare_input (lib.rs:183-187) for n equally sized host images, dealt as described above.  out[n]
 * receives the pages (each lives on its member's device). */
OCRS_API ocrs_status ocrs_group_prepare_input_batch(const ocrs_engine_group* g, const void* const* pixels, size_t n,
                                                    ocrs_pixel_type type, ocrs_dim_order order, int height, int width,
                                                    int channels, ocrs_page** out);
/* The same with the images already resident on member devices (ocrs_device_malloc_on): each is converted where it is. */
OCRS_API ocrs_status ocrs_group_prepare_input_device_batch(const ocrs_engine_group* g, const void* const* d_pixels, size_t n,
                                                           ocrs_pixel_type type, ocrs_dim_order order, int height, int width,
                                                           int channels, ocrs_page** out);
/* OcrEngine::detect_words (lib.rs:193-199); every page is processed on the device it lives on (which must have a
 * member); output as ocrs_engine_detect_words_batch. */
OCRS_API ocrs_status ocrs_group_detect_words_batch(ocrs_engine_group* g, const ocrs_page* const* pages, size_t n_pages,
                                                   float** rects, size_t* offsets);
/* OcrEngine::recognize_text (lib.rs:237-256); arguments and output as ocrs_engine_recognize_text_batch.
 * (find_text_lines is host work: ocrs_engine_find_text_lines_batch serves a group as it is.) */
OCRS_API ocrs_status ocrs_group_recognize_text_batch(ocrs_engine_group* g, const ocrs_page* const* pages, size_t n_pages,
                                                     const size_t* page_line_offsets, const float* line_rects,
                                                     const size_t* line_offsets, size_t n_lines, ocrs_text_char** chars,
                                                     size_t** char_offsets);
/* The per-request gather on its own: payloads[m] / bytes[m] = member m's packed bytes (host memory); *out receives
 * their concatenation in member order through the group's per-request transport, offsets[G + 1] the boundaries. */
OCRS_API ocrs_status ocrs_group_gather(ocrs_engine_group* g, const void* const* payloads, const size_t* bytes, void** out,
                                       size_t* offsets);
/* The final result gather of a stream of requests: the same contract with the transport named per call (AUTO = RCCL
 * when the group has two or more members and RCCL can be had, else host — never an error for lack of RCCL). */
OCRS_API ocrs_status ocrs_group_final_gather(ocrs_engine_group* g, ocrs_gather_mode mode, const void* const* payloads,
                                             const size_t* bytes, void** out, size_t* offsets);
/* Test hook — host-side pre-flight of a multi-GPU deployment on a box with fewer GPUs (bench.py --replay; SURVEY §8e names the
 * host as the expected scaling limiter).  mode 1 = record: calls run as usual and the group keeps every page's word rects and
 * recognised lines, keyed by the host pixels the page was prepared from (ocrs_group_prepare_input_batch).  mode 2 = replay: a
 * member's share of a call does no GPU work — it sleeps seconds[stage] (stage 0 prepare, 1 detect, 2 recognize: what one
 * share takes on one GPU under load) and returns the recorded results of its pages — while dealing, worker threads and their
 * NUMA binding, payload packing, the per-request and final gathers and the reassembly in page order run as in production.
 * mode 0 = off (forgets the records).  Not for concurrent use with calls in flight. */
OCRS_API ocrs_status ocrs_group_set_replay(ocrs_engine_group* g, int mode, const double seconds[3]);
/* What member m has done so far, for diagnosing a multi-GPU run (the host side is the expected scaling
 * limiter): out = {shares of calls it ran, pages those carried, CPU nanoseconds of the host threads that ran them, their wall
 * nanoseconds, NUMA node of its GPU + 1 (0 = the host does not say), CPUs of that node (0 = no binding), shares that ran
 * bound to those CPUs, its device}.  A member's share of a call binds its thread to the CPUs of the NUMA node the member's
 * GPU hangs off (hipDeviceGetPCIBusId -> sysfs numa_node / cpulist) for the duration of the share and restores the thread's
 * mask afterwards; its page-locked staging is first touched from there.  Hosts without that information: no binding. */
OCRS_API ocrs_status ocrs_group_member_stats(const ocrs_engine_group* g, size_t m, uint64_t out[8]);
/* Host-only helpers behind that placement, exported for tests: a sysfs cpu list ("0-3,8") parsed into cpu numbers (cpus may
 * be NULL; *n_cpus = how many the list names); and: node of a PCI device under `sysfs_root` (NULL = "/sys"), the calling
 * thread bound to that node's CPUs inside a scope (*cpus_inside = CPUs in its mask there, -1 = not bound) and restored
 * (*cpus_after). */
OCRS_API ocrs_status ocrs_numa_parse_cpulist(const char* list, int32_t* cpus, size_t capacity, size_t* n_cpus);
OCRS_API ocrs_status ocrs_numa_bind_selftest(const char* sysfs_root, const char* pci_bus_id, int* node, int* cpus_inside, int* cpus_after);
/* Worker threads the group has created so far (they are kept between calls). */
OCRS_API ocrs_status ocrs_group_worker_threads(const ocrs_engine_group* g, size_t* n);
/* Transport of the group's most recent gather: 1 host, 2 RCCL (0: none yet), the payload bytes it moved, and — when it
 * used the host transport — why (valid until the next call of this function on the group; "" otherwise).  Any
 * argument may be NULL. */
OCRS_API ocrs_status ocrs_group_last_gather(const ocrs_engine_group* g, int* transport, size_t* bytes, const char** why_host);

/* ------------------------------------------------------------------------
 * Measurement hooks (bench.py; not part of the reference surface).
 * ---------------------------------------------------------------------- */
/* Device memory helpers so that bench inputs can be made HBM-resident without
 * torch in the loop.  free / upload take pointers of any device. */
OCRS_API ocrs_status ocrs_device_malloc(size_t bytes, void** d_ptr);                 /* on the default device */
OCRS_API ocrs_status ocrs_device_malloc_on(int device, size_t bytes, void** d_ptr);
OCRS_API ocrs_status ocrs_device_free(void* d_ptr);
OCRS_API ocrs_status ocrs_device_upload(void* d_dst, const void* h_src, size_t bytes);
OCRS_API ocrs_status ocrs_device_synchronize(void);
/* Page-locked host memory: ocrs_engine_prepare_input{,_batch} from such a buffer is a DMA transfer that overlaps
 * GPU work (from ordinary memory the runtime stages every copy through a bounce buffer).  An image decoder that
 * writes its output here hands pages over at PCIe speed. */
OCRS_API ocrs_status ocrs_host_malloc(size_t bytes, void** h_ptr);
OCRS_API ocrs_status ocrs_host_free(void* h_ptr);
/* Rates this device sustains: register-only fp32 MFMA loop (TFLOP/s) and a large float4 copy
 * (GB/s, read + write).  Reported by bench.py beside the nominal peaks the roofline uses. */
OCRS_API ocrs_status ocrs_device_measure_peaks(double* mfma_f32_tflops, double* hbm_copy_gbps);

/* Per-stage timers: when enabled, every GPU stage is bracketed by HIP events
 * on the stream it is launched on; ocrs_engine_stage_times returns accumulated
 * milliseconds and launch counts since the last reset.  Stage names:
 * ocrs_stage_name(i), i < ocrs_stage_count(). */
OCRS_API ocrs_status ocrs_engine_enable_timing(ocrs_engine* e, int enable); /* 0 off, 1 stages, 2 stages + kernels */
OCRS_API int ocrs_stage_count(void);
OCRS_API const char* ocrs_stage_name(int stage);
OCRS_API ocrs_status ocrs_engine_stage_times(ocrs_engine* e, double* ms, uint64_t* launches, int reset);

/* Kernel-class timers of the model executor (enable_timing(e, 2)): every launch
 * is bracketed by HIP events on its stream; flops/bytes are the ALGORITHMIC
 * figures of DESIGN.md §6 summed over the launches. */
/* Restrict per-launch kernel timing to the classes whose bit is set (default: all). */
OCRS_API ocrs_status ocrs_engine_set_kernel_timing_mask(ocrs_engine* e, uint32_t mask);
/* Request coalescing (ocrs_engine_params.coalesce): {merged batches run, caller requests they carried} per stage since the engine
 * was created.  requests > batches means calls of different host threads shared launches. */
OCRS_API ocrs_status ocrs_engine_coalesce_stats(const ocrs_engine* e, uint64_t detect[2], uint64_t recognize[2]);
OCRS_API int ocrs_kernel_class_count(void);
OCRS_API const char* ocrs_kernel_class_name(int cls);
OCRS_API ocrs_status ocrs_engine_kernel_stats(ocrs_engine* e, double* ms, uint64_t* launches, double* flops,
                                              double* bytes, int reset);
/* Per class, the part of `flops` that ran on the matrix cores (all of it for the gemm_*_mfma classes; the pointwise
 * convolutions and ConvTranspose of a fused detection block when its MFMA variant ran).  Call BEFORE a resetting
 * ocrs_engine_kernel_stats. */
OCRS_API ocrs_status ocrs_engine_kernel_mfma_flops(ocrs_engine* e, double* mfma_flops);

#ifdef __cplusplus
}
#endif
#endif /* OCRS_AMD_H */
