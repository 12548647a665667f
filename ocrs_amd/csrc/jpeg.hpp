// JPEG hand-off for OcrEngine::prepare_input (the reference decodes on the host with the `image`
// crate before prepare_input, ocrs-cli/src/main.rs:312-333).
//
// Split: the HOST does what is inherently serial — marker parsing and Huffman entropy decoding (baseline / extended
// sequential and progressive, jpeg_host.cpp) — and hands the GPU the quantised DCT coefficients in a sparse form
// (a 64-bit occupancy mask per 8x8 block + the non-zero values: ~0.7 bytes per pixel for a typical 4:2:0 page against
// 3 bytes per pixel of decoded RGB).  The GPU does everything that is per-sample arithmetic (kernels_jpeg.hip):
// dequantisation, the 8x8 inverse DCT, chroma upsampling, YCbCr -> RGB, and from there the engine's own
// prepare_image conversion.  The arithmetic is libjpeg's, operation for operation, so that the pixels equal what
// libjpeg / libjpeg-turbo (and with them PIL, the oracle's decoder) produce:
//   * IDCT: jidctint.c `jpeg_idct_islow` (CONST_BITS 13, PASS1_BITS 2, the twelve FIX_ constants, DESCALE rounding, the
//     post-IDCT range-limit table with its wrap-around behaviour);
//   * upsampling: jdsample.c h2v1 / h2v2 "fancy" (triangle) upsampling with its alternating +1/+2 and +7/+8 rounding and
//     edge replication; 1x1 chroma is copied;
//   * colour: jdcolor.c YCbCr -> RGB with 16-bit fixed-point tables (FIX(1.40200) ... ), ONE_HALF rounding, arithmetic shifts.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace ocrs {
namespace jpeg {

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;   // sampling factors, quantisation table
    int width = 0, height = 0;          // downsampled dimensions in samples: ceil(image * h / hmax), ceil(image * v / vmax)
    int blocks_w = 0, blocks_h = 0;     // block grid incl. the padding of interleaved MCUs (multiples of h / v)
    size_t first_block = 0;             // index of the component's first block in the coefficient arrays
};

struct Coefficients {
    int width = 0, height = 0;          // image
    int ncomp = 0, hmax = 1, vmax = 1;
    bool progressive = false;
    bool ycc = true;                    // 3 components: YCbCr (JFIF / Adobe transform 1) or RGB (Adobe transform 0, ids 'R','G','B')
    Component comp[3];
    uint16_t quant[4][64] = {};         // natural (row-major) order
    // sparse coefficients, blocks in component order, row-major inside a component:
    std::vector<uint64_t> mask;         // bit z set <=> the coefficient with ZIG-ZAG index z is non-zero
    std::vector<uint32_t> offset;       // [nblocks + 1] index of the block's first value (blocks may sit in `values` in any
                                        // order — the order they were decoded in; [nblocks] = values.size())
    std::vector<int16_t> values;        // the non-zero coefficients of a block in ascending zig-zag index
    static const uint8_t kZigzagOfNatural[64];   // natural position (row * 8 + col) -> zig-zag index
    size_t nblocks() const { return mask.size(); }
};

// Parses and entropy-decodes `data`; throws ocrs::Error (OCRS_ERR_IMAGE_SOURCE) on malformed or unsupported streams
// (arithmetic coding, lossless / hierarchical, 12-bit samples, 4 components, sampling other than 4:4:4 / 4:2:2 / 4:2:0 /
// grey) — the caller then decodes on the host as the reference does.
Coefficients decode_coefficients(const uint8_t* data, size_t len);

// One-line description ("1200x1600 progressive 4:4:4"), for errors and the CLI.
std::string describe(const Coefficients& c);

}  // namespace jpeg
}  // namespace ocrs
