"""CPU restatement of the per-sample half of JPEG decoding (TEST INFRASTRUCTURE; only tests/ may import it).

The reference decodes image files on the host with the `image` crate before `prepare_input`
(ocrs-cli/src/main.rs:312-333).  The product splits that work: entropy decoding on the host, everything per-sample on
the GPU (ocrs_amd/csrc/kernels_jpeg.hip).  This module restates the per-sample half in numpy from the published
algorithms of the Independent JPEG Group's reference decoder — the decoder PIL and the `image` crate's consumers are
judged against — so that the GPU kernels have a checker that is not themselves:

  * idct_islow      jidctint.c `jpeg_idct_islow`: Loeffler-Ligtenberg-Moschytz 8-point IDCT, CONST_BITS 13,
                    PASS1_BITS 2, the twelve FIX constants, DESCALE = (x + 2^(n-1)) >> n, the post-IDCT range-limit
                    table of jdmaster.c `prepare_range_limit_table` incl. its wrap-around;
  * upsample_*      jdsample.c `h2v1_fancy_upsample` / `h2v2_fancy_upsample` (triangle filter, alternating rounding
                    constants 1/2 and 8/7, edges replicated; plain replication when the chroma plane is <= 2 wide);
  * ycc_to_rgb      jdcolor.c `build_ycc_rgb_table` / `ycc_rgb_convert` (16-bit fixed point, ONE_HALF, arithmetic shift).

Pinned by: PIL's decoder (libjpeg-turbo) on the reference's own JPEG (ocrs/examples/rust-book.jpg, progressive 4:4:4) and
on encodings of the synthetic pages and of the reference's PNGs in every sampling mode (tests/test_jpeg.py) — equal bytes.
"""
import numpy as np

FIX = dict(f0_298631336=2446, f0_390180644=3196, f0_541196100=4433, f0_765366865=6270, f0_899976223=7373, f1_175875602=9633,
           f1_501321110=12299, f1_847759065=15137, f1_961570560=16069, f2_053119869=16819, f2_562915447=20995, f3_072711026=25172)
CONST_BITS, PASS1_BITS = 13, 2


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct8(v, shift):
    """v: int64 [..., 8] -> [..., 8] (one 1-D pass of jpeg_idct_islow)."""
    F = FIX
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * F["f0_541196100"]
    tmp2 = z1 + z3 * (-F["f1_847759065"])
    tmp3 = z1 + z2 * F["f0_765366865"]
    z2, z3 = v[..., 0], v[..., 4]
    tmp0 = (z2 + z3) << CONST_BITS
    tmp1 = (z2 - z3) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F["f1_175875602"]
    tmp0 = tmp0 * F["f0_298631336"]
    tmp1 = tmp1 * F["f2_053119869"]
    tmp2 = tmp2 * F["f3_072711026"]
    tmp3 = tmp3 * F["f1_501321110"]
    z1 = z1 * (-F["f0_899976223"])
    z2 = z2 * (-F["f2_562915447"])
    z3 = z3 * (-F["f1_961570560"]) + z5
    z4 = z4 * (-F["f0_390180644"]) + z5
    tmp0 = tmp0 + z1 + z3
    tmp1 = tmp1 + z2 + z4
    tmp2 = tmp2 + z2 + z3
    tmp3 = tmp3 + z1 + z4
    out = np.stack([tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3], axis=-1)
    return _descale(out, shift)


def _range_limit_idct(x):
    i = x & 0x3FF
    return np.where(i < 128, i + 128, np.where(i < 512, 255, np.where(i < 896, 0, i - 896))).astype(np.uint8)


def idct_islow(coef, quant):
    """coef: int16 [nblocks, 64] (natural order), quant: [64] -> uint8 [nblocks, 8, 8]."""
    d = coef.astype(np.int64).reshape(-1, 8, 8) * quant.astype(np.int64).reshape(1, 8, 8)
    # pass 1: columns (the 1-D transform runs along axis 1); workspace is a C int
    ws = _idct8(np.swapaxes(d, 1, 2), CONST_BITS - PASS1_BITS)          # [b, col, row-out]
    ws = np.swapaxes(ws, 1, 2).astype(np.int32).astype(np.int64)          # [b, row, col]
    out = _idct8(ws, CONST_BITS + PASS1_BITS + 3)                        # rows
    return _range_limit_idct(out)


def plane_from_blocks(samples, blocks_w, blocks_h):
    return samples.reshape(blocks_h, blocks_w, 8, 8).transpose(0, 2, 1, 3).reshape(blocks_h * 8, blocks_w * 8)


def upsample_h2v1(p, out_w):
    """p: uint8 [h, cw] -> [h, out_w]."""
    cw = p.shape[1]
    q = p.astype(np.int32)
    if cw <= 2:
        return np.repeat(p, 2, axis=1)[:, :out_w]
    left = np.concatenate([q[:, :1], q[:, :-1]], axis=1)
    right = np.concatenate([q[:, 1:], q[:, -1:]], axis=1)
    even = (q * 3 + left + 1) >> 2
    odd = (q * 3 + right + 2) >> 2
    even[:, 0] = q[:, 0]
    odd[:, -1] = q[:, -1]
    out = np.empty((p.shape[0], 2 * cw), np.int32)
    out[:, 0::2], out[:, 1::2] = even, odd
    return out[:, :out_w].astype(np.uint8)


def upsample_h2v2(p, out_h, out_w):
    ch, cw = p.shape
    if cw <= 2:
        return np.repeat(np.repeat(p, 2, axis=0), 2, axis=1)[:out_h, :out_w]
    q = p.astype(np.int32)
    above = np.concatenate([q[:1], q[:-1]], axis=0)
    below = np.concatenate([q[1:], q[-1:]], axis=0)
    out = np.empty((2 * ch, 2 * cw), np.int32)
    for v, far in ((0, above), (1, below)):
        col = q * 3 + far                                   # thiscolsum
        left = np.concatenate([col[:, :1], col[:, :-1]], axis=1)
        right = np.concatenate([col[:, 1:], col[:, -1:]], axis=1)
        even = (col * 3 + left + 8) >> 4
        odd = (col * 3 + right + 7) >> 4
        even[:, 0] = (col[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (col[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2], out[v::2, 1::2] = even, odd
    return out[:out_h, :out_w].astype(np.uint8)


def ycc_to_rgb(y, cb, cr):
    y = y.astype(np.int32)
    cb = cb.astype(np.int32) - 128
    cr = cr.astype(np.int32) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 + -46802 * cr) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def decode_from_coefficients(geom, quant, coef):
    """geom / quant / coef as ocrs_jpeg_coefficients returns them -> RGB8 [H, W, 3] (grey: R = G = B, as into_rgb8)."""
    width, height, ncomp, hmax, vmax, _prog, ycc = [int(v) for v in geom[:7]]
    planes, first = [], 0
    for i in range(ncomp):
        _h, _v, tq, cw, ch, bw, bh = [int(v) for v in geom[7 + 7 * i: 14 + 7 * i]]
        blocks = coef[first: first + bw * bh]
        first += bw * bh
        planes.append(plane_from_blocks(idct_islow(blocks, quant[tq * 64:(tq + 1) * 64]), bw, bh)[:ch, :cw])
    if ncomp == 1:
        return np.repeat(planes[0][:height, :width, None], 3, axis=2)
    y = planes[0][:height, :width]
    if hmax == 1:
        c1, c2 = planes[1][:height, :width], planes[2][:height, :width]
    elif vmax == 1:
        c1, c2 = upsample_h2v1(planes[1], width)[:height], upsample_h2v1(planes[2], width)[:height]
    else:
        c1, c2 = upsample_h2v2(planes[1], height, width), upsample_h2v2(planes[2], height, width)
    if ycc:
        return ycc_to_rgb(y, c1, c2)
    return np.stack([y, c1, c2], axis=-1)
