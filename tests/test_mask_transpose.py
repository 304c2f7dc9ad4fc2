"""The binning's column / row masks by a bit-matrix transpose (csrc/k_raster.hip: rectWord / transposeWords, k_dbin_count).

Host emulation of the wave-level algorithm — a splat's rectangle as a row of the (64 splats x binsX + binsY bits) matrix, five
butterfly stages per half wave (partner = lane ^ d; keep-mask / rotate / merge exactly as the kernel's v_alignbit + v_bfi), the
upper half's word fetched by lane b — against the definition the ballot path implements: mask of column b = the lanes whose
rectangle spans column b.  The GPU side: test_binning_paths_bit_identical and the variants test (MGS_DB_TRANSPOSE=0)."""
import numpy as np

LOW = [0x55555555, 0x33333333, 0x0F0F0F0F, 0x00FF00FF, 0x0000FFFF]


def _rect_word(x0, y0, x1, y1, valid, bins_x, bins_y):
    dx, dy = x1 - x0, y1 - y0
    if not valid or dx < 0 or dy < 0:
        return 0
    cb = ((((2 << (dx & 31)) - 1) << (x0 & 31)) & 0xFFFFFFFF) & ((1 << bins_x) - 1)
    rb = ((((2 << (dy & 31)) - 1) << (y0 & 31)) & 0xFFFFFFFF) & ((1 << bins_y) - 1)
    return (cb | (rb << bins_x)) & 0xFFFFFFFF


def _rotr(v, s):
    s &= 31
    return ((v >> s) | (v << (32 - s))) & 0xFFFFFFFF if s else v


def _transpose_wave(x):
    x = list(x)
    for k in range(5):
        d = 1 << k
        nxt = []
        for lane in range(64):
            up = (lane >> k) & 1
            keep = (~LOW[k] & 0xFFFFFFFF) if up else LOW[k]
            rot = d if up else 32 - d
            p = x[(lane & 32) | ((lane & 31) ^ d)]  # partner inside the 32-lane half
            nxt.append((keep & x[lane]) | (~keep & 0xFFFFFFFF & _rotr(p, rot)))
        x = nxt
    return [(x[(lane + 32) & 63] << 32) | x[lane] for lane in range(64)]  # lane b: the word of lane 32 + b above its own


def test_masks_by_transpose_equal_the_ballot_definition():
    rng = np.random.default_rng(5)
    for bins_x, bins_y in ((8, 9), (15, 17), (15, 9), (1, 1), (31, 1), (1, 31), (16, 16), (3, 29)):
        for _ in range(20):
            rects = []
            for lane in range(64):
                x0, y0 = int(rng.integers(0, bins_x)), int(rng.integers(0, bins_y))
                x1 = int(rng.integers(x0, bins_x)) if rng.random() < 0.9 else x0 - 1  # some inverted rectangles: no bins
                y1 = int(rng.integers(y0, bins_y))
                rects.append((x0, y0, x1, y1, rng.random() < 0.95))
            words = [_rect_word(*r, bins_x, bins_y) for r in rects]
            got = _transpose_wave(words)
            for b in range(bins_x):
                want = sum(1 << l for l, (x0, y0, x1, y1, v) in enumerate(rects) if v and x1 >= x0 and y1 >= y0 and x0 <= b <= x1)
                assert got[b] == want, (bins_x, bins_y, "column", b)
            for b in range(bins_y):
                want = sum(1 << l for l, (x0, y0, x1, y1, v) in enumerate(rects) if v and x1 >= x0 and y1 >= y0 and y0 <= b <= y1)
                assert got[bins_x + b] == want, (bins_x, bins_y, "row", b)
