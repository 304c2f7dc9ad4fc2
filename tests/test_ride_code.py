"""The bin-rectangle codes that ride through the key sort (csrc/kernels_common.h: rideEncode / rideDecode).

Host restatement of the device arithmetic, exhaustive over every bin grid the direct binning accepts (binsX, binsY <= 32,
binsX * binsY <= 256): the branch-free decode — nested thresholds, y0 = r / wS through a 16-bit reciprocal — must invert the
encode for every coded rectangle (1x1, 2x1, 1x2, 2x2).  The GPU side of the same property: test_binning_paths_bit_identical and
the variants test (MGS_RECT_RIDE=0 looks the rectangles up by id instead)."""


def _base(shape, bx, by):
    nb = bx * by
    return [0, nb, nb + (bx - 1) * by, nb + (bx - 1) * by + bx * (by - 1)][shape]


def _decode(code, bx, by):
    b1, b2, b3 = _base(1, bx, by), _base(2, bx, by), _base(3, bx, by)
    c1, c2, c3 = code >= b1, code >= b2, code >= b3
    r = code - (b3 if c3 else (b2 if c2 else (b1 if c1 else 0)))
    dy = 1 if c2 else 0
    dx = 1 if (c3 or (c1 and not c2)) else 0
    w0, w1 = bx, max(bx - 1, 1)
    i0, i1 = (65536 + w0 - 1) // w0, (65536 + w1 - 1) // w1
    ws = w1 if dx else w0
    y0 = ((r * (i1 if dx else i0)) & 0xFFFFFFFF) >> 16
    x0 = r - y0 * ws
    return x0, y0, x0 + dx, y0 + dy


def test_ride_decode_inverts_encode_for_every_grid():
    checked = 0
    for bx in range(1, 33):
        for by in range(1, 33):
            if bx * by > 256:
                continue
            for shape in range(4):
                dx, dy = shape & 1, shape >> 1
                for y0 in range(by - dy):
                    for x0 in range(bx - dx):
                        code = _base(shape, bx, by) + y0 * (bx - dx) + x0
                        assert code < 1024
                        assert _decode(code, bx, by) == (x0, y0, x0 + dx, y0 + dy), (bx, by, shape, x0, y0)
                        checked += 1
    assert checked > 200000
