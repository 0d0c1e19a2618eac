"""The parity-major pixel order of stride-2 transposed launches (csrc/btx_contract_dma.h `par_major`, btx_epilogue.h `par_major_pixel`,
btx_api.hip): restated in Python and checked against the gather rule it shortcuts — no GPU.  (The GPU parity tests of the kernel are
tests/test_gpu_backward.py, the stride-2 cases of CASES.)"""
import itertools


def par_major_pixel(L, mqp, hh, wh, ho, wo):
    """logical pixel -> (class, output pixel raster index)   [btx_epilogue.h par_major_pixel]"""
    cls, q = divmod(L, mqp)
    t, b = divmod(q, wh)
    nb, a = divmod(t, hh)
    return cls, (nb * ho + 2 * a + (cls >> 1)) * wo + 2 * b + (cls & 1)


def tapmask(cls, kh_, kw_, ph, pw, dh, dw):
    """taps a tile of class `cls` walks   [btx_contract_dma.h]"""
    pi, pj = cls >> 1, cls & 1
    return {(kh, kw) for kh in range(kh_) for kw in range(kw_) if ((pi + ph + kh * dh) | (pj + pw + kw * dw)) & 1 == 0}


def gather_taps(oh, ow, kh_, kw_, ph, pw, dh, dw, H, W):
    """taps that reach output pixel (oh, ow) of the transposed stride-2 op under the per-pixel rule of the kernel"""
    out = set()
    for kh in range(kh_):
        for kw in range(kw_):
            th, tw = oh + ph - kh * dh, ow + pw - kw * dw
            if th >= 0 and tw >= 0 and th % 2 == 0 and tw % 2 == 0 and th // 2 < H and tw // 2 < W:
                out.add((kh, kw))
    return out


def test_parity_major_order_is_a_bijection_and_its_tap_rule_covers_the_gather_rule():
    tp = 256
    for (nb, H, W, k, p, d, opad) in [(3, 13, 13, 3, 1, 1, 1), (2, 7, 7, 3, 1, 1, 1), (2, 18, 14, 5, 2, 1, 1), (1, 9, 11, 1, 0, 1, 1),
                                      (2, 8, 8, 3, 2, 2, 1), (1, 6, 10, 4, 1, 1, 0)]:
        ho, wo = (H - 1) * 2 - 2 * p + d * (k - 1) + opad + 1, (W - 1) * 2 - 2 * p + d * (k - 1) + opad + 1
        if ho % 2 or wo % 2:
            continue  # the host takes the raster order there
        hh, wh = ho // 2, wo // 2
        mq = nb * hh * wh
        mqp = (mq + tp - 1) // tp * tp
        seen = set()
        for cls in range(4):
            mask = tapmask(cls, k, k, p, p, d, d)
            union = set()
            for q in range(mq):
                c2, pix = par_major_pixel(cls * mqp + q, mqp, hh, wh, ho, wo)
                assert c2 == cls and pix not in seen
                seen.add(pix)
                t, ow = divmod(pix, wo)
                oh = t % ho
                assert (oh & 1, ow & 1) == (cls >> 1, cls & 1)
                taps = gather_taps(oh, ow, k, k, p, p, d, d, H, W)
                assert taps <= mask, (cls, oh, ow)     # no tap of the pixel is skipped
                union |= taps
            if min(hh, wh) >= k:
                assert union == mask                    # and away from tiny images every walked tap is used by some pixel
        assert seen == set(range(nb * ho * wo))        # every output pixel exactly once
        # a 3x3 / stride-2 / padding-1 layer: 1, 2, 2 and 4 of the 9 taps
        if (k, p, d) == (3, 1, 1):
            assert sorted(len(tapmask(c, 3, 3, 1, 1, 1, 1)) for c in range(4)) == [1, 2, 2, 4]


def test_tiles_are_class_pure_and_class_minor():
    """workgroup t: class t & 3, tile t >> 2 of that class; a tile never holds two classes (classes are padded to whole tiles)"""
    tp, mq = 256, 3136
    mqp = (mq + tp - 1) // tp * tp
    tiles = 4 * mqp // tp
    covered = set()
    for t in range(tiles):
        cls, q0 = t & 3, (t >> 2) * tp
        nvalid = min(tp, mq - q0)
        assert nvalid >= 1
        for pl in range(nvalid):
            L = cls * mqp + q0 + pl
            assert L // mqp == cls
            covered.add(L)
    assert covered == {c * mqp + q for c, q in itertools.product(range(4), range(mq))}
