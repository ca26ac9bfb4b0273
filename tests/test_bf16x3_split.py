"""CPU restatement of the operand splits behind the bf16x3 kernels (csrc/cfl_hip.hip):

* split3 / split_frag -- truncation (h = v & 0xffff0000, ...): the operands split inside the loops of the weight-gradient
  and projection kernels;
* split_pair_rne -- round to nearest even: the kept planes of theta (cfl_wplanes_kernel, the fused Adam tails).

Both are exact (v = h + m + l, every part representable in bf16).  Six retained partial products reproduce a*b to 2^-21
(both operands truncated) / 2^-22 (one operand rounded: the projection on kept planes) in the worst case, and with one
rounded operand the dropped terms are zero-mean; eight of nine reach 2^-29."""
import numpy as np


def split3_trunc(v):
    v = np.asarray(v, np.float32)
    mask = np.uint32(0xFFFF0000)
    h = (v.view(np.uint32) & mask).view(np.float32)
    r = (v - h).astype(np.float32)
    m = (r.view(np.uint32) & mask).view(np.float32)
    l = (r - m).astype(np.float32)
    return h, m, l


def bf16_rne(x):
    """float32 -> nearest bf16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 does)"""
    b = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + 0x7fff + ((b >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3_rne(v):
    v = np.asarray(v, np.float32)
    h = bf16_rne(v)
    r = (v - h).astype(np.float32)
    m = bf16_rne(r)
    l = (r - m).astype(np.float32)
    return h, m, l


def is_bf16(x):
    return ((np.asarray(x, np.float32).view(np.uint32) & np.uint32(0xFFFF)) == 0).all()


def _values():
    rng = np.random.RandomState(0)
    return np.concatenate([rng.randn(200000).astype(np.float32) * 13,
                           np.abs(rng.randn(100000)).astype(np.float32) * 1e-3,
                           (rng.rand(100000).astype(np.float32) - 0.5) * 1e4,
                           np.array([0.0, 1.0, -1.0, 58.388599, 3.0e-30, -7.5e20, 1.00390625, 255.5, 0.998046875], np.float32)])


def test_splits_are_exact_and_bf16_representable():
    v = _values()
    nz = v != 0
    for split, mb, lb in ((split3_trunc, 2.0 ** -7, 2.0 ** -15), (split3_rne, 2.0 ** -8, 2.0 ** -16)):
        h, m, l = split(v)
        assert is_bf16(h) and is_bf16(m) and is_bf16(l)
        # exact: both subtractions are exact in fp32 and the three parts add back to v bit for bit
        assert np.array_equal(v.astype(np.float64) - h.astype(np.float64), (v - h).astype(np.float64))
        assert np.array_equal(l.astype(np.float64) + m.astype(np.float64) + h.astype(np.float64), v.astype(np.float64))
        assert (np.abs(m[nz]) <= np.abs(v[nz]) * mb).all()
        assert (np.abs(l[nz]) <= np.abs(v[nz]) * lb).all()


def test_six_partial_products_match_the_fp32_product():
    rng = np.random.RandomState(1)
    a = (np.abs(rng.randn(300000)) * 5).astype(np.float32)        # one-signed, like post-ReLU features
    b = (rng.randn(300000) * 0.05).astype(np.float32)
    exact = a.astype(np.float64) * b.astype(np.float64)
    for sa, sb, worst, zero_mean in ((split3_trunc, split3_trunc, 2.0 ** -21, False),     # weight gradient: x and dL/dy
                                     (split3_trunc, split3_rne, 2.0 ** -22, True)):       # projection: x and kept planes
        ah, am, al = [t.astype(np.float64) for t in sa(a)]
        bh, bm, bl = [t.astype(np.float64) for t in sb(b)]
        kept = am * bm + al * bh + ah * bl + am * bh + ah * bm + ah * bh
        rel = (kept - exact) / np.maximum(np.abs(exact), 1e-300)
        assert np.abs(rel).max() <= worst            # dropped: am*bl + al*bm + al*bl
        assert np.sqrt((rel ** 2).mean()) <= 2.0 ** -23.5
        if zero_mean:
            # a rounded operand makes the dropped terms zero-mean given the other operand: over a long contraction they cancel
            bpos = np.abs(b).astype(np.float32)
            bh, bm, bl = [t.astype(np.float64) for t in sb(bpos)]
            kept = am * bm + al * bh + ah * bl + am * bh + ah * bm + ah * bh
            ex = a.astype(np.float64) * bpos.astype(np.float64)
            assert abs((kept - ex).sum()) <= 2.0 ** -27 * ex.sum()
        # eight terms (only al*bl dropped) reach 2^-29
        kept8 = kept if zero_mean else kept + am * bl + al * bm
        if not zero_mean:
            assert (np.abs(kept8 - exact) / np.maximum(np.abs(exact), 1e-300)).max() <= 2.0 ** -29
