"""CPU restatement of the operand split behind the bf16x3 weight-gradient kernel
(csrc/cfl_hip.hip split3): v = h + m + l exactly, every part exactly representable in bf16, and the six
retained partial products reproduce a*b to 2^-21 in the worst case (2^-24 rms)."""
import numpy as np


def split3(v):
    v = np.asarray(v, np.float32)
    mask = np.uint32(0xFFFF0000)
    h = (v.view(np.uint32) & mask).view(np.float32)
    r = (v - h).astype(np.float32)
    m = (r.view(np.uint32) & mask).view(np.float32)
    l = (r - m).astype(np.float32)
    return h, m, l


def is_bf16(x):
    return ((np.asarray(x, np.float32).view(np.uint32) & np.uint32(0xFFFF)) == 0).all()


def test_split_is_exact_and_bf16_representable():
    rng = np.random.RandomState(0)
    v = np.concatenate([rng.randn(200000).astype(np.float32) * 13,
                        np.abs(rng.randn(100000)).astype(np.float32) * 1e-3,
                        (rng.rand(100000).astype(np.float32) - 0.5) * 1e4,
                        np.array([0.0, 1.0, -1.0, 58.388599, 3.0e-30, -7.5e20], np.float32)])
    h, m, l = split3(v)
    assert is_bf16(h) and is_bf16(m) and is_bf16(l)
    # exact: the three parts add back to v bit for bit (summed smallest first, every partial sum exact)
    assert np.array_equal((l.astype(np.float64) + m + h).astype(np.float32), v)
    assert np.array_equal(l.astype(np.float64) + m.astype(np.float64) + h.astype(np.float64), v.astype(np.float64))
    nz = v != 0
    assert (np.abs(m[nz]) <= np.abs(v[nz]) * 2.0 ** -7).all()
    assert (np.abs(l[nz]) <= np.abs(v[nz]) * 2.0 ** -15).all()


def test_six_partial_products_match_the_fp32_product():
    rng = np.random.RandomState(1)
    a = (rng.randn(300000) * 5).astype(np.float32)
    b = (rng.randn(300000) * 0.05).astype(np.float32)
    ah, am, al = [t.astype(np.float64) for t in split3(a)]
    bh, bm, bl = [t.astype(np.float64) for t in split3(b)]
    kept = am * bm + al * bh + ah * bl + am * bh + ah * bm + ah * bh
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(kept - exact) / np.maximum(np.abs(exact), 1e-300)
    assert rel.max() <= 2.0 ** -21            # dropped: am*bl + al*bm + al*bl (worst case 2 * 2^-7 * 2^-15)
    assert np.sqrt((rel ** 2).mean()) <= 2.0 ** -23.5
    # eight terms (only al*bl dropped) reach 2^-30
    kept8 = kept + am * bl + al * bm
    assert (np.abs(kept8 - exact) / np.maximum(np.abs(exact), 1e-300)).max() <= 2.0 ** -29
