"""CPU: the arithmetic claims behind the "f32x3" precision (include/dvsg_amd.h, conv_gemm_tile.h X3), checked with NumPy
on the host -- no GPU, no library call.  bfloat16 is emulated on float32 bit patterns (round to nearest even on the upper
16 bits, exactly what v_cvt_pk_bf16_f32 and the host packer of locnet.hip do)."""
import numpy as np


def bf16(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def split3(x):
    p1 = bf16(x)
    r1 = (x - p1).astype(np.float32)
    p2 = bf16(r1)
    r2 = (r1 - p2).astype(np.float32)
    p3 = bf16(r2)
    return p1, p2, p3, r1, r2


def samples():
    rng = np.random.default_rng(7)
    mant = rng.integers(0, 1 << 23, 400000).astype(np.uint32)
    expo = rng.integers(20, 235, 400000).astype(np.uint32)        # 2^-107 .. 2^107
    sign = rng.integers(0, 2, 400000).astype(np.uint32)
    x = ((sign << np.uint32(31)) | (expo << np.uint32(23)) | mant).view(np.float32)
    edge = np.array([0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 2.0 - 2.0 ** -23, 255.0, 1e-3, 3.0e-20, -7.7e19,
                     np.float32(1.00390625), np.float32(1.0039062), np.float32(0.99609375)], dtype=np.float32)
    worst = ((np.uint32(127) << np.uint32(23)) | np.arange(0, 1 << 23, 997, dtype=np.uint32)).view(np.float32)   # a sweep of mantissas
    return np.concatenate([x, edge, worst])


def test_three_bfloat16_pieces_hold_every_float32_exactly():
    x = samples()
    p1, p2, p3, r1, r2 = split3(x)
    # the residuals are exact float32 subtractions (checked in float64) and the third piece leaves nothing
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - p1.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - p2.astype(np.float64))
    assert np.array_equal(p3, r2)
    total = p1.astype(np.float64) + p2.astype(np.float64) + p3.astype(np.float64)
    assert np.array_equal(total, x.astype(np.float64))
    nz = x != 0
    assert (np.abs(p2[nz]) <= np.abs(x[nz]) * 2.0 ** -8).all() and (np.abs(p3[nz]) <= np.abs(x[nz]) * 2.0 ** -16).all()


def test_piece_products_are_exact_in_float32_and_six_terms_miss_at_most_one_multiply_rounding():
    rng = np.random.default_rng(11)
    a = samples()[:300000]
    w = rng.permutation(samples())[:300000]
    keep = (np.abs(a.astype(np.float64) * w.astype(np.float64)) < 1e30) & (np.abs(a.astype(np.float64) * w.astype(np.float64)) > 1e-30)
    a, w = a[keep], w[keep]
    A, W = split3(a)[:3], split3(w)[:3]
    exact = a.astype(np.float64) * w.astype(np.float64)
    six = np.zeros_like(exact)
    for i, j in ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)):
        prod32 = (A[i] * W[j]).astype(np.float32)                     # 8-bit x 8-bit significands: exact in float32
        assert np.array_equal(prod32.astype(np.float64), A[i].astype(np.float64) * W[j].astype(np.float64))
        six += prod32.astype(np.float64)
    miss = np.abs(six - exact)
    nz = exact != 0
    rel = miss[nz] / np.abs(exact[nz])
    assert rel.max() <= 2.0 ** -23                    # the dropped a2 w3 + a3 w2 + a3 w3
    assert np.median(rel) <= 2.0 ** -27
    # the float32 product the reference's unfused multiply forms is off by up to 2^-24 itself
    fl = np.abs((a * w).astype(np.float32).astype(np.float64) - exact)[nz] / np.abs(exact[nz])
    assert fl.max() <= 2.0 ** -24 and rel.mean() < fl.mean()


def test_a_dot_product_from_six_terms_is_as_close_to_float64_as_a_float32_fma_chain():
    """K-long dot products of ReLU-like activations and He-scaled weights, accumulated the way the kernels do: the exact
    path one fused multiply-add per k in float32, f32x3 per 16 k one float32-rounded sum of the large terms and one of the
    five small ones into separate accumulators (the MFMA's internal sum is wider than float32; float64 here), joined at the end."""
    rng = np.random.default_rng(5)
    M, K = 4096, 1152
    a = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    w = (rng.standard_normal(K) * np.sqrt(2.0 / K)).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    acc = np.zeros(M, dtype=np.float32)
    for k in range(K):       # float32 FMA chain: exact product, one rounding per step
        acc = (acc.astype(np.float64) + a[:, k].astype(np.float64) * np.float64(w[k])).astype(np.float32)
    A, W = split3(a)[:3], split3(w)[:3]
    big = np.zeros(M, dtype=np.float32)
    small = np.zeros(M, dtype=np.float32)
    for k0 in range(0, K, 16):
        s = slice(k0, k0 + 16)
        big = (big.astype(np.float64) + (A[0][:, s].astype(np.float64) * W[0][s].astype(np.float64)).sum(1)).astype(np.float32)
        for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1)):
            small = (small.astype(np.float64) + (A[i][:, s].astype(np.float64) * W[j][s].astype(np.float64)).sum(1)).astype(np.float32)
    x3 = (big + small).astype(np.float32)
    e32 = np.abs(acc - ref)
    e3 = np.abs(x3 - ref)
    assert np.sqrt((e3 ** 2).mean()) <= np.sqrt((e32 ** 2).mean()) and e3.max() <= 1.25 * e32.max()
