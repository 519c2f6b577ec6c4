"""The arithmetic of the large-graph training kernels' products (csrc/kernels_train_big.hpp: k_train_fwd_b6, k_train_bwd_dx_b6), restated in
NumPy: every float32 operand is split into three bfloat16 terms by round-to-nearest of the remainder, and a product x w is taken as the six
largest of the nine term products, each exact in float32.  Checked here (no GPU): the split is exact, the dropped terms are O(2^-24 |x w|), a
TRUNCATING split would bias every product the same way (which BatchNormalization's gradients add up coherently - DESIGN.md 6b), and a
k = 160 dot product built this way is as close to float64 as a float32 fma chain is."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32"""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def bf16_trunc(x):
    return (x.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(x, rnd):
    hi = rnd(x); r1 = (x - hi).astype(np.float32)
    mid = rnd(r1); r2 = (r1 - mid).astype(np.float32)
    return hi, mid, rnd(r2)


def six_products(x, w, rnd):
    xh, xm, xl = split3(x, rnd); wh, wm, wl = split3(w, rnd)
    f = np.float64       # (each bf16 x bf16 product has 16 significant bits: exact in float32; summed here in float64 to isolate the dropped terms)
    return wl.astype(f) * xh + wh.astype(f) * xl + wm.astype(f) * xm + wm.astype(f) * xh + wh.astype(f) * xm + wh.astype(f) * xh


def test_three_term_split_is_exact_and_products_are_f32_accurate():
    rng = np.random.default_rng(0)
    x = (rng.normal(size=200_000) * np.exp(rng.uniform(-20, 20, 200_000))).astype(np.float32)
    w = (rng.normal(size=200_000) * np.exp(rng.uniform(-20, 20, 200_000))).astype(np.float32)
    for rnd in (bf16_rne, bf16_trunc):
        hi, mid, lo = split3(x, rnd)
        assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    exact = x.astype(np.float64) * w.astype(np.float64)
    rel_rne = (six_products(x, w, bf16_rne) - exact) / exact            # (signed by the product: < 0 = short of it)
    rel_trunc = (six_products(x, w, bf16_trunc) - exact) / exact
    assert np.abs(rel_rne).max() < 2.0 ** -24                            # measured 4.6e-8: below half an ulp of the product
    assert np.abs(rel_trunc).max() < 8 * 2.0 ** -24                      # measured 4.0e-7: truncated remainders are twice as long
    # round-to-nearest: the dropped terms have no preferred sign; truncation: every product comes out SHORT (a bias, not noise)
    assert abs(rel_rne.mean()) < 1e-9
    assert rel_trunc.max() <= 0.0 and rel_trunc.mean() < -2e-8


def test_dot_products_of_the_c4_layer_width_match_an_f32_chain():
    rng = np.random.default_rng(1)
    K, M = 160, 4000
    X = rng.normal(0.3, 1.0, (M, K)).astype(np.float32); W = (0.3 * rng.normal(size=K)).astype(np.float32)
    ref = X.astype(np.float64) @ W.astype(np.float64)
    chain = np.zeros(M, np.float32)
    for k in range(K): chain = (chain.astype(np.float64) + X[:, k].astype(np.float64) * W[k]).astype(np.float32)      # one rounding per step (an fma chain)
    acc = np.zeros(M, np.float32)
    for k0 in range(0, K, 32):                          # v_mfma_f32_16x16x32_bf16: 32 products per instruction, float32 accumulator between them
        blk = six_products(X[:, k0:k0 + 32], W[None, k0:k0 + 32], bf16_rne).sum(axis=1)
        acc = (acc.astype(np.float64) + blk).astype(np.float32)
    scale = np.abs(X.astype(np.float64)) @ np.abs(W.astype(np.float64))
    e_chain, e_b6 = np.abs(chain - ref) / scale, np.abs(acc - ref) / scale
    assert e_b6.max() <= 1.5e-7 and e_b6.mean() <= e_chain.mean()
