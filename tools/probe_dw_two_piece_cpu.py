"""CPU emulation of a single-accumulator two-piece weight-gradient product (a design study, nothing here runs on the GPU):
dY split into fp16 hi + scaled lo under one scale per batch, activations into hi + unscaled lo + hi * 2^-11 under a static
scale, three piece products into one sum; errors against fp64 next to an fp32 accumulation in chunks of 16 points."""
import numpy as np
rng = np.random.default_rng(0)
def f16(x): return x.astype(np.float16).astype(np.float64)
def split_scaled(x):           # hi + lo_s * 2^-11
    hi = f16(x); lo = f16((x - hi) * 2048.0); return hi, lo
def emul(dY, X, xs=16.0):
    # global scale for dY: max -> 2^14
    s = 2.0 ** (14 - np.floor(np.log2(np.abs(dY).max())))
    a_hi, a_lo = split_scaled(dY * s)
    Xs = X * xs
    b_hi = f16(Xs); b_lo = f16(Xs - b_hi); b_hd = f16(b_hi / 2048.0)
    acc = a_hi.T @ b_hi + a_hi.T @ b_lo + a_lo.T @ b_hd      # exact-ish (float64) sums of exact piece products
    return acc / (s * xs)
def fp32_chunks(dY, X, k=16):   # fp32 accumulate in chunks of k (like an MFMA k-step), sequential over chunks
    acc = np.zeros((dY.shape[1], X.shape[1]), np.float32)
    for i in range(0, dY.shape[0], k):
        acc = (acc.astype(np.float64) + dY[i:i+k].astype(np.float64).T @ X[i:i+k].astype(np.float64)).astype(np.float32)
    return acc.astype(np.float64)
N, M, K = 65536, 48, 48
for name, tail, xscale in (("plain", 0.0, 1.0), ("heavy tail 3 decades", 3.0, 1.0), ("heavy tail 6 decades", 6.0, 1.0), ("small activations", 3.0, 0.02), ("large activations", 3.0, 50.0)):
    X = (np.maximum(rng.standard_normal((N, K)), 0) * xscale).astype(np.float32)
    X[:, :4] *= 1e-3                                   # a few nearly dead units
    mag = 10.0 ** (-tail * rng.random((N, 1)))
    dY = (rng.standard_normal((N, M)) * mag * 1e-4 * (rng.random((N, M)) < 0.5)).astype(np.float32)
    truth = dY.astype(np.float64).T @ X.astype(np.float64)
    e2 = np.abs(emul(dY.astype(np.float64), X.astype(np.float64)) - truth)
    e32 = np.abs(fp32_chunks(dY, X) - truth)
    scale = np.abs(truth).max()
    col = np.abs(truth).max(0)                          # per input unit (column of dW)
    print(f"{name:24s} max err/scale: 2-piece {e2.max()/scale:.2e}  fp32 {e32.max()/scale:.2e} | worst column-relative: 2-piece {(e2.max(0)/col).max():.2e} fp32 {(e32.max(0)/col).max():.2e}")
