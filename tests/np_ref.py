"""Independent numpy restatement of the Cube-CNN maths (test infrastructure).

Written from the reference's formulas with matrix operations instead of the
kernel-shaped loops of oracle/ccnn_oracle.c, so that the two can be checked
against each other (different summation order, np.tanh instead of the
2/(1+1/exp(2x))-1 spelling).  Agreement is therefore to rounding, not bitwise.
"""
import numpy as np

F, K, H, O = 20, 171, 100, 13


def dims(z):
    re = len(range(0, z - 19, 2))          # for (i = 0; i + 19 < z; i += 2)
    mre = (re - 1) // 2 + 1
    return re, mre


def unpack(params, z):
    re, mre = dims(z)
    pout = F * mre
    o = 0
    W0 = params[o:o + F * K].reshape(F, K); o += F * K
    b0 = params[o:o + F]; o += F
    W1 = params[o:o + pout * H].reshape(pout, H).T; o += pout * H      # W1[n, p] = flat[n + 100 p]
    b1 = params[o:o + H]; o += H
    W2 = params[o:o + H * O].reshape(H, O).T; o += H * O               # W2[o, n] = flat[o + 13 n]
    b2 = params[o:o + O]; o += O
    assert o == len(params)
    return W0, b0, W1, b1, W2, b2


def windows(cube, z):
    """[re, 171] view: window t = cube[18 t : 18 t + 171]."""
    re, _ = dims(z)
    return np.stack([cube[18 * t:18 * t + K] for t in range(re)])


def forward(params, cube, z, flip=False):
    re, mre = dims(z)
    W0, b0, W1, b1, W2, b2 = unpack(params, z)
    X = windows(cube, z)
    Wc = W0[:, ::-1] if flip else W0
    a0 = np.tanh(Wc @ X.T + b0[:, None])                               # [F, re]
    pad = np.full((F, 2 * mre), -np.inf, dtype=a0.dtype)
    pad[:, :re] = a0
    pairs = pad.reshape(F, mre, 2)
    arg = np.argmax(pairs, axis=2)                                     # first maximum wins
    pool = np.take_along_axis(pairs, arg[:, :, None], axis=2)[:, :, 0]
    pidx = (2 * np.arange(mre)[None, :] + arg) + re * np.arange(F)[:, None]
    a1 = np.tanh(W1 @ pool.ravel() + b1)
    e = np.exp(W2 @ a1 + b2)
    p = e / e.sum()
    return dict(a0=a0.ravel(), pool=pool.ravel(), pidx=pidx.ravel(), a1=a1, probs=p, X=X)


def grad(params, cube, onehot, z, flip=False):
    """Per-sample gradient in the flat parameter layout, with the reference's
    quirks: delta2 = (p - y)/13, conv grads divided by re, tanh' applied a second
    time on the conv-bias path, and (fp32 variant) a forward-only filter flip."""
    re, mre = dims(z)
    W0, b0, W1, b1, W2, b2 = unpack(params, z)
    f = forward(params, cube, z, flip)
    d2 = (f["probs"] - onehot) / O
    d1 = (W2.T @ d2) * (1 + f["a1"]) * (1 - f["a1"])
    dpool = (W1.T @ d1) * (1 + f["pool"]) * (1 - f["pool"])
    Delta = np.zeros(F * re, dtype=params.dtype)
    Delta[f["pidx"]] = dpool
    Delta = Delta.reshape(F, re)
    a0 = f["a0"].reshape(F, re)
    dW0 = (Delta @ f["X"]) / re
    dB0 = (Delta * (1 + a0) * (1 - a0)).sum(axis=1) / re
    dW1 = np.outer(f["pool"], d1)                                      # flat[n + 100 p]
    dW2 = np.outer(f["a1"], d2)                                        # flat[o + 13 n]
    g = np.concatenate([dW0.ravel(), dB0, dW1.ravel(), d1, dW2.ravel(), d2])
    return g.astype(params.dtype), f


def bce_loss(p, onehot):
    return -(onehot * np.log(p) + (1 - onehot) * np.log(1 - p)).sum() / O
