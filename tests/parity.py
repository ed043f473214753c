"""Shared parity bars of the GPU tests (test infrastructure).

Bars, per PARAMETER GROUP (W0, b0, W1, b1, W2, b2 - never over the whole flat vector, where a
large W1 entry would hide an error in a bias): max |gpu - oracle| <= tol * max |oracle| of the
group; tol = 1e-12 in fp64, 1e-5 in fp32.

The fp32 near-tie rule.  maxpooling picks the first of two neighbouring conv outputs unless the
second is strictly larger (gcn_mbgd_fp32.cu:251-285).  The GPU sums the 171 taps of a conv output
in another order than the reference's serial loop, so two outputs that agree to within that
rounding can compare the other way; the sample's delta then lands on the neighbouring position and
that ONE filter's gradient moves by more than 1e-5 although nothing is wrong.  The rule, never a
wider bar: every group, and every filter of W0 that has no such pair in the batch, must meet 1e-5
against the oracle; a filter that misses it must have a pair whose oracle conv outputs differ by
at most TIE_ABS, and the whole step must meet 1e-5 against the oracle re-run with a subset of
exactly those pairs decided the other way.
"""
import itertools

import numpy as np

import oracle as orc

TOL = {np.float64: 1e-12, np.float32: 1e-5, np.dtype("float64"): 1e-12, np.dtype("float32"): 1e-5}
TIE_ABS = 4e-6          # 32 ulp of 1.0 in fp32: the size of a 171-term sum's reordering error
MAX_TIES_SEARCHED = 8


def groups(d):
    return {"W0": (d.off_w0, d.off_b0), "b0": (d.off_b0, d.off_w1), "W1": (d.off_w1, d.off_b1),
            "b1": (d.off_b1, d.off_w2), "W2": (d.off_w2, d.off_b2), "b2": (d.off_b2, d.nparams)}


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.abs(b).max() if b.size else 0.0
    if den == 0.0:
        return 0.0 if not np.abs(a).any() else np.inf     # an all-zero group (fp32 conv bias) must match exactly
    return float(np.abs(a - b).max() / den)


def group_errs(out, ref, d):
    return {k: relerr(out[lo:hi], ref[lo:hi]) for k, (lo, hi) in groups(d).items()}


def assert_groups(out, ref, d, tol, what=""):
    errs = group_errs(np.asarray(out), np.asarray(ref), d)
    bad = {k: e for k, e in errs.items() if not e <= tol}
    assert not bad, f"{what}: groups over {tol:g}: {bad} (all: {errs})"
    return errs


def filter_errs(out, ref, d):
    """Relative error of each of the 20 filters of W0, against the largest weight of W0."""
    w_out = np.asarray(out[d.off_w0:d.off_b0], dtype=np.float64).reshape(orc.FILTERS, orc.FILTER_SIZE)
    w_ref = np.asarray(ref[d.off_w0:d.off_b0], dtype=np.float64).reshape(orc.FILTERS, orc.FILTER_SIZE)
    return np.abs(w_out - w_ref).max(axis=1) / np.abs(w_ref).max()


def near_ties(a0, d):
    """(sample, filter, pooled index) of every pool pair whose two conv outputs differ by <= TIE_ABS."""
    B = a0.shape[0]
    a = a0.reshape(B, orc.FILTERS, d.re)
    npair = d.re // 2                      # a trailing single position has no partner
    pr = a[:, :, :2 * npair].reshape(B, orc.FILTERS, npair, 2)
    gap = np.abs(pr[..., 0].astype(np.float64) - pr[..., 1].astype(np.float64))
    return [tuple(int(v) for v in t) for t in np.argwhere(gap <= TIE_ABS)]


def check_fp32_step(p_before, p_gpu, lv_gpu, cubes_h, onehot_h, first, B, z, lr):
    """One fp32 MBGD step of the GPU (from p_before to p_gpu) against the oracle, by the rule in
    this module's header.  Returns (oracle's parameters after the step, number of pairs that had
    to be decided the other way - 0 for an ordinary step)."""
    d = orc.dims(z)
    tol = 1e-5
    p_ref = p_before.copy()
    lv_ref, a0 = orc.mbgd_step_f32_ex(p_ref, cubes_h, onehot_h, first, B, z, lr, want_a0=True)
    assert relerr(lv_gpu, lv_ref) <= tol, f"step at {first}: loss {relerr(lv_gpu, lv_ref):.2e}"
    errs = group_errs(p_gpu, p_ref, d)
    ferr = filter_errs(p_gpu, p_ref, d)
    other_bad = {k: e for k, e in errs.items() if k != "W0" and not e <= tol}
    assert not other_bad, f"step at {first}: groups over {tol:g}: {other_bad}"
    bad_f = [int(f) for f in np.nonzero(~(ferr <= tol))[0]]
    if not bad_f:
        return p_ref, 0
    ties = near_ties(a0, d)
    tie_filters = {f for _, f, _ in ties}
    assert set(bad_f) <= tie_filters, (f"step at {first}: filters {bad_f} miss 1e-5 ({ferr[bad_f]}) and have no pool pair "
                                       f"within {TIE_ABS:g} (filters with such pairs: {sorted(tie_filters)})")
    cand = [t for t in ties if t[1] in bad_f]
    assert len(cand) <= MAX_TIES_SEARCHED, f"step at {first}: {len(cand)} near-tie pairs, too many to search"
    for r in range(1, len(cand) + 1):
        for sub in itertools.combinations(cand, r):
            flip = np.zeros((B, d.pout), dtype=np.int32)
            for b, f, m in sub:
                flip[b, f * d.mre + m] = 1
            p_alt = p_before.copy()
            orc.mbgd_step_f32_ex(p_alt, cubes_h, onehot_h, first, B, z, lr, flip=flip)
            if all(e <= tol for e in group_errs(p_gpu, p_alt, d).values()):
                return p_ref, r
    raise AssertionError(f"step at {first}: filters {bad_f} miss 1e-5 ({ferr[bad_f]}) and no decision of the near-tie pairs "
                         f"{cand} reproduces the GPU's step")
