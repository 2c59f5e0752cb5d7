"""Shared helpers for the parity tests."""
import numpy as np


def canon_rows(r, score_col=0):
    """Rows sorted by (-score, remaining columns): removes the tie-order freedom the reference leaves open
    (numpy's unstable argsort()[::-1]) without hiding any other difference."""
    r = np.asarray(r)
    if r.shape[0] == 0:
        return r
    cols = [c for c in range(r.shape[1]) if c != score_col]
    keys = [r[:, c] for c in reversed(cols)] + [-r[:, score_col].astype(np.float64)]
    return r[np.lexsort(keys)]


def match_rois(got, ref, px_tol=1.0, score_tol=1e-3):
    """Fraction of `got` rows that have a partner in `ref` within px_tol on every coordinate and score_tol on the
    score (greedy one-to-one). Used where device and oracle start from head outputs that differ by fp32 rounding,
    so near-tied scores / IoUs at the 0.7 boundary may legitimately resolve differently."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    if got.shape[0] == 0:
        return 1.0 if ref.shape[0] == 0 else 0.0
    used = np.zeros(ref.shape[0], bool)
    hit = 0
    for g in got:
        d = np.abs(ref[:, 1:5] - g[1:5]).max(axis=1)
        ok = (d <= px_tol) & (np.abs(ref[:, 0] - g[0]) <= score_tol) & ~used
        if ok.any():
            used[np.argmax(ok)] = True
            hit += 1
    return hit / float(got.shape[0])


def match_lines(got, ref, px_tol=1.0, score_tol=1e-3):
    """Same number of lines and a one-to-one pairing (greedy nearest) within px_tol on all 8 coordinates and score_tol
    on the score. Not sort-based: two lines whose scores differ in the last ulp may legitimately come out in either
    order from the fp32 C++ and the numpy implementation."""
    got = np.asarray(got, np.float64).reshape(-1, 9)
    ref = np.asarray(ref, np.float64).reshape(-1, 9)
    if got.shape[0] != ref.shape[0]:
        return False
    used = np.zeros(ref.shape[0], bool)
    for g in got:
        ok = (np.abs(ref[:, :8] - g[:8]).max(axis=1) <= px_tol) & (np.abs(ref[:, 8] - g[8]) <= score_tol) & ~used
        if not ok.any():
            return False
        used[np.argmax(ok)] = True
    return True


# Fixtures whose rois are dominated by exactly tied scores (fp32-saturated 1.0: 630 of the 1000 rois of the 80 x 120 case).
# The reference orders ties by numpy's unstable argsort()[::-1] (implementation-defined), this build by ascending index
# (documented deviation): the SET of text lines is identical, their order follows the tie order -- compare those as sets.
TIE_HEAVY = {"s15_80x120"}


def lines_close(tag, got, want, tol):
    """(M,9) text-line records within tol; in input order, or as sets for the tie-heavy fixtures."""
    got, want = np.asarray(got, np.float64).reshape(-1, 9), np.asarray(want, np.float64).reshape(-1, 9)
    if got.shape != want.shape:
        return False
    if got.shape[0] == 0:
        return True
    if tag in TIE_HEAVY:
        return match_lines(got, want, tol, tol) if tol > 0 else np.array_equal(canon_rows(got, 8), canon_rows(want, 8))
    return np.array_equal(got, want) if tol == 0 else float(np.abs(got - want).max()) < tol
