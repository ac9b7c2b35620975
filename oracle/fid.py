"""ORACLE (test infrastructure only; never imported by the product path): Fréchet distance restated independently.

Reference: fid_utils/fid.py:43-66.  For symmetric positive semi-definite C1, C2 the trace of (C1 C2)^(1/2) equals the sum of
the square roots of the eigenvalues of C1^(1/2) C2 C1^(1/2) (a symmetric PSD matrix), which needs no matrix square root of a
non-symmetric product -- a different route to the same number than the reference's scipy.linalg.sqrtm.
Pinned by tests/golden/fid.npz (outputs of the reference's own calc_fid, oracle/make_golden.py::golden_fid).
"""
import numpy as np


def frechet_distance(m1, c1, m2, c2):
    c1 = np.asarray(c1, np.float64); c2 = np.asarray(c2, np.float64)
    w, v = np.linalg.eigh((c1 + c1.T) / 2)
    root1 = (v * np.sqrt(np.clip(w, 0, None))) @ v.T
    inner = root1 @ c2 @ root1
    ev = np.linalg.eigvalsh((inner + inner.T) / 2)
    tr_root = np.sqrt(np.clip(ev, 0, None)).sum()
    d = np.asarray(m1, np.float64) - np.asarray(m2, np.float64)
    return float(d @ d + np.trace(c1) + np.trace(c2) - 2 * tr_root)


def batch_plan(n_sample, batch_size):
    """fid.py:22-27."""
    n_batch = n_sample // batch_size
    resid = n_sample - n_batch * batch_size
    return [batch_size] * n_batch + ([resid] if resid else [])
