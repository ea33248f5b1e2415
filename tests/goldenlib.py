"""Helpers to load tests/golden/*.npz (vectors generated from the real reference)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("asperity", "flexibility", "fluctuations", "loglogavgslope", "spectrum", "xi", "zeromode")
MODEL_CASES = ("g1d", "c1_512", "p2d", "g3d", "g2d_dist", "p2d_geo", "g2d_sig_geo")  # c1_512 = BASELINE configs[0]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def latent(z, prefix):
    return {k: np.asarray(z[f"{prefix}.{k}"]) for k in KEYS if f"{prefix}.{k}" in z.files}


def meta(z):
    d = z["meta.distances"]
    return dict(shape=tuple(int(i) for i in z["meta.shape"]),
                distances=None if np.isnan(d).any() else tuple(float(i) for i in d),
                kind=str(z["meta.kind"]), nonlin=(str(z["meta.nonlin"]) or None),
                n_samples=int(z["meta.n_samples"]), geo=bool(z["meta.geo"]), seed=int(z["meta.seed"]),
                sampling_limit=int(z["meta.sampling_limit"]))


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64 if not np.iscomplexobj(a) else np.complex128)
    b = np.asarray(b, dtype=a.dtype)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def lat_relerr(a, b):
    """max over keys of per-key relative error, scalars compared against the global scale."""
    scale = max(float(np.max(np.abs(b[k]))) for k in b)
    err = 0.0
    for k in b:
        e = float(np.max(np.abs(np.asarray(a[k], dtype=np.float64) - b[k])))
        den = float(np.max(np.abs(b[k])))
        err = max(err, e / max(den, 1e-3 * scale))
    return err
