"""Global configuration (counterpart of reference nifty/config.py:42-80)."""
# sampling_rng: "numpy" = the reference's PCG64 + ziggurat streams, draw for draw.  Fields that live on a GPU are drawn
#               THERE from the host generator's state (nk_pcg64_normal: the same numbers -- bit-identical except in the ziggurat
#               tail |x| > 3.654, 2.7e-4 of the draws, where the device log1p may differ from the host libm by <= 4 ulp --,
#               generator left in the same state;
#               ~25 ms per 1e9 normals instead of ~10 s + the upload);
#               "numpy_host" = the same streams drawn by numpy on the host and uploaded (cross-check of the above);
#               "device" = the fused engine draws its N(0,1) fields with torch's device generator, seeded per sample
#               from the same SeedSequence tree (same statistics and mirrored-pair consistency, different numbers)
_config = dict(hartley_convention="non_canonical_hartley", sampling_rng="numpy")


def get(key):
    return _config[key]


def update(key, value, /):
    if not isinstance(key, str):
        raise TypeError(f"key must be a string; got {key!r}")
    key = key.lower()
    if key == "hartley_convention":
        if not isinstance(value, str):
            raise TypeError(f"value to {key!r} must be a string; got {value!r}")
        if value in ("ducc_hartley", "non_canonical_hartley"):
            value = "non_canonical_hartley"
        elif value in ("ducc_fht", "canonical_hartley"):
            value = "canonical_hartley"
        else:
            raise ValueError(f"invalid value to {key!r}; got {value!r}")
    elif key == "sampling_rng":
        if value not in ("numpy", "numpy_host", "device"):
            raise ValueError(f"invalid value to {key!r}; got {value!r}")
    else:
        raise ValueError(f"unknown configuration key {key!r}")
    _config[key] = value
