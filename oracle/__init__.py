"""Test infrastructure (CPU restatement of the reference's hot path): imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline / parity legs -- never by the product (openglottal_amd/)."""
import os as _os


def reference_band() -> float:
    """The logit band inside which a mask pixel may differ from the reference: the largest difference the REFERENCE shows against
    itself on the 128 fixture frames under another oneDNN summation order (1 thread instead of 8: 2.4e-6; channels_last: 3.475e-5),
    captured by tests/golden/gen_golden.py into unet_full128_self_noise.npz.  Not a literal anywhere in this repository."""
    import numpy as _np

    here = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    band = float(_np.load(_os.path.join(here, "tests", "golden", "unet_full128_self_noise.npz"))["band"])
    assert 1e-6 < band < 1e-4, band
    return band
