"""How far two CORRECT evaluations of the block transform can disagree after binning -- the noise floor every
comparison with an FFTW-backed reference has to be read against ("make parity honest").

The two evaluations are the oracle's flows: the pinned fast flow (what the GPU computes, bit for bit) and the
definition-order DFT (the published contract of fftw_plan_dft_1d).  Per dtype x error bound, on the C2 stand-in (smooth
fp32 field + 1 % noise) and a C3-like fp64 volume:
  * every coefficient pair is within the rounding-noise bound of tests/noise.py, and every differing bin id is
    explained by that noise (an edge or the range limit within reach) -- asserted;
  * the FRACTION of bin ids that differ is measured and held under a ceiling per cell (the table below; DESIGN.md
    section 5 quotes the measured values).
Measured (seeded inputs, so deterministic): fp64 -- no bin id differs at any bound (noise 1e-15 against bin widths
>= 2e-6).  fp32 (noise of a coefficient = a few 1e-7 x ||block||): eb 1e-3: 1.3e-5 of the bin ids differ, 1e-4: 1.1e-4,
1e-5: 1.1e-3 of the in-range ones (80 % of the coefficients are stored exactly there), 1e-6: 1.2 % of the in-range ones
(97 % stored exactly), and the exception COUNT itself moves by one.  So for fp32 "bin ids equal to the reference's" is
not a property any independent transform can have below eb = 1e-3, and north_star's "max |err| / PSNR equal to the
reference within 1e-6 relative" is checked where it is attainable: PSNR (an aggregate) everywhere except fp32 at
eb <= 1e-5, max |err| only through the bound itself."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import workloads as W
from tests.noise import classify_flips

# ceilings on the fraction of differing bin ids (measured values in the comment; seeded inputs, deterministic)
CEIL = {                    # about 1.5 x the measured value: a change of the fast flow's numerics of that size must show
    ("f64", 1e-3): 1e-6,    # 0
    ("f64", 1e-4): 1e-6,    # 0
    ("f64", 1e-5): 1e-6,    # 0
    ("f64", 1e-6): 2e-6,    # 0
    ("f32", 1e-3): 1.9e-5,  # 1.25e-5
    ("f32", 1e-4): 1.65e-4, # 1.08e-4
    ("f32", 1e-5): 3.0e-4,  # 1.98e-4 of all = 1.06e-3 of the in-range bin ids
    ("f32", 1e-6): 3.4e-4,  # 2.24e-4 of all = 1.18e-2 of the in-range bin ids
}


def _inputs(dtype):
    if dtype == np.float32:
        return W.c2()[:64 * 20000]                   # 1.28 M elements of the CESM-sized stand-in, sf = 1
    return W.c3(96)                                  # 0.88 M elements of the C3 formula, sf = 10


@pytest.mark.parametrize("eb", [1e-3, 1e-4, 1e-5, 1e-6])
@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_flip_rate_between_two_correct_transforms(dtype, eb, record_property):
    x = _inputs(dtype)
    a = O.compress(x, eb, O.EC, O.FAST, want_coef=True)
    b = O.compress(x, eb, O.EC, O.NAIVE, want_coef=True)
    flips, illegal = classify_flips(a, b, eb)
    assert illegal == 0
    rate = flips / x.size
    in_range = int((a.bin_index != 255).sum())
    key = ("f64" if dtype == np.float64 else "f32", eb)
    record_property("flip_rate", rate)
    print(f"\nNOISE {key[0]} eb={eb:g}: {flips} of {x.size} bin ids differ ({rate:.3e}; {flips / max(1, in_range):.3e} of the in-range ones), "
          f"cnt {a.cnt} vs {b.cnt}")
    assert rate <= CEIL[key], (key, rate)
    # aggregate quality is insensitive: the reconstructions' PSNR agree to 1e-6 relative wherever the bins carry information
    ra, rb = O.decompress(a, O.FAST), O.decompress(b, O.NAIVE)
    ref = a.scaled.astype(np.float64) * a.sf
    pa, pb = O.psnr(ref.astype(dtype), ra)["psnr"], O.psnr(ref.astype(dtype), rb)["psnr"]
    if not (dtype == np.float32 and eb <= 1e-5):
        assert abs(pa - pb) <= 1e-6 * abs(pa), (pa, pb)
