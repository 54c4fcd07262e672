"""What "agreement to rounding noise" means for two evaluation orders of the same block transform (the oracle's
pinned fast flow, its definition-order DFT, FFTW inside the reference): the shared yardstick of the CPU and GPU tests.

A 64-point transform evaluated in precision eps carries an absolute error of a few eps * ||block||_2 in EVERY
coefficient, however small that coefficient is.  Binning divides by bin_width = 2 eb, so two correct transforms
may put a coefficient into bins that differ by up to noise / bin_width (+1), and may disagree on "in range" when
the coefficient is within noise of +-range_max.  Nothing else may differ."""
import numpy as np

# measured on seeded smooth + noisy fields: max |coef_fast - coef_naive| <= 6 eps ||block||_2 (f32), 4 eps (f64);
# the definition-order sum of 64 products is the noisier of the two
K_NOISE = 12.0


def block_noise(scaled, dtype):
    """Per-block bound on |coef_a - coef_b| between two correct evaluations (length-nblk array)."""
    n = scaled.size
    nblk = (n + 63) // 64
    pad = np.zeros(nblk * 64, np.float64)
    pad[:n] = scaled.astype(np.float64)
    return K_NOISE * float(np.finfo(dtype).eps) * np.sqrt((pad.reshape(nblk, 64) ** 2).sum(axis=1))


def classify_flips(ca, cb, eb):
    """ca, cb: oracle.compress(..., want_coef=True) results of the same input under two flows.
    Returns (flips, illegal): number of differing bin ids, and how many of them the noise bound cannot explain."""
    dtype = ca.dtype
    n = ca.n
    noise = np.repeat(block_noise(ca.scaled, dtype), 64)[:n]
    a = ca.coef.astype(np.float64)
    b = cb.coef.astype(np.float64)
    assert np.all(np.abs(a - b) <= noise + 1e-300), "coefficients differ by more than rounding noise"
    diff = ca.bin_index != cb.bin_index
    flips = int(diff.sum())
    if not flips:
        return 0, 0
    bw = 2.0 * eb
    rmax = 255.0 * eb
    ia = np.nonzero(diff)[0]
    qa = np.floor((a[ia] + rmax) / bw)
    qb = np.floor((b[ia] + rmax) / bw)
    near_limit = (np.abs(np.abs(a[ia]) - rmax) <= noise[ia] * (1 + 1e-9) + 1e-300)
    exc_a = ca.bin_index[ia] == 255
    exc_b = cb.bin_index[ia] == 255
    ok = np.where(exc_a | exc_b, near_limit, np.abs(qa - qb) <= np.ceil(noise[ia] / bw) + 1)
    return flips, int((~ok).sum())
