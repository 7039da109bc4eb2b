"""Synthetic XANES-like spectra + descriptors (SURVEY.md §8d ``make_spectra``).

The reference's real CSVs are not shipped (``/root/reference/.MISSING_LARGE_BLOBS``),
so every test, golden fixture and benchmark uses this generator.  The spectra are
O(1), mostly non-negative, and genuinely rank-correlated with the descriptors so
the rank loss is O(0.1) rather than ~0.  Column 1 of the descriptors is discrete
(4, 5 or 6 -- like a coordination number) so that ties occur in the rank loss.
"""
import numpy as np


def make_spectra(n_rows, n_points=256, n_aux=5, seed=0):
    """Return ``(spec[n_rows, n_points] f64, aux[n_rows, n_aux] f64, grid[n_points])``."""
    rng = np.random.default_rng(seed)
    n_d = max(n_aux, 5)
    d = rng.standard_normal((n_rows, n_d))
    d[:, 1] = rng.integers(4, 7, size=n_rows)
    e = np.linspace(0.0, 1.0, n_points)[None, :]
    d0, d1, d2, d3, d4 = (d[:, i:i + 1] for i in range(5))
    spec = 0.5 + np.arctan(40.0 * (e - 0.25 - 0.02 * d0)) / np.pi
    spec = spec + (0.6 + 0.1 * (d1 - 5.0)) * np.exp(
        -((e - 0.32 - 0.01 * d0) / (0.04 + 0.005 * np.tanh(d3))) ** 2)
    spec = spec + 0.08 * np.sin(2 * np.pi * (3 + 0.5 * np.tanh(d2)) * e + 0.5 * d4) \
        * np.exp(-2 * e) * (e > 0.3)
    for k in range(5, n_d):  # extra descriptors modulate further sinusoid terms
        dk = d[:, k:k + 1]
        spec = spec + 0.02 * np.tanh(dk) * np.sin(2 * np.pi * (k - 1) * e) * (e > 0.3)
    spec = spec + 0.005 * rng.standard_normal((n_rows, n_points))
    grid = 5450.0 + 0.5 * np.arange(n_points)
    return spec, d[:, :n_aux].copy(), grid


def write_csv(path, spec, aux, grid):
    """Write the CSV schema ``sc/clustering/dataloader.py:12-25`` parses:
    two index columns, ``AUX_*`` columns, then ``ENE_<eV>`` columns."""
    n, n_aux = aux.shape
    cols = ["mpid", "site"] + [f"AUX_{k}" for k in range(n_aux)] + [f"ENE_{g:.1f}" for g in grid]
    with open(path, "w") as f:
        f.write(",".join(cols) + "\n")
        for i in range(n):
            row = [f"mp-{i}", "0"] + [repr(float(v)) for v in aux[i]] + [repr(float(v)) for v in spec[i]]
            f.write(",".join(row) + "\n")
