"""Model-selection metrics of the validation styles, on the device.

The reference copies the validation styles to the host every epoch and calls ``scipy.stats.shapiro`` per style
column and ``scipy.stats.spearmanr`` per column pair (``sc/clustering/trainer.py:286-292``).  Here the styles stay
in HBM: ``raae_style_metrics`` ranks the columns and forms every W and rho in two launches, and the host reads
back ``k + k(k-1)/2`` doubles.  Only the Shapiro-Wilk coefficient vector, which depends on ``n`` alone, is built
on the host (once).
"""
import numpy as np
import torch

from . import ops


def _ppnd_as111(p):
    """Normal quantile, Beasley & Springer's AS 111 (the ``ppnd`` scipy's Shapiro-Wilk code carries; it is good
    to ~1e-9, so ``scipy.special.ndtri`` would NOT reproduce scipy's coefficients beyond that)."""
    p = np.asarray(p, dtype=np.float64)
    q = p - 0.5
    r = q * q
    centre = q * (((-25.44106049637 * r + 41.39119773534) * r - 18.61500062529) * r + 2.50662823884) / \
        ((((3.13082909833 * r - 21.06224101826) * r + 23.08336743743) * r - 8.47351093090) * r + 1.0)
    t = np.sqrt(-np.log(np.where(q > 0, 1.0 - p, p)))
    tail = (((2.32121276858 * t + 4.85014127135) * t - 2.29796479134) * t - 2.78718931138) / \
        ((1.63706781897 * t + 3.54388924762) * t + 1.0)
    return np.where(np.abs(q) <= 0.42, centre, np.where(q < 0, -tail, tail))


def _poly(c, x):
    p = x * c[-1]
    for j in range(len(c) - 2, 0, -1):
        p = (p + c[j]) * x
    return c[0] + p


def shapiro_coefficients(n):
    """The ``n // 2`` coefficients ``a`` of the W test for sample size ``n`` (Royston 1992 / AS R94 as run by
    scipy 1.15.3's ``shapiro``): normalised expected normal order statistics with the two outermost replaced by
    polynomial approximations in ``n**-0.5``.  Equal to scipy's own vector to 1 ulp (tests/test_host_cpu.py)."""
    if n < 3:
        raise ValueError("Data must be at least length 3.")
    nn2 = n // 2
    if n == 3:
        return np.array([np.sqrt(0.5)])
    c1 = [0.0, 0.221157, -0.147981, -0.2071190e1, 0.4434685e1, -0.2706056e1]
    c2 = [0.0, 0.42981e-1, -0.293762, -0.1752461e1, 0.5682633e1, -0.3582633e1]
    m = _ppnd_as111((np.arange(1, nn2 + 1) - 0.375) / (n + 0.25))
    summ2 = 2.0 * float(np.sum(m * m))
    ssumm2, rsn = np.sqrt(summ2), 1.0 / np.sqrt(n)
    a = np.zeros(nn2)
    a1 = _poly(c1, rsn) - m[0] / ssumm2
    if n > 5:
        a2 = -m[1] / ssumm2 + _poly(c2, rsn)
        fac = np.sqrt((summ2 - 2.0 * m[0] ** 2 - 2.0 * m[1] ** 2) / (1.0 - 2.0 * a1 ** 2 - 2.0 * a2 ** 2))
        a[1] = a2
        i1 = 2
    else:
        fac = np.sqrt((summ2 - 2.0 * m[0] ** 2) / (1.0 - 2.0 * a1 ** 2))
        i1 = 1
    a[0] = a1
    a[i1:] = -m[i1:] / fac
    return a


class StyleMetrics:
    """Device buffers for one ``(n, k)``; ``launch(z)`` enqueues the two kernels on the current stream (it is
    capturable), ``read()`` returns ``(W[k], rho[k(k-1)/2])`` as numpy float64."""

    def __init__(self, n, k, device):
        self.n, self.k = int(n), int(k)
        self.a = torch.as_tensor(shapiro_coefficients(self.n), dtype=torch.float64).to(device)
        self.work = torch.empty(2 * self.k * self.n, dtype=torch.float64, device=device)
        self.out = torch.zeros(self.k + self.k * (self.k - 1) // 2, dtype=torch.float64, device=device)

    def launch(self, z):
        assert tuple(z.shape) == (self.n, self.k), (tuple(z.shape), self.n, self.k)
        ops.style_metrics(z, self.n, self.k, self.a, self.work, self.out)

    def read(self):
        v = self.out.cpu().numpy()
        return v[:self.k], v[self.k:]
