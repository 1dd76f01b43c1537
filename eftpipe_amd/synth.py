"""SYNTH-PLIN v1: closed-form synthetic linear power spectra.

There is no Boltzmann code on the GPU box, so benchmarks, fixtures and tests all feed the engine
with this seeded, closed-form family of Eisenstein-Hu-shaped spectra with a BAO-like wiggle.
Pure NumPy; used by bench.py, tools/make_fixtures.py and the tests.
"""
from __future__ import annotations

import numpy as np

KIN = np.logspace(-5, 0, 200)  # reference theory.py:562
FIDUCIAL = dict(Om=0.31, h=0.6777, ns=0.9611, A=1.0)
OM_AP = 0.307115  # reference cobaya/yamls/DR16_noric_LRG_NS_LP024_kmax0.20.yaml:32
_GL_X, _GL_W = np.polynomial.legendre.leggauss(64)


def hubble(Om, z):
    """E(z), flat LCDM (same closed form as reference pybird.py:34-36)."""
    return np.sqrt(Om * (1.0 + z) ** 3 + (1.0 - Om))


def da_func(Om, z):
    """int_0^z dz'/E(z') / (1+z) by 64-point Gauss-Legendre (reference pybird.py:39-42 uses quad)."""
    x = 0.5 * z * (_GL_X + 1.0)
    return 0.5 * z * np.sum(_GL_W / hubble(Om, x)) / (1.0 + z)


def growth_rate(Om, z):
    Omz = Om * (1.0 + z) ** 3 / hubble(Om, z) ** 2
    return Omz**0.55


def plin(k, Om=0.31, h=0.6777, ns=0.9611, A=1.0):
    k = np.asarray(k, dtype=float)

    def transfer(kk):
        q = kk / (Om * h)
        L0 = np.log(2.0 * np.e + 1.8 * q)
        C0 = 14.2 + 731.0 / (1.0 + 62.5 * q)
        return L0 / (L0 + C0 * q * q)

    def wiggle(kk):
        return 1.0 + 0.05 * np.sin(kk * 105.0) * np.exp(-((kk / 0.25) ** 2))

    k0 = 0.1
    return 6000.0 * A * (k / k0) ** ns * (transfer(k) / transfer(k0)) ** 2 * wiggle(k) / wiggle(k0)


def cosmology(z=0.7, **params):
    """-> dict(kin, Pin, f, DA, H) for one parameter point."""
    p = dict(FIDUCIAL)
    p.update(params)
    return dict(kin=KIN.copy(), Pin=plin(KIN, **p), f=float(growth_rate(p["Om"], z)),
                DA=float(da_func(p["Om"], z)), H=float(hubble(p["Om"], z)))


def draw_batch(B, z=0.7, seed=12345):
    """B seeded draws -> dict(kin[200], Pin[B,200], f[B], DA[B], H[B])."""
    rng = np.random.default_rng(seed)
    Om = rng.uniform(0.27, 0.35, B)
    h = rng.uniform(0.64, 0.72, B)
    ns = rng.uniform(0.93, 0.99, B)
    A = rng.uniform(0.8, 1.2, B)
    Pin = np.stack([plin(KIN, Om[i], h[i], ns[i], A[i]) for i in range(B)])
    f = np.array([growth_rate(Om[i], z) for i in range(B)])
    DA = np.array([da_func(Om[i], z) for i in range(B)])
    H = np.array([hubble(Om[i], z) for i in range(B)])
    return dict(kin=KIN.copy(), Pin=Pin, f=f, DA=DA, H=H)


def survey_kgrid(Nk):
    """Non-native k grid: 7 native low-k points + linspace(0.02, 0.3, Nk-7)."""
    low = np.array([0.001, 0.005, 0.0075, 0.01, 0.0125, 0.015, 0.0175])
    return np.concatenate([low, np.linspace(0.02, 0.3, Nk - 7)])
