"""Synthetic phantoms and spectra for benchmarks and tests.

The reference bundles no phantom (input/params.txt:8-9 point at files that are not in its
checkout), so benchmark inputs are generated: a water cylinder with bone spheres in air, and
analytic Kramers spectra on the 1-keV grid the bundled spectra use.
"""
import numpy as np

from . import xcompy
from .system import AIR, BONE, WATER, VoxelPhantom, xRaySpectrum


def make_phantom(N, Nz=None, extent=51.2, seed=1234, name='synthetic', z_index=0, n_spheres=12):
    """uint8 ids: 0 air, 1 water cylinder (radius 0.40*extent, axis z), 2 bone spheres."""
    Nz = N if Nz is None else int(Nz)
    d = extent / N
    rng = np.random.default_rng(seed)
    c = (np.arange(N) + 0.5) * d - 0.5 * extent
    zc = (np.arange(Nz) + 0.5) * d - 0.5 * Nz * d
    vol = np.zeros((Nz, N, N), dtype=np.uint8)
    disc = (c[None, :] ** 2 + c[:, None] ** 2) <= (0.40 * extent) ** 2
    vol[:, disc] = 1
    radii = rng.uniform(0.02, 0.06, n_spheres) * extent
    centres = rng.uniform(-0.3, 0.3, (n_spheres, 3)) * extent
    if Nz == 1:
        centres[:, 2] = 0.0
    for r, (cx, cy, cz) in zip(radii, centres):
        ix = np.nonzero(np.abs(c - cx) <= r)[0]
        iy = np.nonzero(np.abs(c - cy) <= r)[0]
        iz = np.nonzero(np.abs(zc - cz) <= r)[0]
        if not (ix.size and iy.size and iz.size):
            continue
        sub = ((c[ix][None, None, :] - cx) ** 2 + (c[iy][None, :, None] - cy) ** 2
               + (zc[iz][:, None, None] - cz) ** 2) <= r * r
        blk = vol[iz[0]:iz[-1] + 1, iy[0]:iy[-1] + 1, ix[0]:ix[-1] + 1]
        blk[sub] = 2
    return VoxelPhantom.from_array(name, vol, [AIR, WATER, BONE], dx=d, dy=d, dz=d, z_index=z_index)


def kramers_spectrum(kVp, total_counts=1.0e6, al_cm=0.25, name=None):
    """I(E) ~ (kVp - E)/E * exp(-mu_Al(E) * al_cm) on E = 1..kVp keV, scaled to total_counts."""
    E = np.arange(1.0, float(kVp) + 1.0)
    mu_al = 2.699 * xcompy.mixatten('Al(100)', E)
    I = np.maximum(kVp - E, 0.0) / E * np.exp(-mu_al * al_cm)
    I *= total_counts / I.sum()
    return xRaySpectrum.from_arrays(name or f'{int(kVp)}kVp_kramers', E, I)


def uniform_grid_spectrum(n_bins=128, e_lo=20.0, e_hi=147.0, total_counts=1.0e6, name='grid128'):
    """Config 5 of BASELINE.json: 128 energy bins on linspace(20, 147)."""
    E = np.linspace(e_lo, e_hi, n_bins)
    I = np.maximum(150.0 - E, 0.0) / E * np.exp(-2.699 * xcompy.mixatten('Al(100)', E) * 0.25)
    I *= total_counts / I.sum()
    return xRaySpectrum.from_arrays(name, E, I)
