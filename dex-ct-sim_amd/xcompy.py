"""Mass-attenuation lookup with the call surface of the reference's ``xcompy``.

The reference calls ``xc.mixatten(formula, E_keV)`` (matdecomp.py:158,
plots.py:138-140,514) on NIST-XCOM tables that live in the un-vendored
``x-tomo-sim`` submodule.  Those tables are not available offline, so this
module offers the same call with two sources:

* user tables: ``register_table(formula_or_element, E_keV, mu_rho)`` or a
  directory named by ``DEXCT_XCOM_DIR`` holding ``<Symbol>.txt`` files with two
  columns (E [keV], mu/rho [cm^2/g]); mixtures are weight-fraction sums,
  log-log interpolated per element;
* otherwise a documented analytic surrogate per element: Klein-Nishina
  incoherent scattering on Z/A free electrons plus a Z^n/E^k photo-electric
  term.  It is smooth (no absorption edges) and is NOT a NIST value; it exists
  so that the engine runs end to end and so that the parity fixtures have a
  frozen, reproducible table.  The HIP kernels only ever see the resulting
  ``mus[M, nE]`` tables, never this function.
"""
import os
import re

import numpy as np

# Z and standard atomic weight of the elements the XCAT / ICRU compositions use.
_ELEMENTS = {
    'H': (1, 1.008), 'He': (2, 4.0026), 'Li': (3, 6.94), 'Be': (4, 9.0122),
    'B': (5, 10.81), 'C': (6, 12.011), 'N': (7, 14.007), 'O': (8, 15.999),
    'F': (9, 18.998), 'Ne': (10, 20.180), 'Na': (11, 22.990), 'Mg': (12, 24.305),
    'Al': (13, 26.982), 'Si': (14, 28.085), 'P': (15, 30.974), 'S': (16, 32.06),
    'Cl': (17, 35.45), 'Ar': (18, 39.948), 'K': (19, 39.098), 'Ca': (20, 40.078),
    'Sc': (21, 44.956), 'Ti': (22, 47.867), 'V': (23, 50.942), 'Cr': (24, 51.996),
    'Mn': (25, 54.938), 'Fe': (26, 55.845), 'Co': (27, 58.933), 'Ni': (28, 58.693),
    'Cu': (29, 63.546), 'Zn': (30, 65.38), 'Mo': (42, 95.95), 'Ag': (47, 107.87),
    'Sn': (50, 118.71), 'I': (53, 126.90), 'Ba': (56, 137.33), 'Gd': (64, 157.25),
    'W': (74, 183.84), 'Pt': (78, 195.08), 'Au': (79, 196.97), 'Pb': (82, 207.2),
}

_N_A = 6.02214076e23
_R_E2 = 7.9407877e-26      # classical electron radius squared [cm^2]
_MEC2 = 510.99895          # electron rest energy [keV]
# photo-electric surrogate per atom: tau = _PE_K * Z**_PE_N / E_keV**_PE_P  [cm^2]
_PE_K, _PE_N, _PE_P = 2.0e-23, 4.4, 3.1

_user_tables = {}
_mix_cache = {}


def parse_formula(formula):
    """'H(11.2)O(88.8)' -> [('H', 0.112), ('O', 0.888)] (weight fractions)."""
    parts = re.findall(r'([A-Z][a-z]?)\(([^)]+)\)', formula)
    if not parts:
        raise ValueError(f'cannot parse material formula {formula!r}')
    w = np.array([float(p[1]) for p in parts], dtype=np.float64)
    w = w / w.sum()
    return [(p[0], wi) for p, wi in zip(parts, w)]


def register_table(name, E_keV, mu_rho):
    """Install a user table for an element symbol or a whole formula string."""
    E = np.asarray(E_keV, dtype=np.float64)
    m = np.asarray(mu_rho, dtype=np.float64)
    if E.ndim != 1 or E.shape != m.shape or np.any(np.diff(E) <= 0):
        raise ValueError('table needs increasing E and matching mu/rho')
    _user_tables[name] = (E, m)
    _mix_cache.clear()          # a table replaced under an existing name must not leave old mixtures behind



def _interp_loglog(E, tab):
    Et, mt = tab
    return np.exp(np.interp(np.log(E), np.log(Et), np.log(mt)))


def _klein_nishina(E):
    a = E / _MEC2
    l = np.log1p(2 * a)
    return 2 * np.pi * _R_E2 * ((1 + a) / a**2 * (2 * (1 + a) / (1 + 2 * a) - l / a)
                                + l / (2 * a) - (1 + 3 * a) / (1 + 2 * a)**2)


def _element(symbol, E):
    if symbol in _user_tables:
        return _interp_loglog(E, _user_tables[symbol])
    d = os.environ.get('DEXCT_XCOM_DIR')
    if d:
        path = os.path.join(d, symbol + '.txt')
        if os.path.exists(path):
            t = np.loadtxt(path)
            register_table(symbol, t[:, 0], t[:, 1])
            return _interp_loglog(E, _user_tables[symbol])
    if symbol not in _ELEMENTS:
        raise KeyError(f'no attenuation data for element {symbol!r}')
    Z, A = _ELEMENTS[symbol]
    per_atom = Z * _klein_nishina(E) + _PE_K * Z**_PE_N / E**_PE_P
    return per_atom * _N_A / A


def mixatten(formula, E_keV):
    """Mass attenuation coefficient [cm^2/g] of a weight-% mixture at E [keV].

    Same signature and units as the reference's xcompy.mixatten
    (matdecomp.py:158).  Returns float64 with the shape of ``E_keV``.
    """
    E = np.asarray(E_keV, dtype=np.float64)
    # the public calls ask for the same (material, energy grid) pairs on every call (0.3 ms of a 0.9 ms get_sino at the
    # reference's own size): keep the last results; register_table() empties the cache, another DEXCT_XCOM_DIR changes the key
    key = (formula, E.shape, E.tobytes(), os.environ.get('DEXCT_XCOM_DIR'))
    hit = _mix_cache.get(key)
    if hit is not None:
        return hit.copy()
    if formula in _user_tables:
        out = _interp_loglog(E, _user_tables[formula])
    else:
        out = np.zeros_like(E)
        for sym, w in parse_formula(formula):
            out = out + w * _element(sym, E)
    if len(_mix_cache) >= 256:
        _mix_cache.clear()
    _mix_cache[key] = out.copy()
    return out
