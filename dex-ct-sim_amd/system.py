"""System model: scanner geometry, voxel phantom, x-ray spectrum, parameter file.

Host-side mirror of the reference's ``xtomosim/system.py``, which is absent from the
reference checkout (un-vendored submodule).  The surface is reconstructed from the call sites:

* ``FanBeamGeometry(N_channels, N_proj, gamma_fan, SID, SDD, h_iso, eid, detector_file)``
  - plots.py:109-111; attributes ``A_iso, N_proj`` (main.py:68), ``det_E, det_eta_E, eid``
  (matdecomp.py:146-147).  README.md:14 calls it ``ScannerGeometry`` (alias below).
* ``VoxelPhantom(name, filename, matcomp_filename, Nx, Ny, Nz, z_index=0)`` - plots.py:124;
  voxel sizes from input/params.txt:13-15; ``M_mono(E)`` - plots.py:251.  README: ``Phantom``.
* ``xRaySpectrum(filename, name)``, ``.E``, ``.I0``, ``.rescale_counts(scale)`` - main.py:67-68,
  matdecomp.py:140,149-150.  README: ``Spectrum``.
* ``read_parameter_file(path)`` -> list of ``[run_id, do_fp, do_bp, ct, phantom, spec, N_matrix,
  FOV, ramp]`` - main.py:89-94; keys input/params.txt:1-37.

These classes hold NumPy data only; device work happens in forward_project / matdecomp.
"""
import csv
import json
import os

import numpy as np

from . import xcompy


def read_half_split(filename):
    """The reference's .bin tables: float32, first half energies [keV], second half values."""
    d = np.fromfile(filename, dtype=np.float32)
    if d.size == 0 or d.size % 2:
        raise ValueError(f'{filename}: expected an even number of float32 values, got {d.size}')
    n = d.size // 2
    return d[:n].astype(np.float64), d[n:].astype(np.float64)


class FanBeamGeometry:
    """Equiangular fan-beam scanner.

    Conventions of this build (the reference's are not observable): the source of view i sits at
    angle beta_i = i * theta_tot / N_proj on a circle of radius SID; channel c looks along
    beta + pi + gamma_c with gamma_c = (c - (N_channels-1)/2) * gamma_fan / N_channels; the
    detector pixel lies on an arc of radius SDD about the source.  ``N_rows`` > 1 stacks
    identical fans along z, row r imaging phantom slice ``z_index + r`` (an extension; the
    reference is single-row); ``cone=True`` instead makes the rows a real 2-D detector seen from a point
    source (cone beam).
    """

    def __init__(self, N_channels=800, N_proj=1200, gamma_fan=0.8230337, SID=60.0, SDD=100.0, h_iso=1.0,
                 eid=True, detector_file=None, theta_tot=2 * np.pi, N_rows=1, cone=False, src_z=0.0):
        self.N_channels = int(N_channels)
        self.N_proj = int(N_proj)
        self.N_rows = int(N_rows)
        # cone=True: true 3-D rays (an extension): the source at height src_z [cm, 0 = centre of the phantom],
        # detector row r at height (r - (N_rows-1)/2) * h on the detector arc, h = h_iso * SDD / SID.
        self.cone = bool(cone)
        self.src_z = float(src_z)
        self.gamma_fan = float(gamma_fan)
        self.theta_tot = float(theta_tot)
        self.SID = float(SID)
        self.SDD = float(SDD)
        self.h_iso = float(h_iso)
        self.eid = bool(eid)
        self.detector_file = detector_file
        self.dgamma = self.gamma_fan / self.N_channels
        self.gammas = (np.arange(self.N_channels) - 0.5 * (self.N_channels - 1)) * self.dgamma
        self.thetas = np.arange(self.N_proj) * (self.theta_tot / self.N_proj)
        self.s_iso = self.SID * self.dgamma            # channel width projected to isocentre [cm]
        self.s = self.SDD * self.dgamma                # channel width on the detector arc [cm]
        self.h = self.h_iso * self.SDD / self.SID      # pixel height on the detector [cm]
        self.A_iso = self.s_iso * self.h_iso           # pixel area at isocentre [cm^2] (main.py:68)
        if detector_file is None:
            self.det_E = np.array([1.0, 1.0e4])
            self.det_eta_E = np.array([1.0, 1.0])      # ideal detector
        else:
            self.det_E, self.det_eta_E = read_half_split(detector_file)

    def row_z(self):
        """Heights [cm] of the detector rows of a cone-beam scan."""
        return (np.arange(self.N_rows) - 0.5 * (self.N_rows - 1)) * self.h

    # float64 tables every implementation (HIP, oracle) starts from
    def view_cs(self):
        return np.ascontiguousarray(np.stack([np.cos(self.thetas), np.sin(self.thetas)], axis=1))

    def chan_cs(self):
        return np.ascontiguousarray(np.stack([np.cos(self.gammas), np.sin(self.gammas)], axis=1))

    def detector_response(self, E):
        """eta(E) (times E for an energy-integrating detector) - matdecomp.py:146-148."""
        r = np.interp(E, self.det_E, self.det_eta_E)
        return r * E if self.eid else r


class Material:
    def __init__(self, name, density, matcomp):
        self.name = name
        self.density = float(density)
        self.matcomp = matcomp


AIR = Material('air', 0.001205, 'C(0.0124)N(75.5268)O(23.1781)Ar(1.2827)')
WATER = Material('water', 1.0, 'H(11.1894)O(88.8106)')
BONE = Material('ICRU bone', 1.92, 'H(3.4)C(15.5)N(4.2)O(43.5)Na(0.1)Mg(0.2)P(10.3)S(0.3)Ca(22.5)')


def read_materials_csv(path):
    """Materials table: rows ``id, name, density [g/cm3], composition`` (header optional).

    The reference's ``xcat_materials.csv`` (input/params.txt:9) is not in its checkout, so the
    column order is this build's definition; ``composition`` uses the weight-% syntax of
    matdecomp.py:13.
    """
    mats = {}
    with open(path, newline='') as f:
        for row in csv.reader(f):
            row = [c.strip() for c in row]
            if len(row) < 4 or not row[0].lstrip('-').isdigit():
                continue
            mats[int(row[0])] = Material(row[1], float(row[2]), row[3])
    if not mats:
        raise ValueError(f'{path}: no material rows')
    n = max(mats) + 1
    return [mats.get(i, Material(f'unused{i}', 0.0, 'H(100)')) for i in range(n)]


class VoxelPhantom:
    """uint8 material-id volume [Nz, Ny, Nx] plus a material table."""

    def __init__(self, name, filename, matcomp_filename, Nx, Ny, Nz, ind=0, dx=0.1, dy=None, dz=None,
                 z_index=None):
        self.name = name
        self.Nx, self.Ny, self.Nz = int(Nx), int(Ny), int(Nz)
        self.dx = float(dx)
        self.dy = float(dx if dy is None else dy)
        self.dz = float(dx if dz is None else dz)
        self.z_index = int(ind if z_index is None else z_index)
        self.filename = filename
        self.matcomp_filename = matcomp_filename
        if filename is not None:
            v = np.fromfile(filename, dtype=np.uint8)
            if v.size != self.Nx * self.Ny * self.Nz:
                raise ValueError(f'{filename}: {v.size} voxels, expected {self.Nx * self.Ny * self.Nz}')
            self.volume = v.reshape(self.Nz, self.Ny, self.Nx)
            self.materials = read_materials_csv(matcomp_filename)
        else:
            self.volume = None
            self.materials = None

    @classmethod
    def from_array(cls, name, volume, materials, dx=0.1, dy=None, dz=None, z_index=0):
        volume = np.ascontiguousarray(volume, dtype=np.uint8)
        if volume.ndim == 2:
            volume = volume[None]
        nz, ny, nx = volume.shape
        self = cls(name, None, None, nx, ny, nz, z_index, dx, dy, dz)
        self.volume = volume
        self.materials = list(materials)
        if int(volume.max()) >= len(self.materials):
            raise ValueError('volume holds a material id without a table entry')
        return self

    # The device-resident copy of the volume (forward_project._projector) is keyed on ``version``: assigning
    # ``phantom.volume`` bumps it; after editing the array in place call ``touch()`` (or forward_project.invalidate()).
    @property
    def volume(self):
        return self._volume

    @volume.setter
    def volume(self, v):
        self._volume = v
        self.version = getattr(self, 'version', 0) + 1

    def touch(self):
        """Tell the engine that ``volume`` was edited in place."""
        self.version = getattr(self, 'version', 0) + 1

    @property
    def n_materials(self):
        return len(self.materials)

    def mu_table(self, E_keV):
        """Linear attenuation [1/cm] of every material id at E: [M, nE] float64."""
        E = np.asarray(E_keV, dtype=np.float64)
        return np.stack([m.density * xcompy.mixatten(m.matcomp, E) for m in self.materials])

    def M_mono(self, E0, z=None):
        """Mono-energetic linear-attenuation map of one slice (plots.py:251): [Ny, Nx] float32."""
        mu = self.mu_table(np.array([float(E0)]))[:, 0]
        return mu[self.volume[self.z_index if z is None else z]].astype(np.float32)


class xRaySpectrum:
    """Polychromatic spectrum: energies [keV] and counts per energy bin (main.py:66-68)."""

    def __init__(self, filename, name, E=None, I0=None):
        self.filename = filename
        self.name = name
        if filename is not None:
            self.E, self.I0 = read_half_split(filename)
        else:
            self.E = np.asarray(E, dtype=np.float64)
            self.I0 = np.asarray(I0, dtype=np.float64)
        self.I0_raw = self.I0.copy()

    @classmethod
    def from_arrays(cls, name, E, I0):
        return cls(None, name, E, I0)

    def rescale_counts(self, scale, verbose=False):
        self.I0 = self.I0 * scale
        if verbose:
            print(f'{self.name}: rescaled counts by {scale:.4e}, total {self.I0.sum():.4e}')

    def bin_widths(self):
        """dE with the first bin spanning 0..E[0] (matdecomp.py:142)."""
        return np.concatenate([[self.E[0]], np.diff(self.E)])


# README.md:14-16 spellings
ScannerGeometry = FanBeamGeometry
Phantom = VoxelPhantom
Spectrum = xRaySpectrum


def _one_run(p, base_dir):
    def rel(path):
        if path in (None, 'NA'):
            return None
        return path if os.path.isabs(path) or os.path.exists(path) else os.path.join(base_dir, path)

    geometry = p.get('scanner_geometry', 'fan_beam')
    if geometry not in ('fan_beam', 'cone_beam'):          # cone_beam: extension, needs "N_rows"
        raise ValueError('scanner_geometry must be "fan_beam" or "cone_beam"')
    mode = p.get('detector_mode', 'eid')
    ct = FanBeamGeometry(N_channels=p['N_channels'], N_proj=p['N_projections'], gamma_fan=p['fan_angle_total'],
                         SID=p['SID'], SDD=p['SDD'], h_iso=p.get('detector_px_height', 1.0), eid=(mode == 'eid'),
                         detector_file=rel(p.get('detector_filename')),
                         theta_tot=p.get('rotation_angle_total', 2 * np.pi), N_rows=p.get('N_rows', 1),
                         cone=(geometry == 'cone_beam'), src_z=p.get('source_z', 0.0))
    ptype = p.get('phantom_type', 'voxel')
    if ptype == 'voxel':
        phantom = VoxelPhantom(p['phantom_id'], rel(p['phantom_filename']), rel(p['matcomp_filename']),
                               p['Nx'], p['Ny'], p['Nz'], p.get('z_index', 0), p['dx'], p['dy'], p['dz'])
    elif ptype == 'synthetic':      # extension: no phantom is bundled with the reference
        from .synthetic import make_phantom
        phantom = make_phantom(p['Nx'], p.get('Nz', 1), extent=p['Nx'] * p['dx'], seed=p.get('phantom_seed', 1234),
                               name=p.get('phantom_id', 'synthetic'), z_index=p.get('z_index', 0))
    else:
        raise ValueError(f'unknown phantom_type {ptype!r}')
    spec = None
    if p.get('spectrum_filename', 'NA') != 'NA':
        spec = xRaySpectrum(rel(p['spectrum_filename']), p.get('spectrum_id', 'spectrum'))
        n = p.get('N_photons_per_cm2_per_scan', 'NA')
        if n != 'NA':
            spec.rescale_counts(float(n) * ct.A_iso / ct.N_proj / spec.I0.sum())
    out = [p['RUN_ID'], bool(p.get('forward_project', True)), bool(p.get('back_project', False)), ct, phantom, spec]
    out += [p.get('N_recon_matrix', 512), p.get('FOV_recon', 50.0), p.get('ramp_filter_percent_Nyquist', 1.0)]
    return out


def read_parameter_file(filename, base_dir=None):
    """JSON parameter file -> list of runs (main.py:89-94).  A file may hold one object or a list.

    Paths in the reference's file are relative to the working directory ('./input/...').  A path that
    does not exist there is looked up under ``base_dir`` (default: the directory above the params
    file's own 'input/' folder); the process working directory is never changed."""
    with open(filename) as f:
        data = json.load(f)
    runs = data if isinstance(data, list) else [data]
    if base_dir is None:
        base_dir = os.path.dirname(os.path.abspath(filename))
        base_dir = os.path.dirname(base_dir) if os.path.basename(base_dir) == 'input' else base_dir
    return [_one_run(p, os.path.abspath(base_dir)) for p in runs]
