"""MI355X-native engine for the dex-ct-sim hot path (Siddon forward projection + Gauss-Newton
basis-material decomposition).  Import as ``dex_ct_sim_amd`` (the repository directory is named
``dex-ct-sim_amd``; ``dex_ct_sim_amd.py`` at the repository root maps the name).

Host code mirrors the reference's call surface; the work runs in ``libdexct_hip.so``
(include/dexct.h) - importing the compute modules fails if that library is not built.
"""
from .system import (FanBeamGeometry, Material, Phantom, ScannerGeometry, Spectrum, VoxelPhantom,  # noqa: F401
                     read_parameter_file, xRaySpectrum)

__all__ = ['FanBeamGeometry', 'ScannerGeometry', 'VoxelPhantom', 'Phantom', 'xRaySpectrum', 'Spectrum', 'Material',
           'read_parameter_file', 'get_sino', 'get_sinos', 'get_recon', 'get_basismat_sinos', 'do_matdecomp_gn']


def __getattr__(name):
    # compute entry points load torch + the HIP library on first use
    if name in ('get_sino', 'get_sinos', 'Projector'):
        from . import forward_project
        return getattr(forward_project, name)
    if name == 'get_recon':
        from . import back_project
        return back_project.get_recon
    if name in ('make_vmi', 'measure_roi', 'vmi_roi_sweep', 'vmi_rmse_sweep'):
        from . import plots
        return getattr(plots, name)
    if name in ('get_basismat_sinos', 'do_matdecomp_gn', 'optimize_sino', 'optimize_sino_cpu'):
        from . import matdecomp
        return getattr(matdecomp, name)
    raise AttributeError(name)
