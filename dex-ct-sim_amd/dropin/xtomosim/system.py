from dex_ct_sim_amd.system import *  # noqa: F401,F403
from dex_ct_sim_amd.system import (FanBeamGeometry, VoxelPhantom, read_parameter_file, xRaySpectrum,  # noqa: F401
                                   ScannerGeometry, Phantom, Spectrum)
