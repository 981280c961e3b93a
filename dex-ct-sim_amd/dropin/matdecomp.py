import _bootstrap  # noqa: F401
from dex_ct_sim_amd.matdecomp import *  # noqa: E402,F401,F403
from dex_ct_sim_amd.matdecomp import (density1, density2, do_matdecomp_gn, get_basismat_sinos, mat1, mat2,  # noqa: E402,F401
                                      matcomp1, matcomp2, optimize_sino, optimize_sino_cpu)
