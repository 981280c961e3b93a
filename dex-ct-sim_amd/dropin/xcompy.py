import _bootstrap  # noqa: F401
from dex_ct_sim_amd.xcompy import mixatten, register_table  # noqa: E402,F401
