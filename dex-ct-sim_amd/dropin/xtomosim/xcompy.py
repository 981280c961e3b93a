from dex_ct_sim_amd.xcompy import mixatten, register_table  # noqa: F401
