from dex_ct_sim_amd.forward_project import get_sino, get_sinos  # noqa: F401
