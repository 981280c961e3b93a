from dex_ct_sim_amd.back_project import get_recon  # noqa: F401
