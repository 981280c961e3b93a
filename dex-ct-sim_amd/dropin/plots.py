"""``from plots import make_vmi, measure_roi, ...`` - the numerical helpers of the reference's analysis script
(plots.py:136-231) served by the engine; the reference's figure cells are not part of it."""
import _bootstrap  # noqa: F401
from dex_ct_sim_amd.plots import (crop_img, get_img_basismats, get_img_ct, get_xcat_mask, make_vmi,  # noqa: E402,F401
                                  measure_roi, vmi_rmse_sweep, vmi_roi_sweep)
