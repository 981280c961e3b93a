"""Import shim: the package directory is ``dex-ct-sim_amd`` (not a valid identifier); this module
loads it under the importable name ``dex_ct_sim_amd`` and replaces itself in sys.modules."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dex-ct-sim_amd')
_spec = importlib.util.spec_from_file_location('dex_ct_sim_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['dex_ct_sim_amd'] = _mod
_spec.loader.exec_module(_mod)
