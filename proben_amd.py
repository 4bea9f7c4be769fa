"""Import shim: the package directory is named after the upstream project
(`multimodal-object-detection-via-probabilistic-ensembling_amd/`), which is not
a valid Python identifier.  `import proben_amd` loads that directory as the
package `proben_amd` (sub-modules import normally: `proben_amd.fusion`, ...).
"""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "multimodal-object-detection-via-probabilistic-ensembling_amd")
_spec = importlib.util.spec_from_file_location(
    "proben_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["proben_amd"] = _mod
_spec.loader.exec_module(_mod)
