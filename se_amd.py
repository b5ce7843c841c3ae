"""Import shim: the package directory required by the repo layout,
`sixty-years-of-frequency-domain-monaural-speech-enhancement_amd/`, is not a
valid Python identifier, so it is registered under the module name `se_amd`.
`import se_amd` (repo root on sys.path) is the supported way in.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "sixty-years-of-frequency-domain-monaural-speech-enhancement_amd")
_spec = importlib.util.spec_from_file_location(
    "se_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["se_amd"] = _mod
_spec.loader.exec_module(_mod)
