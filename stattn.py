"""Import shim: the product package lives in the directory the build contract names,
`video-description-with-spatial-temporal-attention_amd/`, which is not a valid Python
identifier.  `import stattn` registers that directory under the importable name `stattn`
(so `from stattn import model_attention` works from the repo root)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "video-description-with-spatial-temporal-attention_amd")
_spec = importlib.util.spec_from_file_location(
    "stattn", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["stattn"] = _mod
_spec.loader.exec_module(_mod)
