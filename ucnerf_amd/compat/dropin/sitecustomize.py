"""Drop-in switch for the reference's entry scripts (train.py / render.py / eval.py / extract.py / tsdf.py, unchanged).

    PYTHONPATH=/path/to/repo/ucnerf_amd/compat/dropin:/path/to/repo accelerate launch train.py --gin_configs=configs/waymo.gin ...

Python imports `sitecustomize` from PYTHONPATH at interpreter start-up.  A script's own directory sits in front of
PYTHONPATH on sys.path, so the reference's `from internal import models` (train.py:17) would always find ITS
`internal/` package first; this module therefore puts one finder in front of sys.meta_path that answers for the
top-level name `internal` with the overlay package next to this file (compat/dropin/internal/), whose `__path__` falls
through to the caller's own `internal/` for every sub-module it does not carry.  Nothing else is touched: no reference
file is edited, every other import resolves as before.  A `sitecustomize` further down sys.path is still honoured."""
import importlib.abc
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_OVERLAY_INIT = os.path.join(_HERE, "internal", "__init__.py")


class _InternalOverlayFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != "internal" or not os.path.isfile(_OVERLAY_INIT):
            return None
        return importlib.util.spec_from_file_location("internal", _OVERLAY_INIT,
                                                      submodule_search_locations=[os.path.dirname(_OVERLAY_INIT)])


if not any(isinstance(f, _InternalOverlayFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _InternalOverlayFinder())

# chain to the environment's own sitecustomize, if another one exists behind this directory
for _d in sys.path:
    _cand = os.path.join(_d or os.getcwd(), "sitecustomize.py")
    if os.path.isfile(_cand) and os.path.realpath(os.path.dirname(_cand)) != os.path.realpath(_HERE):
        _spec = importlib.util.spec_from_file_location("_chained_sitecustomize", _cand)
        _mod = importlib.util.module_from_spec(_spec)
        try:
            _spec.loader.exec_module(_mod)
        except Exception:                                  # noqa: BLE001  (site.py also swallows a failing sitecustomize)
            pass
        break
