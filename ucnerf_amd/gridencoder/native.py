"""Loader of the `_gridencoder` TORCH EXTENSION (ucnerf_amd/csrc/ext/gridencoder_bindings.cpp): the reference's pybind11
operator module (gridencoder/src/bindings.cpp:5-9) built over the C ABI.  The reference's own grid.py binds to it with
`ucnerf_amd/compat/native` on sys.path (`import _gridencoder as _backend`, grid.py:10); `load()` hands the same module to
this package (`use()` switches GridEncoder to it; the default backend is the ctypes form of the same three functions,
`_backend.py`, which needs no compiler on the deployment box)."""
import importlib.util
import os
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
NATIVE_DIR = os.path.join(os.path.dirname(_HERE), "compat", "native")
_mod = None


def path():
    return os.path.join(NATIVE_DIR, "_gridencoder" + sysconfig.get_config_var("EXT_SUFFIX"))


def load():
    global _mod
    if _mod is None:
        import torch  # noqa: F401  (libtorch must be loaded before the extension's dependencies resolve)
        p = path()
        if not os.path.exists(p):
            raise ImportError(f"{p} not found: build it with ucnerf_amd/csrc/ext/build_ext.sh (g++ against the installed torch)")
        spec = importlib.util.spec_from_file_location("_gridencoder", p)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _mod = mod
    return _mod


def use(native=True):
    """Route ucnerf_amd.gridencoder.GridEncoder through the extension module (True) or the ctypes backend (False)."""
    from . import _backend, grid
    grid._backend = load() if native else _backend
