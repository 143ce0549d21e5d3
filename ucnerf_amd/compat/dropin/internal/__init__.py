"""`internal` as the reference's entry scripts import it (train.py:13-21, render.py:8-15, eval.py, extract.py, tsdf.py):
an OVERLAY package.  `internal.models` and `internal.train_utils` come from this directory (the MI355X ray-march behind
the reference's names); every other sub-module -- configs, datasets, camera_utils, image, utils, vis, checkpoints, coord,
stepfun, ... -- falls through `__path__` to the caller's own `internal/` directory, i.e. the reference's files, unmodified.

The caller's package is the first `internal/` directory on sys.path (the script's directory / the working directory,
as `python train.py` and `accelerate launch train.py` set it up) that is not this one."""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))


def _upstream_internal():
    seen = set()
    for d in list(sys.path) + [os.getcwd()]:
        cand = os.path.realpath(os.path.join(d or os.getcwd(), "internal"))
        if cand in seen or cand == os.path.realpath(_here):
            continue
        seen.add(cand)
        # the reference's package: recognise it by files the overlay relies on
        if all(os.path.isfile(os.path.join(cand, f)) for f in ("configs.py", "stepfun.py", "train_utils.py")):
            return cand
    return None


UPSTREAM = _upstream_internal()
if UPSTREAM is None:
    raise ImportError("ucnerf_amd drop-in overlay: no reference `internal/` package (configs.py, stepfun.py, train_utils.py) "
                      "found on sys.path or in the working directory -- run from the reference's nerf/ directory")
__path__ = [_here, UPSTREAM]
