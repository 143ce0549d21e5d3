"""TEST STAND-IN (not the reference's file): the drop-in overlay recognises the caller's `internal/` package by the presence of
configs.py / stepfun.py / train_utils.py (ucnerf_amd/compat/dropin/internal/__init__.py) and falls through to it for every sub-module it
does not carry.  The GPU box has no /root/reference, so tests/test_dropin_device.py points the overlay at this directory instead."""
from ucnerf_amd.internal.configs import Config  # noqa: F401
