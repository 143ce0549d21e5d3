"""`import _gridencoder` shim: put ucnerf_amd/compat on sys.path and the reference's
gridencoder/grid.py:10 (`import _gridencoder as _backend`) binds to the HIP library."""
from ucnerf_amd.gridencoder._backend import (grad_total_variation, grid_encode_backward,  # noqa: F401
                                            grid_encode_forward)
