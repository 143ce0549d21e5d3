"""CPU oracle for the UC-NeRF ray-march hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the timed CPU baseline.  The shipped path
(``ucnerf_amd``) never imports this package and raises if its HIP library is
missing.

Contents
--------
grid_oracle.c   plain-C restatement of nerf/gridencoder/src/gridencoder.cu
grid_cpu.py     ctypes binding of grid_oracle.c, same call signatures as the
                reference's ``_gridencoder`` pybind module (bindings.cpp:5-9)
grid_numpy.py   second, independent (vectorised numpy) restatement of the same
                kernel; the two must agree bit-for-bit (tests/test_oracle_grid.py)
raymarch.py     torch-CPU restatement of nerf/internal/{stepfun,render,coord,
                math,models,extrinsic_optimizer}.py for the path

Parity status: the reference has NO tests, golden vectors or known-answer
fixtures for this path and its hash-grid op is CUDA-only.
The oracle is therefore pinned against outputs of the reference's own Python
imported in the authoring container (tests/golden/make_golden.py, fixtures
committed under tests/golden/), with grid_oracle.c standing in for the CUDA
kernel.  For the CUDA kernel itself parity is "unpinned" (two independent
restatements only).
"""
