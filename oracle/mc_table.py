"""The marching-cubes case table for the host restatement (oracle/marching.py): taken from its generator,
tools/gen_mc_table.py, which also writes the device kernels' header -- one derivation, two consumers.  TEST INFRASTRUCTURE."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location(
    "gen_mc_table", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_mc_table.py"))
_gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_gen)
build, case_triangles, header_text, EDGES, CORNERS, FACES, HEADER = (_gen.build, _gen.case_triangles, _gen.header_text, _gen.EDGES,
                                                                    _gen.CORNERS, _gen.FACES, _gen.HEADER)
