"""tests/scenes.py -- shim: the synthetic meshes / cameras live in scp_amd/synthetic.py (bench.py and smoke() use them)."""
from scp_amd.synthetic import (LOOK_AT_Z, bottle_like, camera_batch, face_gather, icosphere, octahedron, project,  # noqa: F401
                               random_rotations, raster_inputs)
