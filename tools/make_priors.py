"""tools/make_priors.py -- the five Wild6D category shape priors (config/<cat>_wild6d/<cat>.obj of the reference: DATA files, vertex
and face lists) packed into one npz the package can load on a box without the reference checkout:
    python tools/make_priors.py   ->  self-corr-pose_amd/scp_amd/data/wild6d_priors.npz   (<cat>_v float32 [V,3], <cat>_f int32 [F,3])
Runs in the build container only (reads /root/reference).  bench.py --categories and scp_amd.mesh.category_prior() use the result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
from scp_amd.mesh import read_obj  # noqa: E402

REF = os.environ.get("SCP_REFERENCE", "/root/reference")
out = {}
for cat in ("bottle", "bowl", "camera", "laptop", "mug"):
    v, f = read_obj(os.path.join(REF, "config", cat + "_wild6d", cat + ".obj"))
    out[cat + "_v"], out[cat + "_f"] = np.asarray(v, np.float32), np.asarray(f, np.int32)
    print(cat, out[cat + "_v"].shape, out[cat + "_f"].shape)
path = os.path.join(ROOT, "self-corr-pose_amd", "scp_amd", "data", "wild6d_priors.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path))
