"""tests/posefit_inputs.py -- shim: the seeded pose-fitting inputs live in scp_amd/synthetic.py."""
from scp_amd.synthetic import posefit_inputs, umeyama_case  # noqa: F401
