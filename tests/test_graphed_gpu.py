"""scp_amd/graphed.py (opt-in, SCP_GRAPHS=1, experimental): the forward-only replay wrapper GraphedInference against eager execution.
GraphedSegment (forward + BACKWARD graphs; since round 5 an experiment under tools/graphed_segment.py) has no test on purpose: on this ROCm stack ending the
capture of a backward graph -- whose nodes the autograd engine's device thread records -- takes the interpreter down in
hipStreamEndCapture in every setup tried except the bench workload's (tools/graph_mlp.py: a two-layer torch MLP is enough;
tools/graph_variants.py).  The measurements of the full step under graphs (profiles/r04_host_enqueue.txt) come from that one setup."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_graphed_inference_clones_its_outputs():
    from scp_amd.graphed import GraphedInference
    torch.manual_seed(0)
    net = torch.nn.Linear(32, 8).cuda()
    gi = GraphedInference(lambda x: (net(x), x.sum(1)), warmup=1)
    outs = []
    for it in range(4):
        x = torch.full((4, 32), float(it + 1), device="cuda")
        with torch.no_grad():
            y, s = gi(x)
            assert torch.equal(y, net(x)) and torch.equal(s, x.sum(1))
        outs.append(y)
    assert gi.graph is not None
    assert not torch.equal(outs[-1], outs[-2]), "a later replay must not overwrite what an earlier call returned"
