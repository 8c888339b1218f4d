"""tests/synth.py -- the synthetic training batch of SURVEY.md 8(d) (own code, seeded).

img ~ U(0,1); mask = filled ellipse around the image centre; depth = mask * (500 + 40*smooth noise);
crop intrinsics around f = 755 px (256-px crop), principal point near the centre.  Returned as the
12-tuple MeshNet.forward consumes (after Trainer.batch_reshape's NDC conversion)."""


from scp_amd.synthetic import make_batch  # noqa: F401  (lives in the package: bench.py uses it)
