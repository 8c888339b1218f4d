"""The co-residency screen: kernels of this build stay correct beside bf16-MFMA wavefronts.

gfx950 erratum, characterised in round 5 (csrc/selftest.hip, profiles/r05_packed_fp32_erratum.txt): v_pk_{mul,add,fma}_f32 with op_sel
[0,1] return a wrong low half while a K-doubled 16-bit MFMA executes on the same SIMD.  The overlapped schedule (scp_amd/streams.py)
puts the step's kernels beside the split-bf16 GEMMs all the time, so:
  * positive controls FIRST -- the self-checking packed product must go wrong beside the load on this box, and the rasteriser as it was
    compiled until round 4 (lib/libscp_hip_slpctl.so, SLP-vectorised) must return different images; if the box does not reproduce the
    erratum the screen proves nothing here and says so (skip), it never passes vacuously;
  * then every stage of the training step at the bench batch -- forward AND backward, the ATen glue, clip + fused AdamW, a 1-rank RCCL
    all-reduce -- and the whole step run many times under the load and must reproduce their unloaded results (tests/coresidency.py);
  * the static half (no such instruction in any shipped kernel) is tests/test_capi_symbols.py.
"""
import os
import subprocess
import sys

import pytest
import torch

import coresidency as cr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = int(os.environ.get("SCP_SCREEN_PASSES", "40"))     # suite default; the recorded table (profiles/r05_coresidency_screen.txt) used 300

_CONTROL = {}


def _erratum_reproduces():
    if "lo" not in _CONTROL:
        _CONTROL["alone"] = cr.erratum_counters(0, None, launches=4)
        _CONTROL["fp32"] = cr.erratum_counters(0, 1, launches=4)
        _CONTROL["lo"], _CONTROL["hi"] = cr.erratum_counters(0, 0, launches=16)
    return _CONTROL["lo"] > 0


def _need_control():
    if not _erratum_reproduces():
        pytest.skip("the packed-fp32 erratum did not reproduce on this box (%r): the screen would be vacuous" % (_CONTROL,))


def test_positive_control_the_erratum_form_goes_wrong_only_beside_bf16_mfma():
    """v_pk_mul_f32 op_sel:[0,1]: exact alone and beside fp32 MFMAs, wrong LOW halves beside v_mfma_f32_32x32x16_bf16; the plain form and
    the mirrored selection stay exact beside it"""
    _need_control()
    assert _CONTROL["alone"] == (0, 0) and _CONTROL["fp32"] == (0, 0), _CONTROL
    assert _CONTROL["lo"] > 1000 and _CONTROL["hi"] == 0, _CONTROL
    assert cr.erratum_counters(1, 0, launches=8) == (0, 0)
    assert cr.erratum_counters(2, 0, launches=8) == (0, 0)


def test_positive_control_the_slp_built_rasteriser_fails_the_screen():
    """the SAME screen on the rasteriser as compiled until round 4 (packed fp32 with op_sel [0,1] in its cross products) must see wrong
    images -- run in a subprocess because a process binds one libscp_hip"""
    _need_control()
    ctl = os.path.join(ROOT, "self-corr-pose_amd", "lib", "libscp_hip_slpctl.so")
    assert os.path.exists(ctl), "python self-corr-pose_amd/build.py --control"
    code = ("import sys; sys.path[:0] = [%r, %r]; import coresidency as cr\n"
            "v = cr.raster_victims()['raster_forward/softtex_s1e-3']\n"
            "bad = passes = 0\n"
            "for rnd in range(4):\n"
            "    r = cr.screen(v, 40); bad += r['bad']; passes += r['passes']\n"
            "    if bad >= 3: break\n"
            "print('RESULT', bad, passes, r['deterministic'])\n") % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "self-corr-pose_amd"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SCP_HIP_LIB=ctl), capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert line, out.stdout + out.stderr
    bad, passes, det = line[0].split()[1:]
    assert det == "True", "the control rasteriser is not even deterministic WITHOUT the load"
    if int(bad) < 3:
        # the self-checking kernel (first control) did fail on this box, so the load is real; the wrong lanes just missed the control's
        # few op_sel [0,1] instructions in 160 passes.  Inconclusive, not a failure of the product: say so instead of stopping the suite.
        pytest.skip("the SLP-built control rasteriser showed only %s bad passes of %s under the load" % (bad, passes))


@pytest.mark.parametrize("name", ["raster_forward/softtex_s1e-3", "raster_forward/depth_s1e-4", "raster_forward_backward/softtex_s1e-3",
                                  "raster_forward_backward/depth_s1e-4"])
def test_rasteriser_is_clean_beside_bf16_mfma(name):
    _need_control()
    # 30 passes in the suite (the depth passes take ~0.4 s each under the load: 300 of them were 2 x 126 s of a 10-minute suite, VERDICT
    # r5); the full table is `python tests/coresidency.py 300` -> profiles/r05_coresidency_screen.txt
    r = cr.screen(cr.raster_victims()[name], min(PASSES, int(os.environ.get("SCP_SCREEN_RASTER_PASSES", "30"))))
    assert r["bad"] == 0, r
    if name.startswith("raster_forward/"):
        assert r["deterministic"], r


@pytest.fixture(scope="module")
def victims():
    return cr.StepVictims()


@pytest.mark.parametrize("stage", [s for s in cr.STAGES if s != "step"])
def test_step_stage_is_clean_beside_bf16_mfma(victims, stage):
    """forward + backward of one stage of the B = 32 step, frozen inputs, PASSES times under the load"""
    _need_control()
    r = cr.screen(getattr(victims, stage), PASSES)
    assert r["bad"] == 0, (stage, r)


def test_whole_step_is_clean_beside_bf16_mfma(victims):
    """Trainer.step from one snapshot (every kernel of the step incl. the ATen glue, clip and fused AdamW): 12 loss terms, the clipped
    flat gradient and the updated parameters reproduce the unloaded step"""
    _need_control()
    r = cr.screen(victims.step, max(PASSES // 5, 20))
    assert r["bad"] == 0, r


def test_rccl_all_reduce_is_clean_beside_bf16_mfma(victims):
    """the reduction kernels of a 1-rank RCCL all-reduce over the flat gradient buffer (force_collectives path of FlatGradients)"""
    _need_control()
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29613")
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        src = torch.randn(victims.tr.grads.flat.numel(), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))

        def victim():
            buf = src.clone()
            dist.all_reduce(buf)
            return buf
        r = cr.screen(victim, PASSES)
        assert r["bad"] == 0 and r["deterministic"], r
    finally:
        if created:
            dist.destroy_process_group()
