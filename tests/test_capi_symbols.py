"""CPU-side checks of the drop-in boundary: libscp_hip.so loads and exports exactly what
include/scp_hip.h declares; the Python binding lists the same symbols; no compute is launched."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "scp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    assert "scp_soft_rasterize_forward" in syms and "scp_soft_rasterize_backward" in syms


def test_library_exports_every_declared_symbol():
    from scp_amd import capi
    assert os.path.exists(capi.LIB_PATH), "build with python self-corr-pose_amd/build.py"
    handle = ctypes.CDLL(capi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), name


def test_binding_covers_header():
    from scp_amd import capi
    assert sorted(capi.SYMBOLS) == declared_symbols()
    assert capi.lib().scp_abi_version() == capi.ABI_VERSION


def test_product_path_fails_loudly_without_gpu_tensors():
    """no CPU fallback: CPU tensors are rejected like the reference's CHECK_CUDA does"""
    import torch
    from scp_amd.soft_renderer import functional as srf
    fv = torch.rand(1, 4, 3, 3)
    tex = torch.rand(1, 4, 3, 3)
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(RuntimeError):
        srf.soft_rasterize(fv, tex, 32, texture_type="vertex")
    # ... and so does every other HIP-backed operator of the step
    from scp_amd import dino, mesh, ops
    with pytest.raises(RuntimeError):
        ops.cols_softargmax(torch.rand(1, 4, 5), None, None, torch.rand(2, 4), 10.)
    with pytest.raises(RuntimeError):
        ops.feature_vertex_match(torch.rand(1, 3, 4), torch.rand(1, 5, 3), torch.ones(1, 4), torch.rand(1, 5, 3),
                                 torch.rand(2, 4), 10., 10.)
    with pytest.raises(RuntimeError):
        dino.fused_attention(torch.rand(1, 8, 192), 1, 8, 1, 64, 0.125)
    with pytest.raises(RuntimeError):
        dino.add_layernorm(torch.rand(4, 8), None, torch.nn.LayerNorm(8))
    with pytest.raises(RuntimeError):
        mesh.nearest_sq_dist(torch.rand(1, 4, 3), torch.rand(1, 6, 3))


def test_product_does_not_reference_the_oracle():
    """the shipped package must not import or call anything under oracle/"""
    pkg = os.path.join(ROOT, "self-corr-pose_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, os.path.join(dirpath, f)


def test_bench_and_entry_points_use_the_oracle_only_where_allowed():
    """bench.py may touch oracle/ only in its baseline leg (cpu_baseline / reference_kernels_same_gpu, both after the timed
    region) and must not import the test tree; __graft_entry__ only in smoke() and when BUILDING the checker"""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    allowed = {"cpu_baseline", "reference_kernels_same_gpu", "bench_posefit"}
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                names = [a.name for a in node.names] + ([node.module] if isinstance(node, ast.ImportFrom) and node.module else [])
                if any(n == "oracle" or n.startswith("oracle.") for n in names):
                    assert fn.name in allowed, "bench.py:%s imports the oracle" % fn.name
    for node in tree.body:                      # module level: nothing from oracle/ or tests/
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            names = [a.name for a in node.names] + ([node.module] if isinstance(node, ast.ImportFrom) and node.module else [])
            assert not any(n.split(".")[0] in ("oracle", "tests", "scenes", "synth", "step_case") for n in names), names
    assert 'os.path.join(ROOT, "tests")' not in src and "'tests'" not in src


def test_convolution_launch_plans_host_side():
    """host logic behind the convolution entry points (no launch): which layers of the encoder at B = 32 split K on the split
    main loop, the workspace that implies, and the row tile of the BatchNorm partial sums the caller has to size for"""
    from scp_amd import capi
    L = capi.lib()
    tiles, rows = ctypes.c_int(), ctypes.c_int()

    def plan(h, cin, cout, k=3, stride=1, split=1, n=32):
        L.scp_conv_nhwc_partial_rows(n, h, h, cin, cout, k, stride, split, ctypes.byref(tiles), ctypes.byref(rows))
        return L.scp_conv_nhwc_splitk_workspace(n, h, h, cin, cout, k, stride, split), tiles.value, rows.value

    # layer1 (64 x 64, 64 ch): 256 x 64 tiles, no split; layer2 (32 x 32, 128 ch): 128 x 128 tiles fill the machine, no split
    assert plan(64, 64, 64) == (0, 32 * 64 * 64 // 256, 256)
    assert plan(32, 128, 128) == (0, 32 * 32 * 32 // 128, 128)
    # layer3 (16 x 16, 256 ch, K = 2304): two workgroups per 128 x 128 tile; layer4 (8 x 8, 512 ch, K = 4608): four.  The partial
    # tiles are ksplit x M x Cout floats and the statistics come from the fold kernel in 32-row tiles
    m3, m4 = 32 * 16 * 16, 32 * 8 * 8
    assert plan(16, 256, 256) == (2 * m3 * 256 * 4, m3 // 32, 32)
    assert plan(8, 512, 512) == (4 * m4 * 512 * 4, m4 // 32, 32)
    # the stride-2 layers' K (1152 / 2304) is too short to split (>= 64 chunks per workgroup), 1 x 1 projections never split
    assert plan(32, 128, 256, stride=2)[0] == 0 and plan(16, 256, 512, stride=2)[0] == 0
    assert plan(32, 128, 256, k=1, stride=2)[0] == 0
    # the fp32 main loop never splits K
    assert plan(8, 512, 512, split=0)[0] == 0 and plan(16, 256, 256, split=0)[2] == 64
    # the weight-gradient workspace: ~256 workgroups' partial [Cout, 9, Cin] blocks
    ws = L.scp_conv_nhwc_weight_grad_workspace(32, 64, 64, 64, 64, 3, 1)
    assert ws > 0 and ws % (64 * 9 * 64 * 4) == 0 and 200 <= ws // (64 * 9 * 64 * 4) <= 256
    # stride 2 (split core): 3x3 and 1x1 on even maps with a power-of-two output map; 16 x 16 output here
    ws2 = L.scp_conv_nhwc_weight_grad_workspace(32, 32, 32, 128, 256, 3, 2)
    assert ws2 > 0 and ws2 % (256 * 9 * 128 * 4) == 0
    ws1 = L.scp_conv_nhwc_weight_grad_workspace(32, 32, 32, 128, 256, 1, 2)
    assert ws1 > 0 and ws1 * 9 == ws2
    assert L.scp_conv_nhwc_weight_grad_workspace(32, 30, 30, 128, 256, 3, 2) == 0           # 15 x 15 output: not a power of two
    assert L.scp_conv_nhwc_weight_grad_workspace(32, 32, 32, 128, 256, 1, 1) * 9 == L.scp_conv_nhwc_weight_grad_workspace(32, 32, 32, 128, 256, 3, 1)
    assert L.scp_stem_conv_tiles(32, 256, 256) == 32 * 128 and L.scp_stem_conv_tiles(2, 512, 512) == 2 * 256 * 2
    assert L.scp_stem_conv_weight_grad_workspace(32, 256, 256) == 512 * 64 * 160 * 4
    # the attention's operand planes + tail-query records
    n_pad = 1056
    assert L.scp_vit_attention_split_workspace(32, 1025, 6) == 9 * 32 * 6 * n_pad * 64 * 2 + 32 * 6 * 8 * 8 * 66 * 4


# ---- the static half of the co-residency rule: what the shipped code objects may contain ----------------------------
def _census(path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("packed_census", os.path.join(ROOT, "tools", "packed_census.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.census(path)


def _build_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("scp_build", os.path.join(ROOT, "self-corr-pose_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs llvm-objdump")
def test_no_kernel_carries_the_erratum_form_of_packed_fp32():
    """gfx950: v_pk_{mul,add,fma}_f32 with op_sel [0,1] returns a wrong low half beside a K-doubled 16-bit MFMA (csrc/selftest.hip,
    profiles/r05_packed_fp32_erratum.txt).  No kernel of the shipped library may contain it -- nor any other low-half selection on a
    packed fp32 instruction (none is needed; zero is the easiest number to keep) -- except the self-test that provokes it."""
    from scp_amd import capi
    c = _census(capi.LIB_PATH)
    assert len(c) > 150, len(c)                                   # the disassembly saw the library's kernels
    assert any(k["mfma16"] for k in c.values())                   # ... and recognises the matrix instructions
    selftest = [n for n in c if "packed_fp32_selftest_kernel" in n]
    assert selftest and all(c[n]["bad"] > 0 for n in selftest if "ILi0E" in n), "the detector must see the self-test's erratum form"
    offenders = {n: k for n, k in c.items() if (k["bad"] or k["op_sel"]) and n not in selftest}
    assert not offenders, offenders


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs llvm-objdump")
def test_files_that_are_not_bf16_gemms_are_built_without_packed_fp32():
    """build.py NO_PACKED: the back end cannot emit packed fp32 for them at all (defence in depth behind the op_sel rule)"""
    b = _build_module()
    b.build(verbose=False)
    for src in b.sources():
        if src in b.GEMM_FILES:
            continue
        obj = os.path.join(b.HERE, "build", src[:-4] + ".o")
        c = _census(obj)
        packed = {n: k["packed"] for n, k in c.items() if k["packed"]}
        assert not packed, (src, packed)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs llvm-objdump")
def test_the_positive_control_library_does_carry_it():
    """libscp_hip_slpctl.so = the rasteriser as it was compiled until round 4; tests/test_coresidency_gpu.py needs it to FAIL"""
    b = _build_module()
    c = _census(b.build_slp_control())
    hit = [n for n, k in c.items() if k["bad"] and ("raster_" in n or "face_setup" in n)]
    assert any("face_setup" in n for n in hit) and any("raster_forward" in n for n in hit), hit


def test_library_gemm_scan_is_current_and_clean():
    """static half of the co-residency rule for the LIBRARY GEMMs: the committed scan of the fp32 rocBLAS / hipBLASLt gfx950
    code objects (tools/library_gemm_scan.py -> profiles/r06_library_gemm_scan.txt) names no carrier of the erratum form except rocBLAS's
    complex-single PostGSU helper, its stamp describes THIS installation (so scp_amd.streams defaults to `overlap` here), and a stamp
    that does not match flips the default to `serial`"""
    import json
    import sys
    import warnings
    report = open(os.path.join(ROOT, "profiles", "r06_library_gemm_scan.txt")).read()
    carriers = [l.split() for l in report.splitlines() if l.strip().startswith("ERRATUM-FORM")]
    assert [c[2] for c in carriers] == ["Cijk_C_PostGSU"], carriers             # complex single: no fp32 GEMM launches it
    assert "20169 kernels in 30 code objects" in report
    sys.path.insert(0, os.path.join(ROOT, "self-corr-pose_amd"))
    from scp_amd import streams
    ok, why = streams.stamp_status()
    assert ok, why
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import library_gemm_scan
    want = json.load(open(streams.STAMP))
    now = library_gemm_scan.stamp()
    assert {k: want[k] for k in now} == now
    # an installation the stamp does not describe: serial by default, with a warning that says why
    saved = streams.STAMP
    try:
        streams.STAMP = os.path.join(ROOT, "no_such_stamp.json")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert streams._default_mode() == "serial"
        assert any("serial" in str(x.message) for x in w)
    finally:
        streams.STAMP = saved
    assert streams._default_mode() == "overlap"
