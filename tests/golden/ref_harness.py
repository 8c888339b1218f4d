"""tests/golden/ref_harness.py -- fixture GENERATOR support, runs only in the build container.

Makes the reference (/root/reference, read-only, never copied) importable and runnable on CPU so
that tests/golden/make_golden.py can record input/output vectors from the reference itself:

  * the three SoftRas kernel bodies (soft_rasterize_cuda_kernel.cu lines 22-671: device helpers +
    templated kernels on raw pointers) are read from /root/reference at run time into a temp dir
    and host-compiled UNCHANGED behind ref_shim.hpp (CUDA keyword/builtin stand-ins: empty
    __global__/__device__, thread-index variables, sequential atomicAdd, CUDA's mixed float/double
    min/max overloads, float overloads of exp/sqrt/pow).  g++ -O2, no -mfma, -ffp-contract=off:
    the recorded vectors pin the UN-CONTRACTED fp32 semantics.
  * absl/torchvision/cv2/trimesh/kornia/pytorch3d/skimage/imageio/tensorboard are stubbed so the
    reference's own Python (soft_renderer package, model/*) imports; `.cuda()` is a no-op.

Nothing here is imported by the product, by `-m gpu` tests, by smoke() or by bench.py, and none of
it can run on the GPU box (there is no /root/reference there).  The .so it builds lives in a temp
directory and is not shipped.
"""
import ctypes
import os
import subprocess
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
_KERNEL = os.path.join(REF, "third-party/softras/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu")


# ------------------------------------------------------------------------------------------------
# host build of the reference kernel bodies
# ------------------------------------------------------------------------------------------------
def build_ref_kernels(contract=False):
    work = os.path.join(tempfile.gettempdir(), "scp_ref_build" + ("_fma" if contract else ""))
    os.makedirs(work, exist_ok=True)
    with open(_KERNEL) as f:
        lines = f.readlines()
    body = "".join(lines[21:671])  # the anonymous namespace, lines 22..671
    with open(os.path.join(work, "kernel_body.inc"), "w") as f:
        f.write(body)
    so = os.path.join(work, "libref_softras.so")
    flags = ["-O2", "-std=c++17", "-shared", "-fPIC", "-I", work, "-I", _HERE]
    flags += ["-mfma", "-ffp-contract=fast"] if contract else ["-ffp-contract=off"]
    subprocess.check_call(["g++"] + flags + [os.path.join(_HERE, "ref_shim.cpp"), "-o", so])
    lib = ctypes.CDLL(so)
    fp = ctypes.c_void_p
    i, f32, b = ctypes.c_int, ctypes.c_float, ctypes.c_int
    scal = [i, i, i, i, i, f32, f32, f32, f32, i, f32, f32, i, i, i, b]
    lib.ref_forward.argtypes = [fp] * 5 + scal
    lib.ref_backward.argtypes = [fp] * 8 + scal
    return lib


class _RefRasterizeModule(types.ModuleType):
    """Stands where soft_renderer.cuda.soft_rasterize (the pybind module, cpp:135-138) would."""

    def __init__(self, lib):
        super().__init__("soft_renderer.cuda.soft_rasterize")
        self._lib = lib

    @staticmethod
    def _scal(faces, textures, image_size, near, far, eps, sigma_val, func_id_dist, dist_eps,
              gamma_val, func_id_rgb, func_id_alpha, texture_sample_type, double_side):
        B, F = faces.shape[:2]
        T = textures.shape[2]
        return [B, F, image_size, T, int(np.sqrt(T)), near, far, eps, sigma_val, func_id_dist,
                dist_eps, gamma_val, func_id_rgb, func_id_alpha, texture_sample_type,
                int(bool(double_side))]

    def forward_soft_rasterize(self, faces, textures, faces_info, aggrs_info, soft_colors, *scal):
        for t in (faces, textures, faces_info, aggrs_info, soft_colors):
            assert t.is_contiguous() and t.dtype == torch.float32
        self._lib.ref_forward(faces.data_ptr(), textures.data_ptr(), faces_info.data_ptr(),
                              aggrs_info.data_ptr(), soft_colors.data_ptr(),
                              *self._scal(faces, textures, *scal))
        return [faces_info, aggrs_info, soft_colors]

    def backward_soft_rasterize(self, faces, textures, soft_colors, faces_info, aggrs_info,
                                grad_faces, grad_textures, grad_soft_colors, *scal):
        for t in (faces, textures, soft_colors, faces_info, aggrs_info, grad_faces, grad_textures,
                  grad_soft_colors):
            assert t.is_contiguous() and t.dtype == torch.float32
        self._lib.ref_backward(faces.data_ptr(), textures.data_ptr(), soft_colors.data_ptr(),
                               faces_info.data_ptr(), aggrs_info.data_ptr(), grad_faces.data_ptr(),
                               grad_textures.data_ptr(), grad_soft_colors.data_ptr(),
                               *self._scal(faces, textures, *scal))
        return [grad_faces, grad_textures]


# ------------------------------------------------------------------------------------------------
# stubs for the un-vendored third-party packages
# ------------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Flags:
    """absl.flags stand-in: DEFINE_* just records the default as an attribute of FLAGS."""

    def __init__(self):
        self.__dict__["_vals"] = {}

    def __getattr__(self, k):
        try:
            return self.__dict__["_vals"][k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self.__dict__["_vals"][k] = v

    def flags_into_string(self):
        return "\n".join("--%s=%s" % kv for kv in sorted(self._vals.items()))


def read_obj(path):
    vs, fs = [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                vs.append([float(x) for x in p[1:4]])
            elif p[0] == "f":
                fs.append([int(x.split("/")[0]) - 1 for x in p[1:4]])
    return np.asarray(vs, np.float64), np.asarray(fs, np.int64)


PINNED_ANGLE = [None]   # when set, every rotate() call uses this angle (the reference draws it from the host RNG)


def _rot90_exact(img, angle, interpolation=None, **kw):
    """torchvision.transforms.functional.rotate stand-in: exact for multiples of 90 degrees
    (counter-clockwise, like torchvision); golden runs pin the angle to such a value."""
    if PINNED_ANGLE[0] is not None:
        angle = PINNED_ANGLE[0]
    assert abs(angle - 90.0 * round(angle / 90.0)) < 1e-6, "golden runs must use multiples of 90"
    return torch.rot90(img, int(round(angle / 90.0)) % 4, dims=(-2, -1))


def _make_resnet18_factory():
    import torch.nn as nn

    class BasicBlock(nn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(cout)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(cout)
            self.downsample = None
            if stride != 1 or cin != cout:
                self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False),
                                                nn.BatchNorm2d(cout))

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            y = self.relu(self.bn1(self.conv1(x)))
            y = self.bn2(self.conv2(y))
            return self.relu(y + idt)

    class ResNet18(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            self.layer1 = nn.Sequential(BasicBlock(64, 64, 1), BasicBlock(64, 64, 1))
            self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128, 1))
            self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256, 1))
            self.layer4 = nn.Sequential(BasicBlock(256, 512, 2), BasicBlock(512, 512, 1))
            self.avgpool = nn.AdaptiveAvgPool2d(1)
            self.fc = nn.Linear(512, 1000)

    def resnet18(pretrained=False, **kw):
        return ResNet18()

    return resnet18


class _SamplePointsHook:
    """pytorch3d.ops.sample_points_from_meshes stand-in: the generator injects the barycentric
    sample (face ids + weights) so that the same sample can be replayed by the build."""
    face_idx = None   # [N, P] long
    bary = None       # [N, P, 3]

    @classmethod
    def __call__(cls, meshes, num_samples, return_normals=False):
        verts, faces = meshes._v, meshes._f
        N = verts.shape[0]
        fi = cls.face_idx[:N]
        tri = torch.gather(faces, 1, fi[..., None].expand(-1, -1, 3))          # N,P,3
        pts = torch.gather(verts, 1, tri.reshape(N, -1)[..., None].expand(-1, -1, 3))
        pts = pts.reshape(N, -1, 3, 3)
        out = (pts * cls.bary[:N, :, :, None]).sum(2)
        return (out, None) if return_normals else out


def _knn_points(x, y, lengths1=None, lengths2=None, K=1):
    d = torch.cdist(x, y) ** 2
    dist, idx = d.min(-1)
    return types.SimpleNamespace(dists=dist[..., None], idx=idx[..., None])


def install(contract=False):
    """Install every stub + the host-built rasteriser, extend sys.path like trainer.py:7 does."""
    lib = build_ref_kernels(contract)

    flags = _Flags()

    def _define(name, default, *a, **k):
        setattr(flags, name, default)

    fl = _mod("absl.flags", FLAGS=flags, DEFINE_bool=_define, DEFINE_boolean=_define,
              DEFINE_integer=_define, DEFINE_float=_define, DEFINE_string=_define,
              DEFINE_list=_define, DEFINE_enum=_define)
    _mod("absl.app", run=lambda main: main(None))
    _mod("absl", flags=fl, app=sys.modules["absl.app"])

    class _IM:
        BILINEAR = "bilinear"
        NEAREST = "nearest"

    class _Normalize(torch.nn.Module):
        def __init__(self, mean, std):
            super().__init__()
            self.mean, self.std = mean, std

        def forward(self, x):
            m = torch.tensor(self.mean, dtype=x.dtype, device=x.device)[:, None, None]
            s = torch.tensor(self.std, dtype=x.dtype, device=x.device)[:, None, None]
            return (x - m) / s

    class _Identity(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x):
            return x

    tvf = _mod("torchvision.transforms.functional", rotate=_rot90_exact)
    tvt = _mod("torchvision.transforms", InterpolationMode=_IM, Normalize=_Normalize,
               ColorJitter=_Identity, ToTensor=_Identity, functional=tvf, Compose=_Identity,
               Resize=_Identity)
    tvm = _mod("torchvision.models", resnet18=_make_resnet18_factory())
    _mod("torchvision", transforms=tvt, models=tvm)

    _mod("cv2", sqrt=np.sqrt, circle=lambda *a, **k: None)

    class _Trimesh:
        def __init__(self, vertices=None, faces=None, **k):
            self.vertices, self.faces = vertices, faces

        def export(self, *a, **k):
            return None

    def _load_mesh(path, **k):
        v, f = read_obj(path)
        return _Trimesh(v, f)

    _mod("trimesh", load_mesh=_load_mesh, Trimesh=_Trimesh)

    def _quat_to_rot(q, order=None):
        # only ever called at construction for the single base quaternion; never read in forward
        return torch.eye(3)[None].repeat(q.shape[0], 1, 1)

    kg = _mod("kornia.geometry", quaternion_to_rotation_matrix=_quat_to_rot)
    _mod("kornia", geometry=kg)

    class _Meshes:
        def __init__(self, verts, faces):
            self._v, self._f = verts, faces

    knn = _mod("pytorch3d.ops.knn", knn_points=_knn_points, knn_gather=None)
    p3o = _mod("pytorch3d.ops", sample_points_from_meshes=_SamplePointsHook(), knn=knn,
               knn_points=_knn_points)
    pcl = _mod("pytorch3d.structures.pointclouds", Pointclouds=type("Pointclouds", (), {}))
    p3s = _mod("pytorch3d.structures", Meshes=_Meshes, pointclouds=pcl,
               Pointclouds=pcl.Pointclouds)
    p3l = _mod("pytorch3d.loss")
    _mod("pytorch3d", ops=p3o, structures=p3s, loss=p3l)

    ski = _mod("skimage.io", imread=None, imsave=None)
    _mod("skimage", io=ski)
    _mod("imageio")

    class _SW:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def add_image(self, *a, **k):
            pass

    _mod("torch.utils.tensorboard", SummaryWriter=_SW)

    cuda_pkg = _mod("soft_renderer.cuda")
    cuda_pkg.__path__ = []
    sys.modules["soft_renderer.cuda.soft_rasterize"] = _RefRasterizeModule(lib)
    cuda_pkg.soft_rasterize = sys.modules["soft_renderer.cuda.soft_rasterize"]
    for n in ("load_textures", "create_texture_image", "voxelization"):
        setattr(cuda_pkg, n, _mod("soft_renderer.cuda." + n))

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None

    for p in (REF, REF + "/third-party", REF + "/third-party/softras"):
        if p not in sys.path:
            sys.path.insert(0, p)
    return flags
