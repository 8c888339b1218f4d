"""scp_amd/nets.py -- the networks around the hot path.  The image encoder's 3x3 / 1x1 convolutions, BatchNorms and
activations run on the build's own kernels (scp_amd/fused_conv.py, scp_amd/fused_bn.py); the small heads stay
stock PyTorch.  Module and parameter NAMES follow the reference so that its checkpoints
load and its optimiser's name-based parameter groups (optimizers.py:16-35) keep working.

Restated from (reference file:line):
  torchvision resnet18 (un-vendored; standard topology, torchvision key names) as used by
      model/module/network/image_encoder.py:119-139 (ResNet_Encoder; fc dropped)
  image_encoder.py:141-193   ResNet_Decoder (conv+LeakyReLU(0.1) units, bilinear 2x upsampling)
  network/net_blocks.py:336-359  conv2DBatchNormRelu (with_bn=False as instantiated)
  network/mesh_encoder.py:6-39   STN3d_noBN + MeshEncoder
  network/pose_predictor.py:22-83  PosePredictor (6D rotation, Gram-Schmidt, offsets)
  network/shape_predictor.py:12-43 + third-party/nerf/models.py:336-417, train_utils.py:9-33
      ShapePredictor = conditional MLP on (vertex, shape code)
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .fused_bn import bn_act, maxpool3x3s2
from .fused_conv import conv_bias_leaky, conv_bn_act, stem_conv_bn_act


# ------------------------------------------------------------------------------------------------
# ResNet18 trunk (torchvision key names: conv1, bn1, layer{1..4}.{0,1}.{conv1,bn1,conv2,bn2,downsample})
# ------------------------------------------------------------------------------------------------
class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        # convolution + BatchNorm (+ skip) (+ ReLU) as one op on the own kernels (scp_amd/fused_conv.py); CPU / eval / bf16 take
        # the stock composition inside
        skip = x if self.downsample is None else conv_bn_act(x, self.downsample[0], self.downsample[1])
        y = conv_bn_act(x, self.conv1, self.bn1, relu=True)
        return conv_bn_act(y, self.conv2, self.bn2, skip=skip, relu=True)


class ResNet18Trunk(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        widths = [64, 128, 256, 512]
        cin = 64
        for i, w in enumerate(widths):
            stride = 1 if i == 0 else 2
            setattr(self, "layer%d" % (i + 1), nn.Sequential(_BasicBlock(cin, w, stride), _BasicBlock(w, w, 1)))
            cin = w
        self.fc = None
        for m in self.modules():  # torchvision's default init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


def load_imagenet_resnet18(trunk, path):
    """torchvision's resnet18 state_dict (e.g. resnet18-f37072fd.pth; the key names are the trunk's) -> trunk; `fc.*` dropped
    (image_encoder.py:122-123 builds resnet18(pretrained=True) and sets fc = None)"""
    sd = torch.load(path, map_location="cpu")
    sd = sd.get("state_dict", sd)
    sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}
    missing, unexpected = trunk.load_state_dict(sd, strict=False)
    missing = [k for k in missing if not k.endswith("num_batches_tracked")]
    if missing or unexpected:
        raise RuntimeError("%s is not a torchvision resnet18 state_dict: missing %s unexpected %s" % (path, missing[:5], unexpected[:5]))


class ResNet_Encoder(nn.Module):
    """returns the four pyramid levels (1/4 .. 1/32 resolution: 64, 128, 256, 512 channels).

    The reference starts from ImageNet weights (image_encoder.py:122 `resnet18(pretrained=True)`, downloaded by
    torchvision).  There is no network here, so the weights are read from `weights_path` (a torchvision resnet18
    state_dict on disk).  Like DINO (scp_amd/dino.py) a missing file is an ERROR -- a silent kaiming re-initialisation
    would depart from the reference recipe (its BatchNorm affine parameters are frozen, so gamma=1 / beta=0 would stay
    forever) -- unless a full checkpoint is going to be loaded over it (`will_load_checkpoint`) or synthetic weights were
    asked for explicitly (scp_amd.dino.ALLOW_RANDOM_INIT: bench, tests)."""

    def __init__(self, weights_path=None, will_load_checkpoint=False):
        super().__init__()
        self.resnet = ResNet18Trunk()
        import os
        from . import dino
        if weights_path and os.path.exists(weights_path):
            load_imagenet_resnet18(self.resnet, weights_path)
        elif not (will_load_checkpoint or dino.ALLOW_RANDOM_INIT):
            raise FileNotFoundError(
                "ImageNet resnet18 weights not found at %r (flag resnet18_path; torchvision's resnet18-f37072fd.pth).  Pass "
                "model_path to load a full checkpoint, or set scp_amd.dino.ALLOW_RANDOM_INIT for synthetic weights" % (weights_path,))

    def forward(self, x):
        r = self.resnet
        x = maxpool3x3s2(stem_conv_bn_act(x, r.conv1, r.bn1, relu=True), r.maxpool)
        c2 = r.layer1(x)
        c3 = r.layer2(c2)
        c4 = r.layer3(c3)
        c5 = r.layer4(c4)
        return c2, c3, c4, c5


class _ConvUnit(nn.Module):
    """3x3 conv + LeakyReLU(0.1); parameters live under `.cbr_unit.0` like the reference's"""

    def __init__(self, cin, cout):
        super().__init__()
        self.cbr_unit = nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=True), nn.LeakyReLU(0.1, inplace=True))

    def forward(self, x, stride=1):
        """stride 2: only every other output pixel is wanted -- the same weights as a strided convolution"""
        conv = self.cbr_unit[0]
        if x.is_cuda:
            return conv_bias_leaky(x, conv, 0.1, stride)       # own convolution, bias + LeakyReLU in its epilogue (scp_amd/fused_conv.py)
        if stride == 1:
            return self.cbr_unit(x)
        return F.leaky_relu(F.conv2d(x, conv.weight, conv.bias, stride, 1), 0.1)


class _Upsample2x(torch.autograd.Function):
    """exact-2x bilinear upsampling, NHWC fp32 (or bf16 in configs[4] precision), csrc/upsample.hip: forward (ATen's NHWC kernel
    runs at 0.7 TB/s here) and a gather backward instead of ATen's atomicAdd scatter -- deterministic and ~5x faster"""

    @staticmethod
    def forward(ctx, x):
        import ctypes
        from . import capi
        ctx.shape = x.shape
        n, c, h, w = x.shape
        out = torch.empty((n, c, 2 * h, 2 * w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        fn = capi.lib().scp_upsample2x_bilinear_forward if x.dtype == torch.float32 else capi.lib().scp_upsample2x_bilinear_forward_bf16
        capi.check(fn(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), n, h, w, c, capi.current_stream()),
                   "upsample2x_bilinear_forward")
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        from . import capi
        n, c, h, w = ctx.shape
        g = g if g.is_contiguous(memory_format=torch.channels_last) else g.contiguous(memory_format=torch.channels_last)
        gin = torch.empty(ctx.shape, dtype=g.dtype, device=g.device, memory_format=torch.channels_last)
        fn = capi.lib().scp_upsample2x_bilinear_backward if g.dtype == torch.float32 else capi.lib().scp_upsample2x_bilinear_backward_bf16
        capi.check(fn(ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(gin.data_ptr()), n, h, w, c, capi.current_stream()),
                   "upsample2x_bilinear_backward")
        return gin


class ResNet_Decoder(nn.Module):
    def __init__(self, is_proj=True, out_channel=64, downsample=4):
        super().__init__()
        self.is_proj, self.downsample = is_proj, downsample
        self.upconv5, self.iconv4 = _ConvUnit(512, 256), _ConvUnit(512, 256)
        self.upconv4, self.iconv3 = _ConvUnit(256, 128), _ConvUnit(256, 128)
        self.upconv3, self.iconv2 = _ConvUnit(128, 64), _ConvUnit(128, 64)
        if is_proj:
            self.proj = nn.Conv2d(64 if downsample == 4 else 128, out_channel, 1)

    @staticmethod
    def _up(x, like):
        h, w = like.shape[2:]
        if (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.shape[2] * 2 == h and x.shape[3] * 2 == w and x.shape[1] % 4 == 0
                and x.is_contiguous(memory_format=torch.channels_last)):
            return _Upsample2x.apply(x)
        return F.interpolate(x, like.shape[2:], mode="bilinear", align_corners=False)

    def forward(self, c2, c3, c4, c5, half_res=False):
        """half_res: return the feature map at its even pixels only ([.., ::2, ::2] of the full result, exactly): the last
        3x3 unit runs as a stride-2 convolution and the projection on a quarter of the pixels.  The rotation-cycle loss
        (correspondence.py:91-96) subsamples the rotated images' features this way and uses nothing else of them."""
        c4 = self.iconv4(torch.cat((c4, self.upconv5(self._up(c5, c4))), 1))
        c3 = self.iconv3(torch.cat((c3, self.upconv4(self._up(c4, c3))), 1))
        if self.downsample != 4:
            feat = c3[:, :, ::2, ::2] if half_res else c3
            return self._project(feat) if self.is_proj else feat
        c2 = self.iconv2(torch.cat((c2, self.upconv3(self._up(c3, c2))), 1), stride=2 if half_res else 1)
        return self._project(c2) if self.is_proj else c2

    def _project(self, feat):
        """the 1x1 feature projection (a convolution with bias, no activation): on the GPU the own 1x1 kernel with the bias in its
        epilogue -- LeakyReLU with slope 1 is the identity, exactly -- instead of a library GEMM + bias pass (forward and both
        gradients)"""
        if feat.is_cuda:
            return conv_bias_leaky(feat, self.proj, 1.0)
        return self.proj(feat)


# ------------------------------------------------------------------------------------------------
# mesh feature net
# ------------------------------------------------------------------------------------------------
class STN3d_noBN(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv1d(3, 128, 1)
        self.fc = nn.Linear(128, 9)

    def forward(self, x):  # x: b,3,n
        h = F.relu(self.conv1(x)).max(2)[0]
        eye = torch.eye(3, device=x.device, dtype=x.dtype).reshape(1, 9)
        return (self.fc(h) + eye).view(-1, 3, 3)


class MeshEncoder(nn.Module):
    def __init__(self, n_feat):
        super().__init__()
        self.stn = STN3d_noBN()
        self.conv1 = nn.Conv1d(3, n_feat, 1)

    def forward(self, x):  # b,n,3 -> b,n,c
        trans = self.stn(x.transpose(2, 1))
        x = torch.bmm(x, trans).transpose(2, 1)
        return F.relu(self.conv1(x)).transpose(2, 1)


# ------------------------------------------------------------------------------------------------
# pose head
# ------------------------------------------------------------------------------------------------
def _fc_stack(cin, cout, n):
    layers = []
    for _ in range(n):
        layers.append(nn.Sequential(nn.Linear(cin, cout), nn.LeakyReLU(0.1, inplace=True)))
        cin = cout
    net = nn.Sequential(*layers)
    for m in net.modules():
        if isinstance(m, nn.Linear):
            m.weight.data.normal_(0, 0.02)
            m.bias.data.zero_()
    return net


class PosePredictor(nn.Module):
    def __init__(self, opts, nc_input):
        super().__init__()
        self.offset = opts.depth_offset
        self.use_scale = opts.use_scale
        self.n_hypo = opts.num_multipose_az * opts.num_multipose_el
        assert self.n_hypo == 1
        self.rot_pred_layer = nn.Sequential(_fc_stack(nc_input, 128, 3), nn.Linear(128, 6 * self.n_hypo))
        self.trans_pred_layer = nn.Linear(nc_input, 3 * self.n_hypo)
        if self.use_scale:
            self.scale_pred_layer = nn.Linear(nc_input, 3 * self.n_hypo)
        r_off = [float(r) for r in opts.rotation_offset]
        self.x_offset = nn.Parameter(torch.tensor([r_off[:3]]), requires_grad=False)
        self.y_offset = nn.Parameter(torch.tensor([r_off[3:]]), requires_grad=False)

    def forward(self, feat):
        n = feat.shape[0] * self.n_hypo
        rot6 = self.rot_pred_layer(feat).reshape(n, 6)
        x = F.normalize(rot6[:, :3] + self.x_offset)
        y = rot6[:, 3:] + self.y_offset
        z = F.normalize(torch.cross(x, y, dim=1))
        y = F.normalize(torch.cross(z, x, dim=1))
        rot = torch.stack((x, y, z), 2)
        t = self.trans_pred_layer(feat).reshape(n, 3)
        trans = torch.cat((t[:, :2] * 0.1, t[:, 2:] + self.offset), 1)
        if self.use_scale:
            scale = self.scale_pred_layer(feat).reshape(n, 3) * 0.1 + 1.
        else:
            scale = torch.ones((n, 3), device=feat.device, dtype=feat.dtype)
        return rot, trans, scale


# ------------------------------------------------------------------------------------------------
# shape head: conditional MLP  (vertex xyz ++ shape code) -> xyz offset
# ------------------------------------------------------------------------------------------------
class CondNeRFModel(nn.Module):
    """the 2-layer configuration the reference instantiates (shape_predictor.py:15-24): only the
    layers that exist for num_layers=2, no positional encoding, no view directions"""

    def __init__(self, codesize, hidden_size=256, out_channel=3):
        super().__init__()
        self.codesize = codesize
        self.layer1 = nn.Linear(3 + codesize, hidden_size)
        self.layers_xyz = nn.ModuleList([nn.Linear(hidden_size, hidden_size)])
        self.layers_dir = nn.ModuleList([nn.Linear(hidden_size, hidden_size // 2)])
        self.fc_alpha = nn.Linear(hidden_size, 1)   # present in checkpoints, never used for the shape
        self.fc_rgb = nn.Linear(hidden_size // 2, out_channel)
        self.fc_feat = nn.Linear(hidden_size, hidden_size)

    def forward(self, x):  # [..., 3 + codesize] -> [..., out_channel]
        h = self.layer1(x)                       # (no activation after layer1 in the reference)
        h = F.relu(self.layers_xyz[0](h))
        h = F.relu(self.fc_feat(h))
        h = F.relu(self.layers_dir[0](h))
        return self.fc_rgb(h)


class ShapePredictor(nn.Module):
    def __init__(self, opts):
        super().__init__()
        self.shapenerf = CondNeRFModel(codesize=opts.codedim)
        self.no_deform, self.deform_ratio = opts.no_deform, opts.deform_ratio

    def forward(self, mean_v, shape_code):
        if self.no_deform:
            return mean_v
        code = shape_code[:, None].expand(-1, mean_v.shape[1], -1)
        delta = self.shapenerf(torch.cat((mean_v.detach(), code), -1))
        delta = delta - delta.mean(1, keepdim=True)
        return mean_v + delta * self.deform_ratio
