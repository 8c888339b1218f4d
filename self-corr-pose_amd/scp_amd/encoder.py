"""scp_amd/encoder.py -- image / mesh encoder of the step (model/module/encoder.py:13-52).

Stock PyTorch-ROCm (MIOpen convolutions) by design: north_star keeps everything but the three hot
operators on PyTorch.  Called twice per step (here and from the rotation-cycle loss)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import imgops
from .nets import MeshEncoder, PosePredictor, ResNet_Decoder, ResNet_Encoder, ShapePredictor


class Encoder(nn.Module):
    supports_half_res = True       # encode_img(img, half_res=True), used by Correspondence.compute_rotation_cycle_loss

    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        self.resnet_transform = imgops.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
        self.random_jitter = imgops.ColorJitter(0.2, 0.2, 0.2, 0.05)
        self.backbone = ResNet_Encoder(getattr(opts, "resnet18_path", None), will_load_checkpoint=bool(getattr(opts, "model_path", "")))
        self.featnet = ResNet_Decoder(is_proj=True, out_channel=opts.n_corr_feat,
                                      downsample=opts.img_size // opts.corr_h)
        self.featnet_mesh = MeshEncoder(opts.n_corr_feat)
        self.shape_code_predictor = nn.Linear(512, opts.codedim)
        self.shape_predictor = ShapePredictor(opts)
        self.pose_predictor = PosePredictor(opts, 512)

    def _normalized(self, img):
        # The trunk runs NHWC end to end, but its first layer -- the own 7x7 stem kernel (csrc/conv_stem.hip) -- reads the 3-channel image as
        # NCHW rows and writes NHWC: when it will run, the normalised image is produced NCHW and never converted (it used to be written
        # NHWC and copied back, 2 x 25 MB per pass at B = 32); otherwise (stock stem: bf16 autocast, eval-mode BatchNorm, ...) NHWC.
        from .fused_conv import stem_takes_own_kernels
        r = self.backbone.resnet
        # asked with the metadata of the tensor the stem WILL see (fp32 output of jitter_normalize, the image's own requires_grad, the
        # autocast state the stem will run under: opts.mixed_bf16's or one the caller opened) -- nothing is converted to ask
        nchw = img.is_cuda and stem_takes_own_kernels(
            None, r.conv1, r.bn1, autocast=bool(getattr(self.opts, "mixed_bf16", False)) or torch.is_autocast_enabled(),
            meta=(tuple(img.shape), torch.float32, True, bool(img.requires_grad and torch.is_grad_enabled())))
        x = imgops.jitter_normalize(img, self.random_jitter, self.resnet_transform, channels_last=not nchw)
        if x.is_cuda and not nchw:
            x = x.contiguous(memory_format=torch.channels_last)
        return x

    def _features(self, x, half_res=False):
        """normalised NHWC image -> (img_code [B,512], unit-norm features [B,C,P])"""
        # BASELINE configs[4]: convolutions on the bf16 matrix cores, everything after them in fp32
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(getattr(self.opts, "mixed_bf16", False)) and x.is_cuda):
            c2, c3, c4, c5 = self.backbone(x)
            feat = self.featnet(c2, c3, c4, c5, half_res=half_res)
        img_code = c5.float().mean((2, 3))
        feat = feat.float().contiguous().reshape(x.shape[0], self.opts.n_corr_feat, -1)
        return img_code, F.normalize(feat, 2, 1)

    def encode_img(self, img, half_res=False):
        """half_res: features at the even pixels of the feature map only (nets.ResNet_Decoder.forward)"""
        return self._features(self._normalized(img), half_res)

    def forward(self, img, mean_v, pp_crop, foc_crop):
        return self._heads_from_normalized(self._normalized(img), mean_v, pp_crop, foc_crop)

    def _heads_from_normalized(self, x, mean_v, pp_crop, foc_crop):
        img_code, img_feat = self._features(x)
        pred_v = self.shape_predictor(mean_v, self.shape_code_predictor(img_code))
        mesh_feat = F.normalize(self.featnet_mesh(pred_v.detach()), 2, -1)
        rotation, trans, scale = self.pose_predictor(img_code)
        pred_v = pred_v * scale[:, None]
        # principal-point compensation of the in-plane translation (encoder.py:49)
        # (in place in the reference: evaluated in the promoted dtype -- the loader's intrinsics are float64 -- and rounded
        # back to the translation's)
        xy = (trans[:, :2] - (pp_crop / foc_crop) * trans[:, 2:].detach()).to(trans.dtype)
        translation = torch.cat((xy, trans[:, 2:]), 1)
        return img_feat, mesh_feat, pred_v, rotation.reshape(-1, 3, 3), translation.reshape(-1, 1, 3), scale
