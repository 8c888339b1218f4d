"""scp_amd/flags.py -- the option set of the reference (its absl flags, restated as data).

The reference scatters ~95 `flags.DEFINE_*` over config.py and nine modules and reads them through
`opts = flags.FLAGS`.  absl is not available on the target image, so the same names / defaults are
kept in one table here and `Options` behaves like FLAGS for attribute reads.  `load_flagfile`
parses the reference's `--name=value` flag files (config/*/base_config.txt).

Defaults restated from (reference file:line): config.py:5-27, data/dataloader.py:18-29,
model/model.py:27-39, model/module/correspondence.py:11-18, pretrained_corr.py:13-14,
mesh.py:19-26, weights.py:5-17, network/pose_predictor.py:11-19, network/shape_predictor.py:9-10,
util/base_rot.py:8.
"""
import copy

DEFAULTS = dict(
    # not a reference flag: BASELINE configs[4] "mixed bf16" -- encoder convolutions and ViT linear layers under bf16
    # autocast; SoftRas, correspondence reductions, attention softmax, losses and the optimizer stay fp32
    mixed_bf16=False,
    # not a reference flag: where torchvision's ImageNet resnet18 state_dict lives on disk (the reference lets torchvision
    # download it, image_encoder.py:122; there is no network here).  Missing file = error, see nets.ResNet_Encoder.
    resnet18_path="pretrain/resnet18-f37072fd.pth",
    # model/tester.py:35-37 (the CUB evaluation and the visualisation flags are not provided)
    eval=False, eval_nocs=False,
    # config.py  (num_workers: the reference's default is 8 -- for a step of ~1 s; at 30+ it/s a rank consumes ~1 000 frames/s and a
    # decode worker delivers ~208, and eight ranks share one host: 12 workers per rank delivered 5.8 k img/s in aggregate, 24 gave
    # 23 k (profiles/r04_data_bench8.json), the 8 x 33 it/s node needs 8.5 k => 16 per rank by default)
    train=False, test=False, seed=0, ngpu=1, local_rank=0, num_workers=16, checkpoint_dir="log",
    name="exp", train_list="", test_list="", model_path="", vis_path="", total_iters=10000,
    batch_log_interval=10, save_freq=1, vis_freq=1, batch_size=4, dframe_eval=1, logger="tensorboard",
    # data/dataloader.py
    img_size=256, repeat=8, shuffle_test=False, no_stretch=False, use_occ=False, dataset_path="data",
    dataset_cache_path="data", test_dataset_path="data", dataset_name="Wild6D", category="bottle",
    # model/model.py
    feat_shape=False, flatten_loss=False, camera_loss=False, depth_loss_chamfer=False, use_depth=False,
    surface_texture=False, vert_lr_ratio=0.1, cam_lr_ratio=0.1, learning_rate=0.0001, n_tex_sample=6,
    nz_feat=128, codedim=16, n_corr_feat=16,
    # correspondence / pretrained_corr
    tau_img=10., tau_mesh=10., topk_img=100, topk_mesh=100, corr_h=32, corr_w=32, divide_fn="frame",
    pretrain_k=100,
    # mesh
    symmetry_idx=-1, init_scale=[1, 1, 1], shape_prior=False, shape_prior_path="", prior_deform=False,
    subdivide=3, n_faces=1280,
    # weights
    mask_wt=0.1, tex_wt=0.05, depth_wt=0.05, match_wt=0.01, imatch_wt=0.02, triangle_wt=0.001,
    pullfar_wt=0.001, deform_wt=0.05, symmetry_wt=1., camera_wt=0.005, cycle_loss_wt=0.2,
    cycle_loss_pretrain_wt=0.05, decay_ratio=1.,
    # pose / shape predictors
    use_scale=False, rotation_offset=[0, 0, 0, 0, 0, 0], depth_offset=10., initial_quat_bias_deg=0.,
    baseQuat_elevationBias=0., baseQuat_azimuthBias=0., num_multipose_az=1, num_multipose_el=1,
    no_deform=False, deform_ratio=1., base_rot=[1, 0, 0, 0, 1, 0, 0, 0, 1],
)

# values of config/laptop_wild6d/base_config.txt and config/bottle_wild6d/base_config.txt
# (dataset paths dropped; the prior mesh is supplied by the caller)
_WILD6D_COMMON = dict(
    dataset_name="Wild6D", total_iters=20000, batch_size=8, repeat=4, learning_rate=0.0001,
    depth_offset=5., codedim=64, n_corr_feat=64, corr_h=64, corr_w=64, subdivide=3, init_scale=[1, 1, 1],
    num_multipose_az=1, num_multipose_el=1, mask_wt=0.15, tex_wt=0.05, depth_wt=0.1, triangle_wt=0.002,
    pullfar_wt=0.01, deform_wt=0.4, symmetry_wt=0.5, camera_wt=0.005, match_wt=0.02, imatch_wt=0.02,
    decay_ratio=0.1, tau_mesh=10., tau_img=10., use_depth=True, shape_prior=True, prior_deform=True,
    divide_fn="both", pretrain_k=200)
PRESETS = {
    "laptop_wild6d": dict(_WILD6D_COMMON, category="laptop", symmetry_idx=1, cycle_loss_wt=0.01,
                          cycle_loss_pretrain_wt=0.02, vert_lr_ratio=0.01,
                          rotation_offset=[0.2, 0.0, 0.0, 0.0, -0.2, 0.2], base_rot=[0, 0, 1, 0, -1, 0, -1, 0, 0]),
    "bottle_wild6d": dict(_WILD6D_COMMON, category="bottle", symmetry_idx=0, cycle_loss_wt=0.02,
                          cycle_loss_pretrain_wt=0.05, vert_lr_ratio=0.1,
                          rotation_offset=[0.1, 0.0, 0.0, 0.0, 0.1, -0.1], base_rot=[1, 0, 0, 0, 1, 0, 0, 0, 1]),
    # config/{bowl,camera,mug}_wild6d/base_config.txt
    "bowl_wild6d": dict(_WILD6D_COMMON, category="bowl", symmetry_idx=0, cycle_loss_wt=0.02,
                        cycle_loss_pretrain_wt=0.005, vert_lr_ratio=0.1, cam_lr_ratio=0.2,
                        rotation_offset=[0.2, 0.0, 0.0, 0.0, -0.2, 0.2], base_rot=[1, 0, 0, 0, -1, 0, 0, 0, 1]),
    "camera_wild6d": dict(_WILD6D_COMMON, category="camera", symmetry_idx=-1, cycle_loss_wt=0.02,
                          cycle_loss_pretrain_wt=0.005, vert_lr_ratio=0.1,
                          rotation_offset=[0.2, 0.0, -0.1, 0.0, -0.2, 0.2], base_rot=[1, 0, 0, 0, -1, 0, 0, 0, 1]),
    "mug_wild6d": dict(_WILD6D_COMMON, category="mug", symmetry_idx=1, cycle_loss_wt=0.01,
                       cycle_loss_pretrain_wt=0.02, vert_lr_ratio=0.01,
                       rotation_offset=[0.1, 0.0, 0.0, 0.0, -0.1, 0.1], base_rot=[1, 0, 0, 0, -1, 0, 0, 0, 1]),
}


class Options(object):
    """attribute bag with the reference's flag names; unknown names raise AttributeError"""

    def __init__(self, preset=None, **overrides):
        self.__dict__.update(copy.deepcopy(DEFAULTS))
        if preset is not None:
            self.__dict__.update(copy.deepcopy(PRESETS[preset]))
        for k, v in overrides.items():
            if k not in DEFAULTS:
                raise AttributeError("unknown option %r" % k)
            self.__dict__[k] = v

    def flags_into_string(self):
        return "\n".join("--%s=%s" % (k, v) for k, v in sorted(self.__dict__.items()))


def _convert(name, text):
    default = DEFAULTS[name]
    if isinstance(default, bool):
        return text.strip().lower() in ("1", "true", "yes", "")
    if isinstance(default, int):
        return int(text)
    if isinstance(default, float):
        return float(text)
    if isinstance(default, list):
        return [float(x) for x in text.split(",")]
    return text


def load_flagfile(path, **overrides):
    """parse a reference flag file (`--name=value` per line) into Options"""
    opts = Options()
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if not line.startswith("--"):
                continue
            name, _, value = line[2:].partition("=")
            if name not in DEFAULTS:
                raise AttributeError("unknown flag %r in %s" % (name, path))
            setattr(opts, name, _convert(name, value))
    for k, v in overrides.items():
        setattr(opts, k, v)
    return opts
