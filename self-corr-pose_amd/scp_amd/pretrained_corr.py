"""scp_amd/pretrained_corr.py -- DINO-supervised cycle loss.

API and semantics of model/module/pretrained_corr.py: PretrainedCorrespondence.match :48-104
(mutual nearest neighbours of DINO keys between image pairs, top-k most cycle-consistent target
pixels), compute_cycle_loss :107-140 (soft pixel->pixel map bridged through the vertices, L2 to
the DINO matches).  Batch re-pairing: model/util/loss_utils.py:326-345.

MI355X-first differences, value-preserving: DINO runs once per unique image and the src/tgt lists
are INDEX lists into its token-major keys (the reference pushes 4B images through the ViT
for B unique ones); the [2B,1024,1024] score matrix of :85 is never formed -- score GEMM and both
argmax reductions are one kernel (scp_mutual_nn_fused); `pointcorr` is pooled once per image before
pairing; the bridge `corr` matrix is only formed for the k gathered target columns.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dino, ops, streams
from .correspondence import make_meshgrid
from .dino import DINO
from .losses import pair_indices


class PretrainedCorrespondence(nn.Module):
    def __init__(self, opts, mesh, pretrained=True):
        super().__init__()
        self.opts = opts
        self.mesh = mesh
        self.net = DINO().eval()
        self.img_size, self.feat_size = opts.img_size, opts.img_size // 8
        self.tau_img, self.tau_mesh, self.k = opts.tau_img, opts.tau_mesh, opts.pretrain_k
        self.hf, self.wf = opts.corr_h, opts.corr_w
        self.register_buffer("meshgrid", make_meshgrid(self.hf, self.wf), persistent=False)
        for p in self.net.parameters():
            p.requires_grad = False
        if opts.divide_fn not in ("frame", "instance", "both"):
            raise ValueError
        self.divide_kind = opts.divide_fn
        # test hooks (discrete selections are backend-defined at ties / near-ties)
        self.topk_override = None   # indices [N,k]
        self.nn_override = None     # (bw [N,P], fw [N,P]) mutual-NN argmax indices
        self.last_nn = None
        self.last_topk = None       # the top-k selection of the last match_features call (parity runs inject it on the other side)

    def half_grid(self, bsz):
        """the pixel grid at half the correspondence resolution (pretrained_corr.py:112-114 interpolates the constant
        meshgrid for every pair of every step): computed once per device and expanded"""
        key = (self.meshgrid.device, self.meshgrid.data_ptr())
        if getattr(self, "_half_grid_key", None) != key:
            grid = self.meshgrid.reshape(2, self.hf, self.wf)[None]
            self._half_grid = F.interpolate(grid, (self.hf // 2, self.wf // 2), mode="bilinear")
            self._half_grid_key = key
            if self._half_grid.is_cuda:
                # cached and read by every stream of the step, but enqueued on whichever stream asked first (SCP_STREAMS=overlap: a side
                # stream): drained once, so that no other stream can read it before it exists (seen as a first-step deviation of the
                # bridge's column soft-argmax, tools/first_step_flake.py)
                torch.cuda.current_stream(self._half_grid.device).synchronize()
        return self._half_grid.expand(bsz, -1, -1, -1)

    # -- reference signature: images in, DINO inside --------------------------------------------
    def match(self, src_img, tgt_img, src_mask, tgt_mask, grid):
        bsz = src_img.shape[0]
        keys = self.net.key_tokens(torch.cat((src_img, tgt_img), 0))
        idx = torch.arange(bsz, device=keys.device)
        return self._match_keys(keys, torch.cat((src_mask, tgt_mask), 0), idx, idx + bsz, grid)

    def match_features(self, src_feat, tgt_feat, src_mask, tgt_mask, grid):
        """the matching given per-pair feature maps [N, C, fs, fs] (pretrained_corr.py:76-104); scores through ops.mutual_nn"""
        bsz = src_feat.shape[0]
        fs = self.feat_size
        src_feat, tgt_feat = src_feat.reshape(bsz, src_feat.shape[1], -1), tgt_feat.reshape(bsz, tgt_feat.shape[1], -1)
        src_mask_down = F.interpolate(src_mask[:, None], (fs, fs), mode="nearest").reshape(bsz, -1)
        tgt_mask_down = F.interpolate(tgt_mask[:, None], (fs, fs), mode="nearest").reshape(bsz, -1)
        bw, fw = ops.mutual_nn(src_feat, tgt_feat, src_mask_down, tgt_mask_down)
        return self._select(bw, fw, tgt_mask_down, grid)

    def _match_keys(self, keys, mask, src_idx, tgt_idx, grid):
        """the same from the token-major keys of the unique images [n_images, 1 + fs*fs, C] and the pair lists: score GEMM and both
        argmax reductions in one kernel (ops.mutual_nn_pairs), no per-pair feature gathers, no score tensor"""
        fs = self.feat_size
        mask_down = F.interpolate(mask[:, None].float(), (fs, fs), mode="nearest").reshape(mask.shape[0], -1)
        bw, fw = ops.mutual_nn_pairs(keys, src_idx, tgt_idx, mask_down, tok0=keys.shape[1] - fs * fs)
        return self._select(bw, fw, mask_down[tgt_idx], grid)

    def _select(self, bw, fw, tgt_mask_down, grid):
        """cycle distance of every target pixel through its mutual nearest neighbours, the k most consistent ones
        (pretrained_corr.py:88-104)"""
        bsz = bw.shape[0]
        self.last_nn = (bw, fw)
        if self.nn_override is not None:
            bw, fw = self.nn_override
        cy = torch.gather(fw, -1, bw)
        grid = grid.reshape(bsz, 2, -1)
        pick = lambda idx: torch.gather(grid, -1, idx[:, None].expand(-1, 2, -1))
        match, cycle = pick(bw), pick(cy)
        distance = (cycle - grid).norm(2, 1)
        distance = torch.where(tgt_mask_down > 0, distance, torch.full_like(distance, 1e5))
        if self.topk_override is not None:
            indices = self.topk_override
        else:
            indices = torch.topk(-distance, k=self.k, dim=1).indices
        self.last_distance = distance.detach()
        self.last_topk = indices
        match = torch.gather(match, -1, indices[:, None].expand(-1, 2, -1))
        grid_k = torch.gather(grid, -1, indices[:, None].expand(-1, 2, -1))
        match_mask = torch.gather(tgt_mask_down, -1, indices)
        indices_match = torch.gather(bw, -1, indices)
        return match, grid_k, indices_match, indices, match_mask

    def _keep_tokens(self, mask):
        """the patch tokens inside the object mask at the DINO resolution: the matching masks every other token out of the
        mutual-nearest-neighbour search (pretrained_corr.py:85-89), so their features are never read"""
        fs = self.feat_size
        return F.interpolate(mask[:, None].float(), (fs, fs), mode="nearest").reshape(mask.shape[0], -1) > 0

    def prefetch_features(self, img, mask=None):
        """Start the frozen DINO ViT on a side HIP stream (it depends on nothing but the input images);
        the encoder / correspondence / render work of the main stream overlaps with it and
        compute_cycle_loss joins the stream right before it needs the features."""
        if not img.is_cuda:
            return
        pre = getattr(self, "_prefetched", None)
        if pre is not None and pre[0] is img:
            return                      # already in flight (e.g. started during the previous step's backward)
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = streams.side_stream(img.device)
        self._side_stream.wait_stream(torch.cuda.current_stream(img.device))
        with torch.cuda.stream(self._side_stream):
            streams.crumb("ViT prefetch: start")
            if self.use_graphs and mask is not None and self.nn_override is None and self.topk_override is None and not dino.MIXED_BF16:
                # the whole frozen pass (ViT + matching, ~150 launches) as one HIP-graph replay (scp_amd/graphed.py)
                if getattr(self, "_vit_graph", None) is None:
                    from .graphed import GraphedInference
                    self._vit_graph = GraphedInference(self._keys_and_matches, warmup=2)
                out = self._vit_graph(img, mask)
                keys, matched = out[0], tuple(out[1:])
            else:
                keys = self.net.key_tokens(img, None if mask is None else self._keep_tokens(mask))
                # the mutual-nearest-neighbour matching of the re-paired batch (fused score GEMM + dual argmax, top-k) needs nothing
                # but the keys and the masks either: it stays on the side stream instead of the main stream's critical path
                matched = self._match_pairs(keys, mask) if mask is not None else None
            streams.crumb("ViT prefetch: done")
        img.record_stream(self._side_stream)
        if mask is not None:
            mask.record_stream(self._side_stream)
        self._prefetched = (img, keys, matched)

    use_graphs = False          # Trainer switches it on for fp32 training on the GPU

    @torch.no_grad()
    def _keys_and_matches(self, img, mask):
        keys = self.net.key_tokens(img, self._keep_tokens(mask))
        return (keys,) + tuple(self._match_pairs(keys, mask))

    def _match_pairs(self, keys, mask):
        src_idx, tgt_idx = pair_indices(self.divide_kind, self.opts.batch_size, self.opts.repeat, keys.device)
        return self._match_keys(keys, mask, src_idx, tgt_idx, self.half_grid(src_idx.shape[0]))

    def _features(self, img, mask=None):
        """(DINO keys once per unique image, token-major; their pair matching) -- from the side stream if prefetch_features
        started them"""
        pre = getattr(self, "_prefetched", None)
        self._prefetched = None
        if pre is not None and pre[0] is img:
            main = torch.cuda.current_stream(img.device)
            main.wait_stream(self._side_stream)
            keys, matched = pre[1], pre[2]
            if matched is None:
                matched = self._match_pairs(keys, mask)
            else:
                for t in matched:
                    t.record_stream(main)
            return keys, matched
        keys = self.net.key_tokens(img, None if mask is None else self._keep_tokens(mask))
        return keys, self._match_pairs(keys, mask)

    def compute_cycle_loss(self, img, mask, depth_weight, pointcorr, with_images=False):
        num_verts = pointcorr.shape[-1]
        src_idx, tgt_idx = pair_indices(self.divide_kind, self.opts.batch_size, self.opts.repeat, img.device)
        n = src_idx.shape[0]
        hh, wh = self.hf // 2, self.wf // 2
        grid = self.half_grid(n)

        feats, (pts_src, pts_tgt, indices_src, indices_tgt, mask_k) = self._features(img, mask)      # once per unique image

        # bilinear half-resolution (= exact 2x2 mean) of the score maps, once per image
        pooled = ops.pool2x2_scores(pointcorr, self.hf, self.wf)                                    # b,p,v
        match = ops.vertex_bridge_match(pooled, src_idx, tgt_idx, indices_tgt, depth_weight >= 0.5,
                                        self.half_grid(1).reshape(2, -1), self.tau_img, self.tau_mesh,
                                        precomputed=getattr(pointcorr, "bridge", None))
        cycle_loss = ((match - pts_src).norm(2, 1) * mask_k).mean()
        # the reference also returns the paired images (pretrained_corr.py:139; only its visualisation reads them): two 50 MB
        # gathers per step at B = 32 that the training step never looks at -- on request only
        if with_images:
            return cycle_loss, pts_src, pts_tgt, match, mask_k, img[src_idx], img[tgt_idx]
        return cycle_loss, pts_src, pts_tgt, match, mask_k, None, None
