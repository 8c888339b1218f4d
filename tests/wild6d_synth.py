"""tests/wild6d_synth.py -- writes a tiny dataset in the reference's Wild6D on-disk layout
(data/dataset_wild6d.py:50-75:  <root>/<obj>/<seq>/images/{N.jpg, N-mask.png, N-depth.png} + <seq>/metadata JSON
with K (column-major 3x3), w, h) from a seed, with PIL.  Shared by the fixture generator and the tests; PNG is
lossless and JPEG bytes are reproducible for one PIL build, so both sides decode identical pixels."""
import json
import os

import numpy as np
from PIL import Image


def write_dataset(root, n_obj=2, n_seq=2, n_frames=6, w=160, h=120, seed=0):
    rng = np.random.RandomState(seed)
    names = []
    ys, xs = np.mgrid[0:h, 0:w]
    for o in range(n_obj):
        for s in range(n_seq):
            seq_dir = os.path.join(root, "obj%02d" % o, "seq%02d" % s)
            os.makedirs(os.path.join(seq_dir, "images"), exist_ok=True)
            fx, fy = 180 + 20 * rng.rand(), 175 + 20 * rng.rand()
            K = np.array([[fx, 0, w / 2 + 3 * rng.randn()], [0, fy, h / 2 + 3 * rng.randn()], [0, 0, 1]])
            with open(os.path.join(seq_dir, "metadata"), "w") as f:
                json.dump({"K": K.T.reshape(-1).tolist(), "w": w, "h": h, "fps": 30}, f)
            for n in range(n_frames):
                cx, cy = w * (0.35 + 0.3 * rng.rand()), h * (0.35 + 0.3 * rng.rand())
                # some objects touch the border so that the 1.2-1.5x crop box leaves the frame (zero padding path)
                if n % 3 == 2:
                    cx = w * 0.12
                ax, ay = w * (0.10 + 0.08 * rng.rand()), h * (0.15 + 0.1 * rng.rand())
                inside = ((xs - cx) / ax) ** 2 + ((ys - cy) / ay) ** 2 < 1
                img = (rng.rand(h, w, 3) * 60 + 90 + 50 * np.sin(xs / 9.0 + n)[..., None] + 40 * inside[..., None]).clip(0, 255)
                depth = np.where(inside, 600 + 30 * np.cos(xs / 15.0) + 5 * rng.randn(h, w), 1200 + 10 * rng.randn(h, w))
                depth[rng.rand(h, w) < 0.03] = 0
                base = os.path.join(seq_dir, "images", "%d" % n)
                Image.fromarray(img.astype(np.uint8)).save(base + ".jpg", quality=92)
                Image.fromarray((inside * 255).astype(np.uint8)).save(base + "-mask.png")
                Image.fromarray(depth.astype(np.uint16)).save(base + "-depth.png")
            names.append("synth_%d_%d" % (o, s))
    list_path = os.path.join(root, "..", os.path.basename(root) + "_train_list.txt")
    with open(list_path, "w") as f:
        f.write("\n".join(names) + "\n")
    return os.path.abspath(list_path)


def write_test_set(base, category="laptop", n_obj=1, n_seq=2, n_frames=4, w=160, h=120, seed=3):
    """the reference's test layout: <base>/test_set/<category>/<obj>/<seq>/... and
    <base>/test_set/pkl_annotations/<category>/<category>-<obj>-<seq>.pkl with per-frame rotation / translation / size"""
    import pickle
    root = os.path.join(base, "test_set", category)
    list_path = write_dataset(root, n_obj=n_obj, n_seq=n_seq, n_frames=n_frames, w=w, h=h, seed=seed)
    rng = np.random.RandomState(seed + 1)
    ann_dir = os.path.join(base, "test_set", "pkl_annotations", category)
    os.makedirs(ann_dir, exist_ok=True)
    for obj in sorted(os.listdir(root)):
        for seq in sorted(os.listdir(os.path.join(root, obj))):
            annos = []
            for i in range(n_frames):
                q, _ = np.linalg.qr(rng.randn(3, 3))
                if np.linalg.det(q) < 0:
                    q[:, 0] = -q[:, 0]
                annos.append({"name": "%s/%s/%s/%d" % (category, obj, seq, i), "rotation": q.tolist(),
                              "translation": (np.array([0.0, 0.0, 0.6]) + 0.02 * rng.randn(3)).tolist(),
                              "size": (0.2 + 0.1 * rng.rand(3)).tolist()})
            with open(os.path.join(ann_dir, "%s-%s-%s.pkl" % (category, obj, seq)), "wb") as f:
                pickle.dump({"annotations": annos}, f)
    return root + "/", list_path
