"""scp_amd.soft_renderer.mesh -- `sr.Mesh` as the reference uses it
(third-party/softras/soft_renderer/mesh.py:9-134): a bag of batched vertices / faces / textures
with lazily gathered per-face views.  from_obj/save_obj handle plain v/f OBJ files only."""
import numpy as np
import torch
import torch.nn.functional as F

from . import functional as srf


def _batched(x, dtype, want_dim):
    """tensor (or array) -> device tensor with a leading batch axis: the three shapes the product passes are already batched
    tensors; a single mesh (OBJ loader, tests) gains the axis here"""
    if not torch.is_tensor(x):
        x = torch.as_tensor(np.asarray(x), dtype=dtype)
        x = x.cuda() if torch.cuda.is_available() else x
    return x[None] if x.dim() == want_dim - 1 else x


class Mesh(object):
    """vertices [B,V,3], faces [B,F,3] (int), textures per `texture_type`:
         "vertex"  [B,V,3]        one colour per vertex (every textured pass of the training step)
         "surface" [B,F,R*R,3]    R x R texels per face (the mask pass: R = 1, all ones -- SURVEY F14)
       textures=None means all-ones of the right shape."""

    TEXTURE_RANK = {"surface": 4, "vertex": 3}

    def __init__(self, vertices, faces, textures=None, texture_res=1, texture_type="surface"):
        if texture_type not in self.TEXTURE_RANK:
            raise ValueError("texture type not applicable")
        self._vertices = _batched(vertices, torch.float32, 3)
        self._faces = _batched(faces, torch.int32, 3)
        self.device, self.texture_type = self._vertices.device, texture_type
        self.batch_size, self.num_vertices = self._vertices.shape[:2]
        self.num_faces = self._faces.shape[1]
        if textures is None:
            per_item = (self.num_faces, texture_res ** 2, 3) if texture_type == "surface" else (self.num_vertices, 3)
            self._textures = torch.ones((self.batch_size,) + per_item, dtype=torch.float32, device=self.device)
            self.texture_res = texture_res if texture_type == "surface" else 1
        else:
            self._textures = _batched(textures, torch.float32, self.TEXTURE_RANK[texture_type])
            self.texture_res = int(np.sqrt(self._textures.shape[2]))
        self._cache = {}
        self._fill_back = False
        self._origin = (self._vertices, self._faces, self._textures)

    # -- mutable geometry; any change drops the gathered views
    @property
    def vertices(self):
        return self._vertices

    @vertices.setter
    def vertices(self, v):
        self._vertices = v
        self.num_vertices = v.shape[1]
        self._cache.clear()

    @property
    def faces(self):
        return self._faces

    @faces.setter
    def faces(self, f):
        self._faces = f
        self.num_faces = f.shape[1]
        self._cache.clear()

    @property
    def textures(self):
        return self._textures

    @textures.setter
    def textures(self, t):
        self._textures = t

    def _cached(self, key, fn):
        if key not in self._cache:
            self._cache[key] = fn()
        return self._cache[key]

    @property
    def face_vertices(self):
        return self._cached("fv", lambda: srf.face_vertices(self.vertices, self.faces))

    @property
    def surface_normals(self):
        def compute():
            fv = self.face_vertices
            v10, v12 = fv[:, :, 0] - fv[:, :, 1], fv[:, :, 2] - fv[:, :, 1]
            return F.normalize(torch.cross(v12, v10, dim=2), p=2, dim=2, eps=1e-6)
        return self._cached("sn", compute)

    @property
    def vertex_normals(self):
        return self._cached("vn", lambda: srf.vertex_normals(self.vertices, self.faces))

    @property
    def face_textures(self):
        if self.texture_type == "surface":
            return self.textures
        if self.texture_type == "vertex":
            return srf.face_vertices(self.textures, self.faces)
        raise ValueError("texture type not applicable")

    def fill_back_(self):
        if not self._fill_back:
            self.faces = torch.cat((self.faces, self.faces[:, :, [2, 1, 0]]), dim=1)
            self.textures = torch.cat((self.textures, self.textures), dim=1)
            self._fill_back = True

    def reset_(self):
        self.vertices, self.faces, self.textures = self._origin
        self._fill_back = False

    @classmethod
    def from_obj(cls, filename_obj, normalization=False, load_texture=False, texture_res=1,
                 texture_type="surface"):
        if load_texture:
            raise NotImplementedError("texture atlases are outside the hot path (SURVEY.md section 2)")
        vs, fs = [], []
        with open(filename_obj) as fh:
            for line in fh:
                tok = line.split()
                if tok and tok[0] == "v":
                    vs.append([float(x) for x in tok[1:4]])
                elif tok and tok[0] == "f":
                    idx = [int(x.split("/")[0]) - 1 for x in tok[1:]]
                    for i in range(1, len(idx) - 1):
                        fs.append([idx[0], idx[i], idx[i + 1]])
        v = torch.tensor(vs, dtype=torch.float32)
        if normalization:
            v = v - v.min(0)[0][None]
            v = v / v.abs().max() * 2
            v = v - v.max(0)[0][None] / 2
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        return cls(v.to(dev), torch.tensor(fs, dtype=torch.int32, device=dev), None, texture_res, texture_type)

    def save_obj(self, filename_obj, save_texture=False, texture_res_out=16):
        if self.batch_size != 1:
            raise ValueError("Could not save when batch size >= 1")
        if save_texture:
            raise NotImplementedError("texture atlases are outside the hot path (SURVEY.md section 2)")
        v = self.vertices[0].detach().cpu().numpy()
        f = self.faces[0].detach().cpu().numpy()
        with open(filename_obj, "w") as fh:
            fh.write("# %d vertices, %d faces\n" % (len(v), len(f)))
            for p in v:
                fh.write("v %.8f %.8f %.8f\n" % tuple(p))
            for t in f:
                fh.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))
