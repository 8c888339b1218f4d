"""scp_amd.soft_renderer.mesh -- `sr.Mesh` as the reference uses it
(third-party/softras/soft_renderer/mesh.py:9-134): a bag of batched vertices / faces / textures
with lazily gathered per-face views.  from_obj/save_obj handle plain v/f OBJ files only."""
import numpy as np
import torch
import torch.nn.functional as F

from . import functional as srf


class Mesh(object):
    def __init__(self, vertices, faces, textures=None, texture_res=1, texture_type="surface"):
        if isinstance(vertices, np.ndarray):
            vertices = torch.from_numpy(vertices).float().cuda()
        if isinstance(faces, np.ndarray):
            faces = torch.from_numpy(faces).int().cuda()
        if vertices.dim() == 2:
            vertices = vertices[None]
        if faces.dim() == 2:
            faces = faces[None]
        self._vertices, self._faces = vertices, faces
        self.device = vertices.device
        self.texture_type = texture_type
        self.batch_size, self.num_vertices = vertices.shape[:2]
        self.num_faces = faces.shape[1]
        self._cache = {}
        self._fill_back = False
        if textures is None:
            if texture_type == "surface":
                textures = torch.ones(self.batch_size, self.num_faces, texture_res ** 2, 3,
                                      dtype=torch.float32, device=self.device)
                self.texture_res = texture_res
            elif texture_type == "vertex":
                textures = torch.ones(self.batch_size, self.num_vertices, 3, dtype=torch.float32,
                                      device=self.device)
                self.texture_res = 1
            else:
                raise ValueError("texture type not applicable")
        else:
            if isinstance(textures, np.ndarray):
                textures = torch.from_numpy(textures).float().cuda()
            if textures.dim() == 3 and texture_type == "surface":
                textures = textures[None]
            if textures.dim() == 2 and texture_type == "vertex":
                textures = textures[None]
            self.texture_res = int(np.sqrt(textures.shape[2]))
        self._textures = textures
        self._origin = (vertices, faces, textures)

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
