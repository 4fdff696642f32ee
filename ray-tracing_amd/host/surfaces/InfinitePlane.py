"""Host mirror of InfinitePlane (InfinitePlane.py:5-10): stores the unit normal
and d = -offset/|n| so that the plane is  P.n + d = 0."""
import numpy as np

from surfaces.SurfaceAbs import SurfaceAbs


class InfinitePlane(SurfaceAbs):
    KIND = 3

    def __init__(self, normal, offset, material_index, index):
        super().__init__(material_index, index)
        length = np.linalg.norm(normal)
        self.normal = np.array(normal) / length
        self.offset = -offset / length

    def __repr__(self):
        return "InfinitePlane(n=%s, d=%s, material_index=%s)" % (self.normal.tolist(), self.offset,
                                                                 self.material_index)

    def _prim_params(self):
        n = self.normal
        return np.array([n[0], n[1], n[2], self.offset, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
