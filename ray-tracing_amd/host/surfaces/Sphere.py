"""Host mirror of Sphere (Sphere.py:5-8,135-139)."""
import numpy as np

from surfaces.Object3D import Object3D


class Sphere(Object3D):
    KIND = 1

    def __init__(self, position, radius, material_index, index):
        super().__init__(position, material_index, index)
        self.radius: float = radius

    def __repr__(self):
        return "Sphere(c=%s, r=%s, material_index=%s)" % (self.position.tolist(), self.radius, self.material_index)

    def get_enclosing_values(self):
        return (self.position - self.radius), (self.position + self.radius)

    def _prim_params(self):
        c = np.asarray(self.position, dtype=np.float64)
        return np.array([c[0], c[1], c[2], self.radius ** 2, self.radius, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
