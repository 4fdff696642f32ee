"""Host mirror of Cube (Cube.py:6-16,209-213): axis-aligned box given by centre and edge length."""
import numpy as np

from surfaces.Object3D import Object3D


class Cube(Object3D):
    KIND = 2

    def __init__(self, position, scale, material_index, index):
        super().__init__(position, material_index, index)
        self.scale = scale
        half = scale / 2
        axes = np.eye(3, dtype=int)
        # face centres, named as in the reference (Cube.py:12-14)
        self.right, self.up, self.forward = (self.position + half * axes)
        self.left, self.down, self.backward = (self.position - half * axes)

    def __repr__(self):
        return "Cube(c=%s, scale=%s, material_index=%s)" % (self.position.tolist(), self.scale, self.material_index)

    def get_enclosing_values(self):
        return (self.position - self.scale / 2), (self.position + self.scale / 2)

    def _prim_params(self):
        lo, hi = self.get_enclosing_values()
        c = self.position
        return np.array([lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], c[0], c[1], c[2], self.scale, 0.0, 0.0])

    @staticmethod
    def cube_intersection_mask(rays_sources, rays_directions, bounds_min, bounds_max, outside_hits_only=True):
        """Slab test of Cube.py:95-138 -> (mask (N,1) bool, t_near (N,1)); served by the HIP library."""
        from ray_tracing_amd import native
        return native.slab_mask(rays_sources, rays_directions, bounds_min, bounds_max, outside_hits_only)
