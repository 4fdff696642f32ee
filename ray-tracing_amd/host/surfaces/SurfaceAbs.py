"""Host mirror of SurfaceAbs (SurfaceAbs.py:5-42).  The batched per-surface
operations keep the reference's names; they are served by the HIP library."""
from abc import ABC

import numpy as np


class SurfaceAbs(ABC):
    #: RT_SURF_* kind used by scene_pack (include/rt_amd.h)
    KIND = 0

    def __init__(self, material_index, index):
        self.index = index
        self.material_index = material_index

    def __repr__(self):
        return "%s(material_index=%s)" % (type(self).__name__, self.material_index)

    def get_material_index(self):
        return self.material_index if self.material_index else 0

    # -- batched geometry, same signatures as the reference's abstract methods ------------
    def _prim_params(self):
        raise NotImplementedError

    def intersect_vectorized(self, rays_sources: np.ndarray, rays_directions: np.ndarray) -> np.ndarray:
        """(N,3),(N,3) -> (N,3) hit points, NaN rows where the ray misses."""
        from ray_tracing_amd import native
        return native.prim_intersect(self.KIND, self._prim_params(), rays_sources, rays_directions)

    def calculate_normals(self, rays_interactions: np.ndarray) -> np.ndarray:
        from ray_tracing_amd import native
        return native.prim_normals(self.KIND, self._prim_params(), rays_interactions)
