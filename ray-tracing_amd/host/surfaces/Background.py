"""Host mirror of Background (Background.py:5-44): list position 0, material 0,
'hit' at (1e8,1e8,1e8) which acts as the far clip in the nearest-hit kernel."""
import numpy as np

from surfaces.SurfaceAbs import SurfaceAbs
from util import MAX_RENDER_DISTANCE


class Background(SurfaceAbs):
    KIND = 0

    def __init__(self):
        super().__init__(material_index=0, index=0)

    def intersect_vectorized(self, rays_sources, rays_directions):
        return np.full_like(rays_directions, fill_value=MAX_RENDER_DISTANCE)

    def calculate_normals(self, rays_interactions):
        return np.zeros_like(rays_interactions)
