"""Host mirror of Object3D (Object3D.py:6-32): finite primitives that go into the BSP tree."""
from abc import ABC

import numpy as np

from surfaces.SurfaceAbs import SurfaceAbs


class Object3D(SurfaceAbs, ABC):
    def __init__(self, position, material_index, index):
        super().__init__(material_index, index)
        self.position = np.array(position)

    def get_enclosing_values(self):
        """(smallest corner, biggest corner) of the axis-aligned bounding box."""
        raise NotImplementedError
