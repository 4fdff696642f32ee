"""Host mirror of the Light record (Light.py:7-14).  All lighting arithmetic
(Light.py:35-288) runs in the HIP shade/shadow kernels."""
import numpy as np


class Light:
    def __init__(self, position, color, specular_intensity, shadow_intensity, radius, index: int):
        self.position = np.array(position)
        self.color = np.array(color)
        self.specular_intensity = specular_intensity
        self.shadow_intensity = shadow_intensity
        self.radius = radius
        self.index = index

    def __repr__(self):
        return "Light(%d: pos=%s color=%s spec=%s shadow=%s radius=%s)" % (
            self.index, self.position, self.color, self.specular_intensity, self.shadow_intensity, self.radius)
