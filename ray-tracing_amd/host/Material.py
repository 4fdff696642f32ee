"""Host mirror of the Material record (Material.py:4-11).  The per-hit material
gather of Material.get_materials_base_colors (Material.py:23-62) happens on the
GPU from the table scene_pack builds (integer shininess included)."""
import numpy as np


class Material:
    def __init__(self, diffuse_color, specular_color, reflection_color, shininess, transparency, mat_index):
        self.diffuse_color = np.array(diffuse_color)
        self.specular_color = np.array(specular_color)
        self.reflection_color = np.array(reflection_color)
        self.shininess = np.array(shininess)
        self.transparency = np.array(transparency)
        self.index = mat_index

    def __repr__(self):
        return "Material(%d: kd=%s ks=%s kr=%s n=%s t=%s)" % (
            self.index, self.diffuse_color, self.specular_color, self.reflection_color, self.shininess,
            self.transparency)
