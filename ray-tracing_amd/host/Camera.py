"""Host mirror of the reference's Camera record (Camera.py:4-14)."""
import numpy as np
from util import create_orthonormal_basis


class Camera:
    def __init__(self, position, look_at, up_vector, screen_distance, screen_width):
        self.position = np.array(position)
        self.look_at = np.array(look_at)
        self.up_vector = np.array(up_vector)
        self.screen_distance = screen_distance
        self.screen_width = screen_width
        toward = self.look_at - self.position
        self.look_direction = toward / np.linalg.norm(toward)
        # z = view direction, y = up made perpendicular to z, x = y cross z (Camera.py:13)
        self.z_dir, self.y_dir, self.x_dir = create_orthonormal_basis(self.look_direction, self.up_vector)

    def __repr__(self):
        return "Camera(pos=%s, look_at=%s, up=%s, dist=%s, width=%s)" % (
            self.position, self.look_at, self.up_vector, self.screen_distance, self.screen_width)
