"""Host mirror of the reference's `util` module (util.py:1-132): the constants and
the two basis helpers the scene model needs, plus `save_image`.  Only host-side
set-up runs here; per-ray work is done by the HIP library (see ../native.py)."""
import numpy as np

ColorVector = np.ndarray
Vector = np.ndarray
Matrix = np.ndarray

X_DIRECTION = np.array([1, 0, 0])
Y_DIRECTION = np.array([0, 1, 0])
Z_DIRECTION = np.array([0, 0, 1])

EPSILON = 1e-6                       # util.py:13
MAX_RENDER_DISTANCE = 100_000_000    # util.py:15


def _unit_rows(v, pad=0.0):
    return v / (np.linalg.norm(v, axis=-1, keepdims=True) + pad)


def get_orthonormal_vector(vector1):
    """(1, 1, z) with z chosen so the result is perpendicular to each row of
    `vector1` when its z component is non-zero, (1, 1, 0) otherwise; normalised
    (util.py:113-132, including the non-orthogonal z == 0 case)."""
    out = np.zeros_like(vector1)
    out[:, :2] = 1
    nz = vector1[:, 2] != 0
    out[nz, 2] = -(1 * vector1[nz, 0] + 1 * vector1[nz, 1]) / vector1[nz, 2]
    return out / np.linalg.norm(out, axis=-1, keepdims=True)


def create_orthonormal_basis(vector1, vector2=None):
    """Gram-Schmidt step + cross product with the reference's `+ EPSILON` in both
    normalisations (util.py:48-75), so bases are bit-identical to the reference's."""
    if vector2 is None:
        second = get_orthonormal_vector(vector1)
    else:
        along = np.sum(vector1 * vector2, axis=-1, keepdims=True)
        second = _unit_rows(vector2 - along * vector1, EPSILON)
    third = _unit_rows(np.cross(second, vector1), EPSILON)
    return vector1, second, third


def save_image(image_array, path, height, width):
    """(H*W,3) floats in [0,1] -> 8-bit RGB file; quantisation is truncation of
    x*255 (util.py:94-98).  Accepts an already quantised uint8 array as well."""
    from PIL import Image
    arr = np.asarray(image_array)
    if arr.dtype != np.uint8:
        arr = (arr.reshape((height, width, 3)) * 255).astype(np.uint8)
    Image.fromarray(arr.reshape((height, width, 3))).save(path)
