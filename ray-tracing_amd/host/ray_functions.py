"""Host mirror of the reference's ray_functions module: only the entry point the
drop-in caller uses (get_initial_rays, ray_functions.py:7-43).  It is served by
the k_gen kernel; in the fused frame path (rt_render) the rays never leave the GPU."""
import numpy as np  # noqa: F401


def get_initial_rays(camera, image_width: int, image_height: int):
    """-> (sources (H*W,3), directions (H*W,3)) float64, ray index = row * W + col."""
    from ray_tracing_amd import native, scene_pack
    from surfaces.Sphere import Sphere
    # a camera-only context: one dummy primitive so that the tree has a node
    dummy = [Sphere([0.0, 0.0, 0.0], 1.0, 0, 1)]
    packed = scene_pack.pack_scene(dummy, camera=camera)
    with native.Context(packed) as ctx:
        return ctx.initial_rays(image_width, image_height)
