"""Frame-level helpers over native.Context: load a scene file, render whole
frames on one GPU, or this rank's tiles for a tile-partitioned multi-GPU frame.

Multi-GPU model: pixels are independent, every rank holds the
whole scene, rank r renders the 8x8 tiles t with t % world == r, and the only
exchange step of a frame is the gather of the packed uint8 tiles to rank 0
(torch.distributed: RCCL on GPUs, gloo in the CPU tests), followed by a
de-interleave into the row-major image.  The jitter hash is keyed by the global
pixel index, so the image does not depend on the number of ranks.
"""
import numpy as np

from . import abi, native, scene_pack


class Scene:
    """Parsed scene + tree + packed buffers (host side only)."""

    def __init__(self, camera, settings, surfaces, materials, lights, bsp_tree=None, build="host", device=0):
        """build = "host": the recursive NumPy restatement of BSPNode.build_bsp_tree (runs anywhere);
        build = "device": the same tree built in HBM by rt_bsp_build (raises without a GPU)."""
        if bsp_tree is None:
            from BSPNode import BSPNode
            if build == "device":
                bsp_tree = BSPNode.build_bsp_tree_device(surfaces, device=device)
            elif build == "host":
                bsp_tree = BSPNode.build_bsp_tree(surfaces)
            else:
                raise ValueError("build must be 'host' or 'device'")
        self.camera, self.settings, self.surfaces = camera, settings, surfaces
        self.materials, self.lights, self.bsp_tree = materials, lights, bsp_tree
        self.packed = scene_pack.pack_scene(surfaces, materials, lights, camera, settings, bsp_tree)

    @classmethod
    def from_file(cls, path, build="host", device=0):
        from Parser import parse_scene_file
        camera, settings, _objs, surfaces, materials, lights = parse_scene_file(path)
        return cls(camera, settings, surfaces, materials, lights, build=build, device=device)


class FastScene:
    """Scene file -> tables -> tree with no Python object per surface: rt_scene_parse(_file) tokenizes the text
    into the rt_scene_desc tables (bit-identical to Parser + scene_pack) and rt_scene_build_tree builds the
    reference's BSP tree on the GPU.  Takes a million-line scene in a fraction of a second where the object
    path needs half a minute.  Has the attributes Renderer and bench.py use (`packed`, `settings`)."""

    class _Settings:
        def __init__(self, desc):
            self.background_color = list(desc.background)
            self.root_number_shadow_rays = int(desc.sh_rays)
            self.max_recursions = int(desc.rec_max)

    def __init__(self, path=None, text=None, device=0, threads=0):
        import time
        t0 = time.perf_counter()
        self.packed = native.NativeScene(text=text, path=path, threads=threads)
        t1 = time.perf_counter()
        self.build_stats = self.packed.build_tree(device)
        t2 = time.perf_counter()
        self.settings = FastScene._Settings(self.packed.desc)
        self.timing = dict(parse_ms=(t1 - t0) * 1e3, tree_ms=(t2 - t1) * 1e3, tree_device_ms=self.build_stats["ms_device"])

    @classmethod
    def from_file(cls, path, device=0, threads=0):
        return cls(path=path, device=device, threads=threads)


class Renderer:
    def __init__(self, scene: Scene, device=0):
        self.scene = scene
        self.ctx = native.Context(scene.packed, device=device)

    def close(self):
        self.ctx.close()

    def frame(self, width, height, **kw):
        """Whole frame on this GPU -> dict(rgb8 (H,W,3) uint8, rgb (H,W,3) float32, ids (H,W), stats)."""
        p = abi.render_params(width, height, **kw)
        out = self.ctx.render(p)
        return dict(rgb8=out["rgb8"].reshape(height, width, 3), rgb=out["rgb"].reshape(height, width, 3),
                    ids=out["ids"].reshape(height, width), stats=out["stats"])


# ---------------------------------------------------------------------------- multi-rank helpers
def deinterleave_host(packed_per_rank, width, height, world, tile_w=0, tile_h=0, channels=3):
    """NumPy version of rt_deinterleave_u8: list/array of `world` packed rank buffers -> (H,W,channels)."""
    first = np.asarray(packed_per_rank[0])
    img = np.zeros((height * width, channels), dtype=first.dtype)
    for r in range(world):
        p = abi.render_params(width, height, rank=r, world=world, tile_w=tile_w, tile_h=tile_h)
        pix = native.tile_layout(p)
        buf = np.asarray(packed_per_rank[r]).reshape(-1, channels)
        ok = pix >= 0
        img[pix[ok]] = buf[ok]
    return img.reshape(height, width, channels)


def gather_packed(local, rank, world, dst=0, group=None):
    """The frame's one exchange step: every rank contributes its equally sized packed tile buffer
    (a torch tensor); rank `dst` receives them concatenated (world * n), others get None."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local
    if local.is_cuda and dist.get_backend(group) != "nccl":
        # a backend without device collectives (gloo rehearsals): the same exchange, staged through host memory
        host = local.cpu()
        got = torch.empty(world * host.numel(), dtype=host.dtype) if rank == dst else None
        dist.gather(host, list(got.chunk(world)) if rank == dst else None, dst=dst, group=group)
        return got.to(local.device) if rank == dst else None
    out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device) if rank == dst else None
    dist.gather(local, list(out.chunk(world)) if rank == dst else None, dst=dst, group=group)
    return out
