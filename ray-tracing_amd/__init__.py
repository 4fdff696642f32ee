"""MI355X-native per-pixel hot path of the Ray-Tracing scene format.

  abi         ctypes mirror of include/rt_amd.h
  scene_pack  host objects -> flat float64 SoA scene description
  native      ctypes binding of csrc/librt_amd.so (HIP kernels); fails loudly when missing
  render      frame-level helpers (single GPU and tile-partitioned multi GPU)
  host/       mirror of the reference's module interface (Parser, Camera, BSPNode, surfaces, ...)

Importing this package puts host/ on sys.path so that the reference's bare
module names (`from Parser import parse_scene_file`) resolve to the mirror.
"""
import os as _os
import sys as _sys

PKG_DIR = _os.path.abspath(__path__[0])  # noqa: F821 (set by the import system or by ray_tracing_amd.py)
HOST_DIR = _os.path.join(PKG_DIR, "host")
if HOST_DIR not in _sys.path:
    _sys.path.insert(0, HOST_DIR)
