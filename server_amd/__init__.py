"""server_amd -- MI355X-native image mixer for CasparCG Server (the `accelerator/hip` hot path).

The product is two native libraries built in-tree by server_amd/csrc/Makefile:
  libcghip.so   C ABI (include/cghip.h): device/pools/planes + the fused gfx950 compositor kernel
  libcghost.so  C++ host mixer mirroring core::image_mixer (push/visit/pop/render/create_frame)
This package only binds them with ctypes so tests and bench.py can drive them; there is no Python or CPU
fallback: if the libraries are missing, importing the bindings fails loudly.
"""
from .capi import (BLEND, PF, CGHIPError, Coord, Draw, Format, Mixer, Surface, Transform, default_transform,
                   device_count, lib, host, build, library_paths)

__all__ = ["BLEND", "PF", "CGHIPError", "Coord", "Draw", "Format", "Mixer", "Surface", "Transform",
           "default_transform", "device_count", "lib", "host", "build", "library_paths"]
