"""ctypes binding of include/cghip.h (libcghip.so) and of the C++ host mixer's flat shim (libcghost.so).

`Mixer` mirrors the reference's core::image_mixer surface (core/mixer/image/image_mixer.h:33-57):
push(frame_transform) / visit(const_frame) / pop() / render(format_desc) -- the calls core::mixer makes
(core/mixer/mixer.cpp:62-92). Everything below the Python call is native; nothing here computes pixels.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# CGHIP_LIB_DIR: another build of the SAME two libraries, e.g. the sanitizer builds of `make sanitize`
# (tests/run_sanitized.sh); never a different implementation - there is none
LIBDIR = os.environ.get("CGHIP_LIB_DIR") or HERE
LIBCGHIP = os.path.join(LIBDIR, "libcghip.so")
LIBCGHOST = os.path.join(LIBDIR, "libcghost.so")

PF = dict(gray=0, bgra=1, rgba=2, argb=3, abgr=4, ycbcr=5, ycbcra=6, luma=7, bgr=8, rgb=9, uyvy=10, count=11,
          invalid=12)
BLEND = dict(normal=0, lighten=1, darken=2, multiply=3, average=4, add=5, subtract=6, difference=7, negation=8,
             exclusion=9, screen=10, overlay=11, soft_light=12, hard_light=13, color_dodge=14, color_burn=15,
             linear_dodge=16, linear_burn=17, linear_light=18, vivid_light=19, pin_light=20, hard_mix=21,
             reflect=22, glow=23, phoenix=24, contrast=25, saturation=26, color=27, luminosity=28, mix=29)

CGHIP_RENDER_READBACK = 0x1


class CGHIPError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("cghip status %d: %s" % (status, message))
        self.status = status


class Coord(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("vertex_x", "vertex_y", "texture_x", "texture_y", "texture_r", "texture_q")]


class Transform(C.Structure):  # cghip_transform
    _fields_ = [
        ("opacity", C.c_double), ("contrast", C.c_double), ("brightness", C.c_double), ("saturation", C.c_double),
        ("anchor", C.c_double * 2), ("fill_translation", C.c_double * 2), ("fill_scale", C.c_double * 2),
        ("clip_translation", C.c_double * 2), ("clip_scale", C.c_double * 2), ("angle", C.c_double),
        ("crop_ul", C.c_double * 2), ("crop_lr", C.c_double * 2),
        ("pers_ul", C.c_double * 2), ("pers_ur", C.c_double * 2), ("pers_lr", C.c_double * 2),
        ("pers_ll", C.c_double * 2),
        ("levels_min_input", C.c_double), ("levels_max_input", C.c_double), ("levels_gamma", C.c_double),
        ("levels_min_output", C.c_double), ("levels_max_output", C.c_double),
        ("chroma_enable", C.c_int32), ("chroma_show_mask", C.c_int32),
        ("chroma_target_hue", C.c_double), ("chroma_hue_width", C.c_double),
        ("chroma_min_saturation", C.c_double), ("chroma_min_brightness", C.c_double),
        ("chroma_softness", C.c_double), ("chroma_spill_suppress", C.c_double),
        ("chroma_spill_suppress_saturation", C.c_double),
        ("is_key", C.c_int32), ("invert", C.c_int32), ("is_mix", C.c_int32), ("blend_mode", C.c_int32),
        ("layer_depth", C.c_int32), ("reserved_", C.c_int32),
    ]


class Format(C.Structure):  # cghip_format
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("square_width", C.c_int32),
                ("square_height", C.c_int32), ("bits", C.c_int32), ("field_mode", C.c_int32)]


class Surface(C.Structure):  # cghip_surface
    _fields_ = [("channels", C.c_int32), ("reserved_", C.c_int32)]


class Draw(C.Structure):  # cghip_draw
    _fields_ = [
        ("pixel_format", C.c_int32), ("nplanes", C.c_int32), ("planes", C.c_void_p * 4),
        ("source_surface", C.c_int32), ("target", C.c_int32), ("local_key", C.c_int32), ("layer_key", C.c_int32),
        ("blend_mode", C.c_int32), ("keyer", C.c_int32), ("geometry_is_quad", C.c_int32), ("ncoords", C.c_int32),
        ("coords", Coord * 4), ("transform", Transform), ("aspect_ratio", C.c_double),
    ]


class HostPlane(C.Structure):  # cghost_plane
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32), ("bits", C.c_int32),
                ("data", C.c_void_p)]


def library_paths():
    return [LIBCGHIP, LIBCGHOST]


def build(quiet=True):
    """Compile libcghip.so + libcghost.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", os.path.join(HERE, "csrc")] + (["-s"] if quiet else []), check=True)


_lib = None
_host = None


def _preload_torch_hip():
    # torch bundles its own libamdhip64.so.7; loading it first makes the dynamic loader resolve our
    # DT_NEEDED to the same runtime, so one process never holds two HIP runtimes.
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def lib():
    """libcghip.so with argtypes set. Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBCGHIP):
            raise ImportError("server_amd/libcghip.so is missing: run __graft_entry__.build() "
                              "(make -C server_amd/csrc); the mixer has no CPU fallback")
        _preload_torch_hip()
        L = C.CDLL(LIBCGHIP, mode=C.RTLD_GLOBAL)
        L.cghip_last_error.restype = C.c_char_p
        L.cghip_device_count.argtypes = [C.POINTER(C.c_int)]
        L.cghip_device_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.cghip_device_destroy.argtypes = [C.c_void_p]
        L.cghip_device_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.cghip_device_gc.argtypes = [C.c_void_p]
        L.cghip_device_synchronize.argtypes = [C.c_void_p]
        L.cghip_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.cghip_host_release.argtypes = [C.c_void_p]
        L.cghip_plane_upload_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.POINTER(C.c_void_p)]
        L.cghip_plane_retain.argtypes = [C.c_void_p]
        L.cghip_plane_release.argtypes = [C.c_void_p]
        L.cghip_plane_desc.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 5
        L.cghip_plane_peer_copy_async.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.cghip_mixer_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.cghip_mixer_destroy.argtypes = [C.c_void_p]
        L.cghip_mixer_render_async.argtypes = [C.c_void_p, C.POINTER(Format), C.POINTER(Surface), C.c_int,
                                               C.POINTER(Draw), C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
        L.cghip_mixer_render_noalloc.argtypes = [C.c_void_p, C.POINTER(Format), C.POINTER(Surface), C.c_int,
                                                 C.POINTER(Draw), C.c_int, C.c_void_p, C.c_size_t]
        L.cghip_ticket_wait.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.cghip_ticket_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.cghip_ticket_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.cghip_ticket_release.argtypes = [C.c_void_p]
        L.cghip_debug_plan.argtypes = [C.POINTER(Format), C.POINTER(Surface), C.c_int, C.POINTER(Draw), C.c_int,
                                       C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.cghip_debug_plane_stub.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        _lib = L
    return _lib


def host():
    """libcghost.so: the C++ host mixer (server_amd/host/) behind a flat shim."""
    global _host
    if _host is None:
        lib()
        if not os.path.exists(LIBCGHOST):
            raise ImportError("server_amd/libcghost.so is missing: run __graft_entry__.build()")
        Hh = C.CDLL(LIBCGHOST)
        Hh.cghost_last_error.restype = C.c_char_p
        Hh.cghost_mixer_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        Hh.cghost_mixer_destroy.argtypes = [C.c_void_p]
        Hh.cghost_mixer_push.argtypes = [C.c_void_p, C.POINTER(Transform)]
        Hh.cghost_mixer_visit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(HostPlane), C.POINTER(Coord),
                                          C.c_int, C.c_int]
        Hh.cghost_mixer_pop.argtypes = [C.c_void_p]
        Hh.cghost_mixer_render.argtypes = [C.c_void_p, C.POINTER(Format), C.c_void_p, C.c_size_t, C.c_int,
                                           C.POINTER(C.c_float)]
        Hh.cghost_mixer_flatten.argtypes = [C.c_void_p, C.POINTER(Format), C.POINTER(Draw), C.c_int,
                                            C.POINTER(C.c_int), C.POINTER(Surface), C.POINTER(C.c_int)]
        Hh.cghost_frame_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(HostPlane), C.POINTER(Coord), C.c_int,
                                           C.POINTER(C.c_void_p)]
        Hh.cghost_frame_destroy.argtypes = [C.c_void_p]
        Hh.cghost_mixer_visit_frame.argtypes = [C.c_void_p, C.c_void_p]
        Hh.cghost_mixer_render_async.argtypes = [C.c_void_p, C.POINTER(Format), C.c_int, C.POINTER(C.c_void_p)]
        Hh.cghost_mixer_render_noalloc.argtypes = [C.c_void_p, C.POINTER(Format), C.c_void_p, C.c_size_t]
        Hh.cghost_future_get.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_float)]
        Hh.cghost_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        Hh.cghost_device_gc.argtypes = [C.c_int]
        Hh.cghost_mixer_device_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        _host = Hh
    return _host


def check(rc):
    if rc != 0:
        raise CGHIPError(rc, lib().cghip_last_error().decode(errors="replace"))


def check_host(rc):
    if rc != 0:
        raise CGHIPError(rc, host().cghost_last_error().decode(errors="replace"))


def device_count():
    n = C.c_int(0)
    rc = lib().cghip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def default_transform(**kw):
    """core::image_transform defaults (core/frame/frame_transform.h:77-99)."""
    t = Transform()
    t.opacity = t.contrast = t.brightness = t.saturation = 1.0
    t.fill_scale[0] = t.fill_scale[1] = 1.0
    t.clip_scale[0] = t.clip_scale[1] = 1.0
    t.crop_lr[0] = t.crop_lr[1] = 1.0
    t.pers_ur[0] = 1.0
    t.pers_lr[0] = t.pers_lr[1] = 1.0
    t.pers_ll[1] = 1.0
    t.levels_max_input = t.levels_gamma = t.levels_max_output = 1.0
    t.chroma_spill_suppress_saturation = 1.0
    for k, v in kw.items():
        cur = getattr(t, k)
        if isinstance(cur, C.Array):
            for i, x in enumerate(v):
                cur[i] = x
        elif k == "blend_mode" and isinstance(v, str):
            t.blend_mode = BLEND[v]
        else:
            setattr(t, k, v)
    return t


DEFAULT_COORDS = [(0.0, 0.0, 0.0, 0.0), (1.0, 0.0, 1.0, 0.0), (1.0, 1.0, 1.0, 1.0), (0.0, 1.0, 0.0, 1.0)]
VFLIP_COORDS = [(0.0, 0.0, 0.0, 1.0), (1.0, 0.0, 1.0, 1.0), (1.0, 1.0, 1.0, 0.0), (0.0, 1.0, 0.0, 0.0)]


class Mixer:
    """The C++ accelerator::hip::image_mixer, scripted. gpu=-2 builds a dry-run mixer (flatten only); gpu=-100-g a
    mixer on a device object of its own on GPU g (frames routed between it and other mixers are peer-copied)."""

    def __init__(self, channel_id=1, gpu=0):
        self.H = host()
        h = C.c_void_p()
        check_host(self.H.cghost_mixer_create(channel_id, gpu, C.byref(h)))
        self.h = h
        self._keep = []
        self.last_kernel_ms = 0.0

    def device_info(self):
        """info of the device object this mixer renders on, as a dict (device::info, device.cpp:335-416)"""
        import json
        buf = C.create_string_buffer(4096)
        check_host(self.H.cghost_mixer_device_info(self.h, buf, 4096))
        return json.loads(buf.value.decode())

    def close(self):
        if getattr(self, "h", None):
            self.H.cghost_mixer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, transform):
        check_host(self.H.cghost_mixer_push(self.h, C.byref(transform)))

    def visit(self, fmt, planes, coords=None, via_factory=True):
        """planes: numpy arrays (h,w[,channels]) uint8/uint16; coords: 4 tuples (vx,vy,tx,ty[,r,q])."""
        pf = PF[fmt] if isinstance(fmt, str) else int(fmt)
        hp = (HostPlane * max(1, len(planes)))()
        for i, p in enumerate(planes):
            a = np.ascontiguousarray(p)
            self._keep.append(a)
            hp[i].height, hp[i].width = a.shape[0], a.shape[1]
            hp[i].stride = 1 if a.ndim == 2 else a.shape[2]
            hp[i].bits = 8 * a.dtype.itemsize
            hp[i].data = a.ctypes.data
        cs = DEFAULT_COORDS if coords is None else coords
        cc = (Coord * 4)()
        for i, c in enumerate(cs):
            v = list(c) + ([0.0, 1.0] if len(c) == 4 else [])
            (cc[i].vertex_x, cc[i].vertex_y, cc[i].texture_x, cc[i].texture_y, cc[i].texture_r, cc[i].texture_q) = v
        check_host(self.H.cghost_mixer_visit(self.h, pf, len(planes), hp, cc, len(cs), 1 if via_factory else 0))

    def _host_planes(self, planes, keep):
        hp = (HostPlane * max(1, len(planes)))()
        for i, p in enumerate(planes):
            a = np.ascontiguousarray(p)
            keep.append(a)
            hp[i].height, hp[i].width = a.shape[0], a.shape[1]
            hp[i].stride = 1 if a.ndim == 2 else a.shape[2]
            hp[i].bits = 8 * a.dtype.itemsize
            hp[i].data = a.ctypes.data
        return hp

    def create_frame(self, fmt, planes, coords=None):
        """create_frame + fill + commit: returns a handle to a const_frame whose planes stay resident in HBM."""
        pf = PF[fmt] if isinstance(fmt, str) else int(fmt)
        keep = []
        hp = self._host_planes(planes, keep)
        cc = (Coord * 4)()
        n = 0
        if coords is not None:
            n = 4
            for i, c in enumerate(coords):
                v = list(c) + ([0.0, 1.0] if len(c) == 4 else [])
                (cc[i].vertex_x, cc[i].vertex_y, cc[i].texture_x, cc[i].texture_y, cc[i].texture_r, cc[i].texture_q) = v
        h = C.c_void_p()
        check_host(self.H.cghost_frame_create(self.h, pf, len(planes), hp, cc, n, C.byref(h)))
        return h

    def destroy_frame(self, frame):
        self.H.cghost_frame_destroy(frame)

    def visit_frame(self, frame):
        check_host(self.H.cghost_mixer_visit_frame(self.h, frame))

    def render_async(self, width, height, square_width=None, square_height=None, bits=8, field_mode=0,
                     readback=True):
        fmt = self._format(width, height, square_width, square_height, bits, field_mode)
        h = C.c_void_p()
        check_host(self.H.cghost_mixer_render_async(self.h, C.byref(fmt), 1 if readback else 0, C.byref(h)))
        shape = (height, width, 4)
        return (h, shape, np.uint8 if bits == 8 else np.uint16, readback)

    def get(self, pending, want_output=True):
        h, shape, dt, readback = pending
        out = np.zeros(shape, dtype=dt) if (readback and want_output) else None
        ms = C.c_float(0.0)
        check_host(self.H.cghost_future_get(h, out.ctypes.data if out is not None else None,
                                            out.nbytes if out is not None else 0, C.byref(ms)))
        self.last_kernel_ms = ms.value
        return out

    def pop(self):
        check_host(self.H.cghost_mixer_pop(self.h))

    def _format(self, width, height, square_width, square_height, bits, field_mode):
        return Format(width, height, square_width or width, square_height or height, bits, field_mode)

    def render(self, width, height, square_width=None, square_height=None, bits=8, field_mode=0, readback=True):
        fmt = self._format(width, height, square_width, square_height, bits, field_mode)
        out = np.zeros((height, width, 4), dtype=np.uint8 if bits == 8 else np.uint16) if readback else None
        ms = C.c_float(0.0)
        check_host(self.H.cghost_mixer_render(self.h, C.byref(fmt), out.ctypes.data if readback else None,
                                              out.nbytes if readback else 0, 1 if readback else 0, C.byref(ms)))
        self.last_kernel_ms = ms.value
        self._keep = []
        return out

    def render_noalloc(self, device_ptr, nbytes, width, height, square_width=None, square_height=None, bits=8,
                       field_mode=0):
        """The pending scene into device memory of the caller through cghip_mixer_render_noalloc (blocking)."""
        fmt = self._format(width, height, square_width, square_height, bits, field_mode)
        check_host(self.H.cghost_mixer_render_noalloc(self.h, C.byref(fmt), C.c_void_p(device_ptr), C.c_size_t(nbytes)))
        self._keep = []

    def flatten(self, width, height, square_width=None, square_height=None, bits=8, field_mode=0):
        fmt = self._format(width, height, square_width, square_height, bits, field_mode)
        draws = (Draw * 256)()
        surf = (Surface * 1024)()  # CGHIP_MAX_SURFACES
        nd, ns = C.c_int(0), C.c_int(0)
        check_host(self.H.cghost_mixer_flatten(self.h, C.byref(fmt), draws, 256, C.byref(nd), surf, C.byref(ns)))
        return [draws[i] for i in range(nd.value)], [surf[i].channels for i in range(ns.value)]
