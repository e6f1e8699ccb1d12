"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding of the checker libraries (oracle/liboracle.so, oracle/_ref/libglref.so). Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product (server_amd/)
never does.

The structures mirror oracle/ref_types.h, which restates the reference's value types
(core/frame/frame_transform.h:34-109, core/frame/geometry.h:38-46, core/frame/pixel_format.h:28-76).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_SHADER_DIR = "/root/reference/src/accelerator/ogl/image"

PF = dict(gray=0, bgra=1, rgba=2, argb=3, abgr=4, ycbcr=5, ycbcra=6, luma=7, bgr=8, rgb=9, uyvy=10, count=11,
          invalid=12)
BLEND = dict(normal=0, lighten=1, darken=2, multiply=3, average=4, add=5, subtract=6, difference=7, negation=8,
             exclusion=9, screen=10, overlay=11, soft_light=12, hard_light=13, color_dodge=14, color_burn=15,
             linear_dodge=16, linear_burn=17, linear_light=18, vivid_light=19, pin_light=20, hard_mix=21,
             reflect=22, glow=23, phoenix=24, contrast=25, saturation=26, color=27, luminosity=28, mix=29)


class Coord(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("vertex_x", "vertex_y", "texture_x", "texture_y", "texture_r", "texture_q")]


class Transform(C.Structure):
    _fields_ = [
        ("opacity", C.c_double), ("contrast", C.c_double), ("brightness", C.c_double), ("saturation", C.c_double),
        ("anchor", C.c_double * 2), ("fill_translation", C.c_double * 2), ("fill_scale", C.c_double * 2),
        ("clip_translation", C.c_double * 2), ("clip_scale", C.c_double * 2), ("angle", C.c_double),
        ("crop_ul", C.c_double * 2), ("crop_lr", C.c_double * 2),
        ("pers_ul", C.c_double * 2), ("pers_ur", C.c_double * 2), ("pers_lr", C.c_double * 2),
        ("pers_ll", C.c_double * 2),
        ("lv_min_input", C.c_double), ("lv_max_input", C.c_double), ("lv_gamma", C.c_double),
        ("lv_min_output", C.c_double), ("lv_max_output", C.c_double),
        ("chroma_enable", C.c_int32), ("chroma_show_mask", C.c_int32),
        ("chroma_target_hue", C.c_double), ("chroma_hue_width", C.c_double),
        ("chroma_min_saturation", C.c_double), ("chroma_min_brightness", C.c_double),
        ("chroma_softness", C.c_double), ("chroma_spill_suppress", C.c_double),
        ("chroma_spill_suppress_saturation", C.c_double),
        ("is_key", C.c_int32), ("invert", C.c_int32), ("is_mix", C.c_int32), ("blend_mode", C.c_int32),
        ("layer_depth", C.c_int32), ("pad_", C.c_int32),
    ]


class Plane(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32), ("bits", C.c_int32),
                ("data", C.c_void_p)]


class Frame(C.Structure):
    _fields_ = [("format", C.c_int32), ("nplanes", C.c_int32), ("planes", Plane * 4),
                ("geometry_is_quad", C.c_int32), ("ncoords", C.c_int32), ("coords", Coord * 4)]


class Format(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("square_width", C.c_int32),
                ("square_height", C.c_int32), ("bits", C.c_int32), ("field_mode", C.c_int32)]


class Draw(C.Structure):
    _fields_ = [
        ("pixel_format", C.c_int32), ("nplanes", C.c_int32), ("planes", C.c_void_p * 4),
        ("background", C.c_void_p), ("local_key", C.c_void_p), ("layer_key", C.c_void_p),
        ("is_hd", C.c_int32), ("has_local_key", C.c_int32), ("has_layer_key", C.c_int32),
        ("blend_mode", C.c_int32), ("keyer", C.c_int32), ("invert", C.c_int32),
        ("opacity", C.c_float),
        ("levels", C.c_int32), ("min_input", C.c_float), ("max_input", C.c_float), ("min_output", C.c_float),
        ("max_output", C.c_float), ("gamma", C.c_float),
        ("csb", C.c_int32), ("brt", C.c_float), ("sat", C.c_float), ("con", C.c_float),
        ("chroma", C.c_int32), ("chroma_show_mask", C.c_int32),
        ("chroma_target_hue", C.c_float), ("chroma_hue_width", C.c_float), ("chroma_min_saturation", C.c_float),
        ("chroma_min_brightness", C.c_float), ("chroma_softness", C.c_float), ("chroma_spill_suppress", C.c_float),
        ("chroma_spill_suppress_saturation", C.c_float),
        ("scissor", C.c_int32), ("sc_x", C.c_int32), ("sc_y", C.c_int32), ("sc_w", C.c_int32), ("sc_h", C.c_int32),
        ("field_mode", C.c_int32),
        ("coords", Coord * 4),
    ]


class Backend(C.Structure):
    _fields_ = [
        ("tex_create", C.CFUNCTYPE(C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)),
        ("tex_upload", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)),
        ("tex_read", C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)),
        ("tex_destroy", C.CFUNCTYPE(None, C.c_void_p)),
        ("tex_width", C.CFUNCTYPE(C.c_int, C.c_void_p)),
        ("tex_height", C.CFUNCTYPE(C.c_int, C.c_void_p)),
        ("draw", C.CFUNCTYPE(None, C.POINTER(Draw))),
    ]


def build(quiet=True):
    """Compile the checker libraries (make -C oracle)."""
    subprocess.run(["make", "-C", HERE] + (["-s"] if quiet else []), check=True)


_lib = None
_gl = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.ref_mixer_create.restype = C.c_void_p
        L.ref_mixer_create.argtypes = [C.POINTER(Backend)]
        L.ref_mixer_destroy.argtypes = [C.c_void_p]
        L.ref_mixer_push.argtypes = [C.c_void_p, C.POINTER(Transform)]
        L.ref_mixer_visit.argtypes = [C.c_void_p, C.POINTER(Frame)]
        L.ref_mixer_visit.restype = C.c_int
        L.ref_mixer_pop.argtypes = [C.c_void_p]
        L.ref_mixer_render.argtypes = [C.c_void_p, C.POINTER(Format), C.c_void_p]
        L.ref_mixer_render.restype = C.c_int
        L.ref_mixer_enable_log.argtypes = [C.c_void_p, C.c_int]
        L.ref_mixer_log_size.argtypes = [C.c_void_p]
        L.ref_mixer_log_size.restype = C.c_int
        L.ref_mixer_log.argtypes = [C.c_void_p]
        L.ref_mixer_log.restype = C.POINTER(Draw)
        L.ref_transform_default.argtypes = [C.POINTER(Transform)]
        L.ref_transform_compose.argtypes = [C.POINTER(Transform), C.POINTER(Transform)]
        L.ref_geometry_default.argtypes = [Coord * 4]
        L.ref_geometry_default_vflip.argtypes = [Coord * 4]
        L.cpuref_backend.restype = C.POINTER(Backend)
        L.cpuref_set_threads.argtypes = [C.c_int]
        L.cpuref_set_background_bilinear.argtypes = [C.c_int]
        L.cpuref_set_fetch_delta.argtypes = [C.c_float] * 4
        L.cpuref_set_snap_fetch.argtypes = [C.c_int]
        L.cpuref_set_store_flip.argtypes = [C.c_uint]
        L.cpuref_set_store_flip_draw.argtypes = [C.c_int] * 5
        L.cpuref_draw_count.restype = C.c_int
        L.cpuref_set_pow_ulps.argtypes = [C.c_int]
        _lib = L
    return _lib


def gl_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libglref.so")) and os.path.isdir(REFERENCE_SHADER_DIR)


def gl():
    """The llvmpipe harness (reference GLSL as data). Only exists in the authoring container."""
    global _gl
    if _gl is None:
        if not gl_available():
            raise RuntimeError("llvmpipe harness unavailable (no oracle/_ref/libglref.so or no /root/reference)")
        os.environ["force_compat_shaders"] = "true"
        G = C.CDLL(os.path.join(HERE, "_ref", "libglref.so"))
        G.glref_init.argtypes = [C.c_char_p]
        G.glref_info.restype = C.c_char_p
        G.glref_backend.restype = C.POINTER(Backend)
        rc = G.glref_init(REFERENCE_SHADER_DIR.encode())
        if rc != 0:
            raise RuntimeError("glref_init failed: %d" % rc)
        _gl = G
    return _gl


def backend(name):
    if name == "cpu":
        return lib().cpuref_backend()
    if name == "gl":
        return gl().glref_backend()
    raise ValueError(name)


def default_transform(**kw):
    t = Transform()
    lib().ref_transform_default(C.byref(t))
    set_fields(t, kw)
    return t


def set_fields(t, kw):
    for k, v in kw.items():
        cur = getattr(t, k)
        if isinstance(cur, C.Array):
            for i, x in enumerate(v):
                cur[i] = x
        elif k == "blend_mode" and isinstance(v, str):
            setattr(t, k, BLEND[v])
        else:
            setattr(t, k, v)


def make_frame(fmt, planes, coords=None, vflip=False):
    """planes: list of np arrays shaped (h, w, channels) or (h, w); dtype uint8 or uint16."""
    f = Frame()
    f.format = PF[fmt] if isinstance(fmt, str) else fmt
    f.nplanes = len(planes)
    keep = []
    for i, p in enumerate(planes):
        a = np.ascontiguousarray(p)
        keep.append(a)
        h, w = a.shape[:2]
        f.planes[i].width, f.planes[i].height = w, h
        f.planes[i].stride = 1 if a.ndim == 2 else a.shape[2]
        f.planes[i].bits = 8 * a.dtype.itemsize
        f.planes[i].data = a.ctypes.data
    f.geometry_is_quad = 1
    f.ncoords = 4
    g = (Coord * 4)()
    if vflip:
        lib().ref_geometry_default_vflip(g)
    else:
        lib().ref_geometry_default(g)
    if coords is not None:
        for i, c in enumerate(coords):
            vals = list(c)
            if len(vals) == 4:
                vals += [0.0, 1.0]
            (g[i].vertex_x, g[i].vertex_y, g[i].texture_x, g[i].texture_y, g[i].texture_r, g[i].texture_q) = vals
    for i in range(4):
        f.coords[i] = g[i]
    f._keep = keep
    return f


class RefMixer:
    """Drives the restated reference mixer (push/visit/pop/render) on a backend ('cpu' or 'gl')."""

    def __init__(self, backend_name="cpu"):
        self.L = lib()
        self.be = backend(backend_name)
        self.h = self.L.ref_mixer_create(self.be)
        self._keep = []

    def close(self):
        if self.h:
            self.L.ref_mixer_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def push(self, t):
        self.L.ref_mixer_push(self.h, C.byref(t))

    def visit(self, frame):
        self._keep.append(frame)
        return self.L.ref_mixer_visit(self.h, C.byref(frame))

    def pop(self):
        self.L.ref_mixer_pop(self.h)

    def render(self, width, height, square_width=None, square_height=None, bits=8, field_mode=0, log=False):
        fmt = Format(width, height, square_width or width, square_height or height, bits, field_mode)
        out = np.zeros((height, width, 4), dtype=np.uint8 if bits == 8 else np.uint16)
        self.L.ref_mixer_enable_log(self.h, 1 if log else 0)
        n = self.L.ref_mixer_render(self.h, C.byref(fmt), out.ctypes.data)
        self._keep = []
        self.ndraws = n
        if log:
            p = self.L.ref_mixer_log(self.h)
            self.draw_log = [Draw.from_buffer_copy(p[i]) for i in range(self.L.ref_mixer_log_size(self.h))]
        return out
