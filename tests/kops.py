"""ctypes mirror of server_amd/csrc/cghip_kops.h (KOp), for planner introspection in tests."""
import ctypes as C


class KPlane(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("w", C.c_int32), ("h", C.c_int32), ("channels", C.c_int32), ("bits", C.c_int32),
                ("fw", C.c_float), ("fh", C.c_float)]


class KTri(C.Structure):
    _fields_ = [("e", (C.c_double * 3) * 3), ("a", (C.c_double * 3) * 3)]


class KOp(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("flags", C.c_uint32), ("dst", C.c_int32), ("src_surface", C.c_int32),
        ("local_key", C.c_int32), ("layer_key", C.c_int32), ("pixel_format", C.c_int32), ("blend_mode", C.c_int32),
        ("nplanes", C.c_int32), ("ntri", C.c_int32),
        ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
        ("unit_dx", C.c_int32), ("unit_dy", C.c_int32),
        ("opacity", C.c_float),
        ("lv_min_input", C.c_float), ("lv_inv_range", C.c_float), ("lv_inv_gamma", C.c_float),
        ("lv_min_output", C.c_float), ("lv_max_output", C.c_float),
        ("brt", C.c_float), ("sat", C.c_float), ("con", C.c_float),
        ("ch_target_hue", C.c_float), ("ch_hue_width", C.c_float), ("ch_min_saturation", C.c_float),
        ("ch_min_brightness", C.c_float), ("ch_softness", C.c_float), ("ch_spill", C.c_float),
        ("ch_spill_saturation", C.c_float),
        ("fast_class", C.c_int32), ("ch_inv_spill", C.c_float), ("ch_inv_soft", C.c_float), ("sample_class", C.c_int32),
        ("csx", C.c_int32), ("csy", C.c_int32), ("lv_range", C.c_float), ("ch_soft_range", C.c_float),
        ("ycoef", C.c_float * 8), ("fetch_body", C.c_int32), ("blend_body", C.c_int32),
        ("fused", C.c_int32), ("b_blend_mode", C.c_int32), ("b_local_key", C.c_int32), ("b_layer_key", C.c_int32),
        ("b_flags", C.c_uint32), ("b_opacity", C.c_float),
        ("ax0", C.c_int32), ("ay0", C.c_int32), ("ax1", C.c_int32), ("ay1", C.c_int32),
        ("planes", KPlane * 4), ("tri", KTri * 2),
    ]


KOP_CLEAR, KOP_DRAW_PLANES, KOP_DRAW_SURFACE, KOP_SAVE_KEY, KOP_LOAD_KEY = 0, 1, 2, 3, 4
(KF_LEVELS, KF_CSB, KF_CHROMA, KF_CHROMA_MASK, KF_INVERT, KF_IS_HD, KF_PROJECTIVE, KF_RECT, KF_UNIT_SCALE, KF_ADDITIVE,
 KF_SOLID, KF_EDGE, KF_RED_ONLY, KF_AXIS) = [1 << i for i in range(14)]
FC_NONE, FC_PACKED, FC_PLANAR, FC_SURFACE, FC_SAMPLED, FC_UYVY, FC_BYTES = range(7)
KS_RGBA0, KS_RGBA1, KS_RGBA2, KS_KEY0, KS_KEY1 = range(5)


def plan(fmt, surfaces, draws):
    """Run the product's host planner on a draw list (no GPU). Returns (list[KOp], full)."""
    import server_amd as A
    L = A.lib()
    assert L.cghip_debug_kop_size() == C.sizeof(KOp), (L.cghip_debug_kop_size(), C.sizeof(KOp))
    sa = (A.Surface * len(surfaces))(*[A.Surface(c, 0) for c in surfaces])
    da = (A.Draw * max(1, len(draws)))(*draws)
    out = (KOp * 512)()
    n, full = C.c_int(0), C.c_int(0)
    rc = L.cghip_debug_plan(C.byref(fmt), sa, len(surfaces), da, len(draws), out, 512, C.byref(n), C.byref(full))
    if rc != 0:
        raise A.CGHIPError(rc, L.cghip_last_error().decode())
    global last_plan_fast
    last_plan_fast = bool(full.value & 2)  # every op qualifies for cghip_composite_fast
    return [out[i] for i in range(n.value)], bool(full.value & 1)


last_plan_fast = False


SEG_FIELDS = ("first", "count", "fast", "full", "load_mask", "store_mask", "gx0", "gy0", "gx1", "gy1",
              "key_ld0", "key_ld1", "key_st0", "key_st1")


def plan_segments(fmt, surfaces, draws):
    """The kernel launches the planner cuts a frame into (no GPU): list of dicts with SEG_FIELDS."""
    import server_amd as A
    L = A.lib()
    sa = (A.Surface * len(surfaces))(*[A.Surface(c, 0) for c in surfaces])
    da = (A.Draw * max(1, len(draws)))(*draws)
    out = (C.c_int * (14 * 64))()
    n = L.cghip_debug_plan_segments(C.byref(fmt), sa, len(surfaces), da, len(draws), out, 64)
    if n < 0:
        raise A.CGHIPError(n, L.cghip_last_error().decode())
    assert n <= 64
    return [dict(zip(SEG_FIELDS, out[i * 14:(i + 1) * 14])) for i in range(n)]


def set_cut_policy(cut_cost_scale=1.0, max_segments=8, sampled_fast=True):
    """sampled_fast=False: scaled / rotated / perspective draws stay with the 1-pixel-per-lane kernel (as they did
    before the 4-pixels-per-lane kernel learned to sample), so that scenes mixing them with unit-scale layers still
    exercise the cutting of a frame into launches; today only exotic pixel formats and inverted partial quads do."""
    import server_amd as A
    L = A.lib()
    L.cghip_debug_no_sampled_fast(0 if sampled_fast else 1)
    L.cghip_debug_cut_cost_scale.argtypes = [C.c_double]
    L.cghip_debug_cut_cost_scale.restype = None
    L.cghip_debug_cut_cost_scale(float(cut_cost_scale))
    L.cghip_debug_max_segments(int(max_segments))
