"""Seeded synthetic scenes shared by the oracle tests, the golden generator, the GPU parity tests and bench.py.

A scene is a dict: {name, width, height, [square_width, square_height], bits, ops} where ops is the sequence of
frame_visitor calls core::mixer would make (core/mixer/mixer.cpp:62-68, core/frame/draw_frame.cpp:46-71):
    ("push", {image_transform fields})      ("visit", frame_spec)      ("pop",)
frame_spec = {fmt, planes: [(w, h, channels, kind, seed)], bits, [coords]}.
Field names in transforms follow core/frame/frame_transform.h; the runners translate them to each side's POD.
"""
import numpy as np

MASK64 = (1 << 64) - 1


def _splitmix(idx, seed):
    """Deterministic per-element hash (splitmix64), independent of numpy's Generator streams."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & MASK64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def make_plane(w, h, channels, kind, seed, bits=8):
    """kind: 'noise' (max contrast), 'smooth' (gradients + mild noise), 'video' (legal-range smooth),
    'premul' (4-channel noise, rgb premultiplied by a), 'solid:<v>'."""
    maxv = (1 << bits) - 1
    dt = np.uint8 if bits == 8 else np.uint16
    idx = np.arange(h * w * channels, dtype=np.uint64).reshape(h, w, channels)
    rnd = (_splitmix(idx, seed) >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1)
    if kind == "noise":
        a = rnd
    elif kind in ("smooth", "video"):
        # separable gradients (outer products of 1-D vectors: cheap at 2160p) + mild per-sample noise
        x = np.arange(w, dtype=np.float64)
        y = np.arange(h, dtype=np.float64)
        a = np.empty((h, w, channels), dtype=np.float64)
        for k in range(channels):
            fx = np.sin(x / (9.0 + 2 * k + seed % 5) + k + seed)
            fy = np.cos(y / (7.0 + k + seed % 3) - seed)
            a[..., k] = 0.5 + 0.38 * np.outer(fy, fx) + 0.1 * (x / max(1, w - 1) - 0.5)[None, :]
        a += (rnd - 0.5) * 0.04
        np.clip(a, 0.0, 1.0, out=a)
        if kind == "video":
            a = (16.0 + a * 219.0) / 255.0
    elif kind == "premul":
        assert channels == 4
        a = rnd.copy()
        a[..., :3] = a[..., :3] * a[..., 3:4]
    elif kind.startswith("solid:"):
        a = np.full((h, w, channels), float(kind.split(":")[1]))
    else:
        raise ValueError(kind)
    out = np.floor(a * maxv + 0.5).clip(0, maxv).astype(dt)
    if kind == "premul":  # keep rgb <= a after rounding
        out[..., :3] = np.minimum(out[..., :3], out[..., 3:4])
    return out if channels > 1 else out[..., 0]


def frame(fmt, planes, bits=8, coords=None):
    return {"fmt": fmt, "planes": planes, "bits": bits, "coords": coords}


def bgra(w, h, kind="premul", seed=1, bits=8, coords=None):
    return frame("bgra", [(w, h, 4, kind, seed)], bits, coords)


def ycbcr(w, h, sub="420", kind="video", seed=1, bits=8, alpha=False):
    cw = w if sub == "444" else w // 2
    chh = h // 2 if sub == "420" else h
    planes = [(w, h, 1, kind, seed), (cw, chh, 1, kind, seed + 101), (cw, chh, 1, kind, seed + 202)]
    if alpha:
        planes.append((w, h, 1, "smooth", seed + 303))
    return frame("ycbcra" if alpha else "ycbcr", planes, bits)


def layer(spec, **transform):
    """One stage layer the way core::mixer submits it: layer_depth = 1 (mixer.cpp:66)."""
    t = dict(transform)
    t.setdefault("layer_depth", 1)
    return [("push", t), ("visit", spec), ("pop",)]


def scene(name, width, height, ops, bits=8, square=None, field_mode=0):
    flat = []
    for o in ops:
        flat.extend(o if isinstance(o, list) else [o])
    return {"name": name, "width": width, "height": height, "bits": bits, "ops": flat,
            "square_width": (square or (width, height))[0], "square_height": (square or (width, height))[1],
            "field_mode": field_mode}


def materialize(spec):
    """frame_spec -> list of numpy planes."""
    out = []
    for (w, h, c, kind, seed) in spec["planes"]:
        out.append(make_plane(w, h, c, kind, seed, spec["bits"]))
    if spec["fmt"] == "uyvy" and len(out) == 1:  # av_util.cpp:163-171: the same bytes bound twice
        p = out[0]
        out = [p, p.reshape(p.shape[0], p.shape[1] // 2, 4)]
    return out


CHROMA = dict(chroma_enable=1, chroma_target_hue=120.0, chroma_hue_width=0.1, chroma_min_saturation=0.2,
              chroma_min_brightness=0.2, chroma_softness=0.1, chroma_spill_suppress=36.0,
              chroma_spill_suppress_saturation=0.5)
LEVELS = dict(levels_min_input=0.06, levels_max_input=0.92, levels_gamma=1.2, levels_min_output=0.0,
              levels_max_output=1.0)

W, H = 96, 64


def small_scenes():
    """The parity suite at sizes the oracle finishes in well under a second each."""
    S = []
    bg = layer(bgra(W, H, "premul", 11))
    S.append(scene("identity_bgra", W, H, [layer(bgra(W, H, "premul", 1))]))
    S.append(scene("config1_colour_plus_image", W, H,
                   [layer(bgra(1, 1, "solid:0.6", 2)), layer(bgra(W, H, "premul", 1), opacity=0.73)]))
    S.append(scene("opacity_half_ties", W, H, [bg, layer(bgra(W, H, "premul", 3), opacity=0.5)]))
    S.append(scene("fill_rule_pixel_centres", 64, 32,
                   [layer(bgra(64, 32, "solid:1.0", 1), fill_scale=[3 / 64., 3 / 32.], fill_translation=[2.5 / 64, 2.5 / 32])]))
    S.append(scene("magnify", W, H, [bg, layer(bgra(48, 32, "premul", 4), fill_scale=[1.7, 1.7], fill_translation=[-0.2, -0.3])]))
    S.append(scene("minify", W, H, [bg, layer(bgra(W, H, "premul", 5), fill_scale=[0.37, 0.37], fill_translation=[0.1, 0.2])]))
    S.append(scene("translate_subpixel", W, H, [bg, layer(bgra(W, H, "premul", 6), fill_translation=[0.013, 0.027])]))
    S.append(scene("rotate", W, H, [bg, layer(bgra(W, H, "smooth", 7), angle=0.3, fill_scale=[0.8, 0.8], fill_translation=[0.2, 0.1])]))
    S.append(scene("rotate_anchor_pal_aspect", 90, 72,
                   [layer(bgra(90, 72, "smooth", 8), angle=-0.5, anchor=[0.5, 0.5], fill_scale=[0.7, 0.6], fill_translation=[0.5, 0.5])],
                   square=(128, 72)))
    S.append(scene("perspective", W, H,
                   [bg, layer(bgra(W, H, "smooth", 9), pers_ul=[0.1, 0.05], pers_ur=[0.9, 0.1], pers_lr=[1.0, 0.95], pers_ll=[0.05, 0.9])]))
    S.append(scene("crop", W, H, [bg, layer(bgra(W, H, "premul", 10), crop_ul=[0.1, 0.2], crop_lr=[0.8, 0.9])]))
    S.append(scene("clip_scissor", W, H, [bg, layer(bgra(W, H, "premul", 12), clip_translation=[0.1, 0.2], clip_scale=[0.5, 0.6])]))
    S.append(scene("crop_scale_translate", W, H,
                   [bg, layer(bgra(W, H, "smooth", 13), crop_ul=[0.05, 0.1], crop_lr=[0.9, 0.95], fill_scale=[0.6, 0.5], fill_translation=[0.3, 0.4])]))
    S.append(scene("vflip_geometry", W, H, [layer(frame("bgra", [(W, H, 4, "premul", 14)], 8,
                                                       [(0, 0, 0, 1), (1, 0, 1, 1), (1, 1, 1, 0), (0, 1, 0, 0)]))]))
    S.append(scene("custom_geometry", W, H, [bg, layer(frame("bgra", [(W, H, 4, "smooth", 15)], 8,
                                                             [(0.1, 0.1, 0, 0), (0.8, 0.2, 1, 0), (0.9, 0.9, 1, 1), (0.2, 0.7, 0, 1)]))]))
    S.append(scene("offscreen_cull_and_tiny_opacity", W, H,
                   [bg, layer(bgra(W, H, "premul", 16), fill_translation=[1.5, 0.0]), layer(bgra(W, H, "premul", 17), opacity=0.0005)]))
    # pixel formats
    geo = dict(fill_scale=[0.8, 0.7], fill_translation=[0.1, 0.13], opacity=0.9)
    for kind in ("smooth", "noise"):
        S.append(scene("fmt_gray_" + kind, W, H, [bg, layer(frame("gray", [(48, 32, 1, kind, 20)]), **geo)]))
        S.append(scene("fmt_luma_" + kind, W, H, [bg, layer(frame("luma", [(48, 32, 1, kind, 21)]), **geo)]))
        for f in ("rgba", "argb", "abgr"):
            S.append(scene("fmt_%s_%s" % (f, kind), W, H, [bg, layer(frame(f, [(48, 32, 4, kind, 22)]), **geo)]))
        S.append(scene("fmt_bgr_" + kind, W, H, [bg, layer(frame("bgr", [(48, 32, 3, kind, 23)]), **geo)]))
        S.append(scene("fmt_rgb_" + kind, W, H, [bg, layer(frame("rgb", [(50, 32, 3, kind, 24)]), **geo)]))
        S.append(scene("fmt_ycbcr420_" + kind, W, H, [bg, layer(ycbcr(W, H, "420", kind, 25), opacity=0.8)]))
        S.append(scene("fmt_ycbcr422_geo_" + kind, W, H, [bg, layer(ycbcr(48, 32, "422", kind, 26), **geo)]))
        S.append(scene("fmt_ycbcra444_" + kind, W, H, [bg, layer(ycbcr(W, H, "444", kind, 27, alpha=True))]))
        S.append(scene("fmt_uyvy_" + kind, W, H, [bg, layer(frame("uyvy", [(48, 32, 2, kind, 28)]), **geo)]))
    S.append(scene("ycbcr_hd_matrix", 64, 720, [layer(ycbcr(64, 720, "422", "video", 29))]))
    # adjustments
    S.append(scene("levels_ycbcr", W, H, [bg, layer(ycbcr(W, H, "420", "video", 30), opacity=0.8, **LEVELS)]))
    S.append(scene("levels_output_range", W, H,
                   [bg, layer(bgra(W, H, "noise", 31), levels_min_input=0.1, levels_max_input=0.8, levels_gamma=0.6,
                              levels_min_output=0.1, levels_max_output=0.9)]))
    S.append(scene("csb_bgra", W, H, [bg, layer(bgra(W, H, "premul", 32), brightness=1.2, saturation=0.5, contrast=1.3)]))
    S.append(scene("csb_ycbcr", W, H, [bg, layer(ycbcr(W, H, "420", "video", 33), brightness=0.8, saturation=1.5, contrast=0.7, opacity=0.6)]))
    S.append(scene("chroma_key_video", W, H, [bg, layer(ycbcr(W, H, "422", "smooth", 34), **CHROMA)]))
    S.append(scene("chroma_key_noise", W, H, [bg, layer(bgra(W, H, "noise", 35), **CHROMA)]))
    S.append(scene("chroma_show_mask", W, H, [bg, layer(bgra(W, H, "noise", 36), chroma_show_mask=1, **CHROMA)]))
    S.append(scene("invert", W, H, [bg, layer(bgra(W, H, "premul", 37), invert=1, opacity=0.7)]))
    # keyer layers
    S.append(scene("key_then_fill_same_layer_depth", W, H,
                   [bg, layer(bgra(W, H, "premul", 38), is_key=1), layer(bgra(W, H, "premul", 39), opacity=0.9)]))
    S.append(scene("key_ycbcr_scaled_then_two_fills", W, H,
                   [bg, layer(ycbcr(W, H, "420", "smooth", 40), is_key=1, fill_scale=[0.5, 0.5], fill_translation=[0.25, 0.25]),
                    layer(bgra(W, H, "premul", 41), opacity=0.9), layer(bgra(W, H, "premul", 42), opacity=0.5, blend_mode="screen")]))
    S.append(scene("mix_transition", W, H,
                   [bg, [("push", dict(layer_depth=1)),
                         ("push", dict(is_mix=1, opacity=0.3)), ("visit", bgra(W, H, "premul", 43)), ("pop",),
                         ("push", dict(is_mix=1, opacity=0.7)), ("visit", bgra(W, H, "premul", 44)), ("pop",),
                         ("pop",)]]))
    S.append(scene("local_key_and_fill_in_one_layer", W, H,
                   [bg, [("push", dict(layer_depth=1)),
                         ("push", dict(is_key=1)), ("visit", bgra(W, H, "smooth", 45)), ("pop",),
                         ("push", dict(opacity=0.8)), ("visit", bgra(W, H, "premul", 46)), ("pop",),
                         ("pop",)]]))
    S.append(scene("blend_layer_with_key_and_mix", W, H,
                   [bg, layer(bgra(W, H, "smooth", 47), is_key=1),
                    [("push", dict(layer_depth=1, blend_mode="overlay")),
                     ("push", dict(is_mix=1, opacity=0.4)), ("visit", bgra(W, H, "premul", 48)), ("pop",),
                     ("push", dict(is_mix=1, opacity=0.6)), ("visit", bgra(W, H, "premul", 49)), ("pop",),
                     ("pop",)]]))
    # a layer's mix texture is never reset (image_mixer.cpp:207-225 copies from the named rvalue reference): it is
    # composited again before every later plain item and at the layer's end, and later mix items add to it
    S.append(scene("mix_then_plain_item", W, H,
                   [bg, [("push", dict(layer_depth=1)),
                         ("push", dict(is_mix=1, opacity=0.4)), ("visit", bgra(W, H, "premul", 53)), ("pop",),
                         ("push", dict(opacity=0.5, fill_scale=[0.5, 0.5], fill_translation=[0.25, 0.25])),
                         ("visit", bgra(W, H, "premul", 54)), ("pop",),
                         ("pop",)]]))
    S.append(scene("mix_plain_mix_plain", W, H,
                   [bg, [("push", dict(layer_depth=1)),
                         ("push", dict(is_mix=1, opacity=0.3)), ("visit", bgra(W, H, "premul", 55)), ("pop",),
                         ("push", dict(opacity=0.6)), ("visit", ycbcr(W, H, "422", "video", 56)), ("pop",),
                         ("push", dict(is_mix=1, opacity=0.2)), ("visit", bgra(W, H, "premul", 57)), ("pop",),
                         ("push", dict(opacity=0.3)), ("visit", bgra(W, H, "premul", 58)), ("pop",),
                         ("pop",)]]))
    S.append(scene("mix_then_plain_in_blend_layer", W, H,
                   [bg, [("push", dict(layer_depth=1, blend_mode="screen")),
                         ("push", dict(is_mix=1, opacity=0.4)), ("visit", bgra(W, H, "premul", 59)), ("pop",),
                         ("push", dict(opacity=0.5)), ("visit", bgra(W, H, "premul", 62)), ("pop",),
                         ("pop",)]]))
    S.append(scene("sublayers", W, H,
                   [bg, [("push", dict(layer_depth=1, opacity=0.9)),
                         ("push", dict(layer_depth=1, blend_mode="add", fill_scale=[0.5, 0.5])), ("visit", bgra(W, H, "premul", 50)), ("pop",),
                         ("visit", bgra(W, H, "premul", 51)),
                         ("pop",)]]))
    S.append(scene("nested_transforms_compose", W, H,
                   [bg, [("push", dict(layer_depth=1, fill_scale=[0.8, 0.8], fill_translation=[0.1, 0.1], angle=0.1, opacity=0.9)),
                         ("push", dict(fill_scale=[0.5, 0.9], fill_translation=[0.2, 0.0], brightness=1.1, crop_ul=[0.1, 0.0])),
                         ("visit", bgra(W, H, "smooth", 52)), ("pop",), ("pop",)]]))
    # More than two key textures alive at once (image_mixer.cpp:104-114,153): the key a top-level layer left for the next
    # one (k1) stays pending while that layer's sublayers run their own chain - sublayer S1 leaves kS1 for S2, S2 draws a
    # local key kS2 and a fill that uses kS2 and kS1 - and is consumed by the layer's own item afterwards. The kernels
    # hold two key surfaces: the planner parks the third in HBM across a cut (KOP_SAVE_KEY / KOP_LOAD_KEY).
    def nested(depth, seed):
        inner = [("push", dict(layer_depth=1)),
                 ("push", dict(is_key=1)), ("visit", ycbcr(W, H, "422", "smooth", seed + 1)), ("pop",),
                 ("push", dict(opacity=0.8)), ("visit", bgra(W, H, "premul", seed + 2)), ("pop",),
                 ("pop",)]
        for d in range(depth):
            inner = [("push", dict(layer_depth=1)),
                     ("push", dict(layer_depth=1, is_key=1)), ("visit", bgra(W, H, "smooth", seed + 10 + d)), ("pop",)] + inner + \
                    [("push", dict(opacity=0.9 - 0.1 * d, fill_scale=[0.9, 0.9])), ("visit", ycbcr(W, H, "420", "video", seed + 20 + d)), ("pop",),
                     ("pop",)]
        return inner
    S.append(scene("nested_sublayer_keys", W, H,
                   [bg, layer(bgra(W, H, "smooth", 70), is_key=1), nested(1, 71)]))
    S.append(scene("nested_sublayer_keys_deep", W, H,
                   [bg, layer(ycbcr(W, H, "444", "smooth", 75), is_key=1), nested(3, 76),
                    layer(bgra(W, H, "premul", 79), opacity=0.5)]))
    S.append(scene("empty_scene", W, H, []))
    return S


BLEND_NAMES = ["normal", "lighten", "darken", "multiply", "average", "add", "subtract", "difference", "negation",
               "exclusion", "screen", "overlay", "soft_light", "hard_light", "color_dodge", "color_burn",
               "linear_dodge", "linear_burn", "linear_light", "vivid_light", "pin_light", "hard_mix", "reflect",
               "glow", "phoenix", "contrast", "saturation", "color", "luminosity", "mix"]


def blend_mode_scenes():
    S = []
    for name in BLEND_NAMES:
        S.append(scene("blend_" + name, W, H,
                       [layer(ycbcr(W, H, "420", "video", 60)),
                        layer(bgra(W, H, "smooth", 61), blend_mode=name, opacity=0.8, fill_scale=[0.9, 0.9])]))
    return S


def sixteen_bit_scenes():
    S = []
    bg = layer(bgra(W, H, "smooth", 70, bits=16))
    S.append(scene("b16_identity", W, H, [layer(bgra(W, H, "premul", 71, bits=16))], bits=16))
    S.append(scene("b16_opacity", W, H, [layer(bgra(W, H, "premul", 72, bits=16), opacity=0.73)], bits=16))
    S.append(scene("b16_ycbcr422_stack", W, H,
                   [bg, layer(ycbcr(W, H, "422", "video", 73, bits=16), opacity=0.7),
                    layer(ycbcr(W, H, "422", "video", 74, bits=16), opacity=0.6, **LEVELS)], bits=16))
    S.append(scene("b16_chroma_key", W, H, [bg, layer(ycbcr(W, H, "422", "smooth", 75, bits=16), **CHROMA)], bits=16))
    S.append(scene("b16_key_and_fill", W, H,
                   [bg, layer(ycbcr(W, H, "422", "smooth", 76, bits=16), is_key=1),
                    layer(ycbcr(W, H, "422", "video", 77, bits=16), opacity=0.9)], bits=16))
    S.append(scene("b16_scaled", W, H,
                   [bg, layer(ycbcr(48, 32, "422", "video", 78, bits=16), fill_scale=[0.8, 0.7], fill_translation=[0.1, 0.13])], bits=16))
    S.append(scene("b16_screen_blend", W, H, [bg, layer(ycbcr(W, H, "422", "video", 79, bits=16), blend_mode="screen", opacity=0.8)], bits=16))
    S.append(scene("b16_from_8bit_sources", W, H, [bg, layer(bgra(W, H, "premul", 80), opacity=0.9)], bits=16))
    # 1:1 sources a small fraction of a texel off the pixel grid: a 16-bit source is filtered with fp32 weights, so
    # 1e-3 texel is worth up to ~65 LSB16 on noise and must NOT be snapped to the unit-scale class (8-bit sources
    # have 1/256 weight steps and may be)
    for tag, frac in (("5e-4", 0.0005), ("5e-3", 0.005), ("5e-2", 0.05), ("2e-1", 0.2)):
        S.append(scene("b16_near_unit_" + tag, W, H,
                       [bg, layer(ycbcr(W, H, "422", "noise", 81, bits=16), opacity=0.9,
                                  fill_translation=[(4 + frac) / W, (2 + frac) / H]),
                        layer(bgra(W, H, "noise", 82, bits=16), opacity=0.5,
                              fill_translation=[-(8 + frac) / W, (1 - frac) / H])], bits=16))
    return S


def all_golden_scenes():
    return small_scenes() + blend_mode_scenes() + sixteen_bit_scenes()


# ---------------------------------------------------------------------------------------------- BASELINE configs

def config1(width=1920, height=1080):
    """BASELINE.json configs[0]: colour producer (1x1 BGRA) + BGRA image, opacity + normal blend."""
    return scene("config1", width, height,
                 [layer(bgra(1, 1, "solid:0.6", 2)), layer(bgra(width, height, "premul", 1), opacity=0.73)])


CONFIG2_TABLE = [  # (fill_scale, fill_translation, crop_ul, crop_lr, angle, perspective?, blend)
    ((1.0, 1.0), (0.0, 0.0), (0, 0), (1, 1), 0.0, None, "normal"),
    ((0.6, 0.6), (0.05, 0.05), (0.1, 0.0), (1.0, 0.9), 0.0, None, "screen"),
    ((0.5, 0.5), (0.45, 0.1), (0, 0), (1, 1), 0.0, None, "multiply"),
    ((0.4, 0.4), (0.1, 0.5), (0.0, 0.2), (0.8, 1.0), 0.3, None, "overlay"),
    ((1.2, 1.2), (-0.1, -0.1), (0, 0), (1, 1), 0.0, None, "add"),
    ((0.7, 0.7), (0.2, 0.2), (0, 0), (1, 1), 0.0, ((0.1, 0.05), (0.9, 0.1), (1.0, 0.95), (0.05, 0.9)), "difference"),
    ((0.3, 0.3), (0.6, 0.6), (0, 0), (1, 1), -0.5, None, "lighten"),
    ((0.8, 0.5), (0.1, 0.45), (0, 0), (1, 1), 0.0, ((0.0, 0.1), (1.0, 0.0), (0.9, 1.0), (0.1, 0.9)), "normal"),
]


def config2(width=1920, height=1080):
    """configs[1]: 8 ycbcr 4:2:0 8-bit layers, scale/translate/crop/perspective, mixed blend modes."""
    ops = []
    for i, (fs, ft, cu, cl, ang, pers, blend) in enumerate(CONFIG2_TABLE):
        t = dict(fill_scale=list(fs), fill_translation=list(ft), crop_ul=list(cu), crop_lr=list(cl), angle=ang,
                 blend_mode=blend, opacity=1.0 if i == 0 else 0.85)
        if pers:
            t.update(pers_ul=list(pers[0]), pers_ur=list(pers[1]), pers_lr=list(pers[2]), pers_ll=list(pers[3]))
        ops.append(layer(ycbcr(width, height, "420", "video", 100 + i), **t))
    return scene("config2", width, height, ops)


def config3(width=3840, height=2160, bits=16):
    """configs[2] (headline): 16 ycbcr 4:2:2 layers; 4 is_key layers each preceding its fill; 4 chroma-keyed;
    4 with levels; every fill at opacity < 1 so that no layer is occluded. BGRA16 (or BGRA8) output."""
    ops = []
    for i in range(16):
        t = {}
        group = i % 4  # 0: key  1: fill (keyed by the layer before)  2: chroma  3: levels
        if group == 0:
            t = dict(is_key=1)
        elif group == 1:
            t = dict(opacity=0.8)
        elif group == 2:
            t = dict(opacity=0.75, **CHROMA)
        else:
            t = dict(opacity=0.7, **LEVELS)
        kind = "smooth" if group in (0, 2) else "video"
        ops.append(layer(ycbcr(width, height, "422", kind, 200 + i, bits=bits), **t))
    return scene("config3_%dbit" % bits, width, height, ops, bits=bits)


def config4(width=1920, height=1080, delta=0.4):
    """configs[3]: a MIX transition (two is_mix sources, opacities 1-delta and delta) over a background."""
    return scene("config4", width, height,
                 [layer(bgra(width, height, "premul", 300)),
                  [("push", dict(layer_depth=1)),
                   ("push", dict(is_mix=1, opacity=1.0 - delta)), ("visit", bgra(width, height, "premul", 301)), ("pop",),
                   ("push", dict(is_mix=1, opacity=delta)), ("visit", bgra(width, height, "premul", 302)), ("pop",),
                   ("pop",)]])


def algorithmic_bytes(sc):
    """every source plane byte read once + the frame written once."""
    total = sc["width"] * sc["height"] * 4 * (sc["bits"] // 8)
    for op in sc["ops"]:
        if op[0] == "visit":
            for (w, h, c, _k, _s) in op[1]["planes"]:
                total += w * h * c * (op[1]["bits"] // 8)
    return total


# ---------------------------------------------------------------------------------------------- runners

TRANSFORM_RENAME_ORACLE = {  # product (cghip.h) name -> oracle (ref_types.h) name
    "levels_min_input": "lv_min_input", "levels_max_input": "lv_max_input", "levels_gamma": "lv_gamma",
    "levels_min_output": "lv_min_output", "levels_max_output": "lv_max_output",
}


def run_oracle(sc, backend="cpu", planes_cache=None):
    """Render a scene with the restated reference mixer (oracle/) on the 'cpu' or 'gl' backend."""
    from oracle import oracle as O
    m = O.RefMixer(backend)
    try:
        for op in sc["ops"]:
            if op[0] == "push":
                kw = {TRANSFORM_RENAME_ORACLE.get(k, k): v for k, v in op[1].items()}
                m.push(O.default_transform(**kw))
            elif op[0] == "visit":
                spec = op[1]
                planes = _planes(spec, planes_cache)
                m.visit(O.make_frame(spec["fmt"], planes, coords=spec.get("coords")))
            else:
                m.pop()
        return m.render(sc["width"], sc["height"], sc["square_width"], sc["square_height"], sc["bits"],
                        sc["field_mode"])
    finally:
        m.close()


def _planes(spec, cache):
    if cache is None:
        return materialize(spec)
    key = (spec["fmt"], tuple(spec["planes"]), spec["bits"])
    if key not in cache:
        cache[key] = materialize(spec)
    return cache[key]


def feed_product(mixer, sc, planes_cache=None, via_factory=True):
    import server_amd as A
    for op in sc["ops"]:
        if op[0] == "push":
            mixer.push(A.default_transform(**op[1]))
        elif op[0] == "visit":
            spec = op[1]
            mixer.visit(spec["fmt"], _planes(spec, planes_cache), spec.get("coords"), via_factory)
        else:
            mixer.pop()


def run_product(sc, mixer=None, planes_cache=None, via_factory=True, readback=True):
    """Render a scene with the HIP mixer through libcghost.so -> libcghip.so."""
    import server_amd as A
    own = mixer is None
    if own:
        mixer = A.Mixer(channel_id=1, gpu=0)
    try:
        feed_product(mixer, sc, planes_cache, via_factory)
        return mixer.render(sc["width"], sc["height"], sc["square_width"], sc["square_height"], sc["bits"],
                            sc["field_mode"], readback=readback)
    finally:
        if own:
            mixer.close()


def diff_stats(a, b):
    d = np.abs(a.astype(np.int64) - b.astype(np.int64))
    return {"max": int(d.max()) if d.size else 0, "gt0": float((d > 0).mean()) if d.size else 0.0,
            "gt1": float((d > 1).mean()) if d.size else 0.0, "gt2": float((d > 2).mean()) if d.size else 0.0}
