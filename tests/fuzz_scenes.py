"""Seeded random scenes for the fuzz tests: layer stacks that combine what the fixed suite tests one at a time
(pixel formats x geometry x adjustments x keys / mix / blend modes x output sizes)."""
import random

import scenes as S

FORMATS = ["bgra", "rgba", "argb", "abgr", "gray", "luma", "bgr", "rgb", "uyvy", "ycbcr420", "ycbcr422", "ycbcr444",
           "ycbcra444", "ycbcra422"]


def _source(rng, W, H, bits, seed):
    f = rng.choice(FORMATS)
    kind = rng.choice(["smooth", "noise", "video", "premul"])
    # source size: the frame's, or something unrelated (scaled by the quad)
    if rng.random() < 0.5:
        w, h = W, H
    else:
        w, h = rng.choice([16, 24, 40, 48, 64, 90]), rng.choice([8, 16, 30, 32, 50])
    if f in ("bgra", "rgba", "argb", "abgr"):
        return S.frame(f, [(w, h, 4, kind if kind != "video" else "premul", seed)], bits)
    if kind == "premul":
        kind = "smooth"
    if f in ("gray", "luma"):
        return S.frame(f, [(w, h, 1, kind, seed)], bits)
    if f in ("bgr", "rgb"):
        return S.frame(f, [(w, h, 3, kind, seed)], bits)
    if f == "uyvy":
        w += w & 1
        return S.frame("uyvy", [(w, h, 2, kind, seed)], bits)
    sub = f[-3:]
    w += w & 1
    h += h & 1
    return S.ycbcr(w, h, sub, kind, seed, bits=bits, alpha=f.startswith("ycbcra"))


def _transform(rng, allow_blend=True):
    t = {}
    r = rng.random
    if r() < 0.6:
        t["opacity"] = rng.choice([1.0, 0.5, 0.25, round(r(), 3), 0.73])
    if r() < 0.5:
        t["fill_scale"] = [round(rng.uniform(0.2, 1.4), 3), round(rng.uniform(0.2, 1.4), 3)]
        if r() < 0.15:
            t["fill_scale"][0] = -t["fill_scale"][0]  # mirrored
    if r() < 0.5:
        t["fill_translation"] = [round(rng.uniform(-0.3, 0.7), 3), round(rng.uniform(-0.3, 0.7), 3)]
    if r() < 0.25:
        t["angle"] = round(rng.uniform(-1.0, 1.0), 3)
        if r() < 0.5:
            t["anchor"] = [round(r(), 2), round(r(), 2)]
    if r() < 0.25:
        t["crop_ul"] = [round(rng.uniform(0, 0.3), 3), round(rng.uniform(0, 0.3), 3)]
        t["crop_lr"] = [round(rng.uniform(0.6, 1.0), 3), round(rng.uniform(0.6, 1.0), 3)]
    if r() < 0.2:
        t["clip_translation"] = [round(rng.uniform(0, 0.4), 3), round(rng.uniform(0, 0.4), 3)]
        t["clip_scale"] = [round(rng.uniform(0.2, 0.9), 3), round(rng.uniform(0.2, 0.9), 3)]
    if r() < 0.2:
        d = lambda: round(rng.uniform(-0.12, 0.12), 3)
        t.update(pers_ul=[d(), d()], pers_ur=[1 + d(), d()], pers_lr=[1 + d(), 1 + d()], pers_ll=[d(), 1 + d()])
    if r() < 0.2:
        t.update(levels_min_input=round(rng.uniform(0, 0.2), 3), levels_max_input=round(rng.uniform(0.7, 1.0), 3),
                 levels_gamma=rng.choice([1.0, 0.6, 1.2, 2.2]), levels_min_output=rng.choice([0.0, 0.1]),
                 levels_max_output=rng.choice([1.0, 0.9]))
    if r() < 0.2:
        t.update(brightness=round(rng.uniform(0.6, 1.4), 2), saturation=round(rng.uniform(0.0, 1.6), 2),
                 contrast=round(rng.uniform(0.6, 1.4), 2))
    if r() < 0.15:
        t.update(dict(S.CHROMA, chroma_target_hue=rng.choice([120.0, 240.0, 30.0]),
                      chroma_show_mask=1 if r() < 0.2 else 0))
    if r() < 0.1:
        t["invert"] = 1
    if allow_blend and r() < 0.3:
        t["blend_mode"] = rng.choice(S.BLEND_NAMES)
    return t


def fuzz_scene(seed, bits=8):
    rng = random.Random(1000003 * seed + bits)
    W, H = rng.choice([(96, 64), (100, 37), (64, 48), (130, 50), (48, 90), (256, 36)])
    field_mode = rng.choice([0, 0, 0, 1, 2])
    ops = []
    nl = rng.randint(1, 6)
    s = 7000 + 97 * seed
    for i in range(nl):
        kind = rng.random()
        if kind < 0.12:  # key layer followed by its fill
            ops.append(S.layer(_source(rng, W, H, bits, s + 10 * i), is_key=1, **_transform(rng, False)))
            ops.append(S.layer(_source(rng, W, H, bits, s + 10 * i + 1), **_transform(rng)))
        elif kind < 0.22:  # mix pair inside one layer
            d = round(rng.random(), 2)
            t = _transform(rng)
            t["layer_depth"] = 1
            ops.append([("push", t),
                        ("push", dict(is_mix=1, opacity=1.0 - d)), ("visit", _source(rng, W, H, bits, s + 10 * i + 2)), ("pop",),
                        ("push", dict(is_mix=1, opacity=d)), ("visit", _source(rng, W, H, bits, s + 10 * i + 3)), ("pop",),
                        ("pop",)])
        elif kind < 0.3:  # local key + fill in one layer
            ops.append([("push", dict(layer_depth=1, **({"blend_mode": rng.choice(S.BLEND_NAMES)} if rng.random() < 0.5 else {}))),
                        ("push", dict(is_key=1)), ("visit", _source(rng, W, H, bits, s + 10 * i + 4)), ("pop",),
                        ("push", _transform(rng, False)), ("visit", _source(rng, W, H, bits, s + 10 * i + 5)), ("pop",),
                        ("pop",)])
        else:
            ops.append(S.layer(_source(rng, W, H, bits, s + 10 * i + 6), **_transform(rng)))
    square = rng.choice([None, None, (W * 4 // 3, H)])
    return S.scene("fuzz_%d_%dbit" % (seed, bits), W, H, ops, bits=bits, square=square, field_mode=field_mode)


def fuzz_fast_scene(seed, bits=8, full=False):
    """Random stacks the 4-pixels-per-lane kernel accepts: unit-scale layers of the frame's size (optionally cropped
    or clipped on 4-pixel boundaries), keys, every adjustment; full=True adds blend-mode layers and MIX pairs, which
    need the kernel variant with the layer / mix temporaries."""
    rng = random.Random(7000003 * seed + bits)
    W, H = rng.choice([(96, 64), (128, 36), (260, 50), (516, 22), (64, 90)])
    ops = []
    s = 9000 + 131 * seed
    for i in range(rng.randint(1, 7)):
        f = rng.choice(["bgra", "rgba", "argb", "abgr", "ycbcr420", "ycbcr422", "ycbcr444", "ycbcra444", "ycbcra422", "ycbcra420",
                        "solid"] + (["uyvy", "gray", "luma", "bgr", "rgb"] if bits == 8 else []))  # 8-bit only bodies
        kind = rng.choice(["smooth", "noise", "video"])
        if f == "uyvy":  # what SDI capture delivers
            src = S.frame("uyvy", [(W, H, 2, kind, s + i)], bits)
        elif f in ("gray", "luma", "bgr", "rgb"):
            src = S.frame(f, [(W, H, 1 if f in ("gray", "luma") else 3, kind if kind != "video" else "smooth", s + i)], bits)
        elif f == "solid":  # the colour producer: one 1x1 texel stretched over a rectangle on 4-pixel columns
            src = S.frame(rng.choice(["bgra", "rgba", "argb", "abgr"]), [(1, 1, 4, "noise", s + i)], bits)
        elif f in ("bgra", "rgba", "argb", "abgr"):
            src = S.frame(f, [(W, H, 4, rng.choice(["premul", "noise", "smooth"]), s + i)], bits)
        else:
            src = S.ycbcr(W, H, f[-3:], kind, s + i, bits=bits, alpha=f.startswith("ycbcra"))
        t = {}
        r = rng.random
        if r() < 0.1:
            t["invert"] = 1
        step = 4 if t.get("invert") else rng.choice([1, 4])  # box edges inside a lane's 4 pixels need a non-inverting draw
        if f == "solid" and r() < 0.5:
            x0, x1 = sorted(rng.sample(range(0, W + 1, step), 2))
            y0, y1 = sorted(rng.sample(range(0, H + 1), 2))
            if x1 > x0 and y1 > y0:
                t.update(fill_translation=[x0 / W, y0 / H], fill_scale=[(x1 - x0) / W, (y1 - y0) / H])
        if r() < 0.7:
            t["opacity"] = rng.choice([1.0, 0.5, 0.25, round(r(), 3), 0.73])
        if f != "solid" and r() < 0.3:  # crop (any row)
            x0, x1 = sorted(rng.sample(range(0, W + 1, step), 2))
            y0, y1 = sorted(rng.sample(range(0, H + 1), 2))
            if x1 > x0 and y1 > y0:
                t["crop_ul"], t["crop_lr"] = [x0 / W, y0 / H], [x1 / W, y1 / H]
        if r() < 0.2:
            x0, x1 = sorted(rng.sample(range(0, W + 1, step), 2))
            y0, y1 = sorted(rng.sample(range(0, H + 1), 2))
            if x1 > x0 and y1 > y0:
                t["clip_translation"], t["clip_scale"] = [x0 / W, y0 / H], [(x1 - x0) / W, (y1 - y0) / H]
        if r() < 0.25:
            t.update(levels_min_input=round(rng.uniform(0, 0.2), 3), levels_max_input=round(rng.uniform(0.7, 1.0), 3),
                     levels_gamma=rng.choice([1.0, 1.0, 0.6, 1.2, 2.2]), levels_min_output=rng.choice([0.0, 0.1]),
                     levels_max_output=rng.choice([1.0, 0.9]))
        if r() < 0.2:
            t.update(brightness=round(rng.uniform(0.6, 1.4), 2), saturation=round(rng.uniform(0.0, 1.6), 2),
                     contrast=round(rng.uniform(0.6, 1.4), 2))
        if r() < 0.2:
            t.update(dict(S.CHROMA, chroma_target_hue=rng.choice([120.0, 240.0, 30.0]), chroma_show_mask=1 if r() < 0.2 else 0))
        if r() < 0.25:
            t["is_key"] = 1
        elif full and r() < 0.35:  # the layer goes through a layer temporary and is composited with a blend mode
            t["blend_mode"] = rng.choice(S.BLEND_NAMES)
        if full and not t.get("is_key") and r() < 0.25:  # MIX transition: two is_mix items in one layer
            d = round(rng.random(), 2)
            other = S.frame("bgra", [(W, H, 4, "premul", s + 50 + i)], bits)
            lt = {k: v for k, v in t.items() if k in ("blend_mode", "opacity")}
            lt["layer_depth"] = 1
            it = {k: v for k, v in t.items() if k not in ("blend_mode", "opacity")}
            ops.append([("push", lt),
                        ("push", dict(it, is_mix=1, opacity=1.0 - d)), ("visit", src), ("pop",),
                        ("push", dict(is_mix=1, opacity=d)), ("visit", other), ("pop",),
                        ("pop",)])
            continue
        ops.append(S.layer(src, **t))
    return S.scene("fuzzfast_%d_%dbit" % (seed, bits), W, H, ops, bits=bits, field_mode=rng.choice([0, 0, 1, 2]))


def fuzz_mix_scene(seed, bits=8, blend=False, boxes=False):
    """Unit-scale stacks without blend-mode layers in which about every second fill layer is a MIX transition (two is_mix
    items in one layer, image_mixer.cpp:187-225): what the planner fuses into one step of the lean kernel
    (cghip_plan.cpp: fuse_mix_pairs). The stack is fuzz_fast_scene's (keys, every adjustment, crops, clips, field modes);
    the second item of a pair is another source of the frame's size, sometimes with adjustments of its own.
    blend: about every second of those layers has a blend mode, i.e. the pair is composited into the layer's texture and
    that onto the frame (image_mixer.cpp:118-129): the step in the kernel with one temporary (cghip_composite_fast_one_p).
    boxes: the stack is fuzz_box_scene's - picture-in-picture boxes between the unit-scale layers, and only the latter
    become MIX pairs: frames whose other draws need the sampled bodies (cghip_composite_fast_xp)."""
    base = fuzz_box_scene(seed, bits, False) if boxes else fuzz_fast_scene(seed, bits, False)
    rng = random.Random(6000011 * seed + bits + 3)
    modes = random.Random(7000003 * seed + bits + 5)  # a stream of its own: the scenes without blend modes stay as they are
    W, H = base["width"], base["height"]
    ops = []
    flat = base["ops"]  # scene() flattened the layers: without blend-mode layers and MIX pairs they are push / visit / pop triples
    assert len(flat) % 3 == 0 and all(flat[k][0] == ("push", "visit", "pop")[k % 3] for k in range(len(flat)))
    for i, lay in enumerate([flat[k:k + 3] for k in range(0, len(flat), 3)]):
        t = lay[0][1]
        if t.get("is_key") or t.get("invert") or (boxes and ("fill_scale" in t or "fill_translation" in t)) or rng.random() < (0.2 if boxes else 0.5):
            ops.append(lay)
            continue
        d = rng.choice([0.5, 0.25, round(rng.random(), 2)])
        f = rng.choice(["bgra", "argb", "ycbcr422", "ycbcr420", "ycbcra444"] + (["uyvy", "gray"] if bits == 8 else []))
        sd = 31000 + 17 * seed + i
        if f == "uyvy":
            other = S.frame("uyvy", [(W, H, 2, "video", sd)], bits)
        elif f == "gray":
            other = S.frame("gray", [(W, H, 1, "smooth", sd)], bits)
        elif f in ("bgra", "argb"):
            other = S.frame(f, [(W, H, 4, rng.choice(["premul", "noise"]), sd)], bits)
        else:
            other = S.ycbcr(W, H, f[-3:], "video", sd, bits=bits, alpha=f.startswith("ycbcra"))
        lt = {k: v for k, v in t.items() if k in ("opacity", "layer_depth")}
        if rng.random() < 0.5:
            lt.pop("opacity", None)
        it = {k: v for k, v in t.items() if k not in ("opacity", "layer_depth")}
        ot = {}
        if rng.random() < 0.8:  # the same box as the first item (crop, clip, the colour producer's rectangle): what fuses
            ot.update({k: v for k, v in t.items() if k in ("crop_ul", "crop_lr", "clip_translation", "clip_scale",
                                                            "fill_translation", "fill_scale")})
        if rng.random() < 0.3:
            ot.update(levels_min_input=0.05, levels_max_input=0.9, levels_gamma=rng.choice([1.0, 1.2]), levels_min_output=0.0,
                      levels_max_output=1.0)
        if rng.random() < 0.15:
            ot.update(brightness=1.1, saturation=0.8, contrast=1.2)
        if blend and modes.random() < 0.5:
            lt["blend_mode"] = modes.choice(["screen", "multiply", "overlay", "add", "difference", "lighten", "hard_light", "color"])
        ops.append([("push", lt),
                    ("push", dict(it, is_mix=1, opacity=round(1.0 - d, 2))), ("visit", lay[1][1]), ("pop",),
                    ("push", dict(ot, is_mix=1, opacity=d)), ("visit", other), ("pop",),
                    ("pop",)])
    return S.scene("fuzzmix%s%s_%d_%dbit" % ("b" if blend else "", "x" if boxes else "", seed, bits), W, H, ops, bits=bits, field_mode=base["field_mode"])


def fuzz_box_scene(seed, bits=8, full=False, mirror=False):
    """Random stacks with picture-in-picture boxes: axis-aligned layers at any scale and offset (sources of their own
    size, odd widths and tiny planes included) and 1:1 layers moved by whole pixels off the 4-pixel grid, between
    ordinary unit-scale layers. All of it is the 4-pixels-per-lane kernel's class since it samples boxes itself
    (cghip_kernels.hip: fetch_rect8 / fetch_rect16); full=True puts some of the boxes into blend-mode layers.
    mirror=True flips most boxes horizontally and / or vertically (a negative fill_scale with the translation moved to
    the box's other side: the usual flip in the reference; KF_AXIS admits it, s or t then DECREASE along the row or
    column). The flips come from a generator of their own, so the unmirrored scenes of a seed are what they always were."""
    rng = random.Random(9000011 * seed + bits)
    flips = random.Random(7919 * seed + 31 * bits + 5)
    W, H = rng.choice([(96, 64), (128, 36), (260, 50), (64, 90), (200, 40)])
    ops = []
    s = 12000 + 137 * seed
    for i in range(rng.randint(2, 6)):
        f = rng.choice(["bgra", "rgba", "argb", "abgr", "ycbcr420", "ycbcr422", "ycbcr444", "ycbcra444", "ycbcra422", "ycbcra420"] +
                       (["uyvy", "uyvy"] if bits == 8 else []))  # a live SDI input in a box
        mode = rng.choice(["box", "box", "box", "shift", "plain"])
        if mode == "box":  # a source of its own size; 4:2:x needs even sizes, chroma planes stay >= 2 texels wide
            sw, sh = rng.choice([(W, H), (W // 2, H // 2), (2 * W, 2 * H), (40, 30), (6, 4), (4, 2), (130, 70)])
        else:
            sw, sh = W, H
        kind = rng.choice(["smooth", "noise", "video"])
        if f == "uyvy":
            src = S.frame("uyvy", [(sw, sh, 2, kind, s + i)], bits)
        elif f in ("bgra", "rgba", "argb", "abgr"):
            if mode == "box" and rng.random() < 0.3:
                sw, sh = rng.choice([(2, 2), (3, 5), (37, 21)])  # packed planes may be odd and tiny
            src = S.frame(f, [(sw, sh, 4, rng.choice(["premul", "noise", "smooth"]), s + i)], bits)
        else:
            src = S.ycbcr(sw, sh, f[-3:], kind, s + i, bits=bits, alpha=f.startswith("ycbcra"))
        t = {}
        r = rng.random
        if mode == "box":
            sx, sy = round(rng.uniform(0.15, 1.4), 3), round(rng.uniform(0.15, 1.4), 3)
            t.update(fill_scale=[sx, sy], fill_translation=[round(rng.uniform(-0.3, 0.9), 3), round(rng.uniform(-0.3, 0.9), 3)])
            if r() < 0.3:
                t["crop_ul"], t["crop_lr"] = [round(rng.uniform(0, 0.3), 2), round(rng.uniform(0, 0.3), 2)], \
                                             [round(rng.uniform(0.6, 1.0), 2), round(rng.uniform(0.6, 1.0), 2)]
            if mirror:
                fx, fy = flips.random() < 0.6, flips.random() < 0.4
                sc_, tr_ = t["fill_scale"], t["fill_translation"]
                if fx:
                    sc_[0], tr_[0] = -sc_[0], round(tr_[0] + sc_[0], 3)
                if fy:
                    sc_[1], tr_[1] = -sc_[1], round(tr_[1] + sc_[1], 3)
        elif mode == "shift":  # whole pixels, mostly not multiples of 4
            t["fill_translation"] = [rng.randint(-9, 9) / W, rng.randint(-5, 5) / H]
        if r() < 0.7:
            t["opacity"] = rng.choice([1.0, 0.5, 0.25, round(r(), 3), 0.73])
        if r() < 0.2:
            x0, x1 = sorted(rng.sample(range(0, W + 1), 2))
            y0, y1 = sorted(rng.sample(range(0, H + 1), 2))
            if x1 > x0 and y1 > y0:
                t["clip_translation"], t["clip_scale"] = [x0 / W, y0 / H], [(x1 - x0) / W, (y1 - y0) / H]
        if r() < 0.2:
            t.update(levels_min_input=round(rng.uniform(0, 0.2), 3), levels_max_input=round(rng.uniform(0.7, 1.0), 3),
                     levels_gamma=rng.choice([1.0, 1.0, 0.6, 1.2]), levels_min_output=rng.choice([0.0, 0.1]),
                     levels_max_output=rng.choice([1.0, 0.9]))
        if r() < 0.15:
            t.update(brightness=round(rng.uniform(0.6, 1.4), 2), saturation=round(rng.uniform(0.0, 1.6), 2),
                     contrast=round(rng.uniform(0.6, 1.4), 2))
        if r() < 0.15:
            t.update(dict(S.CHROMA, chroma_target_hue=rng.choice([120.0, 240.0, 30.0])))
        if r() < 0.2:
            t["is_key"] = 1
        elif full and r() < 0.4:
            t["blend_mode"] = rng.choice(S.BLEND_NAMES)
        ops.append(S.layer(src, **t))
    return S.scene("fuzzbox%s_%d_%dbit" % ("m" if mirror else "", seed, bits), W, H, ops, bits=bits,
                   field_mode=rng.choice([0, 0, 0, 1, 2]))


def fuzz_nested_scene(seed, bits=8):
    """Random stacks with NESTED sublayers that carry keys of their own: every nesting level keeps a pending layer key
    (image_mixer.cpp:104-114,153), so three to five key textures are alive at once where the kernels hold two - the
    planner parks the rest in HBM across cuts (KOP_SAVE_KEY / KOP_LOAD_KEY). Sources of both kernel classes."""
    rng = random.Random(5000017 * seed + bits)
    W, H = rng.choice([(96, 64), (128, 36), (64, 48), (130, 50)])
    s = 20000 + 173 * seed
    n = [0]

    def src(unit):
        n[0] += 1
        if unit:
            f = rng.choice(["bgra", "ycbcr422", "ycbcr420", "ycbcr444"])
            if f == "bgra":
                return S.frame("bgra", [(W, H, 4, rng.choice(["premul", "smooth"]), s + n[0])], bits)
            return S.ycbcr(W, H, f[-3:], rng.choice(["smooth", "video"]), s + n[0], bits=bits)
        return _source(rng, W, H, bits, s + n[0])

    def item(**t):
        unit = rng.random() < 0.6
        if not unit and rng.random() < 0.7:
            t.update(_transform(rng, False))
        elif rng.random() < 0.5:
            t["opacity"] = rng.choice([1.0, 0.5, 0.73, round(rng.random(), 3)])
        return [("push", t), ("visit", src(unit)), ("pop",)]

    def layer_with_sublayers(depth):
        ops = [("push", dict(layer_depth=1, **({"blend_mode": rng.choice(S.BLEND_NAMES)} if rng.random() < 0.2 else {})))]
        if depth > 0:
            for _ in range(rng.randint(1, 2)):
                if rng.random() < 0.7:  # a key sublayer in front of the next one
                    ops += [("push", dict(layer_depth=1, is_key=1)), ("visit", src(True)), ("pop",)]
                ops += layer_with_sublayers(depth - 1)
        if rng.random() < 0.6:  # local key + fill
            ops += item(is_key=1)
        ops += item()
        if rng.random() < 0.3:
            ops += item()
        ops.append(("pop",))
        return ops

    ops = [S.layer(S.bgra(W, H, "premul", s))]
    for _ in range(rng.randint(1, 3)):
        if rng.random() < 0.7:
            ops.append(S.layer(src(True), is_key=1))  # leaves a pending key for the next top-level layer
        ops.append(layer_with_sublayers(rng.randint(1, 3)))
    return S.scene("fuzznested_%d_%dbit" % (seed, bits), W, H, ops, bits=bits, field_mode=rng.choice([0, 0, 0, 1, 2]))
