"""CPU suite: the product's HOST logic (C++ push/visit/pop/flatten in server_amd/host/, planner in
server_amd/csrc/cghip_plan.cpp) against the oracle's independent restatement (oracle/ref_scene.c), with no GPU.

Both sides restate image_mixer.cpp:104-299 and image_kernel.cpp:134-375; the oracle's version is pinned through the
llvmpipe goldens (test_oracle_golden.py), so agreement here pins the product's draw order, temporaries, uniforms
and quad math.
"""
import ctypes as C

import numpy as np
import pytest

import kops as K
import scenes as S
import server_amd as A
from oracle import oracle as O

SCENES = [s for s in S.small_scenes() + S.blend_mode_scenes() if s["ops"]]


def product_plan(sc):
    m = A.Mixer(channel_id=1, gpu=-2)  # dry run: no device needed
    try:
        S.feed_product(m, sc, via_factory=False)
        draws, surfaces = m.flatten(sc["width"], sc["height"], sc["square_width"], sc["square_height"], sc["bits"])
        fmt = A.Format(sc["width"], sc["height"], sc["square_width"], sc["square_height"], sc["bits"], 0)
        ops, full = K.plan(fmt, surfaces, draws)
        return draws, surfaces, ops, full
    finally:
        m.close()


def oracle_log(sc):
    m = O.RefMixer("cpu")
    try:
        for op in sc["ops"]:
            if op[0] == "push":
                kw = {S.TRANSFORM_RENAME_ORACLE.get(k, k): v for k, v in op[1].items()}
                m.push(O.default_transform(**kw))
            elif op[0] == "visit":
                m.visit(O.make_frame(op[1]["fmt"], S.materialize(op[1]), coords=op[1].get("coords")))
            else:
                m.pop()
        m.render(sc["width"], sc["height"], sc["square_width"], sc["square_height"], sc["bits"], log=True)
        return m.draw_log
    finally:
        m.close()


def executed_mode(mode):
    """The planner keeps the reference's mode number: soft_light (12) and mix (29) execute BlendNormal, but inside
    `if (blend_mode != 0)` (shader.frag:403,421,427), i.e. with the un-premultiply / re-premultiply round trip that
    mode 0 skips. Only a negative mode (the oracle's 'no blend()' marker) maps to 0."""
    return 0 if mode < 0 else mode


@pytest.mark.parametrize("sc", SCENES, ids=[s["name"] for s in SCENES])
def test_planner_agrees_with_oracle_draw_list(sc):
    _draws, _surf, ops, _full = product_plan(sc)
    log = oracle_log(sc)
    kd = [o for o in ops if o.kind != K.KOP_CLEAR]
    # the oracle logs every draw that reaches GL; the planner additionally drops draws whose pixel box is empty
    W, H = sc["width"], sc["height"]
    assert len(kd) <= len(log)
    it = iter(log)
    for o in kd:
        for d in it:
            if same_draw(o, d):
                break
        else:
            pytest.fail("planner op without a matching oracle draw, in order")
        check_uniforms(o, d)
        check_geometry(o, d, W, H)


def same_draw(o, d):
    return (o.pixel_format == d.pixel_format and executed_mode(d.blend_mode) == o.blend_mode and
            bool(o.flags & K.KF_ADDITIVE) == (d.keyer == 1) and abs(o.opacity - d.opacity) < 1e-7 and
            (o.local_key >= 0) == bool(d.has_local_key) and (o.layer_key >= 0) == bool(d.has_layer_key))


def check_uniforms(o, d):
    assert bool(o.flags & K.KF_INVERT) == bool(d.invert)
    assert bool(o.flags & K.KF_LEVELS) == bool(d.levels)
    assert bool(o.flags & K.KF_CSB) == bool(d.csb)
    assert bool(o.flags & K.KF_CHROMA) == bool(d.chroma)
    if o.kind == K.KOP_DRAW_PLANES:
        assert bool(o.flags & K.KF_IS_HD) == bool(d.is_hd)
    if d.levels:
        assert o.lv_min_input == d.min_input and o.lv_min_output == d.min_output and o.lv_max_output == d.max_output
        assert o.lv_inv_range == pytest.approx(1.0 / (np.float32(d.max_input) - np.float32(d.min_input)), rel=1e-6)
        assert o.lv_inv_gamma == pytest.approx(1.0 / d.gamma, rel=1e-6)
    if d.csb:
        assert (o.brt, o.sat, o.con) == (d.brt, d.sat, d.con)
    if d.chroma:
        assert bool(o.flags & K.KF_CHROMA_MASK) == bool(d.chroma_show_mask)
        for a, b in (("ch_target_hue", "chroma_target_hue"), ("ch_hue_width", "chroma_hue_width"),
                     ("ch_min_saturation", "chroma_min_saturation"), ("ch_min_brightness", "chroma_min_brightness"),
                     ("ch_softness", "chroma_softness"), ("ch_spill", "chroma_spill_suppress"),
                     ("ch_spill_saturation", "chroma_spill_suppress_saturation")):
            assert getattr(o, a) == getattr(d, b), a


def check_geometry(o, d, W, H):
    # varyings of both triangles must reproduce the oracle's transformed quad at its own vertices
    tri_idx = ((0, 1, 2), (0, 2, 3))
    ntri_used = 0
    for t, idx in enumerate(tri_idx):
        xs = [np.float32(d.coords[i].vertex_x) for i in idx]
        ys = [np.float32(d.coords[i].vertex_y) for i in idx]
        area = (xs[1] - xs[0]) * (ys[2] - ys[0]) - (ys[1] - ys[0]) * (xs[2] - xs[0])
        if abs(area) < 1e-9:
            continue
        tri = o.tri[ntri_used]
        ntri_used += 1
        for i in idx:
            c = d.coords[i]
            px = float(np.float32(c.vertex_x)) * W - 0.5
            py = float(np.float32(c.vertex_y)) * H - 0.5
            got = [tri.a[j][0] + tri.a[j][1] * px + tri.a[j][2] * py for j in range(3)]
            # the planner folds a uniform q (2.0 for every parallelogram) into s,t: compare after the divide
            assert got[0] / got[2] == pytest.approx(c.texture_x / c.texture_q, abs=2e-5), (t, i, "s")
            assert got[1] / got[2] == pytest.approx(c.texture_y / c.texture_q, abs=2e-5), (t, i, "t")
            if o.flags & K.KF_PROJECTIVE:
                assert got[2] == pytest.approx(float(np.float32(c.texture_q)), abs=2e-5), (t, i, "q")
    # pixel box: inside the frame and inside the oracle's scissor
    assert 0 <= o.x0 < o.x1 <= W and 0 <= o.y0 < o.y1 <= H
    if d.scissor:
        assert o.x0 >= d.sc_x and o.y0 >= d.sc_y and o.x1 <= d.sc_x + d.sc_w and o.y1 <= d.sc_y + d.sc_h


def test_transform_compose_matches_oracle():
    """image_transform::operator*= (frame_transform.cpp:53-108): C++ host vs oracle C, on seeded random transforms,
    observed through the flattened draw's transform."""
    rng = np.random.default_rng(7)
    for _ in range(25):
        def rnd():
            return dict(opacity=rng.uniform(.2, 1), contrast=rng.uniform(.5, 1.5), brightness=rng.uniform(.5, 1.5),
                        saturation=rng.uniform(.5, 1.5), anchor=list(rng.uniform(-.2, .2, 2)),
                        fill_translation=list(rng.uniform(-.3, .3, 2)), fill_scale=list(rng.uniform(.5, 1.5, 2)),
                        clip_translation=list(rng.uniform(0, .2, 2)), clip_scale=list(rng.uniform(.5, 1, 2)),
                        angle=rng.uniform(-1, 1), crop_ul=list(rng.uniform(0, .2, 2)), crop_lr=list(rng.uniform(.7, 1, 2)),
                        pers_ul=list(rng.uniform(0, .1, 2)), pers_ur=[rng.uniform(.9, 1), rng.uniform(0, .1)],
                        pers_lr=list(rng.uniform(.9, 1, 2)), pers_ll=[rng.uniform(0, .1), rng.uniform(.9, 1)],
                        levels_min_input=rng.uniform(0, .2), levels_max_input=rng.uniform(.8, 1),
                        levels_gamma=rng.uniform(.5, 2), levels_min_output=rng.uniform(0, .2),
                        levels_max_output=rng.uniform(.8, 1), chroma_enable=int(rng.integers(0, 2)),
                        chroma_target_hue=rng.uniform(0, 360), chroma_hue_width=rng.uniform(0, .3),
                        chroma_min_saturation=rng.uniform(0, .5), chroma_min_brightness=rng.uniform(0, .5),
                        chroma_softness=rng.uniform(0, .5), chroma_spill_suppress=rng.uniform(0, 90),
                        chroma_spill_suppress_saturation=rng.uniform(0, 1), invert=int(rng.integers(0, 2)),
                        blend_mode=int(rng.integers(0, 30)))
        a, b, c = rnd(), rnd(), rnd()
        a["layer_depth"] = 1
        m = A.Mixer(1, -2)
        for t in (a, b, c):
            m.push(A.default_transform(**t))
        m.visit("bgra", [np.zeros((4, 4, 4), np.uint8)], via_factory=False)
        draws, _ = m.flatten(64, 64)
        m.close()
        got = draws[0].transform  # the item draw (a blend-mode layer appends its composite draw)
        want = O.default_transform()
        for t in (a, b, c):
            kw = {S.TRANSFORM_RENAME_ORACLE.get(k, k): v for k, v in t.items()}
            O.lib().ref_transform_compose(C.byref(want), C.byref(O.default_transform(**kw)))
        for name, _ in A.Transform._fields_:
            if name == "reserved_":
                continue
            g = getattr(got, name)
            w = getattr(want, S.TRANSFORM_RENAME_ORACLE.get(name, name))
            if isinstance(g, C.Array):
                assert list(g) == pytest.approx(list(w), rel=1e-12, abs=1e-15), name
            else:
                assert g == pytest.approx(w, rel=1e-12, abs=1e-15), name


def test_surface_temporaries_follow_the_reference_renderer():
    # key layer then fill: one 1-channel temporary, consumed as LAYER key by the next layer (image_mixer.cpp:153)
    sc = [s for s in S.small_scenes() if s["name"] == "key_then_fill_same_layer_depth"][0]
    draws, surfaces, ops, full = product_plan(sc)
    assert surfaces == [4, 1] and not full
    assert draws[1].target == 1 and draws[2].layer_key == 1 and draws[2].local_key == -1
    # local key inside one layer is consumed by the first fill as LOCAL key (image_mixer.cpp:200)
    sc = [s for s in S.small_scenes() if s["name"] == "local_key_and_fill_in_one_layer"][0]
    draws, surfaces, ops, full = product_plan(sc)
    assert draws[2].local_key == 1 and draws[2].layer_key == -1
    # MIX: two additive draws into a 4-channel temporary, flushed onto the target with a surface draw (:150,:184-195)
    sc = [s for s in S.small_scenes() if s["name"] == "mix_transition"][0]
    draws, surfaces, ops, full = product_plan(sc)
    assert surfaces == [4, 4] and not full  # the planner fuses the pair into one step of the lean kernel (fuse_mix_pairs)
    assert [o.blend_body for o in ops if o.kind == K.KOP_DRAW_PLANES][-2:] == [7, 2] and not any(o.kind == K.KOP_CLEAR for o in ops)
    A.lib().cghip_debug_no_mix_fusion(1)
    try:
        assert product_plan(sc)[3]  # as the reference draws it: a frame with a temporary
    finally:
        A.lib().cghip_debug_no_mix_fusion(0)
    assert [d.keyer for d in draws[1:3]] == [1, 1] and draws[3].source_surface == 1 and draws[3].target == 0
    # non-normal blend layer: items -> layer temporary, then composited with the layer's mode (:127-139)
    sc = [s for s in S.blend_mode_scenes() if s["name"] == "blend_screen"][0]
    draws, surfaces, ops, full = product_plan(sc)
    assert surfaces == [4, 4] and draws[1].target == 1 and draws[1].blend_mode == 0
    assert draws[2].source_surface == 1 and draws[2].target == 0 and draws[2].blend_mode == A.BLEND["screen"]


def test_fast_path_classification():
    full_frame = S.scene("f", 96, 64, [S.layer(S.ycbcr(96, 64, "422", "video", 1), opacity=0.5)])
    _d, _s, ops, _f = product_plan(full_frame)
    o = [x for x in ops if x.kind == K.KOP_DRAW_PLANES][0]
    assert o.flags & K.KF_RECT and o.flags & K.KF_UNIT_SCALE and (o.unit_dx, o.unit_dy) == (0, 0)
    assert (o.x0, o.y0, o.x1, o.y1) == (0, 0, 96, 64)
    scaled = S.scene("s", 96, 64, [S.layer(S.bgra(96, 64), fill_scale=[0.5, 0.5])])
    o = [x for x in product_plan(scaled)[2] if x.kind == K.KOP_DRAW_PLANES][0]
    assert o.flags & K.KF_RECT and not (o.flags & K.KF_UNIT_SCALE) and (o.x1, o.y1) == (48, 32)
    rotated = S.scene("r", 96, 64, [S.layer(S.bgra(96, 64), angle=0.2)])
    o = [x for x in product_plan(rotated)[2] if x.kind == K.KOP_DRAW_PLANES][0]
    assert not (o.flags & K.KF_RECT)
    cropped = S.scene("c", 96, 64, [S.layer(S.bgra(96, 64), crop_ul=[0.25, 0.25], crop_lr=[0.75, 0.75])])
    o = [x for x in product_plan(cropped)[2] if x.kind == K.KOP_DRAW_PLANES][0]
    assert o.flags & K.KF_UNIT_SCALE and (o.x0, o.y0, o.x1, o.y1) == (24, 16, 72, 48)


def test_planner_rejects_bad_input():
    fmt = A.Format(64, 64, 64, 64, 8, 0)
    with pytest.raises(A.CGHIPError):
        K.plan(A.Format(0, 64, 64, 64, 8, 0), [4], [])
    with pytest.raises(A.CGHIPError):
        K.plan(A.Format(64, 64, 64, 64, 12, 0), [4], [])
    with pytest.raises(A.CGHIPError):
        K.plan(fmt, [1], [])  # surface 0 must be RGBA
    d = A.Draw()
    d.target = 5
    with pytest.raises(A.CGHIPError):
        K.plan(fmt, [4], [d])
    ops, full = K.plan(fmt, [4], [])
    assert ops == [] and not full  # empty scene plans to zero ops


def test_visit_without_open_layer_is_an_error():
    m = A.Mixer(1, -2)
    m.push(A.default_transform())  # layer_depth 0: the reference would dereference an empty layer stack
    with pytest.raises(A.CGHIPError):
        m.visit("bgra", [np.zeros((2, 2, 4), np.uint8)], via_factory=False)
    m.close()


def test_dry_run_mixer_cannot_render():
    m = A.Mixer(1, -2)
    m.push(A.default_transform(layer_depth=1))
    m.visit("bgra", [np.zeros((2, 2, 4), np.uint8)], via_factory=False)
    m.pop()
    with pytest.raises(A.CGHIPError):
        m.render(8, 8)
    m.close()


def test_many_key_layers_do_not_exhaust_surface_slots():
    """Every is_key layer creates a fresh key texture in the reference (image_mixer.cpp:174-183); a rundown with 40
    key + fill pairs names 41 surface slots, which the planner still binds to its two key registers by liveness."""
    W, H = 64, 32
    ops = []
    for i in range(40):
        ops.append(S.layer(S.bgra(W, H, "smooth", 900 + i), is_key=1))
        ops.append(S.layer(S.bgra(W, H, "premul", 950 + i), opacity=0.9))
    sc = S.scene("many_keys", W, H, ops)
    draws, surfaces, kops, full = product_plan(sc)
    assert len(surfaces) == 41 and not full
    assert sum(1 for o in kops if o.kind != K.KOP_CLEAR) == 80
    assert all(o.dst in (0, 3, 4) for o in kops)


@pytest.mark.parametrize("full", [False, True])
def test_fast_class_fuzz_scenes_are_planned_for_the_fast_kernel(full):
    """The GPU fuzz of cghip_composite_fast (tests/test_fuzz.py) only means something if the planner really sends
    those scenes there: every one of them must come out with the 'fast' bit, and the ones with blend-mode layers or
    MIX pairs with 'full' as well."""
    from fuzz_scenes import fuzz_fast_scene
    n_full = 0
    for seed in range(100):
        sc = fuzz_fast_scene(seed, 8, full)
        _d, _s, ops, is_full = product_plan(sc)
        assert K.last_plan_fast or not ops, (seed, [(o.kind, o.fast_class) for o in ops])
        n_full += int(is_full)
    assert (n_full > 20) if full else (n_full == 0)  # 26 since MIX pairs fuse next to fused layers (43 with cghip_debug_no_mix_fusion)


@pytest.mark.parametrize("bits", [8, 16])
def test_box_fuzz_scenes_are_planned_for_the_fast_kernels_sampled_bodies(bits):
    """Picture-in-picture boxes and 1:1 layers off the 4-pixel grid are the fast kernel's own (FC_SAMPLED): the GPU
    test over these scenes (tests/test_fuzz.py) only means something if the planner sends them there."""
    from fuzz_scenes import fuzz_box_scene
    n_sampled = n_axis = n_unit = 0
    for seed in range(160):  # 100 as they are + 60 with mirrored boxes (negative fill_scale: KF_AXIS admits them)
        sc = fuzz_box_scene(seed, bits, seed % 2 == 1, mirror=seed >= 100)
        _d, _s, ops, _full = product_plan(sc)
        assert K.last_plan_fast or not ops, (seed, [(o.kind, o.fast_class, hex(o.flags)) for o in ops])
        for o in ops:
            if o.kind == K.KOP_DRAW_PLANES and o.fast_class == K.FC_SAMPLED:
                n_sampled += 1
                assert not o.flags & K.KF_INVERT and all(o.planes[n].w >= 2 for n in range(o.nplanes))
                if o.pixel_format == 10:  # uyvy: FB_RECT_UYVY8 (+ 1 at unit scale)
                    assert o.fetch_body - 20 == int(bool(o.flags & K.KF_UNIT_SCALE))
                else:
                    base = 16 if o.planes[0].bits == 16 else 12
                    assert base <= o.fetch_body < base + 4
                    assert bool(o.flags & K.KF_UNIT_SCALE) == bool((o.fetch_body - base) & 2)
                n_unit += bool(o.flags & K.KF_UNIT_SCALE)
                n_axis += bool(o.flags & K.KF_AXIS)
    assert n_sampled > 150 and n_axis > 100 and n_unit > 20, (n_sampled, n_axis, n_unit)


# ---------------------------------------------------------------- frames cut into several launches (segments)

def check_segments(ops, segs, W, H):
    """Replays the hand-over of the register surfaces between the launches of one frame with version numbers: every
    op must see, in its launch's registers, the value a single launch would hold; the frame buffer must end up with
    the last RGBA0."""
    assert segs and segs[0]["first"] == 0 and sum(s["count"] for s in segs) == len(ops)
    ZERO = 0
    true = [ZERO] * 5   # value a single launch would hold
    mem = [None] * 5    # value in HBM (None: never written); the key registers' values live in numbered key buffers
    keybuf = {}         # key buffer id -> value
    parked = {}         # surface slot -> value of a key texture that left its register (KOP_SAVE_KEY)
    version = 0
    for si, sg in enumerate(segs):
        assert sg["first"] == (segs[si - 1]["first"] + segs[si - 1]["count"] if si else 0)
        assert si == 0 or sg["count"] > 0
        boxed = (sg["gx0"], sg["gy0"], sg["gx1"], sg["gy1"]) != (0, 0, W, H)
        if boxed:
            assert si > 0 and not sg["fast"] and sg["store_mask"] & ~1 == 0
            assert sg["gx0"] % 64 == 0 and sg["gy0"] % 4 == 0
        if si == 0:
            assert sg["store_mask"] & 1 and sg["load_mask"] == 0
        regs = [ZERO] * 5
        for s in range(5):
            if sg["load_mask"] & (1 << s):
                if s >= 3:
                    b = sg["key_ld%d" % (s - 3)]
                    assert b in keybuf, (si, s, b, "loads a key buffer nobody stored")
                    regs[s] = keybuf[b]
                else:
                    assert mem[s] is not None, (si, s, "loads a surface nobody stored")
                    regs[s] = mem[s]
        seg_ops = ops[sg["first"]:sg["first"] + sg["count"]]
        for oi, o in enumerate(seg_ops):
            if o.kind == K.KOP_CLEAR:
                regs[o.dst] = true[o.dst] = ZERO
                continue
            if o.kind == K.KOP_SAVE_KEY:  # (only with cghip_debug_keep_pseudo_ops) the key texture leaves its register
                parked[o.src_surface] = true[o.dst]
                continue
            if o.kind == K.KOP_LOAD_KEY:  # ... and comes back: the launch must have loaded exactly that value
                true[o.dst] = parked.pop(o.src_surface)
                assert regs[o.dst] == true[o.dst], (si, o.dst, "a parked key texture came back with another value")
                continue
            if not sg["full"]:
                # without temporaries the only composite of a texture is the third op of a fused MIX pair (BB_MIX_HEAD
                # = 7 two ops earlier, in the same launch): the step keeps the mix texture in its own registers
                mix_tail = (o.kind == K.KOP_DRAW_SURFACE and oi >= 2 and seg_ops[oi - 2].kind == K.KOP_DRAW_PLANES and
                            seg_ops[oi - 2].blend_body == 7 and seg_ops[oi - 1].kind == K.KOP_DRAW_PLANES and
                            o.src_surface == 0 and sg["fast"])
                assert o.dst not in (1, 2) and (o.kind != K.KOP_DRAW_SURFACE or mix_tail)
            if o.kind == K.KOP_DRAW_PLANES and o.blend_body == 7 and sg["fast"]:
                assert oi + 2 < len(seg_ops) and seg_ops[oi + 2].kind == K.KOP_DRAW_SURFACE, "a MIX pair's step cut apart"
            if sg["fast"]:
                assert o.fast_class != 0
            if boxed:
                assert sg["gx0"] <= o.x0 and o.x1 <= sg["gx1"] and sg["gy0"] <= o.y0 and o.y1 <= sg["gy1"]
            reads = [o.dst, o.local_key, o.layer_key] + ([o.src_surface] if o.kind == K.KOP_DRAW_SURFACE else [])
            if o.fused:  # a blend-mode layer with one item as ONE op: the composite's keys are read too
                reads += [o.b_local_key, o.b_layer_key]
                if o.kind == K.KOP_DRAW_SURFACE:  # the composite of a MIX pair that is all its blend-mode layer holds
                    assert sg["fast"] and not sg["full"] and oi >= 2 and seg_ops[oi - 2].blend_body == 7 and o.dst == 0
                else:
                    assert o.kind == K.KOP_DRAW_PLANES and (not sg["fast"] or o.blend_body == 6)  # BB_FUSED
                assert o.x0 <= o.ax0 and o.ax1 <= o.x1 and o.y0 <= o.ay0 and o.ay1 <= o.y1
            for s in reads:
                if s >= 0:
                    assert regs[s] == true[s], (si, s, "stale surface")
            version += 1
            regs[o.dst] = true[o.dst] = version
        for s in range(5):
            if sg["store_mask"] & (1 << s):
                # a boxed launch rewrites only its box of RGBA0; outside it the registers were loaded from mem
                if s >= 3:
                    assert sg["key_st%d" % (s - 3)] >= 0
                    keybuf[sg["key_st%d" % (s - 3)]] = regs[s]
                else:
                    mem[s] = regs[s]
    assert mem[0] == true[0] and not parked


def mixed_scene(W, H, bits=8, n_before=6, n_after=6):
    ops = [S.layer(S.bgra(W, H, "premul", 10 + i, bits=bits), opacity=0.9) for i in range(n_before)]
    ops.append(S.layer(S.bgra(W // 2, H // 2, "premul", 40, bits=bits), fill_translation=[0.55, 0.5],
                       fill_scale=[0.3, 0.3]))
    ops += [S.layer(S.bgra(W, H, "premul", 60 + i, bits=bits), opacity=0.8) for i in range(n_after)]
    return S.scene("mixed", W, H, ops, bits=bits)


def plan_with_segments(sc):
    m = A.Mixer(channel_id=1, gpu=-2)
    try:
        S.feed_product(m, sc, via_factory=False)
        draws, surfaces = m.flatten(sc["width"], sc["height"], sc["square_width"], sc["square_height"], sc["bits"])
        fmt = A.Format(sc["width"], sc["height"], sc["square_width"], sc["square_height"], sc["bits"], 0)
        ops, _full = K.plan(fmt, surfaces, draws)
        return ops, K.plan_segments(fmt, surfaces, draws)
    finally:
        m.close()


def test_a_scaled_box_between_full_screen_layers_gets_its_own_small_launch():
    W, H = 1920, 1080
    # today the box is sampled by the 4-pixels-per-lane kernel itself: one launch
    ops, segs = plan_with_segments(mixed_scene(W, H))
    assert [s["fast"] for s in segs] == [1] and ops[6].fast_class == K.FC_SAMPLED
    try:
        K.set_cut_policy(sampled_fast=False)
        _scaled_box_cases(W, H)
    finally:
        K.set_cut_policy()


def _scaled_box_cases(W, H):
    ops, segs = plan_with_segments(mixed_scene(W, H))
    assert [s["fast"] for s in segs] == [1, 0, 1]
    check_segments(ops, segs, W, H)
    box = segs[1]
    assert box["load_mask"] == 1 and box["store_mask"] == 1
    assert (box["gx1"] - box["gx0"]) * (box["gy1"] - box["gy0"]) < W * H // 4
    assert segs[2]["load_mask"] == 1 and segs[2]["store_mask"] == 1
    # one 8-bit fast layer on either side of a full-screen general draw does not pay for two cuts (measured:
    # tests/gpu_segments_probe.py), the frame stays one general launch
    sc = mixed_scene(W, H, n_before=1, n_after=1)
    sc["ops"][3][1].update(fill_translation=[0.0, 0.0], fill_scale=[0.999, 0.999])
    ops, segs = plan_with_segments(sc)
    assert len(segs) == 1 and not segs[0]["fast"]
    check_segments(ops, segs, W, H)
    # small frames are launch-bound: never cut
    ops, segs = plan_with_segments(mixed_scene(96, 64))
    assert len(segs) == 1
    K.set_cut_policy(max_segments=1, sampled_fast=False)
    ops, segs = plan_with_segments(mixed_scene(W, H))
    assert len(segs) == 1 and not segs[0]["fast"]


def test_blend_mode_layers_with_one_item_are_fused_in_general_launches():
    """BASELINE config 2: six of its eight layers are blend-mode layers holding one item: CLEAR, item -> layer
    texture, composite with the mode. In a launch of the general kernel each triple becomes ONE op (KOp::fused), the
    layer texture never exists and the launch no longer needs the kernel variant with the RGBA temporaries."""
    import server_amd as A
    sc = S.config2()
    ops, segs = plan_with_segments(sc)
    assert len(segs) == 1 and not segs[0]["fast"] and not segs[0]["full"]
    assert len(ops) == 8 and sum(o.fused for o in ops) == 6 and all(o.kind == K.KOP_DRAW_PLANES and o.dst == 0 for o in ops)
    for o in ops:
        if o.fused:
            assert o.b_blend_mode != 0 and o.blend_mode == 0 and o.b_opacity == 1.0
    check_segments(ops, segs, sc["width"], sc["height"])
    A.lib().cghip_debug_no_layer_fusion(1)
    try:
        ops3, segs3 = plan_with_segments(sc)
        assert len(ops3) == 20 and segs3[0]["full"] and not any(o.fused for o in ops3)
    finally:
        A.lib().cghip_debug_no_layer_fusion(0)
    # a MIX pair shares its texture between two items: not this fusion; the pair has one of its own (fuse_mix_pairs),
    # after which config 4 needs no temporary either
    ops4, segs4 = plan_with_segments(S.config4())
    assert not any(o.fused for o in ops4) and segs4[0]["fast"] and not segs4[0]["full"]
    assert [o.blend_body for o in ops4] == [0, 7, 2, 0]
    A.lib().cghip_debug_no_mix_fusion(1)
    try:
        ops5, segs5 = plan_with_segments(S.config4())
        assert len(ops5) == 5 and segs5[0]["fast"] and segs5[0]["full"]
    finally:
        A.lib().cghip_debug_no_mix_fusion(0)
    # unit-scale items in blend-mode layers: fused ops of the 4-pixels-per-lane kernel WITHOUT the temporaries
    sc = S.scene("blend_stack", 256, 64, [S.layer(S.bgra(256, 64, "premul", 70 + i), opacity=0.8, blend_mode=m)
                                          for i, m in enumerate(["normal", "screen", "multiply", "overlay"])])
    ops5, segs5 = plan_with_segments(sc)
    assert len(ops5) == 4 and sum(o.fused for o in ops5) == 3 and segs5[0]["fast"] and not segs5[0]["full"]
    assert all(o.blend_body == 6 for o in ops5 if o.fused)


def test_nested_blend_sublayer_with_a_mix_pair_fits_two_temporaries():
    """The kernels hold the target, two RGBA temporaries and two key surfaces in registers (INTEGRATION.md section 4
    documents the limit and the CGHIP_ERR_UNSUPPORTED a frame beyond it gets). The deepest nesting the reference's own
    renderer produces for RGBA temporaries - a MIX pair inside a blend-mode sublayer inside a blend-mode layer, then a
    plain item - never has more than two alive: the planner binds four surface slots to the two registers, and the
    oracle issues the same seven draws."""
    W, H = 64, 32
    inner = [("push", dict(layer_depth=1, blend_mode="multiply")),
             ("push", dict(is_mix=1, opacity=0.4)), ("visit", S.bgra(W, H, "premul", 81)), ("pop",),
             ("push", dict(is_mix=1, opacity=0.6)), ("visit", S.bgra(W, H, "premul", 82)), ("pop",),
             ("pop",)]
    nested = S.scene("nested_blend_mix", W, H,
                     [S.layer(S.bgra(W, H, "premul", 80)),
                      [("push", dict(layer_depth=1, blend_mode="screen"))] + inner +
                      [("push", dict(opacity=0.5)), ("visit", S.bgra(W, H, "premul", 83)), ("pop",), ("pop",)]])
    assert _draw_sequence(nested) == [("P", 0, -1), ("M", 2, -1), ("M", 2, -1), ("S", 1, 2), ("S", 0, 1), ("P", 3, -1),
                                      ("S", 0, 3)]
    A.lib().cghip_debug_no_mix_fusion(1)  # the draws as the renderer issues them
    try:
        _d, surfaces, ops, full = product_plan(nested)
    finally:
        A.lib().cghip_debug_no_mix_fusion(0)
    assert len(surfaces) == 4 and full and {o.dst for o in ops} == {0, 1, 2}
    assert len(oracle_log(nested)) == 7
    # what runs: the pair fused with its sublayer's composite, the item with its layer's - no temporary left at all
    _d, surfaces, ops, full = product_plan(nested)
    assert not full and [(o.kind, o.blend_body, o.fused) for o in ops] == [
        (K.KOP_DRAW_PLANES, 0, 0), (K.KOP_DRAW_PLANES, 7, 0), (K.KOP_DRAW_PLANES, 2, 0), (K.KOP_DRAW_SURFACE, 0, 1),
        (K.KOP_DRAW_PLANES, 6, 1)]


def test_more_than_two_live_key_textures_are_parked_in_hbm_across_cuts():
    """image_mixer.cpp:104-114,153: every nesting level of sublayers has a pending layer key of its own, so a frame can
    have any number of key textures alive at once; the kernels hold two. The planner (round 3; rounds 1 - 2 refused such
    frames with CGHIP_ERR_UNSUPPORTED) parks the texture that is needed again last in HBM: KOP_SAVE_KEY in front of a
    cut, KOP_LOAD_KEY behind a later one. Replayed here with the markers kept in the op list: every draw sees the value
    a renderer with unlimited textures would, and a parked texture comes back unchanged. Pixel parity: the golden
    scenes nested_sublayer_keys / nested_sublayer_keys_deep (CPU: oracle vs llvmpipe; GPU: HIP vs both)."""
    by_name = {s["name"]: s for s in S.small_scenes()}
    L = A.lib()
    for name, n_saves, n_bufs in (("nested_sublayer_keys", 1, 2), ("nested_sublayer_keys_deep", 3, 4)):
        sc = by_name[name]
        draws, surfaces, ops_plain, _full = product_plan(sc)
        assert len([c for c in surfaces if c == 1]) == n_saves + 2  # key textures the reference would create
        L.cghip_debug_keep_pseudo_ops(1)
        try:
            ops, segs = plan_with_segments(sc)
        finally:
            L.cghip_debug_keep_pseudo_ops(0)
        assert sum(o.kind == K.KOP_SAVE_KEY for o in ops) == n_saves == sum(o.kind == K.KOP_LOAD_KEY for o in ops)
        check_segments(ops, segs, sc["width"], sc["height"])
        used = {s[k] for s in segs for k in ("key_ld0", "key_ld1", "key_st0", "key_st1")} - {-1}
        assert used == set(range(n_bufs)), used
        # what the kernels get: no markers, same launches
        ops2, segs2 = plan_with_segments(sc)
        assert all(o.kind <= K.KOP_DRAW_SURFACE for o in ops2) and len(ops2) == len(ops) - 2 * n_saves == len(ops_plain)
        assert len(segs2) == len(segs) and sum(s["count"] for s in segs2) == len(ops2)
        assert len(oracle_log(sc)) == len(draws)


def test_random_nested_key_scenes_replay_correctly():
    """The same replay over 200 random nested scenes (tests/fuzz_scenes.py: fuzz_nested_scene), markers kept."""
    from fuzz_scenes import fuzz_nested_scene
    L = A.lib()
    parked = 0
    L.cghip_debug_keep_pseudo_ops(1)
    try:
        for seed in range(200):
            sc = fuzz_nested_scene(seed, 8)
            ops, segs = plan_with_segments(sc)
            check_segments(ops, segs, sc["width"], sc["height"])
            parked += any(o.kind == K.KOP_SAVE_KEY for o in ops)
    finally:
        L.cghip_debug_keep_pseudo_ops(0)
    assert parked > 100, parked


@pytest.mark.parametrize("blend,boxes", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("bits", [8, 16])
def test_mix_pairs_are_fused_when_the_frame_needs_at_most_one_other_temporary(bits, blend, boxes):
    """fuzz_mix_scene: unit-scale stacks with MIX transitions, with blend=True about half of them inside blend-mode
    layers. The planner fuses a pair (BB_MIX_HEAD = 7 on the first item, the second item and the composite right behind
    it, the mix texture's CLEAR gone) when both items cover one box - all qualifying pairs of a frame, provided the frame
    is then left with at most ONE temporary (the layer texture a pair is composited into, or the mix texture of a pair
    that does not qualify), which the planner calls RGBA1. A pair that is all its blend-mode layer holds is fused with
    the layer's composite as well (`fused` on the pair's composite) when the frame then needs no temporary at all. The
    boxes=True: sampled boxes elsewhere in the frame, which the step tolerates when no temporary is left at all.
    The GPU test over these scenes only means something if most of them take these paths."""
    from fuzz_scenes import fuzz_mix_scene
    pairs = heads = still_full = into_layer = layer_fused = 0
    sampled_too = 0
    for seed in range(300 if boxes else 150):
        sc = fuzz_mix_scene(seed, bits, blend, boxes)
        ops, segs = plan_with_segments(sc)
        n_heads = sum(1 for o in ops if o.kind == K.KOP_DRAW_PLANES and o.blend_body == 7)
        if n_heads and any(o.kind == K.KOP_DRAW_PLANES and (o.fused or 12 <= o.fetch_body < 24) for o in ops):
            sampled_too += 1
            assert not any(s["full"] for s in segs)  # with sampled boxes or fused layers in the frame: only without a temporary
        n_pairs = sum(1 for o in sc["ops"] if o[0] == "push" and o[1].get("is_mix")) // 2
        full = any(s["full"] for s in segs)
        if not (len(segs) == 1 and segs[0]["fast"]):  # a draw outside the aligned class: the frame keeps its temporaries
            assert n_heads == 0
            check_segments(ops, segs, sc["width"], sc["height"])
            continue
        assert n_heads <= n_pairs  # a pair that draws nothing - cropped away, hidden - is no op at all
        if n_heads:  # nothing touches the second temporary any more
            for o in ops:
                assert o.dst != 2 and not (o.kind == K.KOP_DRAW_SURFACE and o.src_surface == 2), (seed, o.kind, o.dst)
        if not blend and n_heads and not full:
            assert not any(o.kind == K.KOP_CLEAR and o.dst in (1, 2) for o in ops)
        for i, o in enumerate(ops):
            if o.kind == K.KOP_DRAW_PLANES and o.blend_body == 7:
                b, s_ = ops[i + 1], ops[i + 2]
                assert b.kind == K.KOP_DRAW_PLANES and s_.kind == K.KOP_DRAW_SURFACE and s_.blend_mode == 0
                assert (o.x0, o.y0, o.x1, o.y1) == (b.x0, b.y0, b.x1, b.y1) and o.dst == b.dst == s_.dst and o.dst in (0, 1)
                into_layer += o.dst == 1
                layer_fused += bool(s_.fused)
                assert not s_.fused or (not full and s_.dst == 0 and s_.b_blend_mode != 0)
        check_segments(ops, segs, sc["width"], sc["height"])
        pairs, heads, still_full = pairs + n_pairs, heads + n_heads, still_full + full
    assert pairs > (120 if boxes else 150) and heads > 0.6 * pairs, (pairs, heads, still_full)
    assert sampled_too > 60 or not boxes, sampled_too
    assert (into_layer + layer_fused > 0.2 * heads and layer_fused > 0.05 * heads) if blend else into_layer == layer_fused == 0, (into_layer, layer_fused, heads)


def test_empty_scene_is_one_launch_that_stores_the_frame():
    fmt = A.Format(64, 64, 64, 64, 8, 0)
    segs = K.plan_segments(fmt, [4], [])
    assert len(segs) == 1 and segs[0]["count"] == 0 and segs[0]["store_mask"] == 1 and segs[0]["load_mask"] == 0
    assert (segs[0]["gx1"], segs[0]["gy1"]) == (64, 64)


@pytest.mark.parametrize("bits", [8, 16])
def test_fuzz_scenes_cut_at_every_class_change_hand_surfaces_over_correctly(bits):
    """With the cut cost forced to zero the planner cuts wherever the kernel class changes; the replay above checks
    that keys and layer / mix temporaries that are live across a cut are stored and loaded, for 300 random scenes."""
    from fuzz_scenes import fuzz_scene, fuzz_fast_scene
    n_cut = n_spill = 0
    try:
        K.set_cut_policy(cut_cost_scale=0.0, max_segments=64, sampled_fast=False)
        for seed in range(300):
            sc = fuzz_scene(seed, bits) if seed % 3 else fuzz_fast_scene(seed, bits, True)
            ops, segs = plan_with_segments(sc)
            check_segments(ops, segs, sc["width"], sc["height"])
            n_cut += len(segs) > 1
            n_spill += any((s["load_mask"] | s["store_mask"]) & ~1 for s in segs)
    finally:
        K.set_cut_policy()
    assert n_cut > 60 and n_spill > 20, (n_cut, n_spill)


def _draw_sequence(sc):
    """(kind, target slot, source slot) per recorded draw: 'S' = composite of a temporary, 'K'/'M'/'P' = key / mix /
    plain item."""
    draws, _surf, _ops, _full = product_plan(sc)
    seq = []
    for d in draws:
        if d.source_surface >= 0:
            seq.append(("S", d.target, d.source_surface))
        else:
            t = d.transform
            seq.append(("K" if t.is_key else "M" if t.is_mix else "P", d.target, -1))
    return seq


def test_mix_texture_is_never_reset_within_a_layer():
    """image_mixer.cpp:207-225 binds local_mix_texture to a NAMED rvalue reference and copies from it
    (common/memory.h:633), so the mix texture survives its composite: it is composited again in front of every later
    plain item (:197) and at the layer's end (:150), and later mix items draw into the SAME texture."""
    by_name = {s["name"]: s for s in S.small_scenes()}
    # [mix A, plain B]: composite, B, composite again
    assert _draw_sequence(by_name["mix_then_plain_item"])[1:] == [("M", 1, -1), ("S", 0, 1), ("P", 0, -1), ("S", 0, 1)]
    # [mix A, plain B, mix C, plain D]: C lands in the same texture (slot 1), which is composited three times
    assert _draw_sequence(by_name["mix_plain_mix_plain"])[1:] == [
        ("M", 1, -1), ("S", 0, 1), ("P", 0, -1), ("M", 1, -1), ("S", 0, 1), ("P", 0, -1), ("S", 0, 1)]
    # inside a blend-mode layer the composites go to the layer texture (slot 1), the mix texture is slot 2
    assert _draw_sequence(by_name["mix_then_plain_in_blend_layer"])[1:] == [
        ("M", 2, -1), ("S", 1, 2), ("P", 1, -1), ("S", 1, 2), ("S", 0, 1)]
    # and the oracle's own log agrees on the number of GL draws
    for name, n in (("mix_then_plain_item", 5), ("mix_plain_mix_plain", 8), ("mix_then_plain_in_blend_layer", 6)):
        assert len(oracle_log(by_name[name])) == n, name
