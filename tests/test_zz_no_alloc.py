"""GPU test of the entry point that allocates nothing (cghip_mixer_render_noalloc, include/cghip.h): through the C ABI
(libcghost.so -> libcghip.so), against the existing entry point (cghip_mixer_render_async) on the same scenes, bit for
bit, on two calls in a row, the second with a smaller input. torch only provides the caller's device memory."""
import numpy as np
import pytest

import kops as K
import scenes as S

pytestmark = pytest.mark.gpu


def _mixed_scene(W, H, bits, n_fill):
    """Unit-scale layers (4-pixels-per-lane kernel) around scaled ones (general kernel), a key drawn on one side of a
    cut and used on the other, a blend-mode layer and a MIX pair: with the cut cost forced to zero the frame becomes
    several launches that hand KEY and RGBA temporaries over through the scratch array."""
    ops = [S.layer(S.ycbcr(W, H, "422", "video", 3000 + i, bits=bits), opacity=0.9) for i in range(n_fill)]
    ops.append(S.layer(S.bgra(W // 2, H // 2, "premul", 3010, bits=bits), fill_translation=[0.55, 0.5], fill_scale=[0.3, 0.3]))
    ops.append(S.layer(S.bgra(W, H, "smooth", 3011, bits=bits), is_key=1))
    ops.append(S.layer(S.bgra(W, H, "premul", 3012, bits=bits), fill_scale=[0.999, 0.999], opacity=0.7))
    ops.append([("push", dict(layer_depth=1, blend_mode="screen")),
                ("push", dict(is_mix=1, opacity=0.4)), ("visit", S.bgra(W, H, "premul", 3013, bits=bits)), ("pop",),
                ("push", dict(opacity=0.5, fill_scale=[0.5, 0.5])), ("visit", S.bgra(W, H, "premul", 3014, bits=bits)), ("pop",),
                ("pop",)])
    ops += [S.layer(S.ycbcr(W, H, "420", "video", 3020 + i, bits=bits), opacity=0.6) for i in range(n_fill)]
    return S.scene("noalloc_%dx%d_%d" % (W, H, bits), W, H, ops, bits=bits)


def _render_both(m, sc, torch):
    """(frame from cghip_mixer_render_async, frame from cghip_mixer_render_noalloc, launches of the latter)"""
    cache = {}
    want = S.run_product(sc, mixer=m, planes_cache=cache)
    nbytes = want.nbytes
    buf = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device="cuda:0")  # poisoned: every byte must be written
    torch.cuda.synchronize()
    S.feed_product(m, sc, cache)
    before = m.device_info()
    m.render_noalloc(buf.data_ptr(), nbytes, sc["width"], sc["height"], sc["square_width"], sc["square_height"],
                     sc["bits"], sc["field_mode"])
    after = m.device_info()
    # the library's pools are how it allocates: not one block more, device or pinned
    for pool in ("device_pool", "pinned_pool"):
        assert after[pool]["blocks"] == before[pool]["blocks"] and after[pool]["bytes"] == before[pool]["bytes"], pool
    assert after["tickets_live"] == before["tickets_live"]
    # ... and the device itself says the same: hipMemGetInfo's free bytes did not move across the call (what the
    # library's counters cannot see: an allocation made by the runtime on its behalf, e.g. to stage a pageable copy)
    assert after["hbm_free_bytes"] == before["hbm_free_bytes"], (before["hbm_free_bytes"], after["hbm_free_bytes"])
    got = buf.cpu().numpy().view(want.dtype).reshape(want.shape)
    return want, got, after["kernel_launches"] - before["kernel_launches"], cache


def test_noalloc_entry_point_equals_render_async_on_two_calls_in_a_row():
    import torch
    import server_amd as A
    m = A.Mixer(channel_id=21, gpu=0)
    K.set_cut_policy(cut_cost_scale=0.0, max_segments=64, sampled_fast=False)  # cut wherever the kernel class changes
    try:
        first = True
        for (W, H, bits, n_fill) in ((1024, 360, 16, 3), (320, 96, 8, 1)):  # the second call is the smaller one
            sc = _mixed_scene(W, H, bits, n_fill)
            want, got, launches, cache = _render_both(m, sc, torch)
            assert launches > 2, launches  # hand-over surfaces went through the scratch array
            assert np.array_equal(got, want), S.diff_stats(got, want)
            if bits == 8:  # and the oracle, to the bar of every other 8-bit frame
                assert np.array_equal(got, S.run_oracle(sc, "cpu", cache))
            first = False
        # a frame that is one launch (no hand-over surfaces) at a size whose spills would not fit
        sc = S.config3(1920, 1080, 16)
        want, got, launches, _ = _render_both(m, sc, torch)
        assert launches == 1 and np.array_equal(got, want)
        # the empty scene is the zero frame
        out = torch.full((64 * 32 * 4,), 0xAB, dtype=torch.uint8, device="cuda:0")
        m.render_noalloc(out.data_ptr(), out.numel(), 64, 32)
        assert not out.cpu().numpy().any()
    finally:
        K.set_cut_policy()
        m.close()


def test_noalloc_entry_point_refuses_what_does_not_fit():
    import torch
    import server_amd as A
    m = A.Mixer(channel_id=22, gpu=0)
    K.set_cut_policy(cut_cost_scale=0.0, max_segments=64, sampled_fast=False)
    try:
        W, H = 3840, 2160  # hand-over surfaces of a cut 2160p 16-bit frame: 2 x 66 MB + keys > 48 MiB of scratch
        sc = _mixed_scene(W, H, 16, 1)
        cache = {}
        S.feed_product(m, sc, cache)
        buf = torch.empty((W * H * 8,), dtype=torch.uint8, device="cuda:0")
        with pytest.raises(A.capi.CGHIPError) as e:
            m.render_noalloc(buf.data_ptr(), buf.numel(), W, H, bits=16)
        assert "scratch" in str(e.value)
        # output buffer too small
        sc = S.config1(256, 64)
        S.feed_product(m, sc, cache)
        with pytest.raises(A.capi.CGHIPError):
            m.render_noalloc(buf.data_ptr(), 256 * 64 * 4 - 1, 256, 64)
        # the mixer is still usable
        assert np.array_equal(S.run_product(sc, mixer=m), S.run_oracle(sc, "cpu"))
    finally:
        K.set_cut_policy()
        m.close()
