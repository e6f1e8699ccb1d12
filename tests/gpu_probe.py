"""Perf probe: kernel time of single-type 16-layer stacks at 2160p (run through gpurun, optionally under rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scenes as S
import server_amd as A
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Channel, run_steps

W, H = int(os.environ.get("PROBE_W", "3840")), int(os.environ.get("PROBE_H", "2160"))
NL = int(os.environ.get("PROBE_LAYERS", "8"))
SUB = os.environ.get("PROBE_SUB", "422")
BLEND_MODE = os.environ.get("PROBE_BLEND", "screen")


def stack(kind, bits):
    t = {"mix_blend": {}, "mix_rect": {}, "uyvy": dict(opacity=0.8), "g_uyvy_rect": dict(opacity=0.8, fill_scale=[0.97, 0.97]),
         "g_uyvy_rot": dict(opacity=0.8, angle=0.05),"plain": dict(opacity=0.8), "key": dict(is_key=1), "chroma": dict(opacity=0.75, **S.CHROMA),
         "levels": dict(opacity=0.7, **S.LEVELS), "bgra": dict(opacity=0.8),
         # general-kernel paths: scaled (axis-aligned), rotated, perspective, unit-scale with a blend mode
         "g_rect": dict(opacity=0.8, fill_scale=[0.97, 0.97]), "g_rot": dict(opacity=0.8, angle=0.05),
         "g_pers": dict(opacity=0.8, pers_ul=[0.02, 0.01], pers_lr=[0.98, 0.99]),
         "g_blend": dict(opacity=0.8, blend_mode=os.environ.get("PROBE_BLEND", "screen")), "g_bgra_rect": dict(opacity=0.8, fill_scale=[0.97, 0.97]),
         "g_unit": dict(opacity=0.8, fill_translation=[1.0 / W, 0.0]),
         # the same inside blend-mode layers: the kernels that also hold the two RGBA temporaries
         "g_rect_blend": dict(opacity=0.8, fill_scale=[0.97, 0.97], blend_mode=os.environ.get("PROBE_BLEND", "screen")),
         "g_rot_blend": dict(opacity=0.8, angle=0.05, blend_mode=os.environ.get("PROBE_BLEND", "screen"))}[kind]
    ops = []
    for i in range(NL):
        spec = S.bgra(W, H, "premul", 500 + i, bits=bits) if "bgra" in kind else S.ycbcr(W, H, SUB, "video", 500 + i, bits=bits)
        if "uyvy" in kind:  # what SDI capture delivers: one U Y0 V Y1 buffer bound twice
            spec = S.frame("uyvy", [(W, H, 2, "video", 500 + i)], bits)
        if kind == "mix_blend":  # a MIX pair inside a blend-mode layer: the kernels with TWO RGBA temporaries
            other = S.bgra(W, H, "premul", 700 + i, bits=bits)
            ops.append([("push", dict(layer_depth=1, blend_mode=BLEND_MODE, opacity=0.8)),
                        ("push", dict(is_mix=1, opacity=0.6)), ("visit", spec), ("pop",),
                        ("push", dict(is_mix=1, opacity=0.4)), ("visit", other), ("pop",), ("pop",)])
            continue
        if kind == "mix_rect":  # MIX pairs between scaled boxes: frames that need the sampled bodies next to the step
            if i % 2:
                ops.append(S.layer(spec, opacity=0.8, fill_scale=[0.97, 0.97]))
            else:
                other = S.bgra(W, H, "premul", 700 + i, bits=bits)
                ops.append([("push", dict(layer_depth=1, opacity=0.8)),
                            ("push", dict(is_mix=1, opacity=0.6)), ("visit", spec), ("pop",),
                            ("push", dict(is_mix=1, opacity=0.4)), ("visit", other), ("pop",), ("pop",)])
            continue
        ops.append(S.layer(spec, **t))
    return S.scene("%s_%d" % (kind, bits), W, H, ops, bits=bits)

which = os.environ.get("PROBE", "plain,key,chroma,levels,bgra").split(",")
if os.environ.get("PROBE_NO_MONO") and hasattr(A.lib(), "cghip_debug_no_mono_kernels"):  # one-format stacks in the kernel with all fetch bodies (A/B of cghip_composite_fast_m)
    A.lib().cghip_debug_no_mono_kernels(1)
if os.environ.get("PROBE_NO_MIX_FUSION"):  # MIX pairs as four draws (A/B of cghip_composite_fast_p / _one_p)
    A.lib().cghip_debug_no_mix_fusion(1)
if os.environ.get("PROBE_NO_LAYER_FUSION"):
    A.lib().cghip_debug_no_layer_fusion(1)
if os.environ.get("PROBE_FORCE_GENERAL"):
    A.lib().cghip_debug_force_general_kernel(1)
BLEND = os.environ.get("PROBE_BLEND", "screen")
for bits in [int(b) for b in os.environ.get("PROBE_BITS", "16,8").split(",")]:
    for kind in which:
        sc = stack(kind, bits)
        ch = Channel(A, S, sc, 0, 1)
        run_steps(ch, int(os.environ.get("PROBE_SETTLE", "2")))
        ms = run_steps(ch, 5 if "PROBE_SETTLE" not in os.environ else 50)
        b = S.algorithmic_bytes(sc)
        print("%-10s %2d-bit  %d layers  kernel %.3f ms  %.0f GB/s  (%.1f ns per Mlayer-px)" % (
            kind, bits, NL, np.mean(ms), b / np.mean(ms) / 1e6, np.mean(ms) * 1e6 / (NL * W * H / 1e6)), flush=True)
        ch.close()
