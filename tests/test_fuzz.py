"""Random-scene parity: (CPU, this container) the oracle restatement against llvmpipe running the reference's own
shader; (GPU) the HIP mixer against the oracle. Same seeded scenes (tests/fuzz_scenes.py)."""
import numpy as np
import pytest

import scenes as S
from fuzz_scenes import fuzz_scene

N_CPU, N_GPU = 60, 150


def _gl():
    from oracle import oracle as O
    return O.gl_available()


@pytest.mark.parametrize("seed", range(N_CPU))
def test_oracle_matches_llvmpipe_on_random_scenes(seed):
    """Pins the restatement beyond the fixed goldens. Needs oracle/_ref/libglref.so and the reference's shader files
    (this container only): skipped elsewhere."""
    if not _gl():
        pytest.skip("llvmpipe harness not available here")
    sc = fuzz_scene(seed, 8)
    cache = {}
    a = S.run_oracle(sc, "cpu", cache)
    b = S.run_oracle(sc, "gl", cache)
    if sc["field_mode"]:  # the field mask is this project's extension: GL draws every row, compare the field's rows
        first = sc["field_mode"] - 1
        assert not a[1 - first::2].any()
        a, b = a[first::2], b[first::2]
    st = S.diff_stats(a, b)
    # Bit-exact except where a texel coordinate lands within one float ulp of a rounding boundary of the 8-bit filter
    # weight (e.g. seed 28: u*256 = 14843.4992 along one column): llvmpipe's interpolation and the restatement's
    # differ in the last bit of s there, the weight moves by 1/256, the sample by 1 LSB, and a steep later stage
    # (multiply / dodge blends, levels gamma, chroma key) can amplify it. Bound the population, not the maximum.
    assert st["gt0"] <= 2e-3 and st["gt1"] <= 5e-4 and st["max"] <= 8, st


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [8, 16])
def test_hip_matches_oracle_on_random_scenes(bits):
    import server_amd as A
    m = A.Mixer(channel_id=3, gpu=0)
    worst = {"max": 0, "gt0": 0.0}
    try:
        for seed in range(N_GPU):
            sc = fuzz_scene(seed, bits)
            cache = {}
            got = S.run_product(sc, mixer=m, planes_cache=cache)
            ref = S.run_oracle(sc, "cpu", cache)
            st = S.diff_stats(got, ref)
            if bits == 8:
                # bit-exact. (This test found, and the kernels now avoid: x*(1/c) for the shader's x/c, a one-step
                # Newton divide, log/exp for pow(r, 1), FMA contraction.) Scenes whose levels use a gamma other than 1
                # may differ by 1 LSB in a few samples per million (v_log/v_exp vs powf): none among these seeds.
                assert st["max"] == 0, (seed, st)
            else:
                # fp32 filter weights: the kernels place unit-scale texels exactly, llvmpipe and the oracle a few ulp
                # off (1 LSB16 on ~1 % of samples); blends that divide by a tiny alpha and the chroma key's thresholds
                # amplify single samples. Bound the population within the north star's +-2 LSB16.
                assert st["gt2"] <= 2e-3, (seed, st)
            worst = {"max": max(worst["max"], st["max"]), "gt0": max(worst["gt0"], st["gt0"])}
    finally:
        m.close()
    print("worst over %d scenes (%d-bit): %s" % (N_GPU, bits, worst))
