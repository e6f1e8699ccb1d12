"""Regenerates tests/golden/goldens.npz: the outputs of the REFERENCE's own shaders
(/root/reference/src/accelerator/ogl/image/shader.{vert,frag}, loaded as data) executed by Mesa llvmpipe through
oracle/gl_backend.c, for every scene of tests/scenes.py::all_golden_scenes(). Inputs are seeded (scenes.py), so
only outputs are stored. Runs only where /root/reference and the llvmpipe harness exist (the authoring
container); the fixtures travel, the reference does not.

    python tests/golden/generate_goldens.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import scenes as S  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    O.build()
    info = O.gl().glref_info().decode()
    out = {}
    for sc in S.all_golden_scenes():
        out[sc["name"]] = S.run_oracle(sc, "gl")
    np.savez_compressed(os.path.join(HERE, "goldens.npz"), **out)
    meta = {"generator": "tests/golden/generate_goldens.py", "gl": info,
            "shaders": "reference accelerator/ogl/image/shader.vert + shader.frag, unmodified, env force_compat_shaders=true",
            "scenes": sorted(out)}
    with open(os.path.join(HERE, "goldens.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote %d goldens (%s)" % (len(out), info))


if __name__ == "__main__":
    main()
