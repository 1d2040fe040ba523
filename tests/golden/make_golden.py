"""Regenerates the oracle-derived fixtures in this directory (run from the repo root:
`python tests/golden/make_golden.py`).  rng_minstd_rand0.json is NOT produced here: it holds the
libstdc++ draws recorded in SURVEY.md section 0.4 and pins the oracle, not the other way round."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_binding as ob  # noqa: E402


def rng_counter():
    L = ob.lib()
    keys = [[1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 1], [7, 3, 11, 259199, 17, 40], [0xFFFFFFFF, 2, 5, 100, 31, 16],
            [12345, 1, 2, 3, 4, 5]]
    vals = [float(np.float32(L.orc_rng_uniform(*k))) for k in keys]
    json.dump({"source": "oracle engine counter RNG (no reference counterpart)", "keys": keys, "values": vals},
              open(os.path.join(HERE, "rng_counter.json"), "w"), indent=1)


if __name__ == "__main__":
    rng_counter()
    print("golden fixtures written to", HERE)
