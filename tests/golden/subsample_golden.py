"""Shrinks a headline-size golden to a committable fixture: every `stride`-th row of z / z_centre plus functionals of the
FULL vectors (column norms and the projection on a fixed deterministic vector), so that the GPU test still checks the whole
solution through size-independent numbers.  Used for fem2d L=8 (229 376 rows: 7 MB as full vectors).

    python tests/golden/subsample_golden.py tests/golden/large_fem2d_L8_p1_0.npz 4
"""
import sys

import numpy as np


def probe(n):
    return np.sin(0.37 * np.arange(n) + 0.11)


def main():
    path, stride = sys.argv[1], int(sys.argv[2])
    d = dict(np.load(path))
    out = {k: v for k, v in d.items() if k not in ("z", "z_centre")}
    n = d["z"].shape[0]
    out["n_full"] = n
    out["stride"] = stride
    for key in ("z", "z_centre"):
        if key in d:
            out[key] = d[key][::stride].copy()
            out[key + "_colnorm"] = np.linalg.norm(d[key], axis=0)
            out[key + "_probe"] = probe(n) @ d[key]
    np.savez_compressed(path, **out)
    print(path, "rows", n, "->", out["z"].shape[0])


if __name__ == "__main__":
    main()
