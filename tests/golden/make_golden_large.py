"""Generates the BASELINE-size golden fixtures tests/golden/large_*.npz from the CPU oracle
(oracle/mgb_oracle.py) -- NOT from the reference: Julia is absent from the build image and the
reference stores no solve outputs, so these vectors pin the oracle/HIP pair at the
headline sizes ("parity unpinned" at solve level).  The oracle needs minutes per case here, which is
why the GPU box compares against these files instead of running it.

    python tests/golden/make_golden_large.py fem2d 7 1.0
    python tests/golden/make_golden_large.py fem2d 7 1.5
    python tests/golden/make_golden_large.py fem3d 4 1.0
    python tests/golden/make_golden_large.py parabolic 6 1.0     (h = 0.1, t1 = 1: snapshots 1, 5, 10)

Only z (and the scalar histories) are stored: fem2d L=7 z is 57 344 x 2 doubles."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import mgb_oracle as O  # noqa: E402

PARABOLIC_KEEP = (1, 5, 10)


def main():
    kind, L, p = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    tag = "%s_L%d_p%s" % (kind, L, str(p).replace(".", "_"))
    t0 = time.time()
    if kind == "parabolic":
        g = O.fem2d(L)
        sol = O.parabolic_solve(g, h=0.1, t1=1.0, p=p)
        out = dict(ts=sol.ts, keep=np.array(PARABOLIC_KEEP))
        for k in PARABOLIC_KEEP:
            out["u_%d" % k] = sol.u[k]
        info = "snapshots %d" % len(sol.u)
    else:
        g = getattr(O, kind)(L)
        sol = O.amgb(g, p=p)
        out = dict(z=sol.z, its=sol.SOL_main["its"], ts=sol.SOL_main["ts"], c_dot_Dz=sol.SOL_main["c_dot_Dz"])
        info = "steps %d |z| %.17g" % (int(sol.SOL_main["its"].sum()), np.linalg.norm(sol.z))
    out["oracle_seconds"] = time.time() - t0
    np.savez_compressed(os.path.join(HERE, "large_%s.npz" % tag), **out)
    print(tag, info, "oracle %.0f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
