#!/usr/bin/env python3
"""Does the end point of a solve depend on harmless changes of rounding order?  Runs the same solve in fresh processes with
different backward-sweep splits (MGB_BWD_SPLIT_NF: another summation order inside the triangular solves, same mathematics)
and leaf sizes, and compares z, the Newton counts and the final t.
usage: python3 tools/robust_sweep.py [L p]  ->  profiles/rN_robust_sweep.txt"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
import mgb_amd as M
sol = M.fem2d_mpi_solve(L=%d, p=%g)
z = M.mpi_to_native(sol).z
np.save(sys.argv[1], z)
print(json.dumps(dict(newton=int(sol.SOL_main["its"].sum()), nt=len(sol.SOL_main["ts"]), t_final=float(sol.SOL_main["ts"][-1]),
                      cdot=float(sol.SOL_main["c_dot_Dz"][-1]))))
'''

if __name__ == "__main__":
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    p = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    variants = [("default", {}), ("bwd_split_96", {"MGB_BWD_SPLIT_NF": "96"}), ("bwd_split_never", {"MGB_BWD_SPLIT_NF": "100000"}),
                ("leaf_96", {"MGB_LEAF": "96"}), ("no_fused_trial", {"MGB_FUSED_TRIAL_ROWS": "0"}),
                ("one_panel_steps", {"MGB_CHOL_STEP2": "0"}),      # front_step2 is bitwise two front_step launches: distance exactly 0
                # ... and so are the two panels as a panel + an update launch (forced for every multi-panel height) and the
                # three-per-CU single-panel tiles (forced for every single-panel launch)
                ("panel_update_pairs", {"MGB_CHOL_STEP2_TILES": "0"}), ("dense_single_tiles", {"MGB_CHOL_DENSE_TILES": "0"})]
    zs = {}
    print("# fem2d L=%d p=%g: one solve per variant (fresh process each)" % (L, p))
    for name, env in variants:
        out = "/tmp/robust_%s.npy" % name
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, L, p), out], env=dict(os.environ, **env), capture_output=True, text=True)
        if r.returncode != 0:
            print(name, "FAILED", r.stderr[-300:])
            continue
        info = json.loads(r.stdout.strip().splitlines()[-1])
        zs[name] = np.load(out)
        d = np.linalg.norm(zs[name] - zs["default"]) / np.linalg.norm(zs["default"]) if "default" in zs else 0.0
        print("%-18s newton %4d  t-steps %3d  t_final %.6g  c.Dz %.15g  |z - z_default|/|z| = %.3e" % (
            name, info["newton"], info["nt"], info["t_final"], info["cdot"], d), flush=True)
