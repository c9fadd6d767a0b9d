"""Parameter sweep of solver="pcg" against the direct solver on one mesh: Newton steps, CG iterations, fallbacks, time, z error."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import mgb_amd as M

kind, L, p = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
solve = getattr(M, kind + "_mpi_solve")
ref = solve(L=L, p=p)
zr = M.mpi_to_native(ref).z
print("%s L=%d p=%g direct: newton %d, %.3f s" % (kind, L, p, int(ref.SOL_main["its"].sum()), ref.SOL_main["t_elapsed"]))
combos = [dict(rtol=1e-9, degree=2), dict(rtol=1e-6, degree=2), dict(rtol=1e-4, degree=2), dict(rtol=1e-2, degree=2),
          dict(rtol=1e-6, degree=4), dict(rtol=1e-4, degree=4), dict(rtol=1e-4, degree=3, maxit=60)]
for extra in sys.argv[4:]:
    combos = [eval("dict(%s)" % extra)]
for c in combos:
    try:
        s = solve(L=L, p=p, solver="pcg", pcg=dict(c))
        z = M.mpi_to_native(s).z
        pc = s.SOL_main["pcg"]
        print("  pcg %-42s newton %4d  cg %6d (%.1f/system)  fallbacks %4d  %.3f s  |z - z_direct| %.2e" % (
            c, int(s.SOL_main["its"].sum()), pc["iterations"], pc["iterations"] / max(pc["solves"], 1), pc["fallbacks"],
            s.SOL_main["t_elapsed"], np.linalg.norm(z - zr) / np.linalg.norm(zr)))
    except Exception as e:
        print("  pcg %s failed: %s" % (c, e))
