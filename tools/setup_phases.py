import sys, time
sys.path.insert(0, ".")
import numpy as np
import mgb_amd as M
L = int(sys.argv[1]) if len(sys.argv) > 1 else 7
be = M.backend_hip(0)
M.AMG(M.fem2d_mpi(2, backend=be), p=1.0).prepare()   # warm code objects / context
def T(msg, t0): be.synchronize(); print("%-28s %.3f s" % (msg, time.time() - t0), flush=True); return time.time()
t = time.time(); t00 = t
gn = M.fem2d(L); t = T("fem2d native (C++ -> scipy)", t)
geo = M.native_to_mpi(gn, backend=be); t = T("native_to_mpi (upload)", t)
A = M.AMG(geo, p=1.0); t = T("AMG()", t)
x = geo.x.to_numpy(); t = T("x.to_numpy", t)
z0 = np.vstack([M.DEFAULT_G[2](xi) for xi in x]).reshape(-1, order="F"); c = np.vstack([M.DEFAULT_F[2](xi) for xi in x]); t = T("python f/g loops", t)
A.set_c(c); t = T("set_c", t)
A.prepare(); t = T("prepare (plan + chol)", t)
print("total %.3f" % (time.time() - t00))
