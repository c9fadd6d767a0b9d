import sys, os, re, subprocess
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0, ROOT)
import mgb_amd as M, io, contextlib
import collections
# verbose=2 prints per Newton step to stderr: capture via fd redirect
r, w = os.pipe()
saved = os.dup(2); os.dup2(w, 2)
sol = M.fem2d_mpi_solve(L=7, p=1.0, verbose=2)
os.dup2(saved, 2); os.close(w)
data = os.fdopen(r).read()
steps = re.findall(r"step=([0-9.e+-]+)", data)
c = collections.Counter(steps)
print("newton", int(sol.SOL_main["its"].sum()), "accepted step sizes:", c.most_common(8))
