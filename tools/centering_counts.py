import sys, numpy as np
sys.path.insert(0, "/root/repo")
import os
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0, ROOT)
import mgb_amd as M
for kind,L,p in (("fem2d",7,1.0),("fem2d",7,1.5),("fem3d",4,1.0),("fem2d",8,1.0)):
    res={}
    for c in ("exact","decrement"):
        s=getattr(M,kind+"_mpi_solve")(L=L,p=p,centering=c)
        res[c]=(M.mpi_to_native(s).z,int(s.SOL_main["its"].sum()),s.SOL_main["t_elapsed"],len(s.SOL_main["ts"]))
    print(kind,L,p,"newton exact %d decrement %d  nt %d/%d  time %.3f/%.3f  |dz| %.2e" % (res["exact"][1],res["decrement"][1],res["exact"][3],res["decrement"][3],res["exact"][2],res["decrement"][2],np.linalg.norm(res["exact"][0]-res["decrement"][0])/np.linalg.norm(res["exact"][0])))
