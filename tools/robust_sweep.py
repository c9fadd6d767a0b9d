import sys, time
sys.path.insert(0,'.')
import numpy as np, mgb_amd as M
def run(kind, L, p):
    t=time.time()
    try:
        sol=getattr(M, kind+"_mpi_solve")(L=L, p=p)
        s=sol.SOL_main
        z=M.mpi_to_native(sol).z
        print("%s L=%d p=%g: steps=%d nt=%d solve=%.2fs wall=%.2fs tfinal=%.3g cdot=%.10g |z|=%.10g"%(kind,L,p,int(s['its'].sum()),len(s['ts']),s['t_elapsed'],time.time()-t,s['ts'][-1],s['c_dot_Dz'][-1],np.linalg.norm(z)), flush=True)
    except Exception as e:
        print("%s L=%d p=%g: FAILED %s"%(kind,L,p,str(e)[:200]), flush=True)
for p in (1.0,1.2,1.5,2.0,3.0,4.0):
    for L in (4,6,7):
        run("fem2d",L,p)
for p in (1.0,1.5,2.0,4.0):
    for L in (6,10,13):
        run("fem1d",L,p)
run("fem2d",8,1.5); run("fem2d",8,2.0)
