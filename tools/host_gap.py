import os, sys, time
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0, os.path.join(ROOT,"dark-archon_amd"))
import numpy as np, torch, pyarchon, archon_synth as S
n=256<<20
x=torch.from_numpy(S.gen_random(n)).cuda(); sa=torch.empty(n,dtype=torch.int32,device="cuda"); bwt=torch.empty(n,dtype=torch.uint8,device="cuda"); base=torch.zeros(1,dtype=torch.int32,device="cuda")
for _ in range(3): pyarchon.forward_dev(x,sa,bwt,base)
torch.cuda.synchronize()
calls=[]; devs=[]; stats_t=[]
t_all0=time.perf_counter()
for _ in range(30):
    t0=time.perf_counter(); pyarchon.forward_dev(x,sa,bwt,base); t1=time.perf_counter()
    st=pyarchon.stats(); t2=time.perf_counter()
    calls.append((t1-t0)*1e3); devs.append(st["ms_total"]); stats_t.append((t2-t1)*1e3)
torch.cuda.synchronize(); t_all=(time.perf_counter()-t_all0)*1e3/30
print("per step wall %.3f ms; call %.3f ms; device %.3f ms; stats() %.4f ms; call - device = %.3f ms" % (t_all, np.mean(calls), np.mean(devs), np.mean(stats_t), np.mean(calls)-np.mean(devs)))
