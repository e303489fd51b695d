import sys, time
sys.path.insert(0, '/root/repo')
import torch
from navierstokes_amd.bubble import DNS, DNSParams, Code3, Code3Params
g = DNS(DNSParams())
for _ in range(20): g.step()
torch.cuda.synchronize(); t=time.time(); sw=[]
for _ in range(200): sw.append(g.step())
torch.cuda.synchronize(); dt=(time.time()-t)/200
print("DNS: mean sweeps", sum(sw)/len(sw), "us/step", dt*1e6, "steps of pipeline", 63+2*(sum(sw)/len(sw)-1))
g = Code3(Code3Params())
for _ in range(20): g.step()
torch.cuda.synchronize(); t=time.time(); sw=[]
for _ in range(100): sw.append(g.step())
torch.cuda.synchronize(); dt=(time.time()-t)/100
import numpy as np
print("code3: mean sweeps", np.mean(np.array(sw),axis=0), "us/step", dt*1e6)
