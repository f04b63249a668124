import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from microhh_amd.model import HotPath
hp = HotPath("drycblles", 512, 512, 512, device="cuda:0", dt=0.5)
hp.cyclic_prognostic(); hp.exec_viscosity(); hp.rhs(); hp.sync()
dbg = torch.zeros(8*8, dtype=torch.int64, device="cuda:0")
os.environ["MHH_EXP_DBGPTR"] = str(dbg.data_ptr())
os.environ["MHH_PRES_LDS"] = "1"
for _ in range(3): hp.pres()
hp.sync()
d = dbg.cpu().numpy().reshape(8, 8)
dif = np.diff(d[:, :7], axis=1)
print("stage 1, cycles per phase (loads + divergence -> LDS, barrier, transform, barrier, split + stores, barrier):")
print(np.median(dif, axis=0).astype(int), " level:", int(np.median(d[1:,0]-d[:-1,0])))
