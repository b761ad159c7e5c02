import faulthandler, sys, time
faulthandler.dump_traceback_later(25, exit=True)
sys.path.insert(0, ".")
import numpy as np
from oracle import cpu_oracle as o
from neighborhoodwatch_amd import KnnEngine
eng = KnnEngine(0)
print("engine created", flush=True)
base, queries = o.gen(41, 0, 2500, 96), o.gen(42, 0, 130, 96)
eng.set_base(base); d, i = eng.search(queries, 20, "cosine"); print("host path ok", flush=True)
import torch
print("torch imported", flush=True)
tb = torch.from_numpy(base).cuda(); print("tb on device", flush=True)
tq = torch.from_numpy(queries).cuda()
torch.cuda.synchronize(); print("sync ok", flush=True)
eng.set_base(tb); print("set_base dev ok", flush=True)
d2, i2 = eng.search(tq, 20, "cosine"); print("search dev ok", flush=True)
print(np.array_equal(i2.cpu().numpy(), i))
