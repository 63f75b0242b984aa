import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from smcp_amd import problems
from smcp_amd.symbolic import Symbolic
t0 = time.time(); symb = Symbolic(problems.nested_block_arrow_pattern()); t1 = time.time()
symb.device_init(0, 1); torch.cuda.synchronize(); t2 = time.time()
F = symb.replicate(8); t3 = time.time()
F.device_init(0, 1); torch.cuda.synchronize(); t4 = time.time()
print("symbolic %.3f  base device_init %.3f  replicate %.3f  forest device_init %.3f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3))
