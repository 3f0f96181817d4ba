# probes HIP-runtime sharing between torch and libfhe_hip.so: torch initialised first, then our context
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init(); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
import numpy as np
import fhe_amd
from oracle import oracle as orc
pr, ps = orc.gen_primes(12, 3)
g = fhe_amd.HipContext(12, 2, 1, pr, ps)
o = orc.Oracle(12, 2, 1, pr, ps)
a = np.random.default_rng(0).integers(0, int(pr[0]), (1, 4096), dtype=np.uint64)
d = torch.from_numpy(a.view(np.int64)).cuda()
g.ntt_forward(d, [0])
print("probe ok", np.array_equal(d.cpu().numpy().view(np.uint64), o.ntt_forward(a.copy(), [0])))
