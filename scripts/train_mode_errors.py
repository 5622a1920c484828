import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import agogo_amd as A
from agogo_amd import capi
from test_train_gpu import make_pair, batch_data
ctx = A.Ctx(0)
K, L, FC, W, H, F, Aspace, B = 256, 1, 32, 19, 19, 18, 362, 2
ot, dt = make_pair(ctx, K, L, FC, W, H, F, Aspace, B)
x, pi, v = batch_data(B, F, H, W, Aspace, seed=77)
co = ot.batch(x, pi, v, lr=0.0)
for mode in (capi.COMPUTE_F32_MFMA, capi.COMPUTE_BF16X3, capi.COMPUTE_WINO_H2):
    dt.set_compute_mode(mode | capi.COMPUTE_FORCE)
    cd = dt.forward_backward(x, pi, v)
    errs = []
    for i in range(ot.num_params()):
        go, gd = ot.get_grad(i), dt.get_grad(i)
        errs.append((ot.param_name(i), float(np.abs(gd - go).max()) / (float(np.abs(go).max()) + 1e-30)))
    print("mode", mode, "cost err", abs(cd - co), " ".join("%s=%.1e" % e for e in errs))
