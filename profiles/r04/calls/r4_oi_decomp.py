"""Round-4 measurement: where the fused out->in kernel's time goes (variants with M loads / parameter loads / V2 stores removed)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import agogo_amd as A

ctx = A.Ctx(0)
S, K, L, B = 19, 256, 20, 512
net = A.Net(ctx, K, L, 2 * K, S, S, 18, S * S + 1, bn_mode=A.capi.BN_IDENTITY)
net.init_random(1337)
for i in range(net.num_params()):
    name, n = net.param_info(i)
    if name.endswith("_gamma"):
        net.set_param(i, np.ones(n, np.float32))
    elif name.endswith("_beta"):
        net.set_param(i, np.zeros(n, np.float32))
net.commit()
net.set_compute_mode(A.capi.COMPUTE_WINO_H2)
net.set_tower_queues(1)
x = torch.randint(-1, 2, (B, 18, S, S), device="cuda").float()
pol = torch.empty((B, S * S + 1), device="cuda")
val = torch.empty((B,), device="cuda")
torch.cuda.synchronize()
forms = [int(f) for f in os.environ.get("FORMS", "1,17,18,20,19,21,22,23").split(",")]
for form in forms:
    net.set_wino_h2_form(form)
    for _ in range(2):
        net.infer_dev(x.data_ptr(), B, pol.data_ptr(), val.data_ptr())
    ctx.sync()
    ctx.prof_enable(True)
    for _ in range(4):
        net.infer_dev(x.data_ptr(), B, pol.data_ptr(), val.data_ptr())
    ctx.sync()
    ctx.prof_enable(False)
    r = {"form": form, "dbg": (form - 16) if form >= 16 else 0}
    for nm, k in (("in", A.capi.PROF_WINO_IN), ("gemm", A.capi.PROF_WINO_GEMM), ("out", A.capi.PROF_WINO_OUT)):
        n_, ms_ = ctx.prof_read(k)
        r[nm + "_ms"] = round(ms_ / max(n_, 1), 4)
    print(json.dumps(r), flush=True)
