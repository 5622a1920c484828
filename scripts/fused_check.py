"""Parity probe for the fused Winograd kernel (conv_wino_fused.hpp): networks of several shapes under AGZ_COMPUTE_WINO_H2 (with
whatever AGZ_WINO_H2_FUSED / _VAR the environment selects) against the fp32-MFMA path and, on three boards, the oracle."""
import json
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np

import agogo_amd as A
from test_net_gpu import make_pair, rand_planes

ctx = A.Ctx(0)
out = []
shapes = [(256, 2, 19, 19, 70, 2), (64, 2, 9, 9, 37, 2), (128, 2, 9, 9, 33, 0), (192, 1, 7, 6, 37, 2), (64, 5, 9, 9, 64, 1), (128, 1, 5, 5, 90, 2),
          (256, 3, 19, 19, 130, 2)]
for (K, L, W, H, B, bn) in shapes:
    F = 18 if W >= 9 else 2
    onet, gnet = make_pair(ctx, K, L, 32, W, H, F, W * H + 1, bn)
    x = rand_planes(B, F, H, W, seed=K + B)
    pf, vf = gnet.infer(x)
    gnet.set_compute_mode(A.capi.COMPUTE_WINO_H2 | A.capi.COMPUTE_FORCE)
    pg, vg = gnet.infer(x)
    idx = [0, B // 2, B - 1]
    po, vo = onet.infer(x[idx])
    out.append({"shape": [K, L, W, H, B, bn], "dpol_f32": float(np.abs(pg - pf).max()), "dval_f32": float(np.abs(vg - vf).max()),
                "dpol_oracle": float(np.abs(pg[idx] - po).max()), "dval_oracle": float(np.abs(vg[idx] - vo).max()),
                "f32_dpol_oracle": float(np.abs(pf[idx] - po).max()), "finite": bool(np.all(np.isfinite(pg)) and np.all(np.isfinite(vg)))})
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("AGZ_")}, "results": out}))
