"""dual.Train step time on one MI355X (config #4 network: 19x19, K=256, 20 blocks; B = dual.Config.BatchSize 256)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import agogo_amd as A

ap = argparse.ArgumentParser()
ap.add_argument("--K", type=int, default=256); ap.add_argument("--L", type=int, default=20)
ap.add_argument("--B", type=int, default=256); ap.add_argument("--size", type=int, default=19)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--x3", action="store_true", help="AGZ_COMPUTE_BF16X3 forward / data-gradient / weight-gradient GEMMs")
ap.add_argument("--wino-h2", action="store_true", help="AGZ_COMPUTE_WINO_H2 forward / data-gradient convolutions (bf16x3 weight gradient)")
ap.add_argument("--hook", type=int, default=-1, help="agz_trainer_set_dma_forward value for the whole run (agz_debug.h)")
ap.add_argument("--fb", action="store_true", help="--hooks: time forward_backward (no SGD step: decomposition hooks that compute garbage leave the weights alone)")
ap.add_argument("--hooks", default="", help="comma list of agz_trainer_set_dma_forward values (agz_debug.h) to time in turn in this process, e.g. 1,9,1,9")
args = ap.parse_args()
S = args.size
ctx = A.Ctx(0)
t = A.Trainer(ctx, args.K, args.L, 2 * args.K, S, S, 18, S * S + 1, args.B)
t0 = time.perf_counter(); t.init_random(1337); t_init = time.perf_counter() - t0
if args.wino_h2:
    t.set_compute_mode(A.capi.COMPUTE_WINO_H2)
elif args.x3:
    t.set_compute_mode(A.capi.COMPUTE_BF16X3)
if args.hook >= 0:
    t.set_dma_forward(args.hook)
rng = np.random.default_rng(0)
x = rng.choice(np.array([-1, 0, 1], np.float32), size=(args.B, 18, S, S)).astype(np.float32)
pi = np.zeros((args.B, S * S + 1), np.float32); pi[np.arange(args.B), rng.integers(0, S * S + 1, args.B)] = 1
v = rng.choice(np.array([-1, 0, 1], np.float32), size=args.B).astype(np.float32)
c = t.batch(x, pi, v)
t0 = time.perf_counter()
for _ in range(args.steps):
    c = t.batch(x, pi, v)
dt = (time.perf_counter() - t0) / args.steps
if args.hooks:
    ab = []
    for hv in [int(q) for q in args.hooks.split(",")]:
        t.set_dma_forward(hv)
        step = (lambda: t.forward_backward(x, pi, v)) if args.fb else (lambda: t.batch(x, pi, v))
        step()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            c2 = step()
        ab.append({"hook": hv, "step_ms": round((time.perf_counter() - t0) / args.steps * 1e3, 3), "cost": c2})
    print(json.dumps({"ab": ab}))
hw = S * S
flops = 3 * (2.0 * 18 * args.K * 9 * hw + args.L * 2 * 2.0 * args.K * args.K * 9 * hw) * args.B  # fwd + dgrad + wgrad
print(json.dumps({"B": args.B, "K": args.K, "L": args.L, "step_ms": dt * 1e3, "examples_per_s": args.B / dt,
                  "tflops": flops / dt / 1e12, "cost": c, "init_s": t_init}))
