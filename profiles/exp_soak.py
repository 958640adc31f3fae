import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from th_rl_amd.batched import GameBatch
import bench
G = 1 << 20
gb = GameBatch(bench.CFG, n_games=G, device="cuda:0", dtype="float32", kernel="wave", seed=7).init_tables()
t = time.time(); out = gb.run(2000); torch.cuda.synchronize(); dt = time.time() - t
c = gb.counter
tot = int(c.sum(dtype=torch.int64).item())
print("2000 episodes x 1M games: %.1f s, %.3e env-steps/s" % (dt, G * 100 * 2000 / dt))
print("visits", tot, "expected", 2 * G * 100 * 2000, "finite", bool(torch.isfinite(gb.q).all().item()))
print("reward log first/last", out["reward_log"][0], out["reward_log"][-1], "eps", gb.eps[:2])
