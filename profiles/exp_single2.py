import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from th_rl_amd.mixed import MixedGameBatch
from th_rl_amd.batched import GameBatch
import bench
cfg = {"agents": bench.CFG["agents"], "environment": bench.CFG["environment"]}
for G in (1, 256, 4096, 16384, 65536):
    E = 2000 if G <= 256 else (200 if G <= 4096 else 40)
    mb = MixedGameBatch(cfg, n_games=G, dtype="float64", seed=1).init_tables()
    mb.run(2); torch.cuda.synchronize()
    t = time.perf_counter(); mb.run(E); torch.cuda.synchronize(); dm = time.perf_counter() - t
    gb = GameBatch(cfg, n_games=G, dtype="float64", seed=1, kernel="generic").init_tables()
    gb.run(2); torch.cuda.synchronize()
    t = time.perf_counter(); gb.run(E); torch.cuda.synchronize(); dg = time.perf_counter() - t
    print("G=%d f64: mixed-wave %.3e env-steps/s, generic %.3e" % (G, G * E * 100 / dm, G * E * 100 / dg), flush=True)
