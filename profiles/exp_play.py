import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from th_rl_amd.batched import GameBatch
import bench
for G in (65536, 1 << 20):
    gb = GameBatch(bench.CFG, n_games=G, device="cuda:0", dtype="float32", kernel="wave", seed=0).init_tables()
    gb.run(25)
    gb.play_greedy(iters=1); torch.cuda.synchronize()
    t = time.perf_counter(); mr, ma = gb.play_greedy(iters=10); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(G, "games: play_greedy", G * 100 * 10 / dt / 1e9, "e9 env-steps/s", dt * 1e3, "ms; mean total reward", mr.sum(axis=1).mean(), flush=True)
