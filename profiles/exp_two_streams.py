"""Experiment: does splitting the batch over two HIP streams of one process help?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from th_rl_amd.batched import GameBatch
import bench
G = 1 << 20
for nsplit in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    gbs = []
    for i in range(nsplit):
        with torch.cuda.stream(streams[i]):
            gbs.append(GameBatch(bench.CFG, n_games=G // nsplit, device="cuda:0", dtype="float32", kernel="wave", seed=0,
                                 game_offset=i * (G // nsplit)).init_tables())
    torch.cuda.synchronize()
    def run(n):
        for _ in range(n):
            for i in range(nsplit):
                with torch.cuda.stream(streams[i]):
                    gbs[i].run(25, sync=False)
    run(1); torch.cuda.synchronize()
    t = time.perf_counter(); run(4); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(nsplit, "streams:", G * 100 * 100 / dt / 1e9, "e9 env-steps/s", flush=True)
    del gbs
