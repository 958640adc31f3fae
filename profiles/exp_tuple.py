"""Throughput of the tuple-chain kernel vs the generic kernel on the three-player shape (golden G5) at 65,536 games."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from th_rl_amd.batched import GameBatch
AG = dict(name="QTable", gamma=0.95, actions=21, states=100, alpha=0.1, eps_end=0.001, epsilon=0.5, eps_step=0.9995, action_range=[0.2, 0.4])
ENV = dict(name="NoisyPriceState", noise_prob=0, a=10, b=1, nplayers=2, max_steps=100)
THREE = {"agents": [dict(AG, actions=11, states=50, action_range=[0.1, 0.3], min_memory=25),
                    dict(AG, actions=21, states=100, action_range=[0.15, 0.35], min_memory=25),
                    dict(AG, actions=5, states=20, action_range=[0.0, 0.3], min_memory=25, max_state=10)],
         "environment": dict(ENV, nplayers=3, max_steps=25)}
TWO_GRIDS = {"agents": [dict(AG, actions=15, min_memory=100), dict(AG, actions=21, action_range=[0.15, 0.45])], "environment": dict(ENV)}
def noisy(cfg, p=0.05):
    return {"agents": cfg["agents"], "environment": dict(cfg["environment"], noise_prob=p)}
for name, cfg, G in (("three players (11/21/5 actions, T=25)", THREE, 65536), ("three players", THREE, 1 << 20),
                     ("two agents, different grids (15/21 actions, T=100)", TWO_GRIDS, 65536),
                     ("three players, noise_prob 0.05", noisy(THREE), 65536),
                     ("two agents, different grids, noise_prob 0.05", noisy(TWO_GRIDS), 65536)):
    for kern in ("tuple", "generic"):
        if kern == "generic" and G > 65536:
            continue
        gb = GameBatch(cfg, n_games=G, dtype="float32", kernel=kern, seed=0).init_tables()
        E = 32 if kern == "tuple" else 8
        gb.run(E, sync=False); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            gb.run(E, sync=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        T = cfg["environment"]["max_steps"]
        print("%-52s G=%8d %-8s %.3e env-steps/s (%.2f ms per %d episodes)" % (name, G, gb.last_kernel, G * T * E / dt, dt * 1e3, E), flush=True)
        del gb
        torch.cuda.empty_cache()
