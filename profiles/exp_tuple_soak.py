"""Soak of the tuple-chain kernel against the one-thread-per-game kernel at scale, in the regimes that stress the replay's
ordering (a converged game rewrites the row it reads next): 65,536 games, 8 launches of 32 episodes each kernel, same seeds --
tables, visit counters and env state must be bit-identical.  Exploring and near-greedy schedules, three players and two grids,
with and without env noise."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from th_rl_amd.batched import GameBatch
AG = dict(name="QTable", gamma=0.95, actions=21, states=100, alpha=0.1, eps_end=0.001, epsilon=0.5, eps_step=0.9995, action_range=[0.2, 0.4])
ENV = dict(name="NoisyPriceState", noise_prob=0, a=10, b=1, nplayers=2, max_steps=100)
def three(eps, noise):
    ag = dict(AG, epsilon=eps, eps_end=min(eps, 0.001))
    return {"agents": [dict(ag, actions=11, states=50, action_range=[0.1, 0.3], min_memory=25),
                       dict(ag, actions=21, states=100, action_range=[0.15, 0.35], min_memory=25),
                       dict(ag, actions=5, states=20, action_range=[0.0, 0.3], min_memory=25, max_state=10)],
            "environment": dict(ENV, nplayers=3, max_steps=25, noise_prob=noise)}
def two(eps, noise):
    ag = dict(AG, epsilon=eps, eps_end=min(eps, 0.001))
    return {"agents": [dict(ag, actions=15, min_memory=100), dict(ag, action_range=[0.15, 0.45])], "environment": dict(ENV, noise_prob=noise)}
G, LAUNCHES, E = 65536, 8, 32
bad = 0
for name, mk in (("three players", three), ("two grids", two)):
    for eps in (0.5, 0.02, 0.0):
        for noise in (0.0, 0.05):
            cfg = mk(eps, noise)
            res = {}
            for kern in ("tuple", "generic"):
                gb = GameBatch(cfg, n_games=G, dtype="float32", kernel=kern, seed=11).init_tables()
                t0 = time.perf_counter()
                for _ in range(LAUNCHES):
                    gb.run(E, sync=False)
                torch.cuda.synchronize()
                res[kern] = (gb.q.clone(), gb.counter.clone(), gb.state.clone(), time.perf_counter() - t0, gb.last_kernel)
                del gb
            a, b = res["tuple"], res["generic"]
            same = bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]))
            bad += not same
            T = cfg["environment"]["max_steps"]
            print("%-14s eps %.2f noise %.2f: %s  (%s %.2f s, %s %.2f s; %d transitions per game)" %
                  (name, eps, noise, "bit-identical" if same else "MISMATCH", a[4], a[3], b[4], b[3], LAUNCHES * E * T), flush=True)
            del res
            torch.cuda.empty_cache()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
