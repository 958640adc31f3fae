import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from th_rl_amd.mixed import MixedGameBatch
import bench
ag = {"name": "Reinforce", "gamma": 0.995, "actions": 21, "states": 1, "action_range": [0.2, 0.4]}
cfg = {"agents": [dict(bench.CFG["agents"][0]), ag], "environment": dict(bench.CFG["environment"])}
mb = MixedGameBatch(cfg, n_games=64, dtype="float32", seed=0).init_tables()
for phase, eps_run in (("early", 9), ("after 2000 episodes", 2000), ("after 10000 episodes", 8000)):
    mb.run(eps_run, per_game_logs=False)
    # the Reinforce ring holds the prices of the steps since its last update
    n = min(mb.count[1], mb.buf_len[1])
    pr = mb.buf[1]["price"][:, :n].t().cpu().numpy().astype(np.float32)       # rings are [G, buf_len]
    d, c32, c64 = [], [], []
    for g in range(64):
        v, cnt = np.unique(pr[:, g], return_counts=True)
        cnt = np.sort(cnt)[::-1]
        d.append(len(v)); c32.append(cnt[:32].sum() / n); c64.append(cnt[:64].sum() / n)
    print(phase, "n=%d distinct states per game: mean %.0f; top-32 cover %.2f, top-64 cover %.2f, eps=%.3f" % (n, np.mean(d), np.mean(c32), np.mean(c64), mb.eps[0]), flush=True)
