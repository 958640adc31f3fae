"""The reference's example config (QTable vs Reinforce) at 65,536 games for 2,000 episodes:
throughput and the learning statistics of the first 1,000 episodes next to the reference's runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from th_rl_amd.mixed import MixedGameBatch
import bench
ag = {"name": "Reinforce", "gamma": 0.995, "actions": 21, "states": 1, "action_range": [0.2, 0.4]}
cfg = {"agents": [dict(bench.CFG["agents"][0]), ag], "environment": dict(bench.CFG["environment"])}
G = 65536
mb = MixedGameBatch(cfg, n_games=G, dtype="float32", seed=3).init_tables()
t = time.time(); out = mb.run(2000, per_game_logs=False); torch.cuda.synchronize(); dt = time.time() - t
r, a = out["reward_log"], out["action_log"]
print("2000 episodes x %d games: %.1f s = %.3e env-steps/s; network updates %d" % (G, dt, G * 100 * 2000 / dt, mb.nn[1].step))
print("first 1000 episodes: reward %s action %s   (reference, 8 runs: 11.99 / 10.82, 0.327 / 0.298)" % (r[:1000].mean(axis=0), a[:1000].mean(axis=0)))
print("episodes 1000-2000: reward %s action %s; finite %s" % (r[1000:].mean(axis=0), a[1000:].mean(axis=0), bool(torch.isfinite(mb.nn[1].params).all().item())))
