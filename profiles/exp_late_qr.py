"""QTable vs Reinforce at 65,536 games: ms per episode early and after 6,000 episodes of training
(policies concentrated), with the hit-rate-gated policy memo on / off (THRL_MIXED_NO_GATED_MEMO)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from th_rl_amd.mixed import MixedGameBatch
import bench
ag = {"name": "Reinforce", "gamma": 0.995, "actions": 21, "states": 1, "action_range": [0.2, 0.4]}
cfg = {"agents": [dict(bench.CFG["agents"][0]), ag], "environment": dict(bench.CFG["environment"])}
mb = MixedGameBatch(cfg, n_games=65536, dtype="float32", seed=0).init_tables()
mb.run(10, per_game_logs=False)
for phase, train in (("early", 0), ("after 6000 episodes", 5970)):
    if train:
        mb.run(train, per_game_logs=False)
    torch.cuda.synchronize(); t = time.perf_counter()
    mb.run(20, per_game_logs=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print("%s: %.2f ms per episode (eps %.3f) gated_memo=%s" % (phase, dt * 1e3, mb.eps[0], os.environ.get("THRL_MIXED_NO_GATED_MEMO") is None), flush=True)
