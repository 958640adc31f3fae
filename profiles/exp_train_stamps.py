"""Phase timestamps of k_nn_reinforce_train (DIAGNOSTIC library build/libthrl_stamp.so: the kernel writes s_memtime at phase
boundaries into the grad_out buffer; results of that build are not used for anything else).
    THRL_LIB=build/libthrl_stamp.so python profiles/exp_train_stamps.py [rr|qr]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from th_rl_amd.mixed import MixedGameBatch
which = sys.argv[1] if len(sys.argv) > 1 else "rr"
ag = {"name": "Reinforce", "gamma": 0.995, "actions": 21, "states": 1, "action_range": [0.2, 0.4]}
first = dict(ag) if which == "rr" else dict(bench.CFG["agents"][0])
cfg = {"agents": [first, dict(ag)], "environment": dict(bench.CFG["environment"], noise_prob=0)}
G = 65536
mb = MixedGameBatch(cfg, n_games=G, dtype="float32", seed=0).init_tables()
mb.run(9, per_game_logs=False)                     # 900 transitions buffered; the next episode triggers the update
# run one more episode by hand so that the update can be called with want_grad (the stamp buffer)
import types
stamps = {}
def wrap(rb):                       # every network's update gets a grad_out buffer: the diagnostic build stamps into it
    orig = rb.train
    def train(price, action, reward, want_grad=False, next_price=None, rows=False):
        g = orig(price, action, reward, want_grad=True, next_price=next_price, rows=rows)
        stamps["g"] = g
        return g
    rb.train = train
for rb in mb.nn.values():
    wrap(rb)
mb.run(1, per_game_logs=False)
torch.cuda.synchronize()
raw = stamps["g"].cpu().numpy().reshape(-1).view(np.uint64)[:G * 16].reshape(G, 16).astype(np.float64)   # 16 accumulated phase times per block
names = ["loads + z-score", "hash insert", "rank / sort", "thresholds", "chunk: zero", "chunk: scatter + returns", "chunk: prefix A/B",
         "chunk: softmax", "chunk: prefix d", "chunk: gather", "finish gradient", "adam prefetch", "norm + adam sweep"]
tot = raw[:, :13].sum(axis=1)
print(which, "median cycles per block (s_memtime ticks, 100 MHz):", np.median(tot), "= %.1f us" % (np.median(tot) / 100.0))
for i, n in enumerate(names):
    print("  %-26s median %8.0f ticks  %5.1f %%" % (n, np.median(raw[:, i]), 100 * np.median(raw[:, i]) / np.median(tot)))
