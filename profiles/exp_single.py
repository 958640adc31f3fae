import json, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from th_rl_amd import trainer
import bench
for kw in ({}, {"dtype": "float32"}, {"n_games": 64}):
    cfg = dict(bench.CFG, training=dict({"epochs": 10000, "print_freq": 10000, "seed": 1}, **kw))
    d = tempfile.mkdtemp(); json.dump(cfg, open(os.path.join(d, "c.json"), "w"))
    t = time.time(); trainer.train_one(os.path.join(d, "run"), os.path.join(d, "c.json")); print(kw, "10000 epochs: %.2f s" % (time.time() - t), flush=True)
