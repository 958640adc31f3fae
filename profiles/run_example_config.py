import json, os, sys, time, tempfile
import numpy as np, pandas
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from th_rl_amd import trainer
cfg = {"agents": [{"name": "QTable", "gamma": 0.95, "actions": 21, "states": 100, "alpha": 0.1, "eps_end": 0.001,
                   "epsilon": 0.5, "eps_step": 0.9995, "action_range": [0.2, 0.4]},
                  {"name": "Reinforce", "gamma": 0.995, "actions": 21, "states": 1, "action_range": [0.2, 0.4]}],
       "environment": {"name": "NoisyPriceState", "noise_prob": 0, "a": 10, "b": 1, "nplayers": 2, "max_steps": 100},
       "training": {"print_freq": 5000, "epochs": 20000}}
for n_games in (1, 64):
    d = tempfile.mkdtemp()
    c = dict(cfg, training=dict(cfg["training"], n_games=n_games, seed=5))
    json.dump(c, open(os.path.join(d, "c.json"), "w"))
    t = time.time()
    trainer.train_one(os.path.join(d, "run"), os.path.join(d, "c.json"))
    el = time.time() - t
    a = pandas.read_csv(os.path.join(d, "run", "log.csv"), header=[0, 1]).to_numpy()
    print("n_games", n_games, "seconds %.1f" % el, "last1000", a[-1000:].mean(axis=0), "first1000", a[:1000].mean(axis=0), flush=True)
