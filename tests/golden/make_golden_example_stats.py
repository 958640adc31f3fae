#!/usr/bin/env python3
"""Fixture G11: learning statistics of the REFERENCE on its own example config (QTable vs Reinforce,
20,000 epochs x 100 steps): six seeded runs of th_rl.trainer.train_one executed here (one CPU core
each, ~750 s per run), reduced to the mean reward / scaled action of both agents over the first and
the last 1,000 epochs, plus the wall time.  Data only.
Usage: python tests/golden/make_golden_example_stats.py [--keep DIR]"""
import multiprocessing
import os
import random
import sys
import tempfile
import time

import numpy

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
CONFIG = os.path.join(REF, "th_rl", "some_path", "configs", "example_config.json")
SEEDS = (1, 2, 3, 4, 5, 6)


def one(args):
    seed, out = args
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import pandas
    import torch
    torch.set_num_threads(1)
    numpy.random.seed(seed); random.seed(seed); torch.manual_seed(seed)
    from th_rl.trainer import train_one
    if not os.path.exists(os.path.join(out, "log.csv")):
        t = time.time()
        train_one(out, CONFIG)
        open(os.path.join(out, "seconds.txt"), "w").write("%.1f" % (time.time() - t))
    sec = float(open(os.path.join(out, "seconds.txt")).read())
    a = pandas.read_csv(os.path.join(out, "log.csv"), header=[0, 1]).to_numpy()
    return seed, a[:1000].mean(axis=0), a[-1000:].mean(axis=0), sec


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else tempfile.mkdtemp()
    jobs = [(s, os.path.join(keep, "run%d" % s)) for s in SEEDS]
    with multiprocessing.get_context("spawn").Pool(len(SEEDS)) as pool:
        res = sorted(pool.map(one, jobs))
    out = dict(seeds=numpy.array([r[0] for r in res]), first1000=numpy.stack([r[1] for r in res]),
               last1000=numpy.stack([r[2] for r in res]), seconds=numpy.array([r[3] for r in res]),
               columns=numpy.array(["reward0", "reward1", "action0", "action1"]), epochs=numpy.array(20000))
    p = os.path.join(HERE, "g11_example_config_stats.npz")
    numpy.savez_compressed(p, **out)
    print("wrote", p, out["first1000"].mean(axis=0), out["last1000"].mean(axis=0), out["seconds"])


if __name__ == "__main__":
    main()
