#!/usr/bin/env python3
"""Fixture G10: the run the reference SHIPS (th_rl/some_path/runs/example_config/0: a QTable vs
Reinforce game trained for 20,000 epochs) as data, plus what the reference's own utils do with it.
Usage: python tests/golden/make_golden_stored_run.py

Copies the run's data files (0.npy, 0_counter.npy, the torch state_dict `1`, config.json, the last
200 rows of log.csv) to tests/golden/ref_run_example_config/ and records, by running the REFERENCE
(utils.load_experiment + utils.play_game, utils.py:12-47), the greedy game those stored agents play
from a seeded initial state, and the log statistics of both shipped runs."""
import os
import random
import shutil
import sys

import numpy

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import pandas  # noqa: E402
import torch  # noqa: E402
import th_rl.utils as ref_utils  # noqa: E402


def main():
    src = os.path.join(REF, "th_rl", "some_path", "runs", "example_config", "0")
    dst = os.path.join(HERE, "ref_run_example_config")
    os.makedirs(dst, exist_ok=True)
    for f in ("0.npy", "0_counter.npy", "1", "config.json"):
        shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
        os.chmod(os.path.join(dst, f), 0o644)
    lines = open(os.path.join(src, "log.csv")).read().splitlines()
    open(os.path.join(dst, "log.csv"), "w").write("\n".join(lines[:2] + lines[-200:]) + "\n")

    out = {}
    config, agents, env, actions_ewm, rewards_ewm = ref_utils.load_experiment(src)
    for seed in (0, 1, 2):
        numpy.random.seed(seed); random.seed(seed); torch.manual_seed(seed)
        env.episode = 0
        acts, rews = ref_utils.play_game(agents, env, iters=1)
        # play_game calls environment.reset() itself: replay the same seed to learn the state it drew
        numpy.random.seed(seed)
        out["play%d_state0" % seed] = numpy.float64(numpy.random.uniform(0, env.a))
        out["play%d_actions" % seed] = acts
        out["play%d_rewards" % seed] = rews
    for run in ("0", "1 "):
        a = pandas.read_csv(os.path.join(REF, "th_rl", "some_path", "runs", "example_config", run, "log.csv"),
                            header=[0, 1]).to_numpy()
        tag = "shipped" + run.strip()
        out[tag + "_first1000"] = a[:1000].mean(axis=0)
        out[tag + "_last1000"] = a[-1000:].mean(axis=0)
    out["ewm_last_actions"] = actions_ewm.to_numpy()[-1]
    out["ewm_last_rewards"] = rewards_ewm.to_numpy()[-1]
    p = os.path.join(HERE, "g10_stored_run.npz")
    numpy.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), {k: numpy.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
