#!/usr/bin/env python3
"""port / reference speed ratio per core, measured in the BUILD container:
the reference's own train_one (imported from /root/reference, which never travels to the GPU box)
and the C oracle, both on CFG, ONE game, float64, one core.  Writes profiles/ref_ratio.json, which
bench.py reports next to cpu_baseline so the GPU-box CPU figure can be read as reference-equivalent.

    python profiles/measure_ref_ratio.py [--epochs 1000]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

AG = dict(name="QTable", gamma=0.95, actions=21, states=100, alpha=0.1, eps_end=0.001, epsilon=0.5,
          eps_step=0.9995, action_range=[0.2, 0.4])
ENV = dict(name="NoisyPriceState", noise_prob=0, a=10, b=1, nplayers=2, max_steps=100)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=1000)
    a = ap.parse_args()
    import numpy as np
    from oracle import oracle as O
    cfgd = {"agents": [dict(AG), dict(AG)], "environment": dict(ENV),
            "training": {"epochs": a.epochs, "print_freq": 10 ** 9}}
    # the C restatement: one game, float64 (the reference's shape), Philox draws
    cfg, eps = O.cfg_from_config(cfgd, 1, 1)
    q, c, s = O.init(cfg, seed=0)
    O.episodes(cfg, q, c, s, eps, O.Memory(cfg), 10, seed=0)
    E = 20 * a.epochs
    t0 = time.perf_counter()
    O.episodes(cfg, q, c, s, eps, O.Memory(cfg), E, seed=0)
    port = E * 100 / (time.perf_counter() - t0)
    # the batch the GPU-box baseline uses: 1,024 games per thread, float32
    cfg, eps = O.cfg_from_config(cfgd, 1024, 0)
    q, c, s = O.init(cfg, seed=0)
    t0 = time.perf_counter()
    O.episodes(cfg, q, c, s, eps, O.Memory(cfg), 40, seed=0)
    port1024 = 1024 * 40 * 100 / (time.perf_counter() - t0)
    # the reference itself (single-threaded pure Python)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import random
    import torch
    from th_rl import trainer as RT
    np.random.seed(0); random.seed(0); torch.manual_seed(0); torch.set_num_threads(1)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "c.json")
        json.dump(cfgd, open(p, "w"))
        t0 = time.perf_counter()
        RT.train_one(os.path.join(d, "run"), p)
        ref = a.epochs * 100 / (time.perf_counter() - t0)
    out = dict(reference_env_steps_per_s_per_core=ref, port_env_steps_per_s_per_core_1game_f64=port,
               port_env_steps_per_s_per_core_1024games_f32=port1024,
               ref_ratio_1game=port / ref, ref_ratio=port1024 / ref, epochs=a.epochs,
               where="build container, one core of an 8-vCPU Intel Xeon @ 2.10 GHz",
               note="ref_ratio = oracle (bench.py's cpu_baseline shape: 1,024 games per thread, float32) / "
                    "reference train_one (one game, float64), env-steps/s on the same core")
    json.dump(out, open(os.path.join(ROOT, "profiles", "ref_ratio.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
