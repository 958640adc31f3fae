#!/usr/bin/env python3
"""The WHOLE training run the reference's config describes (training.epochs = 20,000, example_config.json:36-39)
for 2 x QTable on 1,048,576 parallel games, timed in slices of 1,000 episodes: throughput as epsilon decays from
0.5 to 0.001 and the games settle into fixed points and short cycles (more serial replay passes per episode).

    python3 profiles/full_run_r02.py [--games 1048576] [--episodes 20000] > gpurun_out/full_run.json
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from th_rl_amd.batched import GameBatch

AG = dict(name="QTable", gamma=0.95, actions=21, states=100, alpha=0.1, eps_end=0.001, epsilon=0.5, eps_step=0.9995,
          action_range=[0.2, 0.4])
CFG = {"agents": [dict(AG), dict(AG)],
       "environment": dict(name="NoisyPriceState", noise_prob=0, a=10, b=1, nplayers=2, max_steps=100)}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1 << 20)
    ap.add_argument("--episodes", type=int, default=20000)
    ap.add_argument("--slice", type=int, default=1000)
    ap.add_argument("--chunk", type=int, default=25)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    gb = GameBatch(CFG, n_games=a.games, device=dev, dtype="float32", kernel="wave", seed=0).init_tables()
    gb.run(a.chunk, sync=True)                       # warm-up launch (not part of the run below: tables re-initialised)
    gb = GameBatch(CFG, n_games=a.games, device=dev, dtype="float32", kernel="wave", seed=0).init_tables()
    torch.cuda.synchronize(dev)
    rows, t_all = [], time.perf_counter()
    done = 0
    while done < a.episodes:
        n = min(a.slice, a.episodes - done)
        eps0 = float(gb.eps[0])
        t0 = time.perf_counter()
        k, rew = 0, None
        while k < n:
            e = min(a.chunk, n - k)
            out = gb.run(e, sync=False)
            k += e
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        done += n
        rl = out["reward_log"]                      # [e][2] mean over games, last launch of the slice
        rows.append({"episodes": done, "epsilon_at_slice_start": eps0, "seconds": dt,
                     "env_steps_per_s": a.games * 100.0 * n / dt,
                     "mean_reward_last_episode": [float(rl[-1][0]), float(rl[-1][1])]})
        print("# %6d eps %.4f  %.3e env-steps/s  reward %s" % (done, eps0, rows[-1]["env_steps_per_s"],
              rows[-1]["mean_reward_last_episode"]), file=sys.stderr, flush=True)
    total = time.perf_counter() - t_all
    print(json.dumps({"workload": "2 x QTable (example_config.json parameters) x %d games, %d episodes x 100 steps" % (a.games, a.episodes),
                      "kernel": gb.last_kernel, "episodes_per_launch": a.chunk, "seconds": total,
                      "env_steps": a.games * 100.0 * a.episodes, "env_steps_per_s": a.games * 100.0 * a.episodes / total,
                      "nash_reward_per_agent": 100.0 / 9.0, "cartel_reward_per_agent": 12.5, "slices": rows}, indent=1))
