#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused iterated-pricing-game kernel on MI355X.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric / configs[2]): 2 QTable agents
(example_config.json QTable block) x 2^20 parallel NoisyPriceState games PER GPU,
float32 tables, int32 visit counters, Philox draws, synthetic random-init tables.
One "step" = one episode (T=100 env-steps, both agents acting and learning) of every
game; kernel launches cover --chunk episodes each (tables stay in LDS inside a launch).
Multi-GPU: games are seed-sharded (game_offset = rank * games), no data-path
collective; a gloo group provides only the barrier and the max-over-ranks clock.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG_AGENT = dict(name="QTable", gamma=0.95, actions=21, states=100, alpha=0.1, eps_end=0.001,
                 epsilon=0.5, eps_step=0.9995, action_range=[0.2, 0.4])
CFG = {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)],
       "environment": dict(name="NoisyPriceState", noise_prob=0, a=10, b=1, nplayers=2, max_steps=100)}
T_STEPS = 100
ALGO_BYTES_PER_ENV_STEP = 368.0     # 2 agents x (2 x 21 x 4 + 16) B
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(seconds_target=10.0):
    """The oracle (CPU port of the reference loop, float32 mode) timed on the host
    cores, on a bounded sample of the SAME workload (same config, Philox draws)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    cores = min(os.cpu_count() or 1, 16)
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    games = 1024                    # per thread: 17 MB of tables, cache-friendly like the reference's 1 game
    O.lib()                         # build/load before timing

    def work(k, episodes):
        cfg, eps = O.cfg_from_config(CFG, games, 0)
        q, c, s = O.init(cfg, seed=0, game_offset=k * games)
        mem = O.Memory(cfg)
        t0 = time.perf_counter()
        O.episodes(cfg, q, c, s, eps, mem, episodes, seed=0, game_offset=k * games)   # ctypes releases the GIL
        return time.perf_counter() - t0

    t_probe = work(0, 4)
    E = max(4, int(4 * seconds_target / max(t_probe, 1e-3)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda k: work(k, E), range(cores)))
    wall = time.perf_counter() - t0
    steps = cores * games * E * T_STEPS
    return dict(value=steps / wall, unit="env-steps/s", cores=cores, kind="port",
                sample="%d games x %d episodes x %d steps (%d threads x %d games), oracle float32 mode, "
                       "Philox draws, %.1f s" % (cores * games, E, T_STEPS, cores, games, wall))


def bench_nn(args):
    """BASELINE configs[3]: 2 Reinforce agents (agents.py:119-220) x 65,536 games through
    mixed.MixedGameBatch (--nn-loop fused: thrl_mixed_episodes; unfused: one launch per reference
    call).  --nn-agents qr = the reference's shipped pairing, QTable vs Reinforce.
    steps = episodes; the policy trains every 10 episodes."""
    import torch
    from th_rl_amd.mixed import MixedGameBatch
    G = args.games if args.games != (1 << 20) else 65536
    ag = {"name": "Reinforce", "gamma": 0.995, "actions": 21, "states": 1, "action_range": [0.2, 0.4]}
    first = dict(ag) if args.nn_agents == "rr" else dict(CFG["agents"][0])
    second = {"qq": dict(CFG["agents"][1]), "qa": dict(ag, name="ActorCritic", gamma=0.98),
              "qc": {"name": "CAC", "gamma": 0.98, "states": 1, "action_range": [0.2, 0.4]}}.get(args.nn_agents, dict(ag))
    config = {"agents": [first, second], "environment": dict(CFG["environment"])}
    fused = args.nn_loop == "fused"
    mb = MixedGameBatch(config, n_games=G, dtype="float32", seed=0).init_tables()
    mb.run(args.warmup, fused=fused)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mb.run(args.steps, fused=fused)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    who = {"qr": "QTable vs Reinforce", "rr": "2-agent Reinforce", "qq": "2-agent QTable (mixed kernel)",
           "qa": "QTable vs ActorCritic", "qc": "QTable vs CAC"}[args.nn_agents]
    print(json.dumps({"metric": "env-steps/sec, %s (neural policy) x %d games" % (who, G),
                      "value": G * T_STEPS * args.steps / dt, "unit": "env-steps/s", "n_gpus": 1, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                      "dtype": "f32", "data": "synthetic", "vs_baseline": None,
                      "config": {"workload": "%s x %d games, %s loop, MFMA off" % (who, G, args.nn_loop),
                                 "network_updates": mb.nn[1].step if 1 in mb.nn else 0}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--games", type=int, default=1 << 20, help="games per GPU")
    ap.add_argument("--chunk", type=int, default=25, help="episodes per kernel launch (<=32)")
    ap.add_argument("--nn-loop", default="fused", choices=["fused", "unfused"])
    ap.add_argument("--nn-agents", default="rr", choices=["rr", "qr", "qq", "qa", "qc"])
    ap.add_argument("--kernel", default="wave", choices=["wave", "generic", "auto"])
    ap.add_argument("--workload", default="qtable", choices=["qtable", "nn"],
                    help="qtable = the headline metric (default); nn = BASELINE configs[3]: 2 Reinforce "
                         "agents x 65,536 games through the unfused operator loop (secondary)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counters", action="store_true",
                    help="diagnostic only: run without QTable.counter (NOT the reported workload)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--noise-prob", type=float, default=None,
                    help="diagnostic: NoisyPriceState noise_prob (config value 0; class default 0.05)")
    ap.add_argument("--epsilon", type=float, default=None,
                    help="diagnostic: start from this epsilon instead of the config's 0.5 "
                         "(e.g. 0.001 = the late-training, greedy-dominated regime)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    if world != n_gpus and world > 1:
        n_gpus = world

    if args.workload == "nn":
        return bench_nn(args)

    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_seconds)          # before the GPU is touched

    import torch
    import torch.distributed as dist
    from th_rl_amd.batched import GameBatch

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    # one rank per GPU; the modulo only matters when rehearsing N ranks on a box with fewer GPUs
    dev = torch.device("cuda", (local_rank % max(1, torch.cuda.device_count())) if world > 1 else 0)
    torch.cuda.set_device(dev)

    G = args.games
    chunk = max(1, min(32, args.chunk))
    if args.noise_prob is not None:
        CFG["environment"]["noise_prob"] = float(args.noise_prob)
    gb = GameBatch(CFG, n_games=G, device=dev, dtype="float32", kernel=args.kernel, seed=0,
                   game_offset=rank * G, counters=not args.no_counters).init_tables()

    if args.epsilon is not None:
        gb.eps = [float(args.epsilon)] * len(gb.eps)

    def run_steps(n, events=None):
        done = 0
        while done < n:
            e = min(chunk, n - done)
            if events is not None:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            gb.run(e, sync=False)
            if events is not None:
                b.record()
                events.append((a, b, e))
            done += e

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    run_steps(args.warmup)
    barrier()
    events = []
    t0 = time.perf_counter()
    run_steps(args.steps, events)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])

    if rank == 0:
        total_env_steps = float(n_gpus) * G * T_STEPS * args.steps
        value = total_env_steps / elapsed
        # dominant kernel: k_wave_episodes; HIP events on the launch stream bracket each launch
        full = [(a.elapsed_time(b) * 1e-3, e) for a, b, e in events if e == chunk] or \
               [(a.elapsed_time(b) * 1e-3, e) for a, b, e in events]
        avg_launch_s = sum(t for t, _ in full) / len(full)
        e_launch = full[0][1]
        algo_bytes_launch = ALGO_BYTES_PER_ENV_STEP * G * T_STEPS * e_launch
        achieved = algo_bytes_launch / avg_launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("games") == G and tj.get("episodes_per_launch") == e_launch and tj.get("kernel") == gb.last_kernel:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec, 2-agent PD x 1M parallel games",
            "value": value, "unit": "env-steps/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "2-agent QTable (21 actions x 101 states, example_config.json) x %d "
                                   "parallel NoisyPriceState games per GPU, T=100, fused step+TD kernel"
                                   % G,
                       "games_per_gpu": G, "episodes_per_launch": e_launch, "kernel": gb.last_kernel,
                       "counters": not args.no_counters, "epsilon_start": 0.5 if args.epsilon is None else args.epsilon,
                       "noise_prob": CFG["environment"]["noise_prob"], "parallelism": "seed-sharded x%d, no collective" % n_gpus},
            "roofline": {"bound": "hbm", "kernel": "k_wave_episodes" if gb.last_kernel == "wave" else "k_generic_episodes",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes_launch,
                         # what the HBM actually moved per second (PMC traffic / launch time): tables
                         # stay in LDS for a whole launch, so it is far below `achieved`
                         "traffic_gbps": None if traffic is None else traffic / avg_launch_s / 1e9,
                         "avg_launch_ms": avg_launch_s * 1e3, "launches_timed": len(full)},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
