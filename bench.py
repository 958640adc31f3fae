#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused iterated-pricing-game kernel on MI355X.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric / configs[2]): 2 QTable agents
(example_config.json QTable block) x 2^20 parallel NoisyPriceState games PER GPU,
float32 tables, int32 visit counters, Philox draws, synthetic random-init tables.
One "step" = one episode (T=100 env-steps, both agents acting and learning) of every
game.  A kernel launch covers --chunk episodes (tables stay in LDS inside a launch; 32 at most);
by default the K steps of a timed region are ONE launch when K <= 32 (what a training loop does:
thrl_qtable_episodes cuts a training call into launches of 32 episodes), otherwise equal launches of <= 32.  The
timed region -- exactly K steps between barrier + synchronize on both sides, max over ranks -- is
repeated until >= 4 launches have been timed (K = 20: four regions) and the MEDIAN region is
reported; every region's time is in the line (`region_ms`), the per-launch HIP-event average of
the roofline block is over all of them.
Multi-GPU: games are seed-sharded (game_offset = rank * games), no data-path
collective; a gloo group provides only the barrier and the max-over-ranks clock.

Prints ONE JSON line (rank 0).  Blocks besides the driver contract's fields:
  roofline      bound "hbm" by the contract formula: achieved = 368 B x env-steps per
                launch / mean launch time (HIP events on the launch stream); traffic = HBM bytes per
                launch from the PMC passes in profiles/traffic.json (a per-game + per-episode model
                fitted to two launch sizes, so it is defined for any chunk); `issue` = the kernel's
                REAL bound -- instruction-issue / wait shares from the SQ counters in profiles/
  cpu_baseline  the C oracle (kind "port") on the host cores, bounded sample, + CPU model, core
                counts and the port/reference ratio measured in the build container
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
VALU_F32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: peak FP32 vector
NN_HIDDEN = 256


def _load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def cpu_info():
    """CPU model and core counts of this box (SURVEY 8d-ii asks for both next to the baseline)."""
    model, logical, cores = "unknown", os.cpu_count() or 1, set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    physical = len(cores) or logical
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = logical
    try:                                         # cgroup v2 CPU quota of the container, if any
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            usable = max(1, min(usable, int(int(q) / int(p))))
    except Exception:
        pass
    return dict(cpu_model=model, logical_cores=logical, physical_cores=physical, usable_cores=usable)


def cpu_baseline(seconds_target=10.0, max_threads=128):
    """The oracle (CPU port of the reference loop, float32 mode) timed on the host cores, on a
    bounded sample of the SAME workload (same config, Philox draws): one replica thread per
    physical core this process may use (ctypes releases the GIL), 1,024 games per thread --
    cache-resident like the reference's single game."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    info = cpu_info()
    # one replica per PHYSICAL core (hyper-threads share a core's ALUs), within the CPUs this process
    # may use (affinity mask / cgroup quota)
    cores = max(1, min(info["usable_cores"], info["physical_cores"], max_threads))
    games = 1024                    # per thread: 17 MB of tables
    O.lib()                         # build/load before timing

    def work(k, episodes):
        cfg, eps = O.cfg_from_config(CFG, games, 0)
        q, c, s = O.init(cfg, seed=0, game_offset=k * games)
        mem = O.Memory(cfg)
        t0 = time.perf_counter()
        O.episodes(cfg, q, c, s, eps, mem, episodes, seed=0, game_offset=k * games)
        return time.perf_counter() - t0

    t_probe = work(0, 4)
    E = max(4, int(4 * seconds_target / max(t_probe, 1e-3)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda k: work(k, E), range(cores)))
    wall = time.perf_counter() - t0
    steps = cores * games * E * T_STEPS
    out = dict(value=steps / wall, unit="env-steps/s", cores=cores, kind="port",
               sample="%d games x %d episodes x %d steps (%d threads x %d games), oracle float32 mode, "
                      "Philox draws, %.1f s" % (cores * games, E, T_STEPS, cores, games, wall),
               per_core=steps / wall / cores, single_thread=games * 4 * T_STEPS / t_probe, **info)
    rr = _load_json("ref_ratio.json")
    if rr:
        # the reference itself cannot travel; the ratio port/reference was measured on one core of the
        # build container (profiles/measure_ref_ratio.py), so value / ref_ratio reads as "the reference's
        # own Python loop on this many cores"
        out["ref_ratio"] = rr["ref_ratio"]
        out["reference_equivalent"] = out["value"] / rr["ref_ratio"]
        out["reference_env_steps_per_s_per_core_build_container"] = rr["reference_env_steps_per_s_per_core"]
    return out


def cpu_baseline_nn(kind, seconds_target=10.0):
    """oracle/nn_oracle.py (numpy float32 restatement of the neural agents) + the env formulas driving
    ONE game of the nn workload on one core: sample_action x 2, scale, env.step, append, train_net every
    1,000 transitions (agents.py:159-194).  Scalar port: cores = 1."""
    import numpy as np
    from oracle import nn_oracle as NN
    from oracle import oracle as O
    A, lo, hi, gamma = 21, 0.2, 0.4, 0.995
    rng = np.random.RandomState(0)
    n_nn = 2 if kind == "rr" else 1
    ws = [(rng.uniform(-0.06, 0.06, NN.n_params(A))).astype(np.float32) for _ in range(n_nn)]
    ms = [np.zeros_like(w) for w in ws]; vs = [np.zeros_like(w) for w in ws]; st = [0] * n_nn
    mem = [[] for _ in range(n_nn)]
    qcfg, _ = O.cfg_from_config({"agents": [dict(CFG_AGENT, states=1), dict(CFG_AGENT, states=1)],
                                 "environment": CFG["environment"]}, 1, 1)
    price, steps, t0 = 5.0, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_target:
        for _ in range(T_STEPS):
            acts = [int(NN.sample_action(ws[k], A, [price], [rng.uniform()])[0]) for k in range(n_nn)]
            sc = [NN.scale(a, A, lo, hi) for a in acts]
            if n_nn == 1:                       # a greedy-free stand-in for the tabular opponent's action
                sc = [lo + (hi - lo) * rng.randint(A) / (A - 1.0)] + sc
            nprice, rew = O.env_step(qcfg, sc)
            for k in range(n_nn):
                mem[k].append((price, acts[k], rew[k - n_nn]))
            price = nprice
            steps += 1
        for k in range(n_nn):
            if len(mem[k]) >= 1000:
                pr, ac, rw = zip(*mem[k])
                ws[k], ms[k], vs[k], st[k], _ = NN.train_net(ws[k], ms[k], vs[k], st[k], A, pr, ac, rw, gamma, 0.0)
                mem[k] = []
    wall = time.perf_counter() - t0
    return dict(value=steps / wall, unit="env-steps/s", cores=1, kind="port",
                sample="1 game x %d steps, %d network updates, oracle/nn_oracle.py (numpy float32), %.1f s"
                       % (steps, sum(st), wall), **cpu_info())


def bench_nn(args):
    """BASELINE configs[3]: 2 Reinforce agents (agents.py:119-220) x 65,536 games through
    mixed.MixedGameBatch (--nn-loop fused: thrl_mixed_episodes + the batched update kernels;
    unfused: one launch per reference call).  --nn-agents qr = the reference's shipped pairing,
    QTable vs Reinforce.  steps = episodes; the policy trains every 10 episodes."""
    cpu = None if args.no_cpu_baseline else cpu_baseline_nn(args.nn_agents, args.cpu_seconds)
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
    mb.run(args.warmup, fused=fused, per_game_logs=False) if fused else mb.run(args.warmup, fused=False)
    torch.cuda.synchronize()
    upd0 = {i: rb.step for i, rb in mb.nn.items()}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    mb.run(args.steps, fused=fused, per_game_logs=False) if fused else mb.run(args.steps, fused=False)
    ev1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    who = {"qr": "QTable vs Reinforce", "rr": "2-agent Reinforce", "qq": "2-agent QTable (mixed kernel)",
           "qa": "QTable vs ActorCritic", "qc": "QTable vs CAC"}[args.nn_agents]
    n_nets = sum(1 for k in mb.kinds if k in ("Reinforce", "ActorCritic"))
    updates = sum(rb.step - upd0[i] for i, rb in mb.nn.items())
    env_steps = float(G) * T_STEPS * args.steps
    # Algorithmic work of the neural part (discrete policies 1 -> 256 -> 21, MFMA off): one policy
    # evaluation per network per env-step = 256 + 256*21 FMAs; one update per network = forward + backward
    # (3 x the forward FMAs) over the 1,000 replayed transitions, plus the parameter / Adam-state traffic.
    fwd_fma = NN_HIDDEN + NN_HIDDEN * 21
    n_tr = 1000
    flops = 2.0 * fwd_fma * (n_nets * env_steps + 3.0 * n_tr * updates * G)
    params = 2 * NN_HIDDEN + 21 * NN_HIDDEN + 21
    upd_bytes = float(updates) * G * params * 4 * 6          # params + Adam m, v: read and written once per update
    gpu_s = ev0.elapsed_time(ev1) * 1e-3
    out = {"metric": "env-steps/sec, %s (neural policy) x %d games" % (who, G),
           "value": env_steps / dt, "unit": "env-steps/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "dtype": "f32", "data": "synthetic", "vs_baseline": None,
           "config": {"workload": "%s x %d games, %s loop, MFMA off" % (who, G, args.nn_loop),
                      "network_updates": updates},
           "roofline": {"bound": "valu_f32", "kernel": "k_mixed_wave + k_nn_*_train (whole step)",
                        "achieved": flops / gpu_s / 1e12, "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / gpu_s / 1e12 / VALU_F32_PEAK_TFLOPS, "traffic": upd_bytes / max(updates, 1) if updates else None,
                        "algorithmic_flops": flops, "update_bytes_per_launch_algorithmic": upd_bytes / max(updates, 1),
                        "gpu_time_ms": gpu_s * 1e3,
                        "note": "algorithmic FLOPs (every step evaluates the policy, every update runs forward + "
                                "backward over all 1,000 transitions) / GPU time of the timed region (HIP events); "
                                "the kernels skip work the count includes (policy memo, state folding), so this is a "
                                "rate of useful work, not of executed FLOPs; `traffic` is the algorithmic parameter + "
                                "Adam-state bytes of one update launch (not a PMC reading; PMC: profiles/)"}}
    if cpu is not None:
        out["cpu_baseline"] = cpu
    print(json.dumps(out))


def visible_gpus():
    """GPUs this process could use, WITHOUT initialising the HIP runtime (the launcher must not: its children
    are the ranks).  torch.cuda.device_count() only enumerates on this image."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def self_launch(n, argv, allow_oversubscribe=False, popen=None):
    """`python bench.py --gpus N` without a torch.distributed launcher: spawn N child ranks of this script (one
    per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set),
    relay rank 0's JSON line, return the worst exit code.  The parent never touches the GPU and reports
    nothing of its own -- a line with n_gpus = N only ever comes from N ranks that ran.  Refuses (exit 2) when
    the box shows fewer than N GPUs, unless --allow-oversubscribe (rehearsal: ranks share devices modulo the
    count and the line says so)."""
    import socket
    import subprocess
    have = visible_gpus()
    if have < n and not allow_oversubscribe:
        print("bench.py: --gpus %d but only %d GPU(s) visible: refusing to report an %d-GPU number "
              "(--allow-oversubscribe shares devices for a rehearsal)" % (n, have, n), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    popen = popen or subprocess.Popen
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                           stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0]
    rcs = [p.wait() for p in procs]
    if any(rcs):
        print("bench.py: child ranks exited with %s" % rcs, file=sys.stderr)
        return max(abs(rc) for rc in rcs) or 1
    sys.stdout.write(out0.decode() if isinstance(out0, bytes) else (out0 or ""))
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--games", type=int, default=1 << 20, help="games per GPU")
    ap.add_argument("--chunk", type=int, default=0,
                    help="episodes per kernel launch (<=32); default: the whole region when steps <= 32, else equal "
                         "launches of <= 32 episodes")
    ap.add_argument("--regions", type=int, default=0,
                    help="how many times the timed K-step region is run (median reported); default: enough for >= 4 "
                         "timed launches")
    ap.add_argument("--nn-loop", default="fused", choices=["fused", "unfused"])
    ap.add_argument("--nn-agents", default="rr", choices=["rr", "qr", "qq", "qa", "qc"])
    ap.add_argument("--kernel", default="wave", choices=["wave", "generic", "auto"])
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"],
                    help="table dtype: float32 = the metric's; float64 = the reference's own numerics (diagnostic)")
    ap.add_argument("--workload", default="qtable", choices=["qtable", "nn"],
                    help="qtable = the headline metric (default); nn = BASELINE configs[3]: 2 Reinforce "
                         "agents x 65,536 games through the fused episode + update kernels (secondary)")
    ap.add_argument("--allow-oversubscribe", action="store_true",
                    help="rehearsal only: let --gpus N run on fewer than N GPUs (ranks share devices; the line is "
                         "flagged `oversubscribed`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counters", action="store_true",
                    help="diagnostic only: run without QTable.counter (NOT the reported workload)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--noise-prob", type=float, default=None,
                    help="diagnostic: NoisyPriceState noise_prob (config value 0; class default 0.05)")
    ap.add_argument("--max-steps", type=int, default=None,
                    help="diagnostic: environment.max_steps (config value 100); with the QTable default min_memory = 100 "
                         "a smaller value makes the replay buffer span episodes (training cycles)")
    ap.add_argument("--capacity", type=int, default=None, help="diagnostic: QTable.capacity (default 500)")
    ap.add_argument("--epsilon", type=float, default=None,
                    help="diagnostic: start from this epsilon instead of the config's 0.5 "
                         "(e.g. 0.001 = the late-training, greedy-dominated regime)")
    ap.add_argument("--pretrain", type=int, default=0,
                    help="diagnostic: train this many episodes (untimed, natural epsilon decay) before the warm-up, "
                         "so the timed region runs on TRAINED tables at the epsilon a real run has by then "
                         "(the reference's config trains 20,000 episodes; epsilon is 0.004 after 10,000)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- N child ranks, nothing touches the GPU here
        sys.exit(self_launch(args.gpus, sys.argv[1:], args.allow_oversubscribe))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the GPU count of the line is the number of ranks that RUN, never the flag
    n_gpus = world
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: reporting n_gpus=%d" % (args.gpus, world, world), file=sys.stderr)

    if args.workload == "nn":
        if world > 1:
            print("bench.py: --workload nn is a single-GPU line", file=sys.stderr)
            sys.exit(2)
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
    n_dev = max(1, torch.cuda.device_count())
    if world > n_dev and not args.allow_oversubscribe:
        print("bench.py: %d ranks but %d GPU(s) visible (--allow-oversubscribe for a rehearsal)" % (world, n_dev), file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", (local_rank % n_dev) if world > 1 else 0)
    torch.cuda.set_device(dev)

    G = args.games
    if args.chunk > 0:
        chunk = max(1, min(32, args.chunk))
    else:
        n_launch = (args.steps + 31) // 32
        chunk = max(1, (args.steps + n_launch - 1) // n_launch)
    launches_per_region = (args.steps + chunk - 1) // chunk
    regions = args.regions if args.regions > 0 else max(1, (4 + launches_per_region - 1) // launches_per_region)
    if args.noise_prob is not None:
        CFG["environment"]["noise_prob"] = float(args.noise_prob)
    T_run = T_STEPS
    if args.max_steps is not None:
        CFG["environment"]["max_steps"] = T_run = int(args.max_steps)
    if args.capacity is not None:
        for ag in CFG["agents"]:
            ag["capacity"] = int(args.capacity)
    gb = GameBatch(CFG, n_games=G, device=dev, dtype=args.dtype, kernel=args.kernel, seed=0,
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

    if args.pretrain > 0:
        done = 0
        while done < args.pretrain:
            e = min(32, args.pretrain - done)
            gb.run(e, sync=False)
            done += e
        torch.cuda.synchronize(dev)
    eps_at_start = float(gb.eps[0])
    run_steps(args.warmup)
    barrier()
    events = []
    region_s, region_own = [], []
    for _ in range(regions):
        t0 = time.perf_counter()
        run_steps(args.steps, events)
        torch.cuda.synchronize(dev)
        own = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        per = [own]
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt[0])
            allr = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(allr, torch.tensor([own], dtype=torch.float64))
            per = [float(x[0]) for x in allr]
        region_s.append(el)
        region_own.append(per)
        barrier()
    mid = sorted(range(regions), key=lambda i: region_s[i])[regions // 2]      # the median region (of an even count: the slower middle one)
    elapsed, per_rank = region_s[mid], region_own[mid]

    if rank == 0:
        total_env_steps = float(n_gpus) * G * T_run * args.steps
        value = total_env_steps / elapsed
        # dominant kernel: k_wave_episodes; HIP events on the launch stream bracket each launch
        full = [(a.elapsed_time(b) * 1e-3, e) for a, b, e in events if e == chunk] or \
               [(a.elapsed_time(b) * 1e-3, e) for a, b, e in events]
        times = sorted(t for t, _ in full)
        avg_launch_s = sum(times) / len(times)
        e_launch = full[0][1]
        algo_bytes_launch = ALGO_BYTES_PER_ENV_STEP * G * T_run * e_launch
        achieved = algo_bytes_launch / avg_launch_s / 1e9
        # HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE in separate passes, gfx950
        # corrections: profiles/summarize.py): bytes = G * (per_game + per_game_episode * E)
        traffic, issue = None, None
        tj = _load_json("traffic.json")
        default_cfg = (args.noise_prob in (None, 0.0) and args.epsilon is None and not args.no_counters
                       and args.dtype == "float32" and args.max_steps is None and args.capacity is None
                       and args.pretrain == 0)
        if tj and tj.get("kernel") == gb.last_kernel and default_cfg and "model" in tj:
            m = tj["model"]
            traffic = float(G) * (m["bytes_per_game_per_launch"] + m["bytes_per_game_per_episode"] * e_launch)
            issue = dict(tj.get("issue") or {})
            sc = issue.pop("occupancy_scaling", None)           # keep the line compact: only the fitted share
            if sc:
                issue["latency_share_at_20_waves"] = sc.get("latency_share_at_20_waves")
            issue = issue or None
        out = {
            "metric": "env-steps/sec, 2-agent PD x 1M parallel games",
            "value": value, "unit": "env-steps/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "regions": regions, "region_ms": [t * 1e3 for t in region_s],
            "region_policy": "each region = exactly `steps` steps between barrier + synchronize (max over ranks); "
                             "the median region is reported (of an even count the slower middle one)",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "float32" else "f64", "data": "synthetic",
            "config": {"workload": "2-agent QTable (21 actions x 101 states, example_config.json) x %d "
                                   "parallel NoisyPriceState games per GPU, T=%d, fused step+TD kernel"
                                   % (G, T_run),
                       "games_per_gpu": G, "episodes_per_launch": e_launch, "kernel": gb.last_kernel,
                       "counters": not args.no_counters, "epsilon_start": eps_at_start, "pretrain_episodes": args.pretrain,
                       "noise_prob": CFG["environment"]["noise_prob"], "parallelism": "seed-sharded x%d, no collective" % n_gpus,
                       "devices_visible": n_dev, "oversubscribed": world > n_dev},
            "roofline": {"bound": "hbm", "kernel": "k_wave_episodes" if gb.last_kernel == "wave" else "k_generic_episodes",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes_launch,
                         # what the HBM actually moved per second (PMC traffic / launch time): tables stay
                         # in LDS for a whole launch, so it is far below `achieved` -- the contract fraction
                         # measures algorithmic work, and exceeds 1 once LDS reuse beats the no-reuse bound
                         "traffic_gbps": None if traffic is None else traffic / avg_launch_s / 1e9,
                         "measured_hbm_frac": None if traffic is None else traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS,
                         "avg_launch_ms": avg_launch_s * 1e3, "median_launch_ms": times[len(times) // 2] * 1e3,
                         "launches_timed": len(full),
                         # the kernel's real bound (SQ counters, profiles/): share of the SIMDs' VALU issue
                         # cycles used and the split of wave time into issuing / stalled / waiting
                         "issue": issue},
        }
        if world > 1:
            # N = 1-equivalent figures so a SCALE record can be checked against BENCH directly
            out["per_rank"] = {"games_per_gpu": G, "seconds": per_rank,
                               "value": [G * T_run * args.steps / s for s in per_rank]}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
