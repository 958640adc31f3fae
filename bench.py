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
  roofline      the kernel's BINDING bound (round-3: vector-instruction issue), frac <= 1, recomputable from
                profiles/: achieved = SQ_INSTS_VALU per launch (profiles/traffic.json: instructions per env-step from
                the committed PMC summary x the env-steps of this launch) / live launch time (HIP events on the launch
                stream); peak = 1,024 SIMDs x shader clock / measured issue cost of this kernel's instruction mix
                (profiles/r03_ubench_issue.md priced over profiles/isa_mix.json).  Beside it: `algorithmic_equiv_gbps`
                (the contract's 368 B per env-step / launch time: above the 8 TB/s peak because tables stay in LDS for
                a launch, so it carries no frac), `traffic` (HBM bytes per launch, PMC model) and `measured_hbm_frac`.
                `stale` is true when the loaded library is not the binary the counters were collected on.
  cpu_baseline  the C oracle (kind "port") on the host cores, bounded sample, + CPU model, core
                counts and the port/reference ratio measured in the build container
  secondary     BASELINE configs[3] witnessed by the same run: 2 x Reinforce and QTable vs Reinforce x 65,536 games
                (full sub-lines with their own roofline / cpu_baseline); --no-secondary skips them
  library       which libthrl_hip.so ran (path, source hashes, ablation mask); an ablation build is refused
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
N_SIMD, N_CU = 1024, 256            # MI355X: 256 CUs x 4 SIMD-32
KEY_WAVE = "k_wave_episodes<float,2,1> (headline)"
KEY_MIXED = {"rr": "k_ptuple_episodes<float,NR=2,24,2,lds> (2 x Reinforce)", "qr": "k_ptuple_episodes<float,NR=1,24,2,hbm> (QTable vs Reinforce)",
             "qa": "k_ptuple_episodes<float,NR=1,24,2,hbm> (QTable vs Reinforce)"}


def library_info():
    """Which binary runs (thrl_build_info): an ablation build (phases compiled out, results wrong by construction)
    must never produce a reported number."""
    from th_rl_amd import _lib
    info = _lib.build_info()
    if info["ablate"] != 0:
        print("bench.py: %s is a TIMING-ONLY ablation build (THRL_ABLATE mask %d): refusing to report numbers from it"
              % (info["path"], info["ablate"]), file=sys.stderr)
        sys.exit(2)
    return info


def issue_price(kernel_key):
    """(price, lo, hi, hash) of one vector instruction of `kernel_key` in SIMD cycles: the measured saturated issue costs
    (profiles/r03_ubench_issue.md) over the kernel's static instruction-class mix (profiles/isa_mix.json)."""
    mix = _load_json("isa_mix.json")
    if not mix or kernel_key not in mix.get("kernels", {}):
        return None
    lo, hi = mix["valu_cycles_per_inst_bounds"]
    return mix["kernels"][kernel_key]["valu_cycles_per_inst_static_mix"], lo, hi, mix


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
    ac = kind == "qa"                           # ActorCritic: value head, train_net on (s, a, r, s') with its own gamma
    if ac:
        gamma = 0.98
    ws = [(rng.uniform(-0.06, 0.06, NN.ac_n_params(A) if ac else NN.n_params(A))).astype(np.float32) for _ in range(n_nn)]
    ms = [np.zeros_like(w) for w in ws]; vs = [np.zeros_like(w) for w in ws]; st = [0] * n_nn
    mem = [[] for _ in range(n_nn)]
    qcfg, _ = O.cfg_from_config({"agents": [dict(CFG_AGENT, states=1), dict(CFG_AGENT, states=1)],
                                 "environment": CFG["environment"]}, 1, 1)
    price, steps, t0 = 5.0, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds_target:
        for _ in range(T_STEPS):
            acts = [int(NN.sample_action(NN.ac_split(ws[k], A)[0] if ac else ws[k], A, [price], [rng.uniform()])[0]) for k in range(n_nn)]
            sc = [NN.scale(a, A, lo, hi) for a in acts]
            if n_nn == 1:                       # a greedy-free stand-in for the tabular opponent's action
                sc = [lo + (hi - lo) * rng.randint(A) / (A - 1.0)] + sc
            nprice, rew = O.env_step(qcfg, sc)
            for k in range(n_nn):
                mem[k].append((price, acts[k], rew[k - n_nn], nprice))
            price = nprice
            steps += 1
        for k in range(n_nn):
            if len(mem[k]) >= 1000:
                pr, aa, rw, npr = zip(*mem[k])
                if ac:
                    ws[k], ms[k], vs[k], st[k], _ = NN.ac_train_net(ws[k], ms[k], vs[k], st[k], A, pr, aa, rw, npr, gamma, 0.0)
                else:
                    ws[k], ms[k], vs[k], st[k], _ = NN.train_net(ws[k], ms[k], vs[k], st[k], A, pr, aa, rw, gamma, 0.0)
                mem[k] = []
    wall = time.perf_counter() - t0
    return dict(value=steps / wall, unit="env-steps/s", cores=1, kind="port",
                sample="1 game x %d steps, %d network updates, oracle/nn_oracle.py (numpy float32), %.1f s"
                       % (steps, sum(st), wall), **cpu_info())


def nn_roofline(agents, env_steps, gpu_s, lib):
    """Roofline block of a neural line: the whole step (episode kernel + update kernels) against vector-instruction issue
    and against HBM, from profiles/nn_traffic.json (executed VALU instructions and HBM bytes per env-step of each kernel,
    rocprofv3 --pmc on this shape) x the env-steps of this run / the live GPU time of the timed region.  Executed work
    only: the policy memo and the state folding SKIP work, and skipped work is not counted."""
    nt = (_load_json("nn_traffic.json") or {}).get("pairings", {}).get(agents)
    pr = issue_price(KEY_MIXED.get(agents, ""))
    out = {"bound": "valu_issue", "kernel": "k_ptuple_episodes + k_nn_reinforce_train (whole step)", "unit": "Ginst/s",
           "achieved": None, "peak": None, "frac": None, "traffic": None, "gpu_time_ms": gpu_s * 1e3}
    if not nt or not pr:
        out["note"] = "no PMC summary for this pairing in profiles/nn_traffic.json: only the time is measured here"
        return out
    price, lo, hi, mix = pr
    clock = nt["clock_ghz"]
    simd_cycles = N_SIMD * gpu_s * clock * 1e9
    valu = sum(k["valu_insts_per_env_step"] * (issue_price(k["isa_key"]) or pr)[0] for k in nt["kernels"].values()) * env_steps
    insts = sum(k["valu_insts_per_env_step"] for k in nt["kernels"].values()) * env_steps
    hbm = sum(k["hbm_bytes_per_env_step"] for k in nt["kernels"].values()) * env_steps
    mean_price = valu / insts
    out.update(achieved=insts / gpu_s / 1e9, peak=N_SIMD * clock / mean_price, frac=valu / simd_cycles,
               frac_bounds=[insts * lo / simd_cycles, insts * hi / simd_cycles],
               traffic=hbm, measured_hbm_gbps=hbm / gpu_s / 1e9, measured_hbm_frac=hbm / gpu_s / 1e9 / HBM_PEAK_GBS,
               kernels={n: {"time_share_profiled": k["time_share"], "valu_insts_per_env_step": k["valu_insts_per_env_step"],
                            "hbm_bytes_per_env_step": k["hbm_bytes_per_env_step"], "wait_frac": k.get("wait_frac")}
                        for n, k in nt["kernels"].items()},
               inputs={"clock_ghz": clock, "valu_cycles_per_inst": mean_price, "source": nt.get("source"),
                       "counters_collected_on": nt.get("nn"), "library": lib.get("nn")},
               stale=bool(nt.get("nn") != lib.get("nn") or mix.get("nn") != lib.get("nn")))
    return out


def run_nn(agents, games, steps, warmup, nn_loop="fused", cpu_seconds=10.0, lib=None, noise_prob=0.0):
    """BASELINE configs[3]: neural-policy agents (agents.py:119-220) x 65,536 games through mixed.MixedGameBatch
    (fused: thrl_mixed_episodes + the batched update kernels; unfused: one launch per reference call).
    agents: rr = 2 x Reinforce, qr = the reference's shipped pairing QTable vs Reinforce, qa / qc = QTable vs
    ActorCritic / CAC, qq = 2 x QTable on the mixed kernel.  steps = episodes; the policy trains every 10 episodes.
    Returns the JSON line as a dict."""
    cpu = cpu_baseline_nn(agents, cpu_seconds) if cpu_seconds > 0 else None
    import torch
    from th_rl_amd.mixed import MixedGameBatch
    lib = lib or library_info()
    G = games
    ag = {"name": "Reinforce", "gamma": 0.995, "actions": 21, "states": 1, "action_range": [0.2, 0.4]}
    first = dict(ag) if agents == "rr" else dict(CFG["agents"][0])
    second = {"qq": dict(CFG["agents"][1]), "qa": dict(ag, name="ActorCritic", gamma=0.98),
              "qc": {"name": "CAC", "gamma": 0.98, "states": 1, "action_range": [0.2, 0.4]}}.get(agents, dict(ag))
    config = {"agents": [first, second], "environment": dict(CFG["environment"], noise_prob=noise_prob)}   # (the shipped configs: 0)
    fused = nn_loop == "fused"
    mb = MixedGameBatch(config, n_games=G, dtype="float32", seed=0).init_tables()
    mb.run(warmup, fused=fused, per_game_logs=False) if fused else mb.run(warmup, fused=False)
    torch.cuda.synchronize()
    upd0 = {i: rb.step for i, rb in mb.nn.items()}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    mb.run(steps, fused=fused, per_game_logs=False) if fused else mb.run(steps, fused=False)
    ev1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    who = {"qr": "QTable vs Reinforce", "rr": "2-agent Reinforce", "qq": "2-agent QTable (mixed kernel)",
           "qa": "QTable vs ActorCritic", "qc": "QTable vs CAC"}[agents]
    updates = sum(rb.step - upd0[i] for i, rb in mb.nn.items())
    env_steps = float(G) * T_STEPS * steps
    gpu_s = ev0.elapsed_time(ev1) * 1e-3
    out = {"metric": "env-steps/sec, %s (neural policy) x %d games" % (who, G),
           "value": env_steps / dt, "unit": "env-steps/s", "n_gpus": 1, "steps": steps,
           "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "dtype": "f32", "data": "synthetic", "vs_baseline": None,
           "config": {"workload": "%s x %d games, %s loop, MFMA off%s" %
                                  (who, G, nn_loop, "" if not noise_prob else "; DIAGNOSTIC: env noise_prob %g (the shipped configs: 0)" % noise_prob),
                      "network_updates": updates, "noise_prob": noise_prob},
           "roofline": nn_roofline(agents, env_steps, gpu_s, lib)}
    if cpu is not None:
        out["cpu_baseline"] = cpu
    del mb
    torch.cuda.empty_cache()
    return out


def bench_nn(args):
    lib = library_info()
    G = args.games if args.games != (1 << 20) else 65536
    out = run_nn(args.nn_agents, G, args.steps, args.warmup, args.nn_loop, 0.0 if args.no_cpu_baseline else args.cpu_seconds, lib,
                 noise_prob=float(args.noise_prob or 0.0))
    out["library"] = lib
    emit(json.dumps(out))


_JSON_OUT = None


def protect_stdout():
    """The driver reads ONE JSON line from stdout.  Libraries print there too -- gloo announces "[Gloo] Rank 0 is connected to
    1 peer ranks" through C++ std::cout when a process group forms -- so file descriptor 1 is pointed at stderr for the rest of
    the run and the JSON line goes to the saved, real stdout."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(line):
    out = _JSON_OUT or sys.stdout
    out.write(line + "\n")
    out.flush()


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
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary lines (2 x Reinforce and QTable vs Reinforce x 65,536 games = BASELINE configs[3])")
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
    protect_stdout()
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
    lib = library_info()

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
        # ---- roofline of the dominant kernel: the binding bound is vector-instruction issue.
        # instructions per env-step: committed PMC summary (profiles/traffic.json <- profiles/r03c25_pmc_summary.csv);
        # issue cost per instruction: the microbenchmark priced over the kernel's instruction mix; launch time: live.
        tj = _load_json("traffic.json")
        default_cfg = (args.noise_prob in (None, 0.0) and args.epsilon is None and not args.no_counters
                       and args.dtype == "float32" and args.max_steps is None and args.capacity is None
                       and args.pretrain == 0)
        env_launch = float(G) * T_run * e_launch
        roof = {"bound": "valu_issue", "kernel": "k_wave_episodes" if gb.last_kernel == "wave" else "k_generic_episodes",
                "achieved": None, "peak": None, "unit": "Ginst/s", "frac": None, "traffic": None,
                "avg_launch_ms": avg_launch_s * 1e3, "median_launch_ms": times[len(times) // 2] * 1e3, "launches_timed": len(full),
                # the contract formula of SURVEY 8(d): 368 algorithmic bytes per env-step / launch time.  Tables stay in LDS
                # for a whole launch, so this exceeds the HBM peak: a reuse figure, not a fraction of a limit
                "algorithmic_bytes_per_launch": algo_bytes_launch, "algorithmic_equiv_gbps": achieved,
                "algorithmic_reuse_vs_hbm_peak": achieved / HBM_PEAK_GBS}
        pr = issue_price(KEY_WAVE)
        if tj and tj.get("kernel") == gb.last_kernel and default_cfg and "model" in tj and pr and gb.last_kernel == "wave":
            price, p_lo, p_hi, mix = pr
            iss = tj["issue"]
            ipe, clock = iss["insts_per_env_step"], iss["clock_ghz"]
            simd_cycles = N_SIMD * avg_launch_s * clock * 1e9
            valu = ipe["valu"] * env_launch
            m = tj["model"]
            traffic = float(G) * (m["bytes_per_game_per_launch"] + m["bytes_per_game_per_episode"] * e_launch)
            sc = iss.get("occupancy_scaling") or {}
            roof.update(
                achieved=valu / avg_launch_s / 1e9, peak=N_SIMD * clock / price, frac=valu * price / simd_cycles,
                # the same with every vector instruction at the fast (all-VGPR two-source) / the slow price
                frac_bounds=[valu * p_lo / simd_cycles, valu * p_hi / simd_cycles],
                traffic=traffic, measured_hbm_gbps=traffic / avg_launch_s / 1e9,
                measured_hbm_frac=traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS,
                # the other units, same pricing (scalar unit: one instruction per cycle per CU; LDS: 128 B/clk/CU)
                salu_frac=ipe["salu"] * env_launch * mix["price_cycles"]["salu"] / simd_cycles,
                lds_frac=ipe["lds"] * env_launch * mix["price_cycles"]["lds_b32"] / simd_cycles,
                inputs={"valu_insts_per_env_step": ipe["valu"], "salu": ipe["salu"], "lds": ipe["lds"], "branch": ipe["branch"],
                        "valu_cycles_per_inst": price, "price_bounds": [p_lo, p_hi], "clock_ghz": clock, "simds": N_SIMD,
                        "source": [iss.get("source"), "profiles/r03_ubench_issue.md", "profiles/isa_mix.json"],
                        "counters_collected_on": tj.get("wave"), "library": lib.get("wave")},
                wave_time={"issuing": iss.get("issue_frac"), "parked_on_waitcnt": iss.get("wait_frac"), "stalled_at_issue": iss.get("stall_frac"),
                           "latency_share_at_20_waves": sc.get("latency_share_at_20_waves")},
                stale=bool(tj.get("wave") != lib.get("wave") or mix.get("wave") != lib.get("wave")))
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
            "roofline": roof,
            "library": lib,
        }
        if world > 1:
            # N = 1-equivalent figures so a SCALE record can be checked against BENCH directly
            out["per_rank"] = {"games_per_gpu": G, "seconds": per_rank,
                               "value": [G * T_run * args.steps / s for s in per_rank]}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if world == 1 and default_cfg and not args.no_secondary and args.kernel == "wave" and G == (1 << 20):
            # BASELINE configs[3] in front of the driver: the neural pairings at 65,536 games, after the headline's
            # tensors are released (a few seconds each; their own roofline and cpu_baseline)
            del gb
            torch.cuda.empty_cache()
            # (warm-up = the timed length: the same launch shapes, so no allocation falls into the timed region)
            out["secondary"] = [run_nn(p, 65536, 40, 40, "fused", 0.0 if args.no_cpu_baseline else min(args.cpu_seconds, 5.0), lib)
                                for p in ("rr", "qr", "qa")]
        emit(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
