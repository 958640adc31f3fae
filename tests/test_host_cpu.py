"""CPU-only tests: the C-ABI library loads and exports every declared symbol, the
host logic of the ABI (no GPU needed), the Python mirror of the reference API, and
the world_size-2 sharding path over gloo."""
import ctypes
import json
import os
import re
import socket

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CFG_AGENT = dict(name="QTable", gamma=0.95, actions=21, states=100, alpha=0.1, eps_end=0.001,
                 epsilon=0.5, eps_step=0.9995, action_range=[0.2, 0.4])
CFG_ENV = dict(name="NoisyPriceState", noise_prob=0, a=10, b=1, nplayers=2, max_steps=100)
CFG = {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)], "environment": dict(CFG_ENV),
       "training": {"print_freq": 500, "epochs": 20}}
# the reference's example_config.json content (QTable vs Reinforce), restated as data
EXAMPLE = {"agents": [dict(CFG_AGENT),
                      {"name": "Reinforce", "gamma": 0.995, "actions": 21, "states": 1, "action_range": [0.2, 0.4]}],
           "environment": dict(CFG_ENV), "training": {"print_freq": 500, "epochs": 20000}}


@pytest.fixture(scope="module")
def lib():
    from th_rl_amd import build, _lib
    build.build()          # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from th_rl_amd import _lib
    header = open(os.path.join(ROOT, "include", "thrl.h")).read()
    declared = sorted(set(re.findall(r"\b(thrl_[a-z0-9_]+)\s*\(", header)))
    assert declared and set(declared) == set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.thrl_version() == 3


def test_struct_layout_matches_header(lib):
    """ctypes mirrors of thrl_cfg / thrl_buffers / thrl_run have the C sizes."""
    import subprocess, tempfile
    from th_rl_amd import _lib
    src = '#include <stdio.h>\n#include "thrl.h"\nint main(){printf("%zu %zu %zu %zu\\n",sizeof(thrl_cfg),sizeof(thrl_buffers),sizeof(thrl_run),sizeof(thrl_mixed));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        sizes = list(map(int, subprocess.check_output([os.path.join(d, "s")]).split()))
    assert sizes == [ctypes.sizeof(_lib.Cfg), ctypes.sizeof(_lib.Buffers), ctypes.sizeof(_lib.Run), ctypes.sizeof(_lib.Mixed)]


def test_host_logic_layout_and_kernel_selection(lib):
    from th_rl_amd import _lib
    cfg, eps = _lib.cfg_from_config(CFG, 1 << 20, 0)
    assert eps == [0.5, 0.5]
    assert lib.thrl_table_stride(ctypes.byref(cfg)) == 2 * 101 * 21
    assert lib.thrl_table_offset(ctypes.byref(cfg), 1) == 101 * 21
    assert lib.thrl_workspace_bytes(ctypes.byref(cfg)) > 0
    assert lib.thrl_select_kernel(ctypes.byref(cfg), 0) == _lib.KERNEL_WAVE
    assert lib.thrl_select_kernel(ctypes.byref(cfg), 1) == _lib.KERNEL_WAVE         # injected draws too
    # replay memory holds at most T=100 transitions/agent here: 2*100*G entries * (3*2 + 2*8) bytes
    assert lib.thrl_replay_mem_bytes(ctypes.byref(cfg)) == (1 << 20) * 2 * 100 * 22
    cn = json.loads(json.dumps(CFG)); cn["environment"]["noise_prob"] = 0.05; cn["agents"][0]["alpha"] = 0.5
    cfgn, _ = _lib.cfg_from_config(cn, 64, 0)
    assert lib.thrl_select_kernel(ctypes.byref(cfgn), 0) == _lib.KERNEL_WAVE       # noise + per-agent alpha
    cfg64, _ = _lib.cfg_from_config(CFG, 1 << 20, 1)
    assert lib.thrl_select_kernel(ctypes.byref(cfg64), 0) == _lib.KERNEL_WAVE       # float64 tables: same kernel, QT = double
    # buffers that span episodes (max_steps 30 < min_memory 100: trains every 4th episode on 120 transitions)
    # and deque overflow for BOTH agents are training cycles of the wave kernel
    for mod in (dict(T=30), dict(capboth=64), dict(T=10)):
        c = json.loads(json.dumps(CFG))
        if "T" in mod: c["environment"]["max_steps"] = mod["T"]
        if "capboth" in mod:
            for a in c["agents"]: a["capacity"] = mod["capboth"]; a["min_memory"] = 20
        cfgw, _ = _lib.cfg_from_config(c, 64, 0)
        assert lib.thrl_select_kernel(ctypes.byref(cfgw), 0) == _lib.KERNEL_WAVE, mod
    for mod, why in [(dict(mm=300), "256 transitions"), (dict(T=3), "32 episodes"),
                     (dict(nag=3), "2 agents"), (dict(cap=64), "cycles")]:
        c = json.loads(json.dumps(CFG))
        if "noise" in mod: c["environment"]["noise_prob"] = mod["noise"]
        if "T" in mod: c["environment"]["max_steps"] = mod["T"]
        if "cap" in mod: c["agents"][0]["capacity"] = mod["cap"]
        if "mm" in mod:
            for a in c["agents"]: a["min_memory"] = mod["mm"]
        if "nag" in mod: c["agents"].append(dict(CFG_AGENT)); c["environment"]["nplayers"] = 3
        cfg2, _ = _lib.cfg_from_config(c, 64, mod.get("q", 0))
        assert lib.thrl_select_kernel(ctypes.byref(cfg2), 0) == _lib.KERNEL_GENERIC, mod
        assert why in lib.thrl_last_error().decode(), (mod, lib.thrl_last_error())


def test_workspace_is_sized_from_the_config(lib):
    from th_rl_amd import _lib
    def ws(G, **mod):
        c = json.loads(json.dumps(CFG))
        c["environment"].update(mod.get("env", {}))
        cfg, _ = _lib.cfg_from_config(c, G, mod.get("q", 0))
        return lib.thrl_workspace_bytes(ctypes.byref(cfg))
    big, small = ws(1 << 20), ws(64)
    # 20 resident waves on each of 256 CUs: per wave 32 x 4 log partials + 32 episodes x 2 segments x 64 packed transitions
    assert big == 16384 + 5120 * (1024 + 32 * 2 * 64 * 4) and small == 16384 + 64 * (1024 + 32 * 2 * 64 * 4)
    assert ws(1 << 20, env={"max_steps": 128}) == big and ws(1 << 20, env={"max_steps": 129}) > big
    # float64 tables are twice the LDS per game: 11 resident waves per CU
    assert ws(1 << 20, q=1) == 16384 + 11 * 256 * (1024 + 32 * 2 * 64 * 4)
    c3 = json.loads(json.dumps(CFG)); c3["agents"].append(dict(CFG_AGENT)); c3["environment"]["nplayers"] = 3
    cfg3, _ = _lib.cfg_from_config(c3, 1 << 20, 0)
    assert lib.thrl_workspace_bytes(ctypes.byref(cfg3)) == 16384          # generic kernel: nothing kept in the workspace
    bad, _ = _lib.cfg_from_config(CFG, 4, 0)
    bad.n_agents = 0
    assert lib.thrl_workspace_bytes(ctypes.byref(bad)) == 0


def test_host_entry_points_from_two_threads(lib):
    """Re-entrancy of the host logic (include/thrl.h): two threads hammer thrl_select_kernel /
    thrl_workspace_bytes with different configs; every answer equals the single-threaded one and
    each thread sees its own thrl_last_error()."""
    import threading
    from th_rl_amd import _lib
    wave_cfg, _ = _lib.cfg_from_config(CFG, 1 << 16, 0)
    c3 = json.loads(json.dumps(CFG)); c3["agents"].append(dict(CFG_AGENT)); c3["environment"]["nplayers"] = 3
    gen_cfg, _ = _lib.cfg_from_config(c3, 1 << 16, 0)
    want = {"wave": (lib.thrl_select_kernel(ctypes.byref(wave_cfg), 0), lib.thrl_workspace_bytes(ctypes.byref(wave_cfg))),
            "gen": (lib.thrl_select_kernel(ctypes.byref(gen_cfg), 0), lib.thrl_workspace_bytes(ctypes.byref(gen_cfg)))}
    assert want["wave"][0] == _lib.KERNEL_WAVE and want["gen"][0] == _lib.KERNEL_GENERIC
    bad = []
    def hammer(name, cfg, expect_msg):
        for _ in range(2000):
            k = lib.thrl_select_kernel(ctypes.byref(cfg), 0)
            msg = lib.thrl_last_error().decode()
            w = lib.thrl_workspace_bytes(ctypes.byref(cfg))
            if (k, w) != want[name] or (expect_msg and expect_msg not in msg):
                bad.append((name, k, w, msg))
                return
    ts = [threading.Thread(target=hammer, args=("wave", wave_cfg, None)),
          threading.Thread(target=hammer, args=("gen", gen_cfg, "2 agents"))]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not bad, bad[:2]


def test_error_codes_never_throw(lib):
    from th_rl_amd import _lib
    cfg, _ = _lib.cfg_from_config(CFG, 4, 0)
    cfg.n_agents = 0
    assert lib.thrl_select_kernel(ctypes.byref(cfg), 0) == -1          # THRL_ERR_BAD_CONFIG
    assert b"n_agents" in lib.thrl_last_error()
    cfg, _ = _lib.cfg_from_config(CFG, 4, 0)
    cfg.n_actions[1] = 1
    assert lib.thrl_replay_mem_bytes(ctypes.byref(cfg)) == 0
    cfg, _ = _lib.cfg_from_config(CFG, 4, 0)
    # NULL buffers are rejected before anything touches the device
    assert lib.thrl_qtable_init(ctypes.byref(cfg), None, None, None, 0, 0, None, None) == -2
    run = _lib.Run(); bufs = _lib.Buffers()
    assert lib.thrl_qtable_episodes(ctypes.byref(cfg), ctypes.byref(bufs), ctypes.byref(run), None) == -2
    assert lib.thrl_play_greedy(ctypes.byref(cfg), None, None, 1, 0, 0, None, None, None) == -2
    assert lib.thrl_op_env_step(ctypes.byref(cfg), None, None, None, None, None, None) == -2


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from th_rl_amd.batched import GameBatch
    from th_rl_amd._lib import ThrlError
    with pytest.raises(ThrlError, match="no GPU|no CPU fallback"):
        GameBatch(CFG, n_games=4)
    from th_rl_amd.agents import QTable
    from th_rl_amd.environments import NoisyPriceState
    with pytest.raises(ThrlError):
        QTable(**CFG_AGENT).get_action(np.array([3.0]))
    with pytest.raises(ThrlError):
        NoisyPriceState(**CFG_ENV).step([0.3, 0.3])
    # the neural agents and the mixed-agent batch have no CPU path either
    from th_rl_amd.agents import CAC, ActorCritic, Reinforce
    from th_rl_amd.mixed import MixedGameBatch
    from th_rl_amd.nn import CACBatch, ReinforceBatch
    nn_kw = dict(states=1, actions=21, action_range=[0.2, 0.4])
    for make in (lambda: MixedGameBatch({"agents": [dict(CFG_AGENT, name="QTable"), dict(nn_kw, name="Reinforce")],
                                         "environment": dict(CFG_ENV)}, n_games=2),
                 lambda: ReinforceBatch(2, actions=21), lambda: CACBatch(2),
                 lambda: Reinforce(**nn_kw).get_action(np.array([3.0])),
                 lambda: ActorCritic(**nn_kw).sample_action(np.array([3.0])),
                 lambda: CAC(states=1, action_range=[0.2, 0.4]).get_action(np.array([3.0]))):
        with pytest.raises(ThrlError):
            make()


def test_create_game_reference_schema(tmp_path):
    from th_rl_amd import trainer
    from th_rl_amd.agents import QTable, Reinforce
    from th_rl_amd.environments import NoisyPriceState, PricingGame
    p = tmp_path / "example_config.json"
    p.write_text(json.dumps(EXAMPLE, indent=3))
    np.random.seed(0)
    config, agents, env = trainer.create_game(str(p))
    assert config == EXAMPLE
    assert isinstance(agents[0], QTable) and isinstance(agents[1], Reinforce)
    assert isinstance(env, NoisyPriceState) and PricingGame is NoisyPriceState
    assert trainer.train is trainer.train_one
    q = agents[0]
    assert q.table.shape == (101, 21) and q.table.dtype == np.float64
    assert abs(q.table.mean() - 250.0) < 0.1 and not q.counter.any()
    assert (q.states, q.actions, q.max_state, q.min_memory, q.epsilon) == (100, 21, 10, 100, 0.5)
    assert list(q.action_space) == list(range(21)) and len(q.memory) == 0
    assert 0 <= env.state < 10 and env.max_steps == 100 and env.episode == 0
    nash, cartel = env.get_optimal()
    assert abs(nash - 200 / 9) < 1e-12 and cartel == 25.0
    assert sum(x.numel() for x in agents[1].parameters()) == 5909
    bad = json.loads(json.dumps(EXAMPLE)); bad["environment"]["nplayers"] = 3
    p.write_text(json.dumps(bad))
    with pytest.raises(AssertionError, match="Bad config"):
        trainer.create_game(str(p))


def test_replay_buffer_semantics():
    """deque(maxlen) append / ordered replay / empty / sample (buffers.py:6-41)."""
    from collections import namedtuple
    import torch
    from th_rl_amd.buffers import ReplayBuffer
    Exp = namedtuple("Experience", ["state", "action", "reward", "done", "new_state"])
    rb = ReplayBuffer(5, Exp)
    for k in range(7):
        rb.append(np.array([float(k)]), k, 0.5 * k, True, np.array([k + 1.0]))
    assert len(rb) == 5
    st, ac, rw, nd, ns = rb.replay()
    assert list(ac) == [2, 3, 4, 5, 6] and list(rw) == [1.0, 1.5, 2.0, 2.5, 3.0]
    st, ac, rw, nd, ns = rb.replay(replay_size=2)
    assert list(ac) == [5, 6]
    cast = [torch.float, torch.int64, torch.float, torch.float, torch.float]
    t = list(rb.replay(cast))
    assert t[0].shape == (5, 1) and t[1].dtype == torch.int64 and t[1].tolist() == [2, 3, 4, 5, 6]
    np.random.seed(1)
    _, ac, _, _, _ = rb.sample(3)
    assert len(set(ac)) == 3 and set(ac) <= {2, 3, 4, 5, 6}
    rb.empty()
    assert len(rb) == 0 and rb.buffer.maxlen == 5


def test_qtable_save_load_format(tmp_path):
    from th_rl_amd.agents import QTable
    np.random.seed(3)
    q = QTable(**CFG_AGENT)
    q.counter[3, 4] = 7
    q.save(str(tmp_path / "0"))
    assert sorted(os.listdir(tmp_path)) == ["0.npy", "0_counter.npy"]
    assert os.path.getsize(tmp_path / "0.npy") == 17096
    r = QTable(**CFG_AGENT)
    r.load(str(tmp_path / "0"))
    assert np.array_equal(r.table, q.table) and r.counter[3, 4] == 7 and r.table.dtype == np.float64


def test_shard_range_partitions():
    from th_rl_amd.sharding import shard_range
    for total, world in [(1 << 23, 8), (10, 3), (7, 8), (1 << 20, 1)]:
        blocks = [shard_range(total, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and sum(n for _, n in blocks) == total
        for (o0, n0), (o1, _) in zip(blocks, blocks[1:]):
            assert o0 + n0 == o1
    with pytest.raises(ValueError):
        shard_range(8, 8, 8)


def test_launch_shard_config_is_independent_of_the_shard_size():
    """th_rl_amd.launch: a shard of ONE game must still run float32 tables initialised from Philox by
    global game id (train_one's one-game defaults are float64 + numpy's RNG), and there are never more
    ranks than games."""
    from th_rl_amd.launch import effective_world, shard_training
    cfg = dict(CFG, training={"epochs": 3, "n_games": 2, "seed": 5, "sweep": {"gamma": [0.3, 0.9]}})
    t0, off0, n0 = shard_training(cfg, 0, 2)
    t1, off1, n1 = shard_training(cfg, 1, 2)
    assert (off0, n0, off1, n1) == (0, 1, 1, 1)
    for t in (t0, t1):
        assert t["dtype"] == "float32" and t["philox_init"] is True and t["n_games"] == 1 and t["seed"] == 5
    assert t1["game_offset"] == 1 and t0["sweep"] == {"gamma": [0.3]} and t1["sweep"] == {"gamma": [0.9]}
    assert effective_world(cfg, 8) == 2 and effective_world(cfg, 1) == 1
    three = dict(CFG, training={"n_games": 3, "seed": 1, "dtype": "float64", "game_offset": 10})
    t, off, n = shard_training(three, 1, 2)
    assert (t["dtype"], t["game_offset"], n) == ("float64", 12, 1)
    one = dict(CFG, training={"n_games": 1, "seed": 1})          # an unsharded single game keeps the reference defaults
    t, off, n = shard_training(one, 0, effective_world(one, 4))
    assert t["dtype"] == "float64" and t["philox_init"] is False
    with pytest.raises(ValueError):
        shard_training(cfg, 2, 3)                                 # an empty shard is a caller error
    with pytest.raises(SystemExit):
        shard_training(dict(CFG, training={"n_games": 4}), 0, 2)  # no shared seed


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _shard_worker(rank, world, port, total, E, q):
    import torch.distributed as dist
    from oracle import oracle as O
    from th_rl_amd.sharding import shard_range, aggregate_logs
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    off, n = shard_range(total, rank, world)
    cfg, eps = O.cfg_from_config(CFG, n, 0)
    tab, cnt, st = O.init(cfg, seed=7, game_offset=off)
    out = O.episodes(cfg, tab, cnt, st, eps, O.Memory(cfg), E, seed=7, game_offset=off)
    merged = aggregate_logs(out["reward_log"], n)
    dist.barrier()
    if rank == 0:
        q.put((merged, tab.sum()))
    dist.destroy_process_group()


def test_two_rank_seed_sharding_over_gloo():
    """world_size 2 on CPU: each rank runs its block of games (the oracle stands in for
    the device kernel), logs are merged with a weighted mean; equals the 1-rank run."""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    total, E = 11, 3            # uneven split: 6 + 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, total, E, q)) for r in range(2)]
    [p.start() for p in procs]
    merged, _ = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    cfg, eps = O.cfg_from_config(CFG, total, 0)
    tab, cnt, st = O.init(cfg, seed=7)
    out = O.episodes(cfg, tab, cnt, st, eps, O.Memory(cfg), E, seed=7)
    np.testing.assert_allclose(merged, out["reward_log"], rtol=1e-13)


def test_load_experiment_reads_the_run_the_reference_ships():
    """tests/golden/ref_run_example_config = the data files of th_rl/some_path/runs/example_config/0
    (QTable vs Reinforce, 20,000 epochs; log.csv cut to its last 200 rows).  load_experiment
    (utils.py:12-24) must read them unchanged: table / counter arrays, the torch state_dict of the
    Reinforce agent (loaded with weights_only=True), the log columns."""
    import torch
    from th_rl_amd.utils import load_experiment
    loc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_run_example_config")
    config, agents, env, actions, rewards = load_experiment(loc)
    assert [type(a).__name__ for a in agents] == ["QTable", "Reinforce"] and type(env).__name__ == "NoisyPriceState"
    assert np.array_equal(agents[0].table, np.load(os.path.join(loc, "0.npy"))) and agents[0].table.shape == (101, 21)
    assert agents[0].counter.sum() == 2000000.0              # 20,000 epochs x 100 steps
    sd = torch.load(os.path.join(loc, "1"), weights_only=True)
    assert sorted(sd) == ["fc1.bias", "fc1.weight", "fc_pi.bias", "fc_pi.weight"]
    for k, v in agents[1].state_dict().items():
        assert torch.equal(v, sd[k])
    assert np.array_equal(agents[1].flat_params()[:256], sd["fc1.weight"].numpy().ravel())
    assert list(actions.columns) == ["QTable0", "Reinforce1"] and actions.shape == rewards.shape == (201, 2)
    assert config["training"]["epochs"] == 20000


def test_build_info_names_the_binary(lib):
    """thrl_build_info(): ABI version, ablation mask (0 = the product library) and the hash of the sources the
    binary was built from -- bench.py prints it and refuses an ablation build for any reported number."""
    from th_rl_amd import _lib, build
    info = _lib.build_info()
    assert info["abi"] == 3 and info["ablate"] == 0 and lib.thrl_ablate_mask() == 0
    assert info["src"] == build.source_hash() and re.fullmatch(r"[0-9a-f]{12}", info["src"])
    assert info["wave"] == build.source_hash(build.WAVE_FILES) and info["nn"] == build.source_hash(build.NN_FILES)
    assert info["path"].endswith("libthrl_hip.so")


def test_kernel_variant_ids_are_validated_on_the_host(lib):
    """thrl_run.kernel accepts the two variant-pinning ids; an unknown id is THRL_ERR_BAD_CONFIG (no GPU needed:
    validation happens before any launch)."""
    from th_rl_amd import _lib
    cfg, eps = _lib.cfg_from_config(CFG, 8, 0)
    b, r = _lib.Buffers(), _lib.Run()
    b.q = b.state = 1          # non-NULL; never dereferenced on this path
    r.n_episodes, r.kernel = 1, 7
    assert lib.thrl_qtable_episodes(ctypes.byref(cfg), ctypes.byref(b), ctypes.byref(r), None) == -1
    assert b"unknown kernel id 7" in lib.thrl_last_error()
    assert (_lib.KERNEL_WAVE_PLAIN, _lib.KERNEL_WAVE_GREEDY) == (3, 4)
    header = open(os.path.join(ROOT, "include", "thrl.h")).read()
    assert "THRL_KERNEL_WAVE_PLAIN = 3" in header and "THRL_KERNEL_WAVE_GREEDY = 4" in header


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_gpus_n_spawns_n_ranks_or_refuses(monkeypatch, capsys):
    """`python bench.py --gpus N` as a plain command (no WORLD_SIZE): the parent becomes a launcher that starts N
    child ranks with RANK / WORLD_SIZE / MASTER_* set and relays rank 0's line -- or refuses when fewer than N GPUs are
    visible.  It never prints an N-GPU line from one process (round-2 verdict / advisor finding)."""
    bench = _load_bench()
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None):
            started.append((cmd, env))
            self.rank = int(env["RANK"])
        def communicate(self):
            return (b'{"n_gpus": 2}\n', None)
        def wait(self):
            return 0

    # no GPU visible here: refuses, starts nothing
    monkeypatch.setattr(bench, "visible_gpus", lambda: 0)
    assert bench.self_launch(2, ["--gpus", "2"], popen=FakeProc) == 2 and not started
    assert "refusing" in capsys.readouterr().err
    # enough GPUs: N children, each with its rank and the same rendezvous
    monkeypatch.setattr(bench, "visible_gpus", lambda: 8)
    assert bench.self_launch(4, ["--gpus", "4", "--steps", "3"], popen=FakeProc) == 0
    assert len(started) == 4
    assert [e["RANK"] for _, e in started] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" for _, e in started)
    assert len({e["MASTER_PORT"] for _, e in started}) == 1
    assert all(c[1].endswith("bench.py") and c[2:] == ["--gpus", "4", "--steps", "3"] for c, _ in started)
    assert capsys.readouterr().out.strip() == '{"n_gpus": 2}'
    # a failing child fails the launch and nothing is relayed
    FakeProc.wait = lambda self: 3 if self.rank == 1 else 0
    assert bench.self_launch(2, ["--gpus", "2"], popen=FakeProc) == 3
    assert capsys.readouterr().out == ""


def test_bench_gpus_2_as_a_plain_command_fails_loudly_without_gpus():
    """End to end, in a real child process: no WORLD_SIZE, --gpus 2, no GPU in this container => exit code 2 and no
    JSON line (before the fix: one process printed n_gpus 2 and twice the single-GPU value)."""
    import subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and p.stdout.strip() == "" and "refusing" in p.stderr


def test_oracle_golden_under_sanitizers():
    """AddressSanitizer + UndefinedBehaviorSanitizer build of oracle/thrl_oracle.c (oracle/Makefile,
    libthrl_oracle_san.so) run under the whole golden suite in a child process with libasan preloaded: the
    sanitizers' home on this pool is the CPU build."""
    import subprocess, sys
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    if not (os.path.isabs(asan) and os.path.exists(asan)):
        pytest.skip("gcc has no libasan here")
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", THRL_ORACLE_SANITIZE="1")
    code = ("import sys, pytest; from oracle import oracle as O; "
            "assert O.LIB_PATH.endswith('libthrl_oracle_san.so'); "
            "rc = pytest.main(['-x', '-q', '-p', 'no:cacheprovider', 'tests/test_oracle_golden.py']); "
            "assert 'libthrl_oracle_san.so' in open('/proc/self/maps').read(); sys.exit(int(rc))")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "passed" in p.stdout and "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr


def test_kernel_selection_covers_individual_grids_on_the_host(lib):
    """thrl_select_kernel / thrl_workspace_bytes (host logic, no GPU): two agents on one grid -> the wave kernel; 1-4 agents with
    individual grids in a game (with or without env noise) that trains once per episode -> the tuple-chain kernel; buffers that
    span episodes, five agents or too many action tuples -> the generic kernel."""
    from th_rl_amd import _lib
    def sel(agents, **env):
        conf = {"agents": agents, "environment": dict(CFG_ENV, nplayers=len(agents), **env)}
        cfg, _ = _lib.cfg_from_config(conf, 4096, 0)
        return _lib.KERNEL_NAMES[lib.thrl_select_kernel(ctypes.byref(cfg), 0)], lib.thrl_workspace_bytes(ctypes.byref(cfg))
    a21 = dict(CFG_AGENT)
    a11 = dict(CFG_AGENT, actions=11, states=50, action_range=[0.1, 0.3], min_memory=25)
    a5 = dict(CFG_AGENT, actions=5, states=20, action_range=[0.0, 0.3], min_memory=25)
    assert sel([a21, a21])[0] == "wave"
    k, ws = sel([a11, dict(a21, min_memory=25), a5], max_steps=25)
    # workspace = LUT image + work counter (160 KiB) + the visit log of every resident wave: [32 episodes][25 steps] words of 8 bytes.
    # 4,096 games fill the persistent grid, so the log's size tells the resident waves per CU: the visit histogram overlays the
    # tables and G is u16, which leaves 14 waves per CU (it was 8 with both beside the tables)
    waves, rest = divmod(ws - 160 * 1024, 32 * 25 * 8)
    assert k == "tuple" and rest == 0 and waves == 14 * 256
    k2, ws2 = sel([a11, dict(a21, min_memory=25)], max_steps=25)                            # two agents, different grids
    waves2, rest2 = divmod(ws2 - 160 * 1024, 32 * 25 * 4)                                   # (their visit-log words are 4 bytes)
    assert k2 == "tuple" and rest2 == 0 and waves2 == 16 * 256                              # (the kernel's 4 waves per SIMD)
    assert sel([dict(a21, min_memory=25)], max_steps=25)[0] == "tuple"                     # one agent
    assert sel([a11, dict(a21, min_memory=25), a5], max_steps=25, noise_prob=0.05)[0] == "tuple"
    assert sel([a11, dict(a21, min_memory=25), a5], max_steps=10)[0] == "generic"           # buffers fill every 3rd episode
    assert sel([a5] * 5, max_steps=25)[0] == "generic"                                      # five agents
    assert sel([a21, a21, dict(a21, action_range=[0.0, 0.1])], max_steps=100)[0] == "generic"   # 9,261 action tuples
    assert b"tuples" in lib.thrl_last_error() or b"tuple" in lib.thrl_last_error()


def test_policy_table_scratch_is_sized_on_the_host(lib):
    """thrl_mixed_policy_table_bytes (host logic): QTable vs Reinforce on the 21 x 21 grid -> LUT region + 442 CDF rows per game
    for the tuple-chain kernel; two Reinforce agents on one grid -> the LUT region only (their tables live in LDS); a continuous
    agent in the game -> nothing."""
    from th_rl_amd import _lib
    q = dict(CFG_AGENT)
    r = dict(name="QTable", states=1, actions=21, action_range=[0.2, 0.4], capacity=1, min_memory=1)     # placeholder slot of a neural agent
    G = 1000
    def need(kinds, agents, buf_len, mm):
        cfg, _ = _lib.cfg_from_config({"agents": agents, "environment": dict(CFG_ENV)}, G, 0)
        mx = _lib.Mixed()
        for i, k in enumerate(kinds):
            mx.kind[i] = k; mx.buf_len[i] = buf_len[i]; mx.min_memory[i] = mm[i]
        return int(lib.thrl_mixed_policy_table_bytes(ctypes.byref(cfg), ctypes.byref(mx)))
    qr = need([0, 1], [q, r], [200, 1100], [100, 1000])
    assert qr == 64 * 1024 + G * 442 * 24 * 4
    rr = need([1, 1], [r, r], [1100, 1100], [1000, 1000])
    assert rr == 64 * 1024
    assert need([0, 3], [q, dict(r, actions=2)], [200, 1100], [100, 1000]) == 0


def test_bench_stdout_carries_only_the_json_line():
    """bench.protect_stdout(): whatever a library writes to file descriptor 1 afterwards (gloo prints its rank banner through C++
    std::cout) lands on stderr; emit() writes the JSON line to the real stdout -- the driver parses exactly one line."""
    import subprocess
    import sys
    code = ("import os, sys, json; sys.path.insert(0, %r); import bench; bench.protect_stdout(); "
            "os.write(1, b'[Gloo] Rank 0 is connected to 1 peer ranks\\n'); print('python-level chatter'); "
            "bench.emit(json.dumps({'metric': 'x', 'value': 1}))") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ['{"metric": "x", "value": 1}'], r.stdout
    assert "Gloo" in r.stderr and "chatter" in r.stderr
