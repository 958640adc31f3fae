"""GPU parity tests of the tuple-chain kernel (thrl_tuple_kernel.h): 1-4 QTable agents with individual grids, one
wavefront per game, tables in LDS.  Through the C ABI; bit for bit against the CPU oracle, the generic kernel and the
reference-generated golden fixtures."""
import glob
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O  # noqa: E402  (checker only)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG_AGENT = dict(name="QTable", gamma=0.95, actions=21, states=100, alpha=0.1, eps_end=0.001,
                 epsilon=0.5, eps_step=0.9995, action_range=[0.2, 0.4])
CFG_ENV = dict(name="NoisyPriceState", noise_prob=0, a=10, b=1, nplayers=2, max_steps=100)
# the shape of golden G5 "three_players": three agents with different grids (trainer.py:21-23 allows any nplayers)
THREE = {"agents": [dict(CFG_AGENT, actions=11, states=50, action_range=[0.1, 0.3], min_memory=25),
                    dict(CFG_AGENT, actions=21, states=100, action_range=[0.15, 0.35], min_memory=25),
                    dict(CFG_AGENT, actions=5, states=20, action_range=[0.0, 0.3], min_memory=25, max_state=10)],
         "environment": dict(CFG_ENV, nplayers=3, max_steps=25)}


def _batch(config, G, dtype="float32", kernel="auto", seed=0, game_offset=0):
    from th_rl_amd.batched import GameBatch
    return GameBatch(config, n_games=G, dtype=dtype, kernel=kernel, seed=seed, game_offset=game_offset)


def _oracle(config, G, dtype, q0, s0, E, seed, first_episode=0, eps=None, game_offset=0):
    cfg, eps0 = O.cfg_from_config(config, n_games=G, q_dtype=1 if dtype == "float64" else 0)
    eps = eps0 if eps is None else eps
    q, s = q0.copy(), s0.copy()
    c = np.zeros(q.shape, np.int32)
    out = O.episodes(cfg, q, c, s, eps, O.Memory(cfg), E, seed=seed, first_episode=first_episode, game_offset=game_offset)
    return q, c, s, eps, out


def _cfg(agents, T, **env):
    return {"agents": agents, "environment": dict(CFG_ENV, nplayers=len(agents), max_steps=T, **env)}


A1 = dict(CFG_AGENT, actions=7, states=30, action_range=[0.1, 0.5], min_memory=10)
A2 = dict(CFG_AGENT, actions=21, states=100, action_range=[0.2, 0.4], min_memory=10, alpha=0.3, gamma=0.9)
A3 = dict(CFG_AGENT, actions=4, states=16, action_range=[0.0, 0.25], min_memory=10, epsilon=0.8)
A4 = dict(CFG_AGENT, actions=5, states=40, action_range=[0.05, 0.2], min_memory=10, max_state=12)
A33 = dict(CFG_AGENT, actions=33, states=64, action_range=[0.0, 0.5], min_memory=10)        # > 32 actions: 3 columns per lane


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("label,config,G,E", [
    ("three_players", THREE, 137, 7),
    ("two_agents_different_grids", _cfg([A1, A2], 40), 70, 5),
    ("one_agent", _cfg([dict(A2, action_range=[0.3, 0.9])], 30), 50, 5),
    ("four_agents", _cfg([A1, A3, A4, A3], 20), 41, 6),
    ("T65_two_segments", _cfg([A1, A3], 65), 33, 4),
    ("T130_three_segments", _cfg([A3, A4, A3], 130, a=10), 21, 3),
    ("T256", _cfg([dict(A3, capacity=600), dict(A4, capacity=600)], 256), 9, 2),
    ("A33_three_columns_per_lane", _cfg([A33, A3], 30), 33, 4),
    ("greedy_regime", _cfg([dict(A1, epsilon=0.0, eps_end=0.0), dict(A2, epsilon=0.02, eps_end=0.02), dict(A3, epsilon=0.0, eps_end=0.0)], 50), 90, 6),
    ("many_episodes_two_launches", THREE, 19, 40),
    # env noise (environments.py:28-31): a step with a redrawn intercept leaves the action grid
    ("three_players_noise05", {"agents": THREE["agents"], "environment": dict(THREE["environment"], noise_prob=0.05)}, 137, 7),
    ("four_agents_noise30", _cfg([A1, A3, A4, A3], 20, noise_prob=0.3), 41, 6),
    ("T65_two_segments_noise50", _cfg([A1, A3], 65, noise_prob=0.5), 33, 4),
    ("T130_three_segments_noise10", _cfg([A3, A4, A3], 130, noise_prob=0.1), 21, 3),
    ("every_step_noisy", _cfg([A1, A2], 40, noise_prob=1.0), 40, 5),
    ("one_agent_noise20", _cfg([dict(A2, action_range=[0.3, 0.9])], 30, noise_prob=0.2), 50, 5),
    ("greedy_regime_noise05", _cfg([dict(A1, epsilon=0.0, eps_end=0.0), dict(A2, epsilon=0.02, eps_end=0.02), dict(A3, epsilon=0.0, eps_end=0.0)], 50, noise_prob=0.05), 90, 6),
    ("many_episodes_two_launches_noise", {"agents": THREE["agents"], "environment": dict(THREE["environment"], noise_prob=0.1)}, 19, 40),
])
def test_tuple_kernel_vs_oracle(label, config, G, E, dtype):
    """Philox draws, both table dtypes: tables, visit counters, env state and epsilon bit-identical to the oracle over two
    calls; mean logs to 1e-12 (sums over steps and games are reordered)."""
    gb = _batch(config, G, dtype=dtype, kernel="tuple", seed=17).init_tables()
    assert gb.planned_kernel() == "tuple"
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    o1 = gb.run(E)
    assert o1["kernel"] == "tuple", label
    q, c, s, eps, oo = _oracle(config, G, dtype, q0, s0, E, seed=17)
    assert np.array_equal(gb.states_numpy(), s), label
    assert np.array_equal(gb.counters_numpy(), c), label
    bad = np.flatnonzero((gb.tables_numpy() != q).any(axis=1))
    assert bad.size == 0, "%s: %d games differ (first %s)" % (label, bad.size, bad[:5])
    np.testing.assert_allclose(o1["reward_log"], oo["reward_log"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(o1["action_log"], oo["action_log"], rtol=1e-12, atol=1e-13)
    N = len(config["agents"])
    assert [float(x) for x in gb.eps[:N]] == [float(x) for x in eps[:N]]
    o2 = gb.run(3)                                         # second call: starts on-grid? no -- from the stored price again
    cfg, _ = O.cfg_from_config(config, n_games=G, q_dtype=1 if dtype == "float64" else 0)
    c2 = c.copy()
    oo2 = O.episodes(cfg, q, c2, s, eps, O.Memory(cfg), 3, seed=17, first_episode=E)
    assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c2) and np.array_equal(gb.states_numpy(), s)
    np.testing.assert_allclose(o2["reward_log"], oo2["reward_log"], rtol=1e-12, atol=1e-13)


def _tuple_eligible(path):
    d = np.load(path)
    c = json.loads(str(d["config_json"]))
    ag, env = c["agents"], c["environment"]
    T = env["max_steps"]
    tuples = int(np.prod([a["actions"] for a in ag]))
    return (len(ag) <= 4 and T <= 256 and tuples <= 4096
            and all(a.get("min_memory", 100) <= T <= a.get("capacity", 500) and a["actions"] <= 64 for a in ag))


def _golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "g4_*.npz")) + glob.glob(os.path.join(GOLDEN, "g5_*.npz")) + glob.glob(os.path.join(GOLDEN, "g6_*.npz")))


def test_golden_coverage_of_the_tuple_kernel():
    """Which reference-generated fixtures the kernel can take (the others have replay buffers that span episodes /
    overflow: the generic kernel's), the noisy ones included."""
    names = [os.path.basename(p) for p in _golden_files() if _tuple_eligible(p)]
    assert "g5_three_players_seed9_e30.npz" in names and any(n.startswith("g4_") for n in names), names
    assert "g5_hetero_noise_seed5_e20.npz" in names, names


@pytest.mark.parametrize("path", [p for p in _golden_files() if _tuple_eligible(p)], ids=os.path.basename)
def test_tuple_f64_injected_matches_reference_golden(path):
    """The reference's recorded draws in => the reference's tables / counters / epsilon / state out, bit for bit (float64
    tables); its logs to 1e-12 (an episode's rewards are summed before dividing by T)."""
    d = np.load(path)
    config = json.loads(str(d["config_json"]))
    E, T, N = d["u"].shape
    gb = _batch(config, 1, dtype="float64", kernel="tuple")
    gb.set_tables(d["init_tables"][None, :], [float(d["state0"])])
    inj = dict(u=d["u"][:, :, :, None], choice=d["choice"][:, :, :, None])
    if config["environment"].get("noise_prob", 0.05) > 0:
        inj.update(noise_u=d["noise_u"][:, :, None], noise_a=d["noise_a"][:, :, None])
    out = gb.run(E, inj=inj)
    assert out["kernel"] == "tuple"
    assert np.array_equal(gb.tables_numpy()[0], d["final_tables"])
    assert np.array_equal(gb.counters_numpy()[0].astype(np.float64), d["final_counters"])
    assert np.array_equal(np.array(gb.eps[:N]), d["eps"][-1])
    assert gb.states_numpy()[0] == d["states"][-1, -1]
    np.testing.assert_allclose(out["reward_log"], d["rewards_log"], rtol=1e-12)
    np.testing.assert_allclose(out["action_log"], d["actions_log"], rtol=1e-12)


def test_auto_selects_the_tuple_kernel_and_refusals():
    from th_rl_amd._lib import ThrlError
    gb = _batch(THREE, 64, kernel="auto").init_tables()
    assert gb.run(2)["kernel"] == "tuple"
    # two agents on one grid stay on the two-agent wave kernel
    cfg2 = {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)], "environment": dict(CFG_ENV)}
    assert _batch(cfg2, 64, kernel="auto").init_tables().run(1)["kernel"] == "wave"
    noisy = {"agents": THREE["agents"], "environment": dict(THREE["environment"], noise_prob=0.05)}
    assert _batch(noisy, 8, kernel="auto").init_tables().run(1)["kernel"] == "tuple"
    # more than 4,096 action tuples: the one-thread-per-game kernel
    big = _cfg([dict(A2, actions=40), dict(A2, actions=40), dict(A3, actions=3)], 20)
    with pytest.raises(ThrlError, match="tuple kernel cannot run"):
        _batch(big, 8, kernel="tuple").init_tables().run(1)
    assert _batch(big, 8, kernel="auto").init_tables().run(1)["kernel"] == "generic"


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_tuple_equals_generic_at_65536_games_three_players(dtype):
    """BASELINE configs[1] size for the three-player shape: tuple kernel == generic kernel on device (tables, counters,
    state), sharding invariance via game_offset, every agent makes exactly E*T visits."""
    import torch
    G, E = 65536, 8
    a = _batch(THREE, G, dtype=dtype, kernel="tuple", seed=5).init_tables()
    b = _batch(THREE, G, dtype=dtype, kernel="generic", seed=5).init_tables()
    ra, rb = a.run(E), b.run(E)
    assert ra["kernel"] == "tuple" and rb["kernel"] == "generic"
    assert torch.equal(a.q, b.q) and torch.equal(a.counter, b.counter) and torch.equal(a.state, b.state)
    np.testing.assert_allclose(ra["reward_log"], rb["reward_log"], rtol=1e-12)
    for i in range(3):
        lo, n = a.offsets[i], a.shapes[i][0] * a.shapes[i][1]
        assert bool((a.counter[:, lo:lo + n].sum(dim=1) == E * 25).all())
    half = _batch(THREE, 1000, dtype=dtype, kernel="tuple", seed=5, game_offset=30000).init_tables()
    half.run(E)
    assert torch.equal(half.q, a.q[30000:31000]) and torch.equal(half.counter, a.counter[30000:31000])


def _random_tuple_config(rs, noise=0.0):
    """A random configuration the tuple-chain kernel must take: 1-4 agents, individual grids, buffers that train
    once per episode, <= 4,096 action tuples."""
    while True:
        n = int(rs.choice([1, 2, 3, 3, 4]))
        T = int(rs.choice([1, 7, 25, 64, 65, 100, 129, 200]))
        acts = [int(rs.choice([2, 3, 5, 8, 11, 16, 17, 21, 33, 40])) for _ in range(n)]
        if int(np.prod(acts)) <= 4096:
            break
    agents = []
    for i in range(n):
        lo = float(np.round(rs.uniform(0.0, 0.3), 2))
        agents.append(dict(name="QTable", gamma=float(rs.choice([0.35, 0.9, 0.95, 0.99])), actions=acts[i],
                           states=int(rs.choice([4, 16, 50, 100, 200])), alpha=float(rs.choice([0.05, 0.1, 0.5])),
                           eps_end=float(rs.choice([0.0, 0.001, 0.05])), epsilon=float(rs.choice([0.0, 0.05, 0.5, 1.0])), eps_step=0.99,
                           action_range=[lo, float(np.round(lo + rs.uniform(0.02, 0.6 / n), 2))],
                           max_state=float(rs.choice([10, 10, 12])), min_memory=int(rs.choice([0, 1, max(1, T // 2), T])),
                           capacity=int(rs.choice([T, T + 3, 500]))))
    return {"agents": agents, "environment": dict(name="NoisyPriceState", noise_prob=noise, a=10, b=1, nplayers=n, max_steps=T)}


@pytest.mark.parametrize("case", range(24))
def test_tuple_kernel_fuzz_vs_oracle(case):
    rs = np.random.RandomState(7000 + case)
    config = _random_tuple_config(rs)
    dtype = "float64" if case % 3 == 0 else "float32"
    G, E = int(rs.randint(1, 70)), int(rs.choice([1, 3, 9, 35]))
    if config["environment"]["max_steps"] * G * E > 400000:
        E = 3
    seed, off = int(rs.randint(0, 10 ** 6)), int(rs.randint(0, 1 << 33))
    gb = _batch(config, G, dtype=dtype, kernel="auto", seed=seed, game_offset=off).init_tables()
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    out = gb.run(E)
    label = "%s %s G=%d E=%d -> %s" % (json.dumps(config)[:300], dtype, G, E, out["kernel"])
    two_same = len(config["agents"]) == 2 and all(config["agents"][0][k] == config["agents"][1][k] for k in ("actions", "states", "max_state"))
    assert out["kernel"] == ("wave" if two_same and config["agents"][0]["actions"] <= 32 else "tuple"), label
    q, c, s, eps, oo = _oracle(config, G, dtype, q0, s0, E, seed=seed, game_offset=off)
    assert np.array_equal(gb.tables_numpy(), q), label
    assert np.array_equal(gb.counters_numpy(), c) and np.array_equal(gb.states_numpy(), s), label
    np.testing.assert_allclose(out["reward_log"], oo["reward_log"], rtol=1e-12, atol=1e-13, err_msg=label)
    np.testing.assert_allclose(out["action_log"], oo["action_log"], rtol=1e-12, atol=1e-13, err_msg=label)


@pytest.mark.parametrize("case", range(16))
def test_tuple_kernel_noise_fuzz_vs_oracle(case):
    """Random configurations with env noise, kernel forced: bit-identical to the oracle."""
    rs = np.random.RandomState(9100 + case)
    config = _random_tuple_config(rs, noise=float(rs.choice([0.01, 0.05, 0.2, 0.5, 1.0])))
    dtype = "float64" if case % 3 == 1 else "float32"
    G, E = int(rs.randint(1, 70)), int(rs.choice([1, 3, 9, 35]))
    if config["environment"]["max_steps"] * G * E > 400000:
        E = 3
    seed, off = int(rs.randint(0, 10 ** 6)), int(rs.randint(0, 1 << 33))
    gb = _batch(config, G, dtype=dtype, kernel="tuple", seed=seed, game_offset=off).init_tables()
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    out = gb.run(E)
    label = "%s %s G=%d E=%d -> %s" % (json.dumps(config)[:300], dtype, G, E, out["kernel"])
    assert out["kernel"] == "tuple", label
    q, c, s, eps, oo = _oracle(config, G, dtype, q0, s0, E, seed=seed, game_offset=off)
    assert np.array_equal(gb.tables_numpy(), q), label
    assert np.array_equal(gb.counters_numpy(), c) and np.array_equal(gb.states_numpy(), s), label
    np.testing.assert_allclose(out["reward_log"], oo["reward_log"], rtol=1e-12, atol=1e-13, err_msg=label)
    np.testing.assert_allclose(out["action_log"], oo["action_log"], rtol=1e-12, atol=1e-13, err_msg=label)


def test_tuple_equals_generic_at_65536_games_three_players_with_noise():
    """The reference's default noise_prob (0.05) at BASELINE configs[1] size: tuple kernel == generic kernel on device."""
    import torch
    noisy = {"agents": THREE["agents"], "environment": dict(THREE["environment"], noise_prob=0.05)}
    G, E = 65536, 8
    a = _batch(noisy, G, kernel="tuple", seed=5).init_tables()
    b = _batch(noisy, G, kernel="generic", seed=5).init_tables()
    ra, rb = a.run(E), b.run(E)
    assert ra["kernel"] == "tuple" and rb["kernel"] == "generic"
    assert torch.equal(a.q, b.q) and torch.equal(a.counter, b.counter) and torch.equal(a.state, b.state)
    np.testing.assert_allclose(ra["reward_log"], rb["reward_log"], rtol=1e-12)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("noise", [0.0, 0.1])
def test_tuple_kernel_sweeps_vs_oracle(dtype, noise):
    """Per-game sweeps (gamma, alpha, epsilon schedule, noise_prob: thrl_buffers.sweep_*) on the tuple-chain kernel: three
    players, 40 episodes = two launches (a game's epsilon survives in sweep_eps), then 3 more; bit for bit against the oracle
    given the same arrays, and against the generic kernel."""
    rs = np.random.RandomState(11)
    config = {"agents": THREE["agents"], "environment": dict(THREE["environment"], noise_prob=noise)}
    G, E, N = 29, 40, 3
    sweep = dict(gamma=rs.choice([0.35, 0.9, 0.95], (N, G)), alpha=rs.choice([0.05, 0.1, 0.5], (N, G)),
                 eps=rs.uniform(0.0, 0.9, (N, G)), eps_end=rs.choice([0.0, 0.01], (N, G)), eps_step=rs.choice([0.9, 0.999], (N, G)))
    if noise > 0:
        sweep["noise_prob"] = rs.choice([0.0, 0.05, 0.5, 1.0], G)
    gb = _batch(config, G, dtype=dtype, kernel="auto", seed=31)
    gb.set_sweep(sweep); gb.init_tables()
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    o1 = gb.run(E); o2 = gb.run(3)
    assert o1["kernel"] == "tuple" and o2["kernel"] == "tuple"
    cfg, eps0 = O.cfg_from_config(config, n_games=G, q_dtype=1 if dtype == "float64" else 0)
    q, st, cn, mem = q0.copy(), s0.copy(), np.zeros(q0.shape, np.int32), O.Memory(cfg)
    osw = {k: np.array(v, np.float64) for k, v in sweep.items()}      # copies: the oracle updates eps in place
    eps_start = eps0.copy()
    oo1 = O.episodes(cfg, q, cn, st, eps0, mem, E, seed=31, sweep=osw)
    oo2 = O.episodes(cfg, q, cn, st, eps0, mem, 3, seed=31, first_episode=E, sweep=osw)
    assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), cn)
    assert np.array_equal(gb.states_numpy(), st)
    assert np.array_equal(gb.sweep["eps"].cpu().numpy(), osw["eps"])
    np.testing.assert_allclose(o1["reward_log"], oo1["reward_log"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(o2["action_log"], oo2["action_log"], rtol=1e-12, atol=1e-13)
    gen = _batch(config, G, dtype=dtype, kernel="generic", seed=31)
    gen.set_sweep(sweep); gen.set_tables(q0, s0)
    assert gen.run(E)["kernel"] == "generic"
    gen.run(3)
    assert np.array_equal(gen.tables_numpy(), q) and np.array_equal(gen.sweep["eps"].cpu().numpy(), osw["eps"])
    # sweeps of gamma / alpha only (no per-game epsilon array): epsilon follows the host's schedule
    part = _batch(config, G, dtype=dtype, kernel="tuple", seed=31)
    part.set_sweep(dict(gamma=sweep["gamma"], alpha=sweep["alpha"])); part.set_tables(q0, s0)
    assert part.run(E)["kernel"] == "tuple"
    q2, st2, cn2 = q0.copy(), s0.copy(), np.zeros(q0.shape, np.int32)
    O.episodes(cfg, q2, cn2, st2, eps_start, O.Memory(cfg), E, seed=31, sweep=dict(gamma=osw["gamma"], alpha=osw["alpha"]))
    assert np.array_equal(part.tables_numpy(), q2) and np.array_equal(part.counters_numpy(), cn2)


@pytest.mark.parametrize("label,dtype,noise", [("f32", "float32", 0.0), ("f32_noise05", "float32", 0.05), ("f64", "float64", 0.0)])
def test_tuple_65536_games_one_32_episode_launch_oracle_slices(label, dtype, noise):
    """The measured shape (profiles/exp_tuple.py: three players x 65,536 games, ONE launch of 32 episodes, counters on; every
    resident wave plays ~18 games, so the visit log and the histogram overlay are reused game after game), then four slices of
    1,024 games against the oracle -- Philox is keyed by the global game id, so a slice is a run of its own with game_offset =
    its first game: tables, visit counters and env state bit for bit; every agent's counters sum to the transitions played."""
    config = {"agents": THREE["agents"], "environment": dict(THREE["environment"], noise_prob=noise)}
    G, E, n = 65536, 32, 1024
    gb = _batch(config, G, dtype=dtype, kernel="tuple", seed=5, game_offset=1 << 22).init_tables()
    slices = [(lo, n) for lo in (0, 21000 + 5, 43210, G - n)]
    q_init = [gb.q[lo:lo + n].cpu().numpy() for lo, _ in slices]
    s_init = [gb.state[lo:lo + n].cpu().numpy() for lo, _ in slices]
    out = gb.run(E)
    assert out["kernel"] == "tuple"
    for (lo, _), q0, s0 in zip(slices, q_init, s_init):
        q, c, s, eps, oo = _oracle(config, n, dtype, q0, s0, E, seed=5, game_offset=gb.game_offset + lo)
        assert np.array_equal(gb.q[lo:lo + n].cpu().numpy(), q), "tables of games %d..%d" % (lo, lo + n)
        assert np.array_equal(gb.counter[lo:lo + n].cpu().numpy(), c), "counters of games %d..%d" % (lo, lo + n)
        assert np.array_equal(gb.state[lo:lo + n].cpu().numpy(), s), "states of games %d..%d" % (lo, lo + n)
    assert bool((gb.counter.sum(dim=1) == 3 * E * 25).all())
