"""GPU parity tests: the HIP path (through the C ABI) vs the CPU oracle and the
reference-generated golden fixtures.  Integer/index results and float64 tables
are compared bit for bit; float32 tables are compared bit for bit against the
oracle's float32 mode (same op order), and against the float64 reference within
rtol 1e-5 where stated."""
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
CFG = {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)], "environment": dict(CFG_ENV)}


def _batch(config, G, dtype="float32", kernel="auto", seed=0, game_offset=0):
    from th_rl_amd.batched import GameBatch
    return GameBatch(config, n_games=G, dtype=dtype, kernel=kernel, seed=seed, game_offset=game_offset)


def _oracle_run(config, G, q_dtype, q, state, E, seed=0, game_offset=0, first_episode=0, eps=None, mem=None):
    cfg, eps0 = O.cfg_from_config(config, n_games=G, q_dtype=q_dtype)
    eps = eps0 if eps is None else eps
    mem = O.Memory(cfg) if mem is None else mem
    q = q.copy(); state = state.copy()
    counter = np.zeros(q.shape, np.int32)
    out = O.episodes(cfg, q, counter, state, eps, mem, E, seed=seed, game_offset=game_offset,
                     first_episode=first_episode)
    return q, counter, state, eps, mem, out


def _traj_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "g4_*.npz")) + glob.glob(os.path.join(GOLDEN, "g5_*.npz")) + glob.glob(os.path.join(GOLDEN, "g6_*.npz")))


@pytest.mark.parametrize("path", _traj_files(), ids=os.path.basename)
def test_generic_f64_injected_matches_reference_golden(path):
    """The reference's recorded draws in => the reference's tables/counters/logs out,
    bit for bit (float64 generic kernel)."""
    d = np.load(path)
    config = json.loads(str(d["config_json"]))
    E, T, N = d["u"].shape
    gb = _batch(config, 1, dtype="float64", kernel="generic")
    gb.set_tables(d["init_tables"][None, :], [float(d["state0"])])
    inj = dict(u=d["u"][:, :, :, None], choice=d["choice"][:, :, :, None])
    if gb.cfg.noise_prob > 0:
        inj.update(noise_u=d["noise_u"][:, :, None], noise_a=np.nan_to_num(d["noise_a"][:, :, None]))
    out = gb.run(E, inj=inj, per_game_logs=True)
    assert out["kernel"] == "generic"
    assert np.array_equal(gb.tables_numpy()[0], d["final_tables"])
    assert np.array_equal(gb.counters_numpy()[0].astype(np.float64), d["final_counters"])
    assert np.array_equal(out["game_reward_log"][:, :, 0], d["rewards_log"])
    assert np.array_equal(out["game_action_log"][:, :, 0], d["actions_log"])
    assert np.array_equal(np.array(gb.eps[:N]), d["eps"][-1])
    assert gb.states_numpy()[0] == d["states"][-1, -1]
    np.testing.assert_allclose(out["reward_log"], d["rewards_log"], rtol=1e-14)


def _wave_eligible(path):
    d = np.load(path)
    c = json.loads(str(d["config_json"]))
    ag, T = c["agents"], c["environment"]["max_steps"]
    same = all(ag[0].get(k) == ag[1].get(k) for k in ("states", "actions", "max_state")) if len(ag) == 2 else False
    return (len(ag) == 2 and same and ag[0]["actions"] <= 32 and T <= 256
            and all(a.get("min_memory", 100) <= T <= a.get("capacity", 500) for a in ag))


@pytest.mark.parametrize("path", [p for p in _traj_files() if _wave_eligible(p)], ids=os.path.basename)
def test_wave_f64_injected_matches_reference_golden(path):
    """The reference's own numerics (float64 tables) on the LDS-resident one-wavefront-per-game kernel:
    the reference's recorded draws in => the reference's tables / counters / epsilon / state out, bit
    for bit; its logs to 1e-12 (the kernel sums an episode's rewards before dividing by T)."""
    d = np.load(path)
    config = json.loads(str(d["config_json"]))
    E, T, N = d["u"].shape
    gb = _batch(config, 1, dtype="float64", kernel="wave")
    gb.set_tables(d["init_tables"][None, :], [float(d["state0"])])
    inj = dict(u=d["u"][:, :, :, None], choice=d["choice"][:, :, :, None])
    if gb.cfg.noise_prob > 0:
        inj.update(noise_u=d["noise_u"][:, :, None], noise_a=np.nan_to_num(d["noise_a"][:, :, None]))
    out = gb.run(E, inj=inj)
    assert out["kernel"] == "wave"
    assert np.array_equal(gb.tables_numpy()[0], d["final_tables"])
    assert np.array_equal(gb.counters_numpy()[0].astype(np.float64), d["final_counters"])
    assert np.array_equal(np.array(gb.eps[:N]), d["eps"][-1])
    assert gb.states_numpy()[0] == d["states"][-1, -1]
    np.testing.assert_allclose(out["reward_log"], d["rewards_log"], rtol=1e-12)
    np.testing.assert_allclose(out["action_log"], d["actions_log"], rtol=1e-12)


def test_wave_f64_four_reference_runs_and_generic_kernel_at_size():
    """float64 wave kernel: four different reference runs as one batch of four games (exact), and at
    20,000 games it equals the float64 generic kernel bit for bit (Philox draws, 3 + 2 episodes)."""
    ds = [np.load(os.path.join(GOLDEN, "g4_cfg_seed%d_e12.npz" % s)) for s in range(4)]
    config = json.loads(str(ds[0]["config_json"]))
    gb = _batch(config, 4, dtype="float64", kernel="wave")
    gb.set_tables(np.stack([d["init_tables"] for d in ds]), [float(d["state0"]) for d in ds])
    inj = dict(u=np.stack([d["u"] for d in ds], axis=-1), choice=np.stack([d["choice"] for d in ds], axis=-1))
    out = gb.run(12, inj=inj)
    assert out["kernel"] == "wave"
    for g, d in enumerate(ds):
        assert np.array_equal(gb.tables_numpy()[g], d["final_tables"])
        assert np.array_equal(gb.counters_numpy()[g].astype(np.float64), d["final_counters"])
    np.testing.assert_allclose(out["reward_log"], np.mean([d["rewards_log"] for d in ds], axis=0), rtol=1e-12)
    G = 20000
    a = _batch(CFG, G, dtype="float64", kernel="wave", seed=5).init_tables()
    b = _batch(CFG, G, dtype="float64", kernel="generic", seed=5).init_tables()
    ra = a.run(3); a.run(2); rb = b.run(5)
    assert ra["kernel"] == "wave" and rb["kernel"] == "generic"
    assert np.array_equal(a.tables_numpy(), b.tables_numpy()) and np.array_equal(a.counters_numpy(), b.counters_numpy())
    assert np.array_equal(a.states_numpy(), b.states_numpy())
    np.testing.assert_allclose(ra["reward_log"], rb["reward_log"][:3], rtol=1e-12)


def test_generic_f64_injected_four_games_at_once():
    """Four different reference runs stepped in lockstep as one batch (G=4)."""
    ds = [np.load(os.path.join(GOLDEN, "g4_cfg_seed%d_e12.npz" % s)) for s in range(4)]
    config = json.loads(str(ds[0]["config_json"]))
    gb = _batch(config, 4, dtype="float64", kernel="generic")
    gb.set_tables(np.stack([d["init_tables"] for d in ds]), [float(d["state0"]) for d in ds])
    inj = dict(u=np.stack([d["u"] for d in ds], axis=-1), choice=np.stack([d["choice"] for d in ds], axis=-1))
    out = gb.run(12, inj=inj, per_game_logs=True)
    for g, d in enumerate(ds):
        assert np.array_equal(gb.tables_numpy()[g], d["final_tables"])
        assert np.array_equal(gb.counters_numpy()[g].astype(np.float64), d["final_counters"])
        assert np.array_equal(out["game_reward_log"][:, :, g], d["rewards_log"])
    np.testing.assert_allclose(out["reward_log"], np.mean([d["rewards_log"] for d in ds], axis=0), rtol=1e-13)


def test_generic_f32_injected_tracks_reference_within_tolerance():
    """float32 tables vs the float64 reference: same actions for the first 5
    episodes, Q-values within rtol 1e-5 (the tolerance north_star allows)."""
    d = np.load(os.path.join(GOLDEN, "g4_cfg_seed0_e12.npz"))
    config = json.loads(str(d["config_json"]))
    E = 5
    gb = _batch(config, 1, dtype="float32", kernel="generic")
    gb.set_tables(d["init_tables"][None, :], [float(d["state0"])])
    out = gb.run(E, inj=dict(u=d["u"][:E, :, :, None], choice=d["choice"][:E, :, :, None]), per_game_logs=True)
    assert np.array_equal(out["game_reward_log"][:, :, 0], d["rewards_log"][:E])   # same trajectory
    # float64 reference tables after 5 episodes come from the oracle (pinned to the reference)
    cfg, eps = O.cfg_from_config(config, 1, 1)
    q = d["init_tables"][None, :].copy(); c = np.zeros(q.shape, np.int32); s = np.array([float(d["state0"])])
    O.episodes(cfg, q, c, s, eps, O.Memory(cfg), E, inj_u=np.ascontiguousarray(d["u"][:E, :, :, None]),
               inj_choice=np.ascontiguousarray(d["choice"][:E, :, :, None]))
    np.testing.assert_allclose(gb.tables_numpy()[0].astype(np.float64), q[0], rtol=1e-5, atol=0)
    assert np.array_equal(gb.counters_numpy()[0], c[0])


def test_wave_f32_injected_reference_draws():
    """The FAST kernel on the reference's own recorded draws (4 reference runs as 4 games):
    bit-identical to the oracle's float32 mode for all 12 episodes, and against the float64
    reference itself: identical reward logs (same trajectory) for the first 5 episodes and
    Q-values within rtol 1e-5."""
    ds = [np.load(os.path.join(GOLDEN, "g4_cfg_seed%d_e12.npz" % s)) for s in range(4)]
    config = json.loads(str(ds[0]["config_json"]))
    q0 = np.stack([d["init_tables"] for d in ds]); s0 = np.array([float(d["state0"]) for d in ds])
    inj_u = np.ascontiguousarray(np.stack([d["u"] for d in ds], axis=-1))
    inj_c = np.ascontiguousarray(np.stack([d["choice"] for d in ds], axis=-1))
    gb = _batch(config, 4, dtype="float32", kernel="wave")
    gb.set_tables(q0, s0)
    out5 = gb.run(5, inj=dict(u=inj_u[:5], choice=inj_c[:5]))
    assert out5["kernel"] == "wave"
    q5 = gb.tables_numpy().copy()
    out7 = gb.run(7, inj=dict(u=inj_u[5:], choice=inj_c[5:]))
    cfg, eps = O.cfg_from_config(config, 4, 0)
    q = q0.astype(np.float32); c = np.zeros(q.shape, np.int32); s = s0.copy()
    oo = O.episodes(cfg, q, c, s, eps, O.Memory(cfg), 12, inj_u=inj_u, inj_choice=inj_c)
    assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c)
    assert np.array_equal(gb.states_numpy(), s)
    np.testing.assert_allclose(np.concatenate([out5["reward_log"], out7["reward_log"]]), oo["reward_log"], rtol=1e-12)
    # against the float64 reference itself
    ref_log5 = np.mean([d["rewards_log"][:5] for d in ds], axis=0)
    np.testing.assert_allclose(out5["reward_log"], ref_log5, rtol=1e-12)
    cfg64, eps64 = O.cfg_from_config(config, 4, 1)
    q64 = q0.copy(); c64 = np.zeros(q64.shape, np.int32); s64 = s0.copy()
    O.episodes(cfg64, q64, c64, s64, eps64, O.Memory(cfg64), 5, inj_u=np.ascontiguousarray(inj_u[:5]),
               inj_choice=np.ascontiguousarray(inj_c[:5]))
    np.testing.assert_allclose(q5.astype(np.float64), q64, rtol=1e-5, atol=0)


def test_init_matches_oracle():
    gb = _batch(CFG, 50, dtype="float64", seed=11, game_offset=7).init_tables()
    cfg, _ = O.cfg_from_config(CFG, 50, 1)
    q, c, s = O.init(cfg, seed=11, game_offset=7)
    assert np.array_equal(gb.states_numpy(), s)                   # exact integer->double arithmetic
    np.testing.assert_allclose(gb.tables_numpy(), q, rtol=0, atol=1e-11)   # libm log/sin/cos differ in ulps
    assert not gb.counters_numpy().any()
    tq = gb.tables_numpy()
    assert abs(tq.mean() - 250.0) < 0.02 and abs(tq.std() - 1.0) < 0.02


@pytest.mark.parametrize("kernel,G,E", [("generic", 70, 5), ("wave", 70, 5), ("wave", 333, 20)])
def test_f32_philox_bit_exact_vs_oracle(kernel, G, E):
    """Philox draws, float32 tables: tables, counters, env state bit-identical to
    the oracle's float32 mode; mean logs to 1e-12 (summation order over games)."""
    gb = _batch(CFG, G, dtype="float32", kernel=kernel, seed=5).init_tables()
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    out = gb.run(E)
    assert out["kernel"] == kernel
    q, c, s, eps, mem, oo = _oracle_run(CFG, G, 0, q0, s0, E, seed=5)
    assert np.array_equal(gb.counters_numpy(), c)
    assert np.array_equal(gb.states_numpy(), s)
    assert np.array_equal(gb.tables_numpy(), q)
    np.testing.assert_allclose(out["reward_log"], oo["reward_log"], rtol=1e-12)
    np.testing.assert_allclose(out["action_log"], oo["action_log"], rtol=1e-12)
    assert np.array_equal(np.array(gb.eps[:2]), eps[:2])


def test_wave_equals_generic_and_split_calls():
    """wave kernel == generic kernel bit for bit, and 7+9 episodes in two calls ==
    16 in one (tables stay resident in LDS only inside a launch)."""
    G = 257
    a = _batch(CFG, G, kernel="wave", seed=9).init_tables()
    b = _batch(CFG, G, kernel="generic", seed=9).init_tables()
    c = _batch(CFG, G, kernel="wave", seed=9).init_tables()
    ra = a.run(16); rb = b.run(16); c.run(7); c.run(9)
    assert ra["kernel"] == "wave" and rb["kernel"] == "generic"
    for x in (b, c):
        assert np.array_equal(a.tables_numpy(), x.tables_numpy())
        assert np.array_equal(a.counters_numpy(), x.counters_numpy())
        assert np.array_equal(a.states_numpy(), x.states_numpy())
    np.testing.assert_allclose(ra["reward_log"], rb["reward_log"], rtol=1e-12)


def test_sharding_invariance():
    """game_offset makes results independent of how games are split over GPUs."""
    full = _batch(CFG, 96, kernel="wave", seed=2).init_tables(); full.run(4)
    lo = _batch(CFG, 40, kernel="wave", seed=2, game_offset=0).init_tables(); lo.run(4)
    hi = _batch(CFG, 56, kernel="wave", seed=2, game_offset=40).init_tables(); hi.run(4)
    assert np.array_equal(full.tables_numpy(), np.concatenate([lo.tables_numpy(), hi.tables_numpy()]))
    assert np.array_equal(full.counters_numpy(), np.concatenate([lo.counters_numpy(), hi.counters_numpy()]))


HETERO = {"agents": [dict(CFG_AGENT, gamma=0.35, alpha=0.5, epsilon=0.8), dict(CFG_AGENT)],
          "environment": dict(CFG_ENV, noise_prob=0.05)}
BUFFER = {"agents": [dict(CFG_AGENT, min_memory=70, capacity=90), dict(CFG_AGENT, min_memory=100, capacity=80)],
          "environment": dict(CFG_ENV, max_steps=30)}
THREE = {"agents": [dict(CFG_AGENT, actions=11, states=50, action_range=[0.1, 0.3], min_memory=25),
                    dict(CFG_AGENT, actions=21, states=100, action_range=[0.15, 0.35], min_memory=25),
                    dict(CFG_AGENT, actions=5, states=20, action_range=[0.0, 0.3], min_memory=25, max_state=10)],
         "environment": dict(CFG_ENV, nplayers=3, max_steps=25)}
CAP64 = {"agents": [dict(CFG_AGENT, min_memory=20, capacity=64), dict(CFG_AGENT)], "environment": dict(CFG_ENV)}


@pytest.mark.parametrize("name,config,dtype", [
    ("hetero_noise", HETERO, "float64"), ("hetero_noise", HETERO, "float32"),
    ("buffer_T30", BUFFER, "float64"), ("three_players", THREE, "float64"), ("capacity64", CAP64, "float32")])
def test_generic_philox_other_configs_vs_oracle(name, config, dtype):
    """Noise, heterogeneous agents, 3 players with different grids, buffers that
    span episodes / overflow: generic kernel vs oracle, bit for bit, two calls."""
    G, qd = 37, (1 if dtype == "float64" else 0)
    gb = _batch(config, G, dtype=dtype, kernel="generic", seed=4).init_tables()
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    o1 = gb.run(5, per_game_logs=True)
    o2 = gb.run(4, per_game_logs=True)
    q, c, s, eps, mem, oo1 = _oracle_run(config, G, qd, q0, s0, 5, seed=4)
    cfg, _ = O.cfg_from_config(config, G, qd)
    oo2 = O.episodes(cfg, q, c, s, eps, mem, 4, seed=4, first_episode=5)
    assert np.array_equal(gb.tables_numpy(), q)
    assert np.array_equal(gb.counters_numpy(), c)
    assert np.array_equal(gb.states_numpy(), s)
    assert np.array_equal(o1["game_reward_log"], oo1["game_reward_log"])
    assert np.array_equal(o2["game_action_log"], oo2["game_action_log"])
    assert list(gb.mem_count[:gb.N]) == list(mem.count[:gb.N])


NOISY = {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)], "environment": dict(CFG_ENV, noise_prob=0.05)}
NOISY_HI = {"agents": [dict(CFG_AGENT, action_range=[0.1, 0.6]), dict(CFG_AGENT, gamma=0.35, alpha=0.5, epsilon=0.8)],
            "environment": dict(CFG_ENV, noise_prob=0.5)}


@pytest.mark.parametrize("name,config,G,E", [("hetero_noise", HETERO, 70, 5), ("noise05", NOISY, 333, 20),
                                             ("noise50_ranges", NOISY_HI, 129, 7)])
def test_wave_noise_philox_bit_exact_vs_oracle(name, config, G, E):
    """Environment noise (environments.py:28-31) and per-agent hyper-parameters / action
    ranges on the fused wave kernel: tables, counters, state bit-identical to the oracle."""
    gb = _batch(config, G, dtype="float32", kernel="wave", seed=6).init_tables()
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    out = gb.run(E)
    assert out["kernel"] == "wave"
    q, c, s, eps, mem, oo = _oracle_run(config, G, 0, q0, s0, E, seed=6)
    assert np.array_equal(gb.counters_numpy(), c)
    assert np.array_equal(gb.states_numpy(), s)
    assert np.array_equal(gb.tables_numpy(), q)
    np.testing.assert_allclose(out["reward_log"], oo["reward_log"], rtol=1e-12)
    np.testing.assert_allclose(out["action_log"], oo["action_log"], rtol=1e-12)
    # and identical to the generic kernel
    gg = _batch(config, G, dtype="float32", kernel="generic", seed=6); gg.set_tables(q0, s0); gg.run(E)
    assert np.array_equal(gg.tables_numpy(), q) and np.array_equal(gg.counters_numpy(), c)


def test_wave_noise_injected_reference_run():
    """The reference's recorded noisy run (noise_prob=0.05, heterogeneous agents, golden G5)
    on the wave kernel: bit-identical to the oracle's float32 mode over all 20 episodes; same
    reward log as the float64 reference while the float32 trajectory has not forked (>= 3 episodes)."""
    d = np.load(os.path.join(GOLDEN, "g5_hetero_noise_seed5_e20.npz"))
    config = json.loads(str(d["config_json"]))
    E = d["u"].shape[0]
    inj = dict(u=np.ascontiguousarray(d["u"][:, :, :, None]), choice=np.ascontiguousarray(d["choice"][:, :, :, None]),
               noise_u=np.ascontiguousarray(d["noise_u"][:, :, None]),
               noise_a=np.ascontiguousarray(np.nan_to_num(d["noise_a"][:, :, None])))
    gb = _batch(config, 1, dtype="float32", kernel="wave")
    gb.set_tables(d["init_tables"][None, :], [float(d["state0"])])
    out = gb.run(E, inj=inj)
    assert out["kernel"] == "wave"
    cfg, eps = O.cfg_from_config(config, 1, 0)
    q = d["init_tables"][None, :].astype(np.float32); c = np.zeros(q.shape, np.int32); s = np.array([float(d["state0"])])
    oo = O.episodes(cfg, q, c, s, eps, O.Memory(cfg), E, inj_u=inj["u"], inj_choice=inj["choice"],
                    inj_noise_u=inj["noise_u"], inj_noise_a=inj["noise_a"])
    assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c)
    assert np.array_equal(gb.states_numpy(), s)
    np.testing.assert_allclose(out["reward_log"], oo["reward_log"], rtol=1e-12)
    same = [np.allclose(out["reward_log"][e], d["rewards_log"][e], rtol=1e-12) for e in range(E)]
    assert all(same[:3]), same


def _shape_config(T, A=21, states=100, rng=(0.2, 0.4), noise=0.0, cap=500):
    ag = dict(CFG_AGENT, actions=A, states=states, action_range=list(rng), min_memory=min(T, 100), capacity=cap)
    return {"agents": [dict(ag), dict(ag, alpha=0.3)], "environment": dict(CFG_ENV, max_steps=T, noise_prob=noise)}


@pytest.mark.parametrize("label,config,G,E", [
    ("T1", _shape_config(1), 40, 17),                      # NSEG=1, single step, 17 episodes = 16 + 1 chunks
    ("T37", _shape_config(37), 33, 5),
    ("T64", _shape_config(64), 33, 5),                     # exactly one segment
    ("T65", _shape_config(65), 33, 5),                     # one step into the second segment
    ("T128", _shape_config(128), 21, 3),
    ("T129", _shape_config(129), 21, 3),                   # NSEG=3
    ("T200", _shape_config(200), 13, 3),                   # NSEG=4
    ("T256", _shape_config(256), 9, 2),
    ("A2", _shape_config(50, A=2), 33, 6),                 # smallest action grid
    ("A32", _shape_config(50, A=32), 33, 4),               # all 32 lanes of a half
    ("fullwindow", _shape_config(100, rng=(0.0, 0.5)), 19, 4),          # 101 reachable rows: NRSEG=2
    ("fullwindow_noise_T130", _shape_config(130, rng=(0.0, 0.5), noise=0.2), 11, 3),
    ("states16", _shape_config(30, A=4, states=16, rng=(0.0, 1.0)), 50, 6),   # QTable ctor defaults grid
    ("one_game", _shape_config(100), 1, 3),
])
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_wave_shapes_vs_oracle(label, config, G, E, dtype):
    """Every template variant of the wave kernel (episode length 1..256 -> 1-4 step segments,
    1-2 row segments, 2..32 actions = 1-4 row reads per lane in the replay passes, noise; float32 and
    float64 tables) against the oracle in the same dtype, bit for bit."""
    gb = _batch(config, G, dtype=dtype, kernel="wave", seed=12).init_tables()
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    out = gb.run(E)
    assert out["kernel"] == "wave"
    q, c, s, eps, mem, oo = _oracle_run(config, G, 1 if dtype == "float64" else 0, q0, s0, E, seed=12)
    assert np.array_equal(gb.counters_numpy(), c)
    assert np.array_equal(gb.states_numpy(), s)
    assert np.array_equal(gb.tables_numpy(), q)
    np.testing.assert_allclose(out["reward_log"], oo["reward_log"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(out["action_log"], oo["action_log"], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("eps", [0.0, 0.02])
def test_wave_greedy_regime_cycles_and_fixed_points_vs_oracle(dtype, eps):
    """Late-training regime: (almost) always greedy, so games sit in fixed points and short cycles --
    the replay schedule cuts every group into serial passes and takes the four-identical-transitions
    path (one row read, four dependent updates in registers).  Bit for bit against the oracle."""
    ag = dict(CFG_AGENT, epsilon=eps, eps_end=eps)
    config = {"agents": [dict(ag), dict(ag, alpha=0.3)], "environment": dict(CFG_ENV)}
    G, E = 300, 7
    gb = _batch(config, G, dtype=dtype, kernel="wave", seed=77).init_tables()
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    out = gb.run(E)
    assert out["kernel"] == "wave"
    q, c, s, e2, mem, oo = _oracle_run(config, G, 1 if dtype == "float64" else 0, q0, s0, E, seed=77)
    assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c)
    assert np.array_equal(gb.states_numpy(), s)
    # the regime is what the test is for: most games repeat one cell many times within an episode
    assert (c.max(axis=1) >= 50).mean() > 0.3


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_wave_greedy_variant_equals_generic_kernel_at_16384_games(dtype):
    """The GREEDY variants of the wave kernel (launched once epsilon <= 0.05: all-greedy groups without a table
    build, period-2 passes, cyclic segments of period 1-4 as register recurrences) against the generic kernel --
    independent code, one thread per game, tables in HBM -- on 16,384 games over 40 episodes that start at
    epsilon 0.02: tables, counters, states and epsilon bit for bit; and the regime really is the cyclic one."""
    ag = dict(CFG_AGENT, epsilon=0.02, eps_end=0.001)
    config = {"agents": [dict(ag), dict(ag, alpha=0.3)], "environment": dict(CFG_ENV)}
    G, E = 16384, 40
    a = _batch(config, G, dtype=dtype, kernel="wave", seed=5).init_tables()
    b = _batch(config, G, dtype=dtype, kernel="generic", seed=5).init_tables()
    assert np.array_equal(a.tables_numpy(), b.tables_numpy())
    for n in (25, 15):                       # two launches: the second starts on tables the first one trained
        oa, ob = a.run(n), b.run(n)
        assert oa["kernel"] == "wave" and ob["kernel"] == "generic"
    assert np.array_equal(a.tables_numpy(), b.tables_numpy())
    assert np.array_equal(a.counters_numpy(), b.counters_numpy())
    assert np.array_equal(a.states_numpy(), b.states_numpy())
    assert a.eps == b.eps
    np.testing.assert_allclose(oa["reward_log"], ob["reward_log"], rtol=1e-12)
    c = a.counters_numpy()
    assert (c.max(axis=1) >= 400).mean() > 0.05         # a good share of the games rewrite one cell >= 10 times per episode


def _cycle_config(T, mm, cap, noise=0.0):
    ag = dict(CFG_AGENT, min_memory=mm, capacity=cap)
    return {"agents": [dict(ag), dict(ag, alpha=0.3, gamma=0.9)], "environment": dict(CFG_ENV, max_steps=T, noise_prob=noise)}


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("label,config,k,E", [
    ("T50_trains_every_2nd", _cycle_config(50, 100, 500), 2, 6),
    ("T10_trains_every_10th", _cycle_config(10, 100, 500), 10, 20),
    ("T30_mm70_every_3rd", _cycle_config(30, 70, 500), 3, 6),
    ("T100_deque_overflow_cap64", _cycle_config(100, 20, 64), 1, 4),           # only the last 64 transitions train
    ("T60_cycle_and_overflow", _cycle_config(60, 100, 100), 2, 6),             # 120 appended, 100 kept
    ("never_trains", _cycle_config(40, 100, 50), 1, 5),                        # capacity < min_memory
    ("T100_mm150_two_episodes_noise", _cycle_config(100, 150, 500, noise=0.1), 2, 4),   # 200 transitions: 4 segments
])
def test_wave_training_cycles_vs_oracle(label, config, k, E, dtype):
    """Replay buffers that span episodes (max_steps < min_memory: train_net trains every k-th episode on k*T
    transitions), deque overflow (only the last `capacity` transitions train) and buffers that never reach
    min_memory, on the LDS-resident wave kernel: bit for bit against the oracle, over two calls; a call
    that is not a whole number of cycles runs on the generic kernel and continues exactly."""
    import ctypes
    G, qd = 41, (1 if dtype == "float64" else 0)
    gb = _batch(config, G, dtype=dtype, kernel="auto", seed=14).init_tables()
    assert gb.L.thrl_training_cycle(ctypes.byref(gb.cfg)) == k
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    o1 = gb.run(E)
    assert o1["kernel"] == "wave", label
    q, c, s, eps, mem, oo = _oracle_run(config, G, qd, q0, s0, E, seed=14)
    assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c), label
    assert np.array_equal(gb.states_numpy(), s)
    assert [float(x) for x in gb.eps[:2]] == [float(x) for x in eps[:2]]
    np.testing.assert_allclose(o1["reward_log"], oo["reward_log"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(o1["action_log"], oo["action_log"], rtol=1e-12, atol=1e-13)
    if label == "never_trains":
        assert np.array_equal(q, q0) and not c.any()
    # k + 1 more episodes: not a whole number of cycles -> generic kernel, buffers kept; then back in step
    extra = k + 1 if k > 1 else 2
    o2 = gb.run(extra)
    assert o2["kernel"] == ("generic" if k > 1 else "wave")
    cfg, _ = O.cfg_from_config(config, n_games=G, q_dtype=qd)
    O.episodes(cfg, q, c, s, eps, mem, extra, seed=14, first_episode=E)
    assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c), label
    if k > 1:
        o3 = gb.run(k - 1)                                  # completes the cycle on the generic kernel
        assert o3["kernel"] == "generic"
        O.episodes(cfg, q, c, s, eps, mem, k - 1, seed=14, first_episode=E + extra)
        assert np.array_equal(gb.tables_numpy(), q)
        o4 = gb.run(k)                                      # buffers are empty again: wave kernel
        assert o4["kernel"] == "wave"
        O.episodes(cfg, q, c, s, eps, mem, k, seed=14, first_episode=E + extra + k - 1)
        assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c)


def test_zero_episodes_and_odd_sizes():
    """Empty and ragged inputs: 0 episodes is a no-op; G far below / not a multiple of the
    resident wave count."""
    gb = _batch(CFG, 3, kernel="wave", seed=1).init_tables()
    q0 = gb.tables_numpy().copy()
    out = gb.run(0)
    assert out["reward_log"].shape == (0, 2) and np.array_equal(gb.tables_numpy(), q0) and gb.episode == 0
    gb = _batch(CFG, 5121, kernel="wave", seed=1).init_tables()       # one more game than resident waves
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    gb.run(2)
    q, c, s, eps, mem, oo = _oracle_run(CFG, 5121, 0, q0, s0, 2, seed=1)
    assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c)


def test_per_game_sweeps_vs_oracle():
    """Per-game gamma / alpha / epsilon schedule / noise_prob (the reference's sweep use case,
    main.py:13-21 with configs2.json's gamma=0.35, alpha=0.5, epsilon=0.8) on the wave kernel:
    every game bit-identical to the oracle run with the same per-game arrays, over two calls."""
    G, E = 97, 6
    rs = np.random.RandomState(3)
    config = {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)], "environment": dict(CFG_ENV, noise_prob=0.2)}
    sweep = dict(gamma=rs.choice([0.35, 0.9, 0.95, 0.995], (2, G)), alpha=rs.choice([0.05, 0.1, 0.5], (2, G)),
                 eps=rs.choice([0.2, 0.5, 0.8], (2, G)), eps_end=rs.choice([0.001, 0.02], (2, G)),
                 eps_step=rs.choice([0.9995, 0.99], (2, G)), noise_prob=rs.choice([0.0, 0.05, 0.2], G))
    gb = _batch(config, G, dtype="float32", kernel="wave", seed=21)
    gb.set_sweep(sweep)
    gb.init_tables()
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    o1 = gb.run(E); o2 = gb.run(3)
    assert o1["kernel"] == "wave"
    cfg, eps = O.cfg_from_config(config, G, 0)
    q = q0.copy(); c = np.zeros(q.shape, np.int32); s = s0.copy(); mem = O.Memory(cfg)
    osw = {k: np.array(v, np.float64) for k, v in sweep.items()}          # copies: the oracle updates eps in place
    oo1 = O.episodes(cfg, q, c, s, eps, mem, E, seed=21, sweep=osw)
    oo2 = O.episodes(cfg, q, c, s, eps, mem, 3, seed=21, first_episode=E, sweep=osw)
    assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c)
    assert np.array_equal(gb.states_numpy(), s)
    assert np.array_equal(gb.sweep["eps"].cpu().numpy(), osw["eps"])          # per-game epsilon after 9 episodes
    np.testing.assert_allclose(o2["reward_log"], oo2["reward_log"], rtol=1e-12)
    # a sweep whose arrays all equal the config's scalars is the plain run
    plain = _batch(CFG, 33, kernel="wave", seed=2).init_tables(); plain.run(4)
    same = _batch(CFG, 33, kernel="wave", seed=2)
    same.set_sweep(dict(gamma=np.full(33, 0.95), alpha=np.full((2, 33), 0.1), eps_end=np.full(33, 0.001),
                        eps_step=np.full(33, 0.9995)))
    same.init_tables(); same.run(4)
    assert np.array_equal(plain.tables_numpy(), same.tables_numpy())
    # the generic kernel takes the same arrays and gives the same games
    gen = _batch(config, G, dtype="float32", kernel="generic", seed=21)
    gen.set_sweep(sweep); gen.set_tables(q0, s0)
    og = gen.run(E); gen.run(3)
    assert og["kernel"] == "generic"
    assert np.array_equal(gen.tables_numpy(), q) and np.array_equal(gen.counters_numpy(), c)
    assert np.array_equal(gen.sweep["eps"].cpu().numpy(), osw["eps"])


def test_generic_kernel_sweeps_vs_oracle():
    """Per-game sweeps on the generic kernel (configs the wave kernel cannot take: 3 players, buffers
    that span episodes; float64 and float32), against the oracle given the same arrays, bit for bit;
    a gamma sweep also moves the initial table offset 12.5/(1-gamma) per game."""
    rs = np.random.RandomState(3)
    for config, dtype in ((THREE, "float64"), (BUFFER, "float32")):
        G, E = 23, 6
        N = len(config["agents"])
        sweep = dict(gamma=rs.choice([0.35, 0.9, 0.95], (N, G)), alpha=rs.choice([0.05, 0.1, 0.5], (N, G)),
                     eps=rs.uniform(0.1, 0.9, (N, G)), eps_end=np.full((N, G), 0.01), eps_step=rs.choice([0.9, 0.999], (N, G)))
        gb = _batch(config, G, dtype=dtype, kernel="generic", seed=31)
        gb.set_sweep(sweep); gb.init_tables()
        q0, s0 = gb.tables_numpy(), gb.states_numpy()
        plain = _batch(config, G, dtype=dtype, kernel="generic", seed=31).init_tables()
        off = 12.5 / (1 - sweep["gamma"][0]) - 12.5 / (1 - config["agents"][0]["gamma"])     # agent 0's block moves by this
        rows = (config["agents"][0]["states"] + 1) * config["agents"][0]["actions"]
        np.testing.assert_allclose(q0[:, :rows] - plain.tables_numpy()[:, :rows], np.repeat(off[:, None], rows, 1),
                                   rtol=0, atol=1e-3 if dtype == "float32" else 1e-9)
        out = gb.run(E)
        assert out["kernel"] == "generic"
        cfg, eps0 = O.cfg_from_config(config, n_games=G, q_dtype=1 if dtype == "float64" else 0)
        q, st, cn = q0.copy(), s0.copy(), np.zeros(q0.shape, np.int32)
        osw = {k: np.array(v, np.float64) for k, v in sweep.items()}      # copies: the oracle updates eps in place
        O.episodes(cfg, q, cn, st, eps0, O.Memory(cfg), E, seed=31, sweep=osw)
        assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), cn)
        assert np.array_equal(gb.states_numpy(), st)
        assert np.array_equal(gb.sweep["eps"].cpu().numpy(), osw["eps"])


def test_learning_outcome_matches_reference_statistics():
    """Statistical parity in Philox mode: the reference's seeded 10,000-epoch
    2xQTable run ends at reward ~ 12.1-12.2 and scaled action ~ 0.29 per agent (above the Nash
    11.11: tacit collusion).  The mean over 512 independently seeded games must land there."""
    gb = _batch(CFG, 512, kernel="wave", seed=123).init_tables()
    first = gb.run(500)
    for _ in range(18):
        gb.run(500, logs=False)
    last = gb.run(500)
    assert gb.episode == 10000 and abs(gb.eps[0] - (0.001 + 0.499 * 0.9995 ** 10000)) < 1e-12
    r, a = last["reward_log"].mean(axis=0), last["action_log"].mean(axis=0)
    assert np.all(r > 11.9) and np.all(r < 12.3), r
    assert np.all(a > 0.280) and np.all(a < 0.295), a
    assert np.all(first["reward_log"].mean(axis=0) < 11.5)            # it actually learned
    mr, ma = gb.play_greedy(iters=1)
    total = mr.sum(axis=1).mean()
    assert 22.22 < total < 25.0, total                                  # between Nash and cartel


def test_play_greedy_vs_oracle():
    gb = _batch(CFG, 64, dtype="float32", seed=1).init_tables()
    gb.run(3)
    mr, ma = gb.play_greedy(iters=2)
    cfg, _ = O.cfg_from_config(CFG, 64, 0)
    omr, oma = O.play_greedy(cfg, gb.tables_numpy(), 2, seed=1)
    assert np.array_equal(mr, omr) and np.array_equal(ma, oma)
    st0 = np.random.RandomState(0).uniform(0, 10, (2, 64))
    mr, ma = gb.play_greedy(iters=2, state0=st0)
    omr, oma = O.play_greedy(cfg, gb.tables_numpy(), 2, state0=np.ascontiguousarray(st0))
    assert np.array_equal(mr, omr) and np.array_equal(ma, oma)


def test_wave_forced_on_unsupported_config_fails_loudly():
    from th_rl_amd._lib import ThrlError
    gb = _batch(BUFFER, 8, dtype="float32", kernel="wave").init_tables()
    with pytest.raises(ThrlError, match="wave kernel cannot run"):
        gb.run(1)
    three = {"agents": [dict(CFG_AGENT)] * 3, "environment": dict(CFG_ENV, nplayers=3)}
    with pytest.raises(ThrlError, match="2 agents"):
        _batch(three, 8, dtype="float64", kernel="wave").init_tables().run(1)


def test_full_size_properties_65536_games():
    """BASELINE config[1] size: properties that do not need the oracle.  Every agent
    makes exactly E*T visits; tables stay finite; wave == generic on device."""
    import torch
    G, E = 65536, 6
    a = _batch(CFG, G, kernel="wave", seed=3).init_tables()
    b = _batch(CFG, G, kernel="generic", seed=3).init_tables()
    ra = a.run(E); rb = b.run(E)
    assert ra["kernel"] == "wave"
    half = a.stride // 2
    assert bool((a.counter[:, :half].sum(dim=1) == E * 100).all())
    assert bool((a.counter[:, half:].sum(dim=1) == E * 100).all())
    assert bool(torch.isfinite(a.q).all())
    assert torch.equal(a.q, b.q) and torch.equal(a.counter, b.counter) and torch.equal(a.state, b.state)
    np.testing.assert_allclose(ra["reward_log"], rb["reward_log"], rtol=1e-12)
    # prices live on the action grid: 6 - 0.1*(k0+k1)
    st = a.states_numpy()
    assert st.min() >= 2.0 - 1e-9 and st.max() <= 6.0 + 1e-9
    # sanity band of the early learning curve (reward per agent between Nash-ish and cartel)
    assert 9.0 < ra["reward_log"].mean() < 12.6


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_full_size_properties_1M_games(dtype):
    """BASELINE config[2] size (2^20 games), float32 and the reference's float64: visit-count and finiteness
    invariants; the two dtypes play the same first episode (same draws, tables equal to float32 rounding)."""
    import torch
    G, E = 1 << 20, 3
    a = _batch(CFG, G, dtype=dtype, kernel="wave", seed=8).init_tables()
    r = a.run(E)
    assert r["kernel"] == "wave"
    half = a.stride // 2
    assert bool((a.counter[:, :half].sum(dim=1) == E * 100).all())
    assert bool((a.counter[:, half:].sum(dim=1) == E * 100).all())
    assert bool(torch.isfinite(a.q).all())
    assert 9.0 < r["reward_log"].mean() < 12.6
    st = a.state
    assert float(st.min()) >= 2.0 - 1e-9 and float(st.max()) <= 6.0 + 1e-9


def test_full_size_properties_training_cycles_65536_games():
    """Buffers that span episodes at BASELINE config[1] size: max_steps 50 with min_memory 100 trains every
    2nd episode on 100 transitions; every agent's visit count is exactly the number of trained transitions;
    wave == generic on device."""
    import torch
    config = _cycle_config(50, 100, 500)
    G, E = 65536, 8
    a = _batch(config, G, kernel="wave", seed=4).init_tables()
    b = _batch(config, G, kernel="generic", seed=4).init_tables()
    ra = a.run(E); rb = b.run(E)
    assert ra["kernel"] == "wave" and rb["kernel"] == "generic"
    half = a.stride // 2
    assert bool((a.counter[:, :half].sum(dim=1) == E * 50).all())
    assert torch.equal(a.q, b.q) and torch.equal(a.counter, b.counter) and torch.equal(a.state, b.state)
    np.testing.assert_allclose(ra["reward_log"], rb["reward_log"], rtol=1e-12)
