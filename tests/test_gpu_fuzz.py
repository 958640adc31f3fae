"""Seeded fuzz parity: random all-QTable configurations (2-4 agents, per-agent action grids and
table sizes, ragged T / min_memory / capacity, env noise, both table dtypes) run on the CPU oracle
(Philox draws), on GameBatch (whichever kernel `auto` picks, and the generic kernel explicitly) and on
the one-wavefront-per-game kernel behind MixedGameBatch.  Everything an agent or the env owns must
come out bit-identical on all of them: tables, visit counters, env state, epsilon, replay fill."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O  # noqa: E402  (checker only)


def _random_config(rs):
    n = int(rs.choice([2, 2, 2, 3, 4]))
    T = int(rs.choice([5, 17, 40, 64, 100, 130]))
    agents = []
    same = rs.rand() < 0.5                      # half of the cases: identical agents (wave-kernel shape when n == 2)
    base = None
    for i in range(n):
        if same and base is not None:
            agents.append(dict(base)); continue
        lo = float(np.round(rs.uniform(0.05, 0.3), 2))
        a = dict(name="QTable", gamma=float(rs.choice([0.35, 0.9, 0.95])), actions=int(rs.choice([3, 8, 21, 33])),
                 states=int(rs.choice([10, 50, 100, 120])), alpha=float(rs.choice([0.05, 0.1, 0.3])),
                 eps_end=0.001, epsilon=float(rs.choice([0.1, 0.5, 0.9])), eps_step=0.999,
                 action_range=[lo, float(np.round(lo + rs.uniform(0.05, 0.25), 2))],
                 min_memory=int(rs.choice([1, T // 2 + 1, T, 2 * T + 3])),
                 capacity=int(rs.choice([max(2, T // 2), T + 5, 3 * T, 500])))
        base = a
        agents.append(a)
    env = dict(name="NoisyPriceState", noise_prob=float(rs.choice([0.0, 0.0, 0.1, 0.5])), a=10, b=1, nplayers=n, max_steps=T)
    return {"agents": agents, "environment": env}


@pytest.mark.parametrize("case", range(20))
def test_random_config_all_paths_agree(case):
    from th_rl_amd.batched import GameBatch
    from th_rl_amd.mixed import MixedGameBatch
    rs = np.random.RandomState(1000 + case)
    config = _random_config(rs)
    dtype = "float64" if case % 3 == 0 else "float32"
    G, E = int(rs.randint(3, 45)), int(rs.randint(3, 14))
    seed, off = int(rs.randint(0, 10 ** 6)), int(rs.randint(0, 1000))
    auto = GameBatch(config, n_games=G, dtype=dtype, kernel="auto", seed=seed, game_offset=off).init_tables()
    q0, s0 = auto.tables_numpy().copy(), auto.states_numpy().copy()
    auto.run(E)
    # CPU oracle from the same initial tables / states, same Philox seed
    cfg, eps0 = O.cfg_from_config(config, n_games=G, q_dtype=1 if dtype == "float64" else 0)
    q, st, cn = q0.copy(), s0.copy(), np.zeros(q0.shape, np.int32)
    mem = O.Memory(cfg)
    O.episodes(cfg, q, cn, st, eps0, mem, E, seed=seed, game_offset=off)
    label = "%s %s G=%d E=%d kernel=%s" % (config["environment"], dtype, G, E, auto.last_kernel)
    assert np.array_equal(auto.tables_numpy(), q), label
    assert np.array_equal(auto.counters_numpy(), cn) and np.array_equal(auto.states_numpy(), st), label
    N = len(config["agents"])
    assert [float(x) for x in auto.eps[:N]] == [float(x) for x in eps0[:N]], label
    if auto.last_kernel != "generic":
        gen = GameBatch(config, n_games=G, dtype=dtype, kernel="generic", seed=seed, game_offset=off)
        gen.set_tables(q0, s0)
        gen.run(E)
        assert np.array_equal(gen.tables_numpy(), q) and np.array_equal(gen.counters_numpy(), cn), label
    mixed = MixedGameBatch(config, n_games=G, dtype=dtype, seed=seed, game_offset=off)
    mixed.set_tables(q0, s0)
    out = mixed.run(E)              # fused; the operator loop when one game's tables exceed 64 KiB of LDS
    assert out["kernel"] in ("mixed-fused", "unfused")
    assert np.array_equal(mixed.tables_numpy(), q), label
    assert np.array_equal(mixed.counters_numpy(), cn) and np.array_equal(mixed.states_numpy(), st), label
    assert [float(x) for x in mixed.eps[:N]] == [float(x) for x in eps0[:N]], label


def _random_mixed_config(rs):
    n = int(rs.choice([2, 2, 3, 4]))
    T = int(rs.choice([6, 15, 20, 33]))
    agents, discrete = [], 0
    for i in range(n):
        kind = rs.choice(["QTable", "Reinforce", "ActorCritic", "CAC"])
        if kind in ("Reinforce", "ActorCritic") and discrete == 2:
            kind = "QTable"
        lo = float(np.round(rs.uniform(0.05, 0.3), 2)); hi = float(np.round(lo + rs.uniform(0.05, 0.25), 2))
        if kind == "QTable":
            agents.append(dict(name="QTable", gamma=0.95, actions=int(rs.choice([5, 21, 30])), states=int(rs.choice([20, 100])),
                               alpha=0.1, eps_end=0.001, epsilon=0.5, eps_step=0.999, action_range=[lo, hi],
                               min_memory=int(rs.choice([T, 2 * T])), capacity=int(rs.choice([2 * T + 1, 500]))))
        elif kind == "CAC":
            agents.append(dict(name="CAC", gamma=0.97, states=1, action_range=[lo, hi], min_memory=int(rs.choice([T, 3 * T])),
                               entropy=float(rs.choice([0.0, 0.01]))))
        else:
            discrete += 1
            agents.append(dict(name=kind, gamma=float(rs.choice([0.9, 0.995])), actions=int(rs.choice([4, 21, 27])), states=1,
                               action_range=[lo, hi], min_memory=int(rs.choice([T, 2 * T + 3])), entropy=float(rs.choice([0.0, 0.02]))))
    env = dict(name="NoisyPriceState", noise_prob=float(rs.choice([0.0, 0.0, 0.2])), a=10, b=1, nplayers=n, max_steps=T)
    return {"agents": agents, "environment": env}


@pytest.mark.parametrize("case", range(10))
def test_random_agent_mix_fused_equals_operator_loop(case):
    """Random mixes of QTable / Reinforce / ActorCritic / CAC agents: the fused episode kernel (policy
    memo, state-folding updates and all) against the per-call operator loop, bit for bit, across
    several network updates."""
    from th_rl_amd.mixed import MixedGameBatch
    rs = np.random.RandomState(2000 + case)
    config = _random_mixed_config(rs)
    dtype = "float64" if case % 2 else "float32"
    G, E = int(rs.randint(2, 9)), int(rs.randint(5, 10))
    seed = int(rs.randint(0, 10 ** 6))
    a = MixedGameBatch(config, n_games=G, dtype=dtype, seed=seed).init_tables()
    b = MixedGameBatch(config, n_games=G, dtype=dtype, seed=seed).init_tables()
    ra, rb = a.run(E, fused=True), b.run(E, fused=False)
    label = "%s %s G=%d E=%d" % ([x["name"] for x in config["agents"]], dtype, G, E)
    assert np.array_equal(ra["game_reward_log"], rb["game_reward_log"]), label
    assert np.array_equal(ra["game_action_log"], rb["game_action_log"]), label
    assert np.array_equal(a.tables_numpy(), b.tables_numpy()) and np.array_equal(a.counters_numpy(), b.counters_numpy()), label
    assert np.array_equal(a.states_numpy(), b.states_numpy()) and a.eps == b.eps and a.count == b.count, label
    for i in a.nn:
        assert a.nn[i].step == b.nn[i].step, label
        assert np.array_equal(a.nn[i].params.cpu().numpy(), b.nn[i].params.cpu().numpy()), label
        assert np.array_equal(a.nn[i].adam_v.cpu().numpy(), b.nn[i].adam_v.cpu().numpy()), label
