"""GPU parity tests, round 3: the wave kernel's code variants x training regimes, and oracle-checked slices of
full-size launches.  Everything through the C ABI, bit for bit against the CPU oracle (oracle/ is the checker
only)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O  # noqa: E402  (checker only)

CFG_AGENT = dict(name="QTable", gamma=0.95, actions=21, states=100, alpha=0.1, eps_end=0.001,
                 epsilon=0.5, eps_step=0.9995, action_range=[0.2, 0.4])
CFG_ENV = dict(name="NoisyPriceState", noise_prob=0, a=10, b=1, nplayers=2, max_steps=100)
CFG = {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)], "environment": dict(CFG_ENV)}


def _batch(config, G, dtype="float32", kernel="auto", seed=0, game_offset=0):
    from th_rl_amd.batched import GameBatch
    return GameBatch(config, n_games=G, dtype=dtype, kernel=kernel, seed=seed, game_offset=game_offset)


def _oracle(config, G, dtype, q0, s0, E, seed, game_offset=0, trace=False, sweep=None):
    cfg, eps = O.cfg_from_config(config, n_games=G, q_dtype=1 if dtype == "float64" else 0)
    q, s = q0.copy(), s0.copy()
    c = np.zeros(q.shape, np.int32)
    out = O.episodes(cfg, q, c, s, eps, O.Memory(cfg), E, seed=seed, game_offset=game_offset, trace=trace, sweep=sweep)
    return q, c, s, eps, out


def _quiet_groups_with_a_noisy_step(out, s0, T, max_state=10.0, states=100):
    """Counts, from the oracle's trace, the aligned groups of four steps whose four transitions are the SAME
    (state row, both actions, next row; the game stays in its row) although one of them was a noisy step, i.e.
    had another price and so another reward: the groups the four-identical-transitions shortcut must NOT take
    (advisor finding, round 2)."""
    P = out["trace_price"]                      # [E][T][G] price after each step
    A = out["trace_actions"]                    # [E][T][N][G]
    E, _, G = P.shape
    flat = P.reshape(E * T, G)
    prev = np.concatenate([s0[None, :], flat[:-1]], axis=0).reshape(E, T, G)
    r_prev = np.rint(prev / max_state * states)
    r_next = np.rint(P / max_state * states)
    n = 0
    for t0 in range(0, T - 3, 4):
        sl = slice(t0, t0 + 4)
        same = ((A[:, sl] == A[:, t0:t0 + 1]).all(axis=(1, 2)) & (r_prev[:, sl] == r_prev[:, t0:t0 + 1]).all(axis=1)
                & (r_next[:, sl] == r_prev[:, sl]).all(axis=1))
        differs = (P[:, sl] != P[:, t0:t0 + 1]).any(axis=1)
        n += int((same & differs).sum())
    return n


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("label,T,mm,noise", [("plain", 100, 100, 0.1), ("plain_noise20", 100, 100, 0.2),
                                              ("cycle_T50", 50, 100, 0.1)])
def test_noisy_step_inside_four_identical_transitions_vs_oracle(label, T, mm, noise, dtype):
    """A converged game repeats one transition; with env noise a noisy step that stays in the same row has the same
    (state, actions, next state) but ANOTHER reward (environments.py:29-33).  The replay must not fold such a group
    into the four-identical-transitions path.  Greedy-dominated play (epsilon = eps_end = 0.01) with noise, NOISE
    and NOISE + CYCLE variants, both dtypes: bit for bit against the oracle, and the case really occurs."""
    ag = dict(CFG_AGENT, epsilon=0.01, eps_end=0.01, min_memory=mm)
    config = {"agents": [dict(ag), dict(ag, alpha=0.3)], "environment": dict(CFG_ENV, noise_prob=noise, max_steps=T)}
    G, E = 2500, 8
    gb = _batch(config, G, dtype=dtype, kernel="wave", seed=31).init_tables()
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    out = gb.run(E)
    assert out["kernel"] == "wave"
    q, c, s, eps, oo = _oracle(config, G, dtype, q0, s0, E, seed=31, trace=True)
    n_cases = _quiet_groups_with_a_noisy_step(oo, s0, T)
    assert n_cases >= 5, n_cases                                  # the regime the test is for
    assert np.array_equal(gb.counters_numpy(), c)
    assert np.array_equal(gb.states_numpy(), s)
    bad = np.flatnonzero((gb.tables_numpy() != q).any(axis=1))
    assert bad.size == 0, "%d games differ from the oracle (first: %s); %d noisy same-row groups" % (bad.size, bad[:5], n_cases)
    np.testing.assert_allclose(out["reward_log"], oo["reward_log"], rtol=1e-12)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("eps", [0.0, 0.02, 0.05, 0.12])
@pytest.mark.parametrize("variant", ["wave_plain", "wave_greedy"])
def test_wave_variant_by_regime_matrix_vs_oracle(variant, eps, dtype):
    """{plain, GREEDY} code variants of the wave kernel x epsilon in {0, 0.02, 0.05, 0.12} x {float32, float64}:
    a real run executes the PLAIN variant's fixed-point path between epsilon 0.17 and 0.035 (episodes ~2,200-5,400)
    and the GREEDY one below; the variant is pinned per call (thrl_run.kernel = THRL_KERNEL_WAVE_PLAIN / _GREEDY)
    so every cell of the matrix is reached on purpose.  Bit for bit against the oracle."""
    ag = dict(CFG_AGENT, epsilon=eps, eps_end=eps)
    config = {"agents": [dict(ag), dict(ag, alpha=0.3)], "environment": dict(CFG_ENV)}
    G, E = 300, 7
    gb = _batch(config, G, dtype=dtype, kernel=variant, seed=77).init_tables()
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    out = gb.run(E)
    assert out["kernel"] == "wave"
    q, c, s, e2, oo = _oracle(config, G, dtype, q0, s0, E, seed=77)
    assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c)
    assert np.array_equal(gb.states_numpy(), s)
    np.testing.assert_allclose(out["reward_log"], oo["reward_log"], rtol=1e-12)
    # the regime: many games rewrite one cell over and over within an episode (fixed points / short cycles)
    assert (c.max(axis=1) >= 40).mean() > (0.3 if eps <= 0.02 else 0.05)


def test_wave_greedy_variant_is_refused_where_it_does_not_exist():
    from th_rl_amd._lib import ThrlError
    noisy = {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)], "environment": dict(CFG_ENV, noise_prob=0.05)}
    gb = _batch(noisy, 16, kernel="wave_greedy").init_tables()
    eps0 = list(gb.eps)
    with pytest.raises(ThrlError, match="no greedy-regime variant"):
        gb.run(2)
    assert list(gb.eps) == eps0 and gb.episode == 0                 # a refused call changes nothing
    assert _batch(noisy, 16, kernel="wave_plain").init_tables().run(2)["kernel"] == "wave"


def _greedy_cfg(eps, T=100, mm=100, cap=500, noise=0.0):
    ag = dict(CFG_AGENT, epsilon=eps, eps_end=eps, min_memory=mm, capacity=cap)
    return {"agents": [dict(ag), dict(ag, alpha=0.3, gamma=0.9)], "environment": dict(CFG_ENV, max_steps=T, noise_prob=noise)}


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("label,config,E", [
    ("noise_eps0", _greedy_cfg(0.0, noise=0.05), 6),
    ("noise_eps05", _greedy_cfg(0.05, noise=0.05), 6),
    ("cycle_T50_eps02", _greedy_cfg(0.02, T=50), 8),
    ("cycle_overflow_cap64_eps0", _greedy_cfg(0.0, mm=20, cap=64), 6),
    ("cycle_T50_noise_eps02", _greedy_cfg(0.02, T=50, noise=0.1), 8),
])
def test_noise_and_cycle_variants_in_the_greedy_regime_vs_oracle(label, config, E, dtype):
    """The NOISE, CYCLE and NOISE + CYCLE variants carry their own copies of the fixed-point replay path; at
    epsilon <= 0.05 (where converged games make it the common case) against the oracle, bit for bit."""
    G = 400
    gb = _batch(config, G, dtype=dtype, kernel="wave", seed=41).init_tables()
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    out = gb.run(E)
    assert out["kernel"] == "wave"
    q, c, s, eps, oo = _oracle(config, G, dtype, q0, s0, E, seed=41)
    assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c)
    assert np.array_equal(gb.states_numpy(), s)
    assert (c.max(axis=1) >= 20).mean() > 0.1


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_sweep_variant_in_the_greedy_regime_vs_oracle(dtype):
    """Per-game sweeps (the SWEEP variant, compiled with the noise code) with every game at epsilon <= 0.05."""
    G, E = 257, 6
    rs = np.random.RandomState(9)
    config = {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)], "environment": dict(CFG_ENV, noise_prob=0.1)}
    sweep = dict(gamma=rs.choice([0.9, 0.95], (2, G)), alpha=rs.choice([0.1, 0.3], (2, G)),
                 eps=rs.choice([0.0, 0.01, 0.05], (2, G)), eps_end=np.zeros((2, G)), eps_step=np.full((2, G), 0.999),
                 noise_prob=rs.choice([0.0, 0.1], G))
    gb = _batch(config, G, dtype=dtype, kernel="wave", seed=43)
    gb.set_sweep(sweep)
    gb.init_tables()
    q0, s0 = gb.tables_numpy(), gb.states_numpy()
    out = gb.run(E)
    assert out["kernel"] == "wave"
    osw = {k: np.array(v, np.float64) for k, v in sweep.items()}
    q, c, s, eps, oo = _oracle(config, G, dtype, q0, s0, E, seed=43, sweep=osw)
    assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c)
    assert np.array_equal(gb.states_numpy(), s)
    assert np.array_equal(gb.sweep["eps"].cpu().numpy(), osw["eps"])
    assert (c.max(axis=1) >= 20).mean() > 0.1


def test_greedy_variant_with_two_row_segments_vs_oracle():
    """GREEDY + a 101-row window (NRSEG = 2: the hand-scheduled chain and the composed greedy tables are off,
    cycle detection and the period-2 passes are on) -- advisor finding, round 2."""
    ag = dict(CFG_AGENT, action_range=[0.0, 0.5], epsilon=0.0, eps_end=0.0)
    config = {"agents": [dict(ag), dict(ag, alpha=0.3)], "environment": dict(CFG_ENV)}
    for dtype in ("float32", "float64"):
        G, E = 200, 6
        gb = _batch(config, G, dtype=dtype, kernel="wave_greedy", seed=19).init_tables()
        q0, s0 = gb.tables_numpy(), gb.states_numpy()
        assert gb.run(E)["kernel"] == "wave"
        q, c, s, eps, oo = _oracle(config, G, dtype, q0, s0, E, seed=19)
        assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c)


def _check_slices(gb, config, dtype, E, seed, slices, q_init, s_init):
    """Oracle-compares slices of a big batch: Philox is keyed by the GLOBAL game id, so a slice of the batch is a
    run of its own with game_offset = the slice's first game."""
    import torch
    for (lo, n), q0, s0 in zip(slices, q_init, s_init):
        q, c, s, eps, oo = _oracle(config, n, dtype, q0, s0, E, seed=seed, game_offset=gb.game_offset + lo)
        assert np.array_equal(gb.q[lo:lo + n].cpu().numpy(), q), "tables of games %d..%d" % (lo, lo + n)
        assert np.array_equal(gb.counter[lo:lo + n].cpu().numpy(), c), "counters of games %d..%d" % (lo, lo + n)
        assert np.array_equal(gb.state[lo:lo + n].cpu().numpy(), s), "states of games %d..%d" % (lo, lo + n)
    torch.cuda.synchronize()


def test_benched_shape_1M_games_one_20_episode_launch_oracle_slices():
    """The shape bench.py times (2^20 games, seed 0, ONE launch of 20 episodes, float32, counters on), then eight
    slices of 2,048 games -- first, last and six spread over the batch, so every region of the persistent grid's
    work queue is sampled -- against the oracle: tables, visit counters and env state bit for bit."""
    G, E, n = 1 << 20, 20, 2048
    gb = _batch(CFG, G, dtype="float32", kernel="wave", seed=0).init_tables()
    los = [0, 131072 + 77, 262144, 400000 + 1, 524288 - 1024, 700001, 900000, G - n]
    slices = [(lo, n) for lo in los]
    q_init = [gb.q[lo:lo + n].cpu().numpy() for lo, _ in slices]
    s_init = [gb.state[lo:lo + n].cpu().numpy() for lo, _ in slices]
    out = gb.run(E)
    assert out["kernel"] == "wave"
    _check_slices(gb, CFG, "float32", E, 0, slices, q_init, s_init)
    half = gb.stride // 2
    assert bool((gb.counter[:, :half].sum(dim=1) == E * 100).all())


@pytest.mark.parametrize("label,dtype,noise", [("noise05_f32", "float32", 0.05), ("f64", "float64", 0.0),
                                               ("noise05_f64", "float64", 0.05)])
def test_65536_games_one_launch_oracle_slices(label, dtype, noise):
    """BASELINE configs[1] size for the noise (class default noise_prob 0.05) and float64 variants: one 20-episode
    launch, four oracle-checked slices of 2,048 games."""
    config = {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)], "environment": dict(CFG_ENV, noise_prob=noise)}
    G, E, n = 65536, 20, 2048
    gb = _batch(config, G, dtype=dtype, kernel="wave", seed=3, game_offset=1 << 21).init_tables()
    slices = [(lo, n) for lo in (0, 20000 + 3, 40000, G - n)]
    q_init = [gb.q[lo:lo + n].cpu().numpy() for lo, _ in slices]
    s_init = [gb.state[lo:lo + n].cpu().numpy() for lo, _ in slices]
    out = gb.run(E)
    assert out["kernel"] == "wave"
    _check_slices(gb, config, dtype, E, 3, slices, q_init, s_init)
