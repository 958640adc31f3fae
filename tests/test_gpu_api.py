"""GPU tests of the drop-in boundary: train_one's artefacts, the object-level
protocol (QTable / NoisyPriceState methods run device operators), the unfused
thrl_op_* entry points -- checked against the reference-generated goldens and the
oracle."""
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O  # noqa: E402  (checker only)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG_AGENT = dict(name="QTable", gamma=0.95, actions=21, states=100, alpha=0.1, eps_end=0.001,
                 epsilon=0.5, eps_step=0.9995, action_range=[0.2, 0.4])
CFG_ENV = dict(name="NoisyPriceState", noise_prob=0, a=10, b=1, nplayers=2, max_steps=100)


def _config(epochs, **training):
    return {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)], "environment": dict(CFG_ENV),
            "training": dict({"print_freq": 5, "epochs": epochs}, **training)}


def test_train_one_single_game_artifacts_and_parity(tmp_path, capsys):
    """train_one on the 2xQTable config (BASELINE config[0] shape, float64, 1 game):
    the four artefacts have the reference's formats and the numbers equal the oracle's
    float64 run from the same initial tables / state / Philox seed, bit for bit."""
    import pandas
    from th_rl_amd import trainer
    cfgp = tmp_path / "cfg.json"
    config = _config(12, seed=99)
    cfgp.write_text(json.dumps(config))
    exp = str(tmp_path / "run0")
    np.random.seed(5)
    trainer.train_one(exp, str(cfgp), print_eps=True)
    printed = capsys.readouterr().out.strip().splitlines()
    assert len(printed) == 2 and printed[0].startswith("eps:[") and "| episode:  4 |" in printed[0]
    assert "agents:QTable,QTable" in printed[1] and "| episode:  9 |" in printed[1]
    assert sorted(os.listdir(exp)) == ["0.npy", "0_counter.npy", "1.npy", "1_counter.npy", "config.json", "log.csv"]
    assert json.load(open(os.path.join(exp, "config.json"))) == config
    assert open(os.path.join(exp, "config.json")).read() == json.dumps(config, indent=3)
    head = open(os.path.join(exp, "log.csv")).read().splitlines()[:2]
    assert head == ["rewards,rewards,actions,actions", "0,1,0,1"]            # golden log_csv_head
    assert head == str(np.load(os.path.join(GOLDEN, "g4_cfg_seed0_e12.npz"))["log_csv_head"]).splitlines()
    # same RNG stream -> same initial tables and reset() state as train_one consumed
    np.random.seed(5)
    _, agents, env = trainer.create_game(str(cfgp))
    state0 = env.reset()
    cfg, eps = O.cfg_from_config(config, 1, 1)
    q = np.concatenate([a.table.ravel() for a in agents])[None, :].copy()
    c = np.zeros(q.shape, np.int32); s = np.array([float(state0[0])])
    out = O.episodes(cfg, q, c, s, eps, O.Memory(cfg), 12, seed=99)
    for i in range(2):
        t = np.load(os.path.join(exp, "%d.npy" % i)); k = np.load(os.path.join(exp, "%d_counter.npy" % i))
        assert t.dtype == np.float64 and t.shape == (101, 21) and k.dtype == np.float64 and k.shape == (101, 21)
        assert np.array_equal(t.ravel(), q[0, i * 2121:(i + 1) * 2121])
        assert np.array_equal(k.ravel(), c[0, i * 2121:(i + 1) * 2121].astype(np.float64))
        assert k.sum() == 12 * 100
    log = pandas.read_csv(os.path.join(exp, "log.csv"), header=[0, 1], float_precision="round_trip")
    assert np.array_equal(log["rewards"].to_numpy(), out["game_reward_log"][:, :, 0])
    assert np.array_equal(log["actions"].to_numpy(), out["game_action_log"][:, :, 0])


def test_train_one_batched_games(tmp_path):
    """n_games > 1: fused wave kernel, log = mean over games, batch.pt holds every game,
    and utils.load_experiment reads the run back."""
    import torch
    from th_rl_amd import trainer, utils
    cfgp = tmp_path / "cfg.json"
    config = _config(10, seed=3, n_games=300, print_freq=500)
    cfgp.write_text(json.dumps(config))
    exp = str(tmp_path / "run")
    trainer.train_one(exp, str(cfgp))
    b = torch.load(os.path.join(exp, "batch.pt"), weights_only=True)
    assert b["q"].shape == (300, 4242) and b["q"].dtype == torch.float32 and b["episode"] == 10
    cfg, eps = O.cfg_from_config(config, 300, 0)
    q, c, s = O.init(cfg, seed=3)
    # device init differs from libm in ulps: start the oracle from the device's own tables
    from th_rl_amd.batched import GameBatch
    gb = GameBatch(config, n_games=300, seed=3).init_tables()
    q, s = gb.tables_numpy(), gb.states_numpy()
    out = O.episodes(cfg, q, c, s, eps, O.Memory(cfg), 10, seed=3)
    assert np.array_equal(b["q"].numpy(), q) and np.array_equal(b["counter"].numpy(), c)
    assert np.array_equal(np.load(os.path.join(exp, "0.npy")), q[0, :2121].astype(np.float64).reshape(101, 21))
    config2, agents, env, actions, rewards = utils.load_experiment(exp)
    # reference quirk kept: read_csv takes ONE header row, so log.csv's second header line
    # ("0,1,0,1") is returned as the first data row (utils.py:17-21) -> epochs + 1 rows
    assert list(rewards.columns) == ["QTable0", "QTable1"] and len(rewards) == 11
    assert list(rewards.iloc[0]) == [0.0, 1.0]
    import pandas
    log = pandas.read_csv(os.path.join(exp, "log.csv"), header=[0, 1], float_precision="round_trip")
    np.testing.assert_allclose(log["rewards"].to_numpy(), out["reward_log"], rtol=1e-12)
    assert np.array_equal(agents[1].table.ravel(), q[0, 2121:].astype(np.float64))


def test_checkpoint_resume_is_bit_identical(tmp_path):
    """6 + 9 episodes with a save/load in between == 15 episodes in one go (both kernels,
    including a config whose replay memory spans episodes)."""
    from th_rl_amd.batched import GameBatch
    cfg_a = {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)], "environment": dict(CFG_ENV)}
    cfg_b = {"agents": [dict(CFG_AGENT, min_memory=70, capacity=90), dict(CFG_AGENT)],
             "environment": dict(CFG_ENV, max_steps=30, noise_prob=0.05)}
    for config, dtype in ((cfg_a, "float32"), (cfg_b, "float64")):
        one = GameBatch(config, n_games=50, dtype=dtype, seed=4).init_tables()
        r_one = one.run(15)
        a = GameBatch(config, n_games=50, dtype=dtype, seed=4).init_tables()
        r_a = a.run(6)
        a.save(str(tmp_path / "ck.pt"))
        b = GameBatch(config, n_games=50, dtype=dtype, seed=999).load(str(tmp_path / "ck.pt"))
        assert b.episode == 6 and b.eps == a.eps
        r_b = b.run(9)
        assert np.array_equal(b.tables_numpy(), one.tables_numpy())
        assert np.array_equal(b.counters_numpy(), one.counters_numpy())
        assert np.array_equal(b.states_numpy(), one.states_numpy())
        assert b.eps == one.eps and b.episode == 15 and b.mem_count == one.mem_count
        np.testing.assert_allclose(np.concatenate([r_a["reward_log"], r_b["reward_log"]]), r_one["reward_log"], rtol=1e-12)


def test_train_one_resume(tmp_path):
    import pandas
    from th_rl_amd import trainer
    full = _config(8, seed=3, n_games=64, print_freq=500)
    (tmp_path / "full.json").write_text(json.dumps(full))
    trainer.train_one(str(tmp_path / "full"), str(tmp_path / "full.json"))
    first = _config(3, seed=3, n_games=64, print_freq=500)
    (tmp_path / "first.json").write_text(json.dumps(first))
    trainer.train_one(str(tmp_path / "first"), str(tmp_path / "first.json"))
    second = _config(5, seed=3, n_games=64, print_freq=500, resume=str(tmp_path / "first" / "batch.pt"))
    (tmp_path / "second.json").write_text(json.dumps(second))
    trainer.train_one(str(tmp_path / "second"), str(tmp_path / "second.json"))
    for f in ("0.npy", "1.npy", "0_counter.npy", "1_counter.npy"):
        assert np.array_equal(np.load(tmp_path / "full" / f), np.load(tmp_path / "second" / f)), f
    lf = pandas.read_csv(tmp_path / "full" / "log.csv", header=[0, 1]).to_numpy()
    ls = pandas.read_csv(tmp_path / "second" / "log.csv", header=[0, 1]).to_numpy()
    np.testing.assert_allclose(lf[3:], ls, rtol=1e-12)


def test_sharded_launch_equals_single_process(tmp_path):
    """th_rl_amd.launch with 2 ranks (both mapped onto this box's one GPU): every game and the merged
    log equal a single-process run of all games -- the seed-sharded multi-GPU path."""
    import pandas
    import torch
    from th_rl_amd import trainer
    from th_rl_amd.launch import launch
    cfg = _config(6, seed=17, n_games=101, print_freq=500)          # uneven split: 51 + 50
    (tmp_path / "c.json").write_text(json.dumps(cfg))
    trainer.train_one(str(tmp_path / "one"), str(tmp_path / "c.json"))
    launch(str(tmp_path / "c.json"), str(tmp_path / "two"), gpus=2)
    one = torch.load(tmp_path / "one" / "batch.pt", weights_only=True)
    s0 = torch.load(tmp_path / "two" / "shard0" / "batch.pt", weights_only=True)
    s1 = torch.load(tmp_path / "two" / "shard1" / "batch.pt", weights_only=True)
    assert s0["q"].shape[0] == 51 and s1["q"].shape[0] == 50 and s1["game_offset"] == 51
    assert torch.equal(torch.cat([s0["q"], s1["q"]]), one["q"])
    assert torch.equal(torch.cat([s0["counter"], s1["counter"]]), one["counter"])
    for f in ("0.npy", "1.npy", "0_counter.npy", "config.json"):
        assert open(tmp_path / "two" / f, "rb").read() == open(tmp_path / "one" / f, "rb").read(), f
    l1 = pandas.read_csv(tmp_path / "one" / "log.csv", header=[0, 1]).to_numpy()
    l2 = pandas.read_csv(tmp_path / "two" / "log.csv", header=[0, 1]).to_numpy()
    np.testing.assert_allclose(l2, l1, rtol=1e-12)


def test_sharded_launch_one_game_per_rank_and_more_ranks_than_games(tmp_path):
    """2 games over 3 requested ranks (clamped to 2, one game each) equals 2 games in one process: a
    one-game shard keeps float32 tables and the Philox initialisation keyed by the global game id."""
    import torch
    from th_rl_amd import trainer
    from th_rl_amd.launch import launch
    cfg = _config(5, seed=23, n_games=2, print_freq=500)
    (tmp_path / "c.json").write_text(json.dumps(cfg))
    trainer.train_one(str(tmp_path / "one"), str(tmp_path / "c.json"))
    launch(str(tmp_path / "c.json"), str(tmp_path / "two"), gpus=3)
    assert not (tmp_path / "two" / "shard2").exists()
    one = torch.load(tmp_path / "one" / "batch.pt", weights_only=True)
    s0 = torch.load(tmp_path / "two" / "shard0" / "batch.pt", weights_only=True)
    s1 = torch.load(tmp_path / "two" / "shard1" / "batch.pt", weights_only=True)
    assert s0["q"].dtype == torch.float32 and s1["game_offset"] == 1
    assert torch.equal(torch.cat([s0["q"], s1["q"]]), one["q"])
    assert torch.equal(torch.cat([s0["counter"], s1["counter"]]), one["counter"])
    for f in ("0.npy", "1.npy", "0_counter.npy"):
        assert open(tmp_path / "two" / f, "rb").read() == open(tmp_path / "one" / f, "rb").read(), f


def test_two_batches_on_two_streams_equal_sequential_runs():
    """Re-entrancy on the device side: two independent batches launched back to back on two HIP
    streams (each with its own workspace and work counter) give the results of running them alone."""
    import torch
    from th_rl_amd.batched import GameBatch
    config = {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)], "environment": dict(CFG_ENV)}
    def fresh(seed):
        return GameBatch(config, n_games=3000, seed=seed, kernel="wave").init_tables()
    alone = []
    for seed in (1, 2):
        gb = fresh(seed)
        gb.run(6)
        alone.append((gb.tables_numpy(), gb.counters_numpy(), gb.states_numpy()))
    a, b = fresh(1), fresh(2)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    keep = []
    for _ in range(3):                                  # interleaved launches, 2 episodes each
        with torch.cuda.stream(s1):
            keep.append(a.run(2, sync=False))
        with torch.cuda.stream(s2):
            keep.append(b.run(2, sync=False))
    torch.cuda.synchronize()
    for gb, (q, c, st) in zip((a, b), alone):
        assert np.array_equal(gb.tables_numpy(), q) and np.array_equal(gb.counters_numpy(), c)
        assert np.array_equal(gb.states_numpy(), st)


def test_train_one_sweep(tmp_path):
    """A gamma sweep (the reference's configs2.json value 0.35 vs example_config's 0.95) as one
    batched train_one: game g equals a plain run whose config has that gamma."""
    import torch
    from th_rl_amd import trainer
    from th_rl_amd.batched import GameBatch
    gam = [0.35] * 40 + [0.95] * 40
    cfg = _config(4, seed=9, n_games=80, print_freq=500, sweep={"gamma": gam})
    (tmp_path / "s.json").write_text(json.dumps(cfg))
    trainer.train_one(str(tmp_path / "s"), str(tmp_path / "s.json"))
    b = torch.load(tmp_path / "s" / "batch.pt", weights_only=True)
    assert b["sweep"]["gamma"].shape == (2, 80)
    for gamma, sl in ((0.35, slice(0, 40)), (0.95, slice(40, 80))):
        c2 = _config(4)
        for a in c2["agents"]:
            a["gamma"] = gamma
        # game g of the sweep IS the game a plain run of that config trains: same Philox key, and its
        # initial table starts at that config's 12.5/(1-gamma) (agents.py:29)
        ref = GameBatch(c2, n_games=80, seed=9).init_tables()
        ref.run(4)
        assert np.array_equal(ref.tables_numpy()[sl], b["q"].numpy()[sl])
        assert np.array_equal(ref.counters_numpy()[sl], b["counter"].numpy()[sl])


def test_protocol_env_step_matches_reference_grid():
    """NoisyPriceState.step / QTable.scale / encode through the device operators ==
    the reference on the whole 21x21 action grid (golden G1), incl. noise branch."""
    from th_rl_amd.agents import QTable
    from th_rl_amd.environments import NoisyPriceState
    d = np.load(os.path.join(GOLDEN, "g1_payoff_grid.npz"))
    np.random.seed(0)
    ag = QTable(**CFG_AGENT)
    env = NoisyPriceState(**CFG_ENV)
    assert [ag.scale(k) for k in range(21)] == list(d["scaled"])
    for k0, k1 in [(0, 0), (20, 20), (14, 12), (3, 17), (10, 10), (20, 0)]:
        env.episode = 0
        s, r, done = env.step([ag.scale(k0), ag.scale(k1)])
        assert s.shape == (1,) and s[0] == d["price"][k0, k1] and np.array_equal(r, d["rewards"][k0, k1])
        assert not done and env.episode == 1
        assert ag.encode(s)[0] == d["enc64"][k0, k1]
        assert ag.encode(s.astype("float32"))[0] == d["enc32"][k0, k1]
    g2 = np.load(os.path.join(GOLDEN, "g2_encode.npz"))
    assert np.array_equal(ag.encode(g2["x64"][:40]), g2["e100_64"][:40])
    assert np.array_equal(ag.encode(g2["x32"][:40]), g2["e100_32"][:40])
    env.max_steps = 1; env.episode = 0
    assert env.step([0.3, 0.3])[2]
    # noise branch: same numpy stream as the oracle's injected draws
    envn = NoisyPriceState(**dict(CFG_ENV, noise_prob=1.0))
    np.random.seed(4)
    s, r, _ = envn.step([0.3, 0.25])
    np.random.seed(4)
    nu = np.random.uniform(0, 1); na = np.random.uniform(10 * 0.7, 10)
    cfg, _ = O.cfg_from_config({"agents": [dict(CFG_AGENT)] * 2, "environment": dict(CFG_ENV)}, 1, 1)
    p, rr = O.env_step(cfg, [0.3, 0.25], True, na)
    assert s[0] == p and np.array_equal(r, rr) and nu < 1.0


def test_encode_operator_fast_path_and_boundary_fallback_vs_numpy():
    """thrl_op_encode (and the wave kernel's noisy steps) use a division-free encode that falls back to the
    exact quotient near a rounding boundary.  Checked against numpy's own arithmetic -- QTable.encode is
    numpy.round(state / max_state * states), agents.py:47-49 -- on random prices, on every half-integer
    boundary and on its float neighbours, in float64 and in the float32 form the trainer feeds."""
    import ctypes
    import torch
    from th_rl_amd import _lib
    rs = np.random.RandomState(5)
    ks = np.arange(0, 100) + 0.5
    near = []
    for b in ks / 100.0 * 10.0:                       # prices whose row value sits at k + 0.5
        x = np.float64(b)
        for _ in range(4):
            near += [x, np.nextafter(x, 0.0), np.nextafter(x, 20.0)]
            x = np.nextafter(x, 20.0)
        x32 = np.float32(b)
        near += [np.float64(x32), np.float64(np.nextafter(x32, np.float32(0))), np.float64(np.nextafter(x32, np.float32(20)))]
    price = np.concatenate([rs.uniform(0, 10, 200000), np.array(near, np.float64), [0.0, 10.0, 9.999999999]])
    N = len(price)
    cfg, _ = _lib.cfg_from_config({"agents": [dict(CFG_AGENT)], "environment": dict(CFG_ENV, nplayers=1)}, N, 1)
    L = _lib.load()
    d_p = torch.from_numpy(price).cuda()
    out = torch.zeros(N, dtype=torch.int32, device="cuda")
    for as_f32 in (0, 1):
        _lib.check(L.thrl_op_encode(ctypes.byref(cfg), 0, ctypes.c_void_p(d_p.data_ptr()), as_f32,
                                    ctypes.c_void_p(out.data_ptr()), None), "thrl_op_encode")
        torch.cuda.synchronize()
        st = price.astype(np.float32) if as_f32 else price
        want = np.clip(np.round(st / 10 * 100).astype(np.int64), 0, 100)
        assert np.array_equal(out.cpu().numpy(), want), as_f32


def test_protocol_train_net_matches_reference_known_answers():
    """QTable.memory.append + train_net through thrl_op_td_update == golden G3 cases."""
    from th_rl_amd.agents import QTable
    d = np.load(os.path.join(GOLDEN, "g3_td_known_answers.npz"))
    for name in [str(n) for n in d["case_names"]]:
        g = lambda k: d["%s__%s" % (name, k)]
        mm, cap, alpha, gamma, eps, eps_end, eps_step, states, actions, repeat = g("params")
        np.random.seed(0)
        ag = QTable(states=int(states), actions=int(actions), gamma=gamma, alpha=alpha, eps_end=eps_end,
                    epsilon=eps, eps_step=eps_step, min_memory=int(mm), capacity=int(cap))
        ag.table = g("table0").copy(); ag.counter = 0 * ag.table
        for rep in range(int(repeat)):
            for p, a, r, pn in zip(g("price"), g("action"), g("reward"), g("next_price")):
                ag.memory.append(np.array([p]), np.int64(a), float(r), True, np.array([pn]))
            ag.train_net()
            assert np.array_equal(ag.table, g("table")[rep]), (name, rep)
            assert ag.epsilon == g("eps")[rep]
        assert np.array_equal(ag.counter, g("counter")) and len(ag.memory) == int(g("mem_len"))


def test_protocol_loop_reproduces_reference_trajectory():
    """Driving the duck-typed step loop (trainer.py:46-70) by hand with our objects and
    the same seeds as the golden run reproduces the reference's prices and final tables."""
    import torch
    from th_rl_amd.agents import QTable
    from th_rl_amd.environments import NoisyPriceState
    d = np.load(os.path.join(GOLDEN, "g4_cfg_seed1_e12.npz"))
    np.random.seed(1); random.seed(1); torch.manual_seed(1)
    agents = [QTable(**CFG_AGENT), QTable(**CFG_AGENT)]
    env = NoisyPriceState(**CFG_ENV)
    assert np.array_equal(np.concatenate([a.table.ravel() for a in agents]), d["init_tables"])
    state = env.reset()
    assert state[0] == d["state0"]
    E = 2
    for e in range(E):
        done = False
        env.episode = 0
        t = 0
        while not done:
            acts = [a.sample_action(torch.from_numpy(state.astype("float32"))) for a in agents]
            scaled = [a.scale(x) for a, x in zip(agents, acts)]
            nxt, reward, done = env.step(scaled)
            assert nxt[0] == d["states"][e, t] and np.array_equal(reward, d["rewards"][e, t])
            for a, r, x in zip(agents, reward, acts):
                a.memory.append(state, x, r, not done, nxt)
            state = nxt
            t += 1
        [a.train_net() for a in agents]
        assert [a.epsilon for a in agents] == list(d["eps"][e])
    # tables after 2 episodes == oracle (pinned to the reference) after 2 episodes
    config = json.loads(str(d["config_json"]))
    cfg, eps = O.cfg_from_config(config, 1, 1)
    q = d["init_tables"][None, :].copy(); c = np.zeros(q.shape, np.int32); s = np.array([float(d["state0"])])
    O.episodes(cfg, q, c, s, eps, O.Memory(cfg), E, inj_u=np.ascontiguousarray(d["u"][:E, :, :, None]),
               inj_choice=np.ascontiguousarray(d["choice"][:E, :, :, None]))
    assert np.array_equal(np.concatenate([a.table.ravel() for a in agents]), q[0])
    assert np.array_equal(np.concatenate([a.counter.ravel() for a in agents]), c[0].astype(np.float64))


def test_play_game_protocol_equals_batched_kernel():
    """utils.play_game (object protocol) == thrl_play_greedy (one kernel) == oracle."""
    from th_rl_amd import utils
    from th_rl_amd.agents import QTable
    from th_rl_amd.environments import NoisyPriceState
    from th_rl_amd.batched import GameBatch
    config = {"agents": [dict(CFG_AGENT), dict(CFG_AGENT)], "environment": dict(CFG_ENV, max_steps=15)}
    gb = GameBatch(config, n_games=3, dtype="float64", seed=2).init_tables()
    gb.run(2)
    np.random.seed(8)
    agents = [QTable(**CFG_AGENT), QTable(**CFG_AGENT)]
    for i, a in enumerate(agents):
        a.table = gb.table(1, i)
    env = NoisyPriceState(**config["environment"])
    np.random.seed(21)
    actions, rewards = utils.play_game(agents, env, iters=2)
    assert actions.shape == (30, 2) and rewards.shape == (30, 2)
    np.random.seed(21)
    st0 = np.array([[np.random.uniform(0, 10)] * 3 for _ in range(1)])   # first reset() draw
    mr, ma = utils.play_game_batched(gb, iters=1, state0=st0)
    np.testing.assert_allclose(mr[0, :, 1], rewards[:15].mean(axis=0), rtol=1e-13)
    np.testing.assert_allclose(ma[0, :, 1], actions[:15].mean(axis=0), rtol=1e-13)


def test_stored_reference_run_plays_the_same_greedy_game():
    """The agents the reference ships trained (QTable + Reinforce state_dict) loaded through
    utils.load_experiment and played through utils.play_game on the device == the game the
    reference's own play_game plays with them (fixture G10, recorded by running the reference):
    every scaled action and every reward of three 100-step games, bit for bit."""
    from th_rl_amd.utils import load_experiment, play_game
    loc = os.path.join(GOLDEN, "ref_run_example_config")
    d = np.load(os.path.join(GOLDEN, "g10_stored_run.npz"))
    config, agents, env, _, _ = load_experiment(loc)
    for seed in (0, 1, 2):
        np.random.seed(seed)
        env.episode = 0
        acts, rews = play_game(agents, env, iters=1)
        assert acts.shape == (100, 2) and rews.shape == (100, 2)
        assert np.array_equal(acts, d["play%d_actions" % seed]), seed
        assert np.array_equal(rews, d["play%d_rewards" % seed]), seed


def test_mixed_batch_greedy_play_equals_protocol_play_game():
    """MixedGameBatch.play_greedy (all games at once) == utils.play_game through the object protocol
    on the reference's shipped agents: same states in, same mean reward / action out."""
    from th_rl_amd.mixed import MixedGameBatch
    from th_rl_amd.utils import load_experiment, play_game
    loc = os.path.join(GOLDEN, "ref_run_example_config")
    d = np.load(os.path.join(GOLDEN, "g10_stored_run.npz"))
    config, agents, env, _, _ = load_experiment(loc)
    G = 3
    mb = MixedGameBatch({"agents": config["agents"], "environment": config["environment"]}, n_games=G, dtype="float64")
    flat = np.zeros((G, mb.stride))
    flat[:, mb.offsets[0]:mb.offsets[0] + agents[0].table.size] = agents[0].table.ravel()
    mb.nn[1].set_params(agents[1].flat_params())
    mb.set_tables(flat, [0.0] * G)
    s0 = np.array([[float(d["play%d_state0" % k]) for k in range(G)]])
    mr, ma = mb.play_greedy(iters=1, state0=s0)
    for k in range(G):
        np.testing.assert_allclose(mr[0, :, k], d["play%d_rewards" % k].mean(axis=0), rtol=1e-12)
        np.testing.assert_allclose(ma[0, :, k], d["play%d_actions" % k].mean(axis=0), rtol=1e-12)


def test_main_cli_runs_a_config_directory(tmp_path):
    """python -m th_rl_amd.main --dir <configs> --runs 2 (th_rl/main.py:6-23): one run directory per
    config and run under <dir>/../runs/<config>/<i>, each with the reference's artefacts; a second
    invocation skips finished configs."""
    import subprocess
    import sys
    cdir = tmp_path / "configs"
    cdir.mkdir()
    (cdir / "qq.json").write_text(json.dumps(_config(3, seed=1)))
    mixed = _config(3, seed=2)
    mixed["agents"][1] = {"name": "Reinforce", "gamma": 0.995, "actions": 21, "states": 1, "action_range": [0.2, 0.4],
                          "min_memory": 200}
    (cdir / "qr.json").write_text(json.dumps(mixed))
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-m", "th_rl_amd.main", "--dir", str(cdir), "--runs", "2"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    for conf in ("qq", "qr"):
        for i in ("0", "1"):
            d = tmp_path / "runs" / conf / i
            assert (d / "0.npy").exists() and (d / "config.json").exists() and (d / "log.csv").exists()
    assert (tmp_path / "runs" / "qr" / "0" / "1").exists() and (tmp_path / "runs" / "qq" / "1" / "1_counter.npy").exists()
    again = subprocess.run([sys.executable, "-m", "th_rl_amd.main", "--dir", str(cdir), "--runs", "2"], env=env,
                           capture_output=True, text=True, timeout=300)
    assert again.returncode == 0 and again.stdout.count("Skipping") == 2
