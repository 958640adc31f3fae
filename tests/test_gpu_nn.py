"""GPU tests of the neural policy path (Reinforce, reference agents.py:119-220) against the numpy
oracle and the reference-generated fixture G7.  float32 tolerances (torch's summation order is
unspecified): probabilities rtol 2e-5, clipped gradients rtol 2e-4 + atol 2e-6, Adam moments rtol
1e-3, parameters atol 5e-6 except Adam's sign-sensitive near-zero-gradient elements (<= 0.2 %)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import nn_oracle as NN  # noqa: E402  (checker only)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_reinforce.npz")
CASES = {"cfg": dict(gamma=0.995, entropy=0.0), "ent": dict(gamma=0.35, entropy=0.01)}


def _rb(G, **kw):
    from th_rl_amd.nn import ReinforceBatch
    return ReinforceBatch(G, actions=21, **kw)


@pytest.mark.parametrize("tag", sorted(CASES))
def test_act_probabilities_match_reference(tag):
    d = np.load(GOLDEN)
    probe = d[tag + "_probe_price"]
    rb = _rb(len(probe)).set_params(d[tag + "_w0"])
    act, probs = rb.act(probe, want_probs=True)
    np.testing.assert_allclose(probs.cpu().numpy(), d[tag + "_probe_prob0"], rtol=2e-5, atol=1e-8)
    assert np.array_equal(act.cpu().numpy(), d[tag + "_probe_prob0"].argmax(axis=1))
    rb.set_params(d[tag + "_c1_w"])
    act2, probs2 = rb.act(probe, want_probs=True)
    np.testing.assert_allclose(probs2.cpu().numpy(), d[tag + "_probe_prob2"], rtol=2e-5, atol=1e-8)
    assert np.array_equal(act2.cpu().numpy(), d[tag + "_probe_greedy2"])        # get_action


def test_sampling_matches_oracle_inverse_cdf():
    d = np.load(GOLDEN)
    G = 4096
    rs = np.random.RandomState(5)
    price = rs.randint(20, 61, G) / 10.0
    u = rs.uniform(0, 1, G)
    rb = _rb(G).set_params(d["cfg_w0"])
    a = rb.act(price, u=u).cpu().numpy()
    ref = NN.sample_action(d["cfg_w0"], 21, price, u)
    assert (a == ref).mean() > 0.999            # a float32-rounding tie can move a boundary draw
    assert a.min() >= 0 and a.max() <= 20


@pytest.mark.parametrize("tag", sorted(CASES))
def test_train_net_matches_reference_two_calls(tag):
    d = np.load(GOLDEN)
    kw = CASES[tag]
    rb = _rb(3, gamma=kw["gamma"], entropy=kw["entropy"]).set_params(d[tag + "_w0"])
    for call in range(2):
        if call == 1:        # start the second call from the reference's own state
            import torch
            rb.set_params(d[tag + "_c0_w"])
            rb.adam_m.copy_(torch.from_numpy(np.broadcast_to(d[tag + "_c0_m"], (3, rb.P)).copy()))
            rb.adam_v.copy_(torch.from_numpy(np.broadcast_to(d[tag + "_c0_v"], (3, rb.P)).copy()))
            rb.step = 1
        pr = d["%s_c%d_price" % (tag, call)][:1000]
        tile = lambda x: np.repeat(np.asarray(x)[:, None], 3, axis=1)
        g = rb.train(tile(pr), tile(d["%s_c%d_action" % (tag, call)]), tile(d["%s_c%d_reward" % (tag, call)]),
                     want_grad=True).cpu().numpy()
        for k in range(3):
            np.testing.assert_allclose(g[k], d["%s_c%d_grad" % (tag, call)], rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(rb.adam_m.cpu().numpy()[1], d["%s_c%d_m" % (tag, call)], rtol=1e-3, atol=1e-7)
        np.testing.assert_allclose(rb.adam_v.cpu().numpy()[1], d["%s_c%d_v" % (tag, call)], rtol=1e-3, atol=1e-12)
        diff = np.abs(rb.params.cpu().numpy()[2] - d["%s_c%d_w" % (tag, call)])
        assert (diff > 5e-6).mean() < 0.002 and diff.max() <= 4.1e-4, (float((diff > 5e-6).mean()), float(diff.max()))
        assert rb.step == call + 1


def test_train_many_games_vs_oracle_and_init():
    """Independent weights per game (device init), ragged n: device == numpy oracle per game."""
    G, n = 40, 777
    rb = _rb(G, gamma=0.9, entropy=0.003, seed=4).init()
    w0 = rb.params.cpu().numpy().copy()
    bound2 = 1.0 / 16.0
    assert np.abs(w0[:, :512]).max() <= 1.0 and np.abs(w0[:, 512:]).max() <= bound2 + 1e-7
    assert abs(w0[:, :512].mean()) < 0.02 and abs(w0[:, 512:].std() - bound2 / np.sqrt(3)) < 0.002
    assert not np.array_equal(w0[0], w0[1])
    rs = np.random.RandomState(2)
    price = rs.randint(20, 61, (n, G)) / 10.0
    action = rs.randint(0, 21, (n, G))
    reward = rs.uniform(5, 15, (n, G))
    g = rb.train(price, action, reward, want_grad=True).cpu().numpy()
    w1 = rb.params.cpu().numpy()
    for k in (0, 7, 39):
        ow, om, ov, os_, og = NN.train_net(w0[k], np.zeros(rb.P, np.float32), np.zeros(rb.P, np.float32), 0, 21,
                                           price[:, k], action[:, k], reward[:, k], 0.9, 0.003)
        np.testing.assert_allclose(g[k], og, rtol=2e-4, atol=2e-6)
        diff = np.abs(w1[k] - ow)
        assert (diff > 5e-6).mean() < 0.002 and diff.max() <= 4.1e-4


def _gradients_float64(w, A, price, action, G, entropy_coef):
    """nn_oracle.gradients (agents.py:183-192) evaluated in float64 from the float32 weights, prices and z-scored returns."""
    w1, b1, W2, b2 = [np.asarray(v, np.float64) for v in NN.split(w, A)]
    x = np.asarray(price, np.float64).astype(np.float32).astype(np.float64)[:, None]
    pre32 = (x.astype(np.float32) * NN.split(w, A)[0][None, :] + NN.split(w, A)[1][None, :])      # the activation test is float32's
    pre = x * w1[None, :] + b1[None, :]
    h = np.where(pre32 > 0, pre, 0.0)
    z = h @ W2.T + b2[None, :]
    z = z - z.max(axis=1, keepdims=True)
    p = np.exp(z); p = p / p.sum(axis=1, keepdims=True)
    n = len(p)
    onehot = np.zeros((n, A)); onehot[np.arange(n), np.asarray(action, np.int64)] = 1
    eps = np.finfo(np.float32).eps
    logp = np.log(np.clip(p, eps, 1 - eps))
    Hn = -(p * logp).sum(axis=1, keepdims=True)
    dz = (np.asarray(G, np.float64)[:, None] * (p - onehot) + entropy_coef * p * (logp + Hn)) / n
    dh = (dz @ W2) * (pre32 > 0)
    g = np.concatenate([(dh * x).sum(axis=0), dh.sum(axis=0), (dz.T @ h).ravel(), dz.sum(axis=0)])
    return g * min(1.0, 1.0 / (np.sqrt((g ** 2).sum()) + 1e-6))


@pytest.mark.parametrize("A,states", [(21, 41), (21, 400), (21, 448), (21, 0), (21, -1), (30, 41), (30, 300), (30, 0), (5, 130), (8, 64), (24, 65)])
def test_train_net_every_update_path_vs_oracle(A, states):
    """The update kernel's paths against the numpy restatement of Reinforce.train_net (agents.py:170-194), per game:
    folded on the piecewise-linear form with 1 / 2 / 7 / 16 chunks of distinct states (states = how many distinct prices the
    batch visits; 0 = continuous prices as in a game with env noise: every transition its own state), the plain per-transition
    path (states = -1: 1,300 transitions with continuous prices, more distinct states than the fold takes), both row paddings
    (A <= 24, A <= 32); units with w1 = 0 and thresholds on / beyond the visited prices are planted in game 0."""
    from th_rl_amd.nn import ReinforceBatch
    G, n = 5, (1300 if states < 0 else 1000)
    rb = ReinforceBatch(G, actions=A, gamma=0.93, entropy=0.004, seed=11).init()
    w = rb.params.cpu().numpy().copy()
    w[0, 0:4] = 0.0                                  # fc1.weight = 0: unit active everywhere or nowhere (by its bias)
    w[0, 256:260] = [0.5, -0.5, 0.0, 1e-3]
    w[0, 4:8] = [1.0, -1.0, 2.0, -2.0]               # thresholds exactly on visited prices, below and above the range
    w[0, 260:264] = [-3.0, 3.0, -1.0, 14.0]
    rb.set_params(w)
    w0 = rb.params.cpu().numpy().copy()
    rs = np.random.RandomState(100 + A + states)
    if states > 0:
        grid = np.sort(rs.choice(np.arange(500, 6500), states, replace=False)) / 1000.0
        grid[:2] = [0.5, 3.0]                       # (0.5 * 2 - 1 = 0, 3 * (-1) + 3 = 0: pre-activations of exactly zero)
        price = grid[rs.randint(0, states, (n, G))]
        price[:states, :] = grid[:, None]           # every state is visited
    else:
        price = rs.uniform(0.5, 6.5, (n, G))
    action = rs.randint(0, A, (n, G))
    reward = rs.uniform(5, 15, (n, G))
    g = rb.train(price, action, reward, want_grad=True).cpu().numpy()
    w1 = rb.params.cpu().numpy()
    for k in range(G):
        ow, om, ov, os_, og = NN.train_net(w0[k], np.zeros(rb.P, np.float32), np.zeros(rb.P, np.float32), 0, A,
                                           price[:, k], action[:, k], reward[:, k], 0.93, 0.004)
        # the float32 restatement carries its own summation error (1,000-term float32 sums): 4e-6 absolute on a unit-norm gradient
        np.testing.assert_allclose(g[k], og, rtol=2e-4, atol=4e-6, err_msg="game %d" % k)
        # ... so the same formula in float64 is the sharper check for the folded path, whose sums are exact.  One float32
        # rounding is free: the mean the returns are z-scored with (~140 here, so one ulp moves every return by ~1e-6 sd, agents.py:182).
        # It shifts the gradient along a known direction (d g / d shift, by a difference quotient); the shift is fitted and
        # bounded, the rest must agree tightly.
        Gz = NN.discounted_returns(reward[:, k], 0.93).astype(np.float64)
        g64 = _gradients_float64(w0[k], A, price[:, k], action[:, k], Gz, 0.004)
        if states >= 0:
            direction = (_gradients_float64(w0[k], A, price[:, k], action[:, k], Gz + 1e-4, 0.004) - g64) / 1e-4
            shift = float(direction @ (g[k] - g64) / (direction @ direction))
            assert abs(shift) < 5e-6, (k, shift)
            np.testing.assert_allclose(g[k], g64 + shift * direction, rtol=2e-5, atol=2e-7, err_msg="game %d (float64)" % k)
        diff = np.abs(w1[k] - ow)
        assert (diff > 5e-6).mean() < 0.002 and diff.max() <= 4.1e-4, (k, float((diff > 5e-6).mean()), float(diff.max()))


def test_nn_errors():
    from th_rl_amd._lib import ThrlError
    from th_rl_amd.nn import ReinforceBatch
    with pytest.raises(ThrlError, match="actions"):
        ReinforceBatch(4, actions=40)
    rb = _rb(2).init()
    with pytest.raises(ThrlError, match="transitions"):
        rb.train(np.zeros((2000, 2)), np.zeros((2000, 2), int), np.zeros((2000, 2)))


# ---------------------------------------------------------------- mixed games (QTable vs Reinforce)
from oracle import oracle as O  # noqa: E402

Q_AGENT = dict(name="QTable", gamma=0.95, actions=21, states=100, alpha=0.1, eps_end=0.001, epsilon=0.5,
               eps_step=0.9995, action_range=[0.2, 0.4])
R_AGENT = {"name": "Reinforce", "gamma": 0.995, "actions": 21, "states": 1, "action_range": [0.2, 0.4]}
ENV = dict(name="NoisyPriceState", noise_prob=0, a=10, b=1, nplayers=2, max_steps=100)


@pytest.mark.parametrize("mixed_fused", [False, True])
@pytest.mark.parametrize("dtype,noise", [("float32", 0.0), ("float64", 0.05)])
def test_unfused_operator_loop_equals_fused_kernels(dtype, noise, mixed_fused):
    """An all-QTable game run through the unfused operator loop (one launch per reference call)
    gives bit-identical tables / counters / state / per-game logs to the fused kernels."""
    from th_rl_amd.batched import GameBatch
    from th_rl_amd.mixed import MixedGameBatch
    config = {"agents": [dict(Q_AGENT), dict(Q_AGENT, alpha=0.3)], "environment": dict(ENV, noise_prob=noise, max_steps=40)}
    for a in config["agents"]:
        a["min_memory"] = 40
    G, E = 9, 3
    fused = GameBatch(config, n_games=G, dtype=dtype, kernel="generic", seed=5).init_tables()
    mixed = MixedGameBatch(config, n_games=G, dtype=dtype, seed=5)
    mixed.set_tables(fused.tables_numpy(), fused.states_numpy())
    rf = fused.run(E, per_game_logs=True)
    rm = mixed.run(E, fused=mixed_fused)
    assert np.array_equal(mixed.tables_numpy(), fused.tables_numpy())
    assert np.array_equal(mixed.counters_numpy(), fused.counters_numpy())
    assert np.array_equal(mixed.states_numpy(), fused.states_numpy())
    assert np.array_equal(rm["game_reward_log"], rf["game_reward_log"])
    assert np.array_equal(rm["game_action_log"], rf["game_action_log"])
    assert mixed.eps[:2] == fused.eps[:2]


@pytest.mark.parametrize("dtype,noise,order", [("float64", 0.0, "qr"), ("float32", 0.1, "rq"), ("float64", 0.0, "rr")])
def test_mixed_episode_kernel_equals_operator_loop(dtype, noise, order):
    """thrl_mixed_episodes (one launch per run of episodes between network updates) against the
    per-call operator loop on the same seeds: tables, counters, state, per-game logs, network
    parameters and Adam state after several updates, replay-buffer bookkeeping -- all bit-identical.
    Covers a ring buffer that wraps (capacity < what an episode appends before training)."""
    from th_rl_amd.mixed import MixedGameBatch
    T = 25
    q = dict(Q_AGENT, min_memory=T, capacity=40)
    r = dict(R_AGENT, min_memory=60, entropy=0.01)
    agents = {"qr": [q, r], "rq": [r, dict(q, min_memory=2 * T, capacity=30)], "rr": [r, dict(r, min_memory=T)]}[order]
    config = {"agents": [dict(a) for a in agents], "environment": dict(ENV, max_steps=T, noise_prob=noise)}
    G, E = 5, 8
    a = MixedGameBatch(config, n_games=G, dtype=dtype, seed=21, game_offset=3).init_tables()
    b = MixedGameBatch(config, n_games=G, dtype=dtype, seed=21, game_offset=3).init_tables()
    ra = a.run(3, fused=True); ra2 = a.run(E - 3, fused=True)       # split: state carries across calls
    rb = b.run(E, fused=False)
    assert ra["kernel"] == "mixed-fused" and rb["kernel"] == "unfused"
    assert np.array_equal(np.concatenate([ra["game_reward_log"], ra2["game_reward_log"]]), rb["game_reward_log"])
    assert np.array_equal(np.concatenate([ra["game_action_log"], ra2["game_action_log"]]), rb["game_action_log"])
    assert np.array_equal(a.tables_numpy(), b.tables_numpy())
    assert np.array_equal(a.counters_numpy(), b.counters_numpy())
    assert np.array_equal(a.states_numpy(), b.states_numpy())
    assert a.eps == b.eps and a.count == b.count and a.episode == b.episode
    for i in a.nn:
        assert a.nn[i].step == b.nn[i].step and a.nn[i].step >= 2
        assert np.array_equal(a.nn[i].params.cpu().numpy(), b.nn[i].params.cpu().numpy())
        assert np.array_equal(a.nn[i].adam_m.cpu().numpy(), b.nn[i].adam_m.cpu().numpy())
        assert np.array_equal(a.nn[i].adam_v.cpu().numpy(), b.nn[i].adam_v.cpu().numpy())


def test_qtable_vs_reinforce_game_against_composed_oracle():
    """The reference's own pairing (QTable vs Reinforce): the device loop against a Python loop
    composed from the pinned oracles (C oracle for env / encode / TD, numpy oracle for the MLP),
    fed the same Philox draws.  Tables are bit-identical; network parameters to float32 rounding."""
    import ctypes
    from th_rl_amd.mixed import MixedGameBatch
    T, mm = 30, 60
    config = {"agents": [dict(Q_AGENT, min_memory=T), dict(R_AGENT, min_memory=mm, entropy=0.01)],
              "environment": dict(ENV, max_steps=T)}
    G, E = 3, 5
    mb = MixedGameBatch(config, n_games=G, dtype="float64", seed=11).init_tables()
    q0, s0 = mb.tables_numpy().copy(), mb.states_numpy().copy()
    w0 = mb.nn[1].params.cpu().numpy().copy()
    out = mb.run(E)
    # ---- composed oracle
    qcfg, eps0 = O.cfg_from_config({"agents": [config["agents"][0], dict(Q_AGENT, states=1, actions=21)],
                                    "environment": config["environment"]}, 1, 1)
    for g in range(G):
        table = q0[g, :2121].reshape(101, 21).copy(); counter = np.zeros((101, 21), np.int32)
        w = w0[g].copy(); m = np.zeros_like(w); v = np.zeros_like(w); step = 0
        price = s0[g]; eps = 0.5
        memq = []; memr = []
        for e in range(E):
            rl = np.zeros(2); al = np.zeros(2)
            for t in range(T):
                ctr = [t, e, g, 0]
                xs = O.philox(ctr, [11, 0])
                u0 = xs[0] * 2.0 ** -32; c0 = (xs[1] * 21) >> 32; u1 = xs[2] * 2.0 ** -32
                if u0 < eps:
                    a0 = c0
                else:
                    a0 = int(np.argmax(table[O.encode32(price, 10, 100)]))
                a1 = int(NN.sample_action(w, 21, [price], [u1])[0])
                sc0 = O.scale(a0, 21, 0.2, 0.4); sc1 = NN.scale(a1, 21, 0.2, 0.4)
                nprice, rew = O.env_step(qcfg, [sc0, sc1])
                memq.append((O.encode64(price, 10, 100), a0, rew[0], O.encode64(nprice, 10, 100)))
                memr.append((price, a1, rew[1]))
                rl += rew / T; al += np.array([sc0, sc1]) / T
                price = nprice
            if len(memq) >= T:
                st, ac, rw, ns = zip(*memq)
                O.td_update(table, counter, st, ac, rw, ns, 0.1, 0.95); memq = []
            eps = 0.001 + (eps - 0.001) * 0.9995
            if len(memr) >= mm:
                pr, ac, rw = zip(*memr)
                w, m, v, step, _ = NN.train_net(w, m, v, step, 21, pr, ac, rw, 0.995, 0.01); memr = []
            if e == 0:
                np.testing.assert_allclose(out["game_reward_log"][e, :, g], rl, rtol=1e-13)
        # the float32 policy can flip an inverse-CDF draw after an update, so compare the parts that
        # are insensitive to it only when the whole action sequence agreed
        if np.array_equal(mb.table(g, 0), table):
            assert np.array_equal(mb.counter_of(g, 0), counter.astype(np.float64))
            diff = np.abs(mb.nn[1].params[g].cpu().numpy() - w)
            assert (diff > 1e-5).mean() < 0.01 and diff.max() <= 1e-3, (float((diff > 1e-5).mean()), float(diff.max()))
            matched = True
        else:
            matched = False
        assert matched or E > 2     # before the first network update (episode 2) everything is exact
    assert mb.nn[1].step == 2 and mb.count[1] == 30


def test_train_one_reference_example_config(tmp_path):
    """The reference's shipped example_config.json pairing (QTable vs Reinforce) trains end to end;
    artefacts keep the reference's formats (agents.py:110-112, 215-216; trainer.py:101-110)."""
    import json
    import torch
    from th_rl_amd import trainer, utils
    config = {"agents": [dict(Q_AGENT), dict(R_AGENT)], "environment": dict(ENV),
              "training": {"print_freq": 500, "epochs": 20, "seed": 1}}
    p = tmp_path / "example_config.json"
    p.write_text(json.dumps(config, indent=3))
    exp = str(tmp_path / "run")
    np.random.seed(0); torch.manual_seed(0)
    trainer.train_one(exp, str(p))
    assert sorted(os.listdir(exp)) == ["0.npy", "0_counter.npy", "1", "config.json", "log.csv"]
    sd = torch.load(os.path.join(exp, "1"), weights_only=True)
    assert sorted(sd) == ["fc1.bias", "fc1.weight", "fc_pi.bias", "fc_pi.weight"]
    assert sd["fc_pi.weight"].shape == (21, 256) and sd["fc1.weight"].shape == (256, 1)
    assert np.load(os.path.join(exp, "0_counter.npy")).sum() == 20 * 100
    np.random.seed(0); torch.manual_seed(0)
    _, agents0, _ = trainer.create_game(str(p))
    assert not torch.equal(sd["fc_pi.weight"], agents0[1].state_dict()["fc_pi.weight"])     # it was trained (2 updates)
    cfg2, agents, env, actions, rewards = utils.load_experiment(exp)
    assert list(rewards.columns) == ["QTable0", "Reinforce1"] and len(rewards) == 21
    assert torch.equal(agents[1].state_dict()["fc1.weight"], sd["fc1.weight"])


def test_reinforce_protocol_methods_vs_reference_fixture():
    """agents.Reinforce's object-level methods (pi / get_action / memory.append + train_net) on the
    device reproduce the reference fixture G7."""
    import torch
    from th_rl_amd.agents import Reinforce
    d = np.load(GOLDEN)
    torch.manual_seed(3)
    ag = Reinforce(gamma=0.995, actions=21, states=1, action_range=[0.2, 0.4])
    assert np.array_equal(ag.flat_params(), d["cfg_w0"])          # same torch default init under the same seed
    probe = d["cfg_probe_price"]
    for k in (0, 4, 8):
        np.testing.assert_allclose(ag.pi(torch.tensor([probe[k]], dtype=torch.float32)).numpy(), d["cfg_probe_prob0"][k],
                                   rtol=2e-5, atol=1e-8)
    assert [ag.scale(k) for k in range(21)] == list(d["cfg_scale"])
    for t in range(1000):
        ag.memory.append(np.array([d["cfg_c0_price"][t]]), np.int64(d["cfg_c0_action"][t]), float(d["cfg_c0_reward"][t]),
                         True, np.array([d["cfg_c0_price"][t + 1]]))
    ag.train_net()
    assert len(ag.memory) == 0
    diff = np.abs(ag.flat_params() - d["cfg_c0_w"])
    assert (diff > 5e-6).mean() < 0.002 and diff.max() <= 4.1e-4
    a = ag.sample_action(torch.tensor([3.4], dtype=torch.float32))
    assert 0 <= a <= 20 and isinstance(ag.get_action(np.array([3.4])), int)


# ------------------------------------------------------------------------------------------------
# ActorCritic (reference agents.py:222-330), fixture G8 (tests/golden/g8_actorcritic.npz).
# Tolerances as for Reinforce, gradients rtol 5e-4 (the [N,N] advantage sums 10^6 float32 terms).
GOLDEN_AC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_actorcritic.npz")
CASES_AC = {"cfg": dict(gamma=0.98, entropy=0.0, actions=21), "ent": dict(gamma=0.9, entropy=0.01, actions=15)}


def _ac(G, actions, **kw):
    from th_rl_amd.nn import ActorCriticBatch
    return ActorCriticBatch(G, actions=actions, **kw)


@pytest.mark.parametrize("tag", sorted(CASES_AC))
def test_actorcritic_act_and_train_match_reference(tag):
    import torch
    d = np.load(GOLDEN_AC)
    kw = CASES_AC[tag]; A = kw["actions"]
    ab = _ac(3, A, gamma=kw["gamma"], entropy=kw["entropy"]).set_params(d[tag + "_w0"])
    assert ab.P == NN.ac_n_params(A)
    probe = d[tag + "_probe_price"]
    for j, pr in enumerate(probe):
        _, probs = ab.act(np.full(3, pr), want_probs=True)
        np.testing.assert_allclose(probs.cpu().numpy()[1], d[tag + "_probe_prob0"][j], rtol=2e-5, atol=1e-8)
    tile = lambda x: np.repeat(np.asarray(x)[:, None], 3, axis=1)
    for call in range(2):
        if call == 1:        # start the second call from the reference's own state
            ab.set_params(d[tag + "_c0_w"])
            ab.adam_m.copy_(torch.from_numpy(np.broadcast_to(d[tag + "_c0_m"], (3, ab.P)).copy()))
            ab.adam_v.copy_(torch.from_numpy(np.broadcast_to(d[tag + "_c0_v"], (3, ab.P)).copy()))
            ab.step = 1
        pr = d["%s_c%d_price" % (tag, call)]
        g = ab.train(tile(pr[:1000]), tile(d["%s_c%d_action" % (tag, call)]), tile(d["%s_c%d_reward" % (tag, call)]),
                     want_grad=True, next_price=tile(pr[1:1001])).cpu().numpy()
        for k in range(3):
            np.testing.assert_allclose(g[k], d["%s_c%d_grad" % (tag, call)], rtol=5e-4, atol=2e-6)
        # Adam moments start from the reference's own state, so their error is the gradient's
        # (|dg| <= 2e-6 + 5e-4|g|) propagated: m = .9 m0 + .1 g,  v = .999 v0 + .001 g^2
        gr = np.abs(d["%s_c%d_grad" % (tag, call)].astype(np.float64))
        dg = 2e-6 + 5e-4 * gr
        dm = np.abs(ab.adam_m.cpu().numpy()[1].astype(np.float64) - d["%s_c%d_m" % (tag, call)])
        dv = np.abs(ab.adam_v.cpu().numpy()[1].astype(np.float64) - d["%s_c%d_v" % (tag, call)])
        assert np.all(dm <= 0.1 * dg + 1e-9), float((dm - 0.1 * dg).max())
        assert np.all(dv <= 0.001 * (2 * gr * dg + dg * dg) + 1e-14), float((dv - 0.001 * (2 * gr * dg + dg * dg)).max())
        diff = np.abs(ab.params.cpu().numpy()[2] - d["%s_c%d_w" % (tag, call)])
        assert (diff > 5e-6).mean() < 0.002 and diff.max() <= 4.1e-4, (float((diff > 5e-6).mean()), float(diff.max()))
    for j, pr in enumerate(probe):
        a = ab.set_params(d[tag + "_c1_w"]).act(np.full(3, pr))
        assert int(a.cpu()[0]) == int(d[tag + "_probe_greedy2"][j])


@pytest.mark.parametrize("A,states", [(21, 41), (21, 300), (15, 64), (21, 448), (21, 0), (24, 41), (30, 41)])
def test_actorcritic_train_every_update_path_vs_oracle(A, states):
    """ActorCritic.train_net (agents.py:280-305) through the update kernel's paths against the numpy restatement, per game:
    states > 0: the folded path on the piecewise-linear form (states and next states folded together, the value head as one
    more output column; 1 / 2 / 5 / 7 chunks); states = 0: continuous prices -> the plain per-transition path; A = 24: no free
    column for the value head -> plain path; A = 30: the wider row padding."""
    G, n = 4, 1000
    ab = _ac(G, A, gamma=0.9, entropy=0.005, seed=5).init()
    w0 = ab.params.cpu().numpy().copy()
    rs = np.random.RandomState(300 + A + states)
    if states:
        grid = np.sort(rs.choice(np.arange(500, 6500), states, replace=False)) / 1000.0
        idx = rs.randint(0, states, (n + 1, G)); idx[:states, :] = np.arange(states)[:, None]
        price, nprice = grid[idx[:-1]], grid[idx[1:]]
    else:
        pr = rs.uniform(0.5, 6.5, (n + 1, G)); price, nprice = pr[:-1], pr[1:]
    action = rs.randint(0, A, (n, G)); reward = rs.uniform(5, 15, (n, G))
    g = ab.train(price, action, reward, want_grad=True, next_price=nprice).cpu().numpy()
    w1 = ab.params.cpu().numpy()
    for k in range(G):
        ow, om, ov, os_, og = NN.ac_train_net(w0[k], np.zeros(ab.P, np.float32), np.zeros(ab.P, np.float32), 0, A,
                                              price[:, k], action[:, k], reward[:, k], nprice[:, k], 0.9, 0.005)
        np.testing.assert_allclose(g[k], og, rtol=5e-4, atol=4e-6, err_msg="game %d" % k)
        diff = np.abs(w1[k] - ow)
        assert (diff > 5e-6).mean() < 0.002 and diff.max() <= 4.1e-4, (k, float((diff > 5e-6).mean()), float(diff.max()))



def test_actorcritic_many_games_vs_oracle_and_init():
    """Independent weights per game (device init: fc_v bias 1000), ragged n, 32-action policy."""
    G, n, A = 24, 613, 29
    ab = _ac(G, A, gamma=0.93, entropy=0.002, seed=8).init()
    w0 = ab.params.cpu().numpy().copy()
    P = NN.n_params(A)
    assert np.all(w0[:, P + 256] == 1000.0) and np.abs(w0[:, P:P + 256]).max() <= 1.0 / 16.0 + 1e-7
    assert not np.array_equal(w0[0, P:P + 256], w0[1, P:P + 256])
    rs = np.random.RandomState(5)
    price = rs.randint(20, 61, (n + 1, G)) / 10.0
    action = rs.randint(0, A, (n, G))
    reward = rs.uniform(5, 15, (n, G))
    g = ab.train(price[:n], action, reward, want_grad=True, next_price=price[1:]).cpu().numpy()
    w1 = ab.params.cpu().numpy()
    for k in (0, 11, 23):
        ow, om, ov, os_, og = NN.ac_train_net(w0[k], np.zeros(ab.P, np.float32), np.zeros(ab.P, np.float32), 0, A,
                                              price[:n, k], action[:, k], reward[:, k], price[1:, k], 0.93, 0.002)
        np.testing.assert_allclose(g[k], og, rtol=5e-4, atol=2e-6)
        diff = np.abs(w1[k] - ow)
        assert (diff > 5e-6).mean() < 0.002 and diff.max() <= 4.1e-4


def test_actorcritic_game_fused_equals_operator_loop_and_trains(tmp_path):
    """QTable vs ActorCritic through the fused episode kernel == the per-call operator loop
    (bit-identical, incl. the value-head parameters after two updates); train_one writes the
    reference's artefacts and the ActorCritic state_dict loads back."""
    import json
    import torch
    from th_rl_amd import trainer
    from th_rl_amd.mixed import MixedGameBatch
    T = 20
    q = dict(Q_AGENT, min_memory=T)
    ac = {"name": "ActorCritic", "gamma": 0.98, "actions": 21, "states": 1, "action_range": [0.2, 0.4],
          "min_memory": 50, "entropy": 0.01}
    config = {"agents": [q, ac], "environment": dict(ENV, max_steps=T)}
    a = MixedGameBatch(config, n_games=4, dtype="float64", seed=2).init_tables()
    b = MixedGameBatch(config, n_games=4, dtype="float64", seed=2).init_tables()
    ra, rb = a.run(7, fused=True), b.run(7, fused=False)
    assert a.nn[1].step == b.nn[1].step == 2
    assert np.array_equal(ra["game_reward_log"], rb["game_reward_log"])
    assert np.array_equal(a.tables_numpy(), b.tables_numpy()) and np.array_equal(a.counters_numpy(), b.counters_numpy())
    assert np.array_equal(a.nn[1].params.cpu().numpy(), b.nn[1].params.cpu().numpy())
    cfg = dict(config, training={"epochs": 6, "print_freq": 3, "seed": 4})
    (tmp_path / "c.json").write_text(json.dumps(cfg))
    trainer.train_one(str(tmp_path / "run"), str(tmp_path / "c.json"))
    sd = torch.load(tmp_path / "run" / "1", weights_only=True)
    assert sorted(sd) == ["fc1.bias", "fc1.weight", "fc_pi.bias", "fc_pi.weight", "fc_v.bias", "fc_v.weight"]
    assert sd["fc_v.weight"].shape == (1, 256) and abs(float(sd["fc_v.bias"]) - 1000.0) < 0.01
    assert (tmp_path / "run" / "0.npy").exists() and (tmp_path / "run" / "log.csv").exists()


def test_three_agent_mix_and_unsupported_fallback():
    """Three agents (QTable, Reinforce, QTable with its own grid) take the fused kernel's NA=8
    instantiation: bit-identical to the operator loop.  Three NEURAL agents are beyond the fused
    kernel (two networks fit in registers): fused=True raises thrl_err -3, the default falls back."""
    from th_rl_amd._lib import ThrlError
    from th_rl_amd.mixed import MixedGameBatch
    T = 24
    env = dict(ENV, max_steps=T, nplayers=3, noise_prob=0.2)
    q0 = dict(Q_AGENT, min_memory=T)
    q2 = dict(Q_AGENT, min_memory=2 * T, actions=11, states=50, alpha=0.2, action_range=[0.1, 0.3], capacity=100)
    r = dict(R_AGENT, min_memory=2 * T, actions=17, entropy=0.01)
    config = {"agents": [q0, r, q2], "environment": env}
    a = MixedGameBatch(config, n_games=6, dtype="float32", seed=13).init_tables()
    b = MixedGameBatch(config, n_games=6, dtype="float32", seed=13).init_tables()
    ra, rb = a.run(5, fused=True), b.run(5, fused=False)
    assert ra["kernel"] == "mixed-fused" and a.nn[1].step == b.nn[1].step == 2
    assert np.array_equal(ra["game_reward_log"], rb["game_reward_log"])
    assert np.array_equal(ra["game_action_log"], rb["game_action_log"])
    assert np.array_equal(a.tables_numpy(), b.tables_numpy()) and np.array_equal(a.counters_numpy(), b.counters_numpy())
    assert np.array_equal(a.states_numpy(), b.states_numpy()) and a.eps == b.eps and a.count == b.count
    assert np.array_equal(a.nn[1].params.cpu().numpy(), b.nn[1].params.cpu().numpy())
    three = {"agents": [dict(r), dict(r), dict(r)], "environment": env}
    c = MixedGameBatch(three, n_games=2, dtype="float32", seed=1).init_tables()
    with pytest.raises(ThrlError, match="more than two discrete neural agents"):
        c.run(1, fused=True)
    assert c.run(3)["kernel"] == "unfused" and c.nn[0].step == 1


# ------------------------------------------------------------------------------------------------
# CAC, the continuous actor-critic (reference agents.py:333-442), fixture G9 (g9_cac.npz).
GOLDEN_CAC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_cac.npz")
CASES_CAC = {"cfg": dict(gamma=0.98, entropy=0.0), "ent": dict(gamma=0.9, entropy=0.01)}


def _cac(G, **kw):
    from th_rl_amd.nn import CACBatch
    return CACBatch(G, **kw)


@pytest.mark.parametrize("tag", sorted(CASES_CAC))
def test_cac_heads_and_train_match_reference(tag):
    """pi()/v() heads rtol 2e-5; clipped gradients of the doubly-broadcast loss rtol 1e-3 + atol 3e-6;
    parameters after each Adam step as for the other agents."""
    import torch
    d = np.load(GOLDEN_CAC)
    kw = CASES_CAC[tag]
    cb = _cac(3, gamma=kw["gamma"], entropy=kw["entropy"]).set_params(d[tag + "_c1_w"])
    probe = d[tag + "_probe_price"]
    for j, pr in enumerate(probe):
        a, (mu, sd, v) = cb.act(np.full(3, pr), want_heads=True)
        np.testing.assert_allclose(float(mu[1]), d[tag + "_probe_mu2"][j], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(float(sd[1]), d[tag + "_probe_std2"][j], rtol=2e-5)
        np.testing.assert_allclose(float(v[1]), d[tag + "_probe_value2"][j], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(float(a[1]), 1.0 / (1.0 + np.exp(-float(mu[1]))), rtol=1e-6)   # mean action
    tile = lambda x: np.repeat(np.asarray(x)[:, None], 3, axis=1)
    for call in range(2):
        cb.set_params(d[tag + "_w0"] if call == 0 else d[tag + "_c0_w"])
        if call == 1:
            cb.adam_m.copy_(torch.from_numpy(np.broadcast_to(d[tag + "_c0_m"], (3, cb.P)).copy()))
            cb.adam_v.copy_(torch.from_numpy(np.broadcast_to(d[tag + "_c0_v"], (3, cb.P)).copy()))
        cb.step = call
        pr = d["%s_c%d_price" % (tag, call)]
        g = cb.train(tile(pr[:1000]), tile(d["%s_c%d_action" % (tag, call)]), tile(d["%s_c%d_reward" % (tag, call)]),
                     want_grad=True, next_price=tile(pr[1:1001])).cpu().numpy()
        for k in range(3):
            np.testing.assert_allclose(g[k], d["%s_c%d_grad" % (tag, call)], rtol=1e-3, atol=3e-6)
        diff = np.abs(cb.params.cpu().numpy()[2] - d["%s_c%d_w" % (tag, call)])
        assert (diff > 5e-6).mean() < 0.005 and diff.max() <= 4.1e-4, (float((diff > 5e-6).mean()), float(diff.max()))


def test_cac_many_games_sampling_and_train_vs_oracle():
    G, n = 50, 391
    cb = _cac(G, gamma=0.95, entropy=0.004, seed=6).init()
    w0 = cb.params.cpu().numpy().copy()
    assert np.abs(w0[:, :512]).max() <= 1.0 and np.abs(w0[:, 512:]).max() <= 1.0 / 16.0 + 1e-7
    assert not np.array_equal(w0[0], w0[1])
    rs = np.random.RandomState(9)
    price = rs.uniform(2, 6, G); u1 = rs.uniform(0, 1, G); u2 = rs.uniform(0, 1, G)
    a = cb.act(price, u1=u1, u2=u2).cpu().numpy()
    ref = np.array([NN.cac_sample_action(w0[k], [price[k]], [u1[k]], [u2[k]])[0] for k in range(G)])
    np.testing.assert_allclose(a, ref, rtol=2e-5, atol=1e-6)
    P = rs.randint(20, 61, (n + 1, G)) / 10.0
    act = rs.uniform(0.01, 0.99, (n, G)).astype(np.float32)
    rew = rs.uniform(5, 15, (n, G))
    g = cb.train(P[:n], act, rew, want_grad=True, next_price=P[1:]).cpu().numpy()
    w1 = cb.params.cpu().numpy()
    for k in (0, 17, 49):
        ow, om, ov, os_, og = NN.cac_train_net(w0[k], np.zeros(cb.P, np.float32), np.zeros(cb.P, np.float32), 0,
                                               P[:n, k], act[:, k], rew[:, k], P[1:, k], 0.95, 0.004)
        np.testing.assert_allclose(g[k], og, rtol=1e-3, atol=3e-6)
        diff = np.abs(w1[k] - ow)
        assert (diff > 5e-6).mean() < 0.005 and diff.max() <= 4.1e-4


def test_qtable_vs_cac_game_and_train_one(tmp_path):
    """A QTable-vs-CAC game (fused episode kernel).  With T = 1 the per-game action log is the
    scaled action of the single step: checked against the oracle on the same Philox draws.
    train_one writes the reference's artefacts incl. the CAC state_dict."""
    import json
    import torch
    from th_rl_amd import trainer
    from th_rl_amd.mixed import MixedGameBatch
    cac = {"name": "CAC", "gamma": 0.98, "states": 1, "action_range": [0.2, 0.4], "min_memory": 6, "entropy": 0.01}
    config = {"agents": [dict(Q_AGENT, min_memory=1), cac], "environment": dict(ENV, max_steps=1)}
    G = 5
    mb = MixedGameBatch(config, n_games=G, dtype="float64", seed=31).init_tables()
    w0 = mb.nn[1].params.cpu().numpy().copy(); s0 = mb.states_numpy().copy()
    out = mb.run(8)
    assert out["kernel"] == "mixed-fused" and mb.nn[1].step == 1 and mb.count[1] == 2
    for g in range(G):
        xs = O.philox([0, 0, g, 0], [31, 0])
        a = NN.cac_sample_action(w0[g], [s0[g]], [xs[2] * 2.0 ** -32], [xs[3] * 2.0 ** -32])[0]
        np.testing.assert_allclose(out["game_action_log"][0, 1, g], NN.cac_scale(float(a), 0.2, 0.4), rtol=2e-6)
    al = out["game_action_log"][:, 1, :]
    assert np.all((al > 0.2) & (al < 0.4)) and not np.array_equal(w0, mb.nn[1].params.cpu().numpy())
    cfg = {"agents": [dict(Q_AGENT, min_memory=10), dict(cac, min_memory=20)], "environment": dict(ENV, max_steps=10),
           "training": {"epochs": 5, "print_freq": 5, "seed": 3}}
    (tmp_path / "c.json").write_text(json.dumps(cfg))
    trainer.train_one(str(tmp_path / "run"), str(tmp_path / "c.json"))
    sd = torch.load(tmp_path / "run" / "1", weights_only=True)
    assert sorted(sd) == ["fc1.bias", "fc1.weight", "fc_mu.bias", "fc_mu.weight", "fc_std.bias", "fc_std.weight",
                          "fc_v.bias", "fc_v.weight"]
    from th_rl_amd.agents import CAC
    ag = CAC(**{k: v for k, v in cac.items() if k != "name"})
    ag.load(str(tmp_path / "run" / "1"))
    a = ag.get_action(np.array([3.4]))
    assert 0.0 < a < 1.0 and 0.2 < ag.scale(a) < 0.4 and 0.0 < ag.sample_action(np.array([3.4])) < 1.0


def test_mixed_checkpoint_resume_is_exact(tmp_path):
    """QTable vs ActorCritic: 9 episodes in one go == 3 episodes, save, load into a fresh batch,
    6 more (tables, counters, state, epsilon, replay rings mid-fill, network + Adam state)."""
    from th_rl_amd.mixed import MixedGameBatch
    T = 20
    ac = {"name": "ActorCritic", "gamma": 0.98, "actions": 21, "states": 1, "action_range": [0.2, 0.4],
          "min_memory": 70, "entropy": 0.01}
    config = {"agents": [dict(Q_AGENT, min_memory=2 * T, capacity=50), ac], "environment": dict(ENV, max_steps=T)}
    a = MixedGameBatch(config, n_games=7, dtype="float32", seed=17).init_tables()
    ra = a.run(9)
    b = MixedGameBatch(config, n_games=7, dtype="float32", seed=17).init_tables()
    rb1 = b.run(3)
    assert b.count == [20, 60] and b.nn[1].step == 0 and b.episode == 3      # both replay rings are mid-fill
    b.save(str(tmp_path / "mixed.pt"))
    c = MixedGameBatch(config, n_games=7, dtype="float32", seed=999)            # seed comes from the checkpoint
    c.load(str(tmp_path / "mixed.pt"))
    rc = c.run(6)
    assert np.array_equal(np.concatenate([rb1["game_reward_log"], rc["game_reward_log"]]), ra["game_reward_log"])
    assert np.array_equal(c.tables_numpy(), a.tables_numpy()) and np.array_equal(c.counters_numpy(), a.counters_numpy())
    assert np.array_equal(c.states_numpy(), a.states_numpy()) and c.eps == a.eps and c.count == a.count
    assert c.episode == a.episode == 9 and c.nn[1].step == a.nn[1].step
    assert np.array_equal(c.nn[1].params.cpu().numpy(), a.nn[1].params.cpu().numpy())
    assert np.array_equal(c.nn[1].adam_v.cpu().numpy(), a.nn[1].adam_v.cpu().numpy())


@pytest.mark.parametrize("dtype,noise", [("float64", 0.0), ("float32", 0.15)])
def test_cac_in_fused_kernel_equals_operator_loop(dtype, noise):
    """CAC (continuous actions, network in LDS) next to a QTable and a Reinforce agent in one game:
    the fused kernel == the per-call operator loop, bit for bit, across several network updates."""
    from th_rl_amd.mixed import MixedGameBatch
    T = 16
    cac = {"name": "CAC", "gamma": 0.97, "states": 1, "action_range": [0.15, 0.45], "min_memory": 40, "entropy": 0.02}
    config = {"agents": [dict(cac), dict(Q_AGENT, min_memory=T), dict(R_AGENT, min_memory=3 * T)],
              "environment": dict(ENV, max_steps=T, nplayers=3, noise_prob=noise)}
    a = MixedGameBatch(config, n_games=5, dtype=dtype, seed=23, game_offset=2).init_tables()
    b = MixedGameBatch(config, n_games=5, dtype=dtype, seed=23, game_offset=2).init_tables()
    ra, rb = a.run(7, fused=True), b.run(7, fused=False)
    assert ra["kernel"] == "mixed-fused" and rb["kernel"] == "unfused"
    assert a.nn[0].step == b.nn[0].step == 2 and a.nn[2].step == b.nn[2].step == 2
    assert np.array_equal(ra["game_action_log"], rb["game_action_log"])
    assert np.array_equal(ra["game_reward_log"], rb["game_reward_log"])
    assert np.array_equal(a.tables_numpy(), b.tables_numpy()) and np.array_equal(a.counters_numpy(), b.counters_numpy())
    assert np.array_equal(a.states_numpy(), b.states_numpy()) and a.count == b.count and a.eps == b.eps
    for i in (0, 2):
        assert np.array_equal(a.nn[i].params.cpu().numpy(), b.nn[i].params.cpu().numpy())
    assert np.array_equal(a.buf[0]["action"].cpu().numpy(), b.buf[0]["action"].cpu().numpy())


def test_example_config_learning_statistics_match_the_reference():
    """The reference's example config (QTable vs Reinforce, 20,000 epochs) trained on the device for
    64 games at once (seconds) against the reference's own runs: fixture G11 (six seeded runs made
    here, ~750 s each on a CPU core) plus the two runs the reference ships (G10).  Random streams
    differ by construction, so the check is statistical: the 64-game mean of reward / action over
    the first and the last 1,000 epochs lies within 3 standard errors of the reference's mean
    (+ 1 % slack), per agent."""
    import json
    import tempfile
    import pandas
    from th_rl_amd import trainer
    g10 = np.load(os.path.join(os.path.dirname(GOLDEN), "g10_stored_run.npz"))
    g11 = np.load(os.path.join(os.path.dirname(GOLDEN), "g11_example_config_stats.npz"))
    cfg = json.load(open(os.path.join(os.path.dirname(GOLDEN), "ref_run_example_config", "config.json")))
    cfg["training"].update(n_games=64, seed=5, print_freq=20000)
    d = tempfile.mkdtemp()
    json.dump(cfg, open(os.path.join(d, "c.json"), "w"))
    trainer.train_one(os.path.join(d, "run"), os.path.join(d, "c.json"))
    a = pandas.read_csv(os.path.join(d, "run", "log.csv"), header=[0, 1]).to_numpy()
    assert a.shape == (20000, 4)
    for name, mine in (("first1000", a[:1000].mean(axis=0)), ("last1000", a[-1000:].mean(axis=0))):
        ref = np.concatenate([g11[name], g10["shipped0_" + name][None], g10["shipped1_" + name][None]])
        mean, se = ref.mean(axis=0), ref.std(axis=0, ddof=1) / np.sqrt(len(ref))
        assert np.all(np.abs(mine - mean) <= 3 * se + 0.01 * np.abs(mean)), (name, mine, mean, se)


@pytest.mark.parametrize("distinct", [3, 64, 65, 130, 320, 321, 500])
def test_reinforce_update_state_dedupe_and_plain_path(distinct):
    """The update folds transitions that share a state (up to 320 distinct states per batch, 64 per
    chunk) and takes the plain per-transition path beyond that; all against the numpy oracle, and
    deterministic."""
    G, n = 6, 700
    rs = np.random.RandomState(distinct)
    grid = np.sort(rs.uniform(2.0, 6.0, distinct))
    price = grid[rs.randint(0, distinct, (n, G))]
    price[:distinct, :] = grid[:, None]                      # every state occurs
    action = rs.randint(0, 21, (n, G))
    reward = rs.uniform(5, 15, (n, G))
    outs = []
    for rep in range(2):
        rb = _rb(G, gamma=0.97, entropy=0.01, seed=3).init()
        w0 = rb.params.cpu().numpy().copy()
        g = rb.train(price, action, reward, want_grad=True).cpu().numpy()
        outs.append((g, rb.params.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    for k in (0, 5):
        ow, om, ov, os_, og = NN.train_net(w0[k], np.zeros(rb.P, np.float32), np.zeros(rb.P, np.float32), 0, 21,
                                           price[:, k], action[:, k], reward[:, k], 0.97, 0.01)
        np.testing.assert_allclose(outs[0][0][k], og, rtol=2e-4, atol=2e-6)


def test_example_config_and_configs2_as_one_sweep_equal_two_separate_runs(tmp_path):
    """the reference's research loop is configs x runs, one process each (main.py:13-21), and
    its two shipped configs -- example_config.json and configs2.json, both QTable vs Reinforce -- differ
    in the QTable's gamma / alpha / epsilon and the Reinforce gamma.  Here they run as ONE batch with
    per-game arrays, and every game is bit for bit the game its own config trains alone (tables,
    counters, env state, network parameters and Adam state after two updates, per-game logs)."""
    import json
    import torch
    from th_rl_amd import trainer
    from th_rl_amd.mixed import MixedGameBatch
    ex = {"agents": [dict(Q_AGENT), dict(R_AGENT)], "environment": dict(ENV)}                       # example_config.json
    c2 = {"agents": [dict(Q_AGENT, gamma=0.35, alpha=0.5, epsilon=0.8), dict(R_AGENT, gamma=0.35)],
          "environment": dict(ENV)}                                                                    # configs2.json
    assert (ex["agents"][0]["gamma"], ex["agents"][1]["gamma"]) == (0.95, 0.995)
    Gh, E = 6, 25
    sweep = {"gamma": [[0.95] * Gh + [0.35] * Gh, [0.995] * Gh + [0.35] * Gh],
             "alpha": [[0.1] * Gh + [0.5] * Gh, [0.0] * (2 * Gh)],
             "eps": [[0.5] * Gh + [0.8] * Gh, [0.0] * (2 * Gh)]}
    one = MixedGameBatch(ex, n_games=2 * Gh, seed=3, sweep=sweep).init_tables()
    out = one.run(E)
    assert out["kernel"] == "mixed-fused" and one.nn[1].step == 2
    for cfg, off in ((ex, 0), (c2, Gh)):
        alone = MixedGameBatch(cfg, n_games=Gh, seed=3, game_offset=off).init_tables()
        oa = alone.run(E)
        sl = slice(off, off + Gh)
        assert np.array_equal(one.tables_numpy()[sl], alone.tables_numpy())
        assert np.array_equal(one.counters_numpy()[sl], alone.counters_numpy())
        assert np.array_equal(one.states_numpy()[sl], alone.states_numpy())
        for name in ("params", "adam_m", "adam_v"):
            assert torch.equal(getattr(one.nn[1], name)[sl], getattr(alone.nn[1], name)), name
        assert np.array_equal(out["game_reward_log"][:, :, sl], oa["game_reward_log"])
        assert np.array_equal(one.sweep["eps"][0, sl].cpu().numpy(), np.full(Gh, alone.eps[0]))
    # the same sweep through train_one's JSON key
    cfg = dict(ex, training={"epochs": 12, "print_freq": 500, "n_games": 2 * Gh, "seed": 3, "sweep": sweep})
    (tmp_path / "sweep.json").write_text(json.dumps(cfg))
    trainer.train_one(str(tmp_path / "run"), str(tmp_path / "sweep.json"))
    b = torch.load(tmp_path / "run" / "batch.pt", weights_only=True)
    ref = MixedGameBatch(ex, n_games=2 * Gh, seed=3, sweep=sweep).init_tables(); ref.run(12)
    assert torch.equal(b["q"], ref.q.cpu()) and torch.equal(b["nn"][1]["params"], ref.nn[1].params.cpu())
    assert torch.equal(b["sweep"]["gamma"], ref.sweep["gamma"].cpu())


@pytest.mark.parametrize("pair", ["rr", "qr", "qa", "qc"])
def test_full_size_properties_65536_games_neural(pair):
    """BASELINE config #4 at its full size (65,536 games) for 2 x Reinforce and for QTable vs Reinforce /
    ActorCritic / CAC, through the fused episode kernel and the batched update kernels: properties that
    need no oracle.  Every agent appends exactly E*T transitions and trains every min_memory of them; all
    parameters / Adam moments / tables stay finite and move; visit counters sum to E*T; rewards stay in
    the payoff band of the action grid (price in [2, 6], quantity in [2, 4] -> reward in [4, 24])."""
    import torch
    from th_rl_amd.mixed import MixedGameBatch
    G, E, T = 65536, 20, 100
    nn_ag = {"r": dict(R_AGENT), "a": dict(R_AGENT, name="ActorCritic", gamma=0.98),
             "c": {"name": "CAC", "gamma": 0.98, "states": 1, "action_range": [0.2, 0.4]}}
    first = dict(R_AGENT) if pair[0] == "r" else dict(Q_AGENT)
    config = {"agents": [first, nn_ag[pair[1]]], "environment": dict(ENV)}
    mb = MixedGameBatch(config, n_games=G, dtype="float32", seed=5).init_tables()
    w0 = {i: rb.params.clone() for i, rb in mb.nn.items()}
    out = mb.run(E, per_game_logs=False)
    assert out["kernel"] == "mixed-fused"
    for i, rb in mb.nn.items():
        assert rb.step == E * T // 1000                       # one update per min_memory = 1,000 transitions
        for name in ("params", "adam_m", "adam_v"):
            assert bool(torch.isfinite(getattr(rb, name)).all()), (i, name)
        moved = (rb.params != w0[i]).any(dim=1)
        assert bool(moved.all()), "some game's network never trained"
        assert mb.count[i] == 0                               # buffer emptied by the last update
    if pair[0] == "q":
        c = mb.counter[:, :101 * 21].sum(dim=1)
        assert bool((c == E * T).all())                       # QTable.counter: one visit per transition (agents.py:76)
        assert bool(torch.isfinite(mb.q).all())
    r = out["reward_log"]
    assert r.shape == (E, 2) and np.isfinite(r).all() and (r > 4.0).all() and (r < 24.0).all()
    a = out["action_log"]
    assert ((a >= 0.2) & (a <= 0.4)).all()


# ---------------------------------------------------------------- the tuple-chain kernel for games with neural policies
@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("label,agents,T", [
    ("q_vs_reinforce", [dict(Q_AGENT, min_memory=30, capacity=64), dict(R_AGENT, min_memory=70, entropy=0.01)], 30),
    ("reinforce_vs_q", [dict(R_AGENT, min_memory=50), dict(Q_AGENT, min_memory=25, capacity=500, actions=15)], 25),
    ("two_reinforce", [dict(R_AGENT, min_memory=60, entropy=0.02), dict(R_AGENT, min_memory=35)], 20),
    ("reinforce_ring_wraps", [dict(Q_AGENT, min_memory=25, capacity=40), dict(R_AGENT, min_memory=60, capacity=70)], 25),
    ("two_reinforce_rings_wrap_T64", [dict(R_AGENT, min_memory=150, capacity=170), dict(R_AGENT, min_memory=100, capacity=128)], 64),
    ("q_vs_actorcritic_T70", [dict(Q_AGENT, min_memory=70), {"name": "ActorCritic", "gamma": 0.98, "actions": 21, "states": 1,
                                                             "action_range": [0.2, 0.4], "min_memory": 140}], 70),
    ("reinforce_vs_actorcritic", [dict(R_AGENT, min_memory=60), {"name": "ActorCritic", "gamma": 0.98, "actions": 21, "states": 1,
                                                                 "action_range": [0.2, 0.4], "min_memory": 45}], 15),
])
@pytest.mark.parametrize("noise", [0.0, 0.05, 0.6])
def test_policy_tuple_kernel_equals_general_kernel_and_operator_loop(label, agents, T, dtype, noise):
    """thrl_ptuple.hip (two-agent games with discrete policies: the state is carried as the action pair, policy CDFs are
    looked up; after a step with a redrawn intercept -- env noise -- the state is off the grid and the policies are evaluated
    on its price directly) against the general fused kernel (k_mixed_wave) AND the unfused operator loop, on the same seeds
    over several network updates and two calls: per-game logs, tables, counters, state, replay rings, epsilon, network
    parameters and Adam state all bit-identical."""
    from th_rl_amd.mixed import MixedGameBatch
    if noise and dtype == "float64" and label not in ("q_vs_reinforce", "two_reinforce"):
        pytest.skip("noise x float64: two pairings are enough")
    config = {"agents": [dict(x) for x in agents], "environment": dict(ENV, max_steps=T, noise_prob=noise)}
    G, E = 7, 9
    a = MixedGameBatch(config, n_games=G, dtype=dtype, seed=33, game_offset=11).init_tables()
    b = MixedGameBatch(config, n_games=G, dtype=dtype, seed=33, game_offset=11).init_tables()
    c = MixedGameBatch(config, n_games=G, dtype=dtype, seed=33, game_offset=11).init_tables()
    b.tuple_kernel = False
    ra1 = a.run(4, fused=True); ra2 = a.run(E - 4, fused=True)
    rb = b.run(E, fused=True)
    rc = c.run(E, fused=False)
    assert ra1["episode_kernel"] == "tuple" and ra2["episode_kernel"] == "tuple" and rb["episode_kernel"] == "wave", label
    for other, ro in ((b, rb), (c, rc)):
        assert np.array_equal(np.concatenate([ra1["game_reward_log"], ra2["game_reward_log"]]), ro["game_reward_log"]), label
        assert np.array_equal(np.concatenate([ra1["game_action_log"], ra2["game_action_log"]]), ro["game_action_log"]), label
        assert np.array_equal(a.tables_numpy(), other.tables_numpy()) and np.array_equal(a.counters_numpy(), other.counters_numpy())
        assert np.array_equal(a.states_numpy(), other.states_numpy())
        assert a.eps == other.eps and a.count == other.count and a.episode == other.episode
        for i in a.nn:
            assert a.nn[i].step == other.nn[i].step and a.nn[i].step >= 1
            assert np.array_equal(a.nn[i].params.cpu().numpy(), other.nn[i].params.cpu().numpy()), label
            assert np.array_equal(a.nn[i].adam_m.cpu().numpy(), other.nn[i].adam_m.cpu().numpy())
            # the rings hold the same transitions where both are filled (the QTable agent's ring is not used by the tuple kernel)
            n = min(a.count[i], a.buf_len[i])
            for k in ("price", "action", "reward", "nprice"):
                assert np.array_equal(a.buf[i][k][:, :n].cpu().numpy(), other.buf[i][k][:, :n].cpu().numpy()), (label, k)


@pytest.mark.parametrize("noise", [0.0, 0.2])
@pytest.mark.parametrize("label,agents,T", [
    ("q_vs_reinforce", [dict(Q_AGENT, min_memory=30, capacity=64), dict(R_AGENT, min_memory=70, entropy=0.01)], 30),
    ("reinforce_vs_q", [dict(R_AGENT, min_memory=50), dict(Q_AGENT, min_memory=25, capacity=500, actions=15)], 25),
    ("q_vs_actorcritic_T70", [dict(Q_AGENT, min_memory=70), {"name": "ActorCritic", "gamma": 0.98, "actions": 21, "states": 1,
                                                             "action_range": [0.2, 0.4], "min_memory": 140}], 70),
])
def test_policy_tuple_kernel_takes_per_game_sweeps(label, agents, T, noise):
    """Per-game sweeps (the QTable agent's gamma / alpha / epsilon schedule, the neural agent's gamma / entropy, noise_prob) on
    the tuple-chain kernel against the general fused kernel given the same arrays: logs, tables, counters, state, per-game
    epsilon, rings, network parameters bit-identical over several updates and two calls."""
    from th_rl_amd.mixed import MixedGameBatch
    config = {"agents": [dict(x) for x in agents], "environment": dict(ENV, max_steps=T, noise_prob=noise)}
    G, E = 9, 9
    rs = np.random.RandomState(77)
    sweep = dict(gamma=rs.choice([0.9, 0.95, 0.99], (2, G)), alpha=rs.choice([0.05, 0.1, 0.4], (2, G)),
                 eps=rs.uniform(0.0, 0.8, (2, G)), eps_end=rs.choice([0.0, 0.01], (2, G)), eps_step=rs.choice([0.9, 0.999], (2, G)),
                 entropy=rs.choice([0.0, 0.01], (2, G)))
    if noise:
        sweep["noise_prob"] = rs.choice([0.0, 0.1, 0.7], G)
    a = MixedGameBatch(config, n_games=G, seed=41, sweep=sweep).init_tables()
    b = MixedGameBatch(config, n_games=G, seed=41, sweep=sweep).init_tables()
    b.tuple_kernel = False
    ra1 = a.run(4, fused=True); ra2 = a.run(E - 4, fused=True)
    rb = b.run(E, fused=True)
    assert ra1["episode_kernel"] == "tuple" and ra2["episode_kernel"] == "tuple" and rb["episode_kernel"] == "wave", label
    assert np.array_equal(np.concatenate([ra1["game_reward_log"], ra2["game_reward_log"]]), rb["game_reward_log"]), label
    assert np.array_equal(a.tables_numpy(), b.tables_numpy()) and np.array_equal(a.counters_numpy(), b.counters_numpy())
    assert np.array_equal(a.states_numpy(), b.states_numpy())
    assert np.array_equal(a.sweep["eps"].cpu().numpy(), b.sweep["eps"].cpu().numpy())
    for i in a.nn:
        assert a.nn[i].step == b.nn[i].step and a.nn[i].step >= 1
        assert np.array_equal(a.nn[i].params.cpu().numpy(), b.nn[i].params.cpu().numpy()), label
        n = min(a.count[i], a.buf_len[i])
        for k in ("price", "action", "reward", "nprice"):
            assert np.array_equal(a.buf[i][k][:, :n].cpu().numpy(), b.buf[i][k][:, :n].cpu().numpy()), (label, k)


def test_policy_tuple_kernel_is_taken_with_noise_but_not_with_unfit_buffers():
    from th_rl_amd.mixed import MixedGameBatch
    base = [dict(Q_AGENT, min_memory=25), dict(R_AGENT, min_memory=50)]
    noisy = MixedGameBatch({"agents": [dict(x) for x in base], "environment": dict(ENV, max_steps=25, noise_prob=0.05)}, n_games=4, seed=1).init_tables()
    assert noisy.run(2, fused=True)["episode_kernel"] == "tuple"
    spans = MixedGameBatch({"agents": [dict(Q_AGENT, min_memory=60), dict(R_AGENT, min_memory=50)],
                            "environment": dict(ENV, max_steps=25)}, n_games=4, seed=1).init_tables()     # QTable trains every 3rd episode
    assert spans.run(2, fused=True)["episode_kernel"] == "wave"


@pytest.mark.parametrize("n,G", [(1000, 200), (333, 65), (64, 130), (2, 64)])
def test_returns_prepass_equals_in_kernel_recurrence(n, G):
    """Reinforce.train_net's discounted returns (agents.py:178-181) by the pre-pass kernel (one lane per game, tiles transposed
    through LDS) against the in-kernel form (one thread per block): same operations in the same order, so parameters, Adam
    moments and clipped gradients are bit-identical -- with the scalar gamma and with a per-game gamma sweep, over two updates."""
    from th_rl_amd.nn import ReinforceBatch
    rs = np.random.RandomState(n + G)
    price = rs.choice(np.round(np.linspace(2.0, 6.0, 41), 10), (n, G))
    action = rs.randint(0, 21, (n, G)).astype(np.int32)
    reward = rs.uniform(5.0, 12.5, (n, G))
    for sweep in (False, True):
        outs = []
        for prepass in (True, False):
            rb = ReinforceBatch(G, actions=21, gamma=0.995, entropy=0.01, seed=3).init()
            rb.returns_prepass = prepass
            if sweep:
                rb.set_sweep(gamma=rs.__class__(5).choice([0.9, 0.98, 0.995], G))
            g1 = rb.train(price, action, reward, want_grad=True).cpu().numpy().copy()
            g2 = rb.train(price[::-1].copy(), action, reward[::-1].copy(), want_grad=True).cpu().numpy().copy()
            outs.append((rb.params.cpu().numpy().copy(), rb.adam_m.cpu().numpy().copy(), rb.adam_v.cpu().numpy().copy(), g1, g2))
        for x, y in zip(outs[0], outs[1]):
            assert np.array_equal(x, y), (n, G, sweep)
        assert np.isfinite(outs[0][0]).all() and np.abs(outs[0][3]).max() > 0
