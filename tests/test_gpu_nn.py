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


def test_nn_errors():
    from th_rl_amd._lib import ThrlError
    from th_rl_amd.nn import ReinforceBatch
    with pytest.raises(ThrlError, match="actions"):
        ReinforceBatch(4, actions=40)
    rb = _rb(2).init()
    with pytest.raises(ThrlError, match="transitions"):
        rb.train(np.zeros((2000, 2)), np.zeros((2000, 2), int), np.zeros((2000, 2)))
