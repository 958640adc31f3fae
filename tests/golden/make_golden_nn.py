#!/usr/bin/env python3
"""Golden fixtures G7 / G8 for the neural policy path: runs the REFERENCE's Reinforce
(th_rl/agents.py:119-220) and ActorCritic (:222-330) agents here, records data only.
Usage: python tests/golden/make_golden_nn.py

Recorded: seeded initial parameters (torch default Linear init), action probabilities on
probe states (pi(), agents.py:147-151), two consecutive train_net() calls (agents.py:170-194)
on 1,000 appended transitions each: clipped gradients left in .grad, Adam state
(exp_avg, exp_avg_sq, step) and the parameters afterwards; get_action on probe states.
"""
import os
import random
import sys

import numpy

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import torch  # noqa: E402
import th_rl.agents as ref_agents  # noqa: E402

PARAMS = ["fc1.weight", "fc1.bias", "fc_pi.weight", "fc_pi.bias"]


def flat(d):
    return numpy.concatenate([d[k].detach().numpy().ravel() for k in PARAMS]).astype("float32")


def main():
    global PARAMS
    torch.set_num_threads(1)
    for cls, fname, cases in (
            ("Reinforce", "g7_reinforce.npz",
             (("cfg", dict(gamma=0.995, actions=21, states=1, action_range=[0.2, 0.4])),
              ("ent", dict(gamma=0.35, actions=21, states=1, action_range=[0.2, 0.4], entropy=0.01)))),
            ("ActorCritic", "g8_actorcritic.npz",
             (("cfg", dict(gamma=0.98, actions=21, states=1, action_range=[0.2, 0.4])),
              ("ent", dict(gamma=0.9, actions=15, states=1, action_range=[0.2, 0.4], entropy=0.01))))):
        PARAMS = ["fc1.weight", "fc1.bias", "fc_pi.weight", "fc_pi.bias"] + \
                 (["fc_v.weight", "fc_v.bias"] if cls == "ActorCritic" else [])
        one(cls, fname, cases)
    cac()


def cac():
    """G9: the reference's CAC agent (agents.py:333-442).  pi()/v() on probe states, two train_net()
    calls on 1,000 transitions each (actions in (0,1) as sample_action returns them)."""
    names = ["fc1.weight", "fc1.bias", "fc_mu.weight", "fc_mu.bias", "fc_std.weight", "fc_std.bias",
             "fc_v.weight", "fc_v.bias"]
    fl = lambda d: numpy.concatenate([d[k].detach().numpy().ravel() for k in names]).astype("float32")
    out = {}
    for tag, kw in (("cfg", dict(gamma=0.98, states=1, action_range=[0.2, 0.4])),
                    ("ent", dict(gamma=0.9, states=1, action_range=[0.2, 0.4], entropy=0.01))):
        numpy.random.seed(5); random.seed(5); torch.manual_seed(5)
        ag = ref_agents.CAC(**kw)
        out[tag + "_w0"] = fl(ag.state_dict())
        probe = numpy.linspace(2.0, 6.0, 9)
        out[tag + "_probe_price"] = probe
        rs = numpy.random.RandomState(17)
        for call in range(2):
            price = rs.randint(20, 61, size=1001) / 10.0
            action = rs.uniform(0.02, 0.98, size=1000).astype("float32")
            reward = rs.uniform(5, 15, size=1000)
            for t in range(1000):
                ag.memory.append(numpy.array([price[t]]), float(action[t]), float(reward[t]), True,
                                 numpy.array([price[t + 1]]))
            ag.train_net()
            named = dict(ag.named_parameters())
            out["%s_c%d_price" % (tag, call)] = price
            out["%s_c%d_action" % (tag, call)] = action
            out["%s_c%d_reward" % (tag, call)] = reward
            out["%s_c%d_grad" % (tag, call)] = numpy.concatenate([named[k].grad.numpy().ravel() for k in names]).astype("float32")
            st = ag.optimizer.state
            out["%s_c%d_m" % (tag, call)] = numpy.concatenate([st[named[k]]["exp_avg"].numpy().ravel() for k in names]).astype("float32")
            out["%s_c%d_v" % (tag, call)] = numpy.concatenate([st[named[k]]["exp_avg_sq"].numpy().ravel() for k in names]).astype("float32")
            out["%s_c%d_w" % (tag, call)] = fl(ag.state_dict())
        with torch.no_grad():
            pv = [ag.pi(torch.from_numpy(numpy.array([q]).astype("float32"))) for q in probe]
            out[tag + "_probe_mu2"] = numpy.array([float(a[0]) for a in pv], "float32")
            out[tag + "_probe_std2"] = numpy.array([float(a[1]) for a in pv], "float32")
            out[tag + "_probe_value2"] = numpy.array(
                [float(ag.v(torch.from_numpy(numpy.array([q]).astype("float32")))) for q in probe], "float32")
        out[tag + "_scale"] = numpy.array([ag.scale(k) for k in (0.0, 0.25, 0.7311, 1.0)])
        try:
            ag.get_action(numpy.array([3.4]))
            out[tag + "_get_action_raises"] = numpy.array(0)
        except Exception as e:                       # noqa: BLE001  (recorded as data)
            out[tag + "_get_action_raises"] = numpy.array(1)
            out[tag + "_get_action_error"] = numpy.array(type(e).__name__)
    p = os.path.join(HERE, "g9_cac.npz")
    numpy.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p))


def one(cls, fname, cases):
    out = {}
    for tag, kw in cases:
        numpy.random.seed(3); random.seed(3); torch.manual_seed(3)
        ag = getattr(ref_agents, cls)(**kw)
        nA = kw["actions"]
        sd0 = {k: v.clone() for k, v in ag.state_dict().items()}
        out[tag + "_w0"] = flat(sd0)
        probe = numpy.linspace(2.0, 6.0, 9)
        out[tag + "_probe_price"] = probe
        with torch.no_grad():
            out[tag + "_probe_prob0"] = numpy.stack(
                [ag.pi(torch.from_numpy(numpy.array([p]).astype("float32"))).numpy() for p in probe])
        rs = numpy.random.RandomState(11)
        for call in range(2):
            price = rs.randint(20, 61, size=1001) / 10.0          # on the env's price grid
            action = rs.randint(0, nA, size=1000)
            reward = rs.uniform(5, 15, size=1000)
            for t in range(1000):
                ag.memory.append(numpy.array([price[t]]), numpy.int64(action[t]), float(reward[t]), True,
                                 numpy.array([price[t + 1]]))
            ag.train_net()
            assert len(ag.memory) == 0
            named = dict(ag.named_parameters())
            out["%s_c%d_price" % (tag, call)] = price
            out["%s_c%d_action" % (tag, call)] = action.astype("int64")
            out["%s_c%d_reward" % (tag, call)] = reward
            out["%s_c%d_grad" % (tag, call)] = numpy.concatenate(
                [named[k].grad.numpy().ravel() for k in PARAMS]).astype("float32")
            st = ag.optimizer.state
            out["%s_c%d_m" % (tag, call)] = numpy.concatenate(
                [st[named[k]]["exp_avg"].numpy().ravel() for k in PARAMS]).astype("float32")
            out["%s_c%d_v" % (tag, call)] = numpy.concatenate(
                [st[named[k]]["exp_avg_sq"].numpy().ravel() for k in PARAMS]).astype("float32")
            out["%s_c%d_step" % (tag, call)] = numpy.float64(float(st[named[PARAMS[0]]]["step"]))
            out["%s_c%d_w" % (tag, call)] = flat(ag.state_dict())
        with torch.no_grad():
            out[tag + "_probe_prob2"] = numpy.stack(
                [ag.pi(torch.from_numpy(numpy.array([p]).astype("float32"))).numpy() for p in probe])
        out[tag + "_probe_greedy2"] = numpy.array([ag.get_action(numpy.array([p])) for p in probe], "int64")
        out[tag + "_scale"] = numpy.array([ag.scale(k) for k in range(nA)])
        out[tag + "_kw"] = numpy.array(repr(kw))
        if cls == "ActorCritic":
            with torch.no_grad():
                out[tag + "_probe_value2"] = numpy.stack(
                    [ag.v(torch.from_numpy(numpy.array([q]).astype("float32"))).numpy() for q in probe])
    p = os.path.join(HERE, fname)
    numpy.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p))


if __name__ == "__main__":
    main()
