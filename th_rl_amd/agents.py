"""Agents of the reference (th_rl/agents.py) with the same constructors and protocol.

QTable is the hot-path agent.  Its object-level methods run on the GPU through the
unfused operators of libthrl_hip.so (one game per call); training many games at
once goes through GameBatch / train_one instead.  Host-side state (`table`,
`counter`, `epsilon`, `memory`) keeps the reference's attribute names so
utils.load_experiment / plot_qagent style consumers keep working.

Reinforce, ActorCritic and CAC act and train on the device too
(thrl_nn_* / thrl_ac_* / thrl_cac_*); the torch modules only hold the parameters in the reference's
state_dict format.  Their `reset` / `reset_value` / `reset_pi` methods are deliberately absent: in the
reference they touch layers that do not exist (agents.py:202,207,436) and nothing calls them
(DESIGN.md section 8).
"""
from collections import namedtuple
import random

import numpy
import torch
import torch.nn as nn
import torch.optim as optim

from . import _lib
from .buffers import *  # noqa: F401,F403  (the buffer class name is eval'd, as in the reference)


class QTable:
    def __init__(
        self,
        states=16,
        actions=4,
        action_range=[0, 1],
        gamma=0.99,
        buffer="ReplayBuffer",
        capacity=500,
        max_state=10,
        alpha=0.1,
        eps_end=2e-2,
        epsilon=0.5,
        eps_step=5e-4,
        min_memory=100,
        **kwargs
    ):
        self.table = 12.5 / (1 - gamma) + numpy.random.randn(states + 1, actions)
        self.gamma = gamma
        self.alpha = alpha
        self.action_space = numpy.arange(0, actions)
        self.action_range = action_range
        self.actions = actions
        self.epsilon = epsilon
        self.eps_step = eps_step
        self.eps_end = eps_end
        self.states = states
        self.max_state = max_state
        self.min_memory = min_memory
        self.capacity = capacity
        self.experience = namedtuple(
            "Experience", field_names=["state", "action", "reward", "done", "new_state"]
        )
        self.memory = eval(buffer)(capacity, self.experience)
        self.counter = 0 * self.table
        self._ops = None

    # -- device plumbing ---------------------------------------------------------
    def config_block(self):
        """The JSON block that reconstructs this agent (used to build thrl_cfg)."""
        return dict(name="QTable", states=self.states, actions=self.actions,
                    action_range=list(self.action_range), gamma=self.gamma, capacity=self.capacity,
                    max_state=self.max_state, alpha=self.alpha, eps_end=self.eps_end,
                    epsilon=self.epsilon, eps_step=self.eps_step, min_memory=self.min_memory)

    def _device_ops(self):
        from ._ops import DeviceOps
        cfg, _ = _lib.cfg_from_config(
            {"agents": [self.config_block()], "environment": {"nplayers": 1, "max_steps": 1}}, 1, 1)
        if self._ops is None:
            self._ops = DeviceOps(cfg)
        self._ops.cfg = cfg
        return self._ops

    # -- reference protocol ------------------------------------------------------
    def encode(self, state):
        state = numpy.asarray(state)
        rows = self._device_ops().encode(0, state.astype("float64"), state.dtype == numpy.float32)
        return rows.reshape(state.shape)

    def scale(self, actions):
        """agents.py:51-57; like the reference it takes a scalar index or an array of indices."""
        if numpy.ndim(actions) == 0:
            return self._device_ops().scale(0, actions)
        a = numpy.asarray(actions)
        return numpy.array([self._device_ops().scale(0, int(k)) for k in a.ravel()]).reshape(a.shape)

    def sample_action(self, state):
        """epsilon-greedy; the two stdlib draws are made exactly where the reference makes them."""
        if random.uniform(0, 1) < self.epsilon:
            return random.choice(self.action_space)
        st = state.numpy() if isinstance(state, torch.Tensor) else numpy.asarray(state)
        return self._greedy(st)

    def get_action(self, state):
        return self._greedy(numpy.asarray(state))

    def _greedy(self, st):
        price = float(numpy.asarray(st).reshape(-1)[0])
        return numpy.int64(self._device_ops().greedy_action(0, self.table, price, st.dtype == numpy.float32))

    def train_net(self):
        if len(self.memory) >= self.min_memory:
            price, acts, rwrd, not_done, next_state = self.memory.replay()
            price = numpy.array(price, dtype="float64").reshape(-1)
            next_state = numpy.array(next_state, dtype="float64").reshape(-1)
            acts = numpy.reshape(acts, [-1])
            rwrd = numpy.reshape(rwrd, [-1]).astype("float64")
            self.table, counter = self._device_ops().td_update(
                0, self.table, numpy.zeros_like(self.table), price, acts, rwrd, next_state)
            self.counter = self.counter + counter
            self.memory.empty()
        self.epsilon = self.eps_end + (self.epsilon - self.eps_end) * self.eps_step

    def reset(self, eps_end):
        self.table = 100 / (1 - self.gamma) + numpy.random.randn(self.states, self.actions)
        self.epsilon = 1.0
        self.eps_end = eps_end

    def reset_value(self, eps_end):
        self.table = 100 / (1 - self.gamma) + numpy.random.randn(self.states, self.actions)

    def reset_pi(self, eps_end):
        self.epsilon = 1.0
        self.eps_end = eps_end

    def save(self, loc):
        numpy.save(loc, self.table)
        numpy.save(loc + "_counter", self.counter)

    def load(self, loc):
        self.table = numpy.load(loc + ".npy")
        self.counter = numpy.load(loc + "_counter.npy")


class _NeuralAgentBase(nn.Module):
    """Common part of the reference's torch agents: parameters, memory, scale, save/load.  The torch
    modules only HOLD the parameters (state_dict / save / load keep the reference's format); acting
    and learning run on the device in the subclasses."""

    _scope_note = "this agent class does not act or learn on its own; use Reinforce, ActorCritic or CAC"

    def _common(self, actions, action_range, gamma, buffer, capacity, min_memory, entropy):
        self.gamma = gamma
        self.action_range = action_range
        self.actions = actions
        self.experience = namedtuple(
            "Experience", field_names=["state", "action", "reward", "done", "new_state"]
        )
        self.cast = [torch.float, torch.int64, torch.float, torch.float, torch.float]
        self.memory = eval(buffer)(capacity, self.experience)
        self.min_memory = min_memory
        self.entropy = entropy

    def scale(self, action):
        return action / self.actions * (self.action_range[1] - self.action_range[0]) + self.action_range[0]

    def sample_action(self, state):
        raise NotImplementedError(self._scope_note)

    def get_action(self, state):
        raise NotImplementedError(self._scope_note)

    def train_net(self):
        raise NotImplementedError(self._scope_note)

    def save(self, loc):
        torch.save(self.state_dict(), loc)

    def load(self, loc):
        self.load_state_dict(torch.load(loc, weights_only=True))


class Reinforce(_NeuralAgentBase):
    """The reference's REINFORCE agent (agents.py:119-220).  The torch modules hold the parameters
    (so state_dict / save / load keep the reference's format); acting and train_net run on the GPU
    through thrl_nn_act / thrl_nn_reinforce_train (one game per call here; many games at once
    through th_rl_amd.nn.ReinforceBatch / mixed.MixedGameBatch)."""

    def __init__(self, states=4, actions=2, action_range=[0, 1], gamma=0.98, buffer="ReplayBuffer",
                 capacity=50000, min_memory=1000, entropy=0, **kwargs):
        super().__init__()
        self.data = []
        self.fc1 = nn.Linear(states, 256)
        self.fc_pi = nn.Linear(256, actions)
        self.optimizer = optim.Adam(self.parameters(), lr=2e-4)
        self._common(actions, action_range, gamma, buffer, capacity, min_memory, entropy)
        self.states = states
        self._rb = None

    # -- device plumbing ---------------------------------------------------------
    def flat_params(self):
        return numpy.concatenate([self.fc1.weight.detach().numpy().ravel(), self.fc1.bias.detach().numpy().ravel(),
                                  self.fc_pi.weight.detach().numpy().ravel(),
                                  self.fc_pi.bias.detach().numpy().ravel()]).astype("float32")

    def set_flat_params(self, w):
        w = numpy.asarray(w, "float32")
        A = self.actions
        with torch.no_grad():
            self.fc1.weight.copy_(torch.from_numpy(w[:256].reshape(256, 1).copy()))
            self.fc1.bias.copy_(torch.from_numpy(w[256:512].copy()))
            self.fc_pi.weight.copy_(torch.from_numpy(w[512:512 + A * 256].reshape(A, 256).copy()))
            self.fc_pi.bias.copy_(torch.from_numpy(w[512 + A * 256:].copy()))

    def _device(self):
        if self.states != 1:
            raise _lib.ThrlError("Reinforce on the device needs states == 1")
        if self._rb is None:
            from .nn import ReinforceBatch
            self._rb = ReinforceBatch(1, actions=self.actions, gamma=self.gamma, entropy=self.entropy)
        self._rb.gamma, self._rb.entropy = float(self.gamma), float(self.entropy)
        self._rb.set_params(self.flat_params())
        return self._rb

    # -- reference protocol ------------------------------------------------------
    def pi(self, x, softmax_dim=0):
        """Action probabilities for one state tensor (1,) -- evaluated on the device."""
        rb = self._device()
        _, probs = rb.act(numpy.asarray(x, dtype="float64").reshape(1), want_probs=True)
        return probs[0].cpu()

    def sample_action(self, state):
        u = float(torch.rand(()))                    # the categorical draw (inverse CDF on the device)
        return int(self._device().act(numpy.asarray(state, dtype="float64").reshape(1), u=[u]).cpu()[0])

    def get_action(self, state):
        return int(self._device().act(numpy.asarray(state, dtype="float64").reshape(1)).cpu()[0])

    def train_net(self):
        if len(self.memory) >= self.min_memory:
            states, actions, rewards, done, s_prime = self.memory.replay()
            n = len(actions)
            rb = self._device()
            rb.train(numpy.array(states, dtype="float64").reshape(n, 1), numpy.array(actions).reshape(n, 1),
                     numpy.array(rewards, dtype="float64").reshape(n, 1))
            self.set_flat_params(rb.params.cpu().numpy()[0])
            self.memory.empty()


class ActorCritic(Reinforce):
    """The reference's actor-critic agent (agents.py:222-330): Reinforce's policy head plus the value
    head fc_v (bias initialised to 1000, :243-244).  Acting is Reinforce's; train_net runs
    thrl_ac_train, which reproduces the reference's [N,N]-broadcast advantage (:290)."""

    def __init__(self, states=4, actions=2, action_range=[0, 1], gamma=0.98, buffer="ReplayBuffer",
                 capacity=50000, min_memory=1000, entropy=0, **kwargs):
        _NeuralAgentBase.__init__(self)
        self.data = []
        self.fc1 = nn.Linear(states, 256)
        self.fc_pi = nn.Linear(256, actions)
        self.fc_v = nn.Linear(256, 1)
        self.fc_v.bias.data.fill_(1000.0)
        self.optimizer = optim.Adam(self.parameters(), lr=2e-4)
        self._common(actions, action_range, gamma, buffer, capacity, min_memory, entropy)
        self.states = states
        self._rb = None

    def flat_params(self):
        return numpy.concatenate([Reinforce.flat_params(self), self.fc_v.weight.detach().numpy().ravel(),
                                  self.fc_v.bias.detach().numpy().ravel()]).astype("float32")

    def set_flat_params(self, w):
        w = numpy.asarray(w, "float32")
        P = 512 + self.actions * 256 + self.actions
        Reinforce.set_flat_params(self, w[:P])
        with torch.no_grad():
            self.fc_v.weight.copy_(torch.from_numpy(w[P:P + 256].reshape(1, 256).copy()))
            self.fc_v.bias.copy_(torch.from_numpy(w[P + 256:P + 257].copy()))

    def _device(self):
        if self.states != 1:
            raise _lib.ThrlError("ActorCritic on the device needs states == 1")
        if self._rb is None:
            from .nn import ActorCriticBatch
            self._rb = ActorCriticBatch(1, actions=self.actions, gamma=self.gamma, entropy=self.entropy)
        self._rb.gamma, self._rb.entropy = float(self.gamma), float(self.entropy)
        self._rb.set_params(self.flat_params())
        return self._rb

    def v(self, x):
        """Value head for one state tensor (1,): host float32 arithmetic (not on the training path)."""
        y = torch.relu(self.fc1(torch.as_tensor(x, dtype=torch.float32).reshape(-1)))
        return self.fc_v(y).detach()

    def train_net(self):
        if len(self.memory) >= self.min_memory:
            states, actions, rewards, done, s_prime = self.memory.replay()
            n = len(actions)
            rb = self._device()
            rb.train(numpy.array(states, dtype="float64").reshape(n, 1), numpy.array(actions).reshape(n, 1),
                     numpy.array(rewards, dtype="float64").reshape(n, 1),
                     next_price=numpy.array(s_prime, dtype="float64").reshape(n, 1))
            self.set_flat_params(rb.params.cpu().numpy()[0])
            self.memory.empty()


class CAC(_NeuralAgentBase):
    """The reference's continuous actor-critic (agents.py:333-442): actions are floats in (0,1).
    Acting and train_net run on the device (thrl_cac_act / thrl_cac_train).  get_action returns the
    mean action sigmoid(mu): the reference's own get_action builds Normal(mu, 0) and raises
    ValueError under current torch (recorded in tests/golden/g9_cac.npz)."""

    def __init__(self, states=4, action_range=[0, 1], gamma=0.98, buffer="ReplayBuffer", capacity=50000,
                 min_memory=1000, entropy=0, **kwargs):
        super().__init__()
        self.data = []
        self.fc1 = nn.Linear(states, 256)
        self.fc_mu = nn.Linear(256, 1)
        self.fc_std = nn.Linear(256, 1)
        self.fc_v = nn.Linear(256, 1)
        self.optimizer = optim.Adam(self.parameters(), lr=2e-4)
        self._common(None, action_range, gamma, buffer, capacity, min_memory, entropy)
        del self.actions                                       # the reference's CAC has no such attribute
        self.cast = [torch.float, torch.float, torch.float, torch.float, torch.float]
        self.states = states
        self._rb = None

    _layers = ("fc1", "fc_mu", "fc_std", "fc_v")

    def flat_params(self):
        parts = []
        for name in self._layers:
            layer = getattr(self, name)
            parts += [layer.weight.detach().numpy().ravel(), layer.bias.detach().numpy().ravel()]
        return numpy.concatenate(parts).astype("float32")

    def set_flat_params(self, w):
        w = numpy.asarray(w, "float32")
        o = 0
        with torch.no_grad():
            for name in self._layers:
                layer = getattr(self, name)
                for t in (layer.weight, layer.bias):
                    n = t.numel()
                    t.copy_(torch.from_numpy(w[o:o + n].reshape(tuple(t.shape)).copy()))
                    o += n

    def _device(self):
        if self.states != 1:
            raise _lib.ThrlError("CAC on the device needs states == 1")
        if self._rb is None:
            from .nn import CACBatch
            self._rb = CACBatch(1, gamma=self.gamma, entropy=self.entropy)
        self._rb.gamma, self._rb.entropy = float(self.gamma), float(self.entropy)
        self._rb.set_params(self.flat_params())
        return self._rb

    def scale(self, action):
        return action * (self.action_range[1] - self.action_range[0]) + self.action_range[0]

    def pi(self, x):
        _, (mu, std, _) = self._device().act(numpy.asarray(x, dtype="float64").reshape(1), want_heads=True)
        return mu.cpu(), std.cpu()

    def v(self, x):
        _, (_, _, v) = self._device().act(numpy.asarray(x, dtype="float64").reshape(1), want_heads=True)
        return v.cpu()

    def sample_action(self, state):
        u1, u2 = float(torch.rand(())), float(torch.rand(()))  # Box-Muller inputs for Normal(mu, std).sample()
        return float(self._device().act(numpy.asarray(state, dtype="float64").reshape(1), u1=[u1], u2=[u2]).cpu()[0])

    def get_action(self, state):
        return float(self._device().act(numpy.asarray(state, dtype="float64").reshape(1)).cpu()[0])

    def train_net(self):
        if len(self.memory) >= self.min_memory:
            states, actions, rewards, done, s_prime = self.memory.replay()
            n = len(actions)
            rb = self._device()
            rb.train(numpy.array(states, dtype="float64").reshape(n, 1), numpy.array(actions, dtype="float32").reshape(n, 1),
                     numpy.array(rewards, dtype="float64").reshape(n, 1),
                     next_price=numpy.array(s_prime, dtype="float64").reshape(n, 1))
            self.set_flat_params(rb.params.cpu().numpy()[0])
            self.memory.empty()
