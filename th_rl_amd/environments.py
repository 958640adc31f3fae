"""NoisyPriceState -- the reference's environment class (th_rl/environments.py:4-53),
same constructor and protocol, with the payoff executed on the GPU.

Object-level calls (`env.step(...)`) run the unfused device operator
thrl_op_env_step for one game; the fused multi-game path is GameBatch.run.
The random draws stay where the reference makes them (numpy's global RNG on the
host), so a seeded run consumes the same stream as the reference.
"""
import numpy

from . import _lib


class NoisyPriceState:
    def __init__(self, nplayers, action_range=[0, 1], a=10, b=1, max_steps=1, noise_prob=0.05, **kwargs):
        self.nplayers = nplayers
        self.action_range = action_range
        self.b = b
        self.a = a
        self.max_steps = max_steps
        self.state = self.sample_state()
        self.episode = 0
        self.noise_prob = noise_prob
        self._ops = None

    # -- device plumbing ---------------------------------------------------------
    def _device_ops(self):
        if self._ops is None:
            from ._ops import DeviceOps
            cfg = _lib.Cfg()
            cfg.n_games, cfg.n_agents, cfg.max_steps, cfg.q_dtype = 1, int(self.nplayers), int(self.max_steps), 1
            cfg.env_a, cfg.env_b, cfg.noise_prob = float(self.a), float(self.b), float(self.noise_prob)
            for i in range(int(self.nplayers)):     # agent fields are unused by env_step; keep them valid
                cfg.n_states[i], cfg.n_actions[i], cfg.max_state[i] = 1, 2, 1.0
            self._ops = DeviceOps(cfg)
        self._ops.cfg.env_a, self._ops.cfg.env_b = float(self.a), float(self.b)
        self._ops.cfg.noise_prob = float(self.noise_prob)
        return self._ops

    # -- reference protocol ------------------------------------------------------
    def sample_state(self):
        return numpy.random.uniform(0, self.a)

    def encode(self):
        return numpy.atleast_1d(self.state)

    def step(self, actions):
        """actions: list of N scaled quantities.  -> (state ndarray(1,), rewards ndarray(N,), done)."""
        noise_u = numpy.random.uniform(0, 1)                 # always drawn (environments.py:28)
        noise_a = numpy.random.uniform(self.a * 0.7, self.a) if noise_u < self.noise_prob else 0.0
        price, rewards = self._device_ops().env_step([float(x) for x in actions], noise_u, noise_a)
        self.state = numpy.float64(price)
        self.episode += 1
        done = self.episode >= self.max_steps
        return self.encode(), numpy.array(rewards), done

    def get_optimal(self):
        """(Nash total, cartel total) rewards of the one-shot game; closed form, not on the hot path."""
        n = self.nplayers
        per_nash = (self.a / self.b) * numpy.ones(n,) / (n + 1)
        p_nash = numpy.max([0, self.a - self.b * sum(per_nash)])
        per_cartel = (self.a / self.b) * 0.5 * numpy.ones(n,) / n
        p_cartel = numpy.max([0, self.a - self.b * sum(per_cartel)])
        return sum([p_nash * q for q in per_nash]), sum([p_cartel * q for q in per_cartel])

    def reset(self):
        self.episode = 0
        self.state = self.sample_state()
        return self.encode()


# BASELINE.json's north_star calls the environment "PricingGame"; the reference's only
# environment class is NoisyPriceState.
PricingGame = NoisyPriceState
