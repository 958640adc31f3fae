"""ReinforceBatch: the reference's `Reinforce` agent (th_rl/agents.py:119-220) for G games at once.
Per-game independent 1 -> 256 -> A MLPs, their Adam state and step counter live in HBM; acting and
training run in libthrl_hip.so (thrl_nn_*).  float32, as torch."""
import ctypes

import numpy as np

from . import _lib
from ._lib import ThrlError
from .batched import _require_gpu, _torch


class ReinforceBatch:
    value_head = False          # ActorCriticBatch: fc_v appended to the parameter vector
    _fn = dict(count="thrl_nn_param_count", init="thrl_nn_init", act="thrl_nn_act")

    def __init__(self, n_games, actions=2, gamma=0.98, entropy=0.0, lr=2e-4, device="cuda:0", seed=0,
                 game_offset=0, agent_index=0):
        self.L = _lib.load()
        torch = _torch()
        self.device = _require_gpu(device)
        self.G, self.A = int(n_games), int(actions)
        self.gamma, self.entropy, self.lr = float(gamma), float(entropy), float(lr)
        self.seed, self.game_offset, self.agent_index = int(seed), int(game_offset), int(agent_index)
        self.P = int(getattr(self.L, self._fn["count"])(self.A))
        if self.P == 0:
            raise ThrlError("%s on the device needs 2 <= actions <= 32, got %d" % (type(self).__name__, self.A))
        with torch.cuda.device(self.device):
            self.params = torch.zeros((self.G, self.P), dtype=torch.float32, device=self.device)
            self.adam_m = torch.zeros_like(self.params)
            self.adam_v = torch.zeros_like(self.params)
        self.step = 0
        self.gamma_g = self.entropy_g = None      # per-game sweeps (device float64 [G]) or None
        self.returns_prepass = True               # False: the update kernel computes the returns itself (same results)

    def set_sweep(self, gamma=None, entropy=None):
        """Per-game gamma / entropy coefficient (arrays of length G): a config sweep as one batch."""
        torch = _torch()
        def dev(x):
            if not isinstance(x, torch.Tensor):
                x = np.asarray(x, np.float64)
            return self._dev(x, torch.float64).reshape(self.G).contiguous().clone()
        if gamma is not None:
            self.gamma_g = dev(gamma)
        if entropy is not None:
            self.entropy_g = dev(entropy)
        return self

    def _stream(self):
        return ctypes.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def _dev(self, a, dtype):
        torch = _torch()
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype)

    def init(self):
        torch = _torch()
        with torch.cuda.device(self.device):
            _lib.check(getattr(self.L, self._fn["init"])(self.G, self.A, self._p(self.params), self.seed,
                                                         self.game_offset, self.agent_index, self._stream()),
                       self._fn["init"])
        self.adam_m.zero_(); self.adam_v.zero_(); self.step = 0
        return self

    def set_params(self, w):
        w = np.asarray(w, np.float32).reshape(-1, self.P)
        self.params.copy_(self._dev(np.broadcast_to(w, (self.G, self.P)).copy(), _torch().float32))
        return self

    def act(self, price, u=None, want_probs=False):
        """actions int32 [G] (device tensor); u=None -> greedy get_action."""
        torch = _torch()
        with torch.cuda.device(self.device):
            d_price = self._dev(price, torch.float64).reshape(self.G)
            d_u = None if u is None else self._dev(u, torch.float64).reshape(self.G)
            out = torch.zeros((self.G,), dtype=torch.int32, device=self.device)
            probs = torch.zeros((self.G, self.A), dtype=torch.float32, device=self.device) if want_probs else None
            _lib.check(getattr(self.L, self._fn["act"])(self.G, self.A, self._p(self.params), self._p(d_price),
                                                        self._p(d_u), self._p(out), self._p(probs), self._stream()),
                       self._fn["act"])
        return (out, probs) if want_probs else out

    def _rows(self, a, dtype, rows):
        """The library's layout of a replayed batch: [G, ld] game-major, the first n entries of a row valid.
        rows=False: `a` is [n, G] (transition-major, the reference's replay() order) and is transposed;
        rows=True: `a` is a device tensor [G, n] already -- possibly a view of a replay ring [G, buf_len]
        (unit stride along the row), passed as it is with ld = its row pitch."""
        torch = _torch()
        if rows:
            if not (isinstance(a, torch.Tensor) and a.dim() == 2 and a.shape[0] == self.G and a.dtype == dtype
                    and a.device == self.device and (a.shape[1] <= 1 or a.stride(1) == 1)):
                raise ThrlError("rows=True needs a %s device tensor [G, n] with unit stride along the row" % dtype)
            return a, int(a.shape[1]), int(a.stride(0)) if self.G > 1 else int(a.shape[1])
        t = self._dev(a, dtype)
        n = t.shape[0]
        t = t.reshape(n, self.G).t().contiguous()
        return t, int(n), int(n)

    def train(self, price, action, reward, want_grad=False, next_price=None, rows=False):
        """One train_net update on n transitions per game: arrays [n, G] (rows=True: device tensors [G, n], see
        _rows -- the fused loop hands its replay rings over without a copy)."""
        torch = _torch()
        with torch.cuda.device(self.device):
            d_p, n, ld = self._rows(price, torch.float64, rows)
            d_a, _, lda = self._rows(action, torch.int32, rows)
            d_r, _, ldr = self._rows(reward, torch.float64, rows)
            if not (ld == lda == ldr):
                raise ThrlError("price / action / reward rows must share one pitch")
            grad = torch.zeros_like(self.params) if want_grad else None
            # scratch for the returns pre-pass (one lane per game instead of one thread per block: same bits), kept
            scr = None
            if self.returns_prepass and self.G >= 64:
                need = self.G * ld
                if getattr(self, "_ret_scratch", None) is None or self._ret_scratch.numel() < need:
                    self._ret_scratch = torch.empty((need,), dtype=torch.float32, device=self.device)
                scr = self._ret_scratch
            _lib.check(self.L.thrl_nn_reinforce_train(self.G, self.A, self._p(self.params), self._p(self.adam_m),
                                                      self._p(self.adam_v), self.step, n, ld, self._p(d_p), self._p(d_a),
                                                      self._p(d_r), self.gamma, self.entropy, self.lr,
                                                      self._p(self.gamma_g), self._p(self.entropy_g),
                                                      self._p(grad), self._p(scr), self._stream()), "thrl_nn_reinforce_train")
            torch.cuda.synchronize(self.device)
        self.step += 1
        return grad


class ActorCriticBatch(ReinforceBatch):
    """The reference's `ActorCritic` agent (th_rl/agents.py:222-330) for G games: Reinforce's policy
    network plus the value head fc_v; train() = thrl_ac_train (needs the replayed next states)."""
    value_head = True
    _fn = dict(count="thrl_ac_param_count", init="thrl_ac_init", act="thrl_ac_act")

    def train(self, price, action, reward, want_grad=False, next_price=None, rows=False):
        torch = _torch()
        if next_price is None:
            raise ThrlError("ActorCriticBatch.train needs next_price (the replayed new_state)")
        with torch.cuda.device(self.device):
            d_p, n, ld = self._rows(price, torch.float64, rows)
            d_a, _, lda = self._rows(action, torch.int32, rows)
            d_r, _, ldr = self._rows(reward, torch.float64, rows)
            d_n, _, ldn = self._rows(next_price, torch.float64, rows)
            if not (ld == lda == ldr == ldn):
                raise ThrlError("price / action / reward / next_price rows must share one pitch")
            grad = torch.zeros_like(self.params) if want_grad else None
            _lib.check(self.L.thrl_ac_train(self.G, self.A, self._p(self.params), self._p(self.adam_m),
                                            self._p(self.adam_v), self.step, n, ld, self._p(d_p), self._p(d_a), self._p(d_r),
                                            self._p(d_n), self.gamma, self.entropy, self.lr, self._p(self.gamma_g),
                                            self._p(self.entropy_g), self._p(grad), self._stream()), "thrl_ac_train")
            torch.cuda.synchronize(self.device)
        self.step += 1
        return grad


class CACBatch:
    """The reference's continuous actor-critic `CAC` (th_rl/agents.py:333-442) for G games: 1283
    parameters per game, actions are float32 in (0,1).  act(u1=None) returns the mean action
    sigmoid(mu) -- the reference's get_action raises under current torch (include/thrl.h)."""
    value_head = True

    def __init__(self, n_games, gamma=0.98, entropy=0.0, lr=2e-4, device="cuda:0", seed=0, game_offset=0,
                 agent_index=0, **_):
        self.L = _lib.load()
        torch = _torch()
        self.device = _require_gpu(device)
        self.G, self.P = int(n_games), _lib.CAC_PARAMS
        self.gamma, self.entropy, self.lr = float(gamma), float(entropy), float(lr)
        self.seed, self.game_offset, self.agent_index = int(seed), int(game_offset), int(agent_index)
        with torch.cuda.device(self.device):
            self.params = torch.zeros((self.G, self.P), dtype=torch.float32, device=self.device)
            self.adam_m = torch.zeros_like(self.params)
            self.adam_v = torch.zeros_like(self.params)
        self.step = 0
        self.gamma_g = self.entropy_g = None

    set_sweep = ReinforceBatch.set_sweep
    _stream = ReinforceBatch._stream
    _p = staticmethod(ReinforceBatch._p)
    _dev = ReinforceBatch._dev
    _rows = ReinforceBatch._rows
    set_params = ReinforceBatch.set_params

    def init(self):
        torch = _torch()
        with torch.cuda.device(self.device):
            _lib.check(self.L.thrl_cac_init(self.G, self._p(self.params), self.seed, self.game_offset, self.agent_index,
                                            self._stream()), "thrl_cac_init")
        self.adam_m.zero_(); self.adam_v.zero_(); self.step = 0
        return self

    def act(self, price, u1=None, u2=None, want_heads=False):
        """actions float32 [G] in (0,1) (device tensor); want_heads also returns (mu, std, v)."""
        torch = _torch()
        with torch.cuda.device(self.device):
            d_price = self._dev(price, torch.float64).reshape(self.G)
            d_u1 = None if u1 is None else self._dev(u1, torch.float64).reshape(self.G)
            d_u2 = None if u2 is None else self._dev(u2, torch.float64).reshape(self.G)
            out = torch.zeros((self.G,), dtype=torch.float32, device=self.device)
            heads = [torch.zeros_like(out) for _ in range(3)] if want_heads else [None] * 3
            _lib.check(self.L.thrl_cac_act(self.G, self._p(self.params), self._p(d_price), self._p(d_u1), self._p(d_u2),
                                           self._p(out), self._p(heads[0]), self._p(heads[1]), self._p(heads[2]),
                                           self._stream()), "thrl_cac_act")
        return (out, heads) if want_heads else out

    def train(self, price, action, reward, want_grad=False, next_price=None, rows=False):
        torch = _torch()
        if next_price is None:
            raise ThrlError("CACBatch.train needs next_price (the replayed new_state)")
        with torch.cuda.device(self.device):
            d_p, n, ld = self._rows(price, torch.float64, rows)
            d_a, _, lda = self._rows(action, torch.float32, rows)
            d_r, _, ldr = self._rows(reward, torch.float64, rows)
            d_n, _, ldn = self._rows(next_price, torch.float64, rows)
            if not (ld == lda == ldr == ldn):
                raise ThrlError("price / action / reward / next_price rows must share one pitch")
            grad = torch.zeros_like(self.params) if want_grad else None
            _lib.check(self.L.thrl_cac_train(self.G, self._p(self.params), self._p(self.adam_m), self._p(self.adam_v),
                                             self.step, n, ld, self._p(d_p), self._p(d_a), self._p(d_r), self._p(d_n),
                                             self.gamma, self.entropy, self.lr, self._p(self.gamma_g), self._p(self.entropy_g),
                                             self._p(grad), self._stream()),
                       "thrl_cac_train")
            torch.cuda.synchronize(self.device)
        self.step += 1
        return grad
