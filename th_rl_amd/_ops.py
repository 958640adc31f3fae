"""Single-object (G = 1) access to the unfused device operators of libthrl_hip.so.

The reference's duck-typed protocol calls one agent / one env at a time
(trainer.py:52-62).  These helpers run each such call on the GPU through the
thrl_op_* entry points, so the object-level API has no CPU arithmetic path
either.  (They are slow -- one launch + sync per call; the fast path is
GameBatch.run / train_one.)
"""
import ctypes

import numpy as np

from . import _lib
from .batched import _require_gpu, _torch


class DeviceOps:
    def __init__(self, cfg, device="cuda:0"):
        self.L = _lib.load()
        self.device = _require_gpu(device)
        self.cfg = cfg
        self.torch = _torch()

    def _dev(self, arr, dtype):
        t = self.torch.from_numpy(np.ascontiguousarray(arr)).to(device=self.device, dtype=dtype)
        return t

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def encode(self, agent, prices, as_float32):
        torch = self.torch
        prices = np.atleast_1d(np.asarray(prices, np.float64)).ravel()
        out = np.zeros(len(prices), np.int64)
        with torch.cuda.device(self.device):
            for k, p in enumerate(prices):      # cfg.n_games == 1: one state per launch
                d_p = self._dev([p], torch.float64)
                d_o = torch.zeros(1, dtype=torch.int32, device=self.device)
                _lib.check(self.L.thrl_op_encode(ctypes.byref(self.cfg), agent, self._p(d_p), int(as_float32),
                                                 self._p(d_o), self._stream()), "thrl_op_encode")
                out[k] = int(d_o.cpu()[0])
        return out

    def scale(self, agent, action):
        torch = self.torch
        with torch.cuda.device(self.device):
            d_a = self._dev([int(action)], torch.int32)
            d_o = torch.zeros(1, dtype=torch.float64, device=self.device)
            _lib.check(self.L.thrl_op_scale(ctypes.byref(self.cfg), agent, self._p(d_a), self._p(d_o),
                                            self._stream()), "thrl_op_scale")
            return float(d_o.cpu()[0])

    def greedy_action(self, agent, table, price, as_float32):
        torch = self.torch
        with torch.cuda.device(self.device):
            d_q = self._dev(np.asarray(table, np.float64).reshape(1, -1), torch.float64)
            d_p = self._dev([float(price)], torch.float64)
            d_o = torch.zeros(1, dtype=torch.int32, device=self.device)
            _lib.check(self.L.thrl_op_sample_action(ctypes.byref(self.cfg), agent, self._p(d_q), self._p(d_p),
                                                    0.0, None, None, int(as_float32), self._p(d_o),
                                                    self._stream()), "thrl_op_sample_action")
            return int(d_o.cpu()[0])

    def env_step(self, scaled, noise_u, noise_a):
        torch = self.torch
        n = self.cfg.n_agents
        with torch.cuda.device(self.device):
            d_s = self._dev(np.asarray(scaled, np.float64).reshape(n, 1), torch.float64)
            d_nu = self._dev([float(noise_u)], torch.float64)
            d_na = self._dev([float(noise_a)], torch.float64)
            d_p = torch.zeros(1, dtype=torch.float64, device=self.device)
            d_r = torch.zeros((n, 1), dtype=torch.float64, device=self.device)
            _lib.check(self.L.thrl_op_env_step(ctypes.byref(self.cfg), self._p(d_s), self._p(d_nu), self._p(d_na),
                                               self._p(d_p), self._p(d_r), self._stream()), "thrl_op_env_step")
            return float(d_p.cpu()[0]), d_r.cpu().numpy()[:, 0].copy()

    def td_update(self, agent, table, counter, price, action, reward, next_price):
        """Returns (new_table, new_counter) as float64 numpy (rows, A)."""
        torch = self.torch
        n = len(action)
        shp = np.asarray(table).shape
        with torch.cuda.device(self.device):
            d_q = self._dev(np.asarray(table, np.float64).reshape(1, -1), torch.float64)
            d_c = self._dev(np.asarray(counter).reshape(1, -1).astype(np.int32), torch.int32)
            d_p = self._dev(np.asarray(price, np.float64).reshape(n, 1), torch.float64)
            d_a = self._dev(np.asarray(action).reshape(n, 1).astype(np.int32), torch.int32)
            d_r = self._dev(np.asarray(reward, np.float64).reshape(n, 1), torch.float64)
            d_n = self._dev(np.asarray(next_price, np.float64).reshape(n, 1), torch.float64)
            d_s = torch.zeros((n, 1), dtype=torch.float64, device=self.device)
            _lib.check(self.L.thrl_op_td_update(ctypes.byref(self.cfg), agent, self._p(d_q), self._p(d_c), n,
                                                self._p(d_p), self._p(d_a), self._p(d_r), self._p(d_n),
                                                self._p(d_s), self._stream()), "thrl_op_td_update")
            return (d_q.cpu().numpy().reshape(shp),
                    d_c.cpu().numpy().reshape(shp).astype(np.float64))
