"""GameBatch: G independent pricing games resident in HBM, stepped in lockstep by
libthrl_hip.so.  PyTorch is used only for device memory and streams.

State kept on the device (layouts: include/thrl.h):
    q        [G, stride] float32|float64   QTable.table of every agent of every game
    counter  [G, stride] int32             QTable.counter
    state    [G]         float64           NoisyPriceState.state (last price)
Host-side state that is identical for every game: epsilon per agent, the
replay-buffer fill count, the global episode index.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import ThrlError

_KERNELS = {"auto": _lib.KERNEL_AUTO, "generic": _lib.KERNEL_GENERIC, "wave": _lib.KERNEL_WAVE,
            # the wave kernel with its code variant pinned (same results; include/thrl.h)
            "wave_plain": _lib.KERNEL_WAVE_PLAIN, "wave_greedy": _lib.KERNEL_WAVE_GREEDY, "tuple": _lib.KERNEL_TUPLE}
_WAVE_IDS = (_lib.KERNEL_WAVE, _lib.KERNEL_WAVE_PLAIN, _lib.KERNEL_WAVE_GREEDY)


def _torch():
    try:
        import torch
    except ImportError as e:  # pragma: no cover
        raise ThrlError("th_rl_amd needs PyTorch-ROCm for device memory: %s" % e)
    return torch


def _require_gpu(device):
    torch = _torch()
    if not torch.cuda.is_available():
        raise ThrlError("th_rl_amd: no GPU visible (torch.cuda.is_available() is False). "
                        "The hot path runs only on the HIP device; there is no CPU fallback.")
    return torch.device(device)


class GameBatch:
    def __init__(self, config, n_games=1, device="cuda:0", dtype="float32", seed=0, game_offset=0,
                 kernel="auto", counters=True, sweep=None):
        self.L = _lib.load()
        torch = _torch()
        self.device = _require_gpu(device)
        self.config = config
        self.dtype = {"float32": 0, "float64": 1, "f32": 0, "f64": 1}[str(dtype)]
        self.cfg, self.eps = _lib.cfg_from_config(config, n_games, self.dtype)
        self.G, self.N, self.T = int(n_games), self.cfg.n_agents, self.cfg.max_steps
        self.seed, self.game_offset = int(seed), int(game_offset)
        self.kernel = _KERNELS[kernel]
        self.episode = 0
        self.mem_count = [0] * _lib.MAXA
        self.last_kernel = None
        self.stride = int(self.L.thrl_table_stride(ctypes.byref(self.cfg)))
        if self.stride == 0:
            raise ThrlError("bad config: table stride is 0")
        self.offsets = [int(self.L.thrl_table_offset(ctypes.byref(self.cfg), i)) for i in range(self.N)]
        self.shapes = [(self.cfg.n_states[i] + 1, self.cfg.n_actions[i]) for i in range(self.N)]
        tdt = torch.float64 if self.dtype == 1 else torch.float32
        with torch.cuda.device(self.device):
            self.q = torch.empty((self.G, self.stride), dtype=tdt, device=self.device)
            self.counter = (torch.zeros((self.G, self.stride), dtype=torch.int32, device=self.device)
                            if counters else None)
            self.state = torch.zeros((self.G,), dtype=torch.float64, device=self.device)
            ws = int(self.L.thrl_workspace_bytes(ctypes.byref(self.cfg)))
            self.workspace = torch.empty((ws,), dtype=torch.uint8, device=self.device)
        self.replay_mem = None
        self.initialized = False
        self.sweep = {}
        if sweep:
            self.set_sweep(sweep)

    # ------------------------------------------------------------------ sweeps
    SWEEP_KEYS = ("gamma", "alpha", "eps_end", "eps_step", "eps", "noise_prob")

    def set_sweep(self, sweep):
        """Per-game hyper-parameters: dict of arrays [N, G] (or [G]: same for every agent) for
        gamma / alpha / eps_end / eps_step / eps (starting epsilon), and [G] for noise_prob.
        Keys that are absent keep the config's scalar for every game.  Call before init_tables()
        when gamma is swept (the initial table offset depends on it)."""
        torch = _torch()
        for k, v in sweep.items():
            if k not in self.SWEEP_KEYS:
                raise ThrlError("unknown sweep key %r (known: %s)" % (k, ", ".join(self.SWEEP_KEYS)))
            a = np.asarray(v, np.float64)
            if k == "noise_prob":
                a = a.reshape(self.G)
            else:
                a = np.broadcast_to(a.reshape(-1, self.G), (self.N, self.G)).copy()
            self.sweep[k] = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        if ("eps_end" in self.sweep or "eps_step" in self.sweep) and "eps" not in self.sweep:
            start = np.repeat(np.asarray(self.eps, np.float64)[:, None], self.G, axis=1)
            self.sweep["eps"] = torch.from_numpy(start).to(self.device)
        return self

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        torch = _torch()
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ptr(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def _ensure_replay_mem(self):
        if self.replay_mem is None:
            torch = _torch()
            n = int(self.L.thrl_replay_mem_bytes(ctypes.byref(self.cfg)))
            with torch.cuda.device(self.device):
                self.replay_mem = torch.empty((max(n, 256),), dtype=torch.uint8, device=self.device)

    def planned_kernel(self, injected=False):
        k = self.L.thrl_select_kernel(ctypes.byref(self.cfg), int(bool(injected)))
        if k < 0:
            _lib.check(k, "thrl_select_kernel")
        return _lib.KERNEL_NAMES[k]

    # ------------------------------------------------------------------ init / upload
    def init_tables(self):
        """QTable.__init__ + env.reset() for all games from Philox (thrl_qtable_init).  With a gamma
        sweep every game's table starts at ITS 12.5/(1-gamma) (agents.py:29), so game g of the sweep is
        the game a plain run of that config would initialise."""
        torch = _torch()
        with torch.cuda.device(self.device):
            rc = self.L.thrl_qtable_init(ctypes.byref(self.cfg), self._ptr(self.q), self._ptr(self.counter),
                                         self._ptr(self.state), self.seed, self.game_offset,
                                         self._ptr(self.sweep.get("gamma")), self._stream())
        _lib.check(rc, "thrl_qtable_init")
        self.initialized = True
        return self

    def set_tables(self, q, state, counter=None):
        """Upload host tables [G, stride] (or per-agent list for G==1) and states [G]."""
        torch = _torch()
        q = np.asarray(q)
        self.q.copy_(torch.from_numpy(np.ascontiguousarray(q.reshape(self.G, self.stride))).to(self.q.dtype))
        self.state.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(state, np.float64).reshape(self.G))))
        if self.counter is not None:
            if counter is None:
                self.counter.zero_()
            else:
                self.counter.copy_(torch.from_numpy(np.ascontiguousarray(
                    np.asarray(counter).reshape(self.G, self.stride).astype(np.int32))))
        self.initialized = True
        return self

    # ------------------------------------------------------------------ the hot path
    def run(self, n_episodes, inj=None, per_game_logs=False, sync=True, logs=True):
        """n_episodes of trainer.train_one's loop for all games.  Returns a dict with
        reward_log / action_log [E, N] (mean over games) as numpy (or device tensors
        when sync=False)."""
        torch = _torch()
        if not self.initialized:
            raise ThrlError("GameBatch: call init_tables() or set_tables() first")
        E, N, G, T = int(n_episodes), self.N, self.G, self.T
        b = _lib.Buffers()
        keep = []
        with torch.cuda.device(self.device):
            def dev(a, dt):
                t = torch.from_numpy(np.ascontiguousarray(a)).to(device=self.device, dtype=dt)
                keep.append(t)
                return t
            b.q, b.counter, b.state = self._ptr(self.q), self._ptr(self.counter), self._ptr(self.state)
            r_log = torch.zeros((E, N), dtype=torch.float64, device=self.device) if logs else None
            a_log = torch.zeros((E, N), dtype=torch.float64, device=self.device) if logs else None
            b.reward_log, b.action_log = self._ptr(r_log), self._ptr(a_log)
            g_r = g_a = None
            if per_game_logs:
                g_r = torch.zeros((E, N, G), dtype=torch.float64, device=self.device)
                g_a = torch.zeros((E, N, G), dtype=torch.float64, device=self.device)
                b.game_reward_log, b.game_action_log = self._ptr(g_r), self._ptr(g_a)
            injected = inj is not None
            if injected:
                u = np.asarray(inj["u"], np.float64)
                ch = np.asarray(inj["choice"], np.int8)
                if u.shape != (E, T, N, G) or ch.shape != (E, T, N, G):
                    raise ThrlError("injected draws must have shape [E,T,N,G]=%r" % ((E, T, N, G),))
                b.inj_u, b.inj_choice = self._ptr(dev(u, torch.float64)), self._ptr(dev(ch, torch.int8))
                if self.cfg.noise_prob > 0:
                    nu = np.asarray(inj["noise_u"], np.float64)
                    na = np.asarray(inj["noise_a"], np.float64)
                    if nu.shape != (E, T, G) or na.shape != (E, T, G):
                        raise ThrlError("injected noise draws must have shape [E,T,G]")
                    b.inj_noise_u, b.inj_noise_a = (self._ptr(dev(nu, torch.float64)),
                                                    self._ptr(dev(na, torch.float64)))
            # the wave kernel runs whole training cycles from empty buffers; anything else is the generic
            # kernel's, which keeps the buffers in replay_mem between calls
            cycle = int(self.L.thrl_training_cycle(ctypes.byref(self.cfg)))
            will_generic = (self.kernel == _lib.KERNEL_GENERIC or per_game_logs or cycle == 0
                            or E % max(cycle, 1) != 0
                            or (bool(self.sweep) and not all(self.cfg.min_memory[i] <= T <= self.cfg.capacity[i]
                                                                 for i in range(N)))
                            or any(self.mem_count[i] for i in range(N)))
            if will_generic and self.kernel not in _WAVE_IDS:
                self._ensure_replay_mem()
            if self.replay_mem is not None:
                b.replay_mem, b.replay_mem_bytes = self._ptr(self.replay_mem), self.replay_mem.numel()
            b.workspace, b.workspace_bytes = self._ptr(self.workspace), self.workspace.numel()
            for k in self.SWEEP_KEYS:
                setattr(b, "sweep_" + k, self._ptr(self.sweep.get(k)))
            r = _lib.Run()
            r.seed, r.game_offset, r.first_episode = self.seed, self.game_offset, self.episode
            r.n_episodes, r.kernel = E, self.kernel
            for i in range(N):
                r.eps[i] = self.eps[i]
                r.mem_count[i] = self.mem_count[i]
            rc = self.L.thrl_qtable_episodes(ctypes.byref(self.cfg), ctypes.byref(b), ctypes.byref(r),
                                             self._stream())
            _lib.check(rc, "thrl_qtable_episodes")
            self.eps = [r.eps[i] for i in range(N)]
            self.mem_count = [r.mem_count[i] for i in range(_lib.MAXA)]
            self.episode += E
            self.last_kernel = _lib.KERNEL_NAMES.get(r.kernel_used, "none")
            out = dict(kernel=self.last_kernel)
            if sync:
                torch.cuda.synchronize(self.device)
                if logs:
                    out["reward_log"], out["action_log"] = r_log.cpu().numpy(), a_log.cpu().numpy()
                if per_game_logs:
                    out["game_reward_log"], out["game_action_log"] = g_r.cpu().numpy(), g_a.cpu().numpy()
            else:
                out.update(reward_log=r_log, action_log=a_log, game_reward_log=g_r, game_action_log=g_a,
                           _keep=keep)
        return out

    # ------------------------------------------------------------------ evaluation
    def play_greedy(self, iters=1, state0=None):
        """utils.play_game for every game: per-iteration mean reward / scaled action
        per agent, arrays [iters, N, G]."""
        torch = _torch()
        with torch.cuda.device(self.device):
            mr = torch.zeros((iters, self.N, self.G), dtype=torch.float64, device=self.device)
            ma = torch.zeros((iters, self.N, self.G), dtype=torch.float64, device=self.device)
            s0 = None
            if state0 is not None:
                s0 = torch.from_numpy(np.ascontiguousarray(np.asarray(state0, np.float64)
                                                           .reshape(iters, self.G))).to(self.device)
            rc = self.L.thrl_play_greedy(ctypes.byref(self.cfg), self._ptr(self.q), self._ptr(s0), iters,
                                         self.seed, self.game_offset, self._ptr(mr), self._ptr(ma),
                                         self._stream())
            _lib.check(rc, "thrl_play_greedy")
            torch.cuda.synchronize(self.device)
        return mr.cpu().numpy(), ma.cpu().numpy()

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self):
        """Everything needed to continue the run bit-identically (the reference only saves
        final tables, trainer.py:101-110): device tensors + the host-side run state."""
        return {"q": self.q.cpu(), "counter": None if self.counter is None else self.counter.cpu(),
                "state": self.state.cpu(), "eps": [float(x) for x in self.eps], "episode": int(self.episode),
                "mem_count": [int(x) for x in self.mem_count], "seed": int(self.seed),
                "game_offset": int(self.game_offset), "dtype": int(self.dtype), "n_games": int(self.G),
                "offsets": list(self.offsets), "shapes": [list(x) for x in self.shapes],
                "replay_mem": None if self.replay_mem is None else self.replay_mem.cpu(),
                "sweep": {k: v.cpu() for k, v in self.sweep.items()}}

    def save(self, path):
        _torch().save(self.state_dict(), path)

    def load_state_dict(self, sd):
        torch = _torch()
        if int(sd["n_games"]) != self.G or [list(x) for x in sd["shapes"]] != [list(x) for x in self.shapes] \
                or int(sd["dtype"]) != self.dtype:
            raise ThrlError("checkpoint does not match this GameBatch (games / table shapes / dtype)")
        self.q.copy_(sd["q"])
        self.state.copy_(sd["state"])
        if self.counter is not None:
            if sd["counter"] is None:
                self.counter.zero_()
            else:
                self.counter.copy_(sd["counter"])
        self.eps = [float(x) for x in sd["eps"]]
        self.episode = int(sd["episode"])
        self.mem_count = [int(x) for x in sd["mem_count"]]
        self.seed, self.game_offset = int(sd["seed"]), int(sd["game_offset"])
        for k, v in (sd.get("sweep") or {}).items():
            self.sweep[k] = v.to(self.device)
        if sd.get("replay_mem") is not None:
            self._ensure_replay_mem()
            self.replay_mem.copy_(sd["replay_mem"])
        self.initialized = True
        return self

    def load(self, path):
        return self.load_state_dict(_torch().load(path, weights_only=True))

    # ------------------------------------------------------------------ download
    def tables_numpy(self):
        return self.q.cpu().numpy()

    def counters_numpy(self):
        return None if self.counter is None else self.counter.cpu().numpy()

    def states_numpy(self):
        return self.state.cpu().numpy()

    def table(self, game, agent):
        """Agent's table of one game in the reference's format: float64 (states+1, actions)."""
        r, a = self.shapes[agent]
        o = self.offsets[agent]
        return self.q[game, o:o + r * a].cpu().numpy().astype(np.float64).reshape(r, a)

    def counter_of(self, game, agent):
        r, a = self.shapes[agent]
        o = self.offsets[agent]
        if self.counter is None:
            return np.zeros((r, a))
        return self.counter[game, o:o + r * a].cpu().numpy().astype(np.float64).reshape(r, a)
