"""MixedGameBatch: G games whose agents may be any mix of QTable, Reinforce, ActorCritic and CAC
(the reference's example configs pair a QTable with a Reinforce agent), stepped in lockstep on the
device.  Two equivalent ways to run trainer.train_one's loop (th_rl/trainer.py:46-70):

  * fused (default): thrl_mixed_episodes -- one wavefront per game plays all episodes between two
    network updates in ONE launch (tables in LDS, networks in registers), then the batched update
    kernels (thrl_nn_reinforce_train / thrl_ac_train / thrl_cac_train) run when len(memory) >=
    min_memory;
  * unfused (`run(fused=False)`): the reference's call sequence, one launch per reference call,
        thrl_op_draws -> thrl_op_sample_action | thrl_nn_act -> thrl_op_scale -> thrl_op_env_step
        -> (append) -> per episode thrl_op_td_update | train
    kept as the reference-shaped checker of the fused kernel (bit-identical results, tested).

The random streams are the ones GameBatch's kernels use, so an all-QTable game gives bit-identical
tables on all paths.  Per-game hyper-parameter sweeps: `sweep=` (see set_sweep).
torch is used for device memory and for three one-line float64 element-wise expressions of the
unfused loop (Reinforce.scale and the two log accumulations).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import ThrlError
from .batched import _require_gpu, _torch
from .nn import ActorCriticBatch, CACBatch, ReinforceBatch

NN_DEFAULTS = dict(states=4, actions=2, action_range=[0, 1], gamma=0.98, capacity=50000, min_memory=1000,
                   entropy=0)


class MixedGameBatch:
    def __init__(self, config, n_games=1, device="cuda:0", dtype="float32", seed=0, game_offset=0, sweep=None):
        self.L = _lib.load()
        torch = _torch()
        self.device = _require_gpu(device)
        self.config = config
        self.G = int(n_games)
        self.seed, self.game_offset = int(seed), int(game_offset)
        self.dtype = {"float32": 0, "float64": 1}[str(dtype)]
        self.kinds = [a.get("name", "QTable") for a in config["agents"]]
        for k in self.kinds:
            if k not in ("QTable", "Reinforce", "ActorCritic", "CAC"):
                raise NotImplementedError("device path: agent %r is not supported "
                                          "(QTable, Reinforce, ActorCritic, CAC)" % k)
        # thrl_cfg for the operators: a Reinforce agent occupies a dummy 2-row table slot
        as_q = []
        self.nn_cfg = {}
        for i, a in enumerate(config["agents"]):
            if self.kinds[i] == "QTable":
                as_q.append(dict(a))
            else:
                p = dict(NN_DEFAULTS, **a)
                if int(p["states"]) != 1:
                    raise ThrlError("%s on the device needs states == 1 (the env state is one number)" % self.kinds[i])
                self.nn_cfg[i] = p
                if self.kinds[i] == "CAC":
                    p["actions"] = 2                 # continuous: the table slot is a placeholder
                as_q.append(dict(name="QTable", states=1, actions=int(p["actions"]), action_range=p["action_range"],
                                 capacity=1, min_memory=1))
        qconf = {"agents": as_q, "environment": config["environment"]}
        self.cfg, eps = _lib.cfg_from_config(qconf, self.G, self.dtype)
        self.N, self.T = self.cfg.n_agents, self.cfg.max_steps
        self.eps = list(eps)
        self.stride = int(self.L.thrl_table_stride(ctypes.byref(self.cfg)))
        self.offsets = [int(self.L.thrl_table_offset(ctypes.byref(self.cfg), i)) for i in range(self.N)]
        self.shapes = [(self.cfg.n_states[i] + 1, self.cfg.n_actions[i]) for i in range(self.N)]
        tdt = torch.float64 if self.dtype == 1 else torch.float32
        with torch.cuda.device(self.device):
            self.q = torch.zeros((self.G, self.stride), dtype=tdt, device=self.device)
            self.counter = torch.zeros((self.G, self.stride), dtype=torch.int32, device=self.device)
            self.state = torch.zeros((self.G,), dtype=torch.float64, device=self.device)
        classes = {"Reinforce": ReinforceBatch, "ActorCritic": ActorCriticBatch, "CAC": CACBatch}
        self.nn = {i: classes[self.kinds[i]](
            self.G, actions=int(p["actions"]), gamma=float(p["gamma"]), entropy=float(p["entropy"]), device=device,
            seed=seed, game_offset=game_offset, agent_index=i) for i, p in self.nn_cfg.items()}
        self.cap = [int(self.cfg.capacity[i]) if self.kinds[i] == "QTable" else int(self.nn_cfg[i]["capacity"])
                    for i in range(self.N)]
        self.min_memory = [int(self.cfg.min_memory[i]) if self.kinds[i] == "QTable" else int(self.nn_cfg[i]["min_memory"])
                           for i in range(self.N)]
        # replay buffers hold at most min_memory + T - 1 entries before a train call empties them
        self.buf_len = [min(self.cap[i], self.min_memory[i] + self.T) if self.cap[i] >= self.min_memory[i] else self.cap[i]
                        for i in range(self.N)]
        with torch.cuda.device(self.device):
            # replay rings, game-major [G, buf_len] (include/thrl.h, ABI v3): a game's slots are contiguous, so the
            # episode kernel's 16-step flushes coalesce and an update kernel reads a game's whole batch as one row
            self.buf = [dict(price=torch.zeros((self.G, max(n, 1)), dtype=torch.float64, device=self.device),
                             action=torch.zeros((self.G, max(n, 1)), device=self.device,
                                                dtype=torch.float32 if self.kinds[i] == "CAC" else torch.int32),
                             reward=torch.zeros((self.G, max(n, 1)), dtype=torch.float64, device=self.device),
                             nprice=torch.zeros((self.G, max(n, 1)), dtype=torch.float64, device=self.device))
                        for i, n in enumerate(self.buf_len)]
        # a network update replays the whole buffer: T * ceil(min_memory / T) transitions (the check runs at
        # episode ends only).  The update kernels keep the batch in LDS, so reject a config that cannot
        # train HERE, not at the first update half-way through a run.
        for i, p in self.nn_cfg.items():
            if self.cap[i] < self.min_memory[i] or self.min_memory[i] <= 0:
                continue                      # never trains (or trains on every call with whatever is there)
            n_train = min(self.cap[i], self.T * -(-self.min_memory[i] // self.T))
            limit = 5600 if self.kinds[i] == "CAC" else 1400          # THRL_NN_MAX_TRANSITIONS / CAC's LDS bound
            if n_train > limit:
                raise ThrlError("%s agent %d would train on %d transitions per update (min_memory=%d, max_steps=%d); "
                                "the device update kernel takes at most %d" % (self.kinds[i], i, n_train,
                                                                                self.min_memory[i], self.T, limit))
        self.policy_table = True             # False: the fused kernel evaluates the policy at every step (same results)
        self.tuple_kernel = True             # False: keep the general fused kernel where the tuple-chain one applies (same results)
        self.last_episode_kernel = None
        self._ptab = None
        self.count = [0] * self.N            # appends since the last empty()
        self.episode = 0
        self.initialized = False
        self._fused_launched = False         # True once a fused launch has advanced the state
        self.sweep = {}
        if sweep:
            self.set_sweep(sweep)

    # ------------------------------------------------------------------ sweeps
    SWEEP_KEYS = ("gamma", "alpha", "eps_end", "eps_step", "eps", "noise_prob", "entropy")

    def set_sweep(self, sweep):
        """Per-game hyper-parameters -- the reference's config sweep (one process per config and run,
        main.py:13-21) as ONE batch.  dict of arrays [N, G] (or [G]: the same for every agent) for
        gamma / alpha / eps (starting epsilon) / eps_end / eps_step / entropy, and [G] for noise_prob.
        A row applies to whatever agent sits in that slot: gamma is QTable.gamma or the neural agent's
        discount, alpha / eps* are read for QTable agents only, entropy for neural agents only.  Absent
        keys keep the config's scalar.  Call before init_tables() when gamma is swept."""
        torch = _torch()
        for k, v in sweep.items():
            if k not in self.SWEEP_KEYS:
                raise ThrlError("unknown sweep key %r (known: %s)" % (k, ", ".join(self.SWEEP_KEYS)))
            a = np.asarray(v, np.float64)
            a = a.reshape(self.G) if k == "noise_prob" else np.broadcast_to(a.reshape(-1, self.G), (self.N, self.G)).copy()
            self.sweep[k] = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        if ("eps_end" in self.sweep or "eps_step" in self.sweep) and "eps" not in self.sweep:
            start = np.repeat(np.asarray(self.eps[:self.N], np.float64)[:, None], self.G, axis=1)
            self.sweep["eps"] = torch.from_numpy(start).to(self.device)
        for i, rb in self.nn.items():            # the neural agents' own sweeps go to their update kernels
            rb.set_sweep(gamma=self.sweep["gamma"][i] if "gamma" in self.sweep else None,
                         entropy=self.sweep["entropy"][i] if "entropy" in self.sweep else None)
        return self

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return ctypes.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def init_tables(self):
        torch = _torch()
        with torch.cuda.device(self.device):
            init_gamma = None
            if "gamma" in self.sweep:        # only QTable rows move the table offset 12.5/(1-gamma); a neural agent's
                init_gamma = self.sweep["gamma"].clone()      # placeholder slot keeps its (unused) default
                for i in self.nn:
                    init_gamma[i] = float(self.cfg.gamma[i])
            _lib.check(self.L.thrl_qtable_init(ctypes.byref(self.cfg), self._p(self.q), self._p(self.counter),
                                               self._p(self.state), self.seed, self.game_offset,
                                               self._p(init_gamma), self._stream()),
                       "thrl_qtable_init")
            torch.cuda.synchronize(self.device)          # init_gamma is a temporary
        for rb in self.nn.values():
            rb.init()
        self.initialized = True
        return self

    def set_tables(self, q, state):
        torch = _torch()
        self.q.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(q).reshape(self.G, self.stride))).to(self.q.dtype))
        self.state.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(state, np.float64).reshape(self.G))))
        self.counter.zero_()
        self.initialized = True
        return self

    def states_numpy(self):
        return self.state.cpu().numpy()

    def tables_numpy(self):
        return self.q.cpu().numpy()

    def counters_numpy(self):
        return self.counter.cpu().numpy()

    def table(self, game, agent):
        r, a = self.shapes[agent]; o = self.offsets[agent]
        return self.q[game, o:o + r * a].cpu().numpy().astype(np.float64).reshape(r, a)

    def counter_of(self, game, agent):
        r, a = self.shapes[agent]; o = self.offsets[agent]
        return self.counter[game, o:o + r * a].cpu().numpy().astype(np.float64).reshape(r, a)

    # ------------------------------------------------------------------ greedy evaluation
    def play_greedy(self, iters=1, state0=None):
        """utils.play_game (utils.py:27-47) for every game: each agent's get_action (QTable: argmax
        on the float64-encoded state; Reinforce / ActorCritic: argmax of pi; CAC: the mean action),
        env.step, for `iters` episodes.  Returns per-iteration mean reward and mean scaled action,
        arrays [iters, N, G].  state0 [iters, G] = the states environment.reset() would draw
        (default: uniform(0, a) from numpy RandomState(seed)).  Learning state is not touched."""
        torch = _torch()
        if not self.initialized:
            raise ThrlError("MixedGameBatch: call init_tables() or set_tables() first")
        N, G, T = self.N, self.G, self.T
        cfg, L = ctypes.byref(self.cfg), self.L
        noise = self.cfg.noise_prob > 0
        if state0 is None:
            state0 = np.random.RandomState(self.seed % (2 ** 32)).uniform(0, self.cfg.env_a, (iters, G))
        state0 = np.asarray(state0, np.float64).reshape(iters, G)
        with torch.cuda.device(self.device):
            mr = torch.zeros((iters, N, G), dtype=torch.float64, device=self.device)
            ma = torch.zeros((iters, N, G), dtype=torch.float64, device=self.device)
            u = torch.zeros((N, G), dtype=torch.float64, device=self.device)
            ch = torch.zeros((N, G), dtype=torch.int8, device=self.device)
            nu = torch.zeros((G,), dtype=torch.float64, device=self.device) if noise else None
            na = torch.zeros((G,), dtype=torch.float64, device=self.device) if noise else None
            acts = torch.zeros((N, G), dtype=torch.int32, device=self.device)
            scaled = torch.zeros((N, G), dtype=torch.float64, device=self.device)
            nprice = torch.zeros((G,), dtype=torch.float64, device=self.device)
            reward = torch.zeros((N, G), dtype=torch.float64, device=self.device)
            T_t = torch.tensor(float(T), dtype=torch.float64, device=self.device)
            for it in range(iters):
                price = torch.from_numpy(np.ascontiguousarray(state0[it])).to(self.device)
                for t in range(T):
                    if noise:     # the env draws its noise uniforms in play as well (environments.py:28)
                        _lib.check(L.thrl_op_draws(cfg, self.seed, self.game_offset, (1 << 31) + it, t, self._p(u),
                                                   self._p(ch), None, self._p(nu), self._p(na), self._stream()),
                                   "thrl_op_draws")
                    for i in range(N):
                        lo, hi = float(self.cfg.act_lo[i]), float(self.cfg.act_hi[i])
                        if self.kinds[i] == "QTable":
                            _lib.check(L.thrl_op_sample_action(cfg, i, self._p(self.q), self._p(price), 0.0, None, None,
                                                               0, self._p(acts[i]), self._stream()),
                                       "thrl_op_sample_action")
                            _lib.check(L.thrl_op_scale(cfg, i, self._p(acts[i]), self._p(scaled[i]), self._stream()),
                                       "thrl_op_scale")
                        elif self.kinds[i] == "CAC":
                            scaled[i].copy_(self.nn[i].act(price).to(torch.float64) * (hi - lo) + lo)
                        else:
                            A_t = torch.tensor(float(self.nn[i].A), dtype=torch.float64, device=self.device)
                            scaled[i].copy_(torch.div(self.nn[i].act(price).to(torch.float64), A_t) * (hi - lo) + lo)
                    _lib.check(L.thrl_op_env_step(cfg, self._p(scaled), self._p(nu), self._p(na), self._p(nprice),
                                                  self._p(reward), self._stream()), "thrl_op_env_step")
                    mr[it] += torch.div(reward, T_t)
                    ma[it] += torch.div(scaled, T_t)
                    price = nprice.clone()
            torch.cuda.synchronize(self.device)
        return mr.cpu().numpy(), ma.cpu().numpy()

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self):
        """Everything a continued run needs (plain tensors / numbers: loads with weights_only=True)."""
        return dict(version=2, kind="mixed", n_games=self.G, kinds=list(self.kinds), shapes=[list(x) for x in self.shapes],
                    dtype=self.dtype, seed=self.seed, game_offset=self.game_offset, episode=self.episode,
                    eps=[float(x) for x in self.eps], count=[int(x) for x in self.count],
                    q=self.q.cpu(), counter=self.counter.cpu(), state=self.state.cpu(),
                    buffers=[{k: v.cpu() for k, v in b.items()} for b in self.buf],
                    sweep={k: v.cpu() for k, v in self.sweep.items()},
                    nn={int(i): dict(params=rb.params.cpu(), adam_m=rb.adam_m.cpu(), adam_v=rb.adam_v.cpu(),
                                     step=int(rb.step)) for i, rb in self.nn.items()})

    def save(self, path):
        _torch().save(self.state_dict(), path)

    def load_state_dict(self, sd):
        if sd.get("kind") != "mixed" or int(sd["n_games"]) != self.G or list(sd["kinds"]) != list(self.kinds) \
                or [list(x) for x in sd["shapes"]] != [list(x) for x in self.shapes] or int(sd["dtype"]) != self.dtype:
            raise ThrlError("checkpoint does not match this MixedGameBatch (games / agent kinds / shapes / dtype)")
        self.q.copy_(sd["q"]); self.counter.copy_(sd["counter"]); self.state.copy_(sd["state"])
        self.eps = [float(x) for x in sd["eps"]]
        self.count = [int(x) for x in sd["count"]]
        self.episode = int(sd["episode"])
        self.seed, self.game_offset = int(sd["seed"]), int(sd["game_offset"])
        for b, src in zip(self.buf, sd["buffers"]):
            for k in b:            # version 1 checkpoints hold the rings transition-major [buf_len, G]
                b[k].copy_(src[k].t() if int(sd.get("version", 1)) < 2 else src[k])
        if sd.get("sweep"):
            self.set_sweep({k: v.numpy() for k, v in sd["sweep"].items()})
        for i, rb in self.nn.items():
            src = sd["nn"][int(i)]
            rb.params.copy_(src["params"]); rb.adam_m.copy_(src["adam_m"]); rb.adam_v.copy_(src["adam_v"])
            rb.step = int(src["step"])
            rb.seed, rb.game_offset = self.seed, self.game_offset
        self.initialized = True
        return self

    def load(self, path):
        return self.load_state_dict(_torch().load(path, weights_only=True))

    # ------------------------------------------------------------------ the step loop
    def _append(self, i, price, action, reward, nprice):
        cap = self.buf_len[i]
        if cap <= 0:
            return
        pos = self.count[i] % cap
        b = self.buf[i]
        b["price"][:, pos].copy_(price); b["action"][:, pos].copy_(action)
        b["reward"][:, pos].copy_(reward); b["nprice"][:, pos].copy_(nprice)
        self.count[i] += 1
        if self.count[i] >= 2 * cap:
            self.count[i] -= cap

    def _ordered(self, i):
        """Buffer contents in insertion order (deque semantics) as [G, n] tensors: views of the rings when the
        ring has not wrapped (the usual case: no copy, the update kernels take the row pitch), else gathered."""
        torch = _torch()
        cap = self.buf_len[i]
        n = min(self.count[i], cap)
        start = 0 if self.count[i] <= cap else self.count[i] % cap
        b = self.buf[i]
        if start == 0:
            return n, {k: v[:, :n] for k, v in b.items()}
        idx = (torch.arange(n, device=self.device) + start) % cap
        return n, {k: v.index_select(1, idx).contiguous() for k, v in b.items()}

    def run(self, n_episodes, fused=None, per_game_logs=True):
        """n_episodes for all games.  fused=True: thrl_mixed_episodes, one launch per run of episodes
        between network updates; fused=False: the per-call operator loop (same results); None
        (default): fused unless the library reports the configuration as unsupported by that kernel
        (more than two Reinforce / ActorCritic agents, tables beyond 64 KiB of LDS per game, > 64 actions)."""
        if fused is None:
            first = self.episode
            try:
                return self._run_fused(int(n_episodes), per_game_logs)
            except ThrlError as e:
                # fall back only if the episode kernel itself refused the configuration, i.e. before any
                # launch changed tables / buffers / episode index
                if e.code != _lib.ERR_UNSUPPORTED or self.episode != first or self._fused_launched:
                    raise
                return self._run_unfused(int(n_episodes))
        if fused:
            return self._run_fused(int(n_episodes), per_game_logs)
        return self._run_unfused(int(n_episodes))

    def _run_fused(self, E, per_game_logs=True):
        """per_game_logs=False keeps only the mean over games (reduced on the device, launch by launch):
        what train_one needs, without E x N x G arrays crossing to the host."""
        torch = _torch()
        if not self.initialized:
            raise ThrlError("MixedGameBatch: call init_tables() or set_tables() first")
        N, G, T = self.N, self.G, self.T
        kmax = E if per_game_logs else max(1, (1 << 25) // (N * G))      # <= 256 MiB per log buffer
        with torch.cuda.device(self.device):
            rows = E if per_game_logs else min(E, kmax)
            rlog = torch.zeros((rows, N, G), dtype=torch.float64, device=self.device)
            alog = torch.zeros((rows, N, G), dtype=torch.float64, device=self.device)
            rmean = torch.zeros((E, N), dtype=torch.float64, device=self.device)
            amean = torch.zeros((E, N), dtype=torch.float64, device=self.device)
            if not hasattr(self, "_scratch"):
                self._scratch = [torch.zeros_like(b["price"]) if self.kinds[i] == "QTable" else None
                                 for i, b in enumerate(self.buf)]
            done = 0
            while done < E:
                k = min(E - done, kmax)
                base = done if per_game_logs else 0
                for i in range(N):                     # stop where a network update is due
                    if self.kinds[i] != "QTable" and self.buf_len[i] > 0 and self.buf_len[i] >= self.min_memory[i]:
                        have = min(self.count[i], self.buf_len[i])
                        need = max(1, -(-(self.min_memory[i] - have) // T))
                        k = min(k, need)
                mx = _lib.Mixed()
                for i in range(N):
                    mx.kind[i] = {"QTable": 0, "Reinforce": 1, "ActorCritic": 2, "CAC": 3}[self.kinds[i]]
                    if self.kinds[i] != "QTable":
                        mx.nn_params[i] = self.nn[i].params.data_ptr()
                    b = self.buf[i]
                    mx.buf_price[i], mx.buf_action[i] = b["price"].data_ptr(), b["action"].data_ptr()
                    mx.buf_reward[i], mx.buf_nprice[i] = b["reward"].data_ptr(), b["nprice"].data_ptr()
                    if self._scratch[i] is not None:
                        mx.buf_scratch[i] = self._scratch[i].data_ptr()
                    mx.buf_len[i], mx.min_memory[i], mx.count[i] = self.buf_len[i], self.min_memory[i], self.count[i]
                for key in ("gamma", "alpha", "eps_end", "eps_step", "eps", "noise_prob"):
                    if key in self.sweep:
                        setattr(mx, "sweep_" + key, self.sweep[key].data_ptr())
                if self.policy_table:
                    # scratch of the kernel's policy table (include/thrl.h): allocated once, contents per launch
                    if self._ptab is None:
                        need = int(self.L.thrl_mixed_policy_table_bytes(ctypes.byref(self.cfg), ctypes.byref(mx)))
                        self._ptab = torch.empty((max(need, 4) // 4,), dtype=torch.float32, device=self.device) if need else False
                    if self._ptab is not False:
                        mx.policy_tab, mx.policy_tab_bytes = self._ptab.data_ptr(), self._ptab.numel() * 4
                if not self.tuple_kernel:
                    mx.flags = 1                       # THRL_MIXED_NO_TUPLE_KERNEL
                r = _lib.Run()
                r.seed, r.game_offset, r.first_episode, r.n_episodes = self.seed, self.game_offset, self.episode, k
                for i in range(N):
                    r.eps[i] = self.eps[i]
                _lib.check(self.L.thrl_mixed_episodes(ctypes.byref(self.cfg), ctypes.byref(mx), self._p(self.q),
                                                      self._p(self.counter), self._p(self.state), ctypes.byref(r),
                                                      self._p(rlog[base:]), self._p(alog[base:]), self._stream()),
                           "thrl_mixed_episodes")
                self._fused_launched = True
                self.last_episode_kernel = "tuple" if r.kernel_used == _lib.KERNEL_TUPLE else "wave"
                rmean[done:done + k] = rlog[base:base + k].mean(dim=2)
                amean[done:done + k] = alog[base:base + k].mean(dim=2)
                self.eps = [r.eps[i] for i in range(N)] + self.eps[N:]
                self.count = [mx.count[i] for i in range(N)]
                self.episode += k
                done += k
                for i in range(N):                     # Reinforce.train_net (agents.py:170-194)
                    if self.kinds[i] != "QTable":
                        n, b = self._ordered(i)
                        if n >= self.min_memory[i] and n > 0:
                            self.nn[i].train(b["price"], b["action"], b["reward"], next_price=b["nprice"], rows=True)
                            self.count[i] = 0
            torch.cuda.synchronize(self.device)
            out = dict(kernel="mixed-fused", episode_kernel=self.last_episode_kernel, reward_log=rmean.cpu().numpy(),
                       action_log=amean.cpu().numpy())
            if per_game_logs:
                out.update(game_reward_log=rlog.cpu().numpy(), game_action_log=alog.cpu().numpy())
        return out

    def _run_unfused(self, n_episodes):
        torch = _torch()
        if not self.initialized:
            raise ThrlError("MixedGameBatch: call init_tables() or set_tables() first")
        E, N, G, T = int(n_episodes), self.N, self.G, self.T
        if self.sweep:
            raise ThrlError("per-game sweeps run on the fused path (run(fused=True)); the operator loop takes the "
                            "config's scalars only")
        cfg = ctypes.byref(self.cfg)
        L = self.L
        noise = self.cfg.noise_prob > 0
        with torch.cuda.device(self.device):
            rlog = torch.zeros((E, N, G), dtype=torch.float64, device=self.device)
            alog = torch.zeros((E, N, G), dtype=torch.float64, device=self.device)
            u = torch.zeros((N, G), dtype=torch.float64, device=self.device)
            ch = torch.zeros((N, G), dtype=torch.int8, device=self.device)
            has_cac = "CAC" in self.kinds
            u2 = torch.zeros((N, G), dtype=torch.float64, device=self.device) if has_cac else None
            acts_f = torch.zeros((N, G), dtype=torch.float32, device=self.device) if has_cac else None
            nu = torch.zeros((G,), dtype=torch.float64, device=self.device) if noise else None
            na = torch.zeros((G,), dtype=torch.float64, device=self.device) if noise else None
            acts = torch.zeros((N, G), dtype=torch.int32, device=self.device)
            scaled = torch.zeros((N, G), dtype=torch.float64, device=self.device)
            nprice = torch.zeros((G,), dtype=torch.float64, device=self.device)
            reward = torch.zeros((N, G), dtype=torch.float64, device=self.device)
            price = self.state
            # a tensor divisor: torch turns "/ python_scalar" into "* (1/scalar)", which is not the
            # reference's IEEE division
            T_t = torch.tensor(float(T), dtype=torch.float64, device=self.device)
            for e in range(E):
                for t in range(T):
                    _lib.check(L.thrl_op_draws(cfg, self.seed, self.game_offset, self.episode, t, self._p(u), self._p(ch),
                                               self._p(u2), self._p(nu), self._p(na), self._stream()), "thrl_op_draws")
                    for i in range(N):
                        if self.kinds[i] == "QTable":
                            _lib.check(L.thrl_op_sample_action(cfg, i, self._p(self.q), self._p(price), self.eps[i],
                                                               self._p(u[i]), self._p(ch[i]), 1, self._p(acts[i]),
                                                               self._stream()), "thrl_op_sample_action")
                            _lib.check(L.thrl_op_scale(cfg, i, self._p(acts[i]), self._p(scaled[i]), self._stream()),
                                       "thrl_op_scale")
                        elif self.kinds[i] == "CAC":
                            lo, hi = [float(x) for x in self.nn_cfg[i]["action_range"]]
                            acts_f[i].copy_(self.nn[i].act(price, u1=u[i], u2=u2[i]))
                            # CAC.scale (agents.py:371-375): action * (hi - lo) + lo on the Python float
                            scaled[i].copy_(acts_f[i].to(torch.float64) * (hi - lo) + lo)
                        else:
                            rb = self.nn[i]
                            acts[i].copy_(rb.act(price, u=u[i]))
                            lo, hi = [float(x) for x in self.nn_cfg[i]["action_range"]]
                            # Reinforce.scale (agents.py:153-157): action / actions * (hi - lo) + lo
                            A_t = torch.tensor(float(rb.A), dtype=torch.float64, device=self.device)
                            scaled[i].copy_(torch.div(acts[i].to(torch.float64), A_t) * (hi - lo) + lo)
                    _lib.check(L.thrl_op_env_step(cfg, self._p(scaled), self._p(nu), self._p(na), self._p(nprice),
                                                  self._p(reward), self._stream()), "thrl_op_env_step")
                    for i in range(N):
                        self._append(i, price, acts_f[i] if self.kinds[i] == "CAC" else acts[i], reward[i], nprice)
                    rlog[e] += torch.div(reward, T_t)           # trainer.py:65
                    alog[e] += torch.div(scaled, T_t)           # trainer.py:66
                    price = nprice.clone()
                for i in range(N):                              # [A.train_net() for A in agents]
                    n, b = self._ordered(i)
                    if n >= self.min_memory[i] and n > 0:
                        if self.kinds[i] == "QTable":
                            scratch = torch.zeros((n, G), dtype=torch.float64, device=self.device)
                            bt = {k: v.t().contiguous() for k, v in b.items()}      # the operator form is [n][G]
                            _lib.check(L.thrl_op_td_update(cfg, i, self._p(self.q), self._p(self.counter), n,
                                                           self._p(bt["price"]), self._p(bt["action"]), self._p(bt["reward"]),
                                                           self._p(bt["nprice"]), self._p(scratch), self._stream()),
                                       "thrl_op_td_update")
                            torch.cuda.synchronize(self.device)
                        else:
                            self.nn[i].train(b["price"], b["action"], b["reward"], next_price=b["nprice"], rows=True)
                        self.count[i] = 0
                    if self.kinds[i] == "QTable":                # epsilon decays on every call (agents.py:78)
                        self.eps[i] = self.cfg.eps_end[i] + (self.eps[i] - self.cfg.eps_end[i]) * self.cfg.eps_step[i]
                self.episode += 1
            self.state.copy_(price)
            torch.cuda.synchronize(self.device)
            out = dict(game_reward_log=rlog.cpu().numpy(), game_action_log=alog.cpu().numpy(), kernel="unfused")
        out["reward_log"] = out["game_reward_log"].mean(axis=2)
        out["action_log"] = out["game_action_log"].mean(axis=2)
        return out
