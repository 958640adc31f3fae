"""ctypes loader + numpy-facing wrappers for oracle/thrl_oracle.c.

TEST INFRASTRUCTURE ONLY (see thrl_oracle.c header): the checker for the HIP
path, pinned bit-for-bit against fixtures generated from the reference
(tests/golden/make_golden.py).  Parity status: PINNED.
"""
import ctypes
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# THRL_ORACLE_SANITIZE=1: load the ASan + UBSan build instead (the process must have libasan preloaded;
# tests/test_host_cpu.py::test_oracle_golden_under_sanitizers does that in a child process)
SANITIZE = os.environ.get("THRL_ORACLE_SANITIZE") == "1"
LIB_NAME = "libthrl_oracle_san.so" if SANITIZE else "libthrl_oracle.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
MAXA = 8


class Cfg(ctypes.Structure):
    """Mirror of thrl_cfg (include/thrl.h)."""
    _fields_ = [
        ("n_games", ctypes.c_int32), ("n_agents", ctypes.c_int32),
        ("max_steps", ctypes.c_int32), ("q_dtype", ctypes.c_int32),
        ("env_a", ctypes.c_double), ("env_b", ctypes.c_double),
        ("noise_prob", ctypes.c_double),
        ("n_states", ctypes.c_int32 * MAXA), ("n_actions", ctypes.c_int32 * MAXA),
        ("min_memory", ctypes.c_int32 * MAXA), ("capacity", ctypes.c_int32 * MAXA),
        ("max_state", ctypes.c_double * MAXA), ("gamma", ctypes.c_double * MAXA),
        ("alpha", ctypes.c_double * MAXA), ("eps_end", ctypes.c_double * MAXA),
        ("eps_step", ctypes.c_double * MAXA), ("act_lo", ctypes.c_double * MAXA),
        ("act_hi", ctypes.c_double * MAXA),
    ]


# QTable.__init__ defaults (th_rl/agents.py:13-27) and NoisyPriceState (environments.py:5)
QTABLE_DEFAULTS = dict(states=16, actions=4, action_range=[0, 1], gamma=0.99, capacity=500,
                       max_state=10, alpha=0.1, eps_end=2e-2, epsilon=0.5, eps_step=5e-4,
                       min_memory=100)
ENV_DEFAULTS = dict(action_range=[0, 1], a=10, b=1, max_steps=1, noise_prob=0.05)


def cfg_from_config(config, n_games=1, q_dtype=1):
    """Build (Cfg, eps0) from a reference-schema config dict (all-QTable)."""
    c = Cfg()
    agents = config["agents"]
    env = dict(ENV_DEFAULTS, **config["environment"])
    c.n_games = n_games
    c.n_agents = len(agents)
    c.max_steps = int(env["max_steps"])
    c.q_dtype = q_dtype
    c.env_a = float(env["a"])
    c.env_b = float(env["b"])
    c.noise_prob = float(env["noise_prob"])
    eps0 = np.zeros(MAXA, dtype=np.float64)
    for i, a in enumerate(agents):
        p = dict(QTABLE_DEFAULTS, **a)
        c.n_states[i] = int(p["states"]); c.n_actions[i] = int(p["actions"])
        c.min_memory[i] = int(p["min_memory"]); c.capacity[i] = int(p["capacity"])
        c.max_state[i] = float(p["max_state"]); c.gamma[i] = float(p["gamma"])
        c.alpha[i] = float(p["alpha"]); c.eps_end[i] = float(p["eps_end"])
        c.eps_step[i] = float(p["eps_step"])
        c.act_lo[i] = float(p["action_range"][0]); c.act_hi[i] = float(p["action_range"][1])
        eps0[i] = float(p["epsilon"])
    return c, eps0


def build(force=False):
    src = os.path.join(HERE, "thrl_oracle.c")
    hdr = os.path.join(HERE, "..", "include", "thrl.h")
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return LIB_PATH
    subprocess.check_call(["make", "-s", "-C", HERE, "-B", LIB_NAME])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        L.oracle_table_stride.restype = ctypes.c_size_t
        L.oracle_table_stride.argtypes = [ctypes.POINTER(Cfg)]
        L.oracle_table_offset.restype = ctypes.c_size_t
        L.oracle_table_offset.argtypes = [ctypes.POINTER(Cfg), ctypes.c_int]
        L.oracle_encode64.restype = ctypes.c_int64
        L.oracle_encode64.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int]
        L.oracle_encode32.restype = ctypes.c_int64
        L.oracle_encode32.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int]
        L.oracle_scale.restype = ctypes.c_double
        L.oracle_scale.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
        L.oracle_eps_decay.restype = ctypes.c_double
        L.oracle_eps_decay.argtypes = [ctypes.c_double] * 3
        L.oracle_episodes.restype = ctypes.c_int
        L.oracle_play_greedy.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a, ct=None):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def table_stride(cfg):
    return int(lib().oracle_table_stride(ctypes.byref(cfg)))


def table_offset(cfg, i):
    return int(lib().oracle_table_offset(ctypes.byref(cfg), i))


def philox(ctr, key):
    c = (ctypes.c_uint32 * 4)(*ctr); k = (ctypes.c_uint32 * 2)(*key); o = (ctypes.c_uint32 * 4)()
    lib().oracle_philox4x32_10(c, k, o)
    return list(o)


def encode64(price, max_state, states):
    return int(lib().oracle_encode64(float(price), float(max_state), int(states)))


def encode32(price, max_state, states):
    return int(lib().oracle_encode32(float(price), float(max_state), int(states)))


def scale(action, n_actions, lo, hi):
    return float(lib().oracle_scale(int(action), int(n_actions), float(lo), float(hi)))


def env_step(cfg, scaled, noisy=False, new_a=0.0):
    s = np.ascontiguousarray(scaled, dtype=np.float64)
    price = ctypes.c_double()
    rew = np.zeros(cfg.n_agents, dtype=np.float64)
    lib().oracle_env_step(ctypes.byref(cfg), _p(s), ctypes.c_int(int(noisy)), ctypes.c_double(new_a),
                          ctypes.byref(price), _p(rew))
    return price.value, rew


def get_optimal(cfg):
    n = ctypes.c_double(); c = ctypes.c_double()
    lib().oracle_get_optimal(ctypes.byref(cfg), ctypes.byref(n), ctypes.byref(c))
    return n.value, c.value


def td_update(table, counter, st, ac, rw, ns, alpha, gamma):
    """In-place train_net table update on one (rows, A) table (f32 or f64)."""
    A = table.shape[1]
    st = np.ascontiguousarray(st, np.int32); ac = np.ascontiguousarray(ac, np.int32)
    ns = np.ascontiguousarray(ns, np.int32); rw = np.ascontiguousarray(rw, np.float64)
    fn = lib().oracle_td_update_f64 if table.dtype == np.float64 else lib().oracle_td_update_f32
    assert table.dtype in (np.float64, np.float32)
    fn(_p(table), _p(counter), ctypes.c_int(A), ctypes.c_int(len(st)), _p(st), _p(ac), _p(rw), _p(ns),
       ctypes.c_double(alpha), ctypes.c_double(gamma))


def init(cfg, seed=0, game_offset=0):
    G = cfg.n_games
    stride = table_stride(cfg)
    q = np.zeros((G, stride), dtype=np.float64 if cfg.q_dtype == 1 else np.float32)
    counter = np.zeros((G, stride), dtype=np.int32)
    state = np.zeros(G, dtype=np.float64)
    lib().oracle_init(ctypes.byref(cfg), _p(q), _p(counter), _p(state),
                      ctypes.c_uint64(seed), ctypes.c_uint64(game_offset))
    return q, counter, state


class Memory:
    """The ReplayBuffer contents that survive between calls."""

    def __init__(self, cfg):
        self.capmax = max(1, max(cfg.capacity[i] for i in range(cfg.n_agents)))
        shp = (cfg.n_games, cfg.n_agents, self.capmax)
        self.s = np.zeros(shp, np.int32); self.a = np.zeros(shp, np.int32)
        self.ns = np.zeros(shp, np.int32); self.r = np.zeros(shp, np.float64)
        self.count = np.zeros(MAXA, np.int32)


def episodes(cfg, q, counter, state, eps, mem, n_episodes, seed=0, game_offset=0, first_episode=0,
             inj_u=None, inj_choice=None, inj_noise_u=None, inj_noise_a=None, trace=False, sweep=None):
    """Run n_episodes of all games in place.  Returns dict of logs."""
    G, N, T = cfg.n_games, cfg.n_agents, cfg.max_steps
    g_r = np.zeros((n_episodes, N, G)); g_a = np.zeros((n_episodes, N, G))
    m_r = np.zeros((n_episodes, N)); m_a = np.zeros((n_episodes, N))
    tr_a = np.zeros((n_episodes, T, N, G), np.int32) if trace else None
    tr_p = np.zeros((n_episodes, T, G), np.float64) if trace else None
    for a, dt in ((inj_u, np.float64), (inj_choice, np.int8), (inj_noise_u, np.float64),
                  (inj_noise_a, np.float64)):
        assert a is None or (a.dtype == dt and a.flags["C_CONTIGUOUS"])
    if inj_u is not None:
        assert inj_u.shape == (n_episodes, T, N, G) and inj_choice.shape == inj_u.shape
        if cfg.noise_prob > 0:
            assert inj_noise_u.shape == (n_episodes, T, G) and inj_noise_a.shape == (n_episodes, T, G)
    rc = lib().oracle_episodes(
        ctypes.byref(cfg), _p(q), _p(counter), _p(state), _p(eps), _p(mem.count),
        _p(mem.s), _p(mem.a), _p(mem.ns), _p(mem.r), ctypes.c_int32(mem.capmax),
        ctypes.c_uint64(seed), ctypes.c_uint64(game_offset), ctypes.c_uint64(first_episode),
        ctypes.c_int32(n_episodes), _p(inj_u), _p(inj_choice), _p(inj_noise_u), _p(inj_noise_a),
        _p(g_r), _p(g_a), _p(m_r), _p(m_a), _p(tr_a), _p(tr_p),
        *[_p(None if sweep is None else sweep.get(k)) for k in
          ("gamma", "alpha", "eps_end", "eps_step", "eps", "noise_prob")])
    if rc != 0:
        raise RuntimeError("oracle_episodes failed rc=%d" % rc)
    return dict(game_reward_log=g_r, game_action_log=g_a, reward_log=m_r, action_log=m_a,
                trace_actions=tr_a, trace_price=tr_p)


def play_greedy(cfg, q, iters, state0=None, seed=0, game_offset=0):
    G, N = cfg.n_games, cfg.n_agents
    mr = np.zeros((iters, N, G)); ma = np.zeros((iters, N, G))
    rc = lib().oracle_play_greedy(ctypes.byref(cfg), _p(q), _p(state0), ctypes.c_int32(iters),
                                  ctypes.c_uint64(seed), ctypes.c_uint64(game_offset), _p(mr), _p(ma))
    if rc != 0:
        raise RuntimeError("oracle_play_greedy failed rc=%d" % rc)
    return mr, ma


def load_golden_config(npz):
    return json.loads(str(npz["config_json"]))
