"""ctypes binding of libthrl_hip.so (include/thrl.h).  No CPU fallback: every
compute entry point raises if the library or a GPU is missing."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# THRL_LIB: an alternative build of the SAME ABI (the timing-only ablation variants of profiles/ablate.py)
LIB_PATH = os.environ.get("THRL_LIB") or os.path.join(HERE, "libthrl_hip.so")
MAXA = 8

KERNEL_AUTO, KERNEL_GENERIC, KERNEL_WAVE, KERNEL_WAVE_PLAIN, KERNEL_WAVE_GREEDY, KERNEL_TUPLE = 0, 1, 2, 3, 4, 5
KERNEL_NAMES = {0: "auto", 1: "generic", 2: "wave", 5: "tuple"}
ABI_VERSION = 3


class ThrlError(RuntimeError):
    code = None             # the negative thrl_err when the error came from the library

ERR_UNSUPPORTED = -3


class Cfg(ctypes.Structure):
    """thrl_cfg"""
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


class Buffers(ctypes.Structure):
    """thrl_buffers"""
    _fields_ = [
        ("q", ctypes.c_void_p), ("counter", ctypes.c_void_p), ("state", ctypes.c_void_p),
        ("replay_mem", ctypes.c_void_p), ("replay_mem_bytes", ctypes.c_size_t),
        ("reward_log", ctypes.c_void_p), ("action_log", ctypes.c_void_p),
        ("game_reward_log", ctypes.c_void_p), ("game_action_log", ctypes.c_void_p),
        ("inj_u", ctypes.c_void_p), ("inj_choice", ctypes.c_void_p),
        ("inj_noise_u", ctypes.c_void_p), ("inj_noise_a", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
        ("sweep_gamma", ctypes.c_void_p), ("sweep_alpha", ctypes.c_void_p),
        ("sweep_eps_end", ctypes.c_void_p), ("sweep_eps_step", ctypes.c_void_p),
        ("sweep_eps", ctypes.c_void_p), ("sweep_noise_prob", ctypes.c_void_p),
    ]


class Run(ctypes.Structure):
    """thrl_run"""
    _fields_ = [
        ("seed", ctypes.c_uint64), ("game_offset", ctypes.c_uint64),
        ("first_episode", ctypes.c_uint64), ("n_episodes", ctypes.c_int32),
        ("kernel", ctypes.c_int32), ("eps", ctypes.c_double * MAXA),
        ("mem_count", ctypes.c_int32 * MAXA), ("kernel_used", ctypes.c_int32),
    ]


class Mixed(ctypes.Structure):
    """thrl_mixed"""
    _fields_ = [
        ("kind", ctypes.c_int32 * MAXA), ("nn_params", ctypes.c_void_p * MAXA),
        ("buf_price", ctypes.c_void_p * MAXA), ("buf_action", ctypes.c_void_p * MAXA),
        ("buf_reward", ctypes.c_void_p * MAXA), ("buf_nprice", ctypes.c_void_p * MAXA),
        ("buf_scratch", ctypes.c_void_p * MAXA), ("buf_len", ctypes.c_int32 * MAXA),
        ("min_memory", ctypes.c_int32 * MAXA), ("count", ctypes.c_int32 * MAXA),
        ("sweep_gamma", ctypes.c_void_p), ("sweep_alpha", ctypes.c_void_p),
        ("sweep_eps_end", ctypes.c_void_p), ("sweep_eps_step", ctypes.c_void_p),
        ("sweep_eps", ctypes.c_void_p), ("sweep_noise_prob", ctypes.c_void_p),
        ("policy_tab", ctypes.c_void_p), ("policy_tab_bytes", ctypes.c_size_t), ("flags", ctypes.c_int32),
    ]


# every symbol include/thrl.h declares (tests check the library exports all of them)
SYMBOLS = [
    "thrl_version", "thrl_last_error", "thrl_build_info", "thrl_ablate_mask", "thrl_table_stride", "thrl_table_offset",
    "thrl_replay_mem_bytes", "thrl_workspace_bytes", "thrl_select_kernel", "thrl_training_cycle", "thrl_qtable_init",
    "thrl_qtable_episodes", "thrl_play_greedy", "thrl_op_sample_action", "thrl_op_encode", "thrl_op_scale",
    "thrl_op_env_step", "thrl_op_td_update",
    "thrl_nn_param_count", "thrl_nn_init", "thrl_nn_act", "thrl_nn_reinforce_train", "thrl_op_draws",
    "thrl_mixed_episodes", "thrl_mixed_policy_table_bytes", "thrl_ac_param_count", "thrl_ac_init", "thrl_ac_act", "thrl_ac_train",
    "thrl_cac_init", "thrl_cac_act", "thrl_cac_train",
]
CAC_PARAMS = 1283

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  Device
    pointers and streams are only meaningful inside ONE HIP runtime instance, so torch
    must be imported first and its runtime made globally visible; libthrl_hip.so's
    NEEDED libamdhip64.so.7 then binds to that same instance instead of /opt/rocm's."""
    try:
        import torch
    except ImportError as e:
        raise ThrlError("th_rl_amd needs PyTorch-ROCm (device memory / streams): %s" % e)
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def load():
    """Load libthrl_hip.so; raises ThrlError (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ThrlError(
            "th_rl_amd: %s not found. Build it with `python -m th_rl_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    _share_hip_runtime_with_torch()
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise ThrlError("th_rl_amd: cannot load %s: %s" % (LIB_PATH, e))
    vp, i32, u64, dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64, ctypes.c_double
    cfgp = ctypes.POINTER(Cfg)
    L.thrl_version.restype = ctypes.c_int
    L.thrl_last_error.restype = ctypes.c_char_p
    L.thrl_build_info.restype = ctypes.c_char_p
    L.thrl_ablate_mask.restype = ctypes.c_int
    for n in ("thrl_table_stride", "thrl_replay_mem_bytes", "thrl_workspace_bytes"):
        getattr(L, n).restype = ctypes.c_size_t
        getattr(L, n).argtypes = [cfgp]
    L.thrl_table_offset.restype = ctypes.c_size_t
    L.thrl_table_offset.argtypes = [cfgp, ctypes.c_int]
    L.thrl_select_kernel.restype = ctypes.c_int
    L.thrl_select_kernel.argtypes = [cfgp, ctypes.c_int]
    L.thrl_training_cycle.restype = ctypes.c_int
    L.thrl_training_cycle.argtypes = [cfgp]
    L.thrl_qtable_init.restype = ctypes.c_int
    L.thrl_qtable_init.argtypes = [cfgp, vp, vp, vp, u64, u64, vp, vp]
    L.thrl_qtable_episodes.restype = ctypes.c_int
    L.thrl_qtable_episodes.argtypes = [cfgp, ctypes.POINTER(Buffers), ctypes.POINTER(Run), vp]
    L.thrl_play_greedy.restype = ctypes.c_int
    L.thrl_play_greedy.argtypes = [cfgp, vp, vp, i32, u64, u64, vp, vp, vp]
    L.thrl_op_sample_action.restype = ctypes.c_int
    L.thrl_op_sample_action.argtypes = [cfgp, ctypes.c_int, vp, vp, dbl, vp, vp, ctypes.c_int, vp, vp]
    L.thrl_op_env_step.restype = ctypes.c_int
    L.thrl_op_env_step.argtypes = [cfgp, vp, vp, vp, vp, vp, vp]
    L.thrl_op_encode.restype = ctypes.c_int
    L.thrl_op_encode.argtypes = [cfgp, ctypes.c_int, vp, ctypes.c_int, vp, vp]
    L.thrl_op_scale.restype = ctypes.c_int
    L.thrl_op_scale.argtypes = [cfgp, ctypes.c_int, vp, vp, vp]
    L.thrl_op_td_update.restype = ctypes.c_int
    L.thrl_op_td_update.argtypes = [cfgp, ctypes.c_int, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.thrl_nn_param_count.restype = ctypes.c_size_t
    L.thrl_nn_param_count.argtypes = [ctypes.c_int]
    L.thrl_nn_init.restype = ctypes.c_int
    L.thrl_nn_init.argtypes = [ctypes.c_int, ctypes.c_int, vp, u64, u64, ctypes.c_int, vp]
    L.thrl_nn_act.restype = ctypes.c_int
    L.thrl_nn_act.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp]
    L.thrl_nn_reinforce_train.restype = ctypes.c_int
    L.thrl_nn_reinforce_train.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, i32, i32, i32, vp, vp, vp,
                                          dbl, dbl, dbl, vp, vp, vp, vp, vp]
    L.thrl_op_draws.restype = ctypes.c_int
    L.thrl_op_draws.argtypes = [cfgp, u64, u64, u64, i32, vp, vp, vp, vp, vp, vp]
    L.thrl_cac_init.restype = ctypes.c_int
    L.thrl_cac_init.argtypes = [ctypes.c_int, vp, u64, u64, ctypes.c_int, vp]
    L.thrl_cac_act.restype = ctypes.c_int
    L.thrl_cac_act.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.thrl_cac_train.restype = ctypes.c_int
    L.thrl_cac_train.argtypes = [ctypes.c_int, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp,
                                 ctypes.c_double, ctypes.c_double, ctypes.c_double, vp, vp, vp, vp]
    L.thrl_ac_param_count.restype = ctypes.c_size_t
    L.thrl_ac_param_count.argtypes = [ctypes.c_int]
    L.thrl_ac_init.restype = ctypes.c_int
    L.thrl_ac_init.argtypes = L.thrl_nn_init.argtypes
    L.thrl_ac_act.restype = ctypes.c_int
    L.thrl_ac_act.argtypes = L.thrl_nn_act.argtypes
    L.thrl_ac_train.restype = ctypes.c_int
    L.thrl_ac_train.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp,
                                ctypes.c_double, ctypes.c_double, ctypes.c_double, vp, vp, vp, vp]
    L.thrl_mixed_policy_table_bytes.restype = ctypes.c_size_t
    L.thrl_mixed_policy_table_bytes.argtypes = [cfgp, ctypes.POINTER(Mixed)]
    L.thrl_mixed_episodes.restype = ctypes.c_int
    L.thrl_mixed_episodes.argtypes = [cfgp, ctypes.POINTER(Mixed), vp, vp, vp, ctypes.POINTER(Run), vp, vp, vp]
    if L.thrl_version() != ABI_VERSION:
        raise ThrlError("th_rl_amd: ABI version mismatch (%d)" % L.thrl_version())
    _lib = L
    return L


def build_info():
    """dict(path, abi, ablate, src) of the loaded library (thrl_build_info)."""
    L = load()
    d = dict(kv.split("=", 1) for kv in L.thrl_build_info().decode().split(";"))
    return dict(path=LIB_PATH, abi=int(d["abi"]), ablate=int(d["ablate"]), src=d["src"], wave=d.get("wave"), nn=d.get("nn"))


def check(rc, what):
    if rc != 0:
        msg = load().thrl_last_error().decode("utf-8", "replace")
        err = ThrlError("%s failed (thrl_err %d): %s" % (what, rc, msg))
        err.code = rc
        raise err


# QTable.__init__ defaults (reference th_rl/agents.py:13-27) / NoisyPriceState (environments.py:5)
QTABLE_DEFAULTS = dict(states=16, actions=4, action_range=[0, 1], gamma=0.99, capacity=500,
                       max_state=10, alpha=0.1, eps_end=2e-2, epsilon=0.5, eps_step=5e-4,
                       min_memory=100)
ENV_DEFAULTS = dict(action_range=[0, 1], a=10, b=1, max_steps=1, noise_prob=0.05)


def cfg_from_config(config, n_games, q_dtype):
    """(Cfg, [epsilon_i]) from a reference-schema config dict whose agents are all QTable."""
    agents = config["agents"]
    if len(agents) > MAXA:
        raise ThrlError("at most %d agents per game" % MAXA)
    env = dict(ENV_DEFAULTS, **config["environment"])
    c = Cfg()
    c.n_games = int(n_games)
    c.n_agents = len(agents)
    c.max_steps = int(env["max_steps"])
    c.q_dtype = int(q_dtype)
    c.env_a = float(env["a"])
    c.env_b = float(env["b"])
    c.noise_prob = float(env["noise_prob"])
    eps = []
    for i, a in enumerate(agents):
        if a.get("name", "QTable") != "QTable":
            raise ThrlError("the batched device path handles QTable agents only, got %r" % a.get("name"))
        p = dict(QTABLE_DEFAULTS, **a)
        c.n_states[i] = int(p["states"]); c.n_actions[i] = int(p["actions"])
        c.min_memory[i] = int(p["min_memory"]); c.capacity[i] = int(p["capacity"])
        c.max_state[i] = float(p["max_state"]); c.gamma[i] = float(p["gamma"])
        c.alpha[i] = float(p["alpha"]); c.eps_end[i] = float(p["eps_end"])
        c.eps_step[i] = float(p["eps_step"])
        c.act_lo[i] = float(p["action_range"][0]); c.act_hi[i] = float(p["action_range"][1])
        eps.append(float(p["epsilon"]))
    return c, eps
