#!/usr/bin/env python3
"""Generate golden fixtures by RUNNING THE REFERENCE (read-only, /root/reference).

Only runs in the build container (the reference cannot travel to the GPU box);
only the resulting .npz DATA files are committed -- no reference source, no
bytecode.  Usage:   python tests/golden/make_golden.py

What is recorded (fixtures G1-G6):
  * every draw the hot path consumes: stdlib `random.uniform` / `random.choice`
    inside `QTable.sample_action` (th_rl/agents.py:80-89) and the
    `numpy.random.uniform` draws of `NoisyPriceState.step`
    (th_rl/environments.py:25-39), in call order;
  * the seeded initial tables (`QTable.__init__`, agents.py:29) and the state
    returned by `reset()` (trainer.py:45);
  * per (episode, step): actions, env state (price), rewards;
  * per episode: epsilon after `train_net` (agents.py:78), the trainer's
    rewards_log / actions_log rows (trainer.py:65-66, read back from log.csv);
  * final tables and counters (`<i>.npy`, `<i>_counter.npy`, agents.py:110-112).

The reference itself never seeds anything; this harness seeds numpy / random /
torch so the run is reproducible, then calls the reference's own `train_one`.
"""
import io
import json
import os
import random
import sys
import tempfile

import numpy

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import torch  # noqa: E402
import pandas  # noqa: E402
import th_rl.trainer as ref_trainer  # noqa: E402
import th_rl.agents as ref_agents  # noqa: E402
import th_rl.environments as ref_env  # noqa: E402
import th_rl.buffers as ref_buffers  # noqa: E402

CFG_AGENT = dict(name="QTable", gamma=0.95, actions=21, states=100, alpha=0.1,
                 eps_end=0.001, epsilon=0.5, eps_step=0.9995,
                 action_range=[0.2, 0.4])
CFG_ENV = dict(name="NoisyPriceState", noise_prob=0, a=10, b=1, nplayers=2,
               max_steps=100)


def make_config(agents, env, epochs):
    return {"agents": agents, "environment": env,
            "training": {"print_freq": 10 ** 9, "epochs": epochs}}


class Recorder:
    """Wraps the RNG entry points and the env/agent methods of the reference."""

    def __init__(self):
        self.u = []          # one per sample_action call
        self.choice = []     # aligned with u; -1 when the agent was greedy
        self.noise_u = []    # env.step: uniform(0,1) draw (always consumed)
        self.noise_a = []    # env.step: uniform(0.7a, a) draw or NaN
        self.acts = []       # per step: scaled actions handed to env.step
        self.states = []     # per step: price returned
        self.rewards = []    # per step
        self.eps = []        # per train_net call: epsilon after
        self.in_step = False

    def install(self):
        rec = self
        self._orig = dict(
            uniform=random.uniform, choice=random.choice,
            np_uniform=numpy.random.uniform,
            step=ref_env.NoisyPriceState.step,
            train_net=ref_agents.QTable.train_net,
        )

        def uniform(a, b):
            v = rec._orig["uniform"](a, b)
            rec.u.append(v)
            rec.choice.append(-1)
            return v

        def choice(seq):
            v = rec._orig["choice"](seq)
            rec.choice[-1] = int(v)
            return v

        def np_uniform(lo=0.0, hi=1.0, size=None):
            v = rec._orig["np_uniform"](lo, hi, size)
            if rec.in_step:
                if lo == 0 and hi == 1:
                    rec.noise_u.append(float(v))
                    rec.noise_a.append(float("nan"))
                else:
                    rec.noise_a[-1] = float(v)
            return v

        def step(env, actions):
            rec.in_step = True
            try:
                out = rec._orig["step"](env, actions)
            finally:
                rec.in_step = False
            rec.acts.append([float(a) for a in actions])
            rec.states.append(float(out[0][0]))
            rec.rewards.append([float(r) for r in out[1]])
            return out

        def train_net(agent):
            rec._orig["train_net"](agent)
            rec.eps.append(float(agent.epsilon))

        random.uniform = uniform
        random.choice = choice
        numpy.random.uniform = np_uniform
        ref_env.NoisyPriceState.step = step
        ref_agents.QTable.train_net = train_net

    def uninstall(self):
        random.uniform = self._orig["uniform"]
        random.choice = self._orig["choice"]
        numpy.random.uniform = self._orig["np_uniform"]
        ref_env.NoisyPriceState.step = self._orig["step"]
        ref_agents.QTable.train_net = self._orig["train_net"]


def run_reference(config, seed):
    """Seed, run the reference's train_one, return a dict of arrays."""
    n = len(config["agents"])
    epochs = config["training"]["epochs"]
    T = config["environment"]["max_steps"]
    with tempfile.TemporaryDirectory() as tmp:
        cpath = os.path.join(tmp, "cfg.json")
        with open(cpath, "w") as f:
            json.dump(config, f)
        exp = os.path.join(tmp, "run")

        numpy.random.seed(seed)
        random.seed(seed)
        torch.manual_seed(seed)

        # capture the initial tables / state by wrapping create_game once
        captured = {}
        orig_create = ref_trainer.create_game

        def create_game(path):
            cfg, agents, env = orig_create(path)
            captured["tables"] = [a.table.copy() for a in agents]
            captured["agents"] = agents
            captured["env"] = env
            orig_reset = env.reset

            def reset():
                s = orig_reset()
                captured["state0"] = float(s[0])
                return s
            env.reset = reset
            return cfg, agents, env

        ref_trainer.create_game = create_game
        rec = Recorder()
        rec.install()
        try:
            ref_trainer.train_one(exp, cpath)
        finally:
            rec.uninstall()
            ref_trainer.create_game = orig_create

        final_tables = [numpy.load(os.path.join(exp, "%d.npy" % i)) for i in range(n)]
        final_counters = [numpy.load(os.path.join(exp, "%d_counter.npy" % i)) for i in range(n)]
        log = pandas.read_csv(os.path.join(exp, "log.csv"), header=[0, 1], float_precision="round_trip")
        log_csv_head = open(os.path.join(exp, "log.csv")).read().splitlines()[:2]
        rewards_log = log["rewards"].to_numpy(dtype="float64")
        actions_log = log["actions"].to_numpy(dtype="float64")

    out = dict(
        seed=numpy.int64(seed),
        config_json=numpy.array(json.dumps(config)),
        # tables are stored flat (agent-major concat of (S_i+1, A_i) row-major blocks)
        # because agents may have different grids; shapes in table_shapes [N,2]
        table_shapes=numpy.array([t.shape for t in captured["tables"]], dtype="int64"),
        init_tables=numpy.concatenate([t.ravel() for t in captured["tables"]]),
        state0=numpy.float64(captured["state0"]),
        u=numpy.array(rec.u, dtype="float64").reshape(epochs, T, n),
        choice=numpy.array(rec.choice, dtype="int8").reshape(epochs, T, n),
        noise_u=numpy.array(rec.noise_u, dtype="float64").reshape(epochs, T),
        noise_a=numpy.array(rec.noise_a, dtype="float64").reshape(epochs, T),
        scaled_actions=numpy.array(rec.acts, dtype="float64").reshape(epochs, T, n),
        states=numpy.array(rec.states, dtype="float64").reshape(epochs, T),
        rewards=numpy.array(rec.rewards, dtype="float64").reshape(epochs, T, n),
        eps=numpy.array(rec.eps, dtype="float64").reshape(epochs, n),
        rewards_log=rewards_log, actions_log=actions_log,
        final_tables=numpy.concatenate([t.ravel() for t in final_tables]),
        final_counters=numpy.concatenate([t.ravel() for t in final_counters]),
        log_csv_head=numpy.array("\n".join(log_csv_head)),
    )
    return out


def golden_payoff_grid():
    """G1: env payoff for every (k0,k1) on the 21x21 CFG action grid + encode."""
    numpy.random.seed(0)
    agents = [ref_agents.QTable(**CFG_AGENT) for _ in range(2)]
    env = ref_env.NoisyPriceState(**CFG_ENV)
    A = CFG_AGENT["actions"]
    scaled = numpy.zeros((A,), "float64")
    price = numpy.zeros((A, A), "float64")
    rew = numpy.zeros((A, A, 2), "float64")
    enc64 = numpy.zeros((A, A), "int64")
    enc32 = numpy.zeros((A, A), "int64")
    for k in range(A):
        scaled[k] = agents[0].scale(numpy.int64(k))
    for k0 in range(A):
        for k1 in range(A):
            s, r, _ = env.step([agents[0].scale(numpy.int64(k0)),
                                agents[1].scale(numpy.int64(k1))])
            price[k0, k1] = s[0]
            rew[k0, k1] = r
            enc64[k0, k1] = agents[0].encode(numpy.array(s))[0]
            # the trainer casts to float32 before sample_action (trainer.py:53)
            st = torch.from_numpy(s.astype("float32")).numpy()
            enc32[k0, k1] = agents[0].encode(st)[0]
    nash, cartel = env.get_optimal()
    return dict(scaled=scaled, price=price, rewards=rew, enc64=enc64, enc32=enc32,
                optimal=numpy.array([nash, cartel]))


def golden_encode():
    """G2: encode() vectors, incl. round-half-even ties, f32 and f64 inputs."""
    numpy.random.seed(0)
    ag = ref_agents.QTable(**CFG_AGENT)
    ag16 = ref_agents.QTable(states=16, actions=4, max_state=10)
    x64 = numpy.concatenate([
        numpy.array([0.05, 0.15, 0.25, 0.35, 0.45, 9.95, 9.85, 0.0, 10.0, 5.0]),
        numpy.linspace(2.0, 6.0, 41),
        numpy.random.RandomState(7).uniform(0, 10, 200),
        numpy.arange(0, 33) * (10.0 / 32.0),       # ties for states=16
    ])
    x32 = x64.astype("float32")
    return dict(x64=x64, x32=x32,
                e100_64=ag.encode(x64), e100_32=ag.encode(x32),
                e16_64=ag16.encode(x64), e16_32=ag16.encode(x32))


def golden_td():
    """G3: train_net known answers on hand-built transition lists."""
    out = {}
    cases = {}
    rs = numpy.random.RandomState(11)

    def run_case(name, table0, trans, min_memory, capacity, alpha, gamma,
                 eps=0.5, eps_end=0.001, eps_step=0.9995, states=100, actions=21, repeat=1):
        numpy.random.seed(0)
        ag = ref_agents.QTable(states=states, actions=actions, gamma=gamma, alpha=alpha,
                               eps_end=eps_end, epsilon=eps, eps_step=eps_step,
                               min_memory=min_memory, capacity=capacity)
        ag.table = table0.copy()
        ag.counter = 0 * ag.table
        eps_hist = []
        tabs = []
        for _ in range(repeat):
            for (p, a, r, nd, pn) in trans:
                ag.memory.append(numpy.array([p]), numpy.int64(a), float(r), bool(nd), numpy.array([pn]))
            ag.train_net()
            eps_hist.append(ag.epsilon)
            tabs.append(ag.table.copy())
        cases[name] = dict(
            table0=table0, price=numpy.array([t[0] for t in trans], "float64"),
            action=numpy.array([t[1] for t in trans], "int64"),
            reward=numpy.array([t[2] for t in trans], "float64"),
            next_price=numpy.array([t[4] for t in trans], "float64"),
            params=numpy.array([min_memory, capacity, alpha, gamma, eps, eps_end, eps_step,
                                states, actions, repeat], "float64"),
            table=numpy.stack(tabs), counter=ag.counter.copy(),
            eps=numpy.array(eps_hist), mem_len=numpy.int64(len(ag.memory)))

    # snapshot semantics: two identical transitions from a zero table (SURVEY: 0.1095)
    z = numpy.zeros((101, 21))
    run_case("snapshot_dup", z, [(3.0, 4, 1.0, True, 3.0)] * 2, 2, 500, 0.1, 0.95)
    # live next_max: ns row modified earlier in the same batch
    t0 = 250 + rs.randn(101, 21)
    tr = [(3.0, 4, 1.5, True, 4.0), (4.0, 2, 2.5, True, 3.0), (3.0, 4, 0.5, True, 3.0),
          (3.0, 7, 0.25, True, 4.0), (4.0, 2, 9.0, False, 4.0)]
    run_case("live_next_max", t0, tr, 5, 500, 0.1, 0.95)
    # below min_memory: no update, epsilon still decays; accumulates across calls
    run_case("below_min_memory", t0, tr, 12, 500, 0.3, 0.9, repeat=4)
    # overflow past capacity: deque drops the oldest
    tr_long = [(round(float(rs.uniform(2, 6)), 1), int(rs.randint(0, 21)), float(rs.uniform(5, 15)),
                True, round(float(rs.uniform(2, 6)), 1)) for _ in range(17)]
    run_case("overflow_capacity", t0, tr_long, 30, 40, 0.5, 0.35, repeat=3)
    # capacity < min_memory: never trains
    run_case("never_trains", t0, tr_long, 100, 20, 0.5, 0.35, repeat=2)
    for k, v in cases.items():
        for kk, vv in v.items():
            out["%s__%s" % (k, kk)] = vv
    out["case_names"] = numpy.array(sorted(cases))
    return out


def random_config(k):
    """G6: seeded random configurations (2-4 agents, per-agent grids / table sizes / max_state, ragged
    T, min_memory and capacity, env noise) -- the reference run on shapes nobody picked by hand."""
    rs = numpy.random.RandomState(600 + k)
    n = int(rs.choice([2, 2, 3, 4]))
    T = int(rs.choice([7, 12, 25, 40]))
    agents = []
    for _ in range(n):
        lo = float(numpy.round(rs.uniform(0.0, 0.3), 2))
        agents.append(dict(name="QTable", gamma=float(rs.choice([0.35, 0.9, 0.95])), actions=int(rs.choice([3, 7, 21, 30])),
                           states=int(rs.choice([16, 40, 100])), alpha=float(rs.choice([0.05, 0.1, 0.5])), eps_end=0.02,
                           epsilon=float(rs.choice([0.2, 0.5, 1.0])), eps_step=float(rs.choice([0.9, 0.999])),
                           action_range=[lo, float(numpy.round(lo + rs.uniform(0.05, 0.12), 2))],
                           max_state=float(rs.choice([10, 12])), min_memory=int(rs.choice([1, T, T + 3, 2 * T])),
                           capacity=int(rs.choice([T + 1, 2 * T + 5, 500]))))
    env = dict(name="NoisyPriceState", noise_prob=float(rs.choice([0.0, 0.05, 0.4])), a=10, b=1, nplayers=n, max_steps=T)
    return make_config(agents, env, int(rs.choice([6, 9, 12])))


def main():
    os.makedirs(HERE, exist_ok=True)

    def save(name, d):
        p = os.path.join(HERE, name)
        numpy.savez_compressed(p, **d)
        print("wrote", p, os.path.getsize(p), "bytes")

    if "--only-g6" in sys.argv:
        for k in range(6):
            save("g6_random%d_seed%d.npz" % (k, 60 + k), run_reference(random_config(k), 60 + k))
        return

    save("g1_payoff_grid.npz", golden_payoff_grid())
    save("g2_encode.npz", golden_encode())
    save("g3_td_known_answers.npz", golden_td())

    # G4: full-loop trajectories on CFG
    for seed in (0, 1, 2, 3):
        cfg = make_config([dict(CFG_AGENT), dict(CFG_AGENT)], dict(CFG_ENV), 12)
        save("g4_cfg_seed%d_e12.npz" % seed, run_reference(cfg, seed))
    cfg = make_config([dict(CFG_AGENT), dict(CFG_AGENT)], dict(CFG_ENV), 150)
    save("g4_cfg_seed0_e150.npz", run_reference(cfg, 0))

    # G5: heterogeneous agents (configs2.json QTable params vs example_config), noise on
    a2 = dict(CFG_AGENT, gamma=0.35, alpha=0.5, epsilon=0.8)
    env_noise = dict(CFG_ENV, noise_prob=0.05)
    cfg = make_config([a2, dict(CFG_AGENT)], env_noise, 20)
    save("g5_hetero_noise_seed5_e20.npz", run_reference(cfg, 5))
    env_noise_hi = dict(CFG_ENV, noise_prob=0.5, max_steps=40)
    cfg = make_config([dict(CFG_AGENT, action_range=[0.1, 0.6]), a2], env_noise_hi, 25)
    save("g5_noise50_T40_seed6_e25.npz", run_reference(cfg, 6))

    # buffer corner cases: T < min_memory (accumulate across episodes) with overflow
    a_small = dict(CFG_AGENT, min_memory=100, capacity=80)     # never trains (cap<min)
    a_acc = dict(CFG_AGENT, min_memory=70, capacity=90)        # trains every 3rd episode on 90
    env30 = dict(CFG_ENV, max_steps=30)
    cfg = make_config([a_acc, a_small], env30, 16)
    save("g5_buffer_T30_seed7_e16.npz", run_reference(cfg, 7))
    # T > capacity: only the last `capacity` transitions are replayed
    a_cap = dict(CFG_AGENT, min_memory=20, capacity=64)
    cfg = make_config([a_cap, dict(CFG_AGENT)], dict(CFG_ENV), 10)
    save("g5_capacity64_seed8_e10.npz", run_reference(cfg, 8))
    # different grid sizes per agent + 3 players
    env3 = dict(CFG_ENV, nplayers=3, max_steps=25)
    ags = [dict(CFG_AGENT, actions=11, states=50, action_range=[0.1, 0.3], min_memory=25),
           dict(CFG_AGENT, actions=21, states=100, action_range=[0.15, 0.35], min_memory=25),
           dict(CFG_AGENT, actions=5, states=20, action_range=[0.0, 0.4], min_memory=25, max_state=8)]
    cfg = make_config(ags, env3, 30)
    save("g5_three_players_seed9_e30.npz", run_reference(cfg, 9))
    for k in range(6):
        save("g6_random%d_seed%d.npz" % (k, 60 + k), run_reference(random_config(k), 60 + k))


if __name__ == "__main__":
    main()
