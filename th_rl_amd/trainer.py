"""create_game / train_one -- the reference's trainer entry points (th_rl/trainer.py)
with the same signatures, config schema and output files, driving the GPU.

train_one runs the whole episode loop on the device through GameBatch (fused HIP
kernels).  The JSON schema is the reference's; the optional extra keys in the
"training" block select the batched mode and default to reference behaviour:

    "training": {"epochs": .., "print_freq": ..,
                 "n_games": 1,        # games trained in lockstep (independent replicas)
                 "seed": null,        # Philox seed (null: fresh OS entropy; numpy's global stream is left exactly
                                      # as the reference's train_one leaves it)
                 "philox_init": false, # true: tables / initial state from Philox keyed by (seed, global game id)
                                      # even for ONE game (what a 1-game shard of a sharded run needs)
                 "dtype": null,       # "float64" | "float32" (default f64 for 1 game, f32 otherwise)
                 "device": "cuda:0", "game_offset": 0, "kernel": "auto",
                 "resume": null,     # path of a batch.pt written by an earlier run: continue it
                 "sweep": null}      # per-game hyper-parameters, e.g. {"gamma": [0.35, 0.95, ...]}: arrays of
                                     # length n_games (or [agent][game]) for gamma / alpha / eps / eps_end /
                                     # eps_step / noise_prob (+ entropy for neural agents) -- a config sweep
                                     # (main.py:13-21: one process per config and run) as ONE batched run,
                                     # for all-QTable games and for games with neural agents alike

n_games == 1: tables come from the constructed agents (numpy's global RNG, exactly where
the reference draws them) and the run is float64.  n_games > 1: every game's tables and
initial state come from Philox keyed by (seed, global game id).  Output files are the
reference's four artefacts for game 0 (`<i>.npy`, `<i>_counter.npy`, `config.json`,
`log.csv` -- the log is the MEAN over games), plus `batch.pt` with all games when
n_games > 1.  There is no CPU fallback: without the HIP library or a GPU this raises.
"""
import json
import os
import time

import numpy
import pandas

from th_rl_amd.environments import *  # noqa: F401,F403  (class names are eval'd, as in the reference)
from th_rl_amd.agents import *        # noqa: F401,F403
from th_rl_amd import _lib
from th_rl_amd.batched import GameBatch


def create_game(configpath):
    """JSON -> (config, agents, environment), constructed by name like the reference."""
    config = json.load(open(configpath))
    agents = [eval(agent["name"])(**agent) for agent in config["agents"]]
    assert (
        len(agents) == config["environment"]["nplayers"]
    ), "Bad config. Check number of agents."
    environment = eval(config["environment"]["name"])(**config["environment"])
    return config, agents, environment


def _progress_line(print_eps, eps, elapsed, e, rew, act, names):
    head = ""
    if print_eps:
        head = "eps:{} | ".format(numpy.round(numpy.array(eps) * 1000) / 1000)
    return head + "time:{:2.2f} | episode:{:3d} | reward:{} | agents:{} | actions:{}".format(
        elapsed, e, numpy.round(100 * rew) / 100, ",".join(names), numpy.round(100 * act) / 100)


def resume_is_gamebatch(path):
    """True when `path` is a checkpoint written by GameBatch (so the continued run must use it too)."""
    if not path:
        return False
    import torch
    return torch.load(path, weights_only=True).get("kind") != "mixed"


def _eps_of_game0(batch):
    """Epsilon per agent as game 0 has it: with a per-game epsilon sweep the schedule lives on the device
    (batch.sweep['eps'][agent, game], decayed in the kernels); batch.eps is then only the config's scalar schedule,
    which game 0 never had.  The saved artefacts and the progress line are game 0's (trainer.py:79,101-110)."""
    eps = list(batch.eps)
    sw = getattr(batch, "sweep", None) or {}
    if "eps" in sw:
        col = sw["eps"][:, 0].cpu().numpy()
        for i in range(min(len(eps), len(col))):
            eps[i] = float(col[i])
    return eps


def train_one(exp_path, configpath, loadonly=False, print_eps=False):
    if not os.path.exists(exp_path):
        os.mkdir(os.path.join(exp_path))

    config, agents, environment = create_game(configpath)
    if not all(isinstance(a, (QTable, Reinforce, CAC)) for a in agents) or not isinstance(environment, NoisyPriceState):
        raise NotImplementedError(                      # (ActorCritic is a Reinforce subclass here)
            "train_one: the device path trains QTable, Reinforce, ActorCritic and CAC agents on NoisyPriceState")
    all_tabular = all(isinstance(a, QTable) for a in agents)

    training = config.get("training", {})
    epochs = training.get("epochs", 0)
    print_freq = training.get("print_freq", 500)
    n_games = int(training.get("n_games", 1))
    seed = training.get("seed", None)
    if seed is None:
        # independent of numpy's global stream: under numpy.random.seed(s) the constructors above and
        # environment.reset() below then draw exactly the values the reference's train_one draws
        seed = int(numpy.random.SeedSequence().entropy % (2 ** 31 - 1))
    dtype = training.get("dtype", None) or ("float64" if n_games == 1 else "float32")
    names = [a["name"] for a in config["agents"]]

    resume = training.get("resume", None)
    # Small all-QTable batches in float64 (the default for ONE game = the reference's own use), and those
    # the LDS-resident wave kernel cannot take (other than 2 agents, per-agent grids, T < min_memory), run one
    # wavefront per game through the mixed-agent episode kernel: it keeps train_one's per-step log arithmetic
    # (rewards_log += reward / max_steps, trainer.py:65), so a single game's log.csv is the reference's to the
    # last bit, and it is 4.7x (1 game) to 1.3x (16,384 games) faster than the one-thread-per-game generic
    # kernel, same bits.  Larger float64 batches use the wave kernel's float64 variant (tables, counters,
    # epsilon, state identical; logs to 1e-12).  Explicit "kernel" / "sweep" keys keep GameBatch.
    small_tabular = False
    if all_tabular and "kernel" not in training and not training.get("sweep") and n_games <= 16384 and not resume_is_gamebatch(resume):
        import ctypes
        cfg_probe, _ = _lib.cfg_from_config(config, n_games, {"float32": 0, "float64": 1}[str(dtype)])
        small_tabular = (str(dtype) == "float64"
                         or _lib.load().thrl_select_kernel(ctypes.byref(cfg_probe), 0) == _lib.KERNEL_GENERIC)
    if all_tabular and not small_tabular:
        batch = GameBatch(config, n_games=n_games, device=training.get("device", "cuda:0"), dtype=dtype,
                          seed=seed, game_offset=int(training.get("game_offset", 0)),
                          kernel=training.get("kernel", "auto"), sweep=training.get("sweep", None))
    else:
        # games with neural agents: fused episode kernel + batched network updates (mixed.py)
        from th_rl_amd.mixed import MixedGameBatch
        batch = MixedGameBatch(config, n_games=n_games, device=training.get("device", "cuda:0"), dtype=dtype,
                               seed=seed, game_offset=int(training.get("game_offset", 0)),
                               sweep=training.get("sweep", None))
    if resume:
        batch.load(resume)                              # tables, counters, state, epsilon, episode index
    elif n_games == 1 and not training.get("philox_init", False):
        state = environment.reset()                     # drawn once, as trainer.py:45
        if all_tabular and not small_tabular:
            batch.set_tables(numpy.concatenate([a.table.ravel() for a in agents])[None, :], [float(state[0])])
        else:
            flat = numpy.zeros((1, batch.stride))
            for i, a in enumerate(agents):
                if isinstance(a, QTable):
                    flat[0, batch.offsets[i]:batch.offsets[i] + a.table.size] = a.table.ravel()
                else:
                    batch.nn[i].set_params(a.flat_params())     # torch's default init, as in the reference
            batch.set_tables(flat, [float(state[0])])
    else:
        batch.init_tables()

    rewards_log = numpy.zeros((epochs, len(agents)))
    actions_log = numpy.zeros((epochs, len(agents)))

    t = time.time()
    done = 0
    chunk = max(1, int(print_freq)) if print_freq else epochs
    while done < epochs:
        n = min(chunk - (done % chunk), epochs - done)
        out = batch.run(n) if isinstance(batch, GameBatch) else batch.run(n, per_game_logs=False)
        rewards_log[done:done + n] = out["reward_log"]
        actions_log[done:done + n] = out["action_log"]
        done += n
        if print_freq and not done % print_freq:
            rew = numpy.mean(rewards_log[done - print_freq:done, :], axis=0)
            act = numpy.mean(actions_log[done - print_freq:done, :], axis=0)
            print(_progress_line(print_eps, _eps_of_game0(batch), time.time() - t, done - 1, rew, act, names))
            t = time.time()

    # Store result: the reference's artefacts, from game 0
    for i, a in enumerate(agents):
        if isinstance(a, QTable):
            a.table = batch.table(0, i)
            a.counter = batch.counter_of(0, i)
            a.epsilon = _eps_of_game0(batch)[i]
        else:
            a.set_flat_params(batch.nn[i].params[0].cpu().numpy())
        a.save(os.path.join(exp_path, str(i)))
    environment.state = numpy.float64(batch.states_numpy()[0])

    with open(os.path.join(exp_path, "config.json"), "w") as f:
        json.dump(config, f, indent=3)

    rpd = pandas.DataFrame(data=rewards_log, columns=numpy.arange(len(agents)))
    apd = pandas.DataFrame(data=actions_log, columns=numpy.arange(len(agents)))
    log = pandas.concat([rpd, apd], axis=1, keys=["rewards", "actions"])
    log.to_csv(os.path.join(exp_path, "log.csv"), index=None)

    if n_games > 1 or resume or training.get("checkpoint", False):
        batch.save(os.path.join(exp_path, "batch.pt"))


# BASELINE.json's north_star names the entry point "trainer.train()"; the reference's is train_one.
train = train_one
