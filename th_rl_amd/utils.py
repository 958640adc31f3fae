"""Evaluation helpers on the training path's edge (reference th_rl/utils.py:12-47).
The plotting functions of the reference are out of scope."""
import os

import numpy
import pandas

from th_rl_amd.trainer import create_game


def load_experiment(loc):
    """(config, agents, environment, actions, rewards) from a run directory, as utils.py:12-24."""
    config, agents, environment = create_game(os.path.join(loc, "config.json"))
    for i, agent in enumerate(agents):
        agent.load(os.path.join(loc, str(i)))
    log = pandas.read_csv(os.path.join(loc, "log.csv"))
    names = [a["name"] + str(i) for i, a in enumerate(config["agents"])]
    rcols = [c for c in log.columns if c.startswith("rewards")]
    acols = [c for c in log.columns if c.startswith("actions")]
    rewards = log[rcols].ewm(halflife=1000).mean()
    actions = log[acols].ewm(halflife=1000).mean()
    rewards.columns = names
    actions.columns = names
    return config, agents, environment, actions, rewards


def play_game(agents, environment, iters=1):
    """Greedy rollout through the object protocol (each call runs a device operator)."""
    rewards, actions = [], []
    for _ in range(iters):
        done = False
        next_state = environment.reset()
        while not done:
            acts = [agent.get_action(next_state) for agent in agents]
            scaled_acts = [agent.scale(act) for agent, act in zip(agents, acts)]
            next_state, reward, done = environment.step(scaled_acts)
            rewards.append(reward)
            actions.append(scaled_acts)
    return numpy.array(actions), numpy.array(rewards)


def play_game_batched(batch, iters=1, state0=None):
    """The same rollout for every game of a GameBatch in one kernel (thrl_play_greedy):
    per-iteration mean reward / mean scaled action, arrays [iters, N, G]."""
    return batch.play_greedy(iters=iters, state0=state0)
