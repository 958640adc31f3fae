"""Seed-sharded multi-GPU training of one config on one node.

    python -m th_rl_amd.launch --config cfg.json --out runs/exp --gpus 8

One process per GPU (torch.multiprocessing spawn).  Games are independent, so rank r simply trains
the contiguous block of global game ids `sharding.shard_range(n_games, r, world)` with the matching
`game_offset`; the Philox streams are keyed by the global game id, so every game's result is the
same as in a single-GPU run of all games.  There is NO data-path collective: the only cross-rank
step is the weighted mean of the [epochs, N] logs over a gloo (CPU) group.  Outputs: rank 0 writes
the reference's artefacts for global game 0 and the merged log.csv into --out; every rank writes its
shard checkpoint `--out/shard<r>/batch.pt`.
"""
import argparse
import json
import os
import socket

import numpy
import pandas


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def shard_training(config, rank, world):
    """The `training` block rank `rank` of `world` runs: its contiguous block of global game ids, with
    everything that must not depend on the shard size pinned to what the unsharded run would use --
    the table dtype (train_one's default is float64 for ONE game, float32 otherwise) and the Philox
    initialisation keyed by (seed, global game id) (train_one's one-game default draws the tables
    from numpy's global RNG instead).  Pure host logic (no GPU): tests/test_host_cpu.py."""
    from th_rl_amd.sharding import shard_range
    training = dict(config.get("training", {}))
    total = int(training.get("n_games", world))
    if training.get("seed") is None:
        raise SystemExit("th_rl_amd.launch needs an explicit training.seed (all shards must share it)")
    offset, n_local = shard_range(total, rank, world)
    if n_local < 1:
        raise ValueError("rank %d of %d has no games (n_games=%d): launch() clamps the world first" % (rank, world, total))
    training.setdefault("dtype", None)
    if not training["dtype"]:
        training["dtype"] = "float64" if total == 1 else "float32"
    training.update(n_games=n_local, game_offset=int(training.get("game_offset", 0)) + offset,
                    philox_init=(total > 1) or bool(training.get("philox_init", False)), checkpoint=True)
    sweep = training.get("sweep")
    if sweep:        # slice the per-game arrays to this shard
        def cut(v):
            a = numpy.asarray(v)
            return a[..., offset:offset + n_local].tolist()
        training["sweep"] = {k: cut(v) for k, v in sweep.items()}
    return training, offset, n_local


def effective_world(config, gpus):
    """Never more ranks than games: an empty shard has nothing to run."""
    total = int(config.get("training", {}).get("n_games", gpus))
    return max(1, min(int(gpus), total))


def _worker(rank, world, port, config, out, devices_available):
    import torch
    import torch.distributed as dist
    from th_rl_amd import trainer
    from th_rl_amd.sharding import aggregate_logs
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    training, offset, n_local = shard_training(config, rank, world)
    training["device"] = "cuda:%d" % (rank % max(1, devices_available))
    shard_cfg = dict(config, training=training)
    shard_dir = os.path.join(out, "shard%d" % rank)
    os.makedirs(shard_dir, exist_ok=True)
    cpath = os.path.join(shard_dir, "shard_config.json")
    with open(cpath, "w") as f:
        json.dump(shard_cfg, f, indent=3)
    trainer.train_one(shard_dir, cpath)
    log = pandas.read_csv(os.path.join(shard_dir, "log.csv"), header=[0, 1], float_precision="round_trip")
    merged = aggregate_logs(log.to_numpy(dtype="float64"), n_local)
    dist.barrier()
    if rank == 0:
        n = len(config["agents"])
        for i in range(n):      # global game 0 lives in shard 0
            for suffix in (".npy", "_counter.npy", ""):
                src = os.path.join(shard_dir, str(i) + suffix)
                if os.path.exists(src) and os.path.isfile(src):
                    with open(src, "rb") as fi, open(os.path.join(out, str(i) + suffix), "wb") as fo:
                        fo.write(fi.read())
        with open(os.path.join(out, "config.json"), "w") as f:
            json.dump(config, f, indent=3)
        rpd = pandas.DataFrame(data=merged[:, :n], columns=numpy.arange(n))
        apd = pandas.DataFrame(data=merged[:, n:], columns=numpy.arange(n))
        pandas.concat([rpd, apd], axis=1, keys=["rewards", "actions"]).to_csv(os.path.join(out, "log.csv"), index=None)
    dist.destroy_process_group()


def launch(configpath, out, gpus=None):
    """Train `configpath` with its games sharded over `gpus` processes (default: all visible GPUs)."""
    import torch
    import torch.multiprocessing as mp
    config = json.load(open(configpath))
    avail = torch.cuda.device_count()          # device_count() does not initialise the GPU
    if avail < 1:
        from th_rl_amd._lib import ThrlError
        raise ThrlError("th_rl_amd.launch: no GPU visible; there is no CPU fallback")
    world = effective_world(config, int(gpus or avail))     # a shard needs at least one game
    os.makedirs(out, exist_ok=True)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, config, out, avail), nprocs=world, join=True)
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--config", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--gpus", type=int, default=None, help="processes to launch (default: visible GPUs)")
    a = ap.parse_args()
    launch(a.config, a.out, a.gpus)


if __name__ == "__main__":
    main()
