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


def _worker(rank, world, port, config, out, devices_available):
    import torch
    import torch.distributed as dist
    from th_rl_amd import trainer
    from th_rl_amd.sharding import aggregate_logs, shard_range
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    training = dict(config.get("training", {}))
    total = int(training.get("n_games", world))
    offset, n_local = shard_range(total, rank, world)
    training.update(n_games=n_local, game_offset=int(training.get("game_offset", 0)) + offset,
                    device="cuda:%d" % (rank % max(1, devices_available)), checkpoint=True)
    sweep = training.get("sweep")
    if sweep:        # slice the per-game arrays to this shard
        def cut(v):
            a = numpy.asarray(v)
            return a[..., offset:offset + n_local].tolist()
        training["sweep"] = {k: cut(v) for k, v in sweep.items()}
    if training.get("seed") is None:
        raise SystemExit("th_rl_amd.launch needs an explicit training.seed (all shards must share it)")
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
    world = int(gpus or avail)
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
