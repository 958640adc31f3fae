"""CLI runner, same flags and directory convention as the reference's th_rl/main.py:
for every *.json in --dir, run --runs trainings into <dir>/../runs/<config>/<i>."""
import os

import click

from th_rl_amd.trainer import train_one


@click.command()
@click.option("--runs", default=1, help="Runs per config", type=int)
@click.option("--dir", default="configs", help="Configs dir", type=str)
def main(**params):
    home = os.path.join(os.path.abspath(params["dir"]), "..", "runs")
    if not os.path.exists(home):
        os.mkdir(home)
    for confname in sorted(os.listdir(params["dir"])):
        if ".json" not in confname:
            continue
        stem = confname.replace(".json", "")
        if stem in os.listdir(home):
            print("Skipping {}".format(confname))
            continue
        cpath = os.path.join(home, stem)
        os.mkdir(cpath)
        for i in range(params["runs"]):
            train_one(os.path.join(cpath, str(i)), os.path.join(params["dir"], confname))


if __name__ == "__main__":
    main()
