"""Plain vs GREEDY code variant of the wave kernel on TRAINED tables along a real run (epsilon decays 0.5 -> 0.001): where is the
crossover?  Both variants give identical results; the variant is pinned per call (THRL_KERNEL_WAVE_PLAIN / _GREEDY).
    python profiles/exp_greedy_threshold.py [games]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from th_rl_amd.batched import GameBatch
from th_rl_amd import _lib
import bench
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
gb = GameBatch(bench.CFG, n_games=G, dtype="float32", kernel="wave", seed=0).init_tables()
done = 0
def advance(n):
    global done
    k = 0
    while k < n:
        e = min(32, n - k); gb.run(e, sync=False, logs=False); k += e
    done += n
    torch.cuda.synchronize()
def timed(kern, n=25, reps=3):
    gb.kernel = kern
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gb.run(n, sync=False, logs=False)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    global done
    done += n * reps
    gb.kernel = _lib.KERNEL_WAVE
    return G * 100 * n / sorted(ts)[len(ts) // 2]
for target in (1000, 2000, 3000, 4000, 4500, 5000, 5500, 6000, 7000, 9000):
    advance(target - done)
    eps = gb.eps[0]
    p = timed(_lib.KERNEL_WAVE_PLAIN); g = timed(_lib.KERNEL_WAVE_GREEDY); p2 = timed(_lib.KERNEL_WAVE_PLAIN)
    print("episode %5d eps %.4f  plain %.3e / %.3e  greedy %.3e  -> %s" % (target, eps, p, p2, g, "GREEDY" if g > max(p, p2) else "plain"), flush=True)
