import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from th_rl_amd.nn import ReinforceBatch
G = 65536
rb = ReinforceBatch(G, actions=21, gamma=0.995, seed=1).init()
for n, distinct in ((1000, 41), (500, 41), (250, 41), (1000, 0), (250, 0)):
    if distinct:
        price = torch.randint(20, 61, (n, G), device="cuda").double() / 10.0
    else:
        price = torch.rand((n, G), device="cuda", dtype=torch.float64) * 4 + 2
    action = torch.randint(0, 21, (n, G), device="cuda", dtype=torch.int32)
    reward = torch.rand((n, G), device="cuda", dtype=torch.float64) * 10 + 5
    rb.train(price, action, reward); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): rb.train(price, action, reward)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print("n=%d distinct=%s: %.2f ms per update" % (n, distinct or "all", dt * 1e3), flush=True)
