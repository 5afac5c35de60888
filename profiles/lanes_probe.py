"""Where do in-process lanes lose against two processes?  L threads, each its own env batch of 64/L envs on its own HIP stream, each
looping over the NATIVE part of an env step only (Simulation.multi_step: one ctypes call, GIL released) -- no Python glue inside the
loop besides the redraw of the body force.  python profiles/lanes_probe.py [lanes=2] [steps=20]"""
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidgym_amd  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = 64
dev = torch.device("cuda", 0)
envs = [fluidgym_amd.make("ChannelJet2D-v0", num_envs=B // L, cuda_device=dev) for _ in range(L)]
gen = torch.Generator(device=dev).manual_seed(1)
for i, e in enumerate(envs):
    e.reset(seed=1234 + i, randomize=True)
    b = e._domain.getBlock(0)
    b.setVelocitySource(torch.zeros_like(b.velocity))
    b.velocitySource.normal_(0.0, 2.0, generator=gen)
    e.step(torch.rand(B // L, 1, device=dev, generator=gen) * 2 - 1)
torch.cuda.synchronize()
streams = [torch.cuda.Stream(dev) for _ in range(L)]
n = envs[0]._n_sim_steps


gens = [torch.Generator(device=dev).manual_seed(10 + l) for l in range(L)]


def loop(l, k):
    torch.cuda.set_device(dev)
    e = envs[l]
    src = e._domain.getBlock(0).velocitySource
    with torch.cuda.stream(streams[l]):
        for _ in range(k):
            src.normal_(0.0, 2.0, generator=gens[l])      # (redrawn per env step like the bench: a constant force would spin the flow up)
            e._sim.multi_step(n, {2: e._jets[:, 0], 3: e._jets[:, 1]})
        streams[l].synchronize()


def run(k):
    ts = [threading.Thread(target=loop, args=(l, k)) for l in range(1, L)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    loop(0, k)
    for t in ts:
        t.join()
    return time.perf_counter() - t0


run(3)
el = run(steps)
print(json.dumps({"lanes": L, "native_only_env_steps_per_s": round(B * steps / el, 1), "ms_per_env_step": round(1e3 * el / steps, 3)}))
