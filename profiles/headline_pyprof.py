"""cProfile of the headline loop's host side (one lane): which Python frames the ~0.7 ms per env step outside fg_multi_step are in.
python profiles/headline_pyprof.py [steps=60]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidgym_amd  # noqa: E402,F401
from fluidgym_amd.envs.parallel_env import ParallelFluidEnv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = 64
dev = torch.device("cuda", 0)
penv = ParallelFluidEnv("ChannelJet2D-v0", num_envs=B, lanes=int(os.environ.get("LANES", "1")))
penv.reset(seed=1234, randomize=True)
blks = [e._domain.getBlock(0) for e in penv.lane_envs]
for b in blks:
    b.setVelocitySource(torch.zeros_like(b.velocity))
fgen = torch.Generator(device=dev).manual_seed(4321)
ggen = torch.Generator(device=dev).manual_seed(7)


def one():
    for b in blks:
        b.velocitySource.normal_(0.0, 2.0, generator=fgen)
    penv.step(torch.rand(B, 1, generator=ggen, device=dev) * 2 - 1)


for _ in range(5):
    one()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    one()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
