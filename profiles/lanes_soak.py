"""Soak of ParallelFluidEnv(lanes=2): STEPS env steps of 64 stirred channel envs as two lanes on two HIP streams against the same two
32-env batches stepped alone, one after the other -- bit for bit (fields, observations, rewards), across episode resets, with the body
force redrawn every step.  A stream-ordering slip between the caller's stream, the lanes' streams and the concatenation would show
here.  python profiles/lanes_soak.py [steps=240]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidgym_amd  # noqa: E402
from fluidgym_amd.envs.parallel_env import ParallelFluidEnv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 240
dev = torch.device("cuda", 0)
penv = ParallelFluidEnv("ChannelJet2D-v0", num_envs=64, lanes=2)
plain = [fluidgym_amd.make("ChannelJet2D-v0", num_envs=32) for _ in range(2)]
gen = torch.Generator(device=dev).manual_seed(11)
mismatches, resets = 0, 0


def reset(seed):
    penv.reset(seed=seed, randomize=True)
    for r, e in enumerate(plain):
        e.reset(seed=seed + r, randomize=True)
    for e in penv.lane_envs + plain:
        b = e._domain.getBlock(0)
        b.setVelocitySource(torch.zeros_like(b.velocity))


reset(100)
for k in range(steps):
    if penv.lane_envs[0]._n_steps >= penv.lane_envs[0].episode_length:
        resets += 1
        reset(100 + 7 * resets)
    force = torch.randn(64, 2, 128, 256, device=dev, generator=gen) * 2.0
    for l, e in enumerate(penv.lane_envs):
        e._domain.getBlock(0).velocitySource.copy_(force[32 * l: 32 * l + 32])
    for l, e in enumerate(plain):
        e._domain.getBlock(0).velocitySource.copy_(force[32 * l: 32 * l + 32])
    a = torch.rand(64, 1, device=dev, generator=gen) * 2 - 1
    o, r, term, trunc, info = penv.step(a)
    outs = [e.step(a[32 * i: 32 * i + 32]) for i, e in enumerate(plain)]
    ok = torch.equal(r, torch.cat([x[1] for x in outs]))
    for key in o:
        ok = ok and torch.equal(o[key], torch.cat([x[0][key] for x in outs]))
    for e, q in zip(penv.lane_envs, plain):
        ok = ok and torch.equal(e._block.velocity, q._block.velocity) and torch.equal(e._block.pressure, q._block.pressure)
    mismatches += 0 if ok else 1
print(json.dumps({"steps": steps, "episode_resets": resets, "steps_with_a_mismatch": mismatches}))
penv.close()
