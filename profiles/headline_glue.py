"""Where an env step of the headline workload goes besides the native sim steps: wall time per env step of (a) the bench's loop
(actions on the host + ParallelFluidEnv.step), (b) the same with actions drawn on the GPU, (c) FluidEnv.step alone, (d) the n sim
steps alone (Simulation.multi_step with the jets of a step).  python profiles/headline_glue.py [envs=64] [steps=20]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidgym_amd  # noqa: E402
from fluidgym_amd.envs.parallel_env import ParallelFluidEnv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
penv = ParallelFluidEnv("ChannelJet2D-v0", num_envs=B)
env = penv.local_env
penv.reset(seed=1234, randomize=True)
blk0 = env._domain.getBlock(0)
blk0.setVelocitySource(torch.zeros_like(blk0.velocity))
fgen = torch.Generator(device=dev).manual_seed(4321)
cgen = torch.Generator(device="cpu").manual_seed(7)
ggen = torch.Generator(device=dev).manual_seed(7)
shape = (B,) + tuple(env._zero_action.shape[1:])
sync = lambda: torch.cuda.synchronize(dev)


def timed(fn, n):
    for _ in range(3):
        fn()
    sync(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return 1e3 * (time.perf_counter() - t0) / n


def perturb():
    blk0.velocitySource.normal_(0.0, 2.0, generator=fgen)


def a_host():
    perturb(); penv.step((torch.rand(shape, generator=cgen) * 2 - 1).to(dev))


def a_gpu():
    perturb(); penv.step(torch.rand(shape, generator=ggen, device=dev) * 2 - 1)


def env_only():
    perturb(); env.step(torch.rand(shape, generator=ggen, device=dev) * 2 - 1)


out = {}
out["bench_loop_host_actions_ms"] = timed(a_host, steps)
penv.reset(seed=1234, randomize=True)
out["bench_loop_gpu_actions_ms"] = timed(a_gpu, steps)
penv.reset(seed=1234, randomize=True)
out["fluid_env_step_ms"] = timed(env_only, steps)
penv.reset(seed=1234, randomize=True)
n = env._n_sim_steps
jets = env._jets if hasattr(env, "_jets") else None


def sim_only():
    perturb(); env._sim.multi_step(n, {2: jets[:, 0], 3: jets[:, 1]} if jets is not None else None)


out["sim_steps_only_ms"] = timed(sim_only, steps)
print(json.dumps({k: round(v, 3) for k, v in out.items()}))
penv.close()
