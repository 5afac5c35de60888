"""cProfile of env.step on the host side (where the Python time of an env step goes).  python profiles/env_cprofile.py ENV_ID NUM_ENVS [steps=3]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluidgym_amd  # noqa: E402

env_id, B = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
env = fluidgym_amd.make(env_id, num_envs=B)
env.reset(seed=0)
env.step(env.sample_action())
acts = [env.sample_action() for _ in range(steps)]
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for a in acts:
    env.step(a)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
env.close()
