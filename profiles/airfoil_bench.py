"""Airfoil2D-easy-v0 x num_envs: env-steps/s of the multi-block path (refined BiCGStab)."""
import sys, time; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import fluidgym_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = int(sys.argv[3]) if len(sys.argv) > 3 else 60
env = fluidgym_amd.make("Airfoil2D-easy-v0", num_envs=B, initial_domain_steps=dev, randomize_initial_state=False)
env.reset(seed=0)
a = torch.zeros(B, 3, device="cuda")
env.step(a)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(steps):
    obs, r, _, _, info = env.step(a)
torch.cuda.synchronize(); dt = (time.time() - t0) / steps
print(f"B={B} ms_per_env_step={dt*1e3:.1f} env_steps_per_s={B/dt:.2f} substeps_last={env._sim.last_substeps} its={env._sim.last_iterations} cd={info['drag'][0].item():.3f} cl={info['lift'][0].item():.3f} multilevel={env._domain.multilevel_status()} counters={ {k: v['mean'] for k, v in env._domain.solver_counters().items() if isinstance(v, dict)} }")
