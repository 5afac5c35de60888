"""CylinderJet2D-easy-v0 (the reference's mesh, 14 232 cells) x num_envs: env-steps/s of the multi-block path."""
import sys, time; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import fluidgym_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # pressure solver: 0 CG (reference default), 1 BiCGStab, 2 BiCGStab + fp64 refinement
dev = int(sys.argv[4]) if len(sys.argv) > 4 else 20
env = fluidgym_amd.make("CylinderJet2D-easy-v0", num_envs=B, initial_domain_steps=dev, randomize_initial_state=False, pressure_use_BiCG=mode)
env.reset(seed=0)
a = torch.zeros(B, 1, device="cuda")
env.step(a)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(steps):
    obs, r, _, _, info = env.step(a)
torch.cuda.synchronize(); dt = (time.time() - t0) / steps
print(f"solver={mode} B={B} ms_per_env_step={dt*1e3:.1f} env_steps_per_s={B/dt:.1f} sim_steps_per_env_step={env.n_sim_steps} substeps_last={env._sim.last_substeps} its={env._sim.last_iterations} cd={info['drag'][0].item():.3f}")
