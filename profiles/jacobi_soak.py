"""Soak of the Jacobi-sweep velocity solver (policy advection_jacobi): many env steps with random actions, episode resets included; reports how
many velocity solves the sweeps settled, how many they handed to BiCGStab, unconverged solves, and whether every observation stayed finite.
    python profiles/jacobi_soak.py ENV_ID NUM_ENVS ENV_STEPS [forcing]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fluidgym_amd

env_id, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
forcing = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
env = fluidgym_amd.make(env_id, num_envs=B, **({"initial_domain_steps": 40, "randomize_initial_state": False} if env_id.startswith(("Cylinder", "Airfoil")) else {}))
env.reset(seed=3); env.seed(3)
dom = env._domain
blk = dom.getBlock(0) if hasattr(dom, "getBlock") else None
if forcing > 0 and blk is not None and hasattr(blk, "setVelocitySource"):
    blk.setVelocitySource(torch.zeros_like(blk.velocity))
g = torch.Generator(device="cuda").manual_seed(11)
finite, resets, t0 = True, 0, time.time()
for k in range(steps):
    if forcing > 0 and blk is not None and hasattr(blk, "velocitySource"):
        blk.velocitySource.normal_(0.0, forcing * (1.0 + 2.0 * (k % 7 == 0)), generator=g)      # every seventh step a three times stronger stir
    obs, r, term, trunc, info = env.step(env.sample_action())
    finite = finite and all(bool(torch.isfinite(v).all()) for v in obs.values()) and bool(torch.isfinite(r).all())
    if trunc:
        env.reset(seed=100 + k); resets += 1
solver = dom.solver if hasattr(dom, "solver") else dom
c = solver.solver_counters()
print(json.dumps({"env": env_id, "envs": B, "env_steps": steps, "resets": resets, "finite": finite, "seconds": round(time.time() - t0, 1),
                  "jacobi": solver.advection_jacobi_counts(),
                  "velocity": {k: c["velocity"][k] for k in ("mean", "max", "unconverged", "systems")},
                  "pressure_unconverged": [c[k]["unconverged"] for k in ("pressure0", "pressure1")],
                  "pressure_mean_iterations": [c[k]["mean"] for k in ("pressure0", "pressure1")],
                  "first_iterate": ({k: solver.config_dump().get(k) for k in ("first_iterate_polls", "unstored_pressure_solves", "jacobi_speculation_misses")} if hasattr(solver, "config_dump") else None)}))
env.close()
