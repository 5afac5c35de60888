"""How fast would point Jacobi contract on the velocity systems of the multi-block envs?  (matrix buffers of the last step + the
neighbour table from the library, sweeps in torch; the BiCGStab iteration counts of the same systems beside them)"""
import sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np, fluidgym_amd

def run(env_id, B, dev_steps):
    env = fluidgym_amd.make(env_id, num_envs=B, initial_domain_steps=dev_steps, randomize_initial_state=False)
    env.reset(seed=0)
    g = torch.Generator(device="cpu").manual_seed(3)
    na = env._zero_action.shape[1:]
    for _ in range(2):
        env.step((torch.rand((B,) + tuple(na), generator=g) * 2 - 1).cuda())
    dom = env._domain
    N = dom.n_cells
    A = dom.buffer(0).view(B, 1, N).double(); C = dom.buffer(1).view(B, 4, N).double(); b = dom.buffer(2).view(B, 2, N).double()
    nb = torch.from_numpy(dom.neighbors().astype(np.int64)).cuda()      # [4, N], negative = no neighbour
    ok = (nb >= 0)
    idx = nb.clamp(min=0)
    def offx(x):
        out = torch.zeros_like(x)
        for f in range(4):
            out += C[:, f:f + 1] * torch.where(ok[f], x[..., idx[f]], torch.zeros_like(x))
        return out
    ratio = (C.abs().sum(1, keepdim=True) / A).amax().item()
    c = dom.solver_counters()
    print(f"{env_id}: cells {N}, max row sum|off|/diag {ratio:.3f}, BiCGStab iterations of the env's velocity solves: mean {c['velocity']['mean']:.1f} max {c['velocity']['max']}")
    x = torch.zeros_like(b)
    for k in range(1, 201):
        xn = (b - offx(x)) / A
        rms = (A * (xn - x)).pow(2).mean(dim=2).sqrt().amax().item()
        x = xn
        if k in (1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 200) or rms < 1e-5:
            print(f"   sweep {k:3d}: rms residual {rms:.3e}")
        if rms < 1e-5: break
    env.close()

run("CylinderJet2D-easy-v0", 4, 60)
run("Airfoil2D-easy-v0", 2, 40)
