"""Feasibility of half-batch overlap on the multi-block path: one CylinderJet2D env of B envs against two envs of
B/2 stepped from two host threads on two streams of the same GPU, the second started `stagger` ms after the first so that one half's
on-chip pressure CG (one workgroup per env: 32 of 256 CUs) runs beside the other half's bandwidth-bound kernels.
usage: overlap_halves_exp.py [env_id] [B] [steps] [stagger_ms ...]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fluidgym_amd

env_id = sys.argv[1] if len(sys.argv) > 1 else "CylinderJet2D-easy-v0"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
staggers = [float(v) for v in sys.argv[4:]] or [0.0, 0.5, 1.0]
dev = torch.device("cuda", 0)


def make(n):
    env = fluidgym_amd.make(env_id, num_envs=n, initial_domain_steps=60, randomize_initial_state=False, cuda_device=dev)
    env.reset(seed=0)
    return env


def actions(n, count, seed=7):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [(torch.rand(n, 1, generator=g) * 2 - 1).to(dev) for _ in range(count)]


def run_single():
    env = make(B)
    acts = actions(B, steps + 1)
    env.step(acts[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[1 + k])
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    env.close()
    return B / el, 1e3 * el


def run_halves(stagger_ms):
    envs = [make(B // 2), make(B // 2)]
    acts = [actions(B // 2, steps + 1, seed=7 + h) for h in range(2)]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for h in range(2):
        with torch.cuda.stream(streams[h]):
            envs[h].step(acts[h][0])
    torch.cuda.synchronize()
    done = [0.0, 0.0]

    def work(h):
        torch.cuda.set_device(dev)
        if h == 1 and stagger_ms > 0:
            time.sleep(stagger_ms * 1e-3)
        with torch.cuda.stream(streams[h]):
            for k in range(steps):
                envs[h].step(acts[h][1 + k])
            streams[h].synchronize()
        done[h] = time.perf_counter()

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(h,)) for h in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = (max(done) - t0) / steps
    for e in envs:
        e.close()
    return B / el, 1e3 * el, [1e3 * (d - t0) / steps for d in done]


if __name__ == "__main__":
    v, ms = run_single()
    print(f"OVERLAP single env x{B}: {v:.1f} env-steps/s ({ms:.2f} ms/step)", flush=True)
    for s in staggers:
        v, ms, each = run_halves(s)
        print(f"OVERLAP two halves x{B // 2}, stagger {s} ms: {v:.1f} env-steps/s ({ms:.2f} ms/step; halves {each[0]:.2f} / {each[1]:.2f})", flush=True)
