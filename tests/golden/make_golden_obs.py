"""Golden vectors for the multi-agent observation windows from the reference's own functions.

``envs/util/obs_extraction.py`` cannot be imported as a module here (it pulls in the CUDA extension through
``fluidgym.envs``), but the three window functions are pure torch: they are compiled from the reference file at
generation time (AST, nothing is copied into this repo) and run on random fields.  Only inputs/outputs are written.

    python tests/golden/make_golden_obs.py  ->  tests/golden/reference_obs_windows.npz
"""
import ast
import os

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/src/fluidgym/envs/util/obs_extraction.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_functions(path, names):
    with open(path) as fh:
        tree = ast.parse(fh.read())
    ns = {"torch": torch, "F": F}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    return [ns[n] for n in names]


def main():
    w2d, w2dxz, w3d = load_functions(REF, ["extract_moving_window_2d", "extract_moving_window_2d_x_z", "extract_moving_window_3d"])
    rng = np.random.default_rng(11)
    out = {}
    for name, (n_agents, aw, W, Y) in {"a": (6, 4, 3, 5), "b": (12, 4, 11, 8), "c": (4, 8, 1, 3), "d": (5, 2, 4, 2)}.items():
        f = rng.standard_normal((Y, n_agents * aw)).astype(np.float32)
        out[f"w2d_{name}/field"] = f
        out[f"w2d_{name}/args"] = np.asarray([n_agents, aw, W], np.int32)
        out[f"w2d_{name}/expected"] = w2d(torch.from_numpy(f), n_agents, aw, W).numpy()
    for name, (nx, nz, aw, Wx, Wz, px, pz) in {"a": (8, 4, 2, 3, 3, 2, 1), "b": (8, 4, 2, 3, 3, 3, 1), "c": (6, 6, 3, 1, 1, 0, 0),
                                               "d": (6, 6, 3, 1, 1, 1, 0), "e": (5, 7, 2, 5, 3, 4, 1)}.items():
        f = rng.standard_normal((nz * aw, nx * aw)).astype(np.float32)
        out[f"w2dxz_{name}/field"] = f
        out[f"w2dxz_{name}/args"] = np.asarray([nx, nz, aw, Wx, Wz, px, pz], np.int32)
        out[f"w2dxz_{name}/expected"] = w2dxz(torch.from_numpy(f), nx, nz, aw, Wx, Wz, px, pz).numpy()
    for name, (n, aw, W, Y) in {"a": (4, 2, 3, 3), "b": (3, 4, 1, 2), "c": (5, 2, 4, 2)}.items():
        f = rng.standard_normal((n * aw, Y, n * aw)).astype(np.float32)
        out[f"w3d_{name}/field"] = f
        out[f"w3d_{name}/args"] = np.asarray([n, aw, W], np.int32)
        out[f"w3d_{name}/expected"] = w3d(torch.from_numpy(f), n, aw, W).numpy()
    np.savez_compressed(os.path.join(OUT, "reference_obs_windows.npz"), **out)
    for k in sorted(out):
        if k.endswith("expected"):
            print(k, out[k].shape)


if __name__ == "__main__":
    main()
