"""Golden vectors for the observation resampling from the reference's importable Python.

Run HERE only (needs /root/reference).  ``simulation/pict/data/resample.py`` is imported by file path with stub
modules for the two package imports it makes (the compiled ``PISOtorch`` extension is never called by the
pure-torch function we use):

    sample_multi_coords_to_uniform_grid_diff(data_list, coords_list, out_shape, fill_max_steps=...)

which the reference's own test (``tests/simulation/test_torch_resample.py``) holds equal to the compiled
``SampleTransformedGridLocalToGlobalMulti`` kernel.  Only inputs and expected outputs are written.

    python tests/golden/make_golden_resample.py  ->  tests/golden/reference_resample.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src/fluidgym"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_resample():
    shapes = _load(f"{REF}/simulation/pict/data/shapes.py", "fluidgym.simulation.pict.data.shapes")
    for pkg in ("fluidgym", "fluidgym.simulation", "fluidgym.simulation.pict", "fluidgym.simulation.pict.data",
                "fluidgym.simulation.extensions"):
        sys.modules.setdefault(pkg, types.ModuleType(pkg))
    sys.modules["fluidgym.simulation.extensions"].PISOtorch = types.ModuleType("PISOtorch_stub")
    sys.modules["fluidgym.simulation.pict.data"].shapes = shapes
    return _load(f"{REF}/simulation/pict/data/resample.py", "fluidgym.simulation.pict.data.resample")


def stretched_edges(n, length, rng, amount):
    w = 1.0 + amount * rng.uniform(-1, 1, n)
    e = np.concatenate([[0.0], np.cumsum(w)])
    return (e / e[-1] * length).astype(np.float32)


def vertex_coords(edges):
    """[1, d, (nz+1,) ny+1, nx+1] with channel order x, y(, z) -- the layout of Block.getVertexCoordinates()."""
    d = len(edges)
    grids = np.meshgrid(*[edges[a] for a in reversed(range(d))], indexing="ij")  # (z,) y, x order
    comps = [grids[d - 1 - a] for a in range(d)]
    return np.stack(comps)[None].astype(np.float32)


CASES = {
    # name: (n = (nx, ny[, nz]), lengths, stretch, channels, out_shape (x, y[, z]), fill_max_steps)
    "2d_down": ((24, 16), (3.0, 2.0), 0.3, 2, (12, 8), 0),
    "2d_same": ((16, 12), (2.0, 1.5), 0.0, 1, (16, 12), 0),
    "2d_up_fill": ((10, 6), (2.0, 1.0), 0.4, 2, (32, 20), 16),
    "2d_up_nofill": ((10, 6), (2.0, 1.0), 0.4, 1, (32, 20), 0),
    "2d_aniso_fill2": ((12, 12), (4.0, 1.0), 0.2, 1, (40, 24), 2),
    "3d_down": ((12, 8, 10), (2.0, 1.0, 1.5), 0.3, 3, (6, 5, 4), 0),
    "3d_up_fill": ((6, 5, 4), (2.0, 1.0, 1.5), 0.3, 1, (14, 9, 10), 16),
}


def main():
    res = load_reference_resample()
    rng = np.random.default_rng(2024)
    out = {}
    for name, (n, lengths, stretch, C, oshape, fill) in CASES.items():
        d = len(n)
        edges = [stretched_edges(n[a], lengths[a], rng, stretch) for a in range(d)]
        vc = vertex_coords(edges)
        data = rng.standard_normal((1, C) + tuple(reversed(n))).astype(np.float32)
        y = res.sample_multi_coords_to_uniform_grid_diff([torch.from_numpy(data)], [torch.from_numpy(vc)], list(oshape),
                                                         fill_max_steps=fill)
        for a in range(d):
            out[f"{name}/edges{a}"] = edges[a]
        out[f"{name}/data"] = data
        out[f"{name}/out_shape"] = np.asarray(oshape, np.int32)
        out[f"{name}/fill"] = np.asarray(fill, np.int32)
        out[f"{name}/expected"] = y.numpy()
        print(name, data.shape, "->", tuple(y.shape), "zeros:", int((y == 0).sum()))
    np.savez_compressed(os.path.join(OUT, "reference_resample.npz"), **out)


if __name__ == "__main__":
    main()
