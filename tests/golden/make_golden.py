"""Generate golden fixtures from the importable parts of the reference's *Python*.

Run HERE only (the container that has /root/reference); the GPU box never sees the
reference.  Only data (inputs + expected outputs) is written; no reference source text.

    python tests/golden/make_golden.py

What is pinned:
  * vertex grids of ``simulation/pict/data/shapes.py`` (``make_wall_refined_ortho_grid``,
    ``generate_grid_vertices_2D``, ``extrude_grid_z``, ``make_weights_exp``,
    ``make_weights_cos``) -> our own generators in ``fluidgym_amd/simulation/grids.py``;
  * the TCF wall-normal weights of ``envs/tcf/grid.py::_make_y_weights`` (the function
    object is compiled from the reference file at generation time, nothing is copied);
  * ``envs/util/profiles.py`` inflow / jet profiles.
The solver itself has no importable reference (CUDA only) -> "parity unpinned" there.
"""
import ast
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference/src/fluidgym"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _load_function(path, fn_name):
    """Compile ONE top-level function of a reference module without importing the module
    (its imports need the CUDA extension)."""
    with open(path) as fh:
        tree = ast.parse(fh.read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == fn_name:
            ns = {}
            exec(compile(ast.Module([node], []), path, "exec"), ns)
            return ns[fn_name]
    raise KeyError(fn_name)


def main():
    shapes = _load(f"{REF}/simulation/pict/data/shapes.py", "ref_shapes")
    profiles = _load(f"{REF}/envs/util/profiles.py", "ref_profiles")
    make_y_weights = _load_function(f"{REF}/envs/tcf/grid.py", "_make_y_weights")

    out = {}
    # --- RBC grids (envs/rbc/rbc_env_base.py:190-198): wall-refined in y, base 1.02
    for name, (nx, ny, L, base) in {
        "rbc_96x61": (96, 61, float(np.pi), 1.02),
        "rbc_16x9": (16, 9, 2.0, 1.1),
        "rbc_uniform_12x8": (12, 8, 3.0, 1.0),
    }.items():
        g = shapes.make_wall_refined_ortho_grid(
            nx, ny, corner_lower=(0, -0.5), corner_upper=(L, 0.5), wall_refinement=["-y", "+y"], base=base
        )
        out[f"{name}.coords"] = g.numpy()
        out[f"{name}.args"] = np.array([nx, ny, L, base], dtype=np.float64)
    # 3-D extrusion (rbc_env_base.py:200-208)
    g2 = shapes.make_wall_refined_ortho_grid(
        8, 5, corner_lower=(0, -0.5), corner_upper=(2.0, 0.5), wall_refinement=["-y", "+y"], base=1.02
    )
    g3 = shapes.extrude_grid_z(grid=g2, res_z=6, start_z=0.0, end_z=2.0, weights_z=None, exp_base=1)
    out["rbc3d_8x5x6.coords"] = g3.numpy()
    # --- weights
    out["weights_exp_both_10_1p2"] = np.array(shapes.make_weights_exp(10, 1.2, "BOTH"), dtype=np.float64)
    out["weights_exp_start_7_1p1"] = np.array(shapes.make_weights_exp(7, 1.1, "START"), dtype=np.float64)
    out["weights_exp_end_7_1p1"] = np.array(shapes.make_weights_exp(7, 1.1, "END"), dtype=np.float64)
    out["weights_cos_both_12"] = np.array(shapes.make_weights_cos(12, "BOTH"), dtype=np.float64)
    out["weights_exp_global_16_40"] = np.array(shapes.make_weights_exp_global(16, 40.0, "BOTH"), dtype=np.float64)
    # --- TCF (envs/tcf/grid.py:15-31, 34-81): y weights + 2-D grid + extrusion
    for N, ny_half in ((1, 48), (2, 48), (1, 16), (2, 32)):
        out[f"tcf_y_weights_N{N}_h{ny_half}"] = np.array(make_y_weights(N=N, ny_half=ny_half), dtype=np.float64)
    yw = make_y_weights(N=2, ny_half=16)
    y = len(yw) - 1
    H, L, D, x, z = 2.0, 2 * np.pi, np.pi, 8, 4
    corners = [(-L / 2, -H / 2), (L / 2, -H / 2), (-L / 2, H / 2), (L / 2, H / 2)]
    gt = shapes.generate_grid_vertices_2D([y + 1, x + 1], corners, None, x_weights=yw, dtype=torch.float32)
    gt3 = shapes.extrude_grid_z(gt, z, start_z=-D / 2, end_z=D / 2)
    out["tcf_8x16x4.coords"] = gt3.numpy()
    out["tcf_8x16x4.args"] = np.array([H, L, D, x, 16, 2, z], dtype=np.float64)
    # --- profiles (envs/util/profiles.py:6-90)
    out["jet_profile_h7"] = profiles.get_jet_profile(7, torch.float32, torch.device("cpu")).numpy()
    out["inflow_2d_h2_res16"] = profiles.get_inflow_profile(2.0, 16, 2, torch.float32, torch.device("cpu")).numpy()
    out["inflow_3d_h1_res8_z3"] = profiles.get_inflow_profile(
        1.0, 8, 3, torch.float32, torch.device("cpu"), res_z=3
    ).numpy()

    np.savez_compressed(os.path.join(OUT, "reference_python.npz"), **out)
    print("wrote", os.path.join(OUT, "reference_python.npz"), "with", len(out), "arrays")


if __name__ == "__main__":
    sys.exit(main())
