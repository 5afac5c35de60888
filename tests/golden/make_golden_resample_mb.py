"""Golden vectors for resampling a MULTI-BLOCK curvilinear mesh onto the uniform observation grid: the reference's
pure-torch ``sample_multi_coords_to_uniform_grid_diff`` (simulation/pict/data/resample.py:361-548) applied to the
recorded cylinder mesh (tests/golden/reference_cylinder_grid.npz, resolution 8) with the output shape and fill steps the
cylinder env uses (cylinder_env_base.py:205-209, 321-322).  Only inputs and expected outputs are written.

    python tests/golden/make_golden_resample_mb.py  ->  tests/golden/reference_resample_mb.npz
"""
import os

import numpy as np
import torch

from make_golden_resample import load_reference_resample

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    res = load_reference_resample()
    G = np.load(os.path.join(OUT, "reference_cylinder_grid.npz"))
    rng = np.random.default_rng(77)
    r = 8
    coords = [G[f"r{r}_block{b}"] for b in range(5)]
    out_shape = (int(4 * r / 4.1 * 22.0), 4 * r)  # (x, y) = render_shape[:2]
    out = {"out_shape": np.asarray(out_shape, np.int32)}
    for C, fill in ((2, 16), (1, 0)):
        data = [rng.standard_normal((1, C, c.shape[1] - 1, c.shape[2] - 1)).astype(np.float32) for c in coords]
        y = res.sample_multi_coords_to_uniform_grid_diff([torch.from_numpy(d) for d in data],
                                                         [torch.from_numpy(c[None]) for c in coords], list(out_shape),
                                                         fill_max_steps=fill)
        for b in range(5):
            out[f"c{C}_f{fill}/data{b}"] = data[b]
        out[f"c{C}_f{fill}/expected"] = y.numpy()
        print(C, fill, tuple(y.shape), "zeros:", int((y == 0).sum()))
    # 3-D: the extruded cylinder mesh (z-periodic, 8 cells over [-2, 2]) onto render_shape (x, y, z); the pure-torch
    # implementation splats onto all 8 corners (the compiled kernel onto 6, see oracle/resample_oracle.py)
    coords3 = [G[f"r{r}_3d_block{b}"] for b in range(5)]
    out_shape3 = (out_shape[0], 4 * r, 4 * r)
    out["out_shape3"] = np.asarray(out_shape3, np.int32)
    for C, fill in ((3, 16), (1, 0)):
        data = [rng.standard_normal((1, C, c.shape[1] - 1, c.shape[2] - 1, c.shape[3] - 1)).astype(np.float32) for c in coords3]
        y = res.sample_multi_coords_to_uniform_grid_diff([torch.from_numpy(d) for d in data],
                                                         [torch.from_numpy(c[None]) for c in coords3], list(out_shape3),
                                                         fill_max_steps=fill)
        for b in range(5):
            out[f"3d_c{C}_f{fill}/data{b}"] = data[b]
        out[f"3d_c{C}_f{fill}/expected"] = y.numpy().astype(np.float16 if False else np.float32)
        print("3d", C, fill, tuple(y.shape), "zeros:", int((y == 0).sum()))
    np.savez_compressed(os.path.join(OUT, "reference_resample_mb.npz"), **out)


if __name__ == "__main__":
    main()
