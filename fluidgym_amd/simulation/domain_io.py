"""On-disk domain format of the reference: ``<path>.json`` + ``<path>.npz``.

Same layout as ``pict/util/domain_io.py:64-327`` so that domains written by the reference (the published initial
states, ``fluid_env.py:1044-1190``) load here and files written here load there:

* ``.npz``: tensors stored flat under the keys ``"0", "1", ...`` (shared tensors stored once);
* ``.json``: ``name``, ``spatialDims``, ``viscosity``, ``passiveScalarChannels`` [, ``passiveScalarViscosity``],
  ``blocks`` = list of {``name``, ``velocity``, ``pressure`` [, ``scalar``, ``velocitySource``], ``vertexCoordinates``,
  ``boundaries`` = 2d entries {``type``: FIXED | PERIODIC | CONNECTED | DIRICHLET | DIRICHLET_VARYING, ...}} where every
  tensor field holds the npz key as a string, and ``data_info`` (shape / dtype / device per key).

What loads: one block on a rectilinear vertex grid with FIXED (also the two deprecated DIRICHLET spellings, as the
reference maps them, ``:264-283``) and PERIODIC faces.  CONNECTED boundaries, several blocks, per-block viscosity
fields or a non-rectilinear grid raise ``NotImplementedError`` (multi-block meshes are SURVEY 8f-3).  The files hold ONE
env (tensor batch 1); ``load_domain(..., batch=B)`` replicates it over the env axis and ``save_domain(..., env=b)`` writes
one env of a batched domain.
"""
from __future__ import annotations

import json
from typing import Optional

import numpy as np
import torch

from .domain import BoundaryConditionType, Domain


def _bc_to_string(t: int) -> str:
    if t == BoundaryConditionType.DIRICHLET:
        return "DIRICHLET"
    if t == BoundaryConditionType.NEUMANN:
        return "NEUMANN"
    raise TypeError("Unsupported boundary condition type.")


def _bc_from_string(s: str) -> int:
    if s == "DIRICHLET":
        return BoundaryConditionType.DIRICHLET
    if s == "NEUMANN":
        return BoundaryConditionType.NEUMANN
    raise TypeError("Unsupported boundary condition type.")


def save_domain(domain: Domain, path: str, env: int = 0) -> None:
    """``save_domain`` (domain_io.py:64-185) for env ``env`` of a prepared single-block domain; ``path`` without
    extension."""
    assert domain.IsInitialized(), "PrepareSolve() first: the field tensors live in the solver"
    data = []

    def add(t: torch.Tensor, d: dict, name: str):
        d[name] = str(len(data))
        data.append(t.detach().cpu().contiguous())

    one = lambda t: t[env: env + 1]
    dd = {"name": domain.name, "spatialDims": domain.dims}
    add(domain.viscosity, dd, "viscosity")
    dd["passiveScalarChannels"] = domain.n_scalars
    if domain._scalar_viscosity is not None:
        add(torch.tensor(domain._scalar_viscosity, dtype=torch.float32), dd, "passiveScalarViscosity")
    dd["blocks"] = []
    for blk in domain.getBlocks():
        bd = {"name": blk.name}
        add(one(blk.velocity), bd, "velocity")
        add(one(blk.pressure), bd, "pressure")
        if domain.n_scalars:
            add(one(blk.passiveScalar), bd, "scalar")
        if blk.velocitySource is not None:
            add(one(blk.velocitySource), bd, "velocitySource")
        add(torch.as_tensor(blk.vertexCoordinates, dtype=torch.float32), bd, "vertexCoordinates")
        bd["boundaries"] = []
        for f in range(2 * domain.dims):
            if blk.isFixed(f):
                b = blk.getBoundary(f)
                e = {"type": "FIXED", "velocityType": "DIRICHLET"}
                add(one(b.velocity), e, "velocity")
                if domain.n_scalars:
                    e["passiveScalarType"] = [_bc_to_string(t) for t in b.passiveScalarTypes]
                    add(one(b.passiveScalar), e, "scalar")
            else:
                e = {"type": "PERIODIC"}
            bd["boundaries"].append(e)
        dd["blocks"].append(bd)
    dd["data_info"] = {str(i): {"shape": list(t.shape), "dtype": "float32" if t.dtype == torch.float32 else "float64",
                                "device": "cpu"} for i, t in enumerate(data)}
    np.savez_compressed(path + ".npz", **{str(i): t.numpy() for i, t in enumerate(data)})
    with open(path + ".json", "w") as fh:
        json.dump(dd, fh)


def load_domain(path: str, dtype=None, device=None, with_scalar: bool = True, batch: int = 1, prepare: bool = True) -> Domain:
    """``load_domain`` (domain_io.py:188-327).  Returns a prepared domain (the reference leaves ``PrepareSolve`` to the
    caller because it allocates; here the field tensors only exist afterwards) unless ``prepare=False``."""
    with open(path + ".json") as fh:
        dd = json.load(fh)
    with np.load(path + ".npz") as z:
        data = [torch.from_numpy(np.asarray(z[str(i)])).to(torch.float32) for i in range(len(z.files))]
    if dtype not in (None, torch.float32):
        raise NotImplementedError("the HIP path computes in fp32")
    get = lambda d, name: data[int(d[name])] if name in d else None

    if len(dd["blocks"]) != 1:
        raise NotImplementedError("multi-block domains (connected boundaries) are not built yet (SURVEY 8f-3)")
    n_scal = dd.get("passiveScalarChannels", 1) if with_scalar else 0
    dom = Domain(dd["spatialDims"], get(dd, "viscosity"), passiveScalarChannels=n_scal, name=dd["name"], device=device,
                 batch=batch)
    sv = get(dd, "passiveScalarViscosity") if with_scalar else None
    bd = dd["blocks"][0]
    if "vertexCoordinates" not in bd:
        raise NotImplementedError("blocks stored by transform (no vertex coordinates) cannot be checked for rectilinearity")
    if "viscosity" in bd or "passiveScalarViscosity" in bd:
        raise NotImplementedError("per-block viscosity fields (SGS) are not built")
    blk = dom.CreateBlock(vertexCoordinates=get(bd, "vertexCoordinates"), name=bd["name"])  # raises if not rectilinear
    for f, e in enumerate(bd["boundaries"]):
        t = e["type"]
        if t in ("FIXED", "DIRICHLET", "DIRICHLET_VARYING"):
            has_scalar = bool(n_scal) and "scalar" in e
            bnd = blk.CloseBoundary(f, get(e, "velocity"), get(e, "scalar") if has_scalar else None)
            if has_scalar and "passiveScalarType" in e:
                pst = e["passiveScalarType"]
                bnd.setPassiveScalarType([_bc_from_string(s) for s in (pst if isinstance(pst, list) else [pst] * n_scal)])
        elif t == "PERIODIC":
            continue
        elif t == "CONNECTED":
            raise NotImplementedError("CONNECTED boundaries need multi-block support (SURVEY 8f-3)")
        else:
            raise TypeError("Unknown boundary type: " + t)
    if not prepare:
        return dom
    dom.PrepareSolve()
    if sv is not None:
        dom.setScalarViscosity(sv)
    blk.setVelocity(get(bd, "velocity"))
    blk.setPressure(get(bd, "pressure"))
    if n_scal and "scalar" in bd:
        blk.setPassiveScalar(get(bd, "scalar"))
    if "velocitySource" in bd:
        blk.setVelocitySource(get(bd, "velocitySource"))
    dom.solver.reset_solver_state()
    return dom
