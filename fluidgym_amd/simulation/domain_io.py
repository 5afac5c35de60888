"""On-disk domain format of the reference: ``<path>.json`` + ``<path>.npz``.

Same layout as ``pict/util/domain_io.py:64-327`` so that domains written by the reference (the published initial
states, ``fluid_env.py:1044-1190``) load here and files written here load there:

* ``.npz``: tensors stored flat under the keys ``"0", "1", ...`` (shared tensors stored once);
* ``.json``: ``name``, ``spatialDims``, ``viscosity``, ``passiveScalarChannels`` [, ``passiveScalarViscosity``],
  ``blocks`` = list of {``name``, ``velocity``, ``pressure`` [, ``scalar``, ``velocitySource``], ``vertexCoordinates``,
  ``boundaries`` = 2d entries {``type``: FIXED | PERIODIC | CONNECTED | DIRICHLET | DIRICHLET_VARYING, ...}} where every
  tensor field holds the npz key as a string, and ``data_info`` (shape / dtype / device per key).

What loads: ``load_domain`` -- one block on a rectilinear vertex grid with FIXED (also the two deprecated DIRICHLET spellings,
as the reference maps them, ``:264-283``) and PERIODIC faces; ``load_multiblock_domain`` / ``save_multiblock_domain`` -- several
curvilinear blocks with CONNECTED boundaries (``connectedBlock`` / ``axes``; the cylinder and airfoil envs' initial domains,
SURVEY 8f-3).  Per-block viscosity fields raise ``NotImplementedError``; ``load_domain`` points at ``load_multiblock_domain``
for a file that holds connected blocks.  The files hold ONE
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


def read_domain_file(path: str):
    """The two files of a stored domain as ``(dict, [numpy arrays])`` -- the pure-CPU half of ``load_domain``
    (domain_io.py:188-205): the JSON tree holds npz keys wherever the reference's writer stored a tensor."""
    with open(path + ".json") as fh:
        dd = json.load(fh)
    with np.load(path + ".npz") as z:
        data = [np.asarray(z[str(i)]) for i in range(len(z.files))]
    info = dd.get("data_info", {})
    for i, a in enumerate(data):          # the writer records shape / dtype per tensor (:172-180): cross-check
        meta = info.get(str(i))
        if meta is not None and list(meta["shape"]) != list(a.shape):
            raise ValueError(f"{path}: tensor {i} has shape {list(a.shape)} but data_info says {meta['shape']}")
    return dd, data


def load_domain(path: str, dtype=None, device=None, with_scalar: bool = True, batch: int = 1, prepare: bool = True) -> Domain:
    """``load_domain`` (domain_io.py:188-327).  Returns a prepared domain (the reference leaves ``PrepareSolve`` to the
    caller because it allocates; here the field tensors only exist afterwards) unless ``prepare=False``."""
    dd, arrays = read_domain_file(path)
    data = [torch.from_numpy(a).to(torch.float32) for a in arrays]
    if dtype not in (None, torch.float32):
        raise NotImplementedError("load_domain builds fp32 single-block domains; for float64 construct Domain(dtype=torch.float64) "
                                  "and copy the loaded fields in")
    get = lambda d, name: data[int(d[name])] if name in d else None

    if len(dd["blocks"]) != 1:
        raise NotImplementedError("this file holds a multi-block domain (connected boundaries): load it with "
                                  "load_multiblock_domain (SURVEY 8f-3)")
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
            raise NotImplementedError("CONNECTED boundary: this file holds a multi-block domain, load it with load_multiblock_domain")
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


# --------------------------------------------------------------------------------------------------------------------
# multi-block curvilinear domains (cylinder / airfoil initial states): same file layout, CONNECTED boundaries carry
# ``connectedBlock`` and ``axes`` (domain_io.py:157-160, 306-312)
# --------------------------------------------------------------------------------------------------------------------
_FACE_STR = ["-x", "+x", "-y", "+y", "-z", "+z"]


def save_multiblock_domain(domain, path: str, env: int = 0, name: str = "Domain") -> None:
    """``save_domain`` for env ``env`` of a prepared :class:`~fluidgym_amd.simulation.multiblock.MultiBlockDomain`."""
    data = []

    real = getattr(domain, "dtype", torch.float32)      # float64 domains are stored as float64 (the reference stores the tensors' dtype)

    def add(t, d: dict, key: str):
        d[key] = str(len(data))
        data.append(torch.as_tensor(t).detach().cpu().contiguous().to(real))

    dims = domain.dims
    dd = {"name": name, "spatialDims": dims}
    add(torch.tensor([domain.viscosity]), dd, "viscosity")
    dd["passiveScalarChannels"] = 0
    dd["blocks"] = []
    for blk in domain.blocks:
        bd = {"name": blk.name}
        add(blk.cells(domain.velocity[env: env + 1]), bd, "velocity")
        add(blk.cells(domain.pressure[env: env + 1, None]), bd, "pressure")
        add(blk.coords[None], bd, "vertexCoordinates")
        bd["boundaries"] = []
        for f in range(2 * dims):
            if f in blk.connections:
                other, axes = blk.connections[f]
                e = {"type": "CONNECTED", "connectedBlock": int(other), "axes": [int(a) for a in axes]}
            elif (f >> 1) in blk.periodic_axes:
                e = {"type": "PERIODIC"}
            else:
                e = {"type": "FIXED", "velocityType": "DIRICHLET"}
                shape = [1, dims] + [1 if a == (f >> 1) else blk.size[a] for a in reversed(range(dims))]
                add(blk.boundary(f)[env: env + 1].reshape(shape), e, "velocity")
            bd["boundaries"].append(e)
        dd["blocks"].append(bd)
    dd["data_info"] = {str(i): {"shape": list(t.shape), "dtype": "float64" if real == torch.float64 else "float32", "device": "cpu"}
                       for i, t in enumerate(data)}
    np.savez_compressed(path + ".npz", **{str(i): t.numpy() for i, t in enumerate(data)})
    with open(path + ".json", "w") as fh:
        json.dump(dd, fh)


def load_multiblock_domain(path: str, device=None, batch: int = 1, reference_quirks: bool = True, dtype=None):
    """``load_domain`` for a domain with CONNECTED boundaries: builds and prepares a ``MultiBlockDomain`` holding the stored
    state replicated over ``batch`` envs.  ``dtype``: torch.float32 / torch.float64; None = the dtype the file was written in."""
    from .multiblock import MultiBlockDomain

    dd, arrays = read_domain_file(path)
    if dtype is None:
        dtype = torch.float64 if all(np.asarray(a).dtype == np.float64 for a in arrays) else torch.float32
    npt = np.float64 if dtype == torch.float64 else np.float32
    data = [np.asarray(a, dtype=npt) for a in arrays]
    get = lambda d, key: data[int(d[key])] if key in d else None
    dims = int(dd["spatialDims"])
    if dd.get("passiveScalarChannels", 0):
        raise NotImplementedError("passive scalars on multi-block domains are not built")
    nu = float(np.asarray(get(dd, "viscosity")).reshape(-1)[0])
    dom = MultiBlockDomain(dims, nu, batch=batch, device=device, reference_quirks=reference_quirks, dtype=dtype)
    blocks = []
    for bd in dd["blocks"]:
        if "vertexCoordinates" not in bd:
            raise NotImplementedError("blocks stored by transform only (no vertex coordinates)")
        if "viscosity" in bd:
            raise NotImplementedError("per-block viscosity fields (SGS) are not built")
        blocks.append(dom.CreateBlock(get(bd, "vertexCoordinates"), name=bd["name"]))
    for bi, bd in enumerate(dd["blocks"]):
        for f, e in enumerate(bd["boundaries"]):
            t = e["type"]
            if t in ("FIXED", "DIRICHLET", "DIRICHLET_VARYING"):
                v = np.asarray(get(e, "velocity"))
                blocks[bi].CloseBoundary(f, v.reshape(dims, -1) if v.size > dims else v.reshape(dims, 1))
            elif t == "CONNECTED":
                other = int(e["connectedBlock"])
                if f in blocks[bi].connections:      # made from the other side already
                    continue
                axes = [int(a) for a in e["axes"]]
                blocks[bi].ConnectBlock(f, blocks[other], axes[0], axes[1] if dims > 1 else 0, axes[2] if dims > 2 else 0)
            elif t == "PERIODIC":
                blocks[bi].MakePeriodic(f >> 1)
            else:
                raise TypeError("Unknown boundary type: " + t)
    dom.PrepareSolve()
    for blk, bd in zip(blocks, dd["blocks"]):
        u = torch.as_tensor(get(bd, "velocity"), device=dom.device).reshape(dims, -1)
        p = torch.as_tensor(get(bd, "pressure"), device=dom.device).reshape(-1)
        dom.velocity[:, :, blk.cell_offset: blk.cell_offset + blk.n_cells] = u[None]
        dom.pressure[:, blk.cell_offset: blk.cell_offset + blk.n_cells] = p[None]
    return dom
