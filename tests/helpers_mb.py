"""Mesh specifications shared by the multi-block tests: one description builds the oracle domain (CPU) and the HIP
domain (GPU), so both see the same blocks, connections and boundary values."""
import numpy as np

from oracle import mb_oracle as mbo


class Spec:
    def __init__(self, dims, nu):
        self.dims, self.nu = dims, nu
        self.blocks = []      # vertex coordinate arrays [d, ny+1, nx+1]
        self.fixed = []       # (block, face, velocity [d, face cells])
        self.connections = [] # (b1, face1, b2, face2, axis1)
        self.periodic = []    # (block, axis)

    def oracle(self):
        d = mbo.Domain(self.dims, self.nu)
        for c in self.blocks:
            d.add_block(c)
        for b, f, v in self.fixed:
            d.close(b, f, v)
        for b, a in self.periodic:
            d.make_periodic(b, a)
        for c in self.connections:
            d.connect(*c)  # (b1, face1, b2, face2, axis1[, axis2])
        d.finalize()
        return d

    def native(self, batch=1, reference_quirks=True, non_ortho_flags=25, dtype=None):
        import torch

        from fluidgym_amd.simulation.multiblock import MultiBlockDomain

        dom = MultiBlockDomain(self.dims, self.nu, batch=batch, reference_quirks=reference_quirks, non_ortho_flags=non_ortho_flags,
                               dtype=dtype or torch.float32)
        # (fp64 build: the vertex coordinates as they are -- the oracle computes from the same doubles)
        blks = [dom.CreateBlock(c.astype(np.float64 if dtype == torch.float64 else np.float32)) for c in self.blocks]
        for b, f, v in self.fixed:
            blks[b].CloseBoundary(f, v)
        for b, a in self.periodic:
            blks[b].MakePeriodic(a)
        for b1, f1, b2, f2, a1, *rest in self.connections:
            blks[b1].ConnectBlock(f1, blks[b2], f2, a1, *(rest or [4]))
        dom.PrepareSolve()
        return dom


def rot90(c):
    """Same cells stored with xi' = +eta, eta' = -xi (right-handed)."""
    return np.ascontiguousarray(c[:, :, ::-1].transpose(0, 2, 1))


def split_rotated_channel(nx=12, ny=8, cut=5, nu=0.02):
    x = np.linspace(0.0, 3.0, nx + 1) ** 1.15
    t = np.linspace(-1.0, 1.0, ny + 1)
    y = 0.5 * (np.tanh(1.3 * t) / np.tanh(1.3) + 1.0)
    X, Y = np.meshgrid(x, y)
    coords = np.stack([X, Y])
    yc = 0.5 * (y[1:] + y[:-1])
    inflow = np.stack([4.0 * yc * (1.0 - yc), np.zeros(ny)])
    s = Spec(2, nu)
    s.blocks = [coords[:, :, :cut + 1].copy(), rot90(coords[:, :, cut:])]
    s.fixed = [(0, 0, inflow), (1, 2, inflow)]
    s.connections = [(0, 1, 1, 3, 0)]
    return s


def skewed_pair(nu=0.03, shear=0.3, wobble=0.05, stretch=1.1):
    """Two skewed blocks joined over a shuffled connection, moving lid on top, through-flow left to right."""
    xi = np.linspace(0.0, 1.0, 6)
    eta = np.linspace(0.0, 1.0, 7) ** stretch
    X = xi[None, :] + shear * eta[:, None]
    Y = eta[:, None] + 0.5 * shear * xi[None, :] + wobble * np.sin(3.0 * xi[None, :]) * eta[:, None]
    c1 = np.stack([X, Y])
    X2 = 1.0 + xi[None, :] + shear * eta[:, None]
    Y2 = eta[:, None] + 0.5 * shear * (1.0 + xi[None, :]) + wobble * np.sin(3.0 * (1.0 + xi[None, :])) * eta[:, None]
    # make the two blocks share their interface vertices exactly
    c2 = np.stack([X2, Y2])
    c2[:, :, 0] = c1[:, :, -1]
    ny, nx = len(eta) - 1, len(xi) - 1
    yc = 0.5 * (eta[1:] + eta[:-1])
    through = np.stack([0.5 + 0.2 * yc, 0.05 * yc])
    lid = np.stack([np.full(nx, 0.7), np.zeros(nx)])
    s = Spec(2, nu)
    s.blocks = [c1, rot90(c2)]
    # block 1 is rotated: its -y face is the original +x side (cells run along xi' = original eta),
    # its +x face is the original top (cells run along eta' = reversed original xi)
    s.fixed = [(0, 0, through), (1, 2, through), (0, 3, lid), (1, 1, lid)]
    s.connections = [(0, 1, 1, 3, 0)]
    return s


def twisted_ring(nr=5, nt=8, nu=0.05, twist=0.25, parts=3):
    """An annulus cut into `parts` blocks connected in a ring (xi = angle, eta = radius), radial lines twisted so the
    cells are non-orthogonal; the inner wall rotates."""
    s = Spec(2, nu)
    r = np.linspace(0.5, 1.5, nr + 1) ** 1.0
    for k in range(parts):
        th = np.linspace(2 * np.pi * k / parts, 2 * np.pi * (k + 1) / parts, nt + 1)
        # right-handed with xi = -theta (clockwise), eta = r
        TH = -th[None, :] + twist * (r[:, None] - 0.5)
        X = r[:, None] * np.cos(TH)
        Y = r[:, None] * np.sin(TH)
        s.blocks.append(np.stack([X, Y]))
    for k in range(parts):
        s.connections.append((k, 1, (k + 1) % parts, 0, 2))
        c = s.blocks[k]
        # inner wall (-y face): tangential velocity of a rotating cylinder, evaluated at the face centres
        xm = 0.5 * (c[0, 0, 1:] + c[0, 0, :-1])
        ym = 0.5 * (c[1, 0, 1:] + c[1, 0, :-1])
        s.fixed.append((k, 2, np.stack([-0.8 * ym, 0.8 * xm])))
    return s


def mild_skewed_pair():
    """Same topology, cells within a few degrees of orthogonal: the regime of the reference's meshes, where its CG
    pressure solve converges although the matrix is not exactly symmetric."""
    return skewed_pair(shear=0.04, wobble=0.01)


def polar_ring():
    """Orthogonal O-grid (the cylinder mesh's inner ring is one): curved cells, three connections in a cycle."""
    return twisted_ring(twist=0.0)


def extrude(c2, z):
    """[2, ny+1, nx+1] -> [3, nz+1, ny+1, nx+1]"""
    nzv = len(z)
    xy = np.broadcast_to(c2[:, None], (2, nzv) + c2.shape[1:])
    zz = np.broadcast_to(np.asarray(z)[None, :, None, None], (1, nzv) + c2.shape[1:])
    return np.ascontiguousarray(np.concatenate([xy, zz], axis=0))


def skewed_pair_3d(nu=0.03, nz=3):
    """The skewed two-block mesh extruded along z (periodic), blocks joined +x -> -x; a lid moves in x and z."""
    s2 = skewed_pair(nu)
    c1 = s2.blocks[0]
    # undo the storage rotation of the second block: 3-D keeps both blocks in the same orientation
    c2 = np.ascontiguousarray(s2.blocks[1].transpose(0, 2, 1)[:, :, ::-1])
    z = np.linspace(0.0, 0.9, nz + 1)
    s = Spec(3, nu)
    s.blocks = [extrude(c1, z), extrude(c2, z)]
    ny, nx = c1.shape[1] - 1, c1.shape[2] - 1
    eta = 0.5 * (np.linspace(0.0, 1.0, ny + 1)[1:] + np.linspace(0.0, 1.0, ny + 1)[:-1])
    through = np.zeros((3, nz, ny))
    through[0] = 0.5 + 0.2 * eta[None, :]
    through[2] = 0.1
    lid = np.zeros((3, nz, nx))
    lid[0], lid[2] = 0.7, 0.2
    s.fixed = [(0, 0, through.reshape(3, -1)), (1, 1, through.reshape(3, -1)), (0, 3, lid.reshape(3, -1)), (1, 3, lid.reshape(3, -1))]
    s.periodic = [(0, 2), (1, 2)]
    s.connections = [(0, 1, 1, 0, 2, 4)]
    return s


def odd_channel():
    """Cell count not divisible by four: the solvers' one-cell-per-thread kernels instead of the four-cell ones."""
    return split_rotated_channel(nx=11, ny=7, cut=4)


def cylinder_3d_small(res=4, res_z=3, nu=0.01):
    """The reference's 3-D cylinder mesh (extruded, z-periodic, connections carry two axes) at a small resolution, with
    perturbed Dirichlet values so that walls and inflow exercise every boundary branch."""
    from fluidgym_amd.envs.cylinder_grid import extrude_mesh, make_vortex_street_mesh

    m = extrude_mesh(make_vortex_street_mesh(res), res_z)
    F = {"-x": 0, "+x": 1, "-y": 2, "+y": 3, "-z": 4, "+z": 5}
    s = Spec(3, nu)
    s.blocks = [c.astype(np.float64) for c in m.coords]
    rng = np.random.default_rng(6)
    s.fixed = [(b, F[f], v.astype(np.float64) + 0.1 * rng.standard_normal(v.shape)) for (b, f), v in m.fixed.items()]
    s.connections = [(b1, F[f1], b2, F[f2], F[a1], F[a2]) for b1, f1, b2, f2, a1, a2 in m.connections]
    s.periodic = [(b, 2) for b, _ in m.periodic]
    return s


def cylinder_2d(res=8, nu=0.01, noise=0.02):
    """The reference's five-block cylinder mesh (envs/cylinder/grid.py) in 2-D at a small resolution (``resolution 8``), inflow /
    outflow / wall values perturbed a little; the outflow face is rescaled so that the boundary fluxes balance (what
    balance_boundary_fluxes does for the env)."""
    from fluidgym_amd.envs.cylinder_grid import make_vortex_street_mesh

    m = make_vortex_street_mesh(res)
    F = {"-x": 0, "+x": 1, "-y": 2, "+y": 3}
    s = Spec(2, nu)
    s.blocks = [c.astype(np.float64) for c in m.coords]
    rng = np.random.default_rng(16)
    s.fixed = []
    for (b, f), v in m.fixed.items():
        face_cells = s.blocks[b].shape[2 if F[f] >= 2 else 1] - 1
        v = np.broadcast_to(np.asarray(v, np.float64).reshape(2, -1), (2, face_cells))
        s.fixed.append((b, F[f], v + noise * rng.standard_normal(v.shape)))
    s.connections = [(c[0], F[c[1]], c[2], F[c[3]], F[c[4]]) for c in m.connections]
    fl = face_fluxes(s.oracle())
    # the face with the largest outward flux is the outflow: scale it so that everything balances
    out_key = max(fl, key=lambda k: fl[k])
    k = -sum(v for key, v in fl.items() if key != out_key) / fl[out_key]
    s.fixed = [(b, f, v * k if (b, f) == out_key else v) for b, f, v in s.fixed]
    return s


def face_fluxes(d):
    """Outward contravariant flux through every prescribed face of an oracle domain: {(block, face): flux}."""
    out = {}
    for b, pos in d.cells():
        for f in range(2 * d.d):
            if d.at_bound(b, pos, f) and d.is_empty(b, f):
                out[(b, f)] = out.get((b, f), 0.0) + (1.0 if f & 1 else -1.0) * d.contra_bound(b, f, pos, f >> 1)
    return out


def airfoil_spec(div=4, nu=1e-3, noise=0.1, balanced=False):
    """The airfoil envs' six-block C-mesh (envs/airfoil/grid.py:247-716 in the reference) at ``resolution_div = 4`` (4 110 cells,
    small enough for the per-cell oracle): the mesh with cells of 1e-4 of the typical area where the blocks meet at the nose,
    the one whose pressure matrix is visibly non-symmetric."""
    from fluidgym_amd.envs.airfoil_grid import make_airfoil_mesh

    m = make_airfoil_mesh(resolution_div=div, attack_angle_deg=10.0)
    F = {"-x": 0, "+x": 1, "-y": 2, "+y": 3}
    s = Spec(2, nu)
    s.blocks = [c.astype(np.float64) for c in m.coords]
    rng = np.random.default_rng(9)
    s.fixed = []
    for (b, f), v in m.fixed.items():
        face_cells = s.blocks[b].shape[2 if F[f] >= 2 else 1] - 1
        v = np.broadcast_to(np.asarray(v, np.float64).reshape(2, -1), (2, face_cells))
        s.fixed.append((b, F[f], v + noise * rng.standard_normal(v.shape)))
    s.connections = [(b1, F[f1], b2, F[f2], F[ax]) for b1, f1, b2, f2, ax in m.connections]
    if balanced:
        # what balance_boundary_fluxes does for the env (PISOtorch_simulation.py:188-226): the two outflow faces (tail blocks,
        # +x) are scaled so that the boundary fluxes sum to zero -- the pressure system is then consistent
        fl = face_fluxes(s.oracle())
        outflow = [(4, 1), (5, 1)]
        k = -sum(v for key, v in fl.items() if key not in outflow) / sum(fl[key] for key in outflow)
        s.fixed = [(b, f, v * k if (b, f) in outflow else v) for b, f, v in s.fixed]
    return s
