"""Per-cell viscosity of the velocity system (``Block.setViscosity``; getViscosityBlock, PISO_multiblock_cuda_kernel.cu:1816-1837,
3698-3745, 4342) and the Smagorinsky sub-grid model (``SGSviscosityIncompressibleSmagorinsky``, :6913-6966; gradients :2997-3040)
behind the TCF env's ``C_smag`` / ``use_van_driest`` options (tcf_env.py:441-474): the HIP kernels against the oracle's
restatement, and the env option end to end."""
import numpy as np
import pytest
import torch

from oracle import piso_oracle as O
from tests.helpers import make_case, rel_err

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("dims,n,fixed_axes", [(2, (16, 12), (1,)), (2, (12, 10), (0, 1)), (3, (8, 6, 5), (1,)), (3, (6, 5, 4), (0, 1, 2)),
                                               (2, (12, 8), ())])
def test_smagorinsky_viscosity_matches_the_oracle(dims, n, fixed_axes):
    case = make_case(dims=dims, n=n, fixed_axes=fixed_axes, B=2, seed=7, nu=0.02, vel_scale=0.6)
    ns = case.native()
    nu_t = _np(ns.sgs_smagorinsky(0.17))
    ns.close()
    g = case.grid()
    for b in range(case.B):
        ref = O.sgs_smagorinsky(case.oracle_domain(b, g), 0.17)
        assert ref.max() > 1e-4
        assert rel_err(nu_t[b], ref) < 2e-5, (b, rel_err(nu_t[b], ref))


@pytest.mark.parametrize("dims,n,fixed_axes", [(2, (16, 12), (1,)), (2, (12, 10), (0, 1)), (3, (8, 6, 5), (1,)), (2, (64, 48), (1,))])
def test_per_cell_viscosity_enters_matrix_and_right_hand_side_like_the_oracle(dims, n, fixed_axes):
    case = make_case(dims=dims, n=n, fixed_axes=fixed_axes, B=2, seed=3, nu=0.02, vel_scale=0.4)
    dt = 0.05
    rng = np.random.default_rng(5)
    fields = [0.02 * (1.0 + 4.0 * rng.random(case.shape)) for _ in range(case.B)]       # 0.02 .. 0.1, rough
    ns = case.native()
    ns.set_viscosity_field(torch.as_tensor(np.stack(fields), dtype=torch.float32, device="cuda").contiguous())
    ns.set_advection_start(False)
    ns.setup_advection(dt)
    A = _np(ns.buffer(0, (case.B,) + case.shape))
    off = _np(ns.buffer(1, (case.B, 2 * dims) + case.shape))
    rhs = _np(ns.buffer(2, (case.B, dims) + case.shape))
    info = ns.solve_advection(tol=1e-7)
    assert all(i.converged for i in info)
    x = _np(ns.buffer(3, (case.B, dims) + case.shape))
    # and back to the global viscosity: the scalar-nu kernel again
    ns.set_viscosity_field(None)
    ns.setup_advection(dt)
    A_plain = _np(ns.buffer(0, (case.B,) + case.shape))
    ns.close()
    g = case.grid()
    for b in range(case.B):
        dom = case.oracle_domain(b, g)
        C0, A0, _ = O.build_advection_matrix(dom, dt)
        assert rel_err(A_plain[b], A0) < 1e-5
        dom.viscosity_field = fields[b]
        C, A_ref, offs = O.build_advection_matrix(dom, dt)
        assert rel_err(A_ref, A0) > 1e-2                       # the field matters
        assert rel_err(A[b], A_ref) < 1e-5
        for f in range(2 * dims):
            assert np.abs(off[b, f] - offs[f]).max() < 1e-5 * np.abs(A_ref).max()
        rhs_ref = O.advection_rhs_velocity(dom, dt)
        assert rel_err(rhs[b], rhs_ref) < 2e-5
        for comp in range(dims):
            assert rel_err(x[b, comp], O.solve_direct(C, rhs_ref[comp].ravel()).reshape(case.shape)) < 3e-5


def test_tcf_env_runs_with_the_smagorinsky_model_and_van_driest_damping():
    """``make("TCFSmall3D-both-easy-v0", C_smag=0.1, use_van_driest=True)``: the PRE hook binds nu + C Delta^2 |S| (damped towards the walls) every
    substep; the env steps, the bound field is >= nu everywhere and > nu somewhere, and the step differs from the C_smag = 0 env's."""
    import fluidgym_amd

    out = {}
    for c_smag in (0.0, 0.1):
        env = fluidgym_amd.make("TCFSmall3D-both-easy-v0", num_envs=2, resolution_x_z=16, resolution_y=16, step_length=0.6, use_marl=False,
                                randomize_initial_state=False, C_smag=c_smag, use_van_driest=c_smag != 0.0)
        env.reset(seed=2)
        obs, rew, _, _, _ = env.step(torch.zeros_like(env.sample_action()))
        assert all(torch.isfinite(v).all() for v in obs.values()) and torch.isfinite(rew).all()
        solver = env._domain.solver
        if c_smag:
            visc = solver.viscosity_field
            assert visc is not None and float(visc.min()) >= env._nu * (1 - 1e-6) and float(visc.max()) > 1.05 * env._nu
            # damping: the wall-adjacent rows carry (almost) no eddy viscosity
            assert float((visc[:, :, 0] - env._nu).max()) < 0.2 * float((visc - env._nu).max())
        else:
            assert solver.viscosity_field is None
        out[c_smag] = solver.velocity.clone()
        env.close()
    assert float((out[0.1] - out[0.0]).abs().max()) > 1e-6
