"""Round 6: the pressure CG of one env spread over a cluster of four workgroups (``k_mbc_cluster``, csrc/fg_mb_cluster.hip)
against the one-workgroup-per-env kernels it replaces (``k_mbc_onchip`` / ``k_mbc_l2``, FG_MB_CLUSTER=0) on the reference's
cylinder meshes: the same preconditioned recurrence with the operator applied to ``z`` (``s = P p`` by linearity) and another
summation order, so iteration counts differ by a few per cent (a hundred iterations of fp32 CG on a non-symmetric matrix) and the projected velocities agree to what two Krylov
trajectories at that tolerance do.  Reference role: ``cgSolveGPU`` (cg_solver_kernel.cu:129-471) at the tolerance of
``cylinder_env_base.py:315``."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _run(monkeypatch, mode, resolution, batch=3, maxcl=None, half=None, tol=1e-7):
    from fluidgym_amd.envs.cylinder_grid import build_domain, make_vortex_street_mesh

    monkeypatch.setenv("FG_MB_CLUSTER", mode)      # read once per handle at fg_mb_create
    if maxcl is None:
        monkeypatch.delenv("FG_MB_CL_MAXCL", raising=False)
    else:
        monkeypatch.setenv("FG_MB_CL_MAXCL", str(maxcl))
    if half is None:
        monkeypatch.delenv("FG_MB_CL_HALF", raising=False)
    else:
        monkeypatch.setenv("FG_MB_CL_HALF", str(half))
    mesh = make_vortex_street_mesh(resolution)
    dom = build_domain(mesh, 0.01, batch=batch)
    dom.set_stall_limit(5000)
    assert dom.set_pressure_multilevel() is not None
    g = torch.Generator(device="cpu").manual_seed(11)
    v0 = 0.3 * torch.randn((1,) + tuple(dom.velocity.shape[1:]), generator=g)
    dom.velocity.copy_(v0.expand_as(dom.velocity).to(dom.device))      # every env the same state
    dom.velocity[:, 0] += 1.0
    dom.solver_counters(reset=True)
    dom.make_divergence_free(pressure_tol=tol, max_iterations=5000, pressure_project_mean=True)
    u0 = dom.velocity.cpu().numpy().copy()
    dts = [0.01] * batch
    dts[-1] = 0.0                                                          # the last env sits out
    for _ in range(2):
        dom.piso_step(dts, pressure_tol=tol, advection_tol=tol, pressure_project_mean=True)
    # solves started from an iterate (second non-orthogonal pass, warm-start policy): the kernel's residual pass
    dom.piso_step(dts, pressure_tol=tol, advection_tol=tol, pressure_project_mean=True, pressure_non_ortho_steps=2, pressure_warm_start=True)
    cfg = dom.config_dump()
    out = (u0, dom.velocity.cpu().numpy().copy(), dom.pressure.cpu().numpy().copy(), dom.solver_counters(), cfg)
    dom.close()
    return out


@pytest.mark.parametrize("resolution,half", [(24, None), (32, None), (32, 0)])
def test_cluster_cg_is_the_one_workgroup_cg(monkeypatch, resolution, half):
    """14 232 cells (eight members per thread in 512 threads, the workgroup's rows of the coarse inverse in LDS as fp32) and 23 424
    cells (768 threads, the rows as fp16 -- or, FG_MB_CL_HALF=0, streamed from L2 as fp32): start from zero, start from an iterate,
    an inactive env; identical envs stay bit-identical."""
    u0_c, u_c, p_c, c_c, cfg_c = _run(monkeypatch, "0", resolution)
    u0_k, u_k, p_k, c_k, cfg_k = _run(monkeypatch, "1", resolution, half=half)
    assert cfg_c["cluster_solves"] == 0 and cfg_k["cluster_on"] == 1 and cfg_k["cluster_solves"] > 0 and cfg_k["cluster_fallbacks"] == 0, (cfg_c, cfg_k)
    assert np.isfinite(u_k).all() and np.isfinite(p_k).all()
    assert _rel(u0_k, u0_c) < 2e-5 and _rel(u_k, u_c) < 5e-5, (_rel(u0_k, u0_c), _rel(u_k, u_c))
    for k in ("pressure0", "pressure1"):
        assert abs(c_k[k]["mean"] - c_c[k]["mean"]) <= max(2.0, 0.1 * c_c[k]["mean"]) and c_k[k]["unconverged"] == 0, (c_c, c_k)
    np.testing.assert_array_equal(u_k[2], u0_k[2])   # the inactive env is untouched
    np.testing.assert_array_equal(u_k[0], u_k[1])    # identical envs, identical bits
    np.testing.assert_array_equal(p_k[0], p_k[1])


def test_envs_beyond_the_resident_clusters_queue_inside_the_launch(monkeypatch):
    """A launch holds at most one workgroup per CU (every workgroup a cluster polls must be resident); envs beyond that are taken
    by the clusters in turn.  With the clusters of a launch capped at two, five envs run as 2 + 2 + 1 -- same bits as uncapped."""
    a = _run(monkeypatch, "1", 24, batch=5)
    b = _run(monkeypatch, "1", 24, batch=5, maxcl=2)
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])
    assert b[4]["cluster_fallbacks"] == 0


def test_small_meshes_keep_the_one_workgroup_kernels_unless_forced(monkeypatch):
    """Below 8 k cells a single workgroup with four / eight cells per thread is faster: the default policy leaves such meshes to
    it; FG_MB_CLUSTER=2 runs the cluster on every mesh its tables fit (here 3 558 cells) and gives the same step."""
    u0_c, u_c, p_c, c_c, cfg_c = _run(monkeypatch, "1", 12)
    assert cfg_c["cluster_on"] == 0 and cfg_c["cluster_solves"] == 0
    u0_k, u_k, p_k, c_k, cfg_k = _run(monkeypatch, "2", 12)
    assert cfg_k["cluster_on"] == 1 and cfg_k["cluster_solves"] > 0 and cfg_k["cluster_fallbacks"] == 0
    assert _rel(u0_k, u0_c) < 2e-5 and _rel(u_k, u_c) < 5e-5, (_rel(u0_k, u0_c), _rel(u_k, u_c))
    for k in ("pressure0", "pressure1"):
        assert abs(c_k[k]["mean"] - c_c[k]["mean"]) <= max(2.0, 0.1 * c_c[k]["mean"]) and c_k[k]["unconverged"] == 0, (c_c, c_k)


def test_cluster_cg_at_the_bench_batch(monkeypatch):
    """64 envs = 256 workgroups, one per CU: every env converged, every env the bits of env 0."""
    u0, u, p, c, cfg = _run(monkeypatch, "1", 24, batch=64, tol=1e-5)
    assert cfg["cluster_solves"] > 0 and cfg["cluster_fallbacks"] == 0
    for k in ("pressure0", "pressure1"):
        assert c[k]["unconverged"] == 0
    for b in range(1, 63):
        np.testing.assert_array_equal(u[b], u[0])


def _sweeps_run(monkeypatch, cl_jacobi, env_id, steps=2, B=3, **kw):
    import fluidgym_amd

    monkeypatch.setenv("FG_MB_CL_JACOBI", cl_jacobi)
    env = fluidgym_amd.make(env_id, num_envs=B, randomize_initial_state=False, **kw)
    env.reset(seed=0)
    g = torch.Generator(device="cpu").manual_seed(4)
    na = tuple(env._zero_action.shape[1:])
    for _ in range(steps):
        env.step((torch.rand((B,) + na, generator=g) * 2 - 1).cuda())
    dom = env._domain
    out = (dom.velocity.cpu().numpy().copy(), dom.pressure.cpu().numpy().copy(), dom.advection_jacobi_counts(), dom.solver_counters(), dom.config_dump())
    env.close()
    return out


@pytest.mark.parametrize("env_id", ["CylinderJet2D-easy-v0", "CylinderJet2D-medium-v0"])
def test_cluster_sweeps_are_the_launched_sweeps(monkeypatch, env_id):
    """The velocity systems' Jacobi sweeps by the clusters (``k_mbj_cluster``, one launch per solve) against one launch per sweep
    (``k_mbj_sweep_env``, FG_MB_CL_JACOBI=0): the same arithmetic per cell, the same check points, the same verdict per system --
    only the residual sums the verdict reads are added in another order, so a system whose residual sits at the tolerance at a check
    point may stop one check point apart (seen: one system of 502 on the medium mesh).  Two env steps (50 PISO steps) with random
    jets: every solve settled by the sweeps either way, the same sweeps per solve to a hundredth, fields equal to what two sweeps more
    or less leave."""
    a = _sweeps_run(monkeypatch, "1", env_id, initial_domain_steps=20)
    b = _sweeps_run(monkeypatch, "0", env_id, initial_domain_steps=20)
    assert a[4]["cluster_jacobi_solves"] >= 50 and b[4]["cluster_jacobi_solves"] == 0 and a[4]["cluster_fallbacks"] == 0
    assert a[2] == b[2] and a[2]["handed_to_bicgstab"] == 0, (a[2], b[2])
    va, vb = a[3]["velocity"], b[3]["velocity"]
    assert va["systems"] == vb["systems"] and va["unconverged"] == vb["unconverged"] == 0 and abs(va["mean"] - vb["mean"]) < 0.05, (va, vb)
    assert _rel(a[0], b[0]) < 5e-5 and _rel(a[1], b[1]) < 2e-3, (_rel(a[0], b[0]), _rel(a[1], b[1]))
