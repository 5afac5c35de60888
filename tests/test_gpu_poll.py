"""Host polls on pinned sequence words (csrc/fg_poll.hip, FgPoll): the convergence checks, the flux balance / CFL maximum of
fg_single_step and the multi-block checks publish a sequence number behind their host-pinned results and the host spins on it
instead of calling hipStreamSynchronize.  The wait must not change a single bit of a run: env steps with FG_POLL_SPIN=0 (read at
fg_create / fg_mb_create: the stream synchronisation of rounds 1-3) against the default."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(env_id, monkeypatch, spin, steps=2, **kw):
    import fluidgym_amd

    if spin:
        monkeypatch.delenv("FG_POLL_SPIN", raising=False)
    else:
        monkeypatch.setenv("FG_POLL_SPIN", "0")
    env = fluidgym_amd.make(env_id, **kw)
    try:
        env.seed(3)
        obs, _ = env.reset(seed=11)
        gen = torch.Generator().manual_seed(5)
        out = [{k: v.clone() for k, v in obs.items()}]
        rewards = []
        for _ in range(steps):
            a = env.sample_action() * 0 + (torch.rand(tuple(env.sample_action().shape), generator=gen) * 2 - 1).to(env.cuda_device)
            obs, r, _, _, _ = env.step(a)
            out.append({k: v.clone() for k, v in obs.items()})
            rewards.append(torch.as_tensor(r).clone())
        return out, rewards
    finally:
        env.close()


@pytest.mark.parametrize("env_id,kw", [("ChannelJet2D-v0", dict(num_envs=3)),
                                       ("RBC2D-baseline-v0", dict(num_envs=2)),
                                       ("CylinderJet2D-easy-v0", dict(num_envs=2, initial_domain_steps=5, randomize_initial_state=False))])
def test_spinning_polls_leave_every_bit_of_a_run_where_it_was(env_id, kw, monkeypatch):
    obs_spin, r_spin = _run(env_id, monkeypatch, True, **kw)
    obs_sync, r_sync = _run(env_id, monkeypatch, False, **kw)
    for a, b in zip(obs_spin, obs_sync):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    for a, b in zip(r_spin, r_sync):
        assert torch.equal(a, b)
    assert any(float(o[k].abs().max()) > 0 for o in obs_spin for k in o)


def _run_switch(env_id, monkeypatch, name, value, steps=3, **kw):
    monkeypatch.delenv("FG_POLL_SPIN", raising=False)
    if value is None:
        monkeypatch.delenv(name, raising=False)
    else:
        monkeypatch.setenv(name, value)
    try:
        return _run(env_id, monkeypatch, True, steps=steps, **kw)
    finally:
        monkeypatch.delenv(name, raising=False)


@pytest.mark.parametrize("env_id,kw", [("ChannelJet2D-v0", dict(num_envs=3)), ("RBC2D-baseline-v0", dict(num_envs=2)),
                                       ("TCFSmall3D-both-easy-v0", dict(num_envs=2, resolution_x_z=16, resolution_y=16, step_length=0.6, use_marl=False))])
@pytest.mark.parametrize("switch", ["FG_POLL_WORDS", "FG_DEV_DT"])
def test_result_words_and_the_device_side_sub_step_leave_every_bit_where_it_was(env_id, kw, switch, monkeypatch):
    """Round 6: (a) the polled verdicts, CFL maxima and flux balances travel IN the 8-byte words the host spins on (FgPollOut::gran)
    instead of through a host-pinned mirror behind a system-scope release -- the same values, unpacked to where the mirror form left them
    (FG_POLL_WORDS=0 = the mirror form); (b) the adaptive sub-step is taken on the device by the CFL kernel, in the doubles the host
    uses, and the host reads the maxima after the PISO step (FgDtRule; FG_DEV_DT=0 = before it).  Neither may move a bit of a run
    (read at fg_create): env steps with random actions, adaptive sub-steps that differ per env."""
    on = _run_switch(env_id, monkeypatch, switch, None, **kw)
    off = _run_switch(env_id, monkeypatch, switch, "0", **kw)
    for a, b in zip(on[0], off[0]):
        for k in a:
            assert torch.equal(a[k], b[k]), (switch, k)
    for a, b in zip(on[1], off[1]):
        assert torch.equal(a, b)


def test_native_wall_stress_forcing_is_the_python_hook(monkeypatch):
    """Policy ``native_wall_forcing`` (round 4): the turbulent-channel env's PRE hook -- G_x = mean of the two wall shear stresses
    written into the block's velocity source (tcf_env.py, grid.py:147-176) -- runs natively (``fg_set_wall_stress_forcing``: a
    uniform body force per env recomputed before every PISO step, the whole single_step inside fg_single_step).  Held against the
    interpreter's hook on the same env over two env steps: the forcing value is a mean taken in a different order (fp64 sums
    against torch's fp32 tree), so fields agree to rounding, not to the bit."""
    import fluidgym_amd

    kw = dict(num_envs=2, randomize_initial_state=False, resolution_x_z=16, resolution_y=16, step_length=0.6, use_marl=False)
    outs = {}
    for native in (True, False):
        old = fluidgym_amd.set_solver_policy(native_wall_forcing=native)
        try:
            env = fluidgym_amd.make("TCFSmall3D-both-easy-v0", **kw)
            env.reset(seed=3)                                    # (the domain is built here: the policy is read then)
        finally:
            fluidgym_amd.set_solver_policy(**old)
        try:
            assert env._native_forcing is native
            assert env._sim._native_ok() is native               # (the Python hook keeps the step in the interpreter)
            assert (env._block.velocitySource is None) == native
            a = torch.zeros_like(env.sample_action())
            for _ in range(2):
                _, r, _, _, info = env.step(a)
            outs[native] = (env._block.velocity.clone(), env._block.pressure.clone(), torch.as_tensor(info["wall_stress"]).clone())
        finally:
            env.close()
    (u_n, p_n, tau_n), (u_p, p_p, tau_p) = outs[True], outs[False]
    scale = float(u_p.abs().max())
    assert float((u_n - u_p).abs().max()) < 2e-5 * scale, float((u_n - u_p).abs().max()) / scale
    assert float((p_n - p_p).abs().max()) < 2e-4 * float(p_p.abs().max()) + 1e-7
    assert torch.allclose(tau_n.float(), tau_p.float(), rtol=2e-5, atol=1e-9)
    assert float(tau_p.abs().max()) > 0


def test_native_wall_forcing_also_drives_the_hook_by_hook_path():
    """with a hook that keeps the step out of the fused driver (here a do-nothing POST_PREDICTION hook) the step goes
    through the stepwise entry points -- fg_setup_advection must then compute the wall-stress body force itself (before round 5
    it bound force_uniform as the last fused step had left it: zero or stale, silently)."""
    import fluidgym_amd

    kw = dict(num_envs=2, randomize_initial_state=False, resolution_x_z=16, resolution_y=16, step_length=0.6, use_marl=False)
    outs = {}
    for hooked in (False, True):
        env = fluidgym_amd.make("TCFSmall3D-both-easy-v0", **kw)
        env.reset(seed=3)
        try:
            assert env._native_forcing is True
            calls = []
            if hooked:
                env._sim.prep_fn = {**env._sim.prep_fn, "POST_PREDICTION": [lambda *a, **k: calls.append(1)]}
                assert not env._sim._fused_ok()
            a = torch.zeros_like(env.sample_action())
            for _ in range(2):
                _, r, _, _, info = env.step(a)
            assert bool(calls) == hooked
            outs[hooked] = (env._block.velocity.clone(), torch.as_tensor(info["wall_stress"]).clone())
        finally:
            env.close()
    (u_f, tau_f), (u_h, tau_h) = outs[False], outs[True]
    scale = float(u_f.abs().max())
    assert float((u_h - u_f).abs().max()) < 2e-5 * scale, float((u_h - u_f).abs().max()) / scale
    assert torch.allclose(tau_h.float(), tau_f.float(), rtol=2e-5, atol=1e-9) and float(tau_f.abs().max()) > 0


def test_compacted_krylov_launches_leave_every_bit_where_it_was(monkeypatch):
    """Round 4 (MbSolve::sys_map, mb_bicgstab): while only a few systems of a batch still iterate, the kernels of an iteration are
    launched over those systems only.  The per-system arithmetic does not change, so an airfoil batch whose envs are driven apart
    by different actions (their pressure solves end at different iterations: the compaction does engage) steps to the same bits
    with FG_MB_COMPACT=0 (every launch over all systems, read at fg_mb_create)."""
    import fluidgym_amd

    def run(compact):
        if compact:
            monkeypatch.delenv("FG_MB_COMPACT", raising=False)
        else:
            monkeypatch.setenv("FG_MB_COMPACT", "0")
        env = fluidgym_amd.make("Airfoil2D-easy-v0", num_envs=8, initial_domain_steps=6, randomize_initial_state=False, episode_length=3, resolution_div=2)
        try:
            env.reset(seed=0)
            gen = torch.Generator().manual_seed(9)
            for _ in range(2):
                a = (torch.rand(tuple(env.sample_action().shape), generator=gen) * 2 - 1).to(env.cuda_device)
                _, r, _, _, info = env.step(a)
            c = env._domain.solver_counters()
            return env._domain.velocity.clone(), env._domain.pressure.clone(), torch.as_tensor(r).clone(), c["pressure0"]
        finally:
            env.close()

    u1, p1, r1, c1 = run(True)
    u0, p0, r0, c0 = run(False)
    assert torch.equal(u1, u0) and torch.equal(p1, p0) and torch.equal(r1, r0)
    assert c1["mean"] == c0["mean"] and c1["max"] == c0["max"]
    assert c1["max"] > 1.3 * c1["mean"], c1                      # the solves of the batch do end at different iterations
    assert float((u1[0] - u1[-1]).abs().max()) > 0               # and the envs differ


def test_multi_step_is_the_loop_of_single_steps():
    """fg_multi_step (the sim steps of an env step in one native call, boundary slices bound per step) against the same steps issued one
    by one from Python: every bit of velocity, pressure, observations and rewards; then the env-level replay through get_state."""
    import fluidgym_amd

    def run(loop):
        env = fluidgym_amd.make("ChannelJet2D-v0", num_envs=3)
        env.reset(seed=21)
        if loop:      # the interpreter's loop: what Simulation.multi_step falls back to without the native entry point
            sim = env._sim
            sim.multi_step = lambda n, sched=None: all(
                [[sim._solver.set_boundary_velocity(f, t[k]) for f, t in (sched or {}).items()] and False or sim.single_step() for k in range(n)])
        g = torch.Generator(device="cpu").manual_seed(5)
        out = []
        for _ in range(2):
            obs, r, _, _, info = env.step((torch.rand(3, 1, generator=g) * 2 - 1).cuda())
            out.append((obs["velocity"].clone(), obs["pressure"].clone(), r.clone()))
        ns = env._domain.solver
        res = (ns.velocity.clone(), ns.pressure.clone(), out, env._sim.total_step)
        env.close()
        return res

    a, b = run(False), run(True)
    assert a[3] == b[3] >= 50      # (the reset's random warm-up steps + 2 x 25)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for (v1, p1, r1), (v2, p2, r2) in zip(a[2], b[2]):
        assert torch.equal(v1, v2) and torch.equal(p1, p2) and torch.equal(r1, r2)
