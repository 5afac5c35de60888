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
