"""GPU tests (-m gpu) that close the configuration gaps of BASELINE.json on one MI355X:
  * config 2 (KManipSoloArm @ 4096) at full width with an oracle slice over a whole episode + the auto-reset,
  * config 4's LAST shard (KManipTorso, env ids 57344..65535 = rank 7 of 8) against the oracle, and the whole 65536-env
    job on ONE handle (determinism + invariants across the auto-reset),
  * a wide one-step parity soak (512 envs x 70 steps x 3 models, oracle re-synchronised every step),
  * the counter-based action stream of SURVEY 8d (kmanip_sample_action == the oracle's draw, bit for bit).
Tolerances as in test_gpu_parity.py (float64 both sides; done / masks / counters / float32 actions bit-exact)."""
import numpy as np
import pytest

from conftest import ENVS3
from gym_kmanip_amd.model import KM_DONE_DIVERGED, KM_DONE_TRUNCATED, compile_model

pytestmark = pytest.mark.gpu

TOL_Q, TOL_V, TOL_R = 1e-7, 1e-5, 1e-6


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.mark.parametrize("env", ENVS3)
def test_sample_action_matches_oracle(env):
    """action_space.sample() on the device (Philox keyed (seed; global env id, episode, step)) == the oracle's draw, bit for
    bit: after the reset, `ahead` steps into the future (across the episode boundary), after stepping, and for a shard created
    at an env_id_offset; values are float32 in [-1, 1)."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from oracle.oracle import Oracle
    cm = compile_model(env, auto_reset=True)
    n, off = 96, 5000
    dev = env_hip.KManipEnvHip(cm, num_envs=n, seed=21, env_id_offset=off)
    orc = Oracle(cm, n, seed=21, env_id_offset=off)
    dev.k_reset(); orc.reset()
    stagger = (np.arange(n) % 64).astype(np.int32)
    dev.set_state(step=stagger); orc.set_state(step=stagger)
    seen = []
    for ahead in (0, 1, 63, 64, 130):
        a = dev.sample_action(ahead=ahead).cpu().numpy()
        assert a.dtype == np.float32 and np.array_equal(a, orc.sample_action(ahead)), ahead
        assert (a >= -1).all() and (a < 1).all()
        seen.append(a)
    assert not np.array_equal(seen[0], seen[1]) and not np.array_equal(seen[0], seen[3])      # fresh per step and per episode
    for k in range(3):                                                                       # the stream follows the env's own counters
        a = dev.sample_action()
        dev.step_flat(a)
        ao = orc.sample_action()
        assert np.array_equal(a.cpu().numpy(), ao), k
        orc.step(ao)
    # a shard of the job draws what the whole job draws for the same global env ids
    part = env_hip.KManipEnvHip(cm, num_envs=16, seed=21, env_id_offset=off + 32)
    part.k_reset(); part.set_state(step=stagger[32:48] + 3)
    assert np.array_equal(part.sample_action().cpu().numpy(), dev.sample_action().cpu().numpy()[32:48])
    assert abs(float(seen[0].mean())) < 0.1 and 0.5 < float(seen[0].std()) < 0.65                # U[-1, 1): mean 0, sd 0.577
    dev.k_close(); part.k_close()


def test_config2_full_width_with_oracle_slice():
    """BASELINE config 2 (KManipSoloArm @ 4096, the headline) for 70 control steps on the bench's own action stream: a 16-env
    oracle slice of the full batch every step (obs / reward / done; contact masks and state at checkpoints), bitwise
    determinism against a second handle, invariants."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from oracle.oracle import Oracle
    n, lo, no = 4096, 2048, 16
    a = env_hip.make("KManipSoloArm", num_envs=n, seed=3)
    b = env_hip.make("KManipSoloArm", num_envs=n, seed=3)
    orc = Oracle(a.cm, no, seed=3, env_id_offset=lo)
    a.k_reset(); b.k_reset(); orc.reset()
    nl = a.cm.nlink
    saw_contact = False
    for k in range(70):
        act = a.sample_action()
        assert np.array_equal(act[lo:lo + no].cpu().numpy(), orc.sample_action()), k
        a.step_flat(act); b.step_flat(act.clone())
        oo, ro, do = orc.step(act[lo:lo + no].cpu().numpy())
        assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done), k
        assert np.abs(a.obs[lo:lo + no].cpu().numpy() - oo).max() < TOL_Q, k
        assert np.abs(a.reward[lo:lo + no].cpu().numpy() - ro).max() < TOL_R, k
        assert np.array_equal(a.done[lo:lo + no].cpu().numpy(), do), k
        done = a.done.cpu().numpy()
        assert not (done & KM_DONE_DIVERGED).any() and (done == (KM_DONE_TRUNCATED if k == 63 else 0)).all(), k
        if k in (20, 40, 62, 69):
            mg = a.get_diag()[0]
            assert np.array_equal(mg[lo:lo + no], orc.get_diag()[0]), k
            saw_contact |= bool(mg.any())
    sa, so = a.get_state(), orc.get_state()
    assert np.abs(sa[0][lo:lo + no] - so[0]).max() < TOL_Q and np.abs(sa[1][lo:lo + no] - so[1]).max() < TOL_V
    assert np.array_equal(sa[2][lo:lo + no], so[2]) and (sa[4] == 6).all() and saw_contact
    assert np.abs(np.linalg.norm(sa[0][:, nl + 3:], axis=1) - 1).max() < 1e-12
    obs = a.obs.cpu().numpy()
    assert np.isfinite(obs).all() and (np.abs(obs) <= 1).all()
    a.k_close(); b.k_close()


def test_config4_last_shard_vs_oracle():
    """Rank 7 of BASELINE config 4 (KManipTorso, 65536 envs over 8 ranks): env ids 57344..65535 on one handle, 66 steps across
    the auto-reset, with an oracle on the first and the last 8 env ids of the shard (incl. 65535, the job's last env)."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from oracle.oracle import Oracle
    n, off = 8192, 57344
    a = env_hip.make("KManipTorso", num_envs=n, seed=3, env_id_offset=off)
    o_lo = Oracle(a.cm, 8, seed=3, env_id_offset=off)
    o_hi = Oracle(a.cm, 8, seed=3, env_id_offset=off + n - 8)
    a.k_reset(); o_lo.reset(); o_hi.reset()
    for k in range(66):
        act = a.sample_action()
        an = act.cpu().numpy()
        assert np.array_equal(an[:8], o_lo.sample_action()) and np.array_equal(an[-8:], o_hi.sample_action()), k
        a.step_flat(act)
        obs, rew, done = a.obs.cpu().numpy(), a.reward.cpu().numpy(), a.done.cpu().numpy()
        for sl, orc in ((slice(0, 8), o_lo), (slice(n - 8, n), o_hi)):
            oo, ro, do = orc.step(an[sl])
            assert np.abs(obs[sl] - oo).max() < TOL_Q and np.abs(rew[sl] - ro).max() < TOL_R and np.array_equal(done[sl], do), k
        assert (done == (KM_DONE_TRUNCATED if k == 63 else 0)).all(), k
    sa = a.get_state()
    for sl, orc in ((slice(0, 8), o_lo), (slice(n - 8, n), o_hi)):
        so = orc.get_state()
        assert np.abs(sa[0][sl] - so[0]).max() < TOL_Q and np.abs(sa[1][sl] - so[1]).max() < TOL_V and np.array_equal(sa[2][sl], so[2])
    a.k_close()


def test_config4_whole_job_on_one_handle():
    """All 65536 KManipTorso envs of BASELINE config 4 on ONE handle (the 8-rank run shards exactly these env ids): 66 control
    steps across the auto-reset, two handles bit for bit, the 8192-env shard at rank 7's offset reproducing its slice of the
    whole job bit for bit (what a rank computes does not depend on the shard layout), invariants."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    n, ns, off = 65536, 8192, 57344
    a = env_hip.make("KManipTorso", num_envs=n, seed=3)
    b = env_hip.make("KManipTorso", num_envs=n, seed=3)
    c = env_hip.make("KManipTorso", num_envs=ns, seed=3, env_id_offset=off)
    a.k_reset(); b.k_reset(); c.k_reset()
    nl = a.cm.nlink
    for k in range(66):
        act = a.sample_action()
        assert torch.equal(act[off:off + ns], c.sample_action()), k
        a.step_flat(act); b.step_flat(act.clone()); c.step_flat(act[off:off + ns].contiguous())
        if k % 8 == 0 or k >= 62:
            assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done), k
            assert torch.equal(a.obs[off:off + ns], c.obs) and torch.equal(a.done[off:off + ns], c.done), k
            done = a.done.cpu().numpy()
            assert (done == (KM_DONE_TRUNCATED if k == 63 else 0)).all(), k
    sa, sb, sc = a.get_state(), b.get_state(), c.get_state()
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)
    for x, z in zip(sa, sc):
        assert np.array_equal(x[off:off + ns], z)
    assert (sa[4] == 2).all()
    assert np.abs(np.linalg.norm(sa[0][:, nl + 3:], axis=1) - 1).max() < 1e-12
    obs = a.obs.cpu().numpy()
    assert np.isfinite(obs).all() and (np.abs(obs) <= 1).all()
    assert bool(a.get_diag()[0].any())                                   # contacts are live at this point of the episode
    for e in (a, b, c):
        e.k_close()


@pytest.mark.parametrize("env", ENVS3)
def test_parity_soak_one_step_samples(env):
    """4096 envs x 130 control steps (two episodes per env) with desynchronised episode phases, on the counter-based action stream,
    the oracle re-synchronised to the device state after every step: 532 480 independent one-step parity samples per model (a
    free-running comparison measures the chaos of a cube rocking on a stiff contact, not the kernel).  Contact masks and done bytes
    identical in every sample; float32 ctrl identical except for rounding-boundary flips of one ulp (bounded); IK evaluation counts
    at most one apart.  Two bars on the envs whose ctrl agrees (round 5; the round-4 bar, 1e-7 on qvel for every sample, passed by
    sample size: profiles/r04_parity_soak_8192x400.txt found 2.7e-7 among 3.3 M Torso samples):
      * EVERY sample inside the IK-derived bar.  The IK leaves qpos at its last evaluated point (ik_mujoco.py:34,67) in float64;
        the device's Cholesky-based and the oracle's SVD-based trust-region solves agree to the suite's IK-vs-oracle bar, 1e-7 rad,
        and a position offset d on a kp = 1000, I = 0.01 servo joint is a velocity of up to 400 d / s within the control step:
        |dqpos| < 1e-7, |dqvel| < 4e-5, |dreward| < 1e-7.
      * the TYPICAL sample at roundoff: 99.9 % of the samples below |dqpos| 1e-10, |dqvel| 1e-8 -- a drift of the kernel's
        arithmetic would move the bulk, an IK-limited sample only the tail."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from oracle.oracle import Oracle
    cm = compile_model(env, auto_reset=True)
    n, steps = 4096, 130
    dev = env_hip.KManipEnvHip(cm, num_envs=n, seed=11, env_id_offset=3)
    orc = Oracle(cm, n, seed=11, env_id_offset=3)
    dev.k_reset(); orc.reset()
    stagger = (np.arange(n) % 64).astype(np.int32)
    dev.set_state(step=stagger); orc.set_state(step=stagger)
    dq_all, dv_all = [], []
    worst_r = 0.0
    n_ctrl = 0
    seen = 0
    for k in range(steps):
        act = dev.sample_action()
        an = act.cpu().numpy()
        assert np.array_equal(an, orc.sample_action()), k
        dev.step_flat(act)
        oo, ro, do = orc.step(an, nthreads=16)
        sg, so = dev.get_state(), orc.get_state()
        flip = (sg[2] != so[2]).any(axis=1)       # a float32 rounding flip of ctrl (1 ulp) legitimately moves that env's step by ~1e-5
        if flip.any():
            # two float64 IK results within the IK bar (1e-7 rad) of each other, each rounded to float32: one float32 ulp of rounding
            # on top of the bar (near zero an ulp is far below it: 4e-9 at |ctrl| = 0.04)
            ulp = np.spacing(np.abs(so[2][flip]).astype(np.float32)).astype(np.float64)
            assert (np.abs(sg[2][flip] - so[2][flip]) <= ulp + 1e-7).all(), k
        ok = ~flip
        n_ctrl += int(flip.sum())
        dq_all.append(np.abs(sg[0] - so[0]).max(axis=1)[ok]); dv_all.append(np.abs(sg[1] - so[1]).max(axis=1)[ok])
        worst_r = max(worst_r, float(np.abs(dev.reward.cpu().numpy() - ro)[ok].max()))
        mg, nfg, stg = dev.get_diag(); mo, nfo, sto = orc.get_diag()
        assert np.array_equal(mg, mo), (k, np.where(mg != mo)[0][:8])
        assert np.array_equal(dev.done.cpu().numpy(), do), k
        assert np.array_equal(sg[4], so[4]), k
        assert np.abs(nfg - nfo).max() <= 1 and np.array_equal(stg == -2, sto == -2), k
        seen |= int(np.bitwise_or.reduce(mg))
        orc.set_state(*sg)                         # one-step samples
    dq, dv = np.concatenate(dq_all), np.concatenate(dv_all)
    assert len(dq) > 0.99 * n * steps
    assert dq.max() < 1e-7 and dv.max() < 4e-5 and worst_r < 1e-7, (dq.max(), dv.max(), worst_r)                  # every sample: IK-derived
    assert np.quantile(dq, 0.999) < 1e-10 and np.quantile(dv, 0.999) < 1e-8, (np.quantile(dq, 0.999), np.quantile(dv, 0.999))   # the bulk: roundoff
    assert n_ctrl <= 32, n_ctrl
    assert seen & 0xF, hex(seen)                   # cube-table contacts were in the sample
    dev.k_close()


def test_ik_max_nfev_cap_is_opt_in_and_bounds_the_crawl():
    """KModelDesc.ik_max_nfev (include/kmanip.h): 0 = the reference's least_squares default (100 n evaluations; what every other
    test and the bench run); a positive value caps every ik() call -- a throughput caller's protection against the 350-650-
    evaluation crawls one env in ~10^5 steps runs into.  With the cap the device and the oracle (which honours the same field)
    still agree step for step, no call exceeds it, capped calls report status 0 like SciPy at max_nfev, and calls that needed
    fewer evaluations are bit for bit those of the uncapped handle."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from oracle.oracle import Oracle
    cap, n, steps = 20, 1024, 24
    cm_c = compile_model("KManipSoloArm", auto_reset=True, ik_max_nfev=cap)
    cm_0 = compile_model("KManipSoloArm", auto_reset=True)
    assert cm_0.desc.ik_max_nfev == 0 and cm_c.desc.ik_max_nfev == cap
    dc = env_hip.KManipEnvHip(cm_c, num_envs=n, seed=3); d0 = env_hip.KManipEnvHip(cm_0, num_envs=n, seed=3)
    oc = Oracle(cm_c, n, seed=3)
    dc.k_reset(); d0.k_reset(); oc.reset()
    capped = 0
    for k in range(steps):
        act = dc.sample_action().clone()
        s_pre = dc.get_state()
        d0.set_state(*s_pre); oc.set_state(*s_pre)            # one-step samples from the capped handle's state
        dc.step_flat(act); d0.step_flat(act)
        oc.step(act.cpu().numpy(), nthreads=8)
        mc, nfc, stc = dc.get_diag(); m0, nf0, st0 = d0.get_diag(); mo, nfo, sto = oc.get_diag()
        assert nfc[:, 0].max() <= cap and nf0[:, 0].max() > cap
        hit = nf0[:, 0] > cap
        assert (nfc[hit, 0] == cap).all() and (stc[hit, 0] == 0).all()
        same = ~hit
        sc, s0, so = dc.get_state(), d0.get_state(), oc.get_state()
        assert all(np.array_equal(x[same], y[same]) for x, y in zip(sc, s0))        # calls below the cap: untouched
        assert np.abs(nfc[:, 0] - nfo[:, 0]).max() <= 1 and nfo[:, 0].max() <= cap       # the oracle honours the same cap
        agree = nfc[:, 0] == nfo[:, 0]
        assert np.array_equal(stc[agree, 0] == 0, sto[agree, 0] == 0) and agree.mean() > 0.95
        ok = (sc[2] == so[2]).all(axis=1) & agree
        assert np.abs(sc[0] - so[0])[ok].max() < 1e-6
        capped += int(hit.sum())
    assert capped > 100, capped
    dc.k_close(); d0.k_close()


def test_cost_sorted_wave_slots_change_nothing_but_the_order(monkeypatch):
    """Launches of several residency rounds run their waves in predicted-cost order (k_sort_envs: longest first, from the last
    step's diagnostics).  The order is a permutation of the envs, and an env's bits do not depend on it: a handle with the sort on
    (automatic above 2048 two-arm envs) against one with KMANIP_COST_SORT=0, bit for bit, across an auto-reset."""
    import ctypes as C
    import torch
    from gym_kmanip_amd import env_hip
    n = 4096
    monkeypatch.setenv("KMANIP_WAVE_CLOCKS", "1")
    a = env_hip.make("KManipDualArm", num_envs=n, seed=5)              # sorted (n > 2048)
    monkeypatch.setenv("KMANIP_COST_SORT", "0")
    b = env_hip.make("KManipDualArm", num_envs=n, seed=5)
    monkeypatch.delenv("KMANIP_COST_SORT")
    a.k_reset(); b.k_reset()
    ph = (58 + np.arange(n) % 6).astype(np.int32)
    a.set_state(step=ph); b.set_state(step=ph)
    slot = np.zeros(n, dtype=np.int32)
    orders = set()
    for k in range(10):
        act = a.sample_action().clone()
        a.step_flat(act); b.step_flat(act)
        assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done), k
        assert a.L.kmanip_dbg_wave_clocks(a.h, None, slot.ctypes.data_as(C.POINTER(C.c_int32)), None) == 0
        assert np.array_equal(np.sort(slot), np.arange(n)), k                    # a permutation
        orders.add(slot.tobytes())
    sa, sb = a.get_state(), b.get_state()
    assert all(np.array_equal(x, y) for x, y in zip(sa, sb))
    assert len(orders) > 5                                                       # and it does re-order from step to step
    a.k_close(); b.k_close()


@pytest.mark.parametrize("n,hepb", [(4096, "1"), (4096, "2"), (2048, "1"), (4096, "0"), (4096, "spread"), (2048, "spread")])
def test_heavy_first_dispatch_changes_nothing_but_the_slots(monkeypatch, n, hepb):
    """Single-arm launches of one residency round dispatch the envs predicted heavy (a collider on or near the cube at the end of
    their last step) FIRST and with a wave to themselves (or two per wave), everybody else four (two) per wave behind them: the
    dispatch table is a partition of the env ids rebuilt by every step, and an env's bits depend neither on its slot nor on its
    wave-mates -- a handle with the dispatch (KMANIP_HEAVY_DISPATCH=1) against one without, bit for bit,
    across an auto-reset; the table is a permutation every step and heavy envs do occur.
    hepb = "0": the list-based spread -- the grid of the plain launch, one predicted-heavy env in lane group 0 of the first waves
    and light envs beside it.  hepb = "spread": what a handle of >= 2048 single-arm envs does BY DEFAULT:
    per-env flags, every block of 64 consecutive envs dealt to its own 16 (32) waves so that no wave holds two heavy envs --
    a permutation inside every block, and heavy envs of a block sit in different waves."""
    import ctypes as C
    import torch
    from gym_kmanip_amd import env_hip
    spread = hepb == "spread"
    if not spread:
        monkeypatch.setenv("KMANIP_HEAVY_EPB", hepb)
        monkeypatch.setenv("KMANIP_HEAVY_DISPATCH", "1")
    a = env_hip.make("KManipSoloArm", num_envs=n, seed=5)              # hepb 0 / 1 / 2: launch-wide lists (opt-in experiments: DESIGN.md 3.2)
    monkeypatch.setenv("KMANIP_HEAVY_DISPATCH", "0")
    b = env_hip.make("KManipSoloArm", num_envs=n, seed=5)              # neither: the identity map
    monkeypatch.delenv("KMANIP_HEAVY_DISPATCH")
    S = a.L.kmanip_dbg_wave_slots(a.h)
    assert (S > n if hepb in ("1", "2") else S == n) and b.L.kmanip_dbg_wave_slots(b.h) == n
    a.k_reset(); b.k_reset()
    ph = (40 + np.arange(n) % 24).astype(np.int32)
    a.set_state(step=ph); b.set_state(step=ph)
    slot = np.zeros(S, dtype=np.int32)
    nheavy, epb = [], 4 if n >= 4096 else 2
    for k in range(30):
        act = a.sample_action().clone()
        a.step_flat(act); b.step_flat(act)
        assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done), k
        assert a.L.kmanip_dbg_wave_clocks(a.h, None, slot.ctypes.data_as(C.POINTER(C.c_int32)), None) == 0
        assert np.array_equal(np.sort(slot[slot >= 0]), np.arange(n)), k         # every env exactly once
        per_wave = (slot[:(S // epb) * epb].reshape(-1, epb) >= 0).sum(1)
        if hepb == "0" or spread:
            assert (per_wave[:n // epb] == epb).all() and not per_wave[n // epb:].any()      # the plain grid, every wave full
            nheavy.append(int((slot[:n] != np.arange(n)).sum()))                             # (no heavy env: the identity)
        if spread:
            blocks = slot[:n].reshape(-1, 64)
            assert np.array_equal(np.sort(blocks, axis=1), np.arange(n).reshape(-1, 64)), k   # a permutation inside every 64-env block
        else:
            nheavy.append(int(((per_wave > 0) & (per_wave <= int(hepb)) & (per_wave < epb)).sum()))
    assert all(np.array_equal(x, y) for x, y in zip(a.get_state(), b.get_state()))
    assert np.array_equal(a.get_diag()[0], b.get_diag()[0])
    assert max(nheavy) > 0, nheavy                                               # some envs were dispatched as heavy
    a.k_close(); b.k_close()


def test_interleaved_batches_are_the_envs_of_one_big_batch():
    """pipeline.InterleavedBatches: K handles on K streams, stepped round-robin without synchronising.  Batch i is, bit for bit,
    envs [i * n, (i + 1) * n) of one K * n-env handle (global-id RNG keys), and the stream-pairing probe of the constructor
    (real control steps between a checkpoint and its restore) leaves no trace."""
    import torch
    import gym_kmanip_amd as k
    from gym_kmanip_amd import env_hip
    n, K = 512, 2
    two = k.make_interleaved("KManipSoloArm", n, k=K, seed=9)
    one = env_hip.make("KManipSoloArm", num_envs=K * n, seed=9)
    plain = k.make_interleaved("KManipSoloArm", n, k=K, seed=9, calibrate=False)
    one.k_reset()
    for b in (two, plain):
        for i in range(K):
            with b.on(i):
                b.env[i].k_reset()
    for step in range(70):                                     # across the TimeLimit auto-reset
        act = one.sample_action().clone()                      # (drawn on the default stream: the batches' streams wait for it)
        one.step_flat(act)
        for b in (two, plain):
            for i in range(K):
                b.stream[i].wait_stream(torch.cuda.current_stream())
                with b.on(i):
                    b.env[i].step_flat(act[i * n:(i + 1) * n].contiguous())
        torch.cuda.current_stream().wait_stream(two.stream[0])  # (act's memory is reused by the next clone only after its readers)
        for b in (two, plain):
            for i in range(K):
                torch.cuda.current_stream().wait_stream(b.stream[i])
    two.synchronize(); plain.synchronize(); torch.cuda.synchronize()
    for b in (two, plain):
        for i in range(K):
            sl = slice(i * n, (i + 1) * n)
            assert torch.equal(b.env[i].obs, one.obs[sl]) and torch.equal(b.env[i].reward, one.reward[sl]) and torch.equal(b.env[i].done, one.done[sl])
            for x, y in zip(b.env[i].get_state(), one.get_state()):
                assert np.array_equal(x, y[sl])
    assert two.stream[0] != two.stream[1]
    two.close(); plain.close(); one.k_close()
