"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
 (a) the float64 C oracle on the same seeded inputs, (b) the committed fixtures in tests/golden/
 (SciPy-generated IK vectors, oracle trajectories), (c) size-independent properties at BASELINE sizes.

Tolerances (float64 on both sides; differences come only from operation order -- tree reductions,
Cholesky-vs-SVD in the IK, explicit M^-1 vs factor solves):
   IK vs SciPy          1e-6 rad   (termination knife-edges may shift nfev by one evaluation)
   IK vs oracle         1e-7 rad
   qpos / obs           1e-7       over a full 64-step episode + the auto-reset boundary
   qvel                 1e-5
   reward               1e-6
   ctrl, done, contact masks, step counters: bit-exact
"""
import os

import numpy as np
import pytest

from conftest import ENVS3, GOLDEN
from gym_kmanip_amd.model import KM_DONE_DIVERGED, KM_DONE_TRUNCATED, compile_model

pytestmark = pytest.mark.gpu

TOL_Q, TOL_V, TOL_R = 1e-7, 1e-5, 1e-6


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _mk(env_id, n, seed=0, off=0, **kw):
    from gym_kmanip_amd import env_hip
    from oracle.oracle import Oracle
    cm = compile_model(env_id, **kw)
    return cm, env_hip.KManipEnvHip(cm, num_envs=n, seed=seed, env_id_offset=off), Oracle(cm, n, seed=seed, env_id_offset=off)


def _cmp_state(g, o, k="", resync=None):
    """qpos / qvel within tolerance, step counters and ctrl bit-exact.  ctrl is float32-quantised (env_sim.py:40,71), so a
    free-running comparison has one legitimate failure mode: two float64 IK results that agree to 1e-9 can straddle a
    float32 rounding boundary.  With `resync` (a one-element list used as a counter) such a flip -- every mismatching entry
    within one float32 ulp -- re-synchronises the oracle to the device state instead of failing; callers bound the count.
    Returns the boolean mask of the envs whose ctrl flipped (all False normally): ONLY those envs get the looser bar for this
    step (their step was driven by targets 1.2e-7 apart: ten times the bars); every other env keeps the normal one."""
    sg, so = g.get_state(), o.get_state()
    bad = sg[2] != so[2]
    flipped = bad.any(axis=1)
    if flipped.any():
        ulp = np.spacing(np.abs(so[2][bad]).astype(np.float32)).astype(np.float64)
        assert resync is not None and (np.abs(sg[2][bad] - so[2][bad]) <= ulp).all(), ("ctrl", k, sg[2][bad], so[2][bad])
    tq = np.where(flipped, 10 * TOL_Q, TOL_Q)[:, None]
    tv = np.where(flipped, 10 * TOL_V, TOL_V)[:, None]
    assert (np.abs(sg[0] - so[0]) < tq).all(), ("qpos", k, np.abs(sg[0] - so[0]).max())
    assert (np.abs(sg[1] - so[1]) < tv).all(), ("qvel", k, np.abs(sg[1] - so[1]).max())
    assert np.array_equal(sg[4], so[4]), ("step_idx", k)
    if flipped.any():
        resync[0] += 1
        o.set_state(*sg)
    return flipped


@pytest.mark.parametrize("env", ENVS3)
def test_ik_golden_scipy_gpu(env):
    _torch()
    g = np.load(os.path.join(GOLDEN, "ik_scipy_%s.npz" % env))
    cm, dev, orc = _mk(env, 1)
    nf_mismatch = 0
    for arm in range(2):
        sel = np.where(g["arm"] == arm)[0]
        if len(sel) == 0:
            continue
        n = cm.desc.arm_nq[arm]
        q, qp_after, nfev, st = dev.ik(arm, g["qpos"][sel], g["goal_pos"][sel], g["goal_quat"][sel])
        assert np.abs(q - g["q_out"][sel][:, :n]).max() < 1e-6
        assert np.abs(qp_after - g["qpos_after"][sel]).max() < 1e-6
        nf_mismatch += int((nfev != g["nfev"][sel]).sum())
        failed = g["status"][sel] == -2                      # "IK failed: x0 is infeasible" branch
        assert np.array_equal(st[failed], g["status"][sel][failed]) and (nfev[failed] == 0).all()
        assert np.array_equal(qp_after[failed], g["qpos"][sel][failed])
        for i, s in enumerate(sel):                          # and against the oracle
            qo, qpo, nfo, sto = orc.ik(arm, g["qpos"][s], g["goal_pos"][s], g["goal_quat"][s])
            assert np.abs(q[i] - qo).max() < 1e-7 and np.abs(qp_after[i] - qpo).max() < 1e-7
    assert nf_mismatch <= 3, nf_mismatch
    dev.k_close()


@pytest.mark.parametrize("env", ENVS3)
def test_ik_residual_and_jacobian_vs_scipy_fixture(env):
    """ik_res / ik_jac as the HIP IK evaluates them (coop_eval + its regulariser rows) at x0 against the res0 / jac0 columns
    of the NumPy/SciPy fixtures (ik_mujoco.py:20-97, incl. the deliberately inconsistent 9e-3 regulariser rows): 1e-12."""
    _torch()
    g = np.load(os.path.join(GOLDEN, "ik_scipy_%s.npz" % env))
    cm, dev, orc = _mk(env, 1)
    for arm in range(2):
        sel = np.where(g["arm"] == arm)[0]
        if len(sel) == 0:
            continue
        n = cm.desc.arm_nq[arm]
        res, jac = dev.ik_eval(arm, g["qpos"][sel], g["goal_pos"][sel], g["goal_quat"][sel])
        m = 6 + 2 * n
        assert np.abs(res - g["res0"][sel][:, :m]).max() < 1e-12
        assert np.abs(jac.reshape(len(sel), -1) - g["jac0"][sel][:, :m * n]).max() < 1e-12
        assert np.abs(jac[:, 6:6 + n, :] - 9e-3 * np.eye(n)).max() == 0 and np.abs(jac[:, 6 + n:, :] - 9e-3 * np.eye(n)).max() == 0
    dev.k_close()


@pytest.mark.parametrize("env", ENVS3)
def test_reset_parity(env):
    torch = _torch()
    cm, dev, orc = _mk(env, 32, seed=11, off=1000)
    dev.k_reset(); obs_o = orc.reset()
    sg, so = dev.get_state(), orc.get_state()
    assert np.array_equal(sg[0], so[0]) and np.array_equal(sg[1], so[1]) and np.array_equal(sg[2], so[2])
    assert np.abs(sg[3] - so[3]).max() < 1e-8          # qacc_warmstart of the unactuated mj_forward
    assert np.array_equal(dev.obs.cpu().numpy(), obs_o)
    # masked reset touches only the selected envs and advances their episode (new cube spawn)
    mask = np.zeros(32, dtype=np.uint8); mask[[3, 17]] = 1
    before = dev.get_state()[0].copy()
    dev.k_reset(mask)
    after = dev.get_state()[0]
    changed = np.where(np.abs(after - before).max(axis=1) > 0)[0]
    assert list(changed) == [3, 17]
    dev.k_close()


@pytest.mark.parametrize("solver", ["pgs", "newton"])
@pytest.mark.parametrize("env,n,steps", [("KManipSoloArm", 32, 70), ("KManipDualArm", 16, 66), ("KManipTorso", 16, 66)])
def test_step_parity_vs_oracle(env, n, steps, solver):
    """Full episodes incl. cube landing (contacts), joint-limit hits, IK-infeasible starts and the
    auto-reset at step 64, on identical seeded actions."""
    torch = _torch()
    cm, dev, orc = _mk(env, n, seed=5, off=7, auto_reset=True, solver=solver)
    dev.k_reset(); orc.reset()
    rng = np.random.default_rng(42)
    saw_contact = saw_reset = False
    resync = [0]
    for k in range(steps):
        act = rng.uniform(-1, 1, (n, cm.act_dim)).astype(np.float32)
        dev.step_flat(torch.from_numpy(act).cuda())
        oo, ro, do = orc.step(act)
        flip = _cmp_state(dev, orc, k, resync)                     # (a flipped float32 ctrl entry: the observation's velocity part moves with qvel)
        assert (np.abs(dev.obs.cpu().numpy() - oo) < np.where(flip, 10 * TOL_V, TOL_Q)[:, None]).all(), k      # (only the flipped envs get the loose bar)
        assert (np.abs(dev.reward.cpu().numpy() - ro) < np.where(flip, 10 * TOL_R, TOL_R)).all(), k
        assert np.array_equal(dev.done.cpu().numpy(), do), k
        mg, nfg, stg = dev.get_diag(); mo, nfo, sto = orc.get_diag()
        assert np.array_equal(mg, mo), (k, mg, mo)
        assert np.array_equal(stg == -2, sto == -2) and np.abs(nfg - nfo).max() <= 1, k
        saw_contact |= bool(mg.any()); saw_reset |= bool(do.any())
    assert saw_contact and saw_reset
    # float32 ctrl flips: a handful in 799 k samples (profiles/r03_parity_soak.txt).  These short runs are deterministic: the Newton
    # runs of SoloArm and Torso have none, DualArm's has exactly one (one env, one entry, one ulp); the non-converging PGS variant
    # drifts further from the oracle late in an episode and may straddle a boundary twice.  A flipped env alone gets the loose bar.
    assert resync[0] <= (2 if solver == "pgs" else (1 if env == "KManipDualArm" else 0)), resync
    dev.k_close()


@pytest.mark.parametrize("env", ["KManipDualArm", "KManipTorso"])
def test_block_structure_paths_agree_with_the_full_sweep(env, monkeypatch):
    """Two-arm models: the joint-space inertia is two diagonal blocks (KModelAux.split).  The step then runs the tree passes
    and the inversion one block per DPP row and applies the arm solve's Woodbury shortcut per block; KMANIP_NO_BLOCK_SPLIT=1
    keeps the 20-dof code.  Same mathematics, different operation order: one-step agreement to roundoff on the same states."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    n = 256
    a = env_hip.make(env, num_envs=n, seed=4)
    monkeypatch.setenv("KMANIP_NO_BLOCK_SPLIT", "1")
    b = env_hip.make(env, num_envs=n, seed=4)
    monkeypatch.delenv("KMANIP_NO_BLOCK_SPLIT")
    a.k_reset(); b.k_reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    worst = 0.0
    for k in range(70):
        act = torch.rand((n, a.cm.act_dim), generator=gen, device="cuda") * 2 - 1
        a.step_flat(act); b.step_flat(act)
        sa, sb = a.get_state(), b.get_state()
        same_ctrl = ~(sa[2] != sb[2]).any(axis=1)                 # (a float32 ctrl flip legitimately moves an env by ~1e-5)
        worst = max(worst, float(np.abs(sa[0] - sb[0])[same_ctrl].max()))
        assert np.abs(sa[0] - sb[0])[same_ctrl].max() < 1e-9 and np.abs(sa[1] - sb[1])[same_ctrl].max() < 1e-7, k
        assert np.array_equal(a.get_diag()[0], b.get_diag()[0]) and torch.equal(a.done, b.done), k
        b.set_state(*sa)                                           # one-step samples
    a.k_close(); b.k_close()


def test_touch_reward_parity():
    """touch_reward=True (the reference's unreachable touch / lift terms switched on): reward parity incl. the bonuses, which
    only a FINGER sphere on the cube earns (+1, and +1 more with no cube corner on the table)."""
    torch = _torch()
    from oracle.oracle import Oracle
    import gym_kmanip_amd.model as K
    n = 256
    cm, dev, orc = _mk("KManipSoloArm", n, seed=3, touch_reward=True)
    plain = Oracle(compile_model("KManipSoloArm"), n, seed=3)          # same physics, reward without the touch terms
    dev.k_reset(); orc.reset(); plain.reset()
    rng = np.random.default_rng(1)
    seen = np.zeros(3, dtype=int)
    for k in range(64):
        act = rng.uniform(-1, 1, (n, cm.act_dim)).astype(np.float32)
        dev.step_flat(torch.from_numpy(act).cuda()); ro = orc.step(act)[1]; r0 = plain.step(act)[1]
        mg = dev.get_diag()[0]
        assert np.array_equal(mg, orc.get_diag()[0]), k
        rg = dev.reward.cpu().numpy()
        assert np.abs(rg - ro).max() < TOL_R, (k, np.abs(rg - ro).max())
        finger = (mg & 0x300) != 0; other = ((mg & 0xFFF00) != 0) & ~finger; table = (mg & 0xFF) != 0
        bonus = np.where(finger, K.REWARD_TOUCH_CUBE + np.where(table, 0.0, K.REWARD_LIFT_CUBE), 0.0)
        assert np.abs(rg - r0 - bonus).max() < TOL_R, k
        seen += [int((finger & table).sum()), int((finger & ~table).sum()), int(other.sum())]
        sg = dev.get_state(); orc.set_state(*sg); plain.set_state(*sg)
    assert (seen > 0).all(), seen
    dev.k_close()


def test_forearm_on_table_gpu():
    """The link-collider scenario of tests/test_oracle_dynamics.py on the device: same mask bit, same trajectory."""
    _torch()
    from gym_kmanip_amd import env_hip
    from oracle.oracle import Oracle
    from test_oracle_dynamics import forearm_on_table
    cm = compile_model("KManipSoloArmQPos", auto_reset=False)
    dev = forearm_on_table(cm, cm.desc.nsphere, lambda c: env_hip.KManipEnvHip(c, num_envs=1, seed=0))
    orc = forearm_on_table(cm, cm.desc.nsphere, lambda c: Oracle(c, 1, seed=0))
    torch = _torch()
    for k in range(3):
        act = np.zeros((1, cm.act_dim), dtype=np.float32)
        dev.step_flat(torch.from_numpy(act).cuda()); orc.step(act)
        _cmp_state(dev, orc, k)
        assert int(dev.get_diag()[0][0]) == int(orc.get_diag()[0][0]) == 1 << (20 + 4)
    dev.k_close()


def _crowded(cm):
    """A descriptor whose link spheres crowd the contact slots: they are moved onto the finger links (next to the finger
    spheres), so every finger on the table brings three candidates down at once."""
    from gym_kmanip_amd.model import KModelDesc
    d = KModelDesc.from_buffer_copy(cm.desc)
    nf = 2 * (cm.nlink // 10)
    for s in range(nf, d.nsphere):
        d.sphere_link[s] = d.sphere_link[s % nf]
        for k in range(3):
            d.sphere_pos[s][k] = d.sphere_pos[s % nf][k] + 0.004 * ((s - nf) // nf + 1) * (1 if k == s % 3 else 0)
        d.sphere_radius[s] = 0.012
    return type(cm)(**{**cm.__dict__, "desc": d})


@pytest.mark.parametrize("env,n", [("KManipSoloArm", 32), ("KManipTorso", 16)])
def test_sphere_slot_overflow_parity(env, n):
    """KM_SPHERE_SLOTS / KM_SPHERE_TABLE_SLOTS: with more penetrating spheres than slots, device and oracle keep the same ones (the first in sphere
    order) -- contact masks bit-exact, states within tolerance -- on a model rigged so that this happens often."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from oracle.oracle import Oracle
    from oracle import ik_scipy as S
    cm = _crowded(compile_model(env, auto_reset=True))
    d = cm.desc
    dev = env_hip.KManipEnvHip(cm, num_envs=n, seed=5, env_id_offset=7); orc = Oracle(cm, n, seed=5, env_id_offset=7)
    dev.k_reset(); orc.reset()
    rng = np.random.default_rng(43)
    nss = 2 * (cm.nlink // 10)                                  # KM_SPHERE_SLOTS (sphere-cube)
    nst = 2 if cm.nlink == 10 else 6                            # KM_SPHERE_TABLE_SLOTS
    full = over = 0
    resync = [0]
    for k in range(40):
        act = rng.uniform(-1, 1, (n, cm.act_dim)).astype(np.float32)
        dev.step_flat(torch.from_numpy(act).cuda())
        oo, ro, do = orc.step(act)
        _cmp_state(dev, orc, k, resync)
        mg = dev.get_diag()[0]; mo = orc.get_diag()[0]
        assert np.array_equal(mg, mo), (k, mg, mo)
        assert np.array_equal(dev.done.cpu().numpy(), do), k
        tab = np.array([bin(int(m) >> 20).count("1") for m in mg]); cub = np.array([bin((int(m) >> 8) & 0xFFF).count("1") for m in mg])
        assert tab.max() <= nst and cub.max() <= nss
        full += int((tab == nst).sum())
        qpos = orc.get_state()[0]
        for e in np.where(tab == nst)[0]:                       # slots full: were more spheres down than were kept?
            xpos, xquat, _, _ = orc.fk(qpos[e])
            down = sum((xpos[d.sphere_link[s]] + S.quat2mat(xquat[d.sphere_link[s]]) @ np.array(d.sphere_pos[s]))[2]
                       - d.sphere_radius[s] < d.table_z for s in range(d.nsphere))
            over += int(down > nst)
    assert full > 0 and over > 0, (full, over)
    assert resync[0] <= 1, resync
    dev.k_close()


@pytest.mark.parametrize("env", ENVS3)
def test_golden_trajectory_gpu(env):
    torch = _torch()
    g = np.load(os.path.join(GOLDEN, "traj_%s.npz" % env))
    from gym_kmanip_amd import env_hip
    cm = compile_model(env, auto_reset=True)
    dev = env_hip.KManipEnvHip(cm, num_envs=g["act"].shape[1], seed=int(g["seed"]), env_id_offset=int(g["env_id_offset"]))
    dev.k_reset()
    assert np.abs(dev.obs.cpu().numpy() - g["obs0"]).max() < 1e-12
    for k in range(g["act"].shape[0]):
        dev.step_flat(torch.from_numpy(g["act"][k]).cuda())
        qpos, qvel, ctrl, warm, step = dev.get_state()
        assert np.abs(qpos - g["qpos"][k]).max() < TOL_Q and np.abs(qvel - g["qvel"][k]).max() < TOL_V, k
        assert np.array_equal(ctrl, g["ctrl"][k]), k
        assert np.abs(dev.obs.cpu().numpy() - g["obs"][k]).max() < TOL_Q
        assert np.abs(dev.reward.cpu().numpy() - g["rew"][k]).max() < TOL_R
        assert np.array_equal(dev.done.cpu().numpy(), g["done"][k])
        assert np.array_equal(dev.get_diag()[0], g["mask"][k])
    dev.k_close()


@pytest.mark.parametrize("env", ["KManipSoloArmQPos", "KManipDualArmQPos"])
def test_qpos_action_modes(env):
    torch = _torch()
    cm, dev, orc = _mk(env, 8, seed=2)
    dev.k_reset(); orc.reset()
    rng = np.random.default_rng(9)
    for k in range(6):
        act = rng.uniform(-1, 1, (8, cm.act_dim)).astype(np.float32)
        dev.step_flat(torch.from_numpy(act).cuda()); orc.step(act)
        _cmp_state(dev, orc, k)
    st = dev.get_diag()[2]
    assert all((st[:, a] == -3).all() for a in range(2) if cm.desc.arm_present[a])       # no IK ran
    dev.k_close()


def test_seam_dict_api():
    """k_reset / k_step through the reference's backend seam with dict actions (env_base.py:219-259)."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.gym_shell import KManipEnv
    env = KManipEnv("KManipSoloArm", num_envs=4, seed=1)
    obs, info = env.reset()
    assert list(obs.keys()) == ["q_pos", "q_vel", "cube_pos", "cube_orn"]
    act = {"eer_pos": np.zeros((4, 3), np.float32), "eer_orn": np.zeros((4, 3), np.float32), "grip_r": np.zeros((4, 1), np.float32)}
    for k in range(64):
        obs, reward, terminated, truncated, info = env.step(act)
    assert not np.asarray(terminated).any() and np.asarray(truncated).all() and info["step"] == 64
    assert obs["q_pos"].shape == (4, 10) and obs["cube_orn"].shape == (4, 4) and obs["q_pos"].dtype == np.float64
    env.close()


def test_diverged_flag_and_recovery():
    torch = _torch()
    cm, dev, orc = _mk("KManipSoloArm", 4, seed=1, auto_reset=False)
    dev.k_reset()
    qpos, qvel, ctrl, warm, step = dev.get_state()
    qvel[2, 3] = np.nan
    dev.set_state(qvel=qvel)
    dev.step_flat(torch.zeros((4, 7), dtype=torch.float32, device="cuda"))
    done = dev.done.cpu().numpy()
    assert done[2] & KM_DONE_DIVERGED and not (done[[0, 1, 3]] & KM_DONE_DIVERGED).any()
    q2 = dev.get_state()
    assert np.isfinite(q2[0]).all() and np.isfinite(q2[1]).all() and q2[4][2] == 0    # env 2 was reset
    dev.k_close()


def test_full_size_properties_4096():
    """BASELINE config 2 size (KManipSoloArm @ 4096): determinism, shard independence, invariants."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    n = 4096
    a = env_hip.make("KManipSoloArm", num_envs=n, seed=3)
    b = env_hip.make("KManipSoloArm", num_envs=n, seed=3)
    c = env_hip.make("KManipSoloArm", num_envs=512, seed=3, env_id_offset=1024)   # a shard of the same job
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    a.k_reset(); b.k_reset(); c.k_reset()
    for k in range(12):
        act = torch.rand((n, 7), generator=gen, device="cuda") * 2 - 1
        a.step_flat(act); b.step_flat(act.clone()); c.step_flat(act[1024:1536].contiguous())
    sa, sb, sc = a.get_state(), b.get_state(), c.get_state()
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)                                # bitwise deterministic
    for x, z in zip(sa, sc):
        assert np.array_equal(x[1024:1536], z)                     # results do not depend on the shard layout
    nl = 10
    assert np.abs(np.linalg.norm(sa[0][:, nl + 3:], axis=1) - 1).max() < 1e-12   # unit cube quaternion
    obs = a.obs.cpu().numpy()
    assert np.isfinite(obs).all() and (np.abs(obs[:, :2 * nl + 3]) <= 1).all()
    assert (sa[4] == 12).all() and not a.done.cpu().numpy().any()
    rect = env_hip.make("KManipSoloArm", num_envs=1).cm.desc.table_rect
    cx, cy = sa[0][:, nl], sa[0][:, nl + 1]
    over = (cx > rect[0]) & (cx < rect[1]) & (cy > rect[2]) & (cy < rect[3])
    assert over.mean() > 0.99 and (sa[0][over, nl + 2] > 0.5).all()      # no cube centre below the table top while over it
    for e in (a, b, c):
        e.k_close()


@pytest.mark.parametrize("env_id", ["KManipDualArm", "KManipTorso"])
def test_full_size_configs_3_and_4_shard(env_id):
    """BASELINE config 3 (KManipDualArm @ 8192, IK on both arms) and the per-GPU shard of config 4 (KManipTorso, 65536 envs
    = 8 x 8192) at full size on the 32-lanes-per-env path, 70 control steps (across the 64-step auto-reset):
      * bitwise determinism (two handles, same seed and actions),
      * shard independence: a 1024-env handle created with env_id_offset = 4096 reproduces envs 4096..5119 bit for bit,
        i.e. what rank r of the 8-GPU job computes does not depend on the shard layout,
      * an oracle comparison on a 16-env slice of the same batch (global env ids 4096..4111), every step,
      * invariants: unit cube quaternion, observations inside their Box bounds, done = TimeLimit only."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from oracle.oracle import Oracle
    n, lo, ns, no = 8192, 4096, 1024, 16
    a = env_hip.make(env_id, num_envs=n, seed=3)
    b = env_hip.make(env_id, num_envs=n, seed=3)
    c = env_hip.make(env_id, num_envs=ns, seed=3, env_id_offset=lo)
    orc = Oracle(a.cm, no, seed=3, env_id_offset=lo)
    nl, ad = a.cm.nlink, a.cm.act_dim
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    a.k_reset(); b.k_reset(); c.k_reset(); orc.reset()
    saw_contact = False
    for k in range(70):
        act = torch.rand((n, ad), generator=gen, device="cuda") * 2 - 1
        a.step_flat(act); b.step_flat(act.clone()); c.step_flat(act[lo:lo + ns].contiguous())
        oo, ro, do = orc.step(act[lo:lo + no].cpu().numpy())
        assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done), k
        assert torch.equal(a.obs[lo:lo + ns], c.obs) and torch.equal(a.reward[lo:lo + ns], c.reward) and torch.equal(a.done[lo:lo + ns], c.done), k
        assert np.abs(a.obs[lo:lo + no].cpu().numpy() - oo).max() < TOL_Q, k
        assert np.abs(a.reward[lo:lo + no].cpu().numpy() - ro).max() < TOL_R, k
        assert np.array_equal(a.done[lo:lo + no].cpu().numpy(), do), k
        done = a.done.cpu().numpy()
        assert not (done & KM_DONE_DIVERGED).any(), k
        assert (done == (KM_DONE_TRUNCATED if k == 63 else 0)).all(), k
        if k in (20, 62, 69):
            mg = a.get_diag()[0]; mo = orc.get_diag()[0]
            assert np.array_equal(mg[lo:lo + no], mo), k                 # contact masks bit-exact on the slice
            saw_contact |= bool(mg.any())
    sa, sb, sc, so = a.get_state(), b.get_state(), c.get_state(), orc.get_state()
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)
    for x, z in zip(sa, sc):
        assert np.array_equal(x[lo:lo + ns], z)
    assert np.abs(sa[0][lo:lo + no] - so[0]).max() < TOL_Q and np.abs(sa[1][lo:lo + no] - so[1]).max() < TOL_V
    assert np.array_equal(sa[2][lo:lo + no], so[2])
    assert saw_contact and (sa[4] == 6).all()
    assert np.abs(np.linalg.norm(sa[0][:, nl + 3:], axis=1) - 1).max() < 1e-12
    obs = a.obs.cpu().numpy()
    assert np.isfinite(obs).all() and (np.abs(obs) <= 1).all() and (obs[:, :nl] >= 0).all()
    for e in (a, b, c):
        e.k_close()


@pytest.mark.parametrize("env,cams", [("KManipSoloArm", [0]), ("KManipTorso", [0, 1])])
def test_render_depth_parity(env, cams):
    """Gripper-cam depth (BASELINE config 5) vs the oracle's ray caster on stepped states.  float64 maths on both
    sides, float32 output: equal to 1e-6 m except for rays grazing a primitive's silhouette (allowed: < 0.05 % pixels)."""
    torch = _torch()
    cm, dev, orc = _mk(env, 6, seed=4)
    dev.k_reset(); orc.reset()
    rng = np.random.default_rng(3)
    for k in range(14):
        act = rng.uniform(-1, 1, (6, cm.act_dim)).astype(np.float32)
        dev.step_flat(torch.from_numpy(act).cuda()); orc.step(act)
    qpos = orc.get_state()[0]
    for cam in cams:
        img = dev.render_depth(["grip_r", "grip_l"][cam], 64, 64).cpu().numpy()
        assert img.shape == (6, 64, 64) and img.dtype == np.float32
        for e in range(6):
            ref = orc.render_depth(qpos[e], cam, 64, 64)
            bad = np.abs(img[e] - ref) > 1e-6
            assert bad.mean() < 5e-4, (cam, e, bad.sum())
        assert (img >= cm.desc.cam_znear - 1e-6).all() and (img <= cm.desc.cam_zfar + 1e-6).all()
        assert np.unique(np.round(img, 3)).size > 10          # a real image, not a constant
    dev.k_close()


def test_render_depth_config5_size():
    """BASELINE config 5 size: 2048 envs x 64x64 gripper-cam depth rendered after every control step, across the auto-reset:
    deterministic (two handles), finite, inside [znear, zfar], and equal to the oracle's ray caster on a slice of the batch."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from oracle.oracle import Oracle
    n = 2048
    e = env_hip.make("KManipSoloArm", num_envs=n, seed=1); f = env_hip.make("KManipSoloArm", num_envs=n, seed=1)
    orc = Oracle(e.cm, 8, seed=1, env_id_offset=1000)
    e.k_reset(); f.k_reset(); orc.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(2)
    buf_e = torch.empty((n, 64, 64), dtype=torch.float32, device="cuda"); buf_f = torch.empty_like(buf_e)
    for k in range(66):
        act = torch.rand((n, 7), generator=gen, device="cuda") * 2 - 1
        e.step_flat(act); f.step_flat(act)
        e.render_depth("grip_r", 64, 64, out=buf_e); f.render_depth("grip_r", 64, 64, out=buf_f)
        orc.step(act[1000:1008].cpu().numpy())
        if k in (0, 30, 63, 65):
            assert torch.equal(buf_e, buf_f) and torch.isfinite(buf_e).all(), k
            assert (buf_e >= e.cm.desc.cam_znear - 1e-6).all() and (buf_e <= e.cm.desc.cam_zfar + 1e-6).all()
            qpos = orc.get_state()[0]
            img = buf_e[1000:1008].cpu().numpy()
            for j in range(8):
                bad = np.abs(img[j] - orc.render_depth(qpos[j], 0, 64, 64)) > 1e-6
                assert bad.mean() < 5e-4, (k, j, bad.sum())
    e.k_close(); f.k_close()


@pytest.mark.parametrize("env", ["KManipSoloArm", "KManipTorso"])
def test_fused_and_split_launches_agree(env, monkeypatch):
    """The product path runs before_step inside k_step; KMANIP_IK_UNFUSED=1 keeps it as separate launches.  Same device
    code on the same inputs => bit-identical state/obs/reward."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.lib import KManipError
    n = 96
    cm = compile_model(env)
    envs = {}
    for name, var in (("fused", None), ("split", "KMANIP_IK_UNFUSED")):
        monkeypatch.delenv("KMANIP_IK_UNFUSED", raising=False)
        if var:
            monkeypatch.setenv(var, "1")
        envs[name] = env_hip.KManipEnvHip(cm, num_envs=n, seed=9)      # the switch is read at kmanip_create
    monkeypatch.delenv("KMANIP_IK_UNFUSED", raising=False)
    gen = torch.Generator(device="cuda"); gen.manual_seed(4)
    for e in envs.values():
        e.k_reset()
    for k in range(70):                                                    # crosses the 64-step auto-reset
        act = torch.rand((n, cm.act_dim), generator=gen, device="cuda") * 2 - 1
        for e in envs.values():
            e.step_flat(act)
        f, s = envs["fused"], envs["split"]
        assert torch.equal(f.obs, s.obs) and torch.equal(f.reward, s.reward) and torch.equal(f.done, s.done), k
    for x, y in zip(envs["fused"].get_state(), envs["split"].get_state()):
        assert np.array_equal(x, y)
    df, ds = envs["fused"].get_diag(), envs["split"].get_diag()
    for x, y in zip(df, ds):
        assert np.array_equal(x, y)
    for e in envs.values():
        e.k_close()


@pytest.mark.parametrize("env,n,epb", [("KManipSoloArm", 1, None), ("KManipSoloArm", 5, None), ("KManipSoloArm", 37, "1"),
                                       ("KManipSoloArm", 37, "2"), ("KManipSoloArm", 300, None),
                                       ("KManipDualArm", 3, None), ("KManipDualArm", 21, "1"), ("KManipTorso", 21, "1"), ("KManipTorso", 33, None)])
def test_small_batches_and_launch_shapes(env, n, epb, monkeypatch):
    """Ragged / tiny batches and every envs-per-workgroup launch shape (KMANIP_EPB) against the oracle; on the two-arm models
    also the half-wave workgroup (one env per workgroup) of the block-per-row code."""
    torch = _torch()
    if epb:
        monkeypatch.setenv("KMANIP_EPB", epb)
    cm, dev, orc = _mk(env, n, seed=21, off=7)
    dev.k_reset(); orc.reset()
    rng = np.random.default_rng(n)
    for k in range(20):
        act = rng.uniform(-1, 1, (n, cm.act_dim)).astype(np.float32)
        dev.step_flat(torch.from_numpy(act).cuda()); obs_o, rew_o, done_o = orc.step(act)
        _cmp_state(dev, orc, k)
        assert np.abs(dev.obs.cpu().numpy() - obs_o).max() < TOL_Q
        assert np.abs(dev.reward.cpu().numpy() - rew_o).max() < TOL_R
        assert np.array_equal(dev.done.cpu().numpy(), done_o)
    mg, nf, st = dev.get_diag(); mo, nfo, sto = orc.get_diag()
    assert np.array_equal(mg, mo) and np.array_equal(nf[:, 0], nfo[:, 0]) and np.array_equal(st[:, 0], sto[:, 0])
    dev.k_close()


@pytest.mark.parametrize("env", ENVS3)
def test_scripted_policy_parity(env):
    """the reference's synthetic-data policy (examples/2_synthetic_data.py:28-41) on device vs the
    oracle: eer_pos columns = float32(unit vector site -> cube) at the current state, every other column untouched."""
    torch = _torch()
    cm, dev, orc = _mk(env, 48, seed=13)
    dev.k_reset(); orc.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(2)
    sl = cm.act_slices["eer_pos"]
    for k in range(8):
        raw = torch.rand((48, cm.act_dim), generator=gen, device="cuda") * 2 - 1
        act = dev.scripted_action(raw.clone())
        a = act.cpu().numpy(); r = raw.cpu().numpy()
        keep = np.ones(cm.act_dim, dtype=bool); keep[sl] = False
        assert np.array_equal(a[:, keep], r[:, keep])                       # the sampled columns pass through
        qpos = orc.get_state()[0]
        ref = np.stack([orc.scripted_eer_pos(qpos[e]) for e in range(48)])
        assert np.abs(a[:, sl] - ref.astype(np.float32)).max() <= 1.2e-7    # one float32 ulp of a unit vector
        assert np.abs(np.linalg.norm(a[:, sl].astype(np.float64), axis=1) - 1).max() < 1e-6
        dev.step_flat(act); orc.step(a)
        _cmp_state(dev, orc, k)
    dev.k_close()


def test_scripted_policy_rejected_without_ee_action():
    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.lib import KManipError
    torch = _torch()
    e = env_hip.make("KManipSoloArmQPos", num_envs=4)
    e.k_reset()
    with pytest.raises(KManipError):
        e.scripted_action(torch.zeros((4, e.cm.act_dim), dtype=torch.float32, device="cuda"))
    e.k_close()


@pytest.mark.parametrize("env,n,K", [("KManipSoloArm", 100, 7), ("KManipTorso", 40, 5)])
def test_step_chunk_equals_single_steps(env, n, K):
    """kmanip_step_chunk(K) == K x kmanip_step, bit for bit (same device code, the state merely stays in LDS between
    the steps of a chunk), including across the 64-step auto-reset boundary."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    a = env_hip.make(env, num_envs=n, seed=17); b = env_hip.make(env, num_envs=n, seed=17)
    a.k_reset(); b.k_reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    saw_done = False
    for rounds in range(11):                                             # 11 x 7 = 77 steps > 64
        acts = (torch.rand((K, n, a.cm.act_dim), generator=gen, device="cuda") * 2 - 1).contiguous()
        obs_c, rew_c, done_c = a.step_chunk(acts)
        saw_done |= bool(done_c.any())
        for k in range(K):
            b.step_flat(acts[k])
            assert torch.equal(obs_c[k], b.obs) and torch.equal(rew_c[k], b.reward) and torch.equal(done_c[k], b.done), (rounds, k)
        assert torch.equal(a.obs, b.obs)
    for x, y in zip(a.get_state(), b.get_state()):
        assert np.array_equal(x, y)
    for x, y in zip(a.get_diag(), b.get_diag()):
        assert np.array_equal(x, y)
    assert saw_done == (11 * K >= 64)             # (the SoloArm run crosses the TimeLimit auto-reset inside a chunk; the Torso run stops at step 55)
    a.k_close(); b.k_close()


def test_side_stream_execution():
    """Every entry point takes the caller's stream: stepping on a non-default torch stream gives the same results as on
    the default stream, and the work is really enqueued there (the event recorded on the side stream completes it)."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    n = 256
    a = env_hip.make("KManipSoloArm", num_envs=n, seed=23); b = env_hip.make("KManipSoloArm", num_envs=n, seed=23)
    gen = torch.Generator(device="cuda"); gen.manual_seed(8)
    acts = [(torch.rand((n, 7), generator=gen, device="cuda") * 2 - 1) for _ in range(10)]
    a.k_reset()
    for act in acts:
        a.step_flat(act)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        b.k_reset()
        for act in acts:
            b.step_flat(act)
        ev = torch.cuda.Event(); ev.record(side)
    ev.synchronize()
    assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done)
    for x, y in zip(a.get_state(), b.get_state()):
        assert np.array_equal(x, y)
    a.k_close(); b.k_close()


def test_episode_logger_from_device_buffers(tmp_path):
    """SURVEY 8f rank 3 on the GPU: a full scripted-policy episode logged from the handle's device buffers (no host
    traffic on the step path), then read back: the files hold exactly the actions taken and the observations returned."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.episode_log import EpisodeLogger
    from gym_kmanip_amd.model import MAX_EPISODE_STEPS
    n = 32
    e = env_hip.make("KManipSoloArm", num_envs=n, seed=31, auto_reset=False)
    cm = e.cm
    lg = EpisodeLogger(str(tmp_path), n, cm.nlink, cm.act_dim, device="cuda", env_ids=[0, 7, 31],
                       info={"sim": True, "env_id": "KManipSoloArm"}, backend="npz")
    e.k_reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    acts, qps, qvs = [], [], []
    sq, sv = cm.obs_slices["q_pos"], cm.obs_slices["q_vel"]
    for k in range(MAX_EPISODE_STEPS):
        act = e.scripted_action(generator=gen)
        e.step_flat(act)
        lg.step(act, e.obs[:, sq], e.obs[:, sv])
        acts.append(act.cpu().numpy()); qps.append(e.obs[:, sq].cpu().numpy()); qvs.append(e.obs[:, sv].cpu().numpy())
    assert (e.done.cpu().numpy() & KM_DONE_TRUNCATED).all()
    paths = lg.end_episode()
    assert len(paths) == 3
    z = np.load(paths[1])                                                   # env 7
    assert np.array_equal(z["action"], np.stack(acts)[:, 7])
    assert np.array_equal(z["observations/qpos"], np.stack(qps)[:, 7].astype(np.float32))
    assert np.array_equal(z["observations/qvel"], np.stack(qvs)[:, 7].astype(np.float32))
    e.k_close()


def test_raw_step_rejects_malformed_tensors():
    """step_flat / step_chunk hand raw pointers to the kernel: wrong dtype / device / shape / layout must raise, not run."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.lib import KManipError
    e = env_hip.make("KManipSoloArm", num_envs=8)
    e.k_reset()
    good = torch.zeros((8, 7), dtype=torch.float32, device="cuda")
    e.step_flat(good)
    for bad in (good.double(), good.cpu(), torch.zeros((7, 7), dtype=torch.float32, device="cuda"),
                torch.zeros((8, 14), dtype=torch.float32, device="cuda")[:, ::2], good.cpu().numpy()):
        with pytest.raises(KManipError):
            e.step_flat(bad)
    with pytest.raises(KManipError):
        e.step_chunk(torch.zeros((3, 8, 7), dtype=torch.float64, device="cuda"))
    with pytest.raises(KManipError):
        e.step_chunk(torch.zeros((3, 8, 7), dtype=torch.float32, device="cuda"), obs=torch.zeros((3, 8, 27), dtype=torch.float32, device="cuda"))
    # round-4 entry points: observe / the one-launch camera render / the record selection check their arguments too
    with pytest.raises(KManipError):
        e.observe(obs=torch.zeros((8, 27), dtype=torch.float32, device="cuda"))
    with pytest.raises(KManipError):
        e.observe(reward=torch.zeros(7, dtype=torch.float64, device="cuda"))
    with pytest.raises(KManipError):
        e.select_reward_done_record(0)                         # nothing bound
    with pytest.raises(KManipError):
        e.render_cameras(["grip_l"])                           # the single-arm model has no such camera
    with pytest.raises(KManipError):
        e.render_cameras(["head"], out={"head": torch.zeros((8, 480, 640, 3), dtype=torch.float32, device="cuda")})
    import ctypes as C
    five = (C.c_int32 * 5)(2, 2, 2, 2, 2); ptrs = (C.c_void_p * 5)()
    assert e.L.kmanip_render_rgb_multi(e.h, 5, five, five, five, ptrs, None) != 0      # more jobs than cameras
    assert b"bad arguments" in e.L.kmanip_last_error(e.h)
    e.k_close()
    for n in (0, -3):                                            # an empty / negative batch is refused at create
        with pytest.raises(KManipError):
            env_hip.make("KManipSoloArm", num_envs=n)


def test_checkpoint_restores_the_spawn_stream():
    """get_state + get_episode is a complete checkpoint: a restored handle reproduces the original run bit for bit ACROSS
    the next auto-reset (the cube spawn is keyed by seed, global env id and the episode counter)."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    n = 64
    a = env_hip.make("KManipSoloArm", num_envs=n, seed=41, env_id_offset=100)
    a.k_reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(6)
    acts = [(torch.rand((n, 7), generator=gen, device="cuda") * 2 - 1) for _ in range(100)]
    for k in range(70):                                   # into episode 1
        a.step_flat(acts[k])
    ck = a.checkpoint()
    assert (ck[5] == 1).all() and (ck[4] == 6).all()
    b = env_hip.make("KManipSoloArm", num_envs=n, seed=41, env_id_offset=100)
    b.restore(ck)
    for k in range(70, 100):                              # 70 + 58 = 128: crosses the second auto-reset
        a.step_flat(acts[k]); b.step_flat(acts[k])
        if k in (75, 99):
            assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done), k
    for k in range(100, 130):
        act = acts[k - 100]
        a.step_flat(act); b.step_flat(act)
    assert torch.equal(a.obs, b.obs)
    for x, y in zip(a.checkpoint(), b.checkpoint()):
        assert np.array_equal(x, y)
    assert (a.get_episode() == 2).all()
    # without the episode counter the spawn after the next reset differs
    c = env_hip.make("KManipSoloArm", num_envs=n, seed=41, env_id_offset=100)
    c.set_state(*ck[:5])
    for k in range(70, 130):
        c.step_flat(acts[k % 100])
    assert not torch.equal(a.obs, c.obs)
    for e in (a, b, c):
        e.k_close()


def test_seam_k_step_device_dict_no_host_roundtrip():
    """k_step with a dict of DEVICE tensors == step_flat on the packed action, returns device tensors only (sim_time too),
    and keeps the caller's current device."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    n = 128
    a = env_hip.make("KManipDualArm", num_envs=n, seed=3); b = env_hip.make("KManipDualArm", num_envs=n, seed=3)
    a.k_reset(); b.k_reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    for k in range(66):
        flat = (torch.rand((n, a.cm.act_dim), generator=gen, device="cuda") * 2 - 1)
        d = {key: flat[:, sl] for key, sl in a.cm.act_slices.items()}          # strided device views, reference key names
        terminated, reward, discount, obs, sim_time = a.k_step(d)
        b.step_flat(flat)
        assert torch.equal(a.obs, b.obs) and torch.equal(reward, b.reward)
        assert all(t.is_cuda for t in (terminated, reward, discount, sim_time)) and all(v.is_cuda for v in obs.values())
    assert not terminated.any() and (discount == 1).all()
    # 66 steps = one auto-reset at 64 + 2 steps: sim_time is each env's data.time
    assert torch.allclose(sim_time, torch.full((n,), 2 * 0.02, dtype=torch.float64, device="cuda"), atol=0, rtol=0)
    assert list(obs.keys()) == ["q_pos", "q_vel", "cube_pos", "cube_orn"]
    assert torch.cuda.current_device() == 0
    a.k_close(); b.k_close()


@pytest.mark.parametrize("env", ["KManipSoloArmQPos", "KManipDualArmQPos"])
def test_forearm_cylinder_section_on_the_cube(env):
    """Capsule sections (KModelDesc.sphere_seg): the forearm / elbow housings are the ends of capsules, and against the cube the
    collider is the closest point of the link's segment -- the CYLINDER SECTION between two housings touches the cube, not only
    its end spheres.  The cube is put 2.5 cm beside the MIDDLE of the forearm segment (radius 3 cm, cube half size 2 cm: the
    section overlaps it, both end spheres are 5 cm and more away): the device and the oracle report the same contact bit -- the
    forearm candidate's -- and the same step; with sphere_seg zeroed there is no contact at all."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from oracle.oracle import Oracle
    import copy
    cm = compile_model(env, auto_reset=False)
    names = [s["name"] for s in cm.asset["spheres"]]
    s_fore = names.index("forearm_r")
    seg = np.array(cm.asset["spheres"][s_fore]["seg"])
    assert np.linalg.norm(seg) > 0.08                                   # a real section between the two housings
    nl = cm.nlink
    orc = Oracle(cm, 1, seed=0); orc.reset()
    qpos, qvel, ctrl, warm, step = orc.get_state()
    xpos, xquat, _, _ = orc.fk(qpos[0])
    l = cm.asset["spheres"][s_fore]["link"]
    w, x, y, z = xquat[l]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    mid = xpos[l] + R @ (0.5 * seg)
    axis = R @ seg / np.linalg.norm(seg)
    side = np.cross(axis, [0.0, 0.0, 1.0]); side /= np.linalg.norm(side)
    cube = mid + 0.045 * side                                            # 3 cm radius + 2 cm half size - 0.5 cm of penetration
    qpos[0, nl:nl + 3] = cube; qpos[0, nl + 3:] = [1, 0, 0, 0]
    for end in (xpos[l], xpos[l] + R @ seg):                            # the end spheres alone do not reach the cube
        assert np.linalg.norm(cube - end) > 0.03 + 0.02 * np.sqrt(3) + 1e-3
    dev = env_hip.KManipEnvHip(cm, num_envs=1, seed=0); dev.k_reset()
    dev.set_state(qpos=qpos, qvel=qvel, ctrl=ctrl, warm=warm, step=step); orc.set_state(qpos, qvel, ctrl, warm, step)
    act = np.zeros((1, cm.act_dim), dtype=np.float32)
    dev.step_flat(torch.from_numpy(act).cuda()); orc.step(act)
    bit = 1 << (8 + s_fore)
    mg, mo = int(dev.get_diag()[0][0]), int(orc.get_diag()[0][0])
    assert mg == mo and (mg & bit), (hex(mg), hex(mo))
    sg, so = dev.get_state(), orc.get_state()
    assert np.abs(sg[0] - so[0]).max() < TOL_Q and np.abs(sg[1] - so[1]).max() < TOL_V
    assert np.abs(sg[1][0, nl:nl + 3]).max() > 1e-3                     # the section pushed the cube
    dev.k_close()
    # the same state without the section: nothing touches
    cm0 = copy.deepcopy(cm)
    for s in range(cm0.desc.nsphere):
        for k in range(3):
            cm0.desc.sphere_seg[s][k] = 0.0
    o0 = Oracle(cm0, 1, seed=0); o0.reset(); o0.set_state(qpos, qvel, ctrl, warm, step)
    o0.step(act)
    assert (int(o0.get_diag()[0][0]) & 0xFFF00) == 0


def test_table_rectangle_gpu():
    """The finite table top (kmanip.h table_rect; oracle side: test_the_table_is_a_rectangle): a cube on the table, one straddling its
    edge and one beside it, stepped on the device and in the oracle -- contact masks bit for bit every step, states within the parity
    bars; the cube beside the table falls."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from oracle.oracle import Oracle
    from test_oracle_dynamics import table_edge_states
    cm = compile_model("KManipSoloArmQPos", auto_reset=False)
    nl = cm.nlink
    orc = Oracle(cm, 3, seed=0); orc.reset()
    st = table_edge_states(cm, orc)
    orc.set_state(*st)
    dev = env_hip.KManipEnvHip(cm, num_envs=3, seed=0); dev.k_reset()
    dev.set_state(qpos=st[0], qvel=st[1], ctrl=st[2], warm=st[3], step=st[4])
    act = np.zeros((3, cm.act_dim), dtype=np.float32)
    seen = set()
    for k in range(30):
        dev.step_flat(torch.from_numpy(act).cuda()); orc.step(act)
        mg, mo = dev.get_diag()[0], orc.get_diag()[0]
        assert (mg == mo).all(), (k, [hex(int(x)) for x in mg], [hex(int(x)) for x in mo])
        seen.add(tuple(bin(int(x) & 0xFF).count("1") for x in mg))
        sg, so = dev.get_state(), orc.get_state()
        assert np.abs(sg[0] - so[0]).max() < TOL_Q and np.abs(sg[1] - so[1]).max() < TOL_V, k
    assert (4, 2, 0) in seen
    assert dev.get_state()[0][2, nl + 2] < cm.desc.table_z - 0.5
    dev.k_close()


def test_bound_reward_done_record():
    """kmanip_bind_reward_done_record: every step also writes the packed (reward, done) record of the multi-GPU exchange into the
    bound buffer the CALLER selected (kmanip_select_reward_done_record; the first after the bind) -- the library keeps no parity of
    its own --; reward / done themselves are written as always; unbinding stops it; a chunked launch writes no record."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.lib import KManipError
    n = 37
    e = env_hip.make("KManipSoloArm", num_envs=n, seed=2)
    e.k_reset(); e.set_state(step=(57 + np.arange(n) % 7).astype(np.int32))      # some envs time out during the test
    rec = [torch.full((n, 2), -7.0, dtype=torch.float64, device="cuda") for _ in range(2)]
    with pytest.raises(KManipError):
        e.bind_reward_done_record(rec[0], None)
    e.bind_reward_done_record(rec[0], rec[1])
    saw_done = False
    with pytest.raises(KManipError):
        e.select_reward_done_record(2)
    order = [0, 0, 1, 0, 1, 1]                 # any order: e.g. a step nobody exchanges does not flip anything
    for k in range(6):
        if k > 0:
            e.select_reward_done_record(order[k])
        other = rec[1 - order[k]].clone()
        e.step_flat(e.sample_action())
        assert torch.equal(rec[1 - order[k]], other)
        r = rec[order[k]].cpu().numpy()
        assert np.array_equal(r[:, 0], e.reward.cpu().numpy()) and np.array_equal(r[:, 1], e.done.cpu().numpy().astype(np.float64)), k
        saw_done |= bool(e.done.cpu().numpy().any())
    assert saw_done
    keep = [t.clone() for t in rec]
    e.step_chunk(torch.stack([e.sample_action().clone() for _ in range(2)]))
    e.bind_reward_done_record(None, None)
    e.step_flat(e.sample_action())
    assert all(torch.equal(a, b) for a, b in zip(rec, keep))
    e.k_close()


def test_sampled_step_timing():
    """kmanip_enable_timing(k): events around every k-th step only (an event pair costs the stream ~5 us: bench.py samples long
    windows); the summary counts the sampled steps, their average is a k_step launch's duration, and the steps are the same steps."""
    torch = _torch()
    from gym_kmanip_amd import env_hip
    n = 256
    a = env_hip.make("KManipSoloArm", num_envs=n, seed=5); b = env_hip.make("KManipSoloArm", num_envs=n, seed=5)
    a.k_reset(); b.k_reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(2)
    acts = [torch.rand((n, a.cm.act_dim), generator=gen, device="cuda") * 2 - 1 for _ in range(16)]
    a.enable_timing(4); b.enable_timing(True)
    for act in acts:
        a.step_flat(act); b.step_flat(act)
    _, dyn_a, _, nt_a = a.timing_summary(); _, dyn_b, _, nt_b = b.timing_summary()
    assert (nt_a, nt_b) == (4, 16)
    assert 0.01 < dyn_a / nt_a < 20.0 and 0.01 < dyn_b / nt_b < 20.0           # milliseconds per launch, both plausible
    assert all(np.array_equal(x, y) for x, y in zip(a.get_state(), b.get_state()))
    a.enable_timing(False)
    a.step_flat(acts[0])
    assert a.timing_summary()[3] == 0
    a.k_close(); b.k_close()
