"""The reference's only test is gymnasium's check_env over its 8 env ids (tests/test_env.py:8-24).  gymnasium is not
installed here (or on the GPU box), so this file restates what check_env pins -- reset(seed) signature and return, obs
membership in observation_space (keys / order / shape / dtype / bounds), action sampling, the step 5-tuple's types, render --
on real HIP outputs for all 8 ids, single-env (squeeze) shapes like the reference's, plus the batched and device-resident
variants of the same shell and the TimeLimit(64) truncation gym.make adds (__init__.py:28,247)."""
import os

import numpy as np
import pytest

from gym_kmanip_amd.model import ENV_SPECS

pytestmark = pytest.mark.gpu
ALL_IDS = sorted(ENV_SPECS)


def _member(space, obs):
    assert list(obs.keys()) == list(space.spaces.keys())
    for k, sp in space.spaces.items():
        assert sp.contains(obs[k]), (k, np.asarray(obs[k]).shape, np.asarray(obs[k]).dtype, np.asarray(obs[k]).min(), np.asarray(obs[k]).max())


@pytest.mark.parametrize("env_id", ALL_IDS)
def test_check_env_contract_single_env(env_id):
    from gym_kmanip_amd.gym_shell import KManipEnv
    env = KManipEnv(env_id, num_envs=1, squeeze=True, seed=0)
    obs, info = env.reset(seed=123)
    assert isinstance(info, dict) and info["step"] == 0 and info["q_len"] == env.q_len
    _member(env.observation_space, obs)
    obs_b, _ = env.reset(seed=123)                        # same seed -> same first observation
    for k in obs:
        assert np.array_equal(obs[k], obs_b[k]), k
    obs_c, _ = env.reset(seed=124)                        # another seed -> another cube spawn (state key or pixels)
    assert any(not np.array_equal(obs[k], obs_c[k]) for k in obs)
    obs_d, _ = env.reset()                                # no seed: next episode of the same stream -> again a new spawn
    assert any(not np.array_equal(obs_c[k], obs_d[k]) for k in obs)
    env.action_space.seed(7)
    for k in range(3):
        a = env.action_space.sample()
        assert env.action_space.contains(a)
        o, r, terminated, truncated, info = env.step(a)
        _member(env.observation_space, o)
        assert isinstance(r, float) and isinstance(terminated, bool) and isinstance(truncated, bool) and isinstance(info, dict)
        assert np.isfinite(r) and not terminated and not truncated
        assert info["step"] == k + 1 and abs(float(np.asarray(info["sim_time"]).ravel()[0]) - 0.02 * (k + 1)) < 1e-12
    img = env.render()
    assert img.shape == (480, 640, 3) and img.dtype == np.uint8 and np.unique(img).size > 3
    env.close()


def test_time_limit_truncation_and_cameras_see_the_scene():
    from gym_kmanip_amd.gym_shell import KManipEnv
    env = KManipEnv("KManipSoloArmVision", num_envs=1, squeeze=True, seed=5)
    obs, _ = env.reset(seed=5)
    a = {k: np.zeros(sp.shape, np.float32) for k, sp in env.action_space.spaces.items()}
    for k in range(64):
        obs, r, terminated, truncated, info = env.step(a)
        assert truncated == (k == 63) and not terminated
    head = obs["camera/head"]
    red = (head[..., 0] > 150) & (head[..., 1] < 60) & (head[..., 2] < 60)        # the cube (rgba 1 0 0, scene.xml:20)
    grey = (np.abs(head[..., 0].astype(int) - head[..., 1]) < 3) & (head[..., 0] > 20)   # the table (rgba .2 .2 .2)
    assert red.sum() > 20 and 0.05 < grey.mean() < 0.3            # the 0.8 x 0.4 m table top fills a tenth of the head image, background around it
    assert obs["camera/grip_r"].shape == (40, 60, 3)
    env.close()


@pytest.mark.parametrize("env_id", ["KManipSoloArm", "KManipDualArmVision", "KManipTorsoVision"])
def test_batched_and_device_resident_variants(env_id):
    import torch
    from gym_kmanip_amd.gym_shell import KManipEnv
    n = 5
    host = KManipEnv(env_id, num_envs=n, seed=3)
    dev = KManipEnv(env_id, num_envs=n, seed=3, device_outputs=True)
    oh, _ = host.reset(seed=3); od, _ = dev.reset(seed=3)
    host.action_space.seed(1)
    for k in range(4):
        a = host.action_space.sample()
        act = {key: np.repeat(v[None], n, axis=0) for key, v in a.items()}
        oh, rh, th, trh, ih = host.step(act)
        od, rd, td, trd, idv = dev.step({key: torch.from_numpy(v).cuda() for key, v in act.items()})
        assert list(oh.keys()) == list(od.keys()) == list(host.observation_space.spaces.keys())
        for key, sp in host.observation_space.spaces.items():
            assert oh[key].shape == (n,) + sp.shape and oh[key].dtype == sp.dtype
            assert od[key].is_cuda and np.array_equal(od[key].cpu().numpy().astype(sp.dtype), oh[key]), key
        assert rh.shape == (n,) and rh.dtype == np.float64 and th.shape == (n,) and trh.shape == (n,)
        assert rd.is_cuda and td.is_cuda and trd.is_cuda and np.array_equal(rd.cpu().numpy(), rh)
        assert idv["sim_time"].is_cuda and np.allclose(idv["sim_time"].cpu().numpy(), 0.02 * (k + 1), atol=1e-12)
    host.close(); dev.close()


@pytest.mark.parametrize("env_id", ["KManipSoloArm", "KManipTorso"])
def test_rgb_render_parity_vs_oracle(env_id):
    """uint8 RGB of every camera the model has vs the oracle's restatement of the same ray caster + Lambert shading, on stepped
    states: equal up to one grey level, except silhouette-grazing rays (< 0.1 % of the pixels)."""
    import torch
    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.lib import KManipError
    from gym_kmanip_amd.model import KM_CAM_INDEX, compile_model
    from oracle.oracle import Oracle
    cm = compile_model(env_id)
    n = 4
    dev = env_hip.KManipEnvHip(cm, num_envs=n, seed=4); orc = Oracle(cm, n, seed=4)
    dev.k_reset(); orc.reset()
    rng = np.random.default_rng(3)
    for k in range(14):
        act = rng.uniform(-1, 1, (n, cm.act_dim)).astype(np.float32)
        dev.step_flat(torch.from_numpy(act).cuda()); orc.step(act)
    qpos = orc.get_state()[0]
    for name, ci in KM_CAM_INDEX.items():
        if not cm.desc.cam_present[ci]:
            with pytest.raises(KManipError, match="no camera"):
                dev.render_rgb(name)
            continue
        h, w = (48, 64) if name in ("top", "head") else (40, 60)
        img = dev.render_rgb(name, h, w).cpu().numpy()
        assert img.shape == (n, h, w, 3) and img.dtype == np.uint8
        for e in range(n):
            ref = orc.render_rgb(qpos[e], ci, h, w)
            diff = np.abs(img[e].astype(int) - ref.astype(int)).max(axis=-1)
            assert (diff > 1).mean() < 1e-3, (name, e, (diff > 1).sum())
    dev.k_close()


def test_depth_render_bound_to_the_step():
    """BASELINE config 5 wording: the depth render runs IN the step -- kmanip_bind_step_depth makes every kmanip_step end by
    rendering the state it produced; identical to a separate render_depth call afterwards."""
    import torch
    from gym_kmanip_amd import env_hip
    n = 64
    a = env_hip.make("KManipSoloArm", num_envs=n, seed=2); b = env_hip.make("KManipSoloArm", num_envs=n, seed=2)
    a.k_reset(); b.k_reset()
    buf = a.bind_step_depth("grip_r", 64, 64)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0)
    for k in range(5):
        act = torch.rand((n, 7), generator=gen, device="cuda") * 2 - 1
        a.step_flat(act); b.step_flat(act)
        assert torch.equal(buf, b.render_depth("grip_r", 64, 64)), k
    a.bind_step_depth(None)
    keep = buf.clone()
    a.step_flat(act)
    assert torch.equal(buf, keep)                          # unbound: the buffer is no longer written
    a.k_close(); b.k_close()


def test_episode_logger_with_camera_frames_from_device(tmp_path):
    """log_h5py.cam / step on the GPU path: the selected envs' gripper-camera frames go from kmanip_render_rgb into the
    logger's device ring and come back out of the episode file unchanged (npz tree here: h5py is absent on the GPU box)."""
    import torch
    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.episode_log import EpisodeLogger
    from gym_kmanip_amd.model import CAMERAS, MAX_EPISODE_STEPS
    n = 8
    e = env_hip.make("KManipSoloArmVision", num_envs=n, seed=6, auto_reset=False)
    cm = e.cm
    lg = EpisodeLogger(str(tmp_path), n, cm.nlink, cm.act_dim, device="cuda", env_ids=[1, 6], info={"sim": True}, backend="npz")
    cam = CAMERAS["grip_r"]
    lg.cam(cam)
    e.k_reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    frames = []
    sq, sv = cm.obs_slices["q_pos"], cm.obs_slices["q_vel"]
    for k in range(MAX_EPISODE_STEPS):
        act = e.scripted_action(generator=gen)
        e.step_flat(act)
        img = e.k_render(cam)
        lg.step(act, e.obs[:, sq], e.obs[:, sv], images={cam.log_name: img})
        frames.append(img[6].cpu().numpy())
    z = np.load(lg.end_episode()[1])
    got = z["observations/images/grip_r"]
    assert got.shape == (MAX_EPISODE_STEPS, cam.h, cam.w, 3) and got.dtype == np.uint8
    assert np.array_equal(got, np.stack(frames)) and np.unique(got).size > 3
    e.k_close()


def test_shell_log_h5py_kwargs_and_examples(tmp_path, monkeypatch):
    """env_base.py:82-101,231-263 through the shell: KManipEnv(log_h5py=True, log_prefix=...) makes a fresh log directory,
    writes one episode file per reset for the logged envs (camera frames included for a *Vision id) and flushes on close();
    the two example scripts (the reference's 2_log_with_h5py.py / 2_synthetic_data.py, batched) run end to end."""
    import gym_kmanip_amd.gym_shell as gs
    monkeypatch.setattr(gs, "DATA_DIR", str(tmp_path / "data"))
    env = gs.KManipEnv("KManipSoloArmVision", num_envs=4, log_h5py=True, log_prefix="unit", log_env_ids=[1, 3], log_backend="npz")
    assert os.path.isdir(env.log_dir) and os.path.basename(env.log_dir).startswith("unit.")
    env.action_space.seed(1)
    acts = []
    for ep in range(2):
        env.reset()
        for k in range(5):
            a = {key: np.stack([sp.sample() for _ in range(4)]) for key, sp in env.action_space.spaces.items()}
            acts.append(np.concatenate([a[key].reshape(4, -1) for key in env.action_space.spaces], axis=1))
            obs, rew, term, trunc, info = env.step(a)
    env.close()
    files = sorted(os.listdir(env.log_dir))
    assert files == ["episode_1_env1.npz", "episode_1_env3.npz", "episode_2_env1.npz", "episode_2_env3.npz"], files
    f = np.load(os.path.join(env.log_dir, "episode_2_env3.npz"))
    assert f["observations/qpos"].shape == (64, 10) and f["action"].shape == (64, 7)
    assert np.array_equal(f["action"][:5], np.stack([a[3] for a in acts[5:]]).astype(np.float32))
    assert not f["action"][5:].any()
    assert np.abs(f["observations/qpos"][4] - obs["q_pos"][3].astype(np.float32)).max() == 0
    cam = [c for c in env.cameras if c.name == "grip_r"][0]
    frames = f["observations/images/grip_r"]
    assert frames.shape == (64, cam.h, cam.w, 3) and frames.dtype == np.uint8
    assert np.array_equal(frames[4], obs[cam.log_name][3]) and frames[:5].any() and not frames[5:].any()
    # the examples
    from gym_kmanip_amd.examples import log_episodes, synthetic_data
    d = log_episodes.main(["--env", "KManipSoloArm", "--num-envs", "8", "--episodes", "1", "--log-envs", "0", "2"])
    assert len(os.listdir(d)) == 2
    d = synthetic_data.main(["--num-envs", "64", "--episodes", "1", "--log-envs", "0", "--log-dir", str(tmp_path / "synth")])
    assert len(os.listdir(d)) == 1


def test_rgb_render_reference_resolution_vs_oracle():
    """The camera observations at the reference's own resolutions (head 480 x 640, grip_r 40 x 60: __init__.py:157-161) -- the
    float32, rectangle-culled, four-pixels-per-lane RGB kernel against the oracle's float64 per-pixel ray caster: equal up to one
    grey level except silhouette-grazing rays (< 0.1 % of the pixels), and the culled fast path (table / background only) and
    the full path both occur in the head image."""
    import torch
    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.model import KM_CAM_INDEX, compile_model
    from oracle.oracle import Oracle
    cm = compile_model("KManipSoloArmVision")
    n = 2
    dev = env_hip.KManipEnvHip(cm, num_envs=n, seed=6); orc = Oracle(cm, n, seed=6)
    dev.k_reset(); orc.reset()
    rng = np.random.default_rng(5)
    for k in range(20):
        act = rng.uniform(-1, 1, (n, cm.act_dim)).astype(np.float32)
        dev.step_flat(torch.from_numpy(act).cuda()); orc.step(act)
    qpos = orc.get_state()[0]
    for name in cm.cameras:
        img = dev.render_rgb(name).cpu().numpy()
        h, w = img.shape[1:3]
        assert (h, w) == ((480, 640) if name == "head" else (40, 60))
        for e in range(n):
            ref = orc.render_rgb(qpos[e], KM_CAM_INDEX[name], h, w)
            diff = np.abs(img[e].astype(int) - ref.astype(int)).max(axis=-1)
            assert (diff > 1).mean() < 1e-3, (name, e, int((diff > 1).sum()))
            if name == "head":
                red = (ref[..., 0] > 100) & (ref[..., 1] == 0)
                assert red.any() and (ref[..., 0] == ref[..., 1]).mean() > 0.5      # cube pixels and table / background pixels
    dev.k_close()


def test_render_shows_the_table_rectangle():
    """The table top is a finite rectangle in the camera images too: the fixed `top` camera (0 0 1.3 looking at the table body, fovy
    78: _env_solo_arm.xml:14) sees the whole 0.8 m x 0.4 m top with background around it -- the grey region's bounding box is what the
    pinhole model predicts for the rectangle's corners (within two pixels), and depth beside the table is zfar."""
    import torch
    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.model import compile_model
    cm = compile_model("KManipSoloArmVision")
    d = cm.desc
    dev = env_hip.KManipEnvHip(cm, num_envs=1, seed=0); dev.k_reset()
    h, w = 240, 320
    img = dev.render_rgb("top", h, w).cpu().numpy()[0]
    grey = (img[..., 0] == img[..., 1]) & (img[..., 0] > 20)
    black = img.max(axis=-1) == 0
    assert 0.04 < grey.mean() < 0.3 and black.mean() > 0.3
    rows, cols = np.where(grey)
    # pinhole prediction: camera at (0, 0, 1.3) looking at (0, .6, .5); z = (cam - target) normalised, x = up x z, y = z x x
    co = np.array([0.0, 0.0, 1.3]); to = np.array([0.0, 0.6, 0.5])
    z = (co - to) / np.linalg.norm(co - to); x = np.cross([0, 0, 1.0], z); x /= np.linalg.norm(x); y = np.cross(z, x)
    f = 0.5 * h / np.tan(np.radians(78.0) / 2)
    rr, cc = [], []
    for px in (d.table_rect[0], d.table_rect[1]):
        for py in (d.table_rect[2], d.table_rect[3]):
            v = np.array([px, py, d.table_z]) - co
            zc = -v @ z
            cc.append(v @ x * f / zc + 0.5 * w - 0.5); rr.append(-(v @ y) * f / zc + 0.5 * h - 0.5)
    assert abs(rows.min() - max(min(rr), 0)) <= 2 and abs(rows.max() - min(max(rr), h - 1)) <= 2
    assert abs(cols.min() - max(min(cc), 0)) <= 2 and abs(cols.max() - min(max(cc), w - 1)) <= 2
    dep = dev.render_depth("top", 48, 64).cpu().numpy()[0]
    assert (dep >= d.cam_zfar - 1e-6).mean() > 0.3 and (dep < 2.0).mean() > 0.04
    dev.k_close()


def test_all_cameras_in_one_launch_equal_the_single_renders():
    """kmanip_render_rgb_multi (every camera of a *Vision observation in one launch: grid = envs x cameras) writes, bit for bit,
    what one kmanip_render_rgb call per camera writes -- at the reference resolutions, for a two-arm id's three cameras too."""
    import torch
    from gym_kmanip_amd import env_hip
    for env_id, n in (("KManipSoloArmVision", 24), ("KManipTorsoVision", 8)):
        e = env_hip.make(env_id, num_envs=n, seed=3)
        e.k_reset()
        for k in range(6):
            e.step_flat(e.sample_action())
        multi = e.render_cameras()
        assert list(multi) == e.cm.cameras and len(multi) >= 2
        for cam, img in multi.items():
            assert torch.equal(img, e.render_rgb(cam)), (env_id, cam)
        e.k_close()


def test_render_behind_gives_the_images_of_its_step():
    """pipeline.RenderBehind (kmanip_snapshot_render_state / kmanip_set_render_source): the images rendered on the second stream
    while the next step already runs are, bit for bit, those a render right after their own step produces -- RGB of every camera
    and the depth image; the live state is what the renders read again afterwards."""
    import torch
    from gym_kmanip_amd import env_hip
    from gym_kmanip_amd.pipeline import RenderBehind
    n, steps = 96, 9
    a = env_hip.make("KManipSoloArmVision", num_envs=n, seed=5)
    b = env_hip.make("KManipSoloArmVision", num_envs=n, seed=5)
    a.k_reset(); b.k_reset()
    acts = [a.sample_action(ahead=k).clone() for k in range(steps)]
    rb = RenderBehind(a, depth=("grip_r", 64, 64))
    want = []
    for k in range(steps):
        b.step_flat(acts[k])
        want.append({c: img.clone() for c, img in b.render_cameras().items()})
        want[-1]["depth"] = b.render_depth("grip_r", 64, 64).clone()
    got = []
    for k in range(steps):
        a.step_flat(acts[k])
        assert rb.after_step() == k
        if k:
            got.append({c: img.clone() for c, img in rb.images(k - 1).items()})      # (while step k's render is in flight)
    got.append({c: img.clone() for c, img in rb.images(steps - 1).items()})
    torch.cuda.synchronize()
    assert any(not torch.equal(want[0][c], want[steps - 1][c]) for c in want[0])     # the scene moved: a stale snapshot would show
    for k in range(steps):
        for c in want[k]:
            assert torch.equal(got[k][c], want[k][c]), (k, c)
    with pytest.raises(ValueError):
        rb.images(steps - 3)
    # a consumer on a stream of its own: the render that rewrites set k & 1 waits for the reads enqueued there
    side = torch.cuda.Stream()
    got2 = []
    for k in range(steps):
        a.step_flat(acts[k]); b.step_flat(acts[k])
        j = rb.after_step()
        with torch.cuda.stream(side):
            imgs = rb.images(j)
            for _ in range(4):                          # (a slow consumer: the clone sits behind other work on its stream)
                torch.mm(torch.ones(512, 512, device="cuda"), torch.ones(512, 512, device="cuda"))
            got2.append({c: img.clone() for c, img in imgs.items()})
        want2 = {c: img.clone() for c, img in b.render_cameras().items()}
        want2["depth"] = b.render_depth("grip_r", 64, 64).clone()
        want.append(want2)
    torch.cuda.synchronize()
    for k in range(steps):
        for c in want[steps + k]:
            assert torch.equal(got2[k][c], want[steps + k][c]), (k, c)
    # depth through a snapshot, and back to the live state
    d_live = a.render_depth("grip_r", 64, 64).clone()
    a.snapshot_render_state(0)
    a.step_flat(acts[0]); b.step_flat(acts[0])
    a.set_render_source(0)
    assert torch.equal(a.render_depth("grip_r", 64, 64), d_live)
    a.set_render_source(-1)
    assert torch.equal(a.render_depth("grip_r", 64, 64), b.render_depth("grip_r", 64, 64))
    assert not torch.equal(a.render_depth("grip_r", 64, 64), d_live)
    with pytest.raises(RuntimeError):
        b.set_render_source(1)                  # no snapshot was ever taken on b
    a.k_close(); b.k_close()


def test_synthetic_data_example_logs_the_same_frames_behind_and_in_sequence(tmp_path):
    """examples/synthetic_data.py on a *Vision id: the frames rendered behind the steps (RenderBehind + EpisodeLogger.late_images)
    give, file for file and byte for byte, the datasets the in-sequence render gives."""
    from gym_kmanip_amd.examples import synthetic_data
    common = ["--env", "KManipSoloArmVision", "--num-envs", "32", "--episodes", "1", "--log-envs", "0", "5"]
    da = synthetic_data.main(common + ["--log-dir", str(tmp_path / "behind")])
    db = synthetic_data.main(common + ["--log-dir", str(tmp_path / "seq"), "--render-in-sequence"])
    fa, fb = sorted(os.listdir(da)), sorted(os.listdir(db))
    assert fa == fb and len(fa) == 2
    def read_episode(path):          # (.npz where h5py is not installed -- this image -- else the reference's .hdf5)
        if path.endswith(".npz"):
            return np.load(path)
        import h5py
        return h5py.File(path, "r")
    for name in fa:
        a, b = read_episode(os.path.join(da, name)), read_episode(os.path.join(db, name))
        for k in ("observations/images/head", "observations/images/grip_r", "observations/qpos", "observations/qvel", "action"):
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), (name, k)
        head = np.asarray(a["observations/images/head"])
        assert head.shape == (64, 480, 640, 3) and head.any() and not np.array_equal(head[0], head[-1])
