#!/usr/bin/env python3
"""Scripted data generation on the device -- the reference's examples/2_synthetic_data.py (random action with `eer_pos`
overwritten by the unit vector from the right end effector to the cube) for a whole batch, without the host in the loop:
the policy is a kernel (`kmanip_scripted_action`), the step takes its device action matrix, the logger's rings are device
tensors.  A *Vision id also logs its camera frames (the reference's log_h5py.cam / step): the heuristic acts on the state, so
the frames of step t are rendered BEHIND the steps (pipeline.RenderBehind: a qpos snapshot and a second stream) and reach the
logger while step t + 1 runs -- `--render-in-sequence` renders them before the next step instead.

    python -m gym_kmanip_amd.examples.synthetic_data [--env KManipSoloArm] [--num-envs 4096] [--episodes 10] [--log-envs 0 1 2 3]
"""
import argparse
import os
import time

from gym_kmanip_amd import env_hip
from gym_kmanip_amd.episode_log import EpisodeLogger
from gym_kmanip_amd.model import MAX_EPISODE_STEPS


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="KManipSoloArm")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--episodes", type=int, default=10)
    ap.add_argument("--log-envs", type=int, nargs="*", default=[0, 1, 2, 3])
    ap.add_argument("--log-dir", default=os.path.join(os.getcwd(), "data", "sim_synth"))
    ap.add_argument("--render-in-sequence", action="store_true", help="*Vision ids: render every step's frames before the next step starts")
    args = ap.parse_args(argv)
    import torch
    os.makedirs(args.log_dir, exist_ok=True)
    env = env_hip.make(args.env, num_envs=args.num_envs, auto_reset=False)
    q = env.cm.nlink
    log = EpisodeLogger(args.log_dir, args.num_envs, q, env.cm.act_dim, device=env.obs.device, env_ids=args.log_envs,
                        info={"sim": True, "env": args.env, "policy": "toward-cube heuristic"})
    from gym_kmanip_amd.model import CAMERAS
    from gym_kmanip_amd.pipeline import RenderBehind
    for name in env.cm.cameras:                                                  # (none unless the id is a *Vision one)
        log.cam(CAMERAS[name])
    behind = RenderBehind(env) if (env.cm.cameras and not args.render_in_sequence) else None
    gen = torch.Generator(device=env.obs.device); gen.manual_seed(0)
    t0 = time.time()
    closest = None
    for ep in range(args.episodes):
        env.k_reset()
        due = None                                                               # (step index of the renderer, row of the logger)
        for _ in range(MAX_EPISODE_STEPS):
            act = torch.rand((args.num_envs, env.cm.act_dim), generator=gen, device=env.obs.device) * 2 - 1    # action_space.sample()
            env.scripted_action(act)                                                                           # eer_pos <- unit(cube - eer)
            env.step_flat(act)
            if behind is not None:
                k = behind.after_step()                                          # this step's frames: rendering from now on
                if due is not None:
                    log.late_images(due[1], behind.images(due[0]))               # the previous step's frames have had a whole step
                due = (k, log.step(act, env.obs[:, :q], env.obs[:, q:2 * q], images_later=True))
            else:
                log.step(act, env.obs[:, :q], env.obs[:, q:2 * q], images=env.render_cameras() if env.cm.cameras else None)
        if due is not None:
            log.late_images(due[1], behind.images(due[0]))
        paths = log.end_episode()
        closest = float(env.reward.max())
    torch.cuda.synchronize()
    dt = time.time() - t0
    n = args.episodes * MAX_EPISODE_STEPS * args.num_envs
    print(f"{n} env steps in {dt:.2f} s ({n / dt:.0f} env steps/s incl. logging); best final reward {closest:.3f}; last files: {paths[:2]}")
    env.k_close()
    return args.log_dir


if __name__ == "__main__":
    main()
