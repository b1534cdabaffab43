#!/usr/bin/env python3
"""Random-action episodes logged in the ACT / LeRobot tree -- what the reference's examples/2_log_with_h5py.py does with
one MuJoCo env per process, here with a batch of envs on one MI355X behind the same `KManipEnv` shell.

    python -m gym_kmanip_amd.examples.log_episodes [--env KManipSoloArmVision] [--num-envs 64] [--episodes 2] [--log-envs 0 1]

Every env steps on the device; the logged envs' observation / action rows (and camera frames for the *Vision ids) collect in
device-resident rings and are written once per episode: `episode_<n>_env<e>.hdf5` when h5py is importable, otherwise
`.npz` archives with the same member paths (observations/qpos, observations/qvel, action, observations/images/<cam>,
metadata)."""
import argparse
import os
import pprint

import numpy as np

from gym_kmanip_amd.gym_shell import KManipEnv
from gym_kmanip_amd.model import MAX_EPISODE_STEPS


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="KManipSoloArm")
    ap.add_argument("--num-envs", type=int, default=64)
    ap.add_argument("--episodes", type=int, default=2)
    ap.add_argument("--log-envs", type=int, nargs="*", default=[0])
    ap.add_argument("--log-prefix", default="h5py_test")
    args = ap.parse_args(argv)
    env = KManipEnv(args.env, num_envs=args.num_envs, log_h5py=True, log_prefix=args.log_prefix, log_env_ids=args.log_envs)
    env.action_space.seed(0)
    print(f"Running {args.num_envs} x {args.env} for {args.episodes} episodes of {MAX_EPISODE_STEPS} steps")
    for _ in range(args.episodes):
        env.reset()
        for _ in range(MAX_EPISODE_STEPS):
            action = {k: np.stack([s.sample() for _ in range(args.num_envs)]) for k, s in env.action_space.spaces.items()}
            _, _, terminated, truncated, _ = env.step(action)
            if np.all(terminated | truncated):
                break
    env.close()
    first = sorted(os.listdir(env.log_dir))[0]
    path = os.path.join(env.log_dir, first)
    print(f"Opening {path}")
    if first.endswith(".npz"):
        f = np.load(path, allow_pickle=False)
        print("members:", sorted(f.files))
        pprint.pprint(str(f["metadata"]))
        print(f["observations/qpos"][0], f["observations/qvel"][0], sep="\n")
    else:
        import h5py
        with h5py.File(path, "r") as f:
            print("root level keys:", list(f.keys()))
            pprint.pprint(dict(f["metadata"].attrs.items()))
            print(f["observations/qpos"][0], f["observations/qvel"][0], sep="\n")
    return env.log_dir


if __name__ == "__main__":
    main()
