"""gym_kmanip_amd -- MI355X-native backend for gym-kmanip's simulation hot path (env_sim.step + ik_mujoco.ik).

    import gym_kmanip_amd as k
    env = k.make("KManipSoloArm", num_envs=4096, device_outputs=True)     # KManipEnv-compatible shell (gym_shell.py)
    raw = k.make_backend("KManipTorso", num_envs=8192)                    # the backend itself (k_reset / k_step / k_render)
    two = k.make_interleaved("KManipSoloArm", 4096, k=2)                  # two batches in flight on two streams (pipeline.py)

The reference registers its eight env ids with gymnasium at import time (gym_kmanip/__init__.py:244-483) so that
`gym.make("KManipSoloArm")` works.  gymnasium is not importable in the build image, so nothing is registered on import here;
`register_envs()` does the same registration (same ids, `max_episode_steps = 64`, `nondeterministic=True`) when gymnasium is
there.  Importing this package loads no native code: the HIP library is loaded (and must exist) when an env is created.
"""
from .model import ENV_SPECS, MAX_EPISODE_STEPS

ENV_IDS = sorted(ENV_SPECS)


def make(env_id: str = "KManipSoloArm", **kwargs):
    """A `KManipEnv`-compatible env over the HIP backend (see gym_shell.KManipEnv for the kwargs)."""
    from .gym_shell import KManipEnv
    return KManipEnv(env_id, **kwargs)


def make_backend(env_id: str = "KManipSoloArm", **kwargs):
    """The backend object behind the reference's seam (env_hip.KManipEnvHip: k_reset / k_step / k_render / k_close)."""
    from . import env_hip
    return env_hip.make(env_id, **kwargs)


def make_interleaved(env_id: str = "KManipSoloArm", envs_per_batch: int = 4096, k: int = 2, **kwargs):
    """K independent batches on K streams (pipeline.InterleavedBatches): the shape that fills the GPU -- one batch's launch leaves
    half the SIMD time idle, a second batch in flight takes it (2 x 4096 single-arm envs: 10.9 M env steps/s against 6.5 M)."""
    from .pipeline import InterleavedBatches
    return InterleavedBatches(env_id, envs_per_batch, k=k, **kwargs)


def register_envs(gymnasium=None, suffix: str = ""):
    """Register the eight ids with gymnasium like gym_kmanip/__init__.py:244-483 does.  `gymnasium`: the module (imported when
    None; ImportError if absent).  `suffix` lets both packages be registered side by side (e.g. "-HIP").  Returns the ids."""
    if gymnasium is None:
        import gymnasium                                           # noqa: F811 (absent in the build image)
    ids = []
    for env_id in ENV_IDS:
        gymnasium.register(id=env_id + suffix, entry_point="gym_kmanip_amd.gym_shell:KManipEnv",
                           max_episode_steps=MAX_EPISODE_STEPS, nondeterministic=True,
                           kwargs={"env_id": env_id, "num_envs": 1, "squeeze": True})
        ids.append(env_id + suffix)
    return ids
