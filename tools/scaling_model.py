#!/usr/bin/env python3
"""A MODEL, not a measurement: what the (reward, done) exchange does to weak scaling, from launch times measured on ONE MI355X.

Every rank's k_step launch lasts as long as its slowest wave, and that varies from launch to launch (an IK problem crawling at
the reference's 100 n evaluation limit makes a 0.59 ms launch a 1.5-4 ms one about once in a hundred).  Ranks draw their launch
times independently (different envs), and an all-gather is a rendezvous: whoever arrives first waits.  With the exchange on the
step's stream every step lasts as long as the SLOWEST rank's step; gathered every K steps, ranks run free inside a block and
meet at its end, so a block lasts as long as the slowest rank's SUM over K steps -- the outliers average out.  The per-step
exchange gets the same effect from a ring of D record buffers: a rank waits for the exchange D steps back, not for the last one.  (An exchange that
overlaps the next step does not escape the rendezvous: the collective kernel spins on a compute unit until the last peer has
arrived, and the k_step workgroups it displaced start only then.  Lateness below the slack of an average wave -- the mean wave
runs 0.54 of the launch -- is absorbed; the model below ignores that and is pessimistic for the overlapped per-step exchange.)

   python tools/scaling_model.py profiles/r05_launch_times.txt [profiles/r05_launch_times_cap64.txt ...]
The input is one launch time per line (tests/tools/slow_launches.py with KM_LAUNCH_DUMP); ranks resample it with replacement."""
import sys
import numpy as np


def model(times, world, K, blocks=20000, seed=0, latency_ms=0.0):
    rng = np.random.default_rng(seed)
    t = rng.choice(times, size=(blocks, world, K))            # every rank's K launches of every block
    block = t.sum(axis=2).max(axis=1) + latency_ms             # the block ends when its slowest rank has
    return (K * times.mean()) / block.mean()                   # efficiency against a rank that never waits


def model_ring(times, world, depth, steps=200000, seed=0, latency_ms=0.0):
    """Per-step exchange through a ring of `depth` records (dist.RewardDoneGather(depth=...)): step k of a rank starts when its
    step k-1 has ended AND the exchange of step k-depth is complete (= every rank has ended that step)."""
    rng = np.random.default_rng(seed)
    t = rng.choice(times, size=(steps, world))
    end = np.zeros(world)
    done = np.zeros(steps)                      # when the exchange of step k completed
    for k in range(steps):
        start = end if k < depth else np.maximum(end, done[k - depth])
        end = start + t[k]
        done[k] = end.max() + latency_ms
    return steps * times.mean() / end.max()


def main():
    for path in sys.argv[1:]:
        times = np.loadtxt(path)
        print("%s: %d launches, mean %.4f ms, p50 %.4f, p99 %.4f, max %.4f" % (path, len(times), times.mean(), *np.percentile(times, [50, 99]), times.max()))
        print("  modelled weak-scaling efficiency (whole-job rate / N x the one-GPU rate), rendezvous every K steps:")
        print("    N   " + "".join("K=%-7d" % K for K in (1, 8, 64, 512)))
        for world in (2, 4, 8):
            print("    %-3d " % world + "".join("%-9.3f" % model(times, world, K) for K in (1, 8, 64, 512)))
        print("  per-step exchange through a ring of D records (a rank waits only for the exchange D steps back):")
        print("    N   " + "".join("D=%-7d" % D for D in (1, 2, 4, 8, 16, 64)))
        for world in (2, 4, 8):
            print("    %-3d " % world + "".join("%-9.3f" % model_ring(times, world, D, steps=40000) for D in (1, 2, 4, 8, 16, 64)))


if __name__ == "__main__":
    main()
