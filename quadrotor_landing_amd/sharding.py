"""Multi-GPU partitioning of a batch of independent filters.

Filters never exchange data, so N devices run N disjoint contiguous index ranges with no collective;
only reports and the three RMSE sums are gathered on the host.
"""
import math


def shard_range(batch, rank, world):
    """[lo, hi) of `batch` filters owned by `rank` of `world` (contiguous, sizes differ by at most 1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(int(batch), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def combine_rmse(per_device_sums):
    """Combine per-device (sum |r_err|^2, sum |theta_err|^2, count) into (rmse_r, rmse_theta, count)."""
    er = sum(float(s[0]) for s in per_device_sums)
    eth = sum(float(s[1]) for s in per_device_sums)
    n = sum(float(s[2]) for s in per_device_sums)
    if n <= 0:
        raise ValueError("no filters")
    return math.sqrt(er / n), math.sqrt(eth / n), n
