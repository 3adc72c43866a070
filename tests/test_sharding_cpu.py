"""N > 1 path on CPU: two gloo ranks shard a batch of filters by contiguous index ranges with no
data-path collective, run their shards (here with the oracle standing in for the device, as the
checker), and rank 0 gathers reports and combines the per-rank RMSE sums exactly as bench.py /
the cfg-5 flow does.  The result must equal the single-process run over the whole batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from quadrotor_landing_amd.sharding import combine_rmse, shard_range  # noqa: E402


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make_problem(B, T, seed=5):
    import oracle
    from util import meas_near, rand_imu, rand_states
    rng = np.random.default_rng(seed)
    po = oracle.make_params(update_freq=400.0, direct_orien_method=1)
    x, P = rand_states(rng, B, 15, cov_scale=0.2)
    U = np.stack([rand_imu(rng, B) for _ in range(T)])
    Z = np.stack([meas_near(rng, po, x, ang=0.2, pos=0.05) for _ in range(T)])
    M = np.zeros((T, B), np.uint8); M[3::4] = 1
    truth = np.concatenate([x[:, 0:3], x[:, 6:10]], axis=1)
    return po, x, P, U, Z, M, truth


def _rmse_sums(x, truth):
    from util import qconj, qmul
    er = ((x[:, 0:3] - truth[:, 0:3]) ** 2).sum()
    dq = qmul(qconj(truth[:, 3:7]), x[:, 6:10])
    dq[dq[:, 3] < 0] *= -1
    ang = 2 * np.arctan2(np.linalg.norm(dq[:, :3], axis=1), dq[:, 3])
    return np.array([er, (ang ** 2).sum(), float(x.shape[0])])


def _worker(rank, world, port, B, T, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    po, x, P, U, Z, M, truth = _make_problem(B, T)
    lo, hi = shard_range(B, rank, world)
    xs, Ps = oracle.run_batch(po, x[lo:hi], P[lo:hi], U[:, lo:hi], Z[:, lo:hi], M[:, lo:hi])
    sums = torch.from_numpy(_rmse_sums(xs, truth[lo:hi]))
    # host gather for reporting only; no collective on the data path
    gathered = [None] * world
    dist.gather_object((lo, hi, xs, sums.numpy()), gathered if rank == 0 else None, dst=0)
    dist.barrier()
    if rank == 0:
        xall = np.zeros_like(x)
        parts = []
        for lo_k, hi_k, xk, sk in gathered:
            xall[lo_k:hi_k] = xk
            parts.append(sk)
        np.savez(out_path, x=xall, rmse=np.array(combine_rmse(parts)))
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    for B in (1, 7, 64, 65536, 1048576 + 3):
        for w in (1, 2, 3, 8):
            r = [shard_range(B, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_matches_single_process(tmp_path):
    import oracle
    B, T, world = 96, 12, 2
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(world, _free_port(), B, T, out), nprocs=world, join=True)
    got = np.load(out)
    po, x, P, U, Z, M, truth = _make_problem(B, T)
    xr, _ = oracle.run_batch(po, x, P, U, Z, M)
    np.testing.assert_array_equal(got["x"], xr)  # sharding changes nothing, bit for bit
    s = _rmse_sums(xr, truth)
    np.testing.assert_allclose(got["rmse"], [np.sqrt(s[0] / s[2]), np.sqrt(s[1] / s[2]), s[2]], rtol=1e-12)
