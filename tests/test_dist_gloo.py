"""The N>1 path on CPU: world_size-2 `gloo` process groups running dist.combine_means on
(a) synthetic per-iteration images and (b) the oracle rendering rank-striped iterations of the
dragon scene -- the same striping / pre-advanced blue noise / weighted all-reduce the GPU ranks use
(volumetric-path-tracer_amd/dist.py, bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _iteration_image(it, n):
    return np.random.default_rng(1000 + it).random((n, 3), dtype=np.float32)


def _local_running_mean(rank, world, total_iters, n, nan_at=None):
    """what a rank's tail leaves in its accum buffer: the running mean over its LOCAL iteration index (render_kernel.cu:2278-2287), with
    the NaN guard (:2263) substituting the rank's own running mean for a non-finite sample.  nan_at = (iteration, pixel): that sample is NaN."""
    mean = np.zeros((n, 3), np.float32)
    k = 0
    for it in range(rank, total_iters, world):
        v = _iteration_image(it, n)
        if nan_at is not None and it == nan_at[0]:
            v = v.copy()
            v[nan_at[1]] = np.nan
        bad = ~np.isfinite(v).all(axis=1)
        v[bad] = mean[bad]
        mean = v.copy() if k == 0 else mean + (v - mean) / np.float32(k + 1)
        k += 1
    return mean, k


def _worker_synthetic(rank, world, port, total_iters, out_dir, nan_at):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    first, stride, pre = pkg.dist.stripe(rank, world)
    assert (first, stride, pre) == (rank, world, rank)
    mean, k = _local_running_mean(rank, world, total_iters, 257, nan_at)
    assert k == len(range(first, total_iters, stride))    # a rank beyond the job's iterations (world 8, 5 iterations) renders none: weight 0
    acc = torch.from_numpy(mean.copy())
    pkg.dist.combine_means(acc, k)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), acc.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total_iters", [(2, 8), (2, 5), (3, 7), (3, 64), (8, 64), (8, 13), (8, 5)])
def test_combine_means_striped_ranks(tmp_path, world, total_iters):
    """iteration striping with remainders (spp not a multiple of the ranks; more ranks than iterations): every rank ends with the job's mean"""
    port = _free_port()
    mp.spawn(_worker_synthetic, args=(world, port, total_iters, str(tmp_path), None), nprocs=world, join=True)
    ref = np.mean([_iteration_image(it, 257).astype(np.float64) for it in range(total_iters)], axis=0)
    imgs = [np.load(tmp_path / ("r%d.npy" % r)) for r in range(world)]
    for b in imgs[1:]:
        np.testing.assert_array_equal(imgs[0], b)          # every rank holds the same image
    np.testing.assert_allclose(imgs[0], ref, rtol=4e-6, atol=4e-7)


def test_nan_sample_on_one_rank_pins_the_documented_deviation(tmp_path):
    """DESIGN 5, deviation: a non-finite sample is replaced by the running mean of the RANK that rendered it (the reference, single-GPU,
    substitutes the frame's running mean, render_kernel.cu:2263).  Pinned here: the job's image stays finite, equals the weighted sum of the
    ranks' guarded means exactly as specified, and differs from the single-process result in the one affected pixel by no more than that
    sample's weight in the mean."""
    world, total, n, nan_at = 3, 10, 257, (4, 100)          # iteration 4 belongs to rank 1
    port = _free_port()
    mp.spawn(_worker_synthetic, args=(world, port, total, str(tmp_path), nan_at), nprocs=world, join=True)
    got = np.load(tmp_path / "r0.npy")
    assert np.isfinite(got).all()
    parts = [_local_running_mean(r, world, total, n, nan_at) for r in range(world)]
    spec = sum(m.astype(np.float64) * k for m, k in parts) / float(sum(k for _, k in parts))
    np.testing.assert_allclose(got, spec, rtol=4e-6, atol=4e-7)
    single, _ = _local_running_mean(0, 1, total, n, nan_at)   # one process: the guard substitutes the FRAME's running mean
    diff = np.abs(got.astype(np.float64) - single.astype(np.float64))
    others = np.ones(n, bool); others[nan_at[1]] = False
    assert diff[others].max() <= 4e-6                       # every other pixel: the plain mean either way
    assert 0 < diff[nan_at[1]].max() <= 1.0 / total         # samples are in [0, 1): one substituted sample moves the mean by < 1 / N


def _worker_oracle(rank, world, port, spp, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    import oracle_binding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    sd = pkg.scene.dragon_scene(48, 32, "sun")
    ob = oracle_binding.OracleBinding(sd)
    first, stride, pre = pkg.dist.stripe(rank, world)
    phi = np.float32((1.0 + np.sqrt(np.float32(5.0))) / np.float32(2.0))
    for _ in range(pre):                                   # blue noise pre-advanced `rank` steps (:2320-2325)
        live = min(sd.width * sd.height, 65536)            # a launch advances only the entries its pixels own
        ob.blue_noise[:live, :] = np.fmod(ob.blue_noise[:live] + phi, np.float32(1.0))
    n_local = len(range(first, spp, stride))
    ob.render(n_local, iter_stride=stride, iteration=first, nthreads=1)
    acc = torch.from_numpy(ob.accum.copy())
    pkg.dist.combine_means(acc, n_local)
    np.save(os.path.join(out_dir, "o%d.npy" % rank), acc.numpy())
    dist.destroy_process_group()


def test_striped_oracle_render_equals_single_process(tmp_path, orc):
    """2 ranks x striped iterations + one all-reduce == 1 process rendering every iteration."""
    spp = 6
    port = _free_port()
    mp.spawn(_worker_oracle, args=(2, port, spp, str(tmp_path)), nprocs=2, join=True)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    import oracle_binding
    sd = pkg.scene.dragon_scene(48, 32, "sun")
    ob = oracle_binding.OracleBinding(sd)
    ob.render(spp, nthreads=1)
    a = np.load(tmp_path / "o0.npy"); b = np.load(tmp_path / "o1.npy")
    np.testing.assert_array_equal(a, b)
    assert ob.accum.max() > 0
    np.testing.assert_allclose(a, ob.accum, rtol=1e-5, atol=1e-7)
