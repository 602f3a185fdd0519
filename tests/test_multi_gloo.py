"""N > 1 orchestration (path-tracer_amd/py/multigpu.py) on CPU: two processes, gloo backend, with the
oracle standing in for the device kernels.  Checks the two partitions of SURVEY.md 8(e):
tile split of the heat map reassembles bit-exactly; sample-slice split sums to the same frame as one
process rendering both streams."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import multigpu
import oracle_binding as ob
import ptamd

W, H, SPP, BOUNCES = 48, 40, 6, 3
FRAME_SPP = 7  # strong scaling: split 4 + 3 over two ranks


class OracleBackend:
    """Same interface as multigpu.HipBackend, computed by the CPU oracle into torch CPU tensors."""

    def __init__(self, scene):
        self.scene = scene

    def accumulate(self, params):
        a, _ = ob.accumulate(self.scene, params, seed_mode=params.seed_mode, threads=2)
        return torch.from_numpy(a)

    def resolve(self, accum):
        return torch.from_numpy(ob.resolve(accum.numpy()))

    def counts(self, params):
        c, _ = ob.render_counts(self.scene, params, threads=2)
        return torch.from_numpy(c.astype(np.int32))


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = ptamd.HostScene(2)
    be = OracleBackend(scene)
    img = multigpu.render_radiance(be, ptamd.make_params, W, H, FRAME_SPP, BOUNCES, rank, world, dist)
    cnt = multigpu.render_counts(be, ptamd.make_params, W, H, rank, world, dist)
    # more ranks than samples: rank 1's share is empty and must still take part in the reduce
    one = multigpu.render_radiance(be, ptamd.make_params, W, H, 1, BOUNCES, rank, world, dist)
    # one sample per rank of a two-sample frame: still jittered
    two = multigpu.render_radiance(be, ptamd.make_params, W, H, 2, BOUNCES, rank, world, dist)
    if rank == 0:
        np.save(os.path.join(outdir, "img.npy"), img.numpy())
        np.save(os.path.join(outdir, "cnt.npy"), cnt.numpy())
        np.save(os.path.join(outdir, "one.npy"), one.numpy())
        np.save(os.path.join(outdir, "two.npy"), two.numpy())
    dist.barrier()
    dist.destroy_process_group()
    scene.close()


def test_two_rank_partition_matches_single_process(tmp_path, scene2):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    img = np.load(tmp_path / "img.npy")
    cnt = np.load(tmp_path / "cnt.npy")

    # heat map: tile partition + integer sum == unpartitioned frame, exactly
    full, _ = ob.render_counts(scene2, ptamd.make_params(W, H))
    assert np.array_equal(cnt.astype(np.uint32), full)

    # radiance: the shares rendered in one process, summed in rank order, resolved once
    def expected(frame_spp):
        acc = np.zeros((H, W, 4), dtype=np.float32)
        for r, n in enumerate(multigpu.split_samples(frame_spp, world)):
            p = ptamd.make_params(W, H, spp=n, max_bounces=BOUNCES, seed_mode=ptamd.SEED_STREAM, part_index=r,
                                  part_count=world, frame_spp=frame_spp)
            a, _ = ob.accumulate(scene2, p, seed_mode=ptamd.SEED_STREAM)
            acc += a
        # the split sums to the frame's sample count in every pixel (channel 3 of the accumulation buffer)
        assert (acc[..., 3] == frame_spp).all()
        return ob.resolve(acc)

    assert np.abs(img - expected(FRAME_SPP)).max() < 1e-6
    assert np.abs(np.load(tmp_path / "two.npy") - expected(2)).max() < 1e-6
    # frame of one sample over two ranks == the single-process frame of one sample (no jitter, rank 0's stream)
    single, _ = ob.render(scene2, ptamd.make_params(W, H, spp=1, max_bounces=BOUNCES), seed_mode=ptamd.SEED_REFERENCE_GPU)
    assert np.abs(np.load(tmp_path / "one.npy") - single).max() < 1e-6
    # a one-sample share of a two-sample frame is jittered: it differs from the un-jittered single sample
    p = ptamd.make_params(W, H, spp=1, max_bounces=BOUNCES, seed_mode=ptamd.SEED_STREAM, part_count=2, frame_spp=2)
    a, _ = ob.accumulate(scene2, p, seed_mode=ptamd.SEED_STREAM)
    q = ptamd.make_params(W, H, spp=1, max_bounces=BOUNCES, seed_mode=ptamd.SEED_STREAM, part_count=2)
    b, _ = ob.accumulate(scene2, q, seed_mode=ptamd.SEED_STREAM)
    assert not np.array_equal(a, b)


def test_stream_zero_is_the_reference_seeding(scene2):
    p0 = ptamd.make_params(W, H, spp=4, max_bounces=3, seed_mode=ptamd.SEED_STREAM, part_index=0, part_count=2)
    a, _ = ob.accumulate(scene2, p0, seed_mode=ptamd.SEED_STREAM)
    p = ptamd.make_params(W, H, spp=4, max_bounces=3)
    b, _ = ob.accumulate(scene2, p, seed_mode=ptamd.SEED_REFERENCE_GPU)
    assert np.array_equal(a, b)
    p1 = ptamd.make_params(W, H, spp=4, max_bounces=3, seed_mode=ptamd.SEED_STREAM, part_index=1, part_count=2)
    c, _ = ob.accumulate(scene2, p1, seed_mode=ptamd.SEED_STREAM)
    assert not np.array_equal(a, c)


def test_split_samples():
    assert multigpu.split_samples(100, 8) == [13, 13, 13, 13, 12, 12, 12, 12]
    assert sum(multigpu.split_samples(1024, 8)) == 1024
    assert multigpu.split_samples(5, 1) == [5]
    assert multigpu.split_samples(3, 8) == [1, 1, 1, 0, 0, 0, 0, 0]  # more ranks than samples: empty shares
    for total in (1, 7, 100, 1024):
        for world in (1, 2, 4, 8):
            assert sum(multigpu.split_samples(total, world)) == total
