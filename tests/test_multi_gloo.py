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


def test_stream_ids_of_all_ranks_are_disjoint():
    """bench.py passes ONE sub_streams value on every rank (multigpu.streams_for_frame, from the largest share): the backend
    numbers a rank's streams part_index * sub_streams + k, so per-rank values (7 for a 13-spp share, 6 for a 12-spp one) would
    make rank 3 (ids 21..27) and rank 4 (24..29) trace the same streams twice."""
    for frame_spp in (1, 3, 7, 12, 13, 34, 100, 101, 1024):
        for world in (1, 2, 3, 4, 8):
            ids = multigpu.stream_ids(frame_spp, world)
            flat = [i for r in ids for i in r]
            assert len(flat) == len(set(flat)), (frame_spp, world, ids)
            s = multigpu.streams_for_frame(frame_spp, world)
            for r, (mine, share) in enumerate(zip(ids, multigpu.split_samples(frame_spp, world))):
                assert len(mine) == min(s, share) and all(r * s <= i < (r + 1) * s for i in mine)
    assert multigpu.streams_for_frame(100, 8) == 13 and multigpu.streams_for_frame(100, 3) == 34 and multigpu.streams_for_frame(100, 1) == 100
    assert multigpu.streams_for_frame(1024, 8) == 100 and multigpu.streams_for_frame(100, 1, most=34) == 34
    # what round 2's bench.py did: a value per rank from its own share
    per_rank = [[r * ((n + 1) // 2) + k for k in range((n + 1) // 2)] for r, n in enumerate(multigpu.split_samples(100, 8))]
    flat = [i for r in per_rank for i in r]
    assert len(flat) != len(set(flat))


W8, H8 = 16, 12


def _worker8(rank, world, port, outdir, cache_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()  # all eight ask for the same scene in the same fresh cache directory at once
    scene = ptamd.HostScene(1, scatter_instances=64, cache_dir=cache_dir)
    be = OracleBackend(scene)
    c4 = multigpu.render_radiance(be, ptamd.make_params, W8, H8, 100, 6, rank, world, dist)    # BASELINE config 4: 13/12 spp
    c5 = multigpu.render_radiance(be, ptamd.make_params, W8, H8, 1024, 8, rank, world, dist)   # config 5: 128 spp each
    cnt = multigpu.render_counts(be, ptamd.make_params, W8, H8, rank, world, dist)
    if rank == 0:
        np.save(os.path.join(outdir, "c4.npy"), c4.numpy())
        np.save(os.path.join(outdir, "c5.npy"), c5.numpy())
        np.save(os.path.join(outdir, "cnt8.npy"), cnt.numpy())
    with open(os.path.join(outdir, f"hit{rank}.txt"), "w") as f:
        f.write("1" if scene.cache_hit else "0")
    dist.barrier()
    dist.destroy_process_group()
    scene.close()


def test_eight_ranks_with_the_baseline_splits_and_one_cache_directory(tmp_path):
    """World 8 on a tiny frame of scene 1: BASELINE config 4's split (100 spp -> 13,13,13,13,12,12,12,12) and config 5's
    (1024 spp -> 128 each, 8 bounces), every rank building or loading the flattened scene in ONE fresh cache directory at the
    same moment (bench.py does that on an 8-GPU node: pth_scene_create_cached must survive the race)."""
    world = 8
    port = 31500 + (os.getpid() % 2000)
    cache = tmp_path / "cache"
    cache.mkdir()
    mp.start_processes(_worker8, args=(world, port, str(tmp_path), str(cache)), nprocs=world, join=True, start_method="spawn")
    scene = ptamd.HostScene(1, scatter_instances=64, cache_dir=str(cache))
    assert scene.cache_hit  # a complete entry was left behind, whoever won the race

    def expected(frame_spp, bounces):
        acc = np.zeros((H8, W8, 4), dtype=np.float32)
        for r, n in enumerate(multigpu.split_samples(frame_spp, world)):
            p = ptamd.make_params(W8, H8, spp=n, max_bounces=bounces, seed_mode=ptamd.SEED_STREAM, part_index=r, part_count=world,
                                  frame_spp=frame_spp)
            a, _ = ob.accumulate(scene, p, seed_mode=ptamd.SEED_STREAM)
            acc += a
        assert (acc[..., 3] == frame_spp).all()
        return ob.resolve(acc)

    assert np.abs(np.load(tmp_path / "c4.npy") - expected(100, 6)).max() < 1e-5
    assert np.abs(np.load(tmp_path / "c5.npy") - expected(1024, 8)).max() < 1e-5
    full, _ = ob.render_counts(scene, ptamd.make_params(W8, H8))
    assert np.array_equal(np.load(tmp_path / "cnt8.npy").astype(np.uint32), full)
    scene.close()
