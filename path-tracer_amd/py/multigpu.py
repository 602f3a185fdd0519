"""Frame partitioning across ranks (one process per GPU, torch.distributed: "nccl" = RCCL on ROCm).

The reference is single-GPU; this is the split BASELINE.json asks for:

* radiance -- SAMPLE-SLICE partition: the scene is replicated, rank r path-traces the whole frame
  with RNG stream r (PTX_SEED_STREAM, part_index = r) for its share of the samples, the
  un-normalised float accumulation buffers (W*H*4 f32: r, g, b, sample count) are summed onto
  rank 0 with ONE reduce per frame, and rank 0 divides by the summed sample count, colour-grades and
  tonemaps.  Rank 0 of a 1-rank job reproduces the reference GPU seeding exactly.
* heat map (`-a`) -- TILE partition: rank r traces the 8x8 tiles with tile % world == r and writes
  0 elsewhere, so an integer sum-reduce reassembles the frame exactly (disjoint support).

The render calls go through a `backend` object so that the same orchestration runs on GPUs
(HipBackend: C ABI + RCCL) and, in the CPU tests, on the oracle with the gloo backend.
"""
import numpy as np


def split_samples(total_spp, world):
    """Samples per rank for a fixed-size job: the first (total % world) ranks take one extra."""
    base, extra = divmod(total_spp, world)
    return [base + (1 if r < extra else 0) for r in range(world)]


def streams_for_frame(frame_spp, world, most=100):
    """ptx_params.sub_streams for a frame whose samples are split over `world` ranks: ONE value for all ranks, from the
    largest share -- one stream per sample (work items of one path), at most `most`.  With the pool kernel an item costs
    nothing to start, and the shorter the items the more often a wave's lanes set out together through adjacent pixels:
    100 spp took 313 / 304 / 297 / 293 / 289 ms in 8 / 20 / 34 / 67 / 100 streams (profiles/r03/pool_kernel_ab.txt; pathKernel
    in round 2 was best at items of three samples).  The backend numbers a rank's streams part_index * sub_streams + k, so a
    value that differed between ranks would make their stream ids overlap; a rank whose share is smaller than the value runs
    fewer streams."""
    largest = split_samples(frame_spp, world)[0]
    return max(1, min(most, largest))


def stream_ids(frame_spp, world, most=100):
    """The RNG stream ids every rank traces (what the backend does with streams_for_frame's value): for the tests."""
    s = streams_for_frame(frame_spp, world, most)
    return [[r * s + k for k in range(min(s, share))] for r, share in enumerate(split_samples(frame_spp, world))]


class HipBackend:
    """Device buffers are torch CUDA tensors; kernels are launched through the C ABI on torch's
    current stream so that they are ordered with the RCCL collective."""

    def __init__(self, host_scene, device_index=0):
        import torch
        import ptamd
        self.torch = torch
        self.ptamd = ptamd
        self.host = host_scene
        self.device = torch.device("cuda", device_index)
        self.dscene = ptamd.DeviceScene(host_scene.desc_ptr, device=device_index)
        self._accum = {}

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def accumulate(self, params):
        key = (params.height, params.width)
        if key not in self._accum:
            self._accum[key] = self.torch.empty((params.height, params.width, 4), dtype=self.torch.float32,
                                                device=self.device)
        buf = self._accum[key]
        self.dscene.accumulate_device(self.host.camera, params, buf.data_ptr(), self._stream())
        return buf

    def resolve(self, accum):
        h, w = accum.shape[:2]
        out = self.torch.empty((h, w, 3), dtype=self.torch.float32, device=self.device)
        self.dscene.resolve_device(accum.data_ptr(), w, h, out.data_ptr(), self._stream())
        return out

    def counts(self, params):
        buf = self.torch.empty((params.height, params.width), dtype=self.torch.int32, device=self.device)
        self.dscene.counts_device(self.host.camera, params, buf.data_ptr(), self._stream())
        return buf

    def close(self):
        self.dscene.close()


def render_radiance(backend, make_params, width, height, frame_spp, max_bounces, rank, world, dist=None, sub_streams=0):
    """One frame of `frame_spp` samples per pixel, split over the ranks (strong scaling: the job does not
    grow with the number of GPUs).  Returns the tonemapped (H, W, 3) tensor on rank 0, None elsewhere.

    Every rank tells the backend the FRAME's sample count (ptx_params.frame_spp): whether camera rays are
    jittered is a property of the frame (gpuRender.cu:134), not of a rank's share, and a rank whose share is
    empty (more ranks than samples) still contributes an all-zero buffer to the reduce."""
    import ptamd
    seed_mode = ptamd.SEED_REFERENCE_GPU if world == 1 and sub_streams <= 1 else ptamd.SEED_STREAM
    params = make_params(width, height, spp=split_samples(frame_spp, world)[rank], max_bounces=max_bounces,
                         seed_mode=seed_mode, part_index=rank, part_count=world, sub_streams=sub_streams,
                         frame_spp=frame_spp)
    accum = backend.accumulate(params)
    if world > 1:
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        return backend.resolve(accum)
    return None


def render_counts(backend, make_params, width, height, rank, world, dist=None):
    """`-a` frame: exact integer counts on rank 0 (None elsewhere)."""
    params = make_params(width, height, part_index=rank, part_count=world)
    counts = backend.counts(params)
    if world > 1:
        dist.reduce(counts, dst=0, op=dist.ReduceOp.SUM)
    return counts if rank == 0 else None


def to_numpy(t):
    return np.asarray(t.detach().cpu().numpy()) if hasattr(t, "detach") else np.asarray(t)
