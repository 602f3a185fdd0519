#!/usr/bin/env python3
"""Headline benchmark: Mrays/s at 1500x800, 100 spp, 6 bounces, Scene 1 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (the driver launches N>1 through torch.distributed.run; RCCL = backend "nccl").
A step is one full pass of the hot path over BASELINE's frame: the 100 samples of every pixel are split
over the ranks (sample-slice partition, STRONG scaling: the job is the same 1500x800 x 100-spp frame
at every N; rank r traces split_samples(100, N)[r] samples of every pixel with its own RNG streams),
the un-normalised float accumulation buffers are summed onto rank 0 with one RCCL reduce, and rank 0
divides, colour-grades and tonemaps.  The scene is resident in HBM before the timed region.
(--weak keeps 100 spp per GPU: an N*100-spp frame.)

RNG streams: the reference seeds one serial stream per pixel (gpuRender.cu:130), i.e. 1.2 M
independent chains of 100 samples for this frame -- fewer than four per lane of an MI355X, so the
chip idles through a long tail.  The same sample-slice construct that splits a frame over GPUs is
therefore also used inside a GPU: --sub-streams S (default 100: one stream per sample) gives every pixel S streams of
100/S samples, each a work item of its own (ptx_params.sub_streams; parity against the sum of the
oracle's streams in tests/test_gpu_radiance.py).  The frame with the reference's own seeding
(S = 1, bit-compatible with gpuRender.cu's random sequence) is timed too and reported under
config.reference_seeding.

A ray is one call of Scene::intersect (gpuRender.cu:32): rays per step are counted exactly by an
instrumented launch of the same kernel (paths are deterministic in pixel and seed), outside the
timed region.

roofline (the path kernel; its time is measured live with HIP events on the launch stream):
  frac / achieved   bytes the L2 requested from the memory side per launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE,
                    the gfx950 correction of MI355X_MICROARCH.md; per-ray figure of profiles/traffic.json x this
                    launch's rays) / kernel time, against the 8 TB/s HBM peak.  Infinity-Cache hits are included
                    in the counter and the 98-MB scene lives there, so this is an upper bound of the HBM share.
  l1_level          SURVEY.md section 8d's algorithmic bytes (64*S + 36*T + 144*I + 12*hits + 12*misses + 12*W*H:
                    reference-layout bytes per node step, triangle test, instance entry, face normal, environment
                    texel, output pixel) / kernel time.  These are bytes every step needs at the register file;
                    they are served by L1 / L2 / Infinity Cache re-reads and may exceed the HBM peak.
  memory_path       16-byte lane-accesses (4 per node step and instance entry, 3 per triangle test) against the
                    rate a pure gather of the same shape sustains (tools/ubench_gather2.hip).
  valu              what binds the kernel: VALU wave-instructions per launch (SQ_INSTS_VALU per ray x rays)
                    against 1024 SIMDs x clock / 2.5 cycles per instruction (tools/ubench_valu2.hip), and the
                    share of the 64 lanes that were active in them.

cpu_baseline (rank 0, N=1 only) times the CPU oracle -- this repository's restatement of the
reference's `-d 1` path -- on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd", "py"))

WIDTH, HEIGHT, SPP, BOUNCES, SCENE = 1500, 800, 100, 6, 1
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# What actually bounds the traversal loop: the vector memory path (address processing + L1), whose cost
# is per load instruction and active lane.  A divergent 64-lane load costs ~45 cycles per CU whatever
# its width (tools/ubench_gather2.hip, gpurun_out/ubench_gather3.txt: 180.8 cycles for 4 loads, L1
# resident), i.e. 64 / 45.2 lane-accesses per cycle per CU.
LANE_ACCESS_PEAK_G = 256 * (64 / 45.2) * 2.4  # G lane-accesses/s for 256 CUs at 2.4 GHz
# VALU issue: one plain wave64 instruction per 2.5 cycles and SIMD (profiles/r01/ubench_valu2.txt: 2.4-2.7 for
# fma / mul / add / logic at 4-8 waves per SIMD; min / max / compare / select classes issue slower on their own)
VALU_PEAK_G = 256 * 4 * 2.4 / 2.5  # G wave-instructions/s for 1024 SIMDs at 2.4 GHz
VALU_PEAK_GUIDE_G = 256 * 4 * 2.4 / 2.0  # the same against MI355X_MICROARCH.md's 2 cycles per wave64 VALU instruction


def algorithmic_bytes(st, width, height, hit_bytes=12):
    return (64 * st.node_steps + 36 * st.tri_tests + 144 * st.inst_entries + hit_bytes * st.hits + 12 * st.misses
            + 12 * width * height)


def cpu_baseline(scene, budget_s=None, threads=None):
    """Times the oracle on the host cores: same scene/BVH/resolution/bounces, reduced spp."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_binding as ob
    import ptamd
    if budget_s is None:
        budget_s = float(os.environ.get("PTX_BENCH_CPU_BUDGET", "15"))  # seconds of CPU work (tests shorten it)
    available = os.cpu_count() or 1
    threads = threads or min(available, 16)  # the GPU box's CPU share for one GPU is 16 (of `available` on the machine)
    p = ptamd.make_params(WIDTH, HEIGHT, spp=2, max_bounces=BOUNCES)
    _, st = ob.render(scene, p, seed_mode=ptamd.SEED_REFERENCE_GPU, threads=threads)
    per_spp = st.render_s / 2.0
    spp = int(max(2, min(SPP, budget_s / max(per_spp, 1e-6))))
    if spp > 2:
        p = ptamd.make_params(WIDTH, HEIGHT, spp=spp, max_bounces=BOUNCES)
        _, st = ob.render(scene, p, seed_mode=ptamd.SEED_REFERENCE_GPU, threads=threads)
    return {
        "value": st.rays / st.render_s / 1e6, "unit": "Mrays/s", "cores": threads, "cores_available": available, "kind": "port",
        "sample": f"scene {SCENE} {WIDTH}x{HEIGHT}, {spp} spp of {SPP}, {BOUNCES} bounces, "
                  f"{st.rays} rays in {st.render_s:.2f} s (oracle/pt_oracle.c, pthreads)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-seeding", action="store_true", help="skip the extra frame with the reference's seeding")
    ap.add_argument("--sub-streams", type=int, default=100, help="most RNG streams per pixel and GPU (1 = the reference's seeding)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: 100 spp on every GPU (the job grows with N)")
    ap.add_argument("--spp", type=int, default=SPP, help=argparse.SUPPRESS)  # profiling runs only
    # rehearsal of the N > 1 path on a box with one GPU: every rank on cuda:0, buffers reduced through gloo
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ptamd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def reduce_sum_to_root(t):
        if args.rehearse_on_one_gpu:  # gloo has no device tensors
            h = t.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)  # RCCL over xGMI: W*H*4 floats, once per frame

    def all_reduce(t, op):
        if args.rehearse_on_one_gpu:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    import multigpu
    frame_spp = args.spp * world if args.weak else args.spp            # samples per pixel of the whole job
    spp = multigpu.split_samples(frame_spp, world)[rank]               # this rank's share
    # EVERY rank passes the same sub_streams, derived from the largest share: the backend numbers a rank's streams
    # part_index * sub_streams + k (ptx_backend.h), so ranks that passed different values would overlap (100 spp over 8 GPUs:
    # shares 13,13,13,13,12,12,12,12 -> 7 and 6 streams -> rank 3 would own ids 21..27 and rank 4 ids 24..29).  A rank whose
    # share is smaller than that simply runs fewer streams.
    sub_streams = multigpu.streams_for_frame(frame_spp, world, args.sub_streams)
    # replicated on every GPU; the flattened arrays come from the on-disk cache when an earlier run (or
    # another rank) on this machine has built them (pt_host.h: pth_scene_create_cached)
    import tempfile
    scene = ptamd.HostScene(SCENE, cache_dir=os.path.join(tempfile.gettempdir(), f"ptx_scene_cache_{os.getuid()}"))
    dscene = ptamd.DeviceScene(scene.desc_ptr, device=local_rank)
    params = ptamd.make_params(WIDTH, HEIGHT, spp=spp, max_bounces=BOUNCES, seed_mode=ptamd.SEED_STREAM,
                               part_index=rank, part_count=world, sub_streams=sub_streams, frame_spp=frame_spp)
    accum = torch.zeros((HEIGHT, WIDTH, 4), dtype=torch.float32, device=dev)
    out = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.float32, device=dev)
    out1 = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.float32, device=dev)

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        dscene.accumulate_device(scene.camera, params, accum.data_ptr(), stream)
        if world > 1:
            reduce_sum_to_root(accum)
        if rank == 0:
            dscene.resolve_device(accum.data_ptr(), WIDTH, HEIGHT, out.data_ptr(), stream)

    # exact ray / node / triangle counts of this rank's launch (instrumented kernel, untimed)
    stats = dscene.measure(scene.camera, params) if spp > 0 else ptamd.Stats()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_s = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if spp > 0:
            kernel_s.append(dscene.last_kernel_seconds())  # HIP events on the launch stream
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    rays = torch.tensor([float(stats.rays)], dtype=torch.float64, device=dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        all_reduce(rays, dist.ReduceOp.SUM)
        all_reduce(tmax, dist.ReduceOp.MAX)
    total_rays = rays.item()
    elapsed = tmax.item()

    ref_seed = None
    if rank == 0 and world == 1 and args.sub_streams > 1 and not args.no_reference_seeding:
        # the same frame with the reference's seeding (one stream per pixel, the random sequence of
        # gpuRender.cu:130), timed with the same protocol: warm-up, K steps between synchronisations, kernel
        # time from HIP events
        p1 = ptamd.make_params(WIDTH, HEIGHT, spp=spp, max_bounces=BOUNCES, seed_mode=ptamd.SEED_REFERENCE_GPU)
        st1 = dscene.measure(scene.camera, p1)
        stream = torch.cuda.current_stream().cuda_stream

        def step1():
            dscene.accumulate_device(scene.camera, p1, accum.data_ptr(), stream)
            dscene.resolve_device(accum.data_ptr(), WIDTH, HEIGHT, out1.data_ptr(), stream)

        for _ in range(args.warmup):
            step1()
        torch.cuda.synchronize()
        k1 = []
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step1()
            k1.append(dscene.last_kernel_seconds())
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t1) / args.steps
        ref_seed = {"value": st1.rays / dt1 / 1e6, "unit": "Mrays/s", "ms_per_step": dt1 * 1e3, "kernel_ms": sum(k1) / len(k1) * 1e3,
                    "rays_per_step": st1.rays, "steps": args.steps, "warmup": args.warmup}

    if rank == 0:
        img = out.cpu().numpy()
        if not (img == img).all() or img.min() < 0.0 or img.max() > 1.0 or img.std() == 0.0:
            raise SystemExit("bench.py: rendered frame failed its sanity check")
        avg_kernel = sum(kernel_s) / len(kernel_s)
        alg = algorithmic_bytes(stats, WIDTH, HEIGHT)
        lane_acc = 4 * stats.node_steps + 3 * stats.tri_tests + 4 * stats.inst_entries
        # counters of the committed rocprofv3 --pmc passes (profiles/traffic.json, made by tools/make_traffic.py from
        # profiles/r02): per-ray figures of this kernel on this scene / resolution / bounce count, scaled to this
        # launch's exact ray count (the passes ran the 100-spp frame; a rank of an N-GPU job traces fewer samples)
        traffic = valu = l2_hit = lds_bytes = scratch = None
        kernel_name = "ptx::pathKernel<1, false, true>" if os.environ.get("PTX_QUEUE", "1") == "0" else "ptx::poolKernel<false>"
        counters_from = None   # which committed profile the counter figures come from
        counters_note = "no committed counter profile (profiles/traffic.json)"
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                t = json.load(f)
            same_workload = t.get("workload", "").startswith(f"scene{SCENE}_{WIDTH}x{HEIGHT}_") and t.get("workload", "").endswith(f"_{BOUNCES}b")
            # the counters are per-ray figures of ONE build of ONE kernel with ONE stream count: reported only for that
            if not same_workload:
                counters_note = f"profiles/traffic.json holds {t.get('workload')}, not this workload"
            elif t.get("build_id") != ptamd.build_id():
                counters_note = (f"profiles/traffic.json was measured on build {t.get('build_id')}, the library timed here is "
                                 f"{ptamd.build_id()}: counter figures withheld (tools/profile.sh + tools/make_traffic.py renew them)")
            elif t.get("kernel") != kernel_name or t.get("sub_streams") != sub_streams:
                counters_note = (f"profiles/traffic.json holds {t.get('kernel')} with {t.get('sub_streams')} streams, this run is "
                                 f"{kernel_name} with {sub_streams}: counter figures withheld")
            else:
                counters_from = "profiles/traffic.json (" + t.get("source", "") + ", build " + t["build_id"] + ")"
                counters_note = "per-ray counter figures of the committed rocprofv3 passes x this launch's exact ray count; times are live"
                traffic = t["hbm_bytes_per_ray"] * stats.rays
                l2_hit = t.get("l2_hit_rate")
                lds_bytes = int(t["lds_bytes"]) if t.get("lds_bytes") is not None else None
                scratch = int(t["scratch_size"]) if t.get("scratch_size") is not None else None
                insts = t["valu_insts_per_ray"] * stats.rays
                rate = insts / avg_kernel / 1e9
                valu = {"bound": "VALU issue (wave-instructions)", "insts": insts, "per_ray": t["valu_insts_per_ray"],
                        "achieved": rate, "unit": "G wave-instructions/s",
                        "peak": VALU_PEAK_G, "frac": rate / VALU_PEAK_G,
                        "peak_is": "1024 SIMDs x 2.4 GHz / 2.5 cycles per instruction, measured (profiles/r01/ubench_valu2.txt)",
                        "peak_guide": VALU_PEAK_GUIDE_G, "frac_guide": rate / VALU_PEAK_GUIDE_G,
                        "peak_guide_is": "the same at 2 cycles per wave64 instruction (MI355X_MICROARCH.md)",
                        "lane_utilisation": t.get("valu_lane_utilisation")}
        line = {
            "metric": "Mrays/s at 1500x800, 100spp, 6 bounces, Scene 1",
            "value": total_rays * args.steps / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"scene 1 (SCATTER_STL: 10,000 instances of one {scene.info.triangles_per_mesh}-triangle "
                            f"BLAS, {scene.info.tlas_nodes}-node TLAS) {WIDTH}x{HEIGHT}, {frame_spp} spp per frame"
                            f" ({'+'.join(str(n) for n in multigpu.split_samples(frame_spp, world))} per GPU), {BOUNCES} bounces",
                "geometry": "procedural stand-in bust (models/bust-of-menelaus.stl absent)"
                            if scene.info.stand_in_geometry else "bust-of-menelaus.stl",
                "environment": "analytic stand-in sky (HDR absent)" if scene.info.stand_in_environment else "HDR",
                "partition": "sample slices: every GPU traces its share of each pixel's samples with its own RNG streams, "
                             "RCCL sum-reduce of W*H*4 f32 to rank 0" if world > 1 else "single GPU",
                "rng_streams_per_pixel_per_gpu": sub_streams,
                "reference_seeding": ref_seed,
                "rays_per_step": total_rays, "paths_per_step": WIDTH * HEIGHT * frame_spp,
            },
            "roofline": {
                "bound": "hbm", "achieved": traffic / avg_kernel / 1e9 if traffic else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": traffic / avg_kernel / 1e9 / HBM_PEAK_GBS if traffic else None, "traffic": traffic,
                # frac is COUNTER traffic / time / peak, not SURVEY 8(d)'s algorithmic bytes (those are l1_level below)
                "basis": "counter", "from_committed_profile": counters_from is not None, "counters_from": counters_from,
                "counters_note": counters_note, "build_id": ptamd.build_id(),
                "traffic_is": "bytes requested by the L2 from the memory side (FETCH_SIZE x 2 + WRITE_SIZE), Infinity-Cache hits included",
                "binding": "valu",
                "l2_hit_rate": l2_hit, "lds_bytes_per_block": lds_bytes, "scratch_bytes_per_lane": scratch,
                "kernel": kernel_name, "kernel_ms": avg_kernel * 1e3,
                "l1_level": {"algorithmic_bytes": alg, "achieved": alg / avg_kernel / 1e9, "unit": "GB/s",
                             "note": "bytes the steps consume at the register file; re-served from L1 / L2 / Infinity Cache"},
                "node_steps": stats.node_steps, "tri_tests": stats.tri_tests, "inst_entries": stats.inst_entries,
                "valu": valu,
                # the kernel issues 4 16-byte lane-accesses per node step / instance entry and 3 per triangle test
                "memory_path": {
                    "bound": "vector memory path (lane-accesses: load instructions x active lanes)",
                    "achieved": lane_acc / avg_kernel / 1e9, "peak": LANE_ACCESS_PEAK_G, "unit": "G lane-accesses/s",
                    "frac": lane_acc / avg_kernel / 1e9 / LANE_ACCESS_PEAK_G,
                },
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene)
        print(json.dumps(line), flush=True)

    dscene.close()
    scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
