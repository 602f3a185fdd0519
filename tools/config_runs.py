#!/usr/bin/env python3
"""Times BASELINE.json's other configurations once (GPU box): scene 0 at 1500x800/100spp/6 bounces and
scene 1 at 3840x2160 / 8 bounces (spp given on the command line; the configuration asks for 1024 over 8 GPUs,
i.e. 128 per GPU).  Prints one JSON line per run."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "path-tracer_amd", "py"))
import torch  # before the backend (tests/conftest.py)
torch.cuda.is_available()
import ptamd
spp4k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
import multigpu
cases = [(0, 1500, 800, 100, 6, multigpu.streams_for_frame(100, 1)), (0, 1500, 800, 100, 6, 1), (2, 1500, 800, 100, 6, multigpu.streams_for_frame(100, 1)),
         (1, 3840, 2160, spp4k, 8, multigpu.streams_for_frame(8 * spp4k, 8)),      # one GPU's share of BASELINE's last configuration
         (1, 1500, 800, 13, 6, multigpu.streams_for_frame(100, 8)), (1, 1500, 800, 25, 6, multigpu.streams_for_frame(100, 4)),
         (1, 1500, 800, 50, 6, multigpu.streams_for_frame(100, 2))]                 # a rank's share of the bench frame over 8 / 4 / 2 GPUs
for scene_id, w, h, spp, b, streams in cases:
    s = ptamd.HostScene(scene_id)
    d = ptamd.DeviceScene(s.desc_ptr)
    p = ptamd.make_params(w, h, spp=spp, max_bounces=b, seed_mode=ptamd.SEED_STREAM if streams > 1 else ptamd.SEED_REFERENCE_GPU,
                          sub_streams=streams, frame_spp=100 if spp in (13, 25, 50) else 0, part_count=8 if spp == 13 else (4 if spp == 25 else (2 if spp == 50 else 1)))
    acc = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    st = d.measure(s.camera, p)
    d.accumulate_device(s.camera, p, acc.data_ptr(), None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d.accumulate_device(s.camera, p, acc.data_ptr(), None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"scene": scene_id, "size": [w, h], "spp": spp, "bounces": b, "sub_streams": streams, "rays": st.rays,
                      "frame_ms": dt * 1e3, "Mrays_per_s": st.rays / dt / 1e6, "node_steps": st.node_steps, "tri_tests": st.tri_tests,
                      "inst_entries": st.inst_entries, "mean_w": float(acc[..., 3].mean())}), flush=True)
    d.close(); s.close()
