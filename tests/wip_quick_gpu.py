import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_binding as ob, ptamd
print(ptamd.backend_lib().ptx_version().decode(), flush=True)
for preset, (w, h) in ((2, (1500, 800)), (2, (37, 29)), (0, (750, 400)), (1, (750, 400))):
    t = time.time(); s = ptamd.HostScene(preset); print('scene', preset, 'built', round(time.time() - t, 2), 's', s.info.blas_nodes, s.info.tlas_nodes, s.info.blas_depth, s.info.tlas_depth, flush=True)
    d = ptamd.DeviceScene(s.desc_ptr)
    p = ptamd.make_params(w, h)
    ref, st0 = ob.render_counts(s, p)
    got, st = d.render_counts(s.camera, p)
    print(' counts: oracle max', ref.max(), 'sum', ref.sum(), ' hip max', got.max(), 'sum', got.sum(), ' equal:', np.array_equal(ref, got), ' mismatches', int((ref != got).sum()), ' t_cpu', round(st0.render_s, 4), ' t_gpu', st.render_s, flush=True)
    m = d.measure(s.camera, p, heatmap=True)
    print(' measure', m.as_dict(), flush=True)
    print(' oracle ', st0.as_dict(), flush=True)
    # radiance small
    p = ptamd.make_params(128, 96, spp=8, max_bounces=4)
    t = time.time(); ref, st0 = ob.render(s, p, seed_mode=0); tc = time.time() - t
    got, st = d.render(s.camera, p)
    diff = np.abs(ref - got)
    print(' radiance: rmse', float(np.sqrt(np.mean(diff ** 2))), 'max', float(diff.max()), 'frac>1e-3', float((diff.max(axis=2) > 1e-3).mean()), 't_cpu', round(tc, 3), 't_gpu', st.render_s, flush=True)
    m = d.measure(s.camera, p)
    print(' measure', m.as_dict(), flush=True)
    print(' oracle ', st0.as_dict(), flush=True)
    d.close(); s.close()
