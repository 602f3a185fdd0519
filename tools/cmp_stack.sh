# Checks the global overflow part of the traversal stack: a build with an 8-entry LDS stack must render the
# same bits as the default build (run on the GPU box after tools/build_variants.sh l8 "-DPTX_LDS_STACK=8").
set -e
cd $GRAFT_REPO_ROOT
cat > /tmp/r.py <<'PY'
import sys, os, numpy as np
sys.path.insert(0, "path-tracer_amd/py")
import ptamd
s = ptamd.HostScene(1); d = ptamd.DeviceScene(s.desc_ptr)
img, _ = d.render(s.camera, ptamd.make_params(300, 160, spp=8, max_bounces=6))
cnt, _ = d.render_counts(s.camera, ptamd.make_params(300, 160))
np.save(sys.argv[1], img); np.save(sys.argv[1] + ".cnt.npy", cnt)
PY
timeout -k 10 100 python3 /tmp/r.py /tmp/a.npy
PTX_BACKEND_LIB=$GRAFT_REPO_ROOT/path-tracer_amd/lib/libptx_backend_l8.so timeout -k 10 100 python3 /tmp/r.py /tmp/b.npy
python3 -c "
import numpy as np
a=np.load('/tmp/a.npy'); b=np.load('/tmp/b.npy'); print('radiance identical:', np.array_equal(a,b), float(np.abs(a-b).max()))
a=np.load('/tmp/a.npy.cnt.npy'); b=np.load('/tmp/b.npy.cnt.npy'); print('counts identical:', np.array_equal(a,b), int(a.max()))"
