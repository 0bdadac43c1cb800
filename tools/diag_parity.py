import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from cusift_amd import capi, synth
from cusift_amd.capi import SIFT_POINT_DTYPE, DeviceBuffer
from oracle_binding import Oracle
from parity_utils import canonical_order, ang_diff
o=Oracle(); img=synth.fixture_image()
ctx=capi.Context(0)
for fb in (8,0):
  for thr in (0.1, 1.0):
    kw=dict(num_octaves=6, init_blur=0.0, peak_thresh=thr, edge_thresh=10.0, max_pts=16384, tex_frac_bits=fb)
    want=o.extract(img, **kw)
    prm=capi.default_params(**kw)
    d=DeviceBuffer(ctx, prm.max_pts*588); h=np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
    n=ctx.extract_host(img, prm, d.ptr, h); got=h[:n]
    a,b=canonical_order(want),canonical_order(got)
    dor=ang_diff(a["orientation"].astype(np.float64), b["orientation"].astype(np.float64))
    l2=np.linalg.norm(a["data"].astype(np.float64)-b["data"].astype(np.float64),axis=1)
    ok=dor<1e-3
    print(f"frac_bits={fb} thr={thr} n={n}: ori<1e-3 {ok.mean():.4f} <1e-2 {(dor<1e-2).mean():.4f} max {np.nanmax(dor):.3f} | desc(all) <1e-4 {(l2<1e-4).mean():.4f} | desc(same ori) <1e-4 {(l2[ok]<1e-4).mean():.4f} <1e-3 {(l2[ok]<1e-3).mean():.4f} <1e-2 {(l2[ok]<1e-2).mean():.4f} max {l2[ok].max():.4f}")
