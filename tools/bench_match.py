import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, time
from cusift_amd import capi
from cusift_amd.capi import DeviceBuffer, SIFT_POINT_DTYPE
ctx=capi.Context(0)
rng=np.random.default_rng(0)
for n in (1024, 2048, 4096, 8192, 16384, 32768):
    p=np.zeros(n,dtype=SIFT_POINT_DTYPE); d=np.abs(rng.normal(size=(n,128))).astype(np.float32); p["data"]=d/np.linalg.norm(d,axis=1,keepdims=True)
    d1=DeviceBuffer.from_numpy(ctx,p); d2=DeviceBuffer.from_numpy(ctx,p[::-1].copy())
    for _ in range(3): ctx.match(d1.ptr,n,d2.ptr,n,1)
    ctx.synchronize(); t=time.perf_counter()
    for _ in range(10): ctx.match(d1.ptr,n,d2.ptr,n,1)
    ctx.synchronize(); dt=(time.perf_counter()-t)/10
    print("n=%d: %.3f ms, %.1f TFLOP/s (2*n*n*128 flop), %.1f G pairs/s"%(n, dt*1e3, 2*n*n*128/dt/1e12, n*n/dt/1e9))
