import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from cusift_amd import capi, synth
from cusift_amd.batch import BatchExtractor
B,w,h=64,1920,1080
ex=BatchExtractor(B,w,h,num_octaves=5,init_blur=1.0,peak_thresh=3.0,max_pts=32768)
imgs=np.stack([synth.tile(1000+i,w,h,1.0) for i in range(B)])
d=ex.images_from_numpy(imgs)
pts,cnt=ex.extract(d); torch.cuda.synchronize()
tot=int(torch.clamp(cnt,max=32768).sum())
out=torch.empty((tot+10,588),dtype=torch.uint8,device='cuda')
outc=torch.empty((tot+10,160),dtype=torch.uint8,device='cuda')
for name,fn,o in (("exact",ex.ctx.pack_points,out),("compact",ex.ctx.pack_points_compact,outc)):
    for _ in range(3): fn(pts.data_ptr(),cnt.data_ptr(),B,32768,o.data_ptr(),tot,None)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): fn(pts.data_ptr(),cnt.data_ptr(),B,32768,o.data_ptr(),tot,None)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
    print(name, tot, "records", round(dt*1e3,4), "ms", round(tot*588*2/dt/1e9,1) if name=="exact" else round(tot*(588+160)/dt/1e9,1), "GB/s")
