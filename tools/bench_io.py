import sys, math, numpy as np, torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from slam_amd import synth
from slam_amd.api import Context, default_params
n=512
ctx=Context(synth.default_intrinsics(640,480), default_params(), max_frames=2, max_pairs=1)
bgr=torch.randint(0,256,(n,480,640,3),dtype=torch.uint8,device='cuda'); gray=torch.empty((n,480,640),dtype=torch.uint8,device='cuda')
dep=torch.randint(0,30000,(n,480,640),dtype=torch.int16,device='cuda'); cl=torch.empty((n,480,640,3),dtype=torch.float32,device='cuda')
torch.cuda.synchronize()
for i in range(3):
    ctx.bgr_to_gray_dev(n,bgr.data_ptr(),gray.data_ptr()); ctx.depth_to_point_cloud_dev(n,dep.data_ptr(),1.2,0.9,cl.data_ptr())
ctx.sync(); ctx.profile_enable(True)
for i in range(10):
    ctx.bgr_to_gray_dev(n,bgr.data_ptr(),gray.data_ptr()); ctx.depth_to_point_cloud_dev(n,dep.data_ptr(),1.2,0.9,cl.data_ptr())
p=ctx.profile_read()
npx=n*640*480
for k,b in (("bgr_to_gray",4*npx),("depth_to_cloud",14*npx)):
    ms=p[k]['total_ms']/p[k]['calls']; print(k, round(ms,4),"ms", round(b/ms/1e6,1),"GB/s")
