import sys, os
sys.path.insert(0,'nubomedia-vca_amd'); sys.path.insert(0,'oracle')
import torch
print('torch', torch.__version__, torch.cuda.is_available())
x = torch.zeros(10, device='cuda'); torch.cuda.synchronize()
from nubovca import capi, synth
ctx = capi.Context(0)
print([l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l][:4])
import numpy as np, orc
img = np.random.default_rng(0).integers(0,256,size=(480,640,3),dtype=np.uint8)
print('gray ok', np.array_equal(ctx.bgr2gray(img), orc.bgr2gray(img)))
t = torch.from_numpy(img).cuda(); torch.cuda.synchronize()
out = np.empty((480,640),np.uint8)
import ctypes as C
o = torch.empty((480,640),dtype=torch.uint8,device='cuda')
rc = ctx.L.nvca_bgr2gray(ctx.h, t.data_ptr(), 640,480,640*3,3, capi.MEM_DEVICE, o.data_ptr(), 640)
print('rc', rc, np.array_equal(o.cpu().numpy(), orc.bgr2gray(img)))
