import sys, time, ctypes as C
sys.path.insert(0, 'nubomedia-vca_amd')
import numpy as np, torch
from nubovca import capi, synth
ctx = capi.Context(0); casc = ctx.load_cascade_xml(synth.synthetic_cascade_xml())
fs = capi.FaceStream(ctx, casc, width_to_process=1920, multi_scale_factor=10)
F=32
base=[(200,150,300),(900,400,180),(1400,100,120),(1500,700,240)]
fr=[synth.make_bgr(1920,1080,i,'natural',[(x+8*i,y,s) for x,y,s in base]) for i in range(F)]
keep=[torch.from_numpy(f).cuda() for f in fr]; torch.cuda.synchronize()
frames=[capi.make_frame(t.data_ptr(),1920,1080,1920*3,capi.MEM_DEVICE) for t in keep]
n=F; cap=64
sh=(C.c_void_p*n)(*[fs.h]*n); fa=(capi.Frame*n)(*frames); out=(capi.Rect*(n*cap))(); ids=(C.c_int*(n*cap))(); cnt=(C.c_int*n)()
for _ in range(3): ctx.L.nvca_face_batch_process(ctx.h,n,sh,fa,out,ids,cap,cnt)
ctx.enable_kernel_timing(True)
t0=time.perf_counter()
for _ in range(10): ctx.L.nvca_face_batch_process(ctx.h,n,sh,fa,out,ids,cap,cnt)
t1=time.perf_counter()
kt=ctx.kernel_timing(); ksum=sum(v[0] for v in kt.values())/10
print('raw C call %.3f ms/step; kernels %.3f ms; host+idle %.3f ms' % ((t1-t0)/10*1e3, ksum, (t1-t0)/10*1e3-ksum))
ctx.enable_kernel_timing(False)
t0=time.perf_counter()
for _ in range(10): ctx.L.nvca_face_batch_process(ctx.h,n,sh,fa,out,ids,cap,cnt)
t1=time.perf_counter()
print('raw C call, kernel timing off %.3f ms/step' % ((t1-t0)/10*1e3))
t0=time.perf_counter()
for _ in range(10): ctx.face_batch_process([fs]*F, frames, cap=64)
t1=time.perf_counter()
print('python wrapper %.3f ms/step' % ((t1-t0)/10*1e3))
