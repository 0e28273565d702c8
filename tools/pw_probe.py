import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from mri_epilepsy_diagnosis_amd import ops
dev="cuda"
for (ci,co,d,h,w,k) in [(64,64,20,24,20,1),(128,64,20,24,20,1),(32,16,80,96,80,1),(64,64,20,24,20,3)]:
    x=torch.randn(1,ci,d,h,w,device=dev).contiguous(memory_format=torch.channels_last_3d)
    wt=torch.randn(co,ci,k,k,k,device=dev)*0.05
    for _ in range(3): y=ops.conv3d(x,wt,None,1,k//2,1)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    t0=time.perf_counter(); e0.record()
    for _ in range(20): y=ops.conv3d(x,wt,None,1,k//2,1)
    e1.record(); torch.cuda.synchronize(); t1=time.perf_counter()
    print("conv %dx%dx%d %d->%d @%dx%dx%d: device %.3f ms/call, host wall %.3f ms/call" % (k,k,k,ci,co,d,h,w,e0.elapsed_time(e1)/20,(t1-t0)*1e3/20), flush=True)
