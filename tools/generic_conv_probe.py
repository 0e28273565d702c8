import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import ops
dev = torch.device("cuda")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (ci, co, k, p) in [(8, 1, (3,1,1), (1,0,0)), (8, 2, (3,1,1), (1,0,0)), (8, 8, (3,1,1), (1,0,0)), (8, 1, (1,1,3), (0,0,1)), (1, 1, (1,3,1), (0,1,0)), (1,1,(3,3,3),(1,1,1))]:
    x = torch.randn(4, ci, 160, 192, 160, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(co, ci, *k, device=dev) * 0.1
    b = torch.randn(co, device=dev)
    g = ops._conv_geom(x.shape, w.shape, (1,1,1), p, (1,1,1))
    ms = t(lambda: ops._conv_fwd(g, x, w, b))
    print("fwd %d->%d k%s: %.3f ms  (%.0f GB/s in+out)" % (ci, co, k, ms, (x.numel()+x.numel()//ci*co)*4/ms/1e6), flush=True)
