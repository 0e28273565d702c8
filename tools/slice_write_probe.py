import torch, time, sys
sys.path.insert(0, ".")
from mri_epilepsy_diagnosis_amd import ops
CL = torch.channels_last_3d
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(2, 16, 160, 192, 160, device="cuda").to(dt).contiguous(memory_format=CL)
    g, b = torch.ones(16, device="cuda"), torch.zeros(16, device="cuda")
    m, v = torch.zeros(16, device="cuda"), torch.ones(16, device="cuda")
    buf = torch.empty(2, 48, 160, 192, 160, device="cuda", dtype=dt).contiguous(memory_format=CL)
    with torch.no_grad():
        d = t(lambda: ops.norm_act(x, g, b, None, m, v, "running", 0.1, 1e-5, "relu"))
        s0 = t(lambda: ops.norm_act(x, g, b, None, m, v, "running", 0.1, 1e-5, "relu", out=(buf, 0)))
        s1 = t(lambda: ops.norm_act(x, g, b, None, m, v, "running", 0.1, 1e-5, "relu", out=(buf, 16)))
    by = 2 * x.numel() * x.element_size()
    print("%s norm_act fwd c16 full-res: dense %.3f ms (%.2f TB/s)  into 48-ch slice @0 %.3f ms  @16 %.3f ms" % (dt, d, by / d / 1e9, s0, s1))
