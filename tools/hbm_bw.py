import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
for mb in (256, 1024, 4096):
    x=torch.empty(mb*1024*1024//4, device='cuda', dtype=torch.float32).normal_()
    y=torch.empty_like(x)
    ms=t(lambda: y.copy_(x)); print("copy %5d MB: %.3f ms  %.2f TB/s (r+w)" % (mb, ms, 2*mb/1024/1024/ms*1e3*1.048576))
    ms=t(lambda: x.sum()); print("sum  %5d MB: %.3f ms  %.2f TB/s (read)" % (mb, ms, mb/1024/1024/ms*1e3*1.048576))
    ms=t(lambda: y.fill_(1.0)); print("fill %5d MB: %.3f ms  %.2f TB/s (write)" % (mb, ms, mb/1024/1024/ms*1e3*1.048576))
