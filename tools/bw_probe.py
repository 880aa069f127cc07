import torch, time
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
for mb in (14, 57, 228, 912):
    n=mb*1024*1024//4
    a=torch.empty(n,device='cuda'); b=torch.empty(n,device='cuda')
    tf=t(lambda: a.fill_(1.0)); tc=t(lambda: b.copy_(a)); tr=t(lambda: a.sum())
    print(f"{mb} MB: fill {tf:.1f} us = {mb*1.048576/tf*1e3:.0f} GB/s write | copy {tc:.1f} us = {2*mb*1.048576/tc*1e3:.0f} GB/s r+w | sum {tr:.1f} us = {mb*1.048576/tr*1e3:.0f} GB/s read")
