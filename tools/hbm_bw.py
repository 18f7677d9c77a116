import torch,time
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e-3
for mb in (16.4, 64, 256, 1024):
    n=int(mb*1e6/4)
    x=torch.randn(n,device='cuda'); y=torch.empty_like(x); z=torch.empty_like(x)
    tc=t(lambda: y.copy_(x))
    tr=t(lambda: x.sum())
    tw=t(lambda: y.fill_(1.0))
    ta=t(lambda: torch.add(x,y,out=z))
    print('%7.1f MB: copy %.2f TB/s (%.1f us)  read %.2f TB/s  write %.2f TB/s  add(2r1w) %.2f TB/s'%(mb, 2*n*4/tc/1e12, tc*1e6, n*4/tr/1e12, n*4/tw/1e12, 3*n*4/ta/1e12))
