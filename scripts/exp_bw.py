import torch, time
x = torch.empty(1<<28, dtype=torch.int32, device='cuda')   # 1 GiB
y = torch.empty_like(x)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
gb = x.numel()*4/1e9
print('fill  %.2f TB/s' % (gb/t(lambda: x.fill_(7))/1e3))
print('copy  %.2f TB/s (r+w)' % (2*gb/t(lambda: y.copy_(x))/1e3))
print('read(sum) %.2f TB/s' % (gb/t(lambda: x.sum())/1e3))
