import torch, time
x=torch.empty(64*256*256*32, device='cuda')
y=torch.empty_like(x)
for fn,name,bytes_ in ((lambda: x.fill_(1.0),'fill',x.numel()*4),(lambda: y.copy_(x),'copy',2*x.numel()*4)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/20
    print(name, round(ms,4),'ms', round(bytes_/ms/1e9,2),'TB/s')
