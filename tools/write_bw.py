import torch, time
x = torch.empty(90*1024*1024//4, dtype=torch.float32, device="cuda")
for f,name in ((lambda: x.zero_(),"zero_ 90MB"), (lambda: x.fill_(1.0),"fill_ 90MB")):
    f(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/50
    print(name, round(us,1),"us", round(x.numel()*4/us/1e6,2),"TB/s")
y = torch.empty_like(x)
y.copy_(x); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): y.copy_(x)
e1.record(); torch.cuda.synchronize()
us=e0.elapsed_time(e1)*1e3/50
print("copy 90MB", round(us,1),"us", round(2*x.numel()*4/us/1e6,2),"TB/s (read+write)")
