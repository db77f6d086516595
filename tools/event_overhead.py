import torch, statistics
dev=torch.device('cuda:0')
x=torch.zeros(64,device=dev)
def bracket(fn,n=200):
    ts=[]
    for _ in range(n):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ts.append((a,b))
    torch.cuda.synchronize()
    v=[a.elapsed_time(b)*1e3 for a,b in ts]
    return statistics.median(v), min(v), statistics.mean(v)
print("empty bracket (us): median/min/mean", bracket(lambda: None))
print("tiny kernel (x.add_) bracket:", bracket(lambda: x.add_(1)))
big=torch.zeros(64*1024*1024,device=dev)
print("256 MB fill bracket:", bracket(lambda: big.fill_(1.0),50))
# back-to-back reference for the same fill
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): big.fill_(1.0)
b.record(); torch.cuda.synchronize(); print("256 MB fill back-to-back per launch:", a.elapsed_time(b)/50*1e3)
a.record()
for _ in range(200): x.add_(1)
b.record(); torch.cuda.synchronize(); print("tiny kernel back-to-back per launch:", a.elapsed_time(b)/200*1e3)
