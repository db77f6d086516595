import torch, sys
sys.path.insert(0,'.')
from equiadapt_amd import ops
from equiadapt_amd.images.utils import device_tables
dev=torch.device('cuda:0')
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps
for B in (1,2,4,8,12,32):
    x=torch.randn(B,3,1024,1024,device=dev)
    g=torch.randint(0,8,(B,),device=dev,dtype=torch.int32)
    th,fl=device_tables("canonicalize",4,True,(2048,2048),dev)
    ms=timeit(lambda: ops.canon_transform(x,g,th,fl,512))
    print(f"cfg5 canon_transform B={B:3d}: {ms*1e3:8.1f} us  {2*x.numel()*4/ms/1e6:8.1f} GB/s  frac {2*x.numel()*4/ms/1e6/8000:.3f}")
for B in (1,4,12,100,256):
    x=torch.randn(B,3,224,224,device=dev)
    g=torch.randint(0,8,(B,),device=dev,dtype=torch.int32)
    th,fl=device_tables("canonicalize",8,False,(448,448),dev)
    ms=timeit(lambda: ops.canon_transform(x,g,th,fl,112))
    print(f"224 canon_transform B={B:3d}: {ms*1e3:8.1f} us  {2*x.numel()*4/ms/1e6:8.1f} GB/s")
