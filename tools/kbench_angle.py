import sys, torch
sys.path.insert(0,'.')
from equiadapt_amd import ops, _lib
from equiadapt_amd.images.utils import device_tables
dev=torch.device('cuda:0')
B,S=256,224
x=torch.randn(B,3,S,S,device=dev); go=torch.randn_like(x)
g=torch.randint(0,8,(B,),device=dev,dtype=torch.int32)
th,fl=device_tables("canonicalize",8,False,(2*S,2*S),dev)
def t(fn,reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/reps*1e3
f=lambda: ops.group_action_bwd(x,go,g,th,fl,None,S//2,(S//2,S//2),False,True)
print(f"angle gradient, staged : {t(f):7.1f} us per 256 x 3 x 224^2 (random C8), includes the (B,tiles) partial sum")
_lib.load().eqa_set_option(0,1)
print(f"angle gradient, direct : {t(f):7.1f} us")
_lib.load().eqa_set_option(0,0)
