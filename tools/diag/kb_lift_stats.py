import torch, sys
sys.path.insert(0,'/root/repo')
from equiadapt_amd import ops
dev=torch.device('cuda:0')
x=torch.randn(256,3,96,96,device=dev).contiguous(memory_format=torch.channels_last)
bank=torch.randn(256,3,5,5,device=dev)/8.6
wpk=ops.pack_lift_weights(bank)
def t(fn,n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
print("plain  %.1f us"%t(lambda: ops.lift_conv_nhwc(x,wpk,None,False,5,5)))
print("stats  %.1f us (kernel + seam correction)"%t(lambda: ops.lift_conv_nhwc_stats(x,wpk,5,5)))
