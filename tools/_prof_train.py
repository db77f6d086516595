import sys, torch
sys.path.insert(0, '.')
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
can = bench.build_canonicalizer(dev).train()
opt = torch.optim.SGD(can.parameters(), lr=1e-3)
B = 256
x = torch.randn(B, 3, 224, 224, device=dev); w = torch.randn(B, 3, 224, 224, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    y = can(x)
    loss = (y * w).mean() + can.get_prior_regularization_loss()
    loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::index", "aten::index_put_", "aten::_index_put_impl_", "aten::index_select") or "index" in e.name.lower() and e.device_time_total > 50:
        print(e.name, e.input_shapes, round(e.device_time_total), [f"{s}" for s in (e.stack or [])[:6]])
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
