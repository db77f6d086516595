"""Where the HOST time of a BASELINE configs[4] step goes at its own batch (B = 4): wall time per step, then the same step with the
Python profiler, top functions by cumulative time.   python tools/host_time_cfg5.py"""
import cProfile, os, pstats, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import equiadapt_amd as ea
dev = torch.device("cuda:0")
torch.manual_seed(2)
net5 = ea.ConvNetwork((3, 128, 128), out_channels=16, kernel_size=7, num_layers=3, out_vector_size=128)
hp5 = types.SimpleNamespace(beta=1.0, input_crop_ratio=1.0, resize_shape=128, group_type="roto-reflection", num_rotations=4,
                            artifact_err_wt=0.0, learn_ref_vec=False)
can5 = ea.OptimizedGroupEquivariantImageCanonicalization(net5, hp5, (3, 1024, 1024)).to(dev).eval()
B = int(os.environ.get("B", "4"))
x = torch.randn(B, 3, 1024, 1024, device=dev); pred = torch.randn(B, 1, 1024, 1024, device=dev)
masks = [(torch.rand(3, 1024, 1024, device=dev) > 0.5).to(torch.uint8) for _ in range(B)]
boxes = [torch.tensor([[10.0, 20.0, 200.0, 300.0]] * 3, device=dev) for _ in range(B)]
def step():
    targets = [{"boxes": b, "masks": m} for b, m in zip(boxes, masks)]
    y, t = can5(x, targets)
    return y, t, can5.invert_canonicalization(pred, induced_rep_type="scalar")
with torch.no_grad():
    for _ in range(20): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print(f"B = {B}: {dt * 1e3:.3f} ms per step = {B / dt:.0f} img/s (eager)")
    t0 = time.perf_counter()
    for _ in range(200): step()
    host = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    print(f"host-only time per step (no synchronise inside the loop): {host * 1e3:.3f} ms")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): step()
    pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(45)
