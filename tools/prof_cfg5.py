"""rocprofv3 driver for BASELINE config 5 (1024x1024 D4, masks + boxes): a few steady-state steps at B=32."""
import os, sys, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import equiadapt_amd as ea
dev = torch.device("cuda:0")
torch.manual_seed(2)
net5 = ea.ConvNetwork((3, 128, 128), out_channels=16, kernel_size=7, num_layers=3, out_vector_size=128)
hp5 = types.SimpleNamespace(beta=1.0, input_crop_ratio=1.0, resize_shape=128, group_type="roto-reflection", num_rotations=4,
                            artifact_err_wt=0.0, learn_ref_vec=False)
can5 = ea.OptimizedGroupEquivariantImageCanonicalization(net5, hp5, (3, 1024, 1024)).to(dev).eval()
B = int(os.environ.get("B", "32"))
x = torch.randn(B, 3, 1024, 1024, device=dev); pred = torch.randn(B, 1, 1024, 1024, device=dev)
masks = [(torch.rand(3, 1024, 1024, device=dev) > 0.5).to(torch.uint8) for _ in range(B)]
boxes = [torch.tensor([[10.0, 20.0, 200.0, 300.0]] * 3, device=dev) for _ in range(B)]
with torch.no_grad():
    for _ in range(int(os.environ.get("REPS", "8"))):
        targets = [{"boxes": b.clone(), "masks": m} for b, m in zip(boxes, masks)]
        y, t = can5(x, targets)
        o = can5.invert_canonicalization(pred, induced_rep_type="scalar")
torch.cuda.synchronize()
