import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from equiadapt_amd.images.canonicalization_networks import fftconv
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, Cin, Cout, H, W, relu, in_relu) in [(2, 8, 12, 92, 92, False, False), (3, 16, 8, 92, 92, True, True), (2, 4, 4, 60, 97, True, False), (1, 256, 256, 92, 92, True, True)]:
    x = torch.randn(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 5, 5, device=dev) / (5 * Cin ** 0.5)
    b = torch.randn(Cout, device=dev); ib = torch.randn(Cin, device=dev)
    Bm = fftconv.filter_spectra(w)
    y = fftconv.conv5x5(x, Bm, b, relu, ib if in_relu else None, in_relu)
    xin = torch.relu(x.double() + ib.double()[None, :, None, None]) if in_relu else x.double()
    ref = F.conv2d(xin, w.double(), b.double())
    ref = torch.relu(ref) if relu else ref
    print((B, Cin, Cout, H, W), "max err", (y.double() - ref).abs().max().item(), "scale", ref.abs().max().item())
    S = fftconv.conv5x5(x, Bm, b, relu, ib if in_relu else None, in_relu, sums_k=5)
    OH, OW = H - 4, W - 4
    Sref = torch.stack([torch.stack([ref[:, :, u:u + OH - 4, v:v + OW - 4].sum((-1, -2)) for v in range(5)], -1) for u in range(5)], -2)
    print("   sums rel err", ((S - Sref).abs().max() / Sref.abs().max()).item())
import time
x = torch.randn(256, 256, 92, 92, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(256, 256, 5, 5, device=dev) / 80; b = torch.randn(256, device=dev)
Bm = fftconv.filter_spectra(w)
for _ in range(2): fftconv.conv5x5(x, Bm, b, True, b, True, sums_k=5)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): fftconv.conv5x5(x, Bm, b, True, b, True, sums_k=5)
torch.cuda.synchronize(); print("fft conv + sums, B=256: %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
