import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch.nn.functional as F
from equiadapt_amd import _lib, ops
from equiadapt_amd.images.canonicalization_networks import fftconv
from test_gpu_lift_fft import _spectra_fp64, _unpack_V
dev = torch.device("cuda:0"); lib = _lib.load()
nimg, H0, W0, C = 3, 96, 96, 64
g = torch.Generator().manual_seed(1)
x = torch.randn(nimg, 3, H0, W0, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
bank = (torch.randn(C, 3, 5, 5, generator=g) / 75 ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
bias = torch.randn(C, generator=g).to(dev)
M = nimg * fftconv.tiles(H0 - 4) * fftconv.tiles(W0 - 4)
V = fftconv.spectra_buffer(M, 2 * C, dev); V.fill_(7.0)
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.eqa_lift5_fft48k5_input(x.data_ptr(), bank.data_ptr(), bias.data_ptr(), 1, V.data_ptr(), nimg, H0, W0, C, st), "f")
want = _spectra_fp64(torch.relu(F.conv2d(x.double(), bank.double(), bias.double())))
got = _unpack_V(V, C)
d = (got - want).abs()
print("scale", want.abs().max().item(), "max err", d.max().item())
bad = d > 1e-3
print("bad fraction", bad.float().mean().item())
ky, kx = fftconv.freq_index()
fb = bad.any(dim=2).any(dim=1)            # per frequency
print("bad freq count", fb.sum().item(), "of", fb.numel())
idx = fb.nonzero().flatten()[:40].cpu()
print("first bad (ky,kx):", [(int(ky[i]), int(kx[i])) for i in idx])
cb = bad.any(dim=0).any(dim=0)
print("bad channels", cb.nonzero().flatten().tolist()[:64])
mb = bad.any(dim=0).any(dim=1); print("bad tiles", mb.nonzero().flatten().tolist()[:20])
print("untouched (7.0) fraction", (V == 7.0).float().mean().item())
