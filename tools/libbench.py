import torch, sys
sys.path.insert(0, '/root/repo/tools'); sys.path.insert(0,'/root/repo')
import kbench
for name, M, N, K, epi in kbench.GEMMS:
    A = torch.randn(M, K, device='cuda').half(); W = (torch.randn(N, K, device='cuda') * K ** -0.5).half()
    out = torch.empty(M, N, device='cuda', dtype=torch.float16)
    fn = lambda: torch.mm(A, W.t(), out=out)
    med, mn = kbench.timeit(fn, 5)
    print(f"torch.mm {name:28s} {2.0*M*N*K/1e9:8.1f} GF | {med:7.1f}us {2.0*M*N*K/med/1e6:6.0f}TF", flush=True)
# conv via torch (MIOpen) channels_last f16
import torch.nn.functional as F
for name, n, H, Wd, cin, cout in kbench.CONVS:
    x = torch.randn(n, cin, H, Wd, device='cuda').half().to(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device='cuda') * (9*cin) ** -0.5).half().to(memory_format=torch.channels_last)
    fn = lambda: F.conv2d(x, w, padding=1)
    med, mn = kbench.timeit(fn, 5)
    fl = 2.0*n*H*Wd*cin*cout*9
    print(f"torch.conv {name:26s} {fl/1e9:8.1f} GF | {med:7.1f}us {fl/med/1e6:6.0f}TF", flush=True)
