import sys, torch
sys.path.insert(0, ".")
from rcdms_amd import hip
from tools.kbench import timeit
for name, M, N, K, epi in [("L2 CxC", 2560, 1280, 1280, 5), ("L2 ff-out", 2560, 1280, 5120, 5), ("L2 qkv", 2560, 3840, 1280, 0), ("L3 CxC", 640, 1280, 1280, 5), ("L1 CxC", 10240, 640, 640, 5), ("L1 ffout", 10240, 640, 2560, 5), ("L3 ff-out", 640, 1280, 5120, 5), ("L3 qkv", 640, 3840, 1280, 0)]:
    A = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").half(); out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    line = name
    for sk in (0, 1, 2, 3, 4, 6, 8):
        d = hip.GemmDesc(M, N, K, K, N, N, epi, 1, 0, 1.0, sk)
        ws = torch.zeros(max(hip.gemm_workspace_bytes(d), 16), dtype=torch.uint8, device="cuda")
        med, _ = timeit(lambda: hip.gemm(d, A.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, res.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel()))
        line += f" | sk{sk}: {med:6.1f}us"
    print(line)
for name, n, H, Wd, cin, cout in [("L2 conv 1280", 10, 16, 16, 1280, 1280), ("L2 conv 2560->1280", 10, 16, 16, 2560, 1280), ("L3 conv 1280", 10, 8, 8, 1280, 1280), ("L3 conv 2560", 10, 8, 8, 2560, 1280), ("L1 conv 640", 10, 32, 32, 640, 640), ("L1 conv 1920->640", 10, 32, 32, 1920, 640), ("L2 conv 1920->1280", 10, 16, 16, 1920, 1280)]:
    x = torch.randn(n * H * Wd, cin, device="cuda").half(); w = (torch.randn(cout, 9 * cin, device="cuda") * (9 * cin) ** -0.5).half()
    bias = torch.randn(cout, device="cuda"); out = torch.empty(n * H * Wd, cout, device="cuda", dtype=torch.float16)
    for v in (-1, 1, 2):
        hip.set_igemm_variant(v)
        line = f"{name} v{v}"
        for sk in (0, 1, 2, 3, 4, 6, 8, 12):
            d = hip.ConvDesc(n, H, Wd, cin, cout, 1, 0, cin, cout, 0, 1, 1, 0, 1.0, sk)
            ws = torch.zeros(max(hip.conv3x3_workspace_bytes(d), 16), dtype=torch.uint8, device="cuda")
            med, _ = timeit(lambda: hip.conv3x3(d, x.data_ptr(), w.data_ptr(), bias.data_ptr(), 0, 0, out.data_ptr(), ws.data_ptr(), ws.numel()))
            line += f" | sk{sk}: {med:6.1f}"
        print(line)
hip.set_igemm_variant(-1)
