"""Which hipBLASLt kernels serve the conv layers' implicit-GEMM shapes (run under rocprofv3 --kernel-trace --stats):
the kernel names carry the macro tile (MT..), the MFMA shape (MI..) and the wave tiling."""
import torch
dev = torch.device('cuda:0')
for m, n, k in [(8192, 8192, 8192), (32 * 64 * 64, 512, 4608), (32 * 128 * 128, 256, 2304)]:
    a = torch.randn(m, k, device=dev, dtype=torch.float16); b = torch.randn(n, k, device=dev, dtype=torch.float16)
    for _ in range(5):
        torch.matmul(a, b.t())
    torch.cuda.synchronize()
