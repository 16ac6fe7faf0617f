"""Which kernels the vendor library launches for the GEMMs used as neighbours in tools/diag_conv0_pk.py (run under
`rocprofv3 --kernel-trace`; the tool prints nothing itself -- tools/r04_check.sh vendornames collects the names per size)."""
import torch

for n, dt in ((512, torch.float16), (1024, torch.float16), (2048, torch.float16), (4096, torch.float16), (8192, torch.float16),
              (2048, torch.bfloat16), (2048, torch.float32)):
    a, b = (torch.randn(n, n, device="cuda").to(dt) for _ in range(2))
    c = torch.empty(n, n, device="cuda", dtype=dt)
    for _ in range(3):
        torch.mm(a, b, out=c)
    torch.cuda.synchronize()
