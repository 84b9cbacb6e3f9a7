"""Micro-benchmark of the fp32 MFMA GEMM at the detector's shapes (development aid)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aware_amd._lib import load_library, check
lib = load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
VARIANTS = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else list(range(1, 13))
NP, NF = B * 94, B * 188
shapes = [("mel_f", NF, 128, 256), ("L0_f", NP, 512, 128), ("L1_f", NP, 1024, 512), ("L2", NP, 1024, 1024),
          ("L3_f", NP, 40, 1024), ("L3_b", NP, 1024, 40), ("L1_b", NP, 512, 1024), ("L0_b", NP, 128, 512),
          ("mel_b", NF, 256, 128)]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
tot = {0: 0.0}
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda"); c = torch.empty(M, N, device="cuda")
    ref = None
    line = f"{name:6s} M={M:6d} N={N:5d} K={K:5d} GF={2*M*N*K/1e9:6.2f} |"
    best = (1e9, -1)
    for v in VARIANTS:
        def run():
            check(lib.aware_gemm_nt_variant(C.c_void_p(a.data_ptr()), K, C.c_void_p(b.data_ptr()), K, None, C.c_void_p(c.data_ptr()), N, M, N, K, v, st))
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        if us < best[0]: best = (us, v)
        line += f" v{v}:{us:6.1f}"
    tot[0] += best[0] * (2 if name == "L2" else 1)
    line += f" | best v{best[1]} {best[0]:.1f}us {2*M*N*K/best[0]/1e6:5.1f}TF |"
    tr = torch.empty(M, N, device="cuda")
    for _ in range(3): torch.mm(a, b.T, out=tr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.mm(a, b.T, out=tr)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    line += f" torch {us:6.1f}us {2*M*N*K/us/1e6:5.1f}TF"
    print(line)
print(f"sum of best variant over the 10 GEMMs of one iteration: {tot[0]:.1f} us")
