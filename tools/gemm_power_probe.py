"""Is the bf16x3 conv GEMM bound by issue slots or by the clock the chip holds under MFMA load?
Times the same kernel (one 1024->1024 conv block, B clips of 96 rows, PLAIN epilogue) on three operand sets:
random normal, all zeros, and random with the two low split planes empty (values exactly representable in bf16).
Cycles per MFMA do not depend on the data (MI355X_MICROARCH.md, 'DVFS give-back'); wall time does, through the clock.
Development aid; prints one line per operand set."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aware_amd import runtime as rt

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
K = N = 1024
Tp = 94
M = B * 96
g = torch.Generator(device="cuda").manual_seed(0)


def operands(kind):
    if kind == "zeros":
        return torch.zeros((M, K), device="cuda"), torch.zeros((N, K), device="cuda")
    a = torch.randn((M, K), device="cuda", generator=g)
    w = torch.randn((N, K), device="cuda", generator=g) / 32
    if kind == "bf16-exact":
        a, w = a.bfloat16().float(), w.bfloat16().float()
    return a, w


for kind in ("random", "zeros", "bf16-exact", "random"):
    a, w = operands(kind)
    packed = rt.x3_pack(w)
    for _ in range(5):
        rt.gemm_clip(a, w, None, B, Tp, 0, mode=1, packed=packed)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        rt.gemm_clip(a, w, None, B, Tp, 0, mode=1, packed=packed)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    f32_flop = 2.0 * M * N * K
    print(f"{kind:11s} B={B}: {dt*1e3:.3f} ms/launch  {f32_flop/dt/1e12:.1f} TFLOP/s f32-equivalent  "
          f"{6*f32_flop/dt/1e15:.3f} PFLOP/s on the bf16 pipe")
