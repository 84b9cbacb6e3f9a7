"""Timing only: the five conv-block launches of one embed iteration on the f16 two-term kernel (B clips of Tp pooled frames).
usage: AWARE_HIP_LIB=variants/lib_X.so python tools/h2_time.py [B] [Tp] [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aware_amd.runtime import _ptr, _stream, check, load_library

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Tp = int(sys.argv[2]) if len(sys.argv) > 2 else 94
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
RP = 32 * ((Tp + 31) // 32)
lib = load_library()
g = torch.Generator().manual_seed(1)
shapes = [(512, 128, 1), (1024, 512, 1), (1024, 1024, 1), (1024, 1024, 2), (512, 1024, 2)]
cases = []
for (N, K, epi) in shapes:
    a = torch.randn(B * RP, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = (torch.randn(N, generator=g) * 0.1).cuda() if epi != 2 else None
    act = torch.randn(B * RP, N, generator=g).cuda() if epi == 2 else None
    rstd = (torch.rand(B, N, generator=g) + 0.5).cuda()
    c = torch.empty(B * RP, N, device="cuda")
    nb = int(lib.aware_gemm_clip_h2_workspace_bytes(B, N, K))
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    cases.append((a, w, bias, act, rstd, c, ws, nb, N, K, epi))


def run(case):
    a, w, bias, act, rstd, c, ws, nb, N, K, epi = case
    check(lib.aware_gemm_clip_h2(_ptr(a), K, _ptr(w), K, _ptr(bias), _ptr(c), N, B, Tp, N, K, epi, _ptr(rstd), _ptr(act), None, None, 0,
                                 None, _ptr(ws), nb, _stream()), "h2")


# per-launch kernel time comes from the profiler; here: wall per call incl. the pack + amax pre-pass (constant across variants)
for r in range(rounds):
    out = []
    for case in cases:
        for _ in range(3):
            run(case)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(case)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 20 * 1e3)
    print("round", r, " ".join(f"{t:7.1f}" for t in out), " sum %.1f" % sum(out), flush=True)
