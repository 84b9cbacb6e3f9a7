"""Does the power-limited conv GEMM lose less than proportionally when it is confined to a subset of the CUs?
(hipExtStreamCreateWithCUMask through ctypes; the kernels are launched on that stream through the C ABI.)
If time(75 % of the CUs) < time(100 %) / 0.75, spatial sharing with the HBM-bound DSP kernels of a second batch could pay.
Development probe; prints one line per mask."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aware_amd import runtime as rt

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = N = 1024
Tp = 94
M = B * 96
torch.cuda.init()
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn((M, K), device="cuda", generator=g)
w = torch.randn((N, K), device="cuda", generator=g) / 32
packed = rt.x3_pack(w)
x = torch.randn(256 * 1024 * 1024 // 4, device="cuda", generator=g)          # 256 MB stream copy as the HBM-bound stand-in
y = torch.empty_like(x)


def run(word, label):
    words = (ctypes.c_uint32 * 8)(*([word] * 8))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    if rc:
        print(label, "hipExtStreamCreateWithCUMask rc", rc)
        return
    ext = torch.cuda.ExternalStream(s.value)
    with torch.cuda.stream(ext):
        for _ in range(5):
            rt.gemm_clip(a, w, None, B, Tp, 0, mode=1, packed=packed)
        ext.synchronize()
        t0 = time.time()
        for _ in range(30):
            rt.gemm_clip(a, w, None, B, Tp, 0, mode=1, packed=packed)
        ext.synchronize()
        dt = (time.time() - t0) / 30
        for _ in range(3):
            y.copy_(x)
        ext.synchronize()
        t0 = time.time()
        for _ in range(10):
            y.copy_(x)
        ext.synchronize()
        dc = (time.time() - t0) / 10
    print(f"{label}: CUs/XCD {bin(word).count('1')}/32  gemm {dt*1e3:.3f} ms  ({2.0*M*N*K/dt/1e12:.1f} TFLOP/s f32-eq)   copy {2*x.numel()*4/dc/1e12:.2f} TB/s")


for word, label in ((0xFFFFFFFF, "all"), (0x0FFFFFFF, "7/8"), (0x00FFFFFF, "3/4"), (0x0000FFFF, "1/2"), (0x000000FF, "1/4"), (0xFFFFFFFF, "all")):
    run(word, label)


def masked_stream(word):
    words = (ctypes.c_uint32 * 8)(*([word] * 8))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


print("--- both at once: GEMM on one set of CUs, a 256 MB copy loop on the complement (each rate taken from the run in which the other loop outlasts it)")
for wg, wd, label in ((0x00FFFFFF, 0xFF000000, "3/4 + 1/4"), (0x003FFFFF, 0xFFC00000, "11/16 + 5/16"), (0x000FFFFF, 0xFFF00000, "5/8 + 3/8"),
                      (0x0000FFFF, 0xFFFF0000, "1/2 + 1/2"), (0xFFFFFFFF, 0xFFFFFFFF, "all + all (no partition)")):
    sg, sd = masked_stream(wg), masked_stream(wd)
    res = {}
    for which, ngemm, ncopy in (("gemm", 20, 400), ("copy", 120, 20)):
        eg0, eg1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ed0, ed1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        if which == "gemm":
            with torch.cuda.stream(sd):
                ed0.record(sd)
                for _ in range(ncopy):
                    y.copy_(x)
                ed1.record(sd)
            with torch.cuda.stream(sg):
                eg0.record(sg)
                for _ in range(ngemm):
                    rt.gemm_clip(a, w, None, B, Tp, 0, mode=1, packed=packed)
                eg1.record(sg)
        else:
            with torch.cuda.stream(sg):
                eg0.record(sg)
                for _ in range(ngemm):
                    rt.gemm_clip(a, w, None, B, Tp, 0, mode=1, packed=packed)
                eg1.record(sg)
            with torch.cuda.stream(sd):
                ed0.record(sd)
                for _ in range(ncopy):
                    y.copy_(x)
                ed1.record(sd)
        torch.cuda.synchronize()
        tg, td = eg0.elapsed_time(eg1), ed0.elapsed_time(ed1)
        res[which] = (tg / ngemm, td / ncopy, tg, td)
    tg = res["gemm"][0]
    td = res["copy"][1]
    print(f"{label}: gemm {tg:.3f} ms ({2.0*M*N*K/(tg*1e-3)/1e12:.1f} TFLOP/s; loops {res['gemm'][2]:.0f} vs {res['gemm'][3]:.0f} ms)   "
          f"copy {2*x.numel()*4/(td*1e-3)/1e12:.2f} TB/s (loops {res['copy'][3]:.0f} vs {res['copy'][2]:.0f} ms)")
