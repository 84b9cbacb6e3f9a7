"""Run one GEMM shape/variant repeatedly (for PMC collection)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aware_amd._lib import load_library, check
lib = load_library()
M, N, K, v, reps = [int(x) for x in sys.argv[1:6]]
a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda"); c = torch.empty(M, N, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(reps):
    check(lib.aware_gemm_nt_variant(C.c_void_p(a.data_ptr()), K, C.c_void_p(b.data_ptr()), K, None, C.c_void_p(c.data_ptr()), N, M, N, K, v, st))
torch.cuda.synchronize()
