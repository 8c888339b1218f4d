import ctypes, os, sys, time, subprocess, torch
sys.path.insert(0, "self-corr-pose_amd")
from scp_amd import capi
L = capi.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
M, K, N = 32800, 1536, 384
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
C = torch.empty(M, N, device="cuda"); R = torch.randn(M, N, device="cuda")
def smi(tag):
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    print(tag, [l.strip() for l in out.splitlines() if "sclk" in l or "mclk" in l or "Power" in l or "fclk" in l], flush=True)
smi("idle")
for mode in ("own", "torch"):
    t0 = time.time()
    n = 0
    while time.time() - t0 < 6:
        for _ in range(200):
            if mode == "own":
                L.scp_vit_linear(P(A), P(W), P(b), P(None), P(None), P(R), P(C), M, N, K, 1, capi.current_stream())
            else:
                torch.addmm(R, A, W.t(), out=C)
        n += 200
        if n % 2000 == 0:
            smi(mode + " busy(queue)")
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(mode, "avg ms", dt / n * 1e3, "TF/s", 2 * M * N * K / (dt / n) / 1e12, flush=True)
