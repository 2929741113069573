import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from pointstowood_amd._lib import Epilogue, check, lib, ptr, stream
from tests.test_gpu_ops import _pack_h, _to_h
prec = 0
for M, N, K, flags in [(70000, 512, 512, 0), (32768, 512, 512, 0), (70000, 512, 512, 1), (131072, 512, 512, 0)]:
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / K ** 0.5
    bias, dotw = torch.randn(N, generator=g), torch.randn(N, generator=g)
    dW, wscale, Kp = _pack_h(W, prec)
    ldh_a = (K + 4 + 31) // 32 * 32
    Ah = _to_h(A, prec, ldh_a)
    db, dw = bias.cuda(), dotw.cuda()
    ep = Epilogue(ptr(db), None, None, None, None, None, 0, 1, 0, 0, 0)
    need = int(lib().p2w_gemm_h2_rowdot_ws_bytes(M, N))
    hd = torch.empty((M, N), device="cuda")
    check(lib().p2w_gemm_h2(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), ptr(hd), N, None, 0, flags, stream()))
    ref = (hd.double() @ dw.double() + 0.25)
    for rep in range(3):
        ws = torch.full((need,), 0xFF, dtype=torch.uint8, device="cuda")
        out = torch.full((M,), float("nan"), device="cuda")
        check(lib().p2w_gemm_h2_rowdot(prec, ptr(Ah), ldh_a, ptr(dW), wscale, M, N, K, C.byref(ep), ptr(dw), 0.25, ptr(out), ptr(ws), need, flags, stream()))
        err = (out.double() - ref).abs()
        bad = (err > 1e-3).nonzero().flatten().cpu().numpy()
        print(M, N, K, flags, "rep", rep, "bad rows", len(bad), "max", float(err.max()))
        if len(bad):
            print("  first", bad[:24], " row%256:", sorted(set((bad % 256).tolist()))[:40], " tiles:", sorted(set((bad // 256).tolist()))[:20])
            part = ws.view(torch.float32).view(-1, (M + 255) // 256 * 256)
            pref = torch.stack([(hd[:, 64*s:64*s+64].double() @ dw[64*s:64*s+64].double()) for s in range(N // 64)], 0)
            pe = (part[:, :M].double() - pref).abs()
            bs = (pe > 1e-3).nonzero().cpu().numpy()
            print("  bad (slot,row) count", len(bs), "slots:", sorted(set(bs[:, 0].tolist())), "nan:", int(torch.isnan(part[:, :M]).sum()))
