import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, cnerf_amd
from cnerf_amd import ops, _lib as L
from test_gpu_parity import to_tb16
dev = torch.device("cuda:0")
for (cnt, npi, H, K) in [(1, 32, 32, 32), (1, 64, 32, 32), (1, 37, 64, 32), (2, 1000, 256, 256), (1, 999, 4, 64)]:
    torch.manual_seed(0)
    G, X = torch.randn(cnt, npi, max(H, 32), device=dev), torch.randn(cnt, npi, K, device=dev)
    if H < 32: G[..., H:] = 0
    g16, tiles = to_tb16(G); x16, _ = to_tb16(X)
    dW, cs = torch.zeros(cnt, H, K, device=dev), torch.zeros(cnt, H, device=dev)
    L.check(L.lib().cnerf_weight_grad16(cnt, tiles, H, g16.shape[1], x16.shape[1], L.ptr(g16), L.ptr(x16), L.ptr(dW), L.ptr(cs), None, ops._stream()), "wg16")
    Gh, Xh = G.half().double()[..., :H], X.half().double()
    ref = torch.bmm(Gh.transpose(1, 2), Xh); rcs = Gh.sum(1)
    print((cnt, npi, H, K), "dW err", ((dW.double() - ref).abs().max() / ref.abs().max()).item(), "cs err", ((cs.double() - rcs).abs().max() / rcs.abs().max()).item())
    if npi <= 64 and H == 32:
        # which transposition? compare against alternatives
        alt = torch.bmm(Xh.transpose(1, 2), Gh)
        print("   vs X^T G:", ((dW.double() - alt).abs().max() / alt.abs().max()).item(), " dW[0,:2,:4]", dW[0, :2, :4].tolist(), " ref", ref[0, :2, :4].tolist())
