import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import torch, cnerf_amd
from cnerf_amd import _lib as L, ops
dev = torch.device("cuda:0"); torch.manual_seed(0)
for (cnt, npi, H, K) in [(2, 1000, 256, 256), (1, 37, 64, 32), (3, 5003, 128, 64), (2, 4096, 256, 192), (3, 1048576, 256, 256), (3, 1048576, 256, 32)]:
    G = torch.randn(cnt, npi, H, device=dev); X = torch.randn(cnt, npi, K, device=dev)
    dW = torch.zeros(cnt, H, K, device=dev); cs = torch.zeros(cnt, H, device=dev)
    L.check(L.lib().cnerf_weight_grad(cnt, npi, H, K, L.ptr(G), L.ptr(X), L.ptr(dW), L.ptr(cs), ops._stream()), "wg")
    ref = torch.bmm(G.transpose(1, 2).double(), X.double()); rcs = G.double().sum(1)
    e1 = ((dW.double() - ref).abs().max() / ref.abs().max()).item(); e2 = ((cs.double() - rcs).abs().max() / rcs.abs().max()).item()
    tb = torch.bmm(G.transpose(1, 2), X); eb = ((tb.double() - ref).abs().max() / ref.abs().max()).item()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        dW.zero_(); cs.zero_()
        L.check(L.lib().cnerf_weight_grad(cnt, npi, H, K, L.ptr(G), L.ptr(X), L.ptr(dW), L.ptr(cs), ops._stream()), "wg")
    torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3):
        tb = torch.bmm(G.transpose(1, 2), X); s = G.sum(1)
    torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / 3
    print(f"cnt={cnt} npi={npi} H={H} K={K}: err dW {e1:.1e} (bmm {eb:.1e}) colsum {e2:.1e}  hip {t1*1e3:.2f} ms ({2*cnt*npi*H*K/t1/1e12:.1f} TF/s)  torch bmm+sum {t2*1e3:.2f} ms", flush=True)
