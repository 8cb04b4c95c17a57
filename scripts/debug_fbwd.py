#!/usr/bin/env python3
"""Stage check: cnerf_field_backward (explicit points) vs autograd through the oracle's lookup + MLP."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import scaled_err
from oracle import render_oracle as O
import cnerf_amd
from cnerf_amd import ops, _lib as L
from cnerf_amd.generators import ImplicitGenerator3d
dev = torch.device("cuda:0")
def run(variant, B, n_side, S, V, H, seed=0, dtype=torch.float32):
    torch.manual_seed(seed)
    Z = 32; R = n_side; n = R * R * S
    gen = ImplicitGenerator3d(variant, Z, 32, 4, H); net = gen.siren
    fvol = (torch.randn(B, 32, V, V, V) * 0.5); glob = torch.randn(B, Z)
    pts = (torch.rand(B, n, 3) - 0.5) * 1.3
    up = torch.randn(B, n, 4)
    spec = O.FIELD_SPECS[variant]
    def ref(dt):
        params = {k: v.detach().to(dt).requires_grad_(True) for k, v in net.state_dict().items()}
        fv = fvol.to(dt).requires_grad_(True); gl = glob.to(dt).requires_grad_(True)
        feats = O.trilinear_lookup(fv, pts.to(dt))
        out = O.field_mlp(spec, params, feats, gl)
        loss = (out * up.to(dt)).sum()
        g = torch.autograd.grad(loss, [fv, gl] + list(params.values()))
        return out.detach(), g, list(params.keys())
    out32, g32, keys = ref(torch.float32)
    out64, g64, _ = ref(torch.float64)
    # HIP
    net.to(dev)
    fr, ph = net.film(glob.to(dev)); fr, ph = fr.detach(), ph.detach()
    fcl = ops.channel_last(fvol.to(dev))
    cfg = ops.make_cfg(net, B, [fcl], R, S, 30.0, 0.1, 1.0)
    packed = ops.pack_field(net, cfg); packed_t = ops.pack_field_transposed(net, cfg)
    out = ops.field_forward(net, fcl, fr, ph, pts.to(dev), fvol_is_channel_last=True)
    nl = len(spec.layers)
    N = B * n
    a_feat = torch.empty(N, 32, device=dev); a_h = torch.empty(nl, N, H, device=dev); a_c = torch.empty(nl, N, H, device=dev)
    a_g = torch.empty(nl, N, H, device=dev); a_go = torch.empty(N, 4, device=dev); gfv = torch.zeros_like(fcl)
    ptd = pts.to(dev).contiguous(); upd = up.to(dev).contiguous()
    vs, gvs = ops.volumes_struct([fcl]), ops.volumes_struct([gfv])
    L.check(L.lib().cnerf_field_backward(C.byref(cfg), 2, 0, B, C.byref(vs), L.ptr(packed), L.ptr(packed_t), L.ptr(fr), L.ptr(ph),
            L.ptr(torch.eye(4, device=dev).repeat(B, 1, 1).contiguous()), L.ptr(ptd), None, L.ptr(upd), L.ptr(out), L.ptr(a_feat), L.ptr(a_h), L.ptr(a_c),
            L.ptr(a_g), L.ptr(a_go), C.byref(gvs), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "fbwd")
    torch.cuda.synchronize()
    gfv_cf = ops.channel_first(gfv).cpu()
    print(f"{variant} B={B} n={n} V={V} H={H}: fwd err {scaled_err(out.cpu().numpy(), out32.numpy()):.2e} | fvol grad: hip-vs-ref32 {scaled_err(gfv_cf.numpy(), g32[0].numpy()):.2e}  ref32-vs-ref64 {scaled_err(g32[0].numpy(), g64[0].numpy()):.2e}  hip-vs-ref64 {scaled_err(gfv_cf.numpy(), g64[0].numpy()):.2e}")
    # dW of last hidden layer via act buffers
    X = a_h[nl - 2].view(B, n, H) if nl > 1 else a_feat.view(B, n, 32)
    G = a_g[nl - 1].view(B, n, H)
    dWarg = torch.bmm(G.transpose(1, 2), X)
    f = fr[:, (nl - 1) * H: nl * H] if spec.has_global else torch.ones(B, H, device=dev)
    dW = (f.unsqueeze(-1) * dWarg).sum(0).cpu()
    k = f"network.{nl-1}.layer.weight"; i = keys.index(k) + 2
    print(f"     dW[{nl-1}]: hip-vs-ref32 {scaled_err(dW.numpy(), g32[i].numpy()):.2e}  ref32-vs-ref64 {scaled_err(g32[i].numpy(), g64[i].numpy()):.2e}", flush=True)
run("DOUBLESIREN_FG", 1, 16, 12, 12, 64)
run("SHORTSIREN_FG", 1, 16, 12, 12, 64)
run("SHORTSIREN_FG", 2, 16, 12, 12, 64)
run("SHORTSIREN_FG", 2, 16, 12, 12, 64, seed=3)
run("TALLSIREN_FG", 1, 16, 12, 12, 64)
run("SHORTSIREN_FG", 1, 16, 12, 12, 256)
