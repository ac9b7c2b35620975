#!/usr/bin/env python3
"""Dev tool (GPU): are the operators HIP-graph safe?  Generator forward, forward + backward through D, with zero_grad(set_to_none) and a
capturable fused Adam step inside the capture -- eager vs replay, bit for bit (they are: DESIGN.md section 6, "HIP graphs")."""
import os, sys, copy
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))
import torch
from gan_control_amd.models.op import _backend, weight_cache
from gan_control_amd.models.gan_model import Generator, Discriminator
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
_backend.get().conv_mode = sys.argv[2] if len(sys.argv) > 2 else 'bf16x3'
torch.manual_seed(0)
G = Generator(size, 512, 8, channel_multiplier=2, conv_transpose=True).cuda()
D = Discriminator(size, channel_multiplier=2).cuda()
z = torch.randn(4, 512, device='cuda')
noise = [n.expand(4, -1, -1, -1).contiguous() for n in G.make_noise(1)]
side = torch.cuda.Stream()
_pool = torch.randn(8 * 3 * 1024 * 1024 // 4, device='cuda', generator=torch.Generator(device='cuda').manual_seed(11))
torch.Tensor.normal_ = lambda self, *a, **k: self.copy_(_pool[:self.numel()].view_as(self))
from gan_control_amd.trainers.utils import requires_grad
opt = torch.optim.Adam(G.parameters(), lr=0.0, betas=(0.0, 0.99), fused=True, capturable=True)
def fb_nonoise():
    for p in G.parameters(): p.grad = None
    img, _ = G([z])
    pred, _ = D(img)
    torch.nn.functional.softplus(-pred).mean().backward()
    return img
def fb_zero():
    G.zero_grad(set_to_none=True)
    requires_grad(D, False)
    img, _ = G([z])
    pred, _ = D(img)
    torch.nn.functional.softplus(-pred).mean().backward()
    return img
def fb_opt():
    G.zero_grad(set_to_none=True)
    img, _ = G([z])
    pred, _ = D(img)
    torch.nn.functional.softplus(-pred).mean().backward()
    opt.step()
    return img
def fwd():
    img, _ = G([z], noise=noise)
    return img
def fb():
    for p in G.parameters(): p.grad = None
    img, _ = G([z], noise=noise)
    pred, _ = D(img)
    torch.nn.functional.softplus(-pred).mean().backward()
    return img
for name, fn in (('fwd', fwd), ('fwd+bwd', fb), ('noise=None', fb_nonoise), ('zero_grad + frozen D', fb_zero), ('+ optimizer (lr 0)', fb_opt)):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            ref = fn()
        ref = ref.detach().clone()
        ref_g = {n: p.grad.clone() for n, p in G.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()
    weight_cache.clear()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = fn()
    out.zero_()
    for p in G.parameters():
        if p.grad is not None: p.grad.zero_()
    g.replay(); torch.cuda.synchronize()
    print(name, 'image max abs diff', float((out.detach() - ref).abs().max()), 'ref max', float(ref.abs().max()))
    if ref_g:
        bad = sorted(((float((p.grad - ref_g[n]).abs().max() / (ref_g[n].abs().max() + 1e-20)), n) for n, p in G.named_parameters() if n in ref_g and p.grad is not None), reverse=True)
        print('   grad worst:', bad[:5])
    weight_cache.clear()
