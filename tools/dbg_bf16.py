import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden'); sys.path.insert(0,'/root/repo/tests')
import torch, recipe as R
import graph_physics_amd as gp
from graph_physics_amd import ops
from graph_physics_amd.dense import dense, rms_norm
dev = torch.device('cuda:0')
M,K,N = 500,64,192
x = R.randn((M,K),1); W = R.randn((N,K),2)*0.1; b = R.randn((N,),3)*0.1; W2 = R.randn((N,K),4)*0.1; b2=R.randn((N,),5)*0.1; sc = 1+0.1*R.randn((K,),6)
def ref(mixed):
    h = sc * (x / (x.norm(dim=1, keepdim=True)/K**0.5 + 1e-8))
    if mixed:
        l = torch.nn.functional.gelu(torch.nn.functional.linear(h.bfloat16(), W.bfloat16(), b.bfloat16()))
        r = torch.nn.functional.linear(h.bfloat16(), W2.bfloat16(), b2.bfloat16())
        return (l*r).float()
    return torch.nn.functional.gelu(torch.nn.functional.linear(h, W, b)) * torch.nn.functional.linear(h, W2, b2)
for mode in ('fp32','bf16'):
    ops.set_matrix_precision(mode)
    y = dense(x.to(dev), W.to(dev), b.to(dev), W2=W2.to(dev), b2=b2.to(dev), norm_scale=sc.to(dev), act='gelu').cpu()
    ops.set_matrix_precision('fp32')
    for mixed in (False, True):
        r = ref(mixed)
        print(mode, 'vs mixed' if mixed else 'vs fp32', float((y-r).abs().max()/r.abs().max()))
y = dense(x.to(dev), W.to(dev), b.to(dev)).cpu()
ops.set_matrix_precision('bf16')
y16 = dense(x.to(dev), W.to(dev), b.to(dev)).cpu()
ops.set_matrix_precision('fp32')
r16 = torch.nn.functional.linear(x.bfloat16(), W.bfloat16(), b.bfloat16()).float()
print('plain bf16', float((y16-r16).abs().max()/r16.abs().max()), 'fp32', float((y-torch.nn.functional.linear(x,W,b)).abs().max()))
r32 = torch.nn.functional.linear(x,W,b)
print('y16 vs y32', float((y16-y).abs().max()/y.abs().max()), 'r16 vs r32', float((r16-r32).abs().max()/r32.abs().max()))
xe, We, be = x.bfloat16().float(), W.bfloat16().float(), b.bfloat16().float()
re = torch.nn.functional.linear(xe, We, be).bfloat16().float()
print('y16 vs emulated', float((y16-re).abs().max()/re.abs().max()), 'r16 vs emulated', float((r16-re).abs().max()/re.abs().max()))
def emu(norm, gate, act):
    h = sc * (x / (x.norm(dim=1, keepdim=True)/K**0.5 + 1e-8)) if norm else x
    f = {None: (lambda t: t), 'gelu': torch.nn.functional.gelu, 'relu': torch.relu}[act]
    l = f(torch.nn.functional.linear(h.bfloat16(), W.bfloat16(), b.bfloat16()))
    if gate:
        l = l * torch.nn.functional.linear(h.bfloat16(), W2.bfloat16(), b2.bfloat16())
    return l.float()
for norm in (False, True):
    for gate in (False, True):
        for act in (None, 'relu', 'gelu'):
            ops.set_matrix_precision('bf16')
            y = dense(x.to(dev), W.to(dev), b.to(dev), W2=W2.to(dev) if gate else None, b2=b2.to(dev) if gate else None, norm_scale=sc.to(dev) if norm else None, act=act).cpu()
            ops.set_matrix_precision('fp32')
            r = emu(norm, gate, act)
            print(norm, gate, act, float((y-r).abs().max()/r.abs().max()))
