import sys
sys.path[:0]=['/root/repo/danbo-pytorch_amd']
import torch
from core import train_path, hip_ops as ops
DEV="cuda:0"
gen = torch.Generator(device="cpu").manual_seed(17)
M=1487
for K1,K2,N in ((432,0,448),(448,0,448),(432,448,448)):
    lin=torch.nn.Linear(K1+K2,N).to(DEV)
    x1=torch.randn(M,K1,generator=gen).to(DEV)
    x2=torch.randn(M,K2,generator=gen).to(DEV) if K2 else None
    xin = x1 if x2 is None else torch.cat([x1,x2],-1)
    ref=torch.relu(xin.double()@lin.weight.double().t()+lin.bias.double())
    packed,shape=ops.linear16_pack(lin.weight.detach(), K1=K1 if K2 else None)
    y=ops.linear16(x1,packed,shape,lin.bias.detach(),relu=True,x2=x2)
    d=(y.double()-ref).abs()
    print(K1,K2,N,"direct max dev",float(d.max()),"ref max",float(ref.abs().max()), "bad cols", (d.max(0).values>1e-4).nonzero().flatten().tolist()[:10], "bad rows", int((d.max(1).values>1e-4).sum()))
    y2=train_path.linear16(lin,x1.requires_grad_(True),relu=True,x2=x2)
    print("   via Fn", float((y2.double()-ref).abs().max()))
