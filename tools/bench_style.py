import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from one_to_many_gan_amd import _hip as H
from tools.bench_conv import timeit
dev="cuda"; B=16; WD=6; Ci=Co=256
w=torch.rand(B,WD,device=dev); ws=torch.randn(Ci,WD,device=dev); bs=torch.ones(Ci,device=dev)
q=torch.rand(Co,Ci,device=dev); qt=q.t().contiguous()
s=torch.empty(B,Ci,device=dev); d=torch.empty(B,Co,device=dev)
t=timeit(lambda: H.style_fwd(w,ws,bs,qt,s,d,Ci,0.4,1e-8)); print(f"style_fwd {t*1e6:.1f} us")
sums=torch.randn(B,2,Co,device=dev); dots=torch.randn(B,Ci,device=dev)
e=torch.empty(B,Co,device=dev); gs=torch.empty(B,Ci,device=dev); gw=torch.empty(B,WD,device=dev)
gws=torch.empty(Ci,WD,device=dev); gbs=torch.empty(Ci,device=dev); gq=torch.zeros(Co,Ci,device=dev)
t=timeit(lambda: H.style_bwd(sums,None,dots,s,d,q,w,ws,e,gs,gw,gws,gbs,gq,Ci,0.4)); print(f"style_bwd (3 launches) {t*1e6:.1f} us")
