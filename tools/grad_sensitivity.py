import sys, time, torch
sys.path.insert(0,'/root/repo')
from oracle import torch_oracle as O
from stereo_toolbox_amd.models import ACVNet
from stereo_toolbox_amd.utils import fill_state_dict, synthetic_tensor
torch.set_num_threads(16)
H,W,D,B = [int(v) for v in sys.argv[1:5]]
eps = float(sys.argv[5]) if len(sys.argv)>5 else 1e-6
LOSS_W=(0.5,0.5,0.7,1.0)
m=ACVNet(D); sd=m.state_dict(); fill_state_dict(sd)
left,right=synthetic_tensor((B,3,H,W),1),synthetic_tensor((B,3,H,W),2)
gt=synthetic_tensor((B,H,W),3,lo=0.0,hi=float(D-2))
with torch.no_grad():
    cxf=O.Ctx({k:v.clone() for k,v in sd.items()},True)
    gl,_=O.features_gwc(cxf,left,False); gr,_=O.features_gwc(cxf,right,False)
def run(dtype, pert=0.0, seed=0):
    s_={k:(v.detach().clone().to(dtype).requires_grad_("running" not in k) if v.is_floating_point() else v.clone()) for k,v in sd.items()}
    f_=[t.detach().clone().to(dtype) for t in (gl,gr)]
    if pert:
        g=torch.Generator().manual_seed(seed)
        f_=[t*(1+pert*torch.randn(t.shape,generator=g,dtype=torch.float64).to(dtype)) for t in f_]
    f_=[t.requires_grad_() for t in f_]
    preds=O.acvnet_aggregate(O.Ctx(s_,True),f_[0],f_[1],D,H,W)
    O.smooth_l1_multi(preds,gt.to(dtype),D,LOSS_W).backward()
    return s_
t=time.time()
r64=run(torch.float64); print('fp64',time.time()-t)
r32=run(torch.float32)
rp=run(torch.float64,eps,1)
rows=[]
for k,v in r64.items():
    if not v.is_floating_point() or v.grad is None or k.startswith('feature_extraction'): continue
    sc=v.grad.abs().max().item()
    e32=(r32[k].grad.double()-v.grad).abs().max().item()/sc
    ep=(rp[k].grad-v.grad).abs().max().item()/sc
    rows.append((ep,e32,k))
rows.sort(reverse=True)
print("top by perturbation sensitivity (rel to tensor max): pert%g  fp32-oracle-err  name"%eps)
for r in rows[:12]: print("%.3e %.3e %s"%r)
for r in rows:
    if 'dres2.conv4.0.0.weight' in r[2] or 'dres2.attention_block' in r[2]: print("-> %.3e %.3e %s"%r)
