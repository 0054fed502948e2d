import sys, time, torch, json
sys.path.insert(0,'/root/repo')
from oracle import torch_oracle as O
from stereo_toolbox_amd.models import ACVNet
from stereo_toolbox_amd.utils import fill_state_dict, synthetic_tensor
from stereo_toolbox_amd.losses import masked_smooth_l1_multi
from tests.emu_util import emu_product_path
torch.set_num_threads(16)
H,W,D,B = [int(v) for v in sys.argv[1:5]]
LOSS_W=(0.5,0.5,0.7,1.0)
m=ACVNet(D); sd=m.state_dict(); fill_state_dict(sd); m.load_state_dict(sd); m.train()
sd={k:v.clone() for k,v in sd.items()}
left,right=synthetic_tensor((B,3,H,W),1),synthetic_tensor((B,3,H,W),2)
gt=synthetic_tensor((B,H,W),3,lo=0.0,hi=float(D-2))
with torch.no_grad():
    cxf=O.Ctx({k:v.clone() for k,v in sd.items()},True)
    gl,_=O.features_gwc(cxf,left,False); gr,_=O.features_gwc(cxf,right,False)
def run(dtype):
    s_={k:(v.detach().clone().to(dtype).requires_grad_("running" not in k) if v.is_floating_point() else v.clone()) for k,v in sd.items()}
    f_=[t.detach().clone().to(dtype).requires_grad_() for t in (gl,gr)]
    preds=O.acvnet_aggregate(O.Ctx(s_,True),f_[0],f_[1],D,H,W)
    O.smooth_l1_multi(preds,gt.to(dtype),D,LOSS_W).backward()
    return s_,f_
r64,f64=run(torch.float64); r32,f32=run(torch.float32)
t=time.time()
dfe=[t_.clone().requires_grad_() for t_ in (gl,gr)]
with emu_product_path():
    preds=m.aggregate(dfe[0],dfe[1],H,W)
    masked_smooth_l1_multi(preds,gt,D,LOSS_W).backward()
print('emu',time.time()-t)
rows=[]
for k,p in m.named_parameters():
    if k.startswith('feature_extraction'): continue
    g64=r64[k].grad; sc=g64.abs().max().item()
    rows.append(((p.grad.double()-g64).abs().max().item()/sc,(r32[k].grad.double()-g64).abs().max().item()/sc,k))
for i in range(2):
    sc=f64[i].grad.abs().max().item()
    rows.append(((dfe[i].grad.double()-f64[i].grad).abs().max().item()/sc,(f32[i].grad.double()-f64[i].grad).abs().max().item()/sc,'d_feature[%d]'%i))
rows.sort(reverse=True)
print("emu-vs-fp64  oracle32-vs-fp64  name (rel to tensor max)")
for r in rows[:15]: print("%.3e %.3e %s"%r)
for r in rows:
    if 'dres2.conv4.0.0.weight' in r[2]: print("-> %.3e %.3e %s"%r)
