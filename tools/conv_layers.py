import sys, torch, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd import nn_conv
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
dev='cuda'
torch.manual_seed(0)
net=MANO_OCR_stage('ResNet_stage_18',21,True).to(dev)
render=Render('synthetic','nyu',(588.03,587.07,320.,240.),(640,480)).to(dev)
step=RenderSupervisedStep(net,render,Config)
p,c,cube=synthetic_batch(32,dev,0); tgt=step.make_targets(p,c,cube)
step(tgt); 
nn_conv.RECORD=[]; step(tgt); torch.cuda.synchronize(); recs,nn_conv.RECORD=nn_conv.RECORD,None
agg=collections.OrderedDict()
for r in recs:
    agg.setdefault(r,0); agg[r]+=1
rows=[]
for r,n in agg.items():
    us,fl,_=nn_conv.replay(r,iters=5)
    rows.append((us*n,n,us,fl/us/1e6,r))
rows.sort(reverse=True)
tot=sum(x[0] for x in rows)
print('total conv us/step',tot)
for t,n,us,tf,r in rows[:40]:
    print(f'{t:8.0f}us n={n} each {us:7.1f}us {tf:6.1f}TF {nn_conv.kernel_name(r):28s} {r[0]} B{r[1]} in{r[2]}x{r[3]}x{r[4]} out{r[5]}x{r[6]}x{r[7]} k{r[8]} s{r[10]} d{r[11]}')
