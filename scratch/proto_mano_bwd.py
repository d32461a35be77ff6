# prototype of the hand-derived MANO backward (chain + rodrigues) vs autograd of the oracle
import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from oracle import hand_ref as H
torch.manual_seed(0)
def rod_fwd(th):
    t=th+1e-8; a=np.sqrt((t*t).sum()); n=th/a; h=a*0.5
    q=np.concatenate([[np.cos(h)], np.sin(h)*n]); qn_=np.sqrt((q*q).sum()); qn=q/qn_
    w,x,y,z=qn
    R=np.array([[w*w+x*x-y*y-z*z, 2*x*y-2*w*z, 2*w*y+2*x*z],[2*w*z+2*x*y, w*w-x*x+y*y-z*z, 2*y*z-2*w*x],[2*x*z-2*w*y, 2*w*x+2*y*z, w*w-x*x-y*y+z*z]])
    return R,(t,a,n,h,q,qn_,qn)
def quat_bwd(qn,G):
    w,x,y,z=qn
    gw=2*w*(G[0,0]+G[1,1]+G[2,2])+2*(-z*G[0,1]+y*G[0,2]+z*G[1,0]-x*G[1,2]-y*G[2,0]+x*G[2,1])
    gx=2*x*(G[0,0]-G[1,1]-G[2,2])+2*(y*G[0,1]+z*G[0,2]+y*G[1,0]-w*G[1,2]+z*G[2,0]+w*G[2,1])
    gy=2*y*(-G[0,0]+G[1,1]-G[2,2])+2*(x*G[0,1]+w*G[0,2]+x*G[1,0]+z*G[1,2]-w*G[2,0]+z*G[2,1])
    gz=2*z*(-G[0,0]-G[1,1]+G[2,2])+2*(-w*G[0,1]+x*G[0,2]+w*G[1,0]+y*G[1,2]+x*G[2,0]+y*G[2,1])
    return np.array([gw,gx,gy,gz])
def rod_bwd(th,G):
    R,(t,a,n,h,q,qn_,qn)=rod_fwd(th)
    gqn=quat_bwd(qn,G)
    gq=(gqn-qn*(qn@gqn))/qn_
    gn=np.sin(h)*gq[1:]; gs=n@gq[1:]
    gh=-np.sin(h)*gq[0]+np.cos(h)*gs
    ga=gh*0.5-(gn@th)/(a*a)
    return gn/a+ga*(t/a)
th=np.random.randn(3); G=np.random.randn(3,3)
tt=torch.tensor(th,dtype=torch.float64,requires_grad=True)
Rt=H.rodrigues(tt.view(1,3))
g,=torch.autograd.grad((Rt[0]*torch.tensor(G)).sum(),tt)
print('rod', np.abs(rod_bwd(th,G)-g.numpy()).max())
# chain
parents=[-1,0,1,2,0,4,5,0,7,8,0,10,11,0,13,14]
Rs=torch.randn(1,16,3,3,dtype=torch.float64,requires_grad=True); Js=torch.randn(1,16,3,dtype=torch.float64,requires_grad=True)
import oracle.hand_ref as hr
# make oracle chain float64-friendly
def chain64(Rs,Js):
    B=1; G=[None]*16
    bottom=torch.tensor([0.,0,0,1],dtype=torch.float64).view(1,1,4)
    rigid=lambda R,t: torch.cat([torch.cat([R,t.unsqueeze(-1)],2),bottom],1)
    G[0]=rigid(Rs[:,0],Js[:,0])
    for i in range(1,16):
        p=parents[i]; G[i]=G[p]@rigid(Rs[:,i],Js[:,i]-Js[:,p])
    G=torch.stack(G,1)
    Jh=torch.cat([Js,torch.zeros(1,16,1,dtype=torch.float64)],2).unsqueeze(-1)
    corr=G@Jh
    A=G-torch.cat([torch.zeros(1,16,4,3,dtype=torch.float64),corr],3)
    return G,A
Gm,A=chain64(Rs,Js)
gA=torch.randn(1,16,3,4,dtype=torch.float64)
gRs_ref,gJs_ref=torch.autograd.grad((A[:,:,:3,:]*gA).sum(),(Rs,Js))
# manual
Gn=Gm.detach().numpy()[0]; Rn=Rs.detach().numpy()[0]; Jn=Js.detach().numpy()[0]; gAn=gA.numpy()[0]
Rg=Gn[:,:3,:3]; 
gRg=np.zeros((16,3,3)); gt=np.zeros((16,3)); gJ=np.zeros((16,3)); gR=np.zeros((16,3,3))
for i in range(16):
    gRg[i]=gAn[i,:,:3]-np.outer(gAn[i,:,3],Jn[i]); gt[i]=gAn[i,:,3]; gJ[i]+=-Rg[i].T@gAn[i,:,3]
for i in range(15,0,-1):
    p=parents[i]
    gR[i]=Rg[p].T@gRg[i]; gRg[p]+=gRg[i]@Rn[i].T
    d=Jn[i]-Jn[p]; gd=Rg[p].T@gt[i]; gRg[p]+=np.outer(gt[i],d); gt[p]+=gt[i]; gJ[i]+=gd; gJ[p]-=gd
gR[0]=gRg[0]; gJ[0]+=gt[0]
print('chain', np.abs(gR-gRs_ref.numpy()[0]).max(), np.abs(gJ-gJs_ref.numpy()[0]).max())
