import numpy as np, torch, sys
sys.path.insert(0,'.')
from dsf_amd.render_model.mano_layer import MANO_SMPL
g=np.load('tests/golden/reference_golden.npz')
m=MANO_SMPL('synthetic','nyu').cuda()
P=torch.tensor(g['mano_params']).cuda()
v,j,Rs=m.forward(P[:,48:58],P[:,3:48],P[:,:3],get_skin=True)
print('Rs',np.abs(Rs.cpu().numpy()-g['mano_fwd_Rs']).max())
print('j',np.abs(j.cpu().numpy()-g['mano_fwd_joints']).max())
d=np.abs(v.cpu().numpy()-g['mano_fwd_verts'])
print('v',d.max(), d.max(axis=(1,2)), 'argmax vert', np.unravel_index(d.argmax(), d.shape))
# zero pose/shape
Z=torch.zeros(1,62).cuda()
v0,j0,_=m.forward(Z[:,48:58],Z[:,3:48],Z[:,:3],get_skin=True)
from oracle import hand_ref as H
from dsf_amd.assets import build_synthetic_mano
hm=H.HandModel(build_synthetic_mano(0))
vo,jo,_=H.mano_forward(hm,torch.zeros(1,10),torch.zeros(1,45),torch.zeros(1,3))
print('zero', np.abs(v0.cpu().numpy()-vo.numpy()).max())
# only shape
for name,sl in [('beta',slice(48,58)),('theta',slice(3,48)),('rot',slice(0,3))]:
    Q=torch.zeros(2,62); Q[:,sl]=P[:2,sl].cpu()
    vq,_,_=m.forward(Q[:,48:58].cuda(),Q[:,3:48].cuda(),Q[:,:3].cuda(),get_skin=True)
    voq,_,_=H.mano_forward(hm,Q[:,48:58],Q[:,3:48],Q[:,:3])
    print(name, np.abs(vq.cpu().numpy()-voq.numpy()).max())
