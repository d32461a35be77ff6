import torch, sys, os
sys.path.insert(0,'.')
from dsf_amd import ops
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.train_step import synthetic_batch
render=Render('synthetic','nyu',(588.03,587.07,320.,240.),(640,480)).cuda()
mano=render.mano_layer
B=32
p,c,cube=synthetic_batch(B,'cuda',123)
with torch.no_grad():
    v,_=mano.get_mano_vertices(p[:,:3],p[:,3:48],p[:,48:58],p[:,58:62],1/125)
    verts=(v*cube.unsqueeze(1)/2+c.unsqueeze(1)).contiguous()
    c2,M,_,_=ops.crop_setup(c,cube,render.cam,128)
    minv=torch.linalg.inv_ex(M)[0].contiguous(); cz=c2[:,2].contiguous(); cbz=cube[:,2].contiguous()
    run=lambda: ops.RenderCropFunction.apply(verts,mano.faces_i32,minv,render.resize_rowmap,cz,cbz,render.cam,640,128)
    for dbg in ('0','1','2','3'):
        os.environ['DSF_DBG']=dbg
        for _ in range(5): run()
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        img,p2f=run()
        extra=''
        if dbg=='3': extra=f' total hits per sample {( (p2f+1).clamp(min=0).view(B,-1)[:, ::64].sum(1).float().mean().item()):.0f} (tile-lane0 sums)'
        print(f'dbg {dbg}: {e0.elapsed_time(e1)/50*1e3:.1f} us{extra}')
    # empty kernel launch overhead reference
    os.environ['DSF_DBG']='0'
