import torch, sys, os
sys.path.insert(0,'.')
from dsf_amd import ops
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.train_step import synthetic_batch
render=Render('synthetic','nyu',(588.03,587.07,320.,240.),(640,480)).cuda()
mano=render.mano_layer
for B in (32,128):
    p,c,cube=synthetic_batch(B,'cuda',123)
    with torch.no_grad():
        v,_=mano.get_mano_vertices(p[:,:3],p[:,3:48],p[:,48:58],p[:,58:62],1/125)
        verts=(v*cube.unsqueeze(1)/2+c.unsqueeze(1)).contiguous()
        c2,M,_,_=ops.crop_setup(c,cube,render.cam,128)
        minv=torch.linalg.inv_ex(M)[0].contiguous(); cz=c2[:,2].contiguous(); cbz=cube[:,2].contiguous()
        run=lambda: ops.RenderCropFunction.apply(verts,mano.faces_i32,minv,render.resize_rowmap,cz,cbz,render.cam,640,128)
        for tgt in ('256','512','1024','2048'):
            os.environ['DSF_CROP_WG_TARGET']=tgt
            for _ in range(5): run()
            torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): run()
            e1.record(); torch.cuda.synchronize()
            img,p2f=run()
            print(f'B{B} target {tgt}: {e0.elapsed_time(e1)/50*1e3:.1f} us  fg frac {(p2f>=0).float().mean().item():.3f}')
