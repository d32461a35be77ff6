import sys, os, torch, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.hourglass import PoseNetMANO
from dsf_amd.train_step import MeshLossStep, GraphedStep, synthetic_batch, Config
from dsf_amd import _lib as L, nn_conv
L.set_deterministic(True)
r = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).cuda()
torch.manual_seed(0)
o = MeshLossStep(PoseNetMANO(1, 21).cuda(), r, Config, n_points=512)
p, c, cube = synthetic_batch(4, "cuda", seed=2)
t = o.make_targets(p, c, cube)
g = GraphedStep(o, t, warmup=2)
torch.cuda.synchronize()
snap = {"static": {k: v.clone() for k, v in g.static.items() if torch.is_tensor(v)},
        "state": {k: v.clone() for k, v in o.net.state_dict().items()},
        "render": {k: v.clone() for k, v in r.state_dict().items()}}
def terms():
    torch.cuda.synchronize(); return {k: round(float(v), 6) for k, v in g.terms.items()}
g.graph.replay(); print("replay 1", terms())
for k, v in g.static.items():
    if torch.is_tensor(v) and not torch.equal(v, snap["static"][k]): print("static input changed:", k)
ch = [k for k, v in o.net.state_dict().items() if not torch.equal(v, snap["state"][k])]
print("net state changed:", len(ch), [k for k in ch if "running" not in k and "num_batches" not in k][:10])
print("render state changed:", [k for k, v in r.state_dict().items() if not torch.equal(v, snap["render"][k])])
g.graph.replay(); print("replay 2", terms())
o.net.load_state_dict(snap["state"]); 
g.graph.replay(); print("replay 3 after restoring the net state (incl. BN statistics)", terms())
o.net.eval(); o.net.train()
