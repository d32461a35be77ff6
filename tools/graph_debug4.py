import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.hourglass import PoseNetMANO
from dsf_amd.train_step import MeshLossStep, GraphedStep, synthetic_batch, Config
from dsf_amd import _lib as L
L.set_deterministic(True)
render = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).cuda()
def make():
    torch.manual_seed(0)
    return MeshLossStep(PoseNetMANO(1, 21).cuda(), render, Config, n_points=512)
e, o = make(), make()
p, c, cube = synthetic_batch(4, "cuda", seed=2)
t = e.make_targets(p, c, cube)
g = GraphedStep(o, t, warmup=2)
for _ in range(2): print("eager warm", {k: round(float(v), 5) for k, v in e(t)[1].items()})
for i in range(4):
    le, te = e(t); lg, tg = g(t)
    print(i, float(le), float(lg), {k: (round(float(te[k]), 5), round(float(tg[k]), 5)) for k in te})
print("---- eager only, fresh model, own Render")
render2 = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).cuda()
torch.manual_seed(0)
e2 = MeshLossStep(PoseNetMANO(1, 21).cuda(), render2, Config, n_points=512)
print([round(float(e2(t)[0]), 6) for _ in range(6)])
print("---- graph only, fresh model, own Render")
render3 = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).cuda()
torch.manual_seed(0)
o3 = MeshLossStep(PoseNetMANO(1, 21).cuda(), render3, Config, n_points=512)
g3 = GraphedStep(o3, t, warmup=2)
print([round(float(g3(t)[0]), 6) for _ in range(4)])
