import sys, os, torch
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
orig = nn_conv._x6_image
log = []
def spy(weight, wk, mode):
    cache = weight.__dict__.get("_dsf_x6")
    hit = cache is not None and mode in cache and cache[mode][0] == (weight._version, nn_conv._EPOCH, wk.data_ptr()) and weight.__dict__.get("_dsf_managed", False)
    if torch.cuda.is_current_stream_capturing():
        log.append((tuple(weight.shape), mode, hit, None if cache is None or mode not in cache else (cache[mode][0], (weight._version, nn_conv._EPOCH, wk.data_ptr())), weight.__dict__.get("_dsf_managed", False)))
    return orig(weight, wk, mode)
nn_conv._x6_image = spy
g = GraphedStep(o, t, warmup=2)
print("image requests during capture:", len(log), "misses:", sum(not l[2] for l in log))
for l in [l for l in log if not l[2]][:8]: print(l)
