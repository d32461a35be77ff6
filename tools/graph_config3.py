"""Config 3 (B = 64 hourglass-2 + mesh losses): eager vs GraphedStep ms per step."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.hourglass import PoseNetMANO
from dsf_amd.train_step import MeshLossStep, GraphedStep, synthetic_batch, Config
render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
torch.manual_seed(0)
step = MeshLossStep(PoseNetMANO(2, 21).cuda(), render, Config)
p, c, cube = synthetic_batch(64, "cuda", seed=9)
tgt = step.make_targets(p, c, cube)
def timeit(f, n=20, w=5):
    for _ in range(w): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
e = timeit(lambda: step(tgt))
g = GraphedStep(step, tgt)
print("config 3 eager %.2f ms/step | graph replay %.2f ms/step | nodes %s" % (e, timeit(lambda: g(tgt)), g.node_types))
