import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import RenderSupervisedStep, synthetic_batch, Config
dev = 'cuda'
torch.manual_seed(0)
net = MANO_OCR_stage('ResNet_stage_18', 21, True).to(dev)
render = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480), inverse=sys.argv[1] if len(sys.argv) > 1 else 'torch').to(dev)
opt = torch.optim.AdamW(net.parameters(), lr=Config.lr, weight_decay=Config.weight_decay, capturable=True)
step = RenderSupervisedStep(net, render, Config, optimizer=opt)
p, c, cube = synthetic_batch(32, dev, 0); tgt = step.make_targets(p, c, cube)
def timeit(f, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(5): step(tgt)
torch.cuda.current_stream().wait_stream(s)
print('eager ms', timeit(lambda: step(tgt)))
g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.graph(g):
    loss, _ = step(tgt)
print('capture s', time.perf_counter() - t0)
g.replay(); torch.cuda.synchronize()
print('loss after replay', float(loss))
print('graph ms', timeit(lambda: g.replay()))
print('loss', float(loss))
