"""Node types of the captured config-3 step; the kernels right after each memset node (DOT dump)."""
import sys, os, re, ctypes, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.model.hourglass import PoseNetMANO
from dsf_amd.model.backbone import MANO_OCR_stage
from dsf_amd.train_step import MeshLossStep, RenderSupervisedStep, synthetic_batch, Config
r = Render('synthetic', 'nyu', (588.03, 587.07, 320., 240.), (640, 480)).cuda()
torch.manual_seed(0)
if os.environ.get("K", "3") == "3":
    o = MeshLossStep(PoseNetMANO(1, 21).cuda(), r, Config, n_points=512)
else:
    o = RenderSupervisedStep(MANO_OCR_stage('ResNet_stage_18', 21, True).cuda(), r, Config)
p, c, cube = synthetic_batch(4, "cuda", seed=2)
t = o.make_targets(p, c, cube)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    o(t); o(t)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph(keep_graph=True)
g.enable_debug_mode()
with torch.cuda.graph(g):
    o.forward_backward(t)
raw = g.raw_cuda_graph()
hip = ctypes.CDLL("libamdhip64.so")
n = ctypes.c_size_t(0)
print("hipGraphGetNodes rc", hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)), "nodes", n.value)
arr = (ctypes.c_void_p * n.value)()
hip.hipGraphGetNodes(ctypes.c_void_p(raw), arr, ctypes.byref(n))
types = collections.Counter()
for a in arr:
    ty = ctypes.c_int(-1); hip.hipGraphNodeGetType(ctypes.c_void_p(a), ctypes.byref(ty)); types[ty.value] += 1
print("node types (0 kernel, 1 memcpy, 2 memset):", dict(types))
g.debug_dump("/tmp/g.dot")
txt = open("/tmp/g.dot").read()
print("dot bytes", len(txt))
labels = dict(re.findall(r'"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', txt))
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
ms = [k for k, v in labels.items() if "MEMSET" in v.upper()]
print("memset nodes in dot:", len(ms))
nxt = collections.defaultdict(list)
for a, b in edges: nxt[a].append(b)
for m in ms[:20]:
    print(labels[m][:80].replace("\n", " "), "->", [labels.get(b, b)[:90].replace("\n", " ") for b in nxt[m]])
