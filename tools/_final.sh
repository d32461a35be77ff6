cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_stress.py -m gpu -x -q 2>&1 | tail -2
python tools/soak.py 1500 all 2>&1 | grep -v amdgpu > gpurun_out/soak_r05.txt; tail -14 gpurun_out/soak_r05.txt
for c in 3 5; do python bench.py --config $c --init fitted > gpurun_out/r05_bench_config${c}_fitted.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r05_bench_config${c}_fitted.json').read().strip().splitlines()[-1]); print('config $c fitted', d['value'], d['unit'], d['ms_per_step'], 'ms')"; done
python - <<PY
import torch, sys
sys.path.insert(0, ".")
from dsf_amd.render_model.mano_layer import Render
from dsf_amd.train_step import synthetic_batch
import bench
render = Render("synthetic", "nyu", (588.03, 587.07, 320.0, 240.0), (640, 480)).cuda()
for g in bench.geometry_rooflines(render, 32):
    if g["id"] in ("K6", "K7"): print(g["id"], g["avg_launch_us"], "us")
PY
