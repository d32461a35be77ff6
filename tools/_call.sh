R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c26; mkdir -p $O; cd $R
timeout 2400 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -n 4 $O/pytest_gpu.log
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['kernel'],d['roofline']['frac'],d['roofline']['avg_launch_us'],d.get('roofline_critical',{}).get('frac'))"
timeout 900 python3 tools/ab_env.py --config 2 --var DSF_X6_WRWP_WGS --values 512 256 384 --rounds 8 > $O/ab_wgsp_c2.txt 2>&1; tail -n 3 $O/ab_wgsp_c2.txt
