"""kernel-trace csv -> torch (at::native) kernels per step with full functor names.  usage: kglue.py trace.csv n_steps"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1]))); nst = float(sys.argv[2])
names = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r['Kernel_Name']
    if 'at::native' not in n and 'rocclr' not in n: continue
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    f = re.findall(r"(CUDAFunctor\w*<[^>]*>|\w+Functor<[^>]*>|direct_copy_kernel_cuda|\w+_kernel_cuda|CatArray\w+|reduce_kernel<[^,]*,[^,]*, at::native::ReduceOp<[^,]*, at::native::\w+|rocclr_\w+|index\w*|\w+_kernel_impl\w*)", n)
    key = (n.split("<")[0].replace("void at::native::", "")[:40] + " | " + ",".join(f[:2]))[:150]
    names[key][0] += 1; names[key][1] += d
tot = sum(v[1] for v in names.values())
print("torch/runtime kernels: %.3f ms/step, %.0f launches/step" % (tot / nst / 1e6, sum(v[0] for v in names.values()) / nst))
for k, v in sorted(names.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%7.1f x %7.1f us = %7.1f us  %s" % (v[0] / nst, v[1] / v[0] / 1e3, v[1] / nst / 1e3, k))
