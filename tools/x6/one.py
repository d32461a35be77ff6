"""One conv_x6 forward shape, a few launches (target of rocprofv3 --pmc runs): python3 tools/x6/one.py B H Ci Co k stride pad"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from try_fwd import run
a = [int(v) for v in sys.argv[1:8]]
run(*a, check=False)
