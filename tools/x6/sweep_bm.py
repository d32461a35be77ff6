import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from try_fwd import run
for shape in ((32, 32, 128, 128, 3, 1, 1), (32, 16, 256, 256, 3, 1, 1), (32, 8, 512, 512, 3, 1, 1), (32, 32, 256, 256, 4, 2, 1), (32, 16, 256, 512, 3, 2, 1)):
    for sp in (1, 2, 4, 8):
        run(*shape, splits=sp, check=False)
    run(*shape, splits=0, check=(shape[1] == 8))
