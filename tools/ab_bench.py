import sys, runpy
sys.path.insert(0, '/root/repo')
from sipnet_amd import _lib
if sys.argv[1] != 'product':
    _lib.use_library(sys.argv[1])
sys.argv = ['bench.py'] + sys.argv[2:]
runpy.run_path('/root/repo/bench.py', run_name='__main__')
