"""Run tools/dbg_sweep17.py against an alternative build of the library: python tools/dbg_alt.py <lib.so> seeds..."""
import sys, os, runpy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from andvaranaut_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dbg_sweep17.py"), run_name="__main__")
