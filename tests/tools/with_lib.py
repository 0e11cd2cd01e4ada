"""Run a tool against a variant library: python tests/tools/with_lib.py tests/tools/libpw_var_X.so tests/tools/tool.py args..."""
import pathlib
import runpy
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from pywindow_amd import _lib

_lib.LIB_PATH = pathlib.Path(sys.argv[1]).resolve()
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
