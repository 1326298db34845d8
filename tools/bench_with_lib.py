"""Runs bench.py against another build of the library (same-box A/B of compile-time choices; tuning only).
usage: bench_with_lib.py /path/to/libvqahot_variant.so [bench.py arguments]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vqa_transfer_externaldata_amd import _lib  # noqa: E402

_lib._LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
