#!/usr/bin/env python3
"""bench.py against an experiment variant of the library: RT_LIB=<path to librt_amd_exp_*.so> python3 tools/trace_exp.py <bench args>
(csrc/Makefile: make librt_amd_exp.so EXPFLAGS=... EXPNAME=...).  The shipped bench.py reads no such knob."""
import importlib, os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
rt = importlib.import_module("gpu-raytracing_amd")
if os.environ.get("RT_LIB"):
    rt.LIB_PATH = os.path.abspath(os.environ["RT_LIB"])
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
