"""Experiment driver: tools/prof_render.py with another build of the library (path in argv[1])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytrace_amd
raytrace_amd.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
exec(open(os.path.join(ROOT, "tools", "prof_render.py")).read())
