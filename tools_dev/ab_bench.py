"""Dev: run bench.py against a differently built library (same-box A/B): EXCEL_AB_LIB=tools_dev/ab/X.so python tools_dev/ab_bench.py [bench args]"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("EXCEL_AB_LIB"):
    import excel_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["EXCEL_AB_LIB"])
sys.argv = ["bench.py"] + sys.argv[1:] + ([] if any(a.startswith("--power-seconds") for a in sys.argv[1:]) else ["--power-seconds", "0"])   # (A/B runs skip the power side-line)
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
