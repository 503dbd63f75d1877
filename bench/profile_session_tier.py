"""cProfile of the per-batch session tier (tfsingle.py's loop: one sess.run with feed + fetches per batch)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["tfsingle.py", "--epochs", "3", "--logs_path", "/tmp/prof_logs"] + sys.argv[1:]
import runpy  # noqa: E402

pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tfsingle.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.print_callers("_cuda_getDeviceCount")
st.print_callers("is_available")
st.sort_stats("cumtime").print_stats("session.py|mnist_mlp.py|summary.py|hip_backend.py|mlp.py", 25)
