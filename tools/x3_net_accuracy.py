"""fp32 parity mode on the BASELINE net against float64: max |HIP - f64| next to the reference's own fp32 error (the numbers of
tests/test_round2_goldens.py::test_tolerance_bookkeeping_k784_goldens), and the layer-2 check of the bench path over several epochs."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import subprocess
r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_round2_goldens.py", "-m", "gpu", "-q", "-k", "tolerance_bookkeeping", "-s"], capture_output=True, text=True)
print(r.stdout[-1500:])
for f in ("gpurun_out/tolerance_bookkeeping.json",):
    if os.path.exists(f):
        print(open(f).read())
