"""TEST INFRASTRUCTURE, not a test: runs the reference's own optimiser (oracle/_ref/libcuba_ref_lm.so -- its CUDA sources compiled for
gfx950 as they are) on the KITTI-00 shape under the samples' protocol, so that `rocprofv3 --kernel-trace --stats -- python
tests/ref_kernels_profile_run.py` lists the reference's kernels with their durations on this GPU (scripts/r05/run_z9.sh).  The numbers
complement tests/test_ref_lm.py::test_reference_stage_times_on_this_gpu, which reads the reference's own stage timers."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import ref_lm                               # noqa: E402
from test_ref_lm import full_size_cases                 # noqa: E402

if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "kitti00_full"
    make, rk, iters = full_size_cases()[name]
    r = ref_lm.run(make(), rk, iters, nruns=2)
    print(name, "chi2", r["chi2"][0], "->", r["chi2"][-1], "stage timers (s):", r["profile"])
