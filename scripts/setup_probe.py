"""Host-side set-up cost of the plume case (FFM_TIMING=1 prints the stages).  usage: setup_probe.py n"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffm_import import ffm
n = int(sys.argv[1])
ctx = ffm.Context(0)
t0 = time.time(); case = ffm.Plume(ctx, (n, n, n), h=0.05, deltaT=1e-3); print("Plume(%d^3) %.1f s" % (n, time.time() - t0))
case.close()
