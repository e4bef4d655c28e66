"""Start-up costs of a process that uses the C ABI (dev tool, GPU box).  usage: startup_probe.py [--torch]"""
import time, sys, os
t0 = time.time()
sys.path.insert(0, os.getcwd())
import numpy
t1 = time.time()
from smcounter_amd import cli
t2 = time.time()
from smcounter_amd import _lib
_lib.load(with_torch="--torch" in sys.argv)
t2b = time.time()
from smcounter_amd.engine import Engine
e = Engine(0)
t3 = time.time()
from smcounter_amd import synth
cfg = synth.CONFIGS["C2"]; P = synth.params_for(cfg)
db = synth.generate_native(cfg, 0, 100, P)
t4 = time.time()
R = e.call_batch_host(db, P)
t5 = time.time()
R = e.call_batch_host(db, P)
t6 = time.time()
print("numpy %.2f | package imports %.2f | bind ABI %.2f | Engine(0) %.2f | synth %.2f | first call %.2f | second call %.3f | torch imported: %s"
      % (t1 - t0, t2 - t1, t2b - t2, t3 - t2b, t4 - t3, t5 - t4, t6 - t5, "torch" in sys.modules))
