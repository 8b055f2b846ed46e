import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import numpy as np, torch, gen
from hhrl_vae_amd.algorithm import HATD3Triple, WorkerTD3
from hhrl_vae_amd import _lib
which = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64          # usage: td3_prof.py <worker|triple> [batch]
mk = lambda ug: WorkerTD3(gen.cav_cfg(cap=4096, batch=batch), use_graph=ug) if which == 'worker' else HATD3Triple(gen.light_cfg(cap=2048, batch=batch), use_graph=ug)
net = mk(False)
n = 2048
net.memory[:n] = np.random.RandomState(1).uniform(0, 1, size=(n, net.fam.row_width)); net.pointer = n
for _ in range(4): net.learn()
torch.cuda.synchronize()
lib = _lib.load(); lib.hhrl_prof_reset(); lib.hhrl_prof_enable(1)
t0 = time.perf_counter()
K = 20
for _ in range(K): net.learn()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
lib.hhrl_prof_enable(0)
rep = _lib.prof_report()
tot = sum(v['ms'] for v in rep.values()); nl = sum(v['launches'] for v in rep.values())
print(which, 'wall ms/update (profiling on)', 1e3*dt/K, 'gpu-kernel ms/update', tot/K, 'launches/update', nl/K)
for k, v in sorted(rep.items(), key=lambda kv: -kv[1]['ms']): print('  %-32s %6.1f launches %8.3f ms/update  %6.1f us avg  %6.1f TFLOP/s' % (k, v['launches']/K, v['ms']/K, 1e3*v['ms']/v['launches'], v['flops']/(v['ms']*1e-3)/1e12 if v['ms'] > 0 else 0.0))
lib.hhrl_prof_reset()
if os.environ.get('TD3_PROF_ONLY') == '1': sys.exit(0)
for ug in (False, True, "auto"):
    net = mk(ug)
    net.memory[:n] = np.random.RandomState(1).uniform(0, 1, size=(n, net.fam.row_width)); net.pointer = n
    for it in range(800):
        net.learn()
        f = net.launch_forms()
        if it >= 30 and len(f) == (2 if which == 'worker' else 7) and 'tuning' not in f.values(): break
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): net.learn()
    torch.cuda.synchronize()
    print('wall ms/update (no profiling, graphs=%s, %d graphs)' % (ug, len(net._graphs)), 1e3*(time.perf_counter()-t0)/100)
