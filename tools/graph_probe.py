#!/usr/bin/env python3
"""GPU box: does a HIP graph of one log-psi forward (torch.cuda.graph around the C-ABI call) beat the eager launches?
    python tools/graph_probe.py [batch ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsolid_amd import network, systems

cell, klist = systems.build('bcc_li')
net_kw = dict(systems.DETNET_DEFAULTS)
net = network.make_solid_fermi_net(klist=klist, simulation_cell=cell, method_name='eval_slogdet', **net_kw)
params = net.init(0)
for B in [int(a) for a in sys.argv[1:]] or [512, 4096]:
    x = torch.as_tensor(systems.synthetic_walkers(cell, B), device='cuda')
    ref = net.apply(params, x).clone()
    torch.cuda.synchronize()
    def timeit(fn, n=50):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    t_eager = timeit(lambda: net.apply(params, x))
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): net.apply(params, x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = net.apply(params, x)
        t_graph = timeit(g.replay)
        g.replay(); torch.cuda.synchronize()
        print(f'B={B}: eager {t_eager:.3f} ms, graph replay {t_graph:.3f} ms, identical: {torch.equal(out, ref)}', flush=True)
    except Exception as e:
        print(f'B={B}: eager {t_eager:.3f} ms, graph capture failed: {type(e).__name__}: {str(e)[:300]}', flush=True)
