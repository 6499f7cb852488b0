"""How long does the HOST take to enqueue one World.step launch (vmas_world_step_n returns before the GPU is done)?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
sc, w = bench.build_world(B, torch.device("cuda", 0), 4, 0, 0)
be = w._get_backend()
forces = bench.make_forces(w, 100, 1234, torch.device("cuda", 0))
be.step_n(100, forces); torch.cuda.synchronize()
for n in (100, 1000):
    f = forces[:100]
    t0 = time.perf_counter()
    for _ in range(n // 100):
        be.step_n(100, f)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B} n={n}: host enqueue {1e6*(t1-t0)/n:.2f} us/launch, until done {1e6*(t2-t0)/n:.2f} us/launch")
