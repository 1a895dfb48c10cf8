import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import essentials_amd as ea
ctx = ea.Context(0)
g = ea.Graph.rmat(ctx, int(sys.argv[1]) if len(sys.argv) > 1 else 22, 16, 1, 7)
d = torch.empty(g.n_rows, dtype=torch.int32, device="cuda")
w = torch.empty(g.n_rows, dtype=torch.float32, device="cuda")
for i in range(4):
    t0 = time.perf_counter(); _, st = ea.bfs(ctx, g, 0, d); t1 = time.perf_counter()
    _, st2 = ea.sssp(ctx, g, 0, w); t2 = time.perf_counter()
    print(f"bfs wall {1e3*(t1-t0):8.2f} ms enact {st.elapsed_ms:6.2f} | sssp wall {1e3*(t2-t1):8.2f} ms enact {st2.elapsed_ms:6.2f}", flush=True)
for f in (1.5, 0.1):
    t0 = time.perf_counter(); _, st = ea.bfs(ctx, g, 0, d, ea.Options(frontier_sizing_factor=f)); t1 = time.perf_counter()
    print(f"sizing factor {f}: bfs wall {1e3*(t1-t0):8.2f} ms enact {st.elapsed_ms:6.2f}")
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
p = ctypes.c_void_p()
for mb in (64, 805):
    t0 = time.perf_counter(); hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(mb << 20)); t1 = time.perf_counter(); hip.hipFree(p); t2 = time.perf_counter()
    print(f"hipMalloc {mb} MB: {1e3*(t1-t0):.2f} ms, hipFree {1e3*(t2-t1):.2f} ms")
