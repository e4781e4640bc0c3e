"""Cost of a cross-stream dependency on this device: N tiny kernels on one stream against N tiny kernels alternating between two
streams with an event record + wait between each (the shape a split pair/step launch order would need)."""
import time, torch
x = torch.zeros(64, device="cuda"); y = torch.zeros(64, device="cuda")
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream(priority=-1)
N = 4000
def same():
    with torch.cuda.stream(s1):
        for _ in range(N): x.add_(1.0)
def cross():
    evs = [torch.cuda.Event() for _ in range(N)]
    for i in range(N):
        s = s1 if i % 2 == 0 else s2
        if i: s.wait_event(evs[i - 1])
        with torch.cuda.stream(s): x.add_(1.0)
        evs[i].record(s)
def cross2():   # two independent chains (x on s1/s2 alternating, y on s2/s1): what two groups in anti-phase would issue
    evx = [torch.cuda.Event() for _ in range(N)]; evy = [torch.cuda.Event() for _ in range(N)]
    for i in range(N):
        sx = s1 if i % 2 == 0 else s2; sy = s2 if i % 2 == 0 else s1
        if i: sx.wait_event(evx[i - 1]); sy.wait_event(evy[i - 1])
        with torch.cuda.stream(sx): x.add_(1.0)
        with torch.cuda.stream(sy): y.add_(1.0)
        evx[i].record(sx); evy[i].record(sy)
for name, f in (("same stream", same), ("alternating streams, event between", cross), ("two chains in anti-phase", cross2)):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: {(t2 - t0) / N * 1e6:.2f} us per kernel (host enqueue {(t1 - t0) / N * 1e6:.2f} us)", flush=True)
