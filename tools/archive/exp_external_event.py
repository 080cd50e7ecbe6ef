"""Does an EXTERNAL event record captured INSIDE a hipGraph order work on another stream at replay time?  torch refuses
torch.cuda.Event(external=True) on ROCm ("External events are disallowed in rocm"), so go to HIP directly:
hipEventRecordWithFlags(ev, capturing stream, hipEventRecordExternal) inside the capture, hipStreamWaitEvent(side, ev) after
each replay (the mechanism a graph-compatible overlapped all-reduce would need)."""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
x = torch.zeros(1 << 24, device="cuda")
y = torch.zeros(1 << 24, device="cuda")
out = torch.zeros(4, device="cuda")
side = torch.cuda.Stream()
def mk():
    e = ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(e), 2) == 0        # hipEventDisableTiming
    return e
ev, ev2 = mk(), mk()
def rec(e):
    r = hip.hipEventRecordWithFlags(e, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 1)   # hipEventRecordExternal
    assert r == 0, "hipEventRecordWithFlags -> %d" % r
def body():
    x.add_(1.0)
    rec(ev)
    for _ in range(50):
        y.add_(1.0)
    rec(ev2)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body(); body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
torch.cuda.synchronize()
x.zero_(); y.zero_()
for rep in range(3):
    g.replay()
    with torch.cuda.stream(side):
        assert hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), ev, 0) == 0
        out[0] = x[0]
        out[1] = y[0]
        assert hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), ev2, 0) == 0
        out[2] = y[0]
    torch.cuda.synchronize()
    print("replay", rep, "x seen after ev:", float(out[0]), "(want", rep + 1, ") y seen after ev:", float(out[1]), "(overlap if <", 50 * (rep + 1), ") y after ev2:", float(out[2]), "(want", 50 * (rep + 1), ")")
