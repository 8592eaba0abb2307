"""Where a config-5 decode (N = 10^6, layered NMS) spends its time: per-kernel-class event times of the decoder's own
profile hooks next to the wall time of a step.  python tools/config5_breakdown.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _qldpc_loader import load
q = load()
from bench import make_frames

f5 = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
n5, k5 = 1000000, 800000
code = q.Code.ira(n5, k5, 0.125, 11, 3, 7)
enc = q.Encoder(code, "IRA", device=0)
cw, rx = make_frames(q, torch, code, enc, f5, 0.02, 5000, dev)
mag = torch.full((f5,), q.bsc_llr(0.02), dtype=torch.float32, device=dev)
cls = torch.zeros(n5, dtype=torch.uint8, device=dev); cls[k5:] = q.VN_PINNED
out = torch.empty((f5, (n5 + 31) // 32), dtype=torch.int32, device=dev)
for synd in (True, False):
    d = q.Decoder(code, k5, 50, rule="NMS", rule_param=0.75, enable_syndrome=synd, n_frames=f5, device=0, schedule="hlayered")
    d.set_stream(torch.cuda.current_stream(dev))
    def step():
        d.load_bits(rx, mag, cls); d.run(); d.fetch_packed(out)
    step(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / 5 * 1e3
    d.profile(True); d.profile_clear(); step(); torch.cuda.synchronize()
    ks = d.profile_read()
    print("early exit" if synd else "fixed 50", "frames", f5, "wall %.3f ms/step (unprofiled)" % wall, "sweeps", d.last_run_iterations)
    for k in ks:
        if k["launches"]:
            print("   %-16s %4d scopes  %8.3f ms  %7.1f us each  moved %.2f TB/s" % (k["name"], k["launches"], k["total_ms"], k["total_ms"] / k["launches"] * 1e3, k["moved_bytes"] / max(k["total_ms"], 1e-9) / 1e9))
    print("   sum of scopes %.3f ms" % sum(k["total_ms"] for k in ks))
    del d
