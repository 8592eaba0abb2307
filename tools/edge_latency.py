import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import _qldpc_loader
q = _qldpc_loader.load()
import torch
torch.cuda.init()
code = q.Code.ira(65536, 52429, 0.125, 11, 3, 7)
enc = q.Encoder(code, "IRA")
rng = np.random.default_rng(1)
for F in (1, 2, 4, 8):
    cw = enc.encode(rng.integers(0, 2, (F, enc.K)))
    noisy = cw.copy(); noisy[:, :enc.K] ^= rng.random((F, enc.K)) < 0.02
    bits = torch.from_numpy(q.pack_bits(noisy).view(np.int32)).cuda()
    mag = torch.full((F,), q.bsc_llr(0.02), dtype=torch.float32, device="cuda")
    cls = torch.zeros(code.N, dtype=torch.uint8, device="cuda"); cls[enc.K:] = 1
    dec = q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, n_frames=F, engine="edges")
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            dec.load_bits(bits, mag, cls); dec.run(); out = dec.fetch_packed()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    it, ok = dec.fetch_status()
    print("WGS=%s F=%d %.1f us/decode iters %s ok %s correct %s" % (os.environ.get("QLDPC_EDGE_WGS"), F, dt * 1e6, it.cpu().numpy().tolist(), bool(ok.all()), bool((out.cpu().numpy().view(np.uint32) == q.pack_bits(cw)).all())), flush=True)
