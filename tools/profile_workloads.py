#!/usr/bin/env python3
"""Workloads for rocprofv3 that bench.py's default run does not reach (developer tool): the edge-parallel engine on one block (the
daemon's case: qe_cn / qe_vn), the horizontal-layered schedule on the config-2 batch in fp32 and 8-bit (qk_cn_layer / qi_cn_layer),
and an early-exit decode with compaction (REMAP check pass, qk_compact_*).
  rocprofv3 --kernel-trace --stats -d out -- python3 tools/profile_workloads.py
  rocprofv3 --pmc FETCH_SIZE -d out_f -- python3 tools/profile_workloads.py     (and WRITE_SIZE, separately)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import _qldpc_loader  # noqa: E402
import bench  # noqa: E402

q = _qldpc_loader.load()
dev = torch.device("cuda", 0)
F = int(os.environ.get("FRAMES", "4096"))
code = q.Code.ira(65536, 52429, 0.125, 11, 3, 7)
enc = q.Encoder(code, "IRA")
cw, rx = bench.make_frames(q, torch, code, enc, F, 0.02, 1000, dev)
mag = torch.full((F,), q.bsc_llr(0.02), dtype=torch.float32, device=dev)
cls = torch.zeros(code.N, dtype=torch.uint8, device=dev)
cls[enc.K:] = 1


def run(dec, n, reps):
    for _ in range(reps):
        dec.load_bits(rx[:n], mag[:n], cls)
        dec.run()
        dec.fetch_packed()
    torch.cuda.synchronize()
    it, ok = dec.fetch_status()
    return float(it.float().mean()), int(ok.sum())


print("edges, 1 block x 20:", run(q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, n_frames=1, engine="edges"), 1, 20))
print("edges SPA, 1 block x 20:", run(q.Decoder(code, enc.K, 50, rule="SPA", n_frames=1, engine="edges"), 1, 20))
print("hlayered f32, %d frames, 10 sweeps fixed x 2:" % F, run(q.Decoder(code, enc.K, 10, rule="NMS", rule_param=0.75, n_frames=F, schedule="hlayered", enable_syndrome=False), F, 2))
print("hlayered i8, %d frames, 10 sweeps fixed x 2:" % F, run(q.Decoder(code, enc.K, 10, rule="NMS", rule_param=0.75, n_frames=F, schedule="hlayered", enable_syndrome=False, msg_dtype="i8"), F, 2))
print("flooding f32 early exit + compaction x 2:", run(q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, n_frames=F), F, 2))
print("flooding i8 early exit + compaction x 2:", run(q.Decoder(code, enc.K, 50, rule="NMS", rule_param=0.75, n_frames=F, msg_dtype="i8"), F, 2))
