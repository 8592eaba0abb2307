#!/usr/bin/env python3
"""bench.py -- reconciled-key throughput of the batched LDPC decoder on MI355X.

Workload (BASELINE.json configs[1]): rate-0.8 N = 65536 DVB-like IRA code (K 52429, M 13107,
E 235925), flooding normalised min-sum (alpha 0.75), 50 BP iterations, QBER 2 %, 4096 synthetic
sifted-key frames per GPU.  One "step" = one pass of the hot path over one batch whose packed bits
are already in HBM: frame formation (LLRs from bits + QBER) -> 50 iterations -> packed hard
decisions (+ for N > 1 the RCCL gather of decoded blocks to rank 0).

value = K bits of every successfully reconciled frame / wall time (whole job, all ranks), Mbit/s.
The headline run executes all 50 iterations for every frame (enable_syndrome = 0: no work skipped);
the AFF3CT-default early-exit mode is reported beside it under "early_exit".

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# PEG-built codes (the sessions' mother codes, the second code of the FER leg) are kept as checksummed edge lists here, so that Alice's and
# Bob's sessions of the config-3 leg build each code once (construction is outside every timed region either way)
import tempfile  # noqa: E402

os.environ.setdefault("QLDPC_CODE_CACHE", os.path.join(tempfile.gettempdir(), "qldpc_code_cache_%d" % os.getuid()))
os.makedirs(os.environ["QLDPC_CODE_CACHE"], exist_ok=True)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_frames(q, torch, code, enc, frames, qber, seed, device):
    """Synthetic sifted-key epochs: Alice's codewords, Bob's copy through a BSC(qber) on the key VNs,
    parity VNs disclosed (pinned).  Everything is generated on the device, packed MSB-first."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    K, N = enc.K, code.N
    Wk, Wn = (K + 31) // 32, (N + 31) // 32
    w8 = torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.float16, device=device)

    def pack(bits, W):      # bits [F, n] bool -> int32 words [F, W] (bytes by an exact fp16 dot product, then four bytes per word)
        F, n = bits.shape
        pad = W * 32 - n
        if pad:
            bits = torch.cat([bits, torch.zeros((F, pad), dtype=bits.dtype, device=device)], 1)
        by = (bits.view(F, W * 4, 8).to(torch.float16) @ w8).to(torch.int32).view(F, W, 4)
        return (by[:, :, 0] << 24) | (by[:, :, 1] << 16) | (by[:, :, 2] << 8) | by[:, :, 3]

    tail = K & 31
    cw_chunks, rx_chunks = [], []
    step = 1024
    for lo in range(0, frames, step):
        n = min(step, frames - lo)
        info = torch.randint(-2 ** 31, 2 ** 31, (n, Wk), generator=g, device=device, dtype=torch.int64).to(torch.int32)      # iid uniform key bits, already packed
        if tail:
            info[:, -1] &= -(1 << (32 - tail))
        cw = enc.encode_packed(info)
        noise = pack(torch.rand((n, K), generator=g, device=device) < qber, Wn)      # flips only on key VNs 0..K-1; parity bits are disclosed exactly
        cw_chunks.append(cw)
        rx_chunks.append(cw ^ noise)
    return torch.cat(cw_chunks), torch.cat(rx_chunks)


def clopper_pearson_upper(k, n, conf=0.95):
    """one-sided upper confidence bound on a binomial proportion (k events in n trials)"""
    from scipy.stats import beta
    return 1.0 if k >= n else float(beta.ppf(conf, k + 1, n - k))


def cpu_baseline(code, frames_llr_fn, rule, param, n_ite, K):
    """The CPU oracle (kind 'port': scalar fp32 C restatement of the AFF3CT decoder, OpenMP over
    frames) on this host's cores, bounded sample of the same workload."""
    from oracle import oracle as O
    import numpy as np
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    var, chk = code.edges()
    og = O.Graph.from_edges(code.N, code.M, var, chk)
    llr = frames_llr_fn(cores)
    t0 = time.time()
    O.decode(og, llr, rule, param, n_ite, "flooding", False, 1, n_threads=cores)
    t1 = time.time() - t0
    reps = max(1, min(64, int(round(12.0 / max(t1, 1e-3)))))
    n = cores * reps
    llr = frames_llr_fn(n)
    t0 = time.time()
    r = O.decode(og, llr, rule, param, n_ite, "flooding", False, 1, n_threads=cores)
    dt = time.time() - t0
    ok = int((r["synd_ok"] == 1).sum())
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return dict(value=ok * K / dt / 1e6, unit="Mbit/s", cores=cores, cpu_model=model, kind="port",
                sample="%d frames of the same workload (flooding %s %.2f, %d iterations fixed), %.1f s wall" % (n, rule, param, n_ite, dt)), r, llr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (default: 4096 = BASELINE config 2 on one GPU; 32768 = config 4's shard when WORLD_SIZE > 1)")
    ap.add_argument("--no-config3", action="store_true", help="skip the config-3 leg (multi-rate stream through the reconciliation sessions)")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 leg (N = 10^6 horizontal-layered)")
    ap.add_argument("--config5-frames", default="64,256", help="batches of the config-5 leg: SURVEY 8d's 64 frames, then (optional) a chip-filling batch")
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--k", type=int, default=52429)
    ap.add_argument("--qber", type=float, default=0.02)
    ap.add_argument("--n-ite", type=int, default=50)
    ap.add_argument("--alpha", type=float, default=0.75)
    ap.add_argument("--rule", default="NMS", help="update rule (the headline workload is NMS)")
    ap.add_argument("--schedule", default="flooding", help="flooding (headline) | hlayered (experiment)")
    ap.add_argument("--frames-per-lane", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-early", action="store_true", help="skip the early-exit leg")
    ap.add_argument("--no-fp16", action="store_true", help="skip the fp16-message-storage variant leg")
    ap.add_argument("--no-layered", action="store_true", help="skip the horizontal-layered leg on the headline code")
    ap.add_argument("--no-int8", action="store_true", help="skip the 8-bit fixed-point variant leg")
    ap.add_argument("--no-spa", action="store_true", help="skip the flooding-SPA leg on the headline code (the harness's default rule)")
    ap.add_argument("--no-fer-deep", action="store_true", help="skip the deep frame-error-rate leg (>= 2^20 frames at the headline QBER + waterfall points)")
    ap.add_argument("--fer-frames", type=int, default=1 << 20, help="frames of the deep FER point at the headline QBER")
    ap.add_argument("--peg", type=int, default=2, help="PEG depth of the second code the FER / config-3 legs report beside the seeded shuffle (0 = skip)")
    ap.add_argument("--msg-dtype", default="f32", choices=["f32", "f16", "i8"], help="experiment: message storage of the main legs (the contract run uses f32)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import _qldpc_loader
    q = _qldpc_loader.load()
    sys.path.insert(0, os.path.join(ROOT, "qcrypto-ldpc_amd"))
    import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libqldpc has no CPU fallback")
    # one process per GPU; QLDPC_DIST_BACKEND=gloo lets several ranks rehearse the N > 1 path on a box with fewer GPUs
    backend = os.environ.get("QLDPC_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    comm_device = device if backend == "nccl" else torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()

    # BASELINE configs[1]: 4 096 frames on one GPU; configs[3]: 262 144 frames over 8 GPUs = 32 768 per GPU, the shard every rank of an
    # N > 1 run decodes (weak scaling: the per-GPU batch stays config 4's whatever N is)
    F = args.frames if args.frames > 0 else (4096 if world == 1 else 32768)
    total_frames = F * world
    lo, hi = shard.frame_range(total_frames, world, rank)
    code = q.Code.ira(args.n, args.k, 0.125, 11, 3, 7)
    enc = q.Encoder(code, "IRA", device=local_rank)
    K, N = enc.K, code.N
    t0 = time.time()
    cw, rx = make_frames(q, torch, code, enc, F, args.qber, 1000 + rank, device)
    mag = torch.full((F,), q.bsc_llr(args.qber), dtype=torch.float32, device=device)
    cls = torch.zeros(N, dtype=torch.uint8, device=device)
    cls[K:] = q.VN_PINNED
    torch.cuda.synchronize()
    if rank == 0:
        log("frames ready: %d per GPU (%.1f s), code N=%d K=%d M=%d E=%d" % (F, time.time() - t0, N, K, code.M, code.E))

    def make_decoder(enable_syndrome, msg_dtype=None, schedule=None, rule=None):
        msg_dtype = msg_dtype or args.msg_dtype
        d = q.Decoder(code, K, args.n_ite, rule=rule or args.rule, rule_param=args.alpha if rule is None else 0.0, enable_syndrome=enable_syndrome,
                      n_frames=F, device=local_rank, frames_per_lane=args.frames_per_lane, msg_dtype=msg_dtype, schedule=schedule or args.schedule)
        d.set_stream(torch.cuda.current_stream(device))
        return d

    out = torch.empty((F, (N + 31) // 32), dtype=torch.int32, device=device)
    gbuf = shard.gather_buffer(out, total_frames, dst=0) if world > 1 else None      # rank 0's receive buffer, allocated once

    def step(dec):
        dec.load_bits(rx, mag, cls)
        dec.run()
        dec.fetch_packed(out)
        if world > 1:
            shard.gather_blocks(out, total_frames, dst=0, out=gbuf)      # the one collective of the job: RCCL gather of decoded blocks

    def timed(dec, steps, warmup):
        for _ in range(warmup):
            step(dec)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            step(dec)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        if world > 1:
            td = torch.tensor([dt], dtype=torch.float64, device=comm_device)
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
            dt = float(td.item())
        return dt

    def verdicts(dec):
        it, ok = dec.fetch_status()
        good = ((out == cw).all(dim=1)) & (ok == 1)
        stats = torch.tensor([float(good.sum()), float(it.sum()), float(F)], dtype=torch.float64, device=comm_device)
        if world > 1:
            dist.all_reduce(stats)
        return stats.tolist()

    # ---- headline: all n_ite iterations executed for every frame --------------------------------
    dec = make_decoder(False)
    dec.profile(True)
    for _ in range(args.warmup):
        step(dec)
    dec.profile_clear()
    dt = timed(dec, args.steps, 0)
    kstats = {s["name"]: s for s in dec.profile_read()}
    dec.profile(False)
    good, it_sum, n_all = verdicts(dec)
    out_headline = out.clone()            # the later legs reuse `out`; the CPU cross-check below is about THIS leg
    value = good * K * args.steps / dt / 1e6
    fer = 1.0 - good / n_all
    if args.schedule != "flooding":          # experiment mode: one kernel family, no CPU / fp16 legs
        kstats["cn_update"] = kstats["vn_update"] = kstats["layer_update"]
    if args.schedule != "flooding" or args.msg_dtype != "f32":
        args.no_fp16 = args.no_int8 = args.no_cpu = True
    cn = kstats["cn_update"]
    vn = kstats["vn_update"]
    cn_avg_s = cn["total_ms"] / cn["launches"] * 1e-3
    cn_bytes = cn["alg_bytes"] / cn["launches"]
    achieved = cn_bytes / cn_avg_s / 1e9
    vn_achieved = (vn["alg_bytes"] / vn["launches"]) / (vn["total_ms"] / vn["launches"] * 1e-3) / 1e9
    vn_moved = (vn["moved_bytes"] / vn["launches"]) / (vn["total_ms"] / vn["launches"] * 1e-3) / 1e9
    distinct = {id(s): s for s in kstats.values()}.values()       # hlayered aliases cn / vn to the layer pass: count it once
    step_bytes = sum(s["alg_bytes"] for s in distinct) / args.steps
    step_moved = sum(s["moved_bytes"] for s in distinct) / args.steps
    fixed_iters = dec.last_run_iterations
    del dec
    torch.cuda.empty_cache()

    # ---- AFF3CT-default mode: per-frame syndrome early exit --------------------------------------
    early = None
    if not args.no_early:
        dec = make_decoder(True)
        dte = timed(dec, max(1, args.steps), 1)
        g2, its2, n2 = verdicts(dec)
        st = dec.last_run_stats()
        early = dict(value=g2 * K * max(1, args.steps) / dte / 1e6, unit="Mbit/s", fer=1.0 - g2 / n2,
                     avg_iterations=its2 / n2, iterations_launched=dec.last_run_iterations,
                     ms_per_step=dte / max(1, args.steps) * 1e3,
                     # sum of the frames' iteration counts / lane-iterations the groups executed (rank 0's shard)
                     useful_work=float(dec.fetch_status()[0].sum().item()) / max(1, st["lane_iterations"]), compactions=st["compactions"])
        del dec
        torch.cuda.empty_cache()

    # ---- variants (NOT the headline): narrower message storage -------------------------------------------------
    # f16: messages rounded to binary16 in HBM, fp32 arithmetic.  i8: 8-bit fixed-point min-sum (quantiser 8 steps per LLR
    # unit, messages saturating at +-127).  Both are FER-tolerance class against the AFF3CT float build and bit-exact
    # against the oracle run with the same arithmetic (tests/test_parity_gpu.py, tests/test_i8_gpu.py).
    def variant(msg_dtype, rule=None):
        res = {}
        for name, synd in (("fixed", False), ("early_exit", True)):
            dec = make_decoder(synd, msg_dtype, rule=rule)
            dec.profile(True)
            step(dec)
            dec.profile_clear()
            dth = timed(dec, max(1, args.steps), 0)
            ks = {s["name"]: s for s in dec.profile_read()}
            dec.profile(False)
            gh, ith, nh = verdicts(dec)
            sth = dec.last_run_stats()
            cnh, vnh = ks["cn_update"], ks["vn_update"]
            res[name] = dict(value=gh * K * max(1, args.steps) / dth / 1e6, unit="Mbit/s", fer=1.0 - gh / nh, avg_iterations=ith / nh,
                             ms_per_step=dth / max(1, args.steps) * 1e3,
                             useful_work=(float(dec.fetch_status()[0].sum().item()) / max(1, sth["lane_iterations"])) if synd else 1.0, compactions=sth["compactions"],
                             cn_update_GBs=(cnh["alg_bytes"] / cnh["launches"]) / (cnh["total_ms"] / cnh["launches"] * 1e-3) / 1e9,
                             vn_update_GBs=(vnh["alg_bytes"] / vnh["launches"]) / (vnh["total_ms"] / vnh["launches"] * 1e-3) / 1e9)
            del dec
            torch.cuda.empty_cache()
        return res

    # ---- the same code, rule and iteration cap on AFF3CT's OTHER schedule: horizontal layered (Decoder_LDPC_BP_horizontal_layered; the harness keeps
    # the case commented out, VAR/main.cpp (alist-v1.0.1):220-258).  Not the headline -- the reference runs flooding -- but the fastest bit-exact fp32
    # operating point of this library: half the iterations, and min-sum sweeps run on the compressed check state (qldpc_kernels_cst.h).
    def layered():
        res = {}
        for name, synd in (("fixed", False), ("early_exit", True)):
            dec = make_decoder(synd, "f32", "hlayered")
            dec.profile(True)
            step(dec)
            dec.profile_clear()
            dtl = timed(dec, max(1, args.steps), 0)
            lay = {s["name"]: s for s in dec.profile_read()}["layer_update"]
            dec.profile(False)
            gl, itl, nl = verdicts(dec)
            res[name] = dict(value=gl * K * max(1, args.steps) / dtl / 1e6, unit="Mbit/s", fer=1.0 - gl / nl, avg_sweeps=itl / nl, sweeps_launched=dec.last_run_iterations,
                             ms_per_step=dtl / max(1, args.steps) * 1e3, avg_sweep_ms=lay["total_ms"] / lay["launches"])
            if not synd:      # with early exit finished groups leave the sweeps: the bytes of a launch are not those of the full batch, no fraction is quoted
                res[name].update(roofline_frac=lay["alg_bytes"] / (lay["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 roofline_moved_frac=lay["moved_bytes"] / (lay["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
            del dec
            torch.cuda.empty_cache()
        res["workload"] = "the headline code and batch, horizontal-layered %s(%.2f) over %d colour layers, <= %d sweeps" % (args.rule, args.alpha, code.n_layers, args.n_ite)
        res["note"] = "roofline_frac prices 4 E rows per sweep (section 8(d)); min-sum sweeps move 2 E + 8 M rows (compressed check state): roofline_moved_frac"
        return res

    lay2 = layered() if (not args.no_layered and world == 1 and args.schedule == "flooding" and args.msg_dtype == "f32") else None

    fp16 = variant("f16") if not args.no_fp16 else None
    int8 = variant("i8") if (not args.no_int8 and args.rule in ("MS", "OMS", "NMS")) else None
    # SURVEY 8(d) config 2 names SPA beside NMS: Update_rule_SPA is the harness's default (BS/src/main.cpp:193) and the rule the reference's
    # published 0.241 Mb/s was measured with.  Same code, frames, schedule and iteration cap as the headline; fp32; tolerance class (one exp,
    # one rcp and one log per edge on the hardware's transcendental units), so NOT the bit-exact headline.
    spa = None
    if not args.no_spa and args.rule != "SPA" and args.schedule == "flooding" and args.msg_dtype == "f32":
        spa = variant("f32", rule="SPA")
        for leg in ("fixed", "early_exit"):
            spa[leg]["cn_update_frac"] = spa[leg]["cn_update_GBs"] / HBM_PEAK_GBS
        spa["workload"] = "the headline code and batch, flooding SPA (the harness default), fp32, <= %d iterations" % args.n_ite
        spa["parity_note"] = "tolerance class against the oracle (hardware exp / log / rcp); the oracle's SPA is pinned to AFF3CT by the reference's known-answer vector"

    # ---- the FER half of the metric: depth that means something -------------------------------------------------------------------
    # Early-exit mode (AFF3CT's default), fresh seeds per batch.  Per point: frames, frame errors (no valid codeword OR a codeword other
    # than Alice's), UNDETECTED errors (syndrome zero and word != Alice's: what a QKD user needs beside the CRC), FER with its one-sided
    # 95 % Clopper-Pearson upper bound, mean / max iterations.  The reference publishes FE / FRA per QBER from 30 frames a point on
    # DVB-S2 rate 0.8 (BS/data_dvb/data3 (DVB S2)/DVB_S2_N_64800_K_51840_CR_0.8.txt:31-36: SPA 0 / 30 up to 3.0 %, 30 / 30 at 3.5 %;
    # NMS with factor 1: 30 / 30 at 1.5 %, :24-25) -- printed beside ours as context (another H: DVB-S2 tables are AFF3CT built-ins).
    def fer_point(code_, enc_, rule, param, qber, n_frames, seed0, batch=4096, schedule="flooding"):
        dec_ = q.Decoder(code_, enc_.K, args.n_ite, rule=rule, rule_param=param, enable_syndrome=True, n_frames=batch, device=local_rank, schedule=schedule)
        dec_.set_stream(torch.cuda.current_stream(device))
        mag_ = torch.full((batch,), q.bsc_llr(qber), dtype=torch.float32, device=device)
        out_ = torch.empty((batch, (code_.N + 31) // 32), dtype=torch.int32, device=device)
        fails = undet = done_frames = 0
        it_sum, it_max = 0.0, 0
        t_dec = 0.0
        b_i = 0
        while done_frames < n_frames:
            cw_, rx_ = make_frames(q, torch, code_, enc_, batch, qber, seed0 + b_i, device)
            torch.cuda.synchronize()
            t = time.perf_counter()
            dec_.load_bits(rx_, mag_, cls)
            dec_.run()
            dec_.fetch_packed(out_)
            it_, ok_ = dec_.fetch_status()
            torch.cuda.synchronize()
            t_dec += time.perf_counter() - t
            same = (out_ == cw_).all(dim=1)
            fails += int((~(same & (ok_ == 1))).sum())
            undet += int(((ok_ == 1) & ~same).sum())
            it_sum += float(it_.sum())
            it_max = max(it_max, int(it_.max()))
            done_frames += batch
            b_i += 1
        del dec_
        torch.cuda.empty_cache()
        return dict(qber=qber, rule="%s(%.2f)" % (rule, param) if rule != "SPA" else "SPA", frames=done_frames, frame_errors=fails, undetected_errors=undet,
                    fer=fails / done_frames, fer_upper_95=clopper_pearson_upper(fails, done_frames), avg_iterations=it_sum / done_frames, max_iterations=it_max,
                    decode_Mbit_s=(done_frames - fails) * enc_.K / max(t_dec, 1e-9) / 1e6)

    def fer_deep():
        t_all = time.perf_counter()
        res = {"workload": "rate-%.1f N=%d IRA LDPC, flooding, <= %d iterations with AFF3CT's early exit, batches of 4096 fresh frames" % (K / N, N, args.n_ite),
               "headline_qber": fer_point(code, enc, args.rule, args.alpha, args.qber, args.fer_frames, 100000)}
        wf = []
        for rule, param in ((args.rule, args.alpha), ("SPA", 0.0)):
            for qb in (0.025, 0.0275, 0.03, 0.0325):
                wf.append(fer_point(code, enc, rule, param, qb, 65536, 200000 + int(qb * 1e5)))
        res["waterfall_seeded_shuffle"] = wf
        # the layered schedule (the layered_schedule leg's operating point: sweeps on the compressed check state) on the same frames as the flooding points above
        res["layered_schedule"] = [fer_point(code, enc, args.rule, args.alpha, args.qber, args.fer_frames, 100000, schedule="hlayered")] + \
                                  [fer_point(code, enc, args.rule, args.alpha, qb, 65536, 200000 + int(qb * 1e5), schedule="hlayered") for qb in (0.0275, 0.03)]
        if args.peg > 0:
            # the same degree profile with the information part grown by progressive edge growth: what the sessions use with peg_depth
            code_p = q.Code.ira_peg(args.n, args.k, 0.125, 11, 3, args.peg, 7)
            enc_p = q.Encoder(code_p, "IRA", device=local_rank)
            res["waterfall_peg_depth_%d" % args.peg] = [fer_point(code_p, enc_p, rule, param, qb, 65536, 300000 + int(qb * 1e5))
                                                         for rule, param in ((args.rule, args.alpha), ("SPA", 0.0)) for qb in (0.03, 0.0325)]
        res["reference_rows"] = {"source": "BS/data_dvb/data3 (DVB S2)/DVB_S2_N_64800_K_51840_CR_0.8.txt (AFF3CT DVB-S2 rate 0.8, 30 frames per point: FE / FRA)",
                                 "SPA": {"0.020": "0/30", "0.025": "0/30", "0.030": "0/30", "0.035": "30/30"}, "NMS(1.00)": {"0.010": "0/30", "0.015": "30/30"}}
        res["seconds"] = time.perf_counter() - t_all
        return res

    # ---- BASELINE config 5: N = 10^6 irregular LDPC, horizontal-layered schedule, per-sweep syndrome early termination ----------
    def config5(f5=64):
        n5, k5 = 1000000, 800000
        code5 = q.Code.ira(n5, k5, 0.125, 11, 3, 7)
        enc5 = q.Encoder(code5, "IRA", device=local_rank)
        cw5, rx5 = make_frames(q, torch, code5, enc5, f5, args.qber, 5000, device)
        mag5 = torch.full((f5,), q.bsc_llr(args.qber), dtype=torch.float32, device=device)
        cls5 = torch.zeros(n5, dtype=torch.uint8, device=device)
        cls5[k5:] = q.VN_PINNED
        out5 = torch.empty((f5, (n5 + 31) // 32), dtype=torch.int32, device=device)
        res = {}
        for name, synd in (("fixed", False), ("early_exit", True)):
            d5 = q.Decoder(code5, k5, args.n_ite, rule="NMS", rule_param=args.alpha, enable_syndrome=synd, n_frames=f5, device=local_rank, schedule="hlayered")
            d5.set_stream(torch.cuda.current_stream(device))

            def step5():
                d5.load_bits(rx5, mag5, cls5)
                d5.run()
                d5.fetch_packed(out5)
            d5.profile(True)
            step5()
            d5.profile_clear()
            torch.cuda.synchronize()
            t = time.perf_counter()
            n5steps = 3
            for _ in range(n5steps):
                step5()
            torch.cuda.synchronize()
            dt5 = time.perf_counter() - t
            ks = {s_["name"]: s_ for s_ in d5.profile_read()}
            d5.profile(False)
            it5, ok5 = d5.fetch_status()
            good5 = float((((out5 == cw5).all(dim=1)) & (ok5 == 1)).sum())
            lay = ks["layer_update"]
            cst5 = lay["moved_bytes"] < lay["alg_bytes"]
            res[name] = dict(value=good5 * k5 * n5steps / dt5 / 1e6, unit="Mbit/s", fer=1.0 - good5 / f5, avg_sweeps=float(it5.float().mean()),
                             sweeps_launched=d5.last_run_iterations, ms_per_step=dt5 / n5steps * 1e3,
                             roofline=dict(bound="hbm", kernel=("qk_cn_layer_cst (min-sum on the compressed check state: 2 E posterior rows + 8 M state rows moved per sweep "
                                                                "instead of section 8(d)'s 4 E rows; one sweep = all colour layers)" if cst5 else "qk_cn_layer (one sweep = all colour layers)"),
                                           peak=HBM_PEAK_GBS, unit="GB/s",
                                           alg_bytes_per_sweep=lay["alg_bytes"] / lay["launches"], moved_bytes_per_sweep=lay["moved_bytes"] / lay["launches"],
                                           avg_sweep_ms=lay["total_ms"] / lay["launches"], sweeps=lay["launches"],
                                           achieved=lay["alg_bytes"] / (lay["total_ms"] * 1e-3) / 1e9, frac=lay["alg_bytes"] / (lay["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           moved=lay["moved_bytes"] / (lay["total_ms"] * 1e-3) / 1e9, moved_frac=lay["moved_bytes"] / (lay["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS))
            del d5
            torch.cuda.empty_cache()
        res["workload"] = "N=%d K=%d IRA LDPC (E=%d, %d colour layers), horizontal-layered NMS(%.2f), <= %d sweeps, syndrome test every sweep, QBER %.1f %%, %d frames" % (
            n5, k5, code5.E, code5.n_layers, args.alpha, args.n_ite, args.qber * 100, f5)
        # HBM traffic of one sweep from the committed PMC summary of this workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950-corrected;
        # tools/gpu/r3_profiles2.sh): a constant read from profiles/, labelled as such -- per sweep = per-launch bytes of the layer kernels x launches per sweep
        try:
            pmc5 = json.load(open(os.path.join(ROOT, "profiles", "r03_config5_cst_pmc_hbm_traffic.json")))
            lk = {k_: v_ for k_, v_ in pmc5["kernels"].items() if k_.startswith("qk_cn_layer_cst")}
            if lk and f5 == pmc5["workload"]["frames"] and cst5:
                per_sweep = sum(v_["hbm_bytes_corrected"] * v_["launches"] for v_ in lk.values()) / (sum(v_["launches"] for v_ in lk.values()) / code5.n_layers)
                for name in ("fixed", "early_exit"):
                    res[name]["roofline"]["traffic"] = per_sweep
                    res[name]["roofline"]["traffic_source"] = "profiles/r03_config5_cst_pmc_hbm_traffic.json (" + pmc5["source"].split(" -- ", 1)[-1] + ")"
        except Exception as ex:      # noqa: BLE001
            log("no PMC traffic file for config 5: %s" % ex)
        res["roofline_note"] = ("frac prices SURVEY 8(d)'s algorithmic bytes (4 E message / posterior rows per sweep) against the measured sweep time; the min-sum sweep keeps a compressed "
                                "check state and MOVES 2 E + 8 M rows (0.61 of them on this code), so frac can pass 1 on full launches -- moved_frac is the share of the HBM peak actually used")
        return res

    # ---- BASELINE config 3: multi-rate H set {0.5, 0.7, 0.8, 0.9} chosen per epoch from the estimated QBER, a stream of epochs through
    #      the reconciliation sessions (what the ecd2 handlers call), HOST buffers in and out ----------------------------------------
    def config3(peg_depth=2, rate_gap=None, gap_profile=0, schedule="auto"):
        epochs_n, key_bits, batch = 512, 52429, 512      # every rate group of the stream is one batch of its decoder
        rng = np.random.default_rng(42)
        qbers = rng.uniform(0.005, 0.06, epochs_n).astype(np.float32)
        alice = rng.integers(0, 2, (epochs_n, key_bits)).astype(np.uint8)
        bob = alice ^ (rng.random((epochs_n, key_bits)) < qbers[:, None])
        aw, bw = q.pack_bits(alice), q.pack_bits(bob)
        kw = dict(device=local_rank, max_blocks=batch, peg_depth=peg_depth, rate_gap=rate_gap, gap_profile=gap_profile, schedule=schedule)
        ra, rb = q.Recon(**kw), q.Recon(**kw)
        keys = [aw[i] for i in range(epochs_n)]
        ra.encode_blocks(keys, [key_bits] * epochs_n, qbers)                  # builds the codes of the table (warm-up)
        t = time.perf_counter()
        msgs, pars = ra.encode_blocks(keys, [key_bits] * epochs_n, qbers)
        t_enc = time.perf_counter() - t
        groups = {}
        for i, m in enumerate(msgs):
            groups.setdefault((m.rate_index, m.code_k, m.code_m), []).append(i)
        # Bob: ONE qldpc_recon_decode_blocks call for the whole stream, host buffers in and out.  The library groups the epochs by code,
        # runs the rate groups side by side (a lane = host worker + compute stream + copy stream each), stages the next batch while one
        # decodes, and verifies (CRC-32, corrected-bit count) on the device.  Arguments are marshalled once, outside the timed region:
        # what is timed is the C call (qcrypto-ldpc_amd/host/qldpc_stream.c times the same call from C).
        call = rb.prepare_decode([bw[i] for i in range(epochs_n)], [key_bits] * epochs_n, qbers, msgs, pars)
        call.run()                                                            # Bob's decoders of the table (warm-up)
        reps, dts = 3, []
        for _ in range(reps):
            call.reset()
            t = time.perf_counter()
            call.run()
            dts.append(time.perf_counter() - t)
        dt3 = sum(dts) / reps
        ok = (call.status == 0) & np.array([(call.keys[i] == aw[i]).all() for i in range(epochs_n)])
        undetected = int(((call.status == 0) & ~ok).sum())
        iters = call.iterations.copy()
        rb.profile(True)
        call.reset()
        call.run()                                                            # the same stream again for the per-kernel split
        ks = {s_["name"]: s_ for s_ in rb.profile_read()}
        rb.profile(False)
        # first-round failures get the withheld parity bits (the plugin's verdict 2): recovered or not, and what that leaks
        leaked = np.array([q.Recon.leaked_bits(m) for m in msgs], dtype=np.int64)
        first_fail = int((~ok).sum())
        for i in np.nonzero(~ok)[0]:
            if msgs[i].n_punct == 0:
                continue
            m2, p2 = ra.encode_planned(aw[i], key_bits, msgs[i], 0)
            good2, fixed2, _, _, _ = rb.decode(bw[i], key_bits, float(qbers[i]), m2, p2)
            leaked[i] = q.Recon.leaked_bits(m2)
            if good2 and (fixed2 == aw[i]).all():
                ok[i] = True
        leak = int(leaked[ok].sum())
        kern_ms = sum(s_["total_ms"] for s_ in ks.values())
        hot = [ks[k_] for k_ in ("cn_update", "vn_update", "layer_update") if k_ in ks and ks[k_]["launches"]]
        layered3 = "layer_update" in ks and ks["layer_update"]["launches"] > 0
        hot_bytes = sum(s_["alg_bytes"] for s_ in hot)
        return dict(value=float(epochs_n - first_fail) * key_bits / dt3 / 1e6, unit="Mbit/s of sifted key, host buffers in and out (PCIe, CRC and packing included)",
                    fer=float(first_fail) / epochs_n, fer_after_second_round=float(1.0 - ok.mean()), undetected_errors=undetected,
                    leaked_fraction=float(leak) / max(1.0, float(ok.sum()) * key_bits),
                    timed_from_c_too="qcrypto-ldpc_amd/host/qldpc_stream.c (the same call with nothing but the C ABI in the timed region: 1 650 - 1 900 Mbit/s on its own synthetic stream)",
                    configured_efficiency=1.4, avg_iterations=float(iters.mean()), ms_total=dt3 * 1e3, ms_best=min(dts) * 1e3, alice_encode_ms=t_enc * 1e3,
                    epochs_per_rate={("%.1f" % ra.rates[k_[0]]): len(v) for k_, v in sorted(groups.items())},
                    # SURVEY 8d bytes of the check + variable passes over the wall time of the whole call (copies, staging, verification included)
                    wall_frac=hot_bytes / dt3 / 1e9 / HBM_PEAK_GBS,
                    # the rate groups run side by side on their own streams, so per-launch event times overlap and their sum says nothing about the device:
                    # this roofline prices the check + variable passes' SURVEY 8d bytes against the WALL time of the call (= wall_frac)
                    roofline=dict(bound="hbm", kernel=("qk_cn_layer sweeps" if layered3 else "qk_cn_flood + qk_vn_flood") + " of the session decoders (SPA, early exit, batches of <= %d blocks, rate groups side by side)" % batch,
                                  peak=HBM_PEAK_GBS, unit="GB/s", alg_bytes=hot_bytes, moved_bytes=sum(s_["moved_bytes"] for s_ in hot),
                                  achieved=hot_bytes / dt3 / 1e9, frac=hot_bytes / dt3 / 1e9 / HBM_PEAK_GBS, against="wall time of the decode_blocks call (copies, staging, verification included)",
                                  summed_kernel_ms=sum(s_["total_ms"] for s_ in hot), summed_all_kernels_ms=kern_ms,
                                  note="summed_*_ms add per-launch event times of kernels that overlap on the device: they exceed ms_total and are not a duration"),
                    workload="%d epochs x %d bits, QBER ~ U[0.5 %%, 6 %%] (seed 42), rate per epoch from {0.5, 0.7, 0.8, 0.9} (f = 1.4), mother code K = 57344 (%s) shortened + punctured per epoch, "
                             "%s SPA, one decode_blocks call for the stream, gap profile %d" % (epochs_n, key_bits, "PEG depth %d" % peg_depth if peg_depth else "seeded shuffle",
                                                                                                  "horizontal-layered" if layered3 else "flooding", gap_profile),
                    schedule_note="the sessions decode batches on the horizontal-layered schedule (qldpc_recon_cfg.schedule = auto): about half as many passes over the same bytes per pass as flooding, "
                                  "so wall_frac (section 8(d) bytes of the passes executed / wall time / 8 TB/s) is lower at a higher throughput; flooding_schedule is the same stream on round 2's schedule")

    cfg3 = cfg5 = ferd = None
    if rank == 0 and world == 1 and args.schedule == "flooding" and args.msg_dtype == "f32" and (N, K) == (65536, 52429):
        if not args.no_config5:
            f5s = [int(x) for x in args.config5_frames.split(",") if x]
            cfg5 = config5(f5s[0])              # batch 64 (one frame group: 6 667 waves per colour layer)
            for f5 in f5s[1:]:                  # the same code with enough frames to fill the chip
                big = config5(f5)
                # (early exit with several groups: finished groups leave the sweeps, so a launch's bytes are not the full batch's -- no fraction quoted)
                cfg5["at_%d_frames" % f5] = {k_: {kk: big[k_][kk] for kk in ("value", "unit", "fer", "avg_sweeps", "ms_per_step")} |
                                             ({"roofline_frac": big[k_]["roofline"]["frac"], "roofline_moved_frac": big[k_]["roofline"]["moved_frac"]} if (k_ == "fixed" or f5 <= 64) else {})
                                             for k_ in ("fixed", "early_exit")}
        if not args.no_config3:
            cfg3 = config3()      # the sessions' default: PEG-built mother codes (depth 2), gaps as calibrated for them (leak 0.29 of the key)
            keep = ("value", "fer", "fer_after_second_round", "leaked_fraction", "ms_total", "avg_iterations", "wall_frac", "epochs_per_rate", "workload")
            fast = config3(gap_profile=1)      # the same codes planned with round 2's gaps: fewer iterations, leak 0.305
            cfg3["peg_mothers_round2_gaps"] = {k_: fast[k_] for k_ in keep}
            shuf = config3(peg_depth=0)        # round 2's codes and gaps: the seeded socket shuffle, for comparison
            cfg3["seeded_shuffle_mothers"] = {k_: shuf[k_] for k_ in keep}
            flood = config3(schedule="flooding")      # the default's codes and gaps on the flooding schedule (the sessions' schedule until this round)
            cfg3["flooding_schedule"] = {k_: flood[k_] for k_ in keep}
        if not args.no_fer_deep:
            ferd = fer_deep()

    # ---- CPU baseline (rank 0, N = 1 only) -------------------------------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        rx_host = None

        def llr_fn(n):
            nonlocal rx_host
            n = min(n, F)
            bits = q.unpack_bits(rx[:n].cpu().numpy().view(np.uint32), N)
            m = np.float32(q.bsc_llr(args.qber))
            llr = np.where(bits == 1, -m, m).astype(np.float32)
            cwb = q.unpack_bits(cw[:n].cpu().numpy().view(np.uint32), N)
            llr[:, K:] = np.where(cwb[:, K:] == 1, -np.float32(q.CONFIRMED_BIT_LLR), np.float32(q.CONFIRMED_BIT_LLR))
            return llr

        cpu, ref, ref_llr = cpu_baseline(code, llr_fn, args.rule, args.alpha, args.n_ite, K)
        # the same sample through the GPU must give the same words (cheap cross-check, not timed)
        nref = ref["hard"].shape[0]
        got = q.unpack_bits(out_headline[:nref].cpu().numpy().view(np.uint32), N)
        cpu["gpu_matches_oracle_on_sample"] = bool((got == ref["hard"]).all())

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process, so this is the
    # committed rocprofv3 --pmc summary of the SAME command (profiles/), used only when the workload matches.
    traffic, traffic_src, vn_traffic = None, None, None
    if rank == 0:
        try:
            pmc_path = os.path.join("profiles", "r03_fixed50_pmc_hbm_traffic.json")
            pmc = json.load(open(os.path.join(ROOT, pmc_path)))
            w = pmc["workload"]
            fpl = args.frames_per_lane or 1
            if (w["frames"], w["frames_per_lane"], w["N"], w["E"]) == (F, fpl, N, code.E) and args.msg_dtype == "f32" and args.rule == "NMS":
                # the steady-state kernels of the fixed-50 leg, whatever further template arguments they have grown
                cnk = [k_ for k_ in pmc["kernels"] if k_.startswith("qk_cn_flood<%d, 20, 0, float, false" % fpl)]
                vnk = [k_ for k_ in pmc["kernels"] if k_.startswith("qk_vn_flood<%d," % fpl) and ", 1, float, true>" in k_]
                if len(cnk) == 1:
                    traffic = pmc["kernels"][cnk[0]]["hbm_bytes_corrected"]
                    traffic_src = pmc_path + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of: " + pmc["source"].split(" -- ", 1)[-1] + "; gfx950-corrected)"
                if vnk:
                    vn_traffic = sum(pmc["kernels"][k_]["hbm_bytes_corrected"] for k_ in vnk)      # one VN pass = its degree buckets' launches
        except Exception as ex:      # noqa: BLE001
            log("no PMC traffic file for this workload: %s" % ex)
        if traffic is None and (F, N, args.msg_dtype, args.rule, args.schedule) == (4096, 65536, "f32", "NMS", "flooding"):
            log("warning: roofline.traffic is null although the workload is the profiled one -- kernel names changed? re-run tools/pmc_summary.py")

    # what this device copies at (read once + written once, non-temporal) in the decoder's access shape: the measured ceiling beside the data sheet's
    probe = None
    if rank == 0:
        probe = {"rows_256B_GBs": q.copy_probe(4 << 30, 6, False, local_rank), "wide_16B_per_lane_GBs": q.copy_probe(4 << 30, 6, True, local_rank),
                 "what": "hand-written copy kernels of libqldpc (qldpc_copy_probe), 4 GiB read + 4 GiB written per launch, bytes read + bytes written per second"}
        probe["frac_of_copy"] = achieved / max(probe["rows_256B_GBs"], probe["wide_16B_per_lane_GBs"])
    if rank == 0:
        line = {
            "metric": "reconciled_key_Mbit_s",
            "value": value,
            "unit": "Mbit/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "f16": "f32 (binary16 message storage)", "i8": "int8"}[args.msg_dtype],
            "data": "synthetic",
            "fer": fer,
            # the decode rules of the min-sum family (the headline's NMS included) hold no AFF3CT vector in the reference: bit-exact means
            # against the CPU oracle (oracle/), which is pinned to AFF3CT by the reference's flooding-SPA known-answer vector only
            "parity_note": "NMS / OMS / MS: parity unpinned against AFF3CT (no reference vector); bit-exact means against the oracle",
            "sifted_key_Mbit_s": value * N / K,        # the same frames counted with all N VNs
            "iterations_executed": fixed_iters,
            "frames_per_step": int(n_all),
            "config": {
                "workload": "rate-%.1f N=%d IRA LDPC (K=%d, M=%d, E=%d), flooding %s(%.2f), %d iterations fixed, QBER %.1f %%, "
                            "%d frames per GPU (%s), packed bits resident in HBM" % (K / N, N, K, code.M, code.E, args.rule, args.alpha, args.n_ite,
                                                                                     args.qber * 100, F, "BASELINE configs[1]" if (world == 1 and F == 4096) else
                                                                                     ("BASELINE configs[3]: 262144 / 8 frames per GPU" if F == 32768 else "custom batch")),
                "frames_per_gpu": F, "qber": args.qber, "n_ite": args.n_ite, "rule": args.rule, "alpha": args.alpha,
                "parallelism": "frame-sharded x%d, RCCL gather of decoded blocks only" % world,
            },
            "roofline": {
                "bound": "hbm", "kernel": "qk_cn_flood (check-node update)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "alg_bytes_per_launch": cn_bytes, "avg_launch_ms": cn_avg_s * 1e3, "launches": cn["launches"], "copy_probe": probe,
                # "achieved" prices the SURVEY 8d algorithmic bytes (an N x 4 B LLR array per pass); "moved" the bytes the pass really
                # fetches with coded LLRs (N / 8 B of received-bit ballots per frame)
                "vn_update": {"achieved": vn_achieved, "frac": vn_achieved / HBM_PEAK_GBS, "moved": vn_moved, "moved_frac": vn_moved / HBM_PEAK_GBS,
                              "alg_bytes_per_pass": vn["alg_bytes"] / vn["launches"], "moved_bytes_per_pass": vn["moved_bytes"] / vn["launches"],
                              "avg_pass_ms": vn["total_ms"] / vn["launches"], "passes": vn["launches"], "traffic": vn_traffic},
                "whole_step": {"alg_bytes": step_bytes, "achieved": step_bytes / (dt / args.steps) / 1e9,
                               "frac": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                               "moved_bytes": step_moved, "moved": step_moved / (dt / args.steps) / 1e9,
                               "moved_frac": step_moved / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
            },
            "cpu_baseline": cpu,
            "early_exit": early,
            "layered_schedule": lay2,
            "fp16_messages": fp16,
            "int8_messages": int8,
            "spa_rule": spa,
            "fer_deep": ferd,
            "config3_multirate_stream": cfg3,
            "config5_layered_1e6": cfg5,
            "reference_context": {"aff3ct_spa_1thread_debug_Mbit_s": 0.241, "cascade_daemon_Mbit_s": 0.3},
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
