#!/usr/bin/env python3
"""bench.py — SeedDelta FFN decode throughput on MI355X (BASELINE.json metric, config 3).

The unit of work is one token (T=1) through the SeedDelta op of one FFN tensor — by default Mistral-7B
``blk.23.ffn_down`` (14336 -> 4096), block32 K=1536 (nb=48), base L=4096 B=4, row_scale: 13,189,120
algorithmic bytes per token. Inputs are resident in HBM when the timed region starts. To
keep the 256 MB Infinity Cache from serving the payload, the loop rotates over ``--payloads`` (default 32
=> 420 MB) distinct payloads and activations, and a "step" is ONE ROTATION: ``--payloads`` tokens, one
launch each (``config.tokens_per_step``). ``--steps K --warmup W`` are honoured exactly as given.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU, either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
environment) or, when ``--gpus N`` is given without them, spawned by this script as N fresh child processes
before anything touches a GPU. The FFN path does not shard — one model fits one GPU — so ranks run
independent replicas ("replicas only", weak scaling), no data-path collective; the reported value is all
ranks' tokens / max-over-ranks time. The scan leg shards (strong scaling, one all_gather).

Prints ONE JSON line on rank 0 with the throughput, a ``roofline`` object for the dominant kernel
(payload-streaming residual kernel; duration from HIP events recorded by the library on the launch
stream) and a ``cpu_baseline`` object (the reference's own translation unit from oracle/_ref when that
prebuilt .so is present, else the oracle port; rank 0, N=1 only; bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

CLOCK_SETTLE_MS = 60.0        # untimed launches before the W warm-up steps of the headline leg (clock ramp)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 achievable)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256, help="timed steps; one step = one rotation over --payloads distinct payloads")
    ap.add_argument("--warmup", type=int, default=16, help="untimed warm-up steps (rotations)")
    ap.add_argument("--workload", default="cfg3_mistral_down_block32_k1536")
    ap.add_argument("--payloads", type=int, default=32, help="distinct payloads rotated per loop (cache honesty)")
    ap.add_argument("--tokens", type=int, default=1, help="tokens per step (1 = decode)")
    ap.add_argument("--launch", choices=["graph", "eager"], default="eager",
                    help="eager: one fused launch per step; graph: the same launches recorded once into a "
                         "hipGraph and replayed (captured fused launches use the reserved hand-off tag)")
    ap.add_argument("--no-fuse", action="store_true", help="force the two-launch path (A/B)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-scan", action="store_true", help="skip the scan (candidates/s) leg")
    ap.add_argument("--no-ffn", action="store_true", help="skip the gated-FFN-block leg")
    ap.add_argument("--no-stack", action="store_true", help="skip the persistent FFN-stack leg")
    ap.add_argument("--stack-layers", type=int, default=32)
    ap.add_argument("--stack-reps", type=int, default=40)
    ap.add_argument("--no-prefill", action="store_true", help="skip the prefill (T=512) leg")
    ap.add_argument("--prefill-reps", type=int, default=20)
    ap.add_argument("--ffn-layers", type=int, default=16, help="distinct synthetic Mistral-7B layers rotated in the FFN-block leg")
    ap.add_argument("--scan-layers", type=int, default=32, help="synthetic Mistral-7B layers in the scan job")
    ap.add_argument("--scan-payload-sample", type=int, default=144,
                    help="candidates re-run with every output row selected (payload emission), reported beside the scan rate")
    ap.add_argument("--scan-threads", type=int, default=16, help="host threads preparing candidates per rank")
    ap.add_argument("--host-io", action="store_true",
                    help="also time the PCIe-inclusive boundary (host x in, host y out, one sync per call: what "
                         "the shim pays per graph node); reported as a separate field, never as value")
    return ap.parse_args()


def cpu_baseline(workload: str, tokens: int, budget_s: float):
    """Reference CPU path on this host's cores, bounded sample. kind = 'reference' when the prebuilt
    oracle/_ref/libsdref.so (the reference's own TU) is present, else 'port' (oracle/sd_oracle.c)."""
    import oracle_py as O
    from llama_cpp_amd.payload import config_payload, make_x
    p = config_payload(workload, seed=1234)
    x = make_x(p.n_in, tokens, seed=1)
    use_ref = O.ref_available()
    if not use_ref:
        O.oracle_lib()
    cores = min(16, os.cpu_count() or 1)       # the GPU box gives a 16-core share per GPU
    y = np.empty((tokens, p.n_out), dtype=np.float32)
    O.forward(p, x, nth=cores, use_ref=use_ref, out=y)      # warm
    n, t0 = 0, time.perf_counter()
    while True:
        O.forward(p, x, nth=cores, use_ref=use_ref, out=y)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 2000:
            break
    note = ""
    if tokens == 1 and not p.is_tall:
        note = "; wide base path splits tokens across threads, so at T=1 one thread does all the work (src/llama-seeddelta.cpp:1129-1131)"
    return {
        "value": n * tokens / el, "unit": "tok/s", "cores": cores,
        "kind": "reference" if use_ref else "port",
        "sample": f"{n} evaluations of {workload} at T={tokens} in {el:.1f}s, {cores} threads{note}",
    }


# K schedules of policies/policy.mistral7b.base.json (gate/up 128..768, down 256..1536): two points each
# K schedules of the reference's Mistral policy (tools/seeddelta-build/policies/policy.mistral7b.base.json:15-16,31-32,46-47)
SCAN_K = {"ffn_gate": (128, 256, 384, 512, 640, 768), "ffn_up": (128, 256, 384, 512, 640, 768),
          "ffn_down": (256, 512, 768, 1024, 1280, 1536)}
MISTRAL = {"n_embd": 4096, "n_ff": 14336}


_SCAN_BASE = {}


def scan_dims(layer, kind):
    return (MISTRAL["n_ff"], MISTRAL["n_embd"]) if kind == "ffn_down" else (MISTRAL["n_embd"], MISTRAL["n_ff"])


def scan_model(layer, kind):
    """Synthetic Mistral-7B FFN tensor in Q8_0: 0.02*N(0,1) with an XOR-circulant component, one
    draw per tensor kind; layers differ by a rotation of the output rows (cheap to make, distinct
    contents, same statistics)."""
    from llama_cpp_amd.dense import make_dense
    from llama_cpp_amd.scan import KIND_ID
    if kind not in _SCAN_BASE:
        n_in, n_out = scan_dims(layer, kind)
        _SCAN_BASE[kind] = make_dense(n_in, n_out, "q8_0", seed=1234 + KIND_ID[kind], structure=0.3)
    return np.roll(_SCAN_BASE[kind], 97 * layer, axis=0)


def scan_variants(metrics_only=True):
    """metrics_only: the layer-scan / autogate use — a report entry per candidate, the builder's output
    GGUF discarded (scripts/seeddelta-layer-scan.py), so only the rows the metric passes sample are
    selected; the entry is bit-identical to the full selection (tests/test_gpu_scan.py)."""
    from llama_cpp_amd.scan import ScanParams
    return {0: ScanParams(scheme="coo", base=False, row_scale=False, eval_cols=64, eval_x=32, metrics_only=metrics_only),   # proxy on
            1: ScanParams(scheme="coo", base=True, row_scale=True, eval_cols=64, eval_x=32, base_perm_trials=4,
                          metrics_only=metrics_only)}


def scan_host_threads(args, world):
    """Host threads preparing candidates per rank: the flag, capped at this process's share of the host's cores."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(2, min(args.scan_threads, cores // max(1, world)))


def run_scan(args, rank, world, dev):
    """The sweep of config 5, scaled to --scan-layers layers: (layer, tensor, K) candidates in two
    variants (COO without base + FFN proxy; COO with base + row scale), dealt to ranks as whole (layer, tensor)
    groups, evaluated on each rank's GPU, one all_gather of the records (RCCL). Strong scaling: the job is fixed.
    Three timed passes over the SAME candidates:
      upload_inclusive  each rank materialises and uploads only its own layers' weights inside the clock, then runs
                        the sweep with every output row selected (payload emission: the reference's only form);
      value             the same payload-emission sweep with the weights already resident;
      metrics_only      report entries only (rows the metric passes sample), weights resident."""
    import torch
    from llama_cpp_amd import scan as SC
    from llama_cpp_amd import seeddelta as _sd
    one_dev = os.environ.get("SDH_BENCH_ONE_DEVICE") == "1"
    threads = scan_host_threads(args, world)
    lite, full = scan_variants(True), scan_variants(False)
    cands = []
    for v in lite:
        for c in SC.enumerate_candidates(range(args.scan_layers), SC.KINDS, SCAN_K):
            c.variant = v
            cands.append(c)
    cost = lambda c: (2.0 if c.variant else 1.0) + c.K / 1536.0       # noqa: E731
    # all trials of one (layer, tensor) stay on one rank: its dense weight is uploaded once;
    # the same deal the world-size-2 gloo test exercises (tests/test_scan_host.py)
    mine, counts = SC.shard_groups(cands, rank, world, cost)
    ev = SC.Evaluator(scan_model, lite[0], max_resident=3 * max(1, args.scan_layers) + 3, dims_of=scan_dims)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    def max_over_ranks(el):
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([el], dtype=torch.float64, device="cpu" if one_dev else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return el

    def sweep(variants, upload_first=False):
        """barrier, clock, [this rank's weights], the sweep, the gather, barrier, clock; max over ranks."""
        _sd.load_library().sdh_scan_memo_clear()          # every timed sweep makes every distinct draw itself
        barrier()
        t0 = time.perf_counter()
        if upload_first:
            ev.preload([(c.layer, k) for c in mine for k in (SC.KINDS if c.kind != "ffn_down" else (c.kind,))], threads=threads)
        res = ev.run_many(mine, threads=threads, params_of=lambda c: variants[c.variant])
        local = np.stack([SC.to_record(c, r) for c, r in zip(mine, res)]) if mine else np.zeros((0, SC.RECORD_LEN))
        allr = SC.gather_records(local, counts, device=None if one_dev else dev)
        barrier()
        el = max_over_ranks(time.perf_counter() - t0)
        assert len(allr) == len(cands)
        return el

    # warm-up on the worker pool the timed runs use, on a layer outside the sweep: every worker creates its scan
    # handle (stream, events, device buffers at the largest K), as in a scan that has been running for a while
    wl = args.scan_layers + rank
    warm = []
    for v in lite:
        for c in SC.enumerate_candidates([wl], SC.KINDS, SCAN_K):
            c.variant = v
            warm.append(c)
    for _ in range(max(1, -(-threads // 6))):
        ev.run_many(warm, threads=threads, params_of=lambda c: full[c.variant])
        ev.run_many(warm, threads=threads, params_of=lambda c: lite[c.variant])
    for key in [k for k in list(ev._cache) if k[0] == wl]:             # the warm-up layer leaves the device
        ev._cache.pop(key).close()
        ev._order.remove(key)

    el_up = sweep(full, upload_first=True)
    dense_bytes = sum(w.stored_bytes for w in ev._cache.values())
    el_full = sweep(full)
    el_lite = sweep(lite)
    ev.close()
    # bytes actually read: HBM traffic per candidate from the committed counter passes (tools/gpu/gpu_scan_pmc.sh:
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over the sweep, FETCH_SIZE doubled as the guide prescribes)
    traffic, tsrc = None, None
    for name in ("r03_scan_traffic.json", "r02_scan_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic, tsrc = json.load(fh), "profiles/" + name
            break
    # algorithmic bytes per candidate, SURVEY §8d: the quantised tensor read once per K schedule and the idx / val rows
    # stored once per schedule at its largest K (every smaller K is a prefix of the stored row: work shared across a K
    # grid is never counted twice), int32 / fp32 staging, + the candidate's own row scales
    q8_row = lambda n_in: n_in // 32 * 34                                                    # noqa: E731
    def alg_full(c):
        n_in, n_out = scan_dims(c.layer, c.kind)
        ks = SCAN_K[c.kind]
        return (n_out * q8_row(n_in) + n_out * 8 * max(ks)) / len(ks) + n_out * 4
    # metrics-only: per K schedule the fit's sampled row blocks once + the rows the metric passes sample
    def alg_lite(c):
        n_in, n_out = scan_dims(c.layer, c.kind)
        L = 1 << (min(n_in, n_out) - 1).bit_length()
        fit = (min(2048, L) * -(-max(n_in, n_out) // L) * q8_row(min(n_in, L))) if c.variant else 0
        rows = 64 * (2 + 2) * q8_row(n_in)
        return (fit + rows) / len(SCAN_K[c.kind])
    a_full, a_lite = sum(map(alg_full, cands)), sum(map(alg_lite, cands))
    peak = HBM_PEAK_GBS * world
    roof = lambda a, el: {"bound": "hbm", "achieved": a / el / 1e9, "peak": peak, "unit": "GB/s", "frac": a / el / 1e9 / peak}  # noqa: E731
    r_full = roof(a_full, el_full)
    per_c = traffic.get("payload", {}).get("hbm_bytes_per_candidate") if traffic else None
    r_full.update({"traffic": per_c, "traffic_source": tsrc if per_c else None,
                   "algorithmic_bytes_per_candidate": a_full / len(cands),
                   "note": "achieved = (quantised tensor + idx/val rows at the largest K, once per K schedule, + row scales per "
                           "candidate) / sweep time; traffic = measured HBM bytes per candidate from the committed counter passes"})
    r_lite = roof(a_lite, el_lite)
    r_lite.update({"traffic": traffic["sweep"].get("hbm_bytes_per_candidate") if traffic else None,
                   "traffic_source": tsrc, "algorithmic_bytes_per_candidate": a_lite / len(cands)})
    return {"metric": "scan candidates/s", "value": len(cands) / el_full, "unit": "candidates/s", "n_gpus": world,
            "scaling": "strong", "candidates": len(cands), "seconds": el_full, "roofline": r_full,
            "upload_inclusive": {"value": len(cands) / el_up, "unit": "candidates/s", "seconds": el_up,
                                 "dense_bytes_uploaded_per_rank": dense_bytes,
                                 "note": "each rank materialises and uploads only its own layers' Q8_0 weights inside "
                                         "the clock (a real scan's dominant cost; it shards with the candidates)"},
            "metrics_only": {"value": len(cands) / el_lite, "unit": "candidates/s", "seconds": el_lite, "roofline": r_lite,
                             "note": "report entries only: rows the metric passes sample (layer-scan / autogate use)"},
            "config": {"workload": f"cfg5: {args.scan_layers} synthetic Mistral-7B layers x gate/up/down x "
                                   f"K {SCAN_K} x {{coo no-base + proxy, coo base + row_scale}}, eval_cols=64 eval_x=32, Q8_0; "
                                   "every output row selected per K (payload emission)",
                       "host_threads_per_rank": threads, "dense_bytes_resident_per_rank": dense_bytes,
                       "collective": "one all_gather of %d-double records" % SC.RECORD_LEN}}


def timed_on_stream(fn, reps, stream, seeddelta, warm=None):
    """Run fn() reps times on `stream` (an explicit, non-default torch stream whose handle every library call
    receives) and time the region twice: HIP events recorded by the library ON THAT STREAM (sdh_profile_span_*)
    and the synchronised host clock. Returns (event_ms, host_ms). The two must agree: events on another stream
    would time host enqueue, not the kernels (the flaw of the round-1 FFN figure)."""
    import torch
    for _ in range(max(2, reps // 8) if warm is None else warm):
        fn()
    stream.synchronize()
    torch.cuda.synchronize()
    seeddelta.profile_span_begin(stream.cuda_stream)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ev_ms, _ = seeddelta.profile_span_end(stream.cuda_stream)       # waits for the closing event
    stream.synchronize()
    host_ms = (time.perf_counter() - t0) * 1e3
    return ev_ms, host_ms


def run_ffn(args, dev):
    """Gated FFN block at T=1: one Mistral-7B layer = gate, up (4096->14336) and down
    (14336->4096), COO K=128 with base, SiLU. Compares sdh_ffn_forward (gate|up in one launch, h kept on
    the device, then down) with the same three nodes + activation issued one by one. Layers rotate so
    that the payloads (17 MB per layer) do not stay in the 256 MB Infinity Cache. Everything is launched and
    timed on one explicit stream."""
    import torch
    from llama_cpp_amd import seeddelta
    from llama_cpp_amd.payload import make_payload
    NL = max(1, args.ffn_layers)
    n_embd, n_ff, K = 4096, 14336, 128
    layers, alg = [], 0
    for i in range(NL):
        ps = [make_payload(n_embd, n_ff, scheme="coo", K=K, base=True, seed=7000 + 3 * i),
              make_payload(n_embd, n_ff, scheme="coo", K=K, base=True, seed=7001 + 3 * i),
              make_payload(n_ff, n_embd, scheme="coo", K=K, base=True, seed=7002 + 3 * i)]
        alg = sum(p.algorithmic_bytes(1) for p in ps)
        layers.append([seeddelta.SeedDeltaWeight(p) for p in ps])
        del ps
    x = torch.randn((1, n_embd), dtype=torch.float32, device=dev)
    y = torch.empty((1, n_embd), dtype=torch.float32, device=dev)
    g = torch.empty((1, n_ff), dtype=torch.float32, device=dev)
    u = torch.empty((1, n_ff), dtype=torch.float32, device=dev)
    h = torch.empty((1, n_ff), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    st = torch.cuda.Stream(device=dev)
    sp = st.cuda_stream
    assert sp != 0

    def block():
        for wg, wu, wd in layers:
            seeddelta.ffn_forward_device(wg, wu, wd, seeddelta.ACT_SILU, x.data_ptr(), n_embd, 1, y.data_ptr(), n_embd, 0, sp)

    def nodes():
        for wg, wu, wd in layers:
            wg.forward_device(x.data_ptr(), n_embd, 1, g.data_ptr(), n_ff, 0, sp)
            wu.forward_device(x.data_ptr(), n_embd, 1, u.data_ptr(), n_ff, 0, sp)
            seeddelta.glu_device(seeddelta.ACT_SILU, g.data_ptr(), u.data_ptr(), h.data_ptr(), n_ff, sp)
            wd.forward_device(h.data_ptr(), n_ff, 1, y.data_ptr(), n_embd, 0, sp)

    reps = max(8, 2048 // NL)
    ev_b, host_b = timed_on_stream(block, reps, st, seeddelta)
    fused = seeddelta.ffn_last_fused()
    ev_n, host_n = timed_on_stream(nodes, reps, st, seeddelta)
    us_block, us_nodes = ev_b * 1e3 / (reps * NL), ev_n * 1e3 / (reps * NL)
    # small decode batches (speculative / batched decode) and the start of the tiled path, same layers
    by_tokens = {}
    for T in (2, 4, 8, 16):
        xT = torch.randn((T, n_embd), dtype=torch.float32, device=dev)
        yT = torch.empty((T, n_embd), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()

        def block_t():
            for wg, wu, wd in layers:
                seeddelta.ffn_forward_device(wg, wu, wd, seeddelta.ACT_SILU, xT.data_ptr(), n_embd, T, yT.data_ptr(), n_embd, 0, sp)

        ev_t, _ = timed_on_stream(block_t, max(4, reps // 8), st, seeddelta)
        by_tokens[str(T)] = ev_t * 1e3 / (max(4, reps // 8) * NL)
        del xT, yT
    # one ubatch of prefill through the same layers: the fp32 node forms (SDH_FWD_NO_MFMA), what sdh_ffn_forward chooses by itself
    # (from 320 tokens on: every node as a folded f16 contraction), and that form on request (SDH_FWD_MFMA_FOLD: gate, up with
    # the activation in its epilogue, down)
    prefill_layer = {}
    if not args.no_prefill:
        T = 512
        xT = torch.randn((T, n_embd), dtype=torch.float32, device=dev)
        yT = torch.empty((T, n_embd), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        use = layers[:min(NL, 4)]
        for name, fl in (("fp32", seeddelta.FWD_NO_MFMA), ("default", 0), ("mfma_fold", seeddelta.FWD_MFMA_FOLD)):
            def block_p():
                for wg, wu, wd in use:
                    seeddelta.ffn_forward_device(wg, wu, wd, seeddelta.ACT_SILU, xT.data_ptr(), n_embd, T, yT.data_ptr(), n_embd, fl, sp)
            ev_p, _ = timed_on_stream(block_p, 24, st, seeddelta, warm=30 if name == "fp32" else 24)    # (the first variant also brings the clocks up;
            prefill_layer[name] = ev_p / (24 * len(use))                                                 # 24 + 24 blocks of four layers: ~25 ms per variant, settled clocks)
        prefill_layer = {"tokens": T, "ms_per_layer": prefill_layer,
                         "tokens_per_s_fold": T / prefill_layer["mfma_fold"] * 1e3,
                         "mfma_frac_fold": 2.0 * 3 * n_embd * n_ff * T / (prefill_layer["mfma_fold"] * 1e-3) / (MFMA_F16_PEAK_TFLOPS * 1e12)}
        del xT, yT
    for ws in layers:
        for w in ws:
            w.close()
    out = {"metric": "gated FFN block decode, layers/s (Mistral-7B layer, COO K=128 + base, SiLU, T=1)",
           "value": 1e6 / us_block, "unit": "layers/s", "us_per_layer": us_block,
           "us_per_layer_host_clock": host_b * 1e3 / (reps * NL),
           "us_per_layer_node_by_node": us_nodes, "us_per_layer_node_by_node_host_clock": host_n * 1e3 / (reps * NL),
           "speedup_vs_node_by_node": us_nodes / us_block,
           "gate_up_single_launch": bool(fused), "distinct_layers": NL, "us_per_layer_by_tokens": by_tokens,
           "prefill_layer": prefill_layer,
           "algorithmic_bytes_per_layer": alg,
           "timing": "HIP events recorded by the library on the explicit launch stream; host clock (synchronised) beside it",
           "roofline": {"bound": "hbm", "achieved": alg / us_block / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / us_block / 1e3 / HBM_PEAK_GBS, "traffic": None,
                        "note": "two launches per layer (gate|up, down): launch-latency bound like the single-tensor decode"}}
    # the event span can only be shorter than the host clock by the enqueue lead; a large gap means mis-timing
    out["timing_consistent"] = bool(us_block <= out["us_per_layer_host_clock"] * 1.02 and us_block >= 0.80 * out["us_per_layer_host_clock"])
    return out


def run_ffn_stack(args, dev):
    """FFN stack leg: 32 distinct synthetic Mistral-7B layers (gate, up 4096->14336, down
    14336->4096, COO K=128 + base, SiLU), x_{l+1} = FFN_l(x_l), evaluated by ONE persistent launch per token
    (sdh_ffn_stack_forward). 32 layers x 17.46 MB = 559 MB of payload per token: more than twice the 256 MB
    Infinity Cache, so every layer streams from HBM on every call. Launched and timed on one explicit stream."""
    import torch
    from llama_cpp_amd import seeddelta
    from llama_cpp_amd.payload import make_payload
    NL = max(1, args.stack_layers)
    n_embd, n_ff, K = 4096, 14336, 128
    layers, alg = [], 0
    # A chain of random layers has no norm between its blocks (a model has one), so each layer's down row_scale is
    # rescaled by one scalar that brings |x_{l+1}|_rms back to 1: activations stay O(1) through all layers (a chain
    # that overflows would measure the non-finite counter's atomics, one that decays would measure zeros).
    import numpy as _np
    xh = _np.random.default_rng(77).standard_normal((1, n_embd)).astype(_np.float32)
    x0 = xh.copy()
    for i in range(NL):
        pg = make_payload(n_embd, n_ff, scheme="coo", K=K, base=True, seed=9000 + 3 * i)
        pu = make_payload(n_embd, n_ff, scheme="coo", K=K, base=True, seed=9001 + 3 * i)
        pd = make_payload(n_ff, n_embd, scheme="coo", K=K, base=True, seed=9002 + 3 * i)
        wg, wu = seeddelta.SeedDeltaWeight(pg), seeddelta.SeedDeltaWeight(pu)
        h = seeddelta.ffn_gate_up_host(wg, wu, xh, seeddelta.ACT_SILU)
        wd0 = seeddelta.SeedDeltaWeight(pd)
        # (calibrated with the STACK's arithmetic: the stack kernel is bit-identical to the fused launches, SDH_FWD_NO_DIRECT. A chain of
        # 32 random layers without norms is chaotic — calibrated along the direct launch's trajectory, whose low bits differ, the
        # stack's own trajectory leaves it within ~15 layers and overflows: 36,864 non-finite outputs and 17.6 instead of 14.6 us per layer)
        _fl = seeddelta.FWD_NO_DIRECT
        y0 = wd0.forward_host(h, _fl)
        wd0.close()
        c = 1.0 / max(float(_np.sqrt(_np.mean(y0.astype(_np.float64) ** 2))), 1e-20)
        pd.row_scale = (pd.row_scale.astype(_np.float32) * _np.float32(c)).astype(_np.float16)
        wd = seeddelta.SeedDeltaWeight(pd)
        xh = wd.forward_host(h, _fl)
        alg = sum(p.algorithmic_bytes(1) for p in (pg, pu, pd))
        layers.append((wg, wu, wd))
        del pg, pu, pd
    stack = seeddelta.FfnStack(layers)
    x = torch.from_numpy(x0).to(dev)
    y = torch.empty((1, n_embd), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    st = torch.cuda.Stream(device=dev)
    sp = st.cuda_stream

    def call():
        stack.forward_device(seeddelta.ACT_SILU, x.data_ptr(), y.data_ptr(), 0, sp)

    reps = max(4, args.stack_reps)
    ev_ms, host_ms = timed_on_stream(call, reps, st, seeddelta, warm=3)
    nf, timeouts = stack.status()
    y_rms = float(torch.sqrt(torch.mean(y.double() ** 2)).item())
    us_layer = ev_ms * 1e3 / (reps * NL)
    # the same layers one sdh_ffn_forward per layer (two launches per layer), for the speed-up
    a, b = x.clone(), torch.empty_like(x)

    def per_layer():
        for wg, wu, wd in layers:
            seeddelta.ffn_forward_device(wg, wu, wd, seeddelta.ACT_SILU, a.data_ptr(), n_embd, 1, b.data_ptr(), n_embd, 0, sp)

    ev2, _ = timed_on_stream(per_layer, max(2, reps // 4), st, seeddelta, warm=1)
    us_layer_launches = ev2 * 1e3 / (max(2, reps // 4) * NL)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_ffn_stack_mistral_coo_k128.json")
    if os.path.exists(tpath):
        with open(tpath) as fh:
            traffic = json.load(fh).get("hbm_bytes_per_layer")
    stack.close()
    for t in layers:
        for w in t:
            w.close()
    return {"metric": "FFN stack decode, layers/s (Mistral-7B layers, COO K=128 + base, SiLU, T=1, one persistent launch per token)",
            "value": 1e6 / us_layer, "unit": "layers/s", "us_per_layer": us_layer,
            "us_per_layer_host_clock": host_ms * 1e3 / (reps * NL), "us_per_token_%d_layers" % NL: us_layer * NL,
            "us_per_layer_two_launches_per_layer": us_layer_launches, "speedup_vs_launches": us_layer_launches / us_layer,
            "layers": NL, "calls_timed": reps, "algorithmic_bytes_per_layer": alg, "nonfinite": nf, "handoff_timeouts": timeouts, "output_rms": y_rms,
            "timing": "HIP events recorded by the library on the explicit launch stream / (calls x layers); host clock beside it",
            "roofline": {"bound": "hbm", "achieved": alg / us_layer / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / us_layer / 1e3 / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "ffn_stack_kernel (one launch walks all layers)"}}


PREFILL_WORKLOADS = ("cfg4_mistral_gate_coo_k128", "cfg4_mistral_gate_block16_k128",
                     "cfg4_mistral_down_coo_k128", "cfg4_mistral_down_block16_k128",
                     "cfg3_mistral_down_block32_k1536")
MFMA_F16_PEAK_TFLOPS = 2500.0      # dense f16/bf16 MFMA (MI355X_MICROARCH.md); never the 2:1-sparsity figure
VALU_F32_PEAK_TFLOPS = 157.3


def run_prefill(args, dev):
    """Prefill leg (config 4, and config 3's tensor): Mistral-7B gate / down, K=128, COO vs block16, and the down
    tensor at block 32 / K=1536 (long rows: LDS rows kernel), base on, one ubatch of T=512 tokens per call. Per workload: the default path (FWHT base + sparse x dense residual) and the W0-on-MFMA
    base (SDH_FWD_W0_MFMA), each timed on one explicit stream with the library's events. Rooflines: HBM for the
    algorithmic bytes (payload as stored + 4*T*(n_in+n_out)); fp32 VALU for the residual's
    multiply-adds (2*K*n_out*T); dense f16 MFMA for the W0 contraction (2*n_in*n_out*T)."""
    import torch
    from llama_cpp_amd import seeddelta
    from llama_cpp_amd.payload import config_payload
    T = 512
    st = torch.cuda.Stream(device=dev)
    sp = st.cuda_stream
    legs = {}
    for name in PREFILL_WORKLOADS:
        p = config_payload(name)
        w = seeddelta.SeedDeltaWeight(p)
        alg = p.algorithmic_bytes(T)
        K = p.d_idx.shape[1] if p.d_idx is not None else p.b_idx.shape[1] * p.b_val.shape[2]
        n_in, n_out = p.n_in, p.n_out
        del p
        g = torch.Generator(device="cpu").manual_seed(5)
        x = torch.randn((T, n_in), generator=g, dtype=torch.float32).to(dev)
        y = torch.empty((T, n_out), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        leg = {"algorithmic_bytes": alg, "tokens": T}
        # rows_lds: SDH_FWD_NO_SPMM, the residual through the LDS rows kernels (the fast path's arithmetic: within the
        # tolerance of tests/test_gpu_forward.py, each token bit-identical to its one-token result) instead of the tiled
        # SpMM, whose rows are bit-identical to the CPU op; the gated FFN entry points use it for COO payloads
        # mfma_fold: SDH_FWD_MFMA_FOLD, the whole op as one dense contraction (residual added to the W0 tile in LDS)
        # default: what sdh_forward chooses by itself — at 512 tokens the folded contraction for every Mistral payload class;
        # fp32: SDH_FWD_NO_MFMA, the default's choice among the fp32 forms (bit-exact tiled SpMM, or the rows kernel for long rows)
        for _ in range(400):                                  # ~35 ms of load: the chip reaches its steady clocks only after 20+ ms (the first variant read 10 % slow behind 40 calls)
            w.forward_device(x.data_ptr(), n_in, T, y.data_ptr(), n_out, flags=0, stream=sp)
        st.synchronize()
        for tag, flags in (("default", 0), ("fp32", seeddelta.FWD_NO_MFMA), ("w0_mfma", seeddelta.FWD_W0_MFMA),
                           ("rows_lds", seeddelta.FWD_NO_SPMM), ("mfma_fold", seeddelta.FWD_MFMA_FOLD)):
            fn = lambda: w.forward_device(x.data_ptr(), n_in, T, y.data_ptr(), n_out, flags=flags, stream=sp)   # noqa: E731
            ev_ms, host_ms = timed_on_stream(fn, args.prefill_reps, st, seeddelta, warm=10)      # (the first variant after an upload also warms the clocks)
            ms = ev_ms / args.prefill_reps
            d = {"ms": ms, "ms_host_clock": host_ms / args.prefill_reps, "tok_per_s": T / ms * 1e3,
                 "hbm": {"achieved": alg / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / ms / 1e6 / HBM_PEAK_GBS},
                 "residual_valu": {"achieved": 2.0 * K * n_out * T / ms / 1e9, "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": 2.0 * K * n_out * T / ms / 1e9 / VALU_F32_PEAK_TFLOPS}}
            if tag in ("w0_mfma", "mfma_fold"):
                fl = 2.0 * n_in * n_out * T
                d["mfma"] = {"achieved": fl / ms / 1e9, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": fl / ms / 1e9 / MFMA_F16_PEAK_TFLOPS,
                             "note": "whole call (W0 contraction + residual epilogue) against the dense f16 MFMA peak"}
            leg[tag] = d
        legs[name] = leg
        w.close()
        del x, y
        torch.cuda.empty_cache()
    return {"metric": "prefill ubatch (T=512) per tensor, Mistral-7B (config 4: K=128; config 3: block 32, K=1536)", "workloads": legs,
            "timing": "HIP events recorded by the library on the explicit launch stream / reps",
            "reference_cpu_seconds": {"cfg4_mistral_gate_coo_k128": 1.21, "cfg4_mistral_gate_block16_k128": 0.83,
                                      "cfg4_mistral_down_coo_k128": 0.35, "cfg4_mistral_down_block16_k128": 0.24,
                                      "source": "SURVEY.md §8a, compiled reference, 8 threads, build container"}}


def scan_cpu_baseline(budget_s):
    """The oracle's restatement of the builder on one gate tensor, in the unit the GPU's payload-emission figure
    uses: K schedules of a tensor — the base fitted once (sd_build.cpp:444-451), then per K a trial that selects
    EVERY output row (the reference has no other form) and takes the weight- and activation-space metrics
    (:843-913). kind 'port': sd_build.cpp's run_trial cannot be compiled here."""
    import scan_py as S
    W = scan_model(0, "ffn_gate")
    cores = min(16, os.cpu_count() or 1)
    n, t0 = 0, time.perf_counter()
    while True:
        o = S.Engine("oracle", 1234)
        fit = o.fit_base(W, 2048, 4)
        for K in SCAN_K["ffn_gate"]:
            idx, val, rs = S.select_rows(W, fit, 0, K, 32, True, nth=cores)
            o.eval_w(W, fit, 0, idx, val, K, 32, rs, 64)
            o.eval_x(W, fit, 0, idx, val, K, 32, rs, 64, 32)
            n += 1
            if time.perf_counter() - t0 >= 3 * budget_s:
                break
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 60:
            break
    return {"value": n / el, "unit": "candidates/s", "cores": cores, "kind": "port",
            "compare_with": "scan.value (every output row selected: the only form the reference has); "
                            "metrics_only has no CPU counterpart",
            "sample": f"{n} candidates = K schedule {SCAN_K['ffn_gate']} of ffn_gate 4096->14336 Q8_0 (COO, base fitted once: "
                      f"2048 samples x 4 perms, row_scale, eval_cols=64, eval_x=32) in {el:.1f}s; base fit and evals "
                      f"single-threaded as in the reference, row loop on {cores} threads"}


def spawn_ranks(n: int) -> int:
    """--gpus N without a launcher: start N child ranks (fresh interpreters, one per GPU) BEFORE this process
    makes any GPU call, pass them the rendezvous through the environment torch.distributed.run would set, wait,
    and return the worst exit status. Rank 0 prints the JSON line on the shared stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    import torch
    import llama_cpp_amd  # noqa: F401
    from llama_cpp_amd import seeddelta
    from llama_cpp_amd.payload import config_payload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: SDH_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo (RCCL refuses
    # two ranks on one device); the driver's real multi-GPU runs take the default nccl (= RCCL) path
    rehearsal = os.environ.get("SDH_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    seeddelta.init(local_rank)
    dev = torch.device("cuda", local_rank)

    T = args.tokens
    NP = max(1, args.payloads)
    weights, xs, ys = [], [], []
    alg_bytes = None
    for i in range(NP):
        p = config_payload(args.workload, seed=1234 + i)
        w = seeddelta.SeedDeltaWeight(p)
        weights.append(w)
        alg_bytes = p.algorithmic_bytes(T)
        g = torch.Generator(device="cpu").manual_seed(1 + i)
        xs.append(torch.randn((T, p.n_in), generator=g, dtype=torch.float32).to(dev))
        ys.append(torch.empty((T, p.n_out), dtype=torch.float32, device=dev))
        n_in, n_out = p.n_in, p.n_out
        del p
    torch.cuda.synchronize()

    fwd_flags = seeddelta.FWD_NO_FUSE if args.no_fuse else 0
    fwd_flags |= int(os.environ.get("SDH_BENCH_FLAGS", "0"))        # dev: extra SDH_FWD_* bits (A/B runs)
    fused = (T <= 4 and not args.no_fuse)       # 2..4 tokens: one launch too when the payload qualifies (else two kernels)

    def run_rotation(stream_ptr):
        for i in range(NP):
            weights[i].forward_device(xs[i].data_ptr(), n_in, T, ys[i].data_ptr(), n_out, flags=fwd_flags,
                                      stream=stream_ptr)

    # one step = one rotation over the NP distinct payloads; the driver's --steps / --warmup are taken as given
    steps = max(1, args.steps)
    warm = max(0, args.warmup)
    launches = steps * NP
    cur = torch.cuda.Stream(device=dev)         # explicit launch stream: kernels, span events and graphs all live on it
    sp = cur.cuda_stream
    assert sp != 0

    graph = None
    if args.launch == "graph":
        run_rotation(sp)                        # workspaces + the fused kernel's residency census, outside capture
        cur.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=cur):
            run_rotation(torch.cuda.current_stream().cuda_stream)

    def loop(n_rot):
        if graph is not None:
            with torch.cuda.stream(cur):
                for _ in range(n_rot):
                    graph.replay()
        else:
            for _ in range(n_rot):
                run_rotation(sp)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    run_rotation(sp)                            # one-time set-up (census, first-use tables) is not a warm-up step
    # the GPU's clocks take ~20 ms of load to settle (the same command read 7.8 us per launch from a cold
    # start with --steps 20 and 7.1 us after 50 ms of launches): an untimed settling run of fixed length, whatever W is
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < CLOCK_SETTLE_MS:
        loop(8)
        torch.cuda.synchronize()
    loop(warm)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    # the timed region is also bracketed by two HIP events recorded by the library on the launch stream
    seeddelta.profile_span_begin(sp)
    t0 = time.perf_counter()
    loop(steps)
    span_ms, _ = seeddelta.profile_span_end(sp)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # roofline of the dominant kernel. Fused decode = ONE kernel per token, so its average launch
    # duration is the HIP-event span of the timed region / launches (back-to-back launch period; the
    # rocprofv3 kernel-trace average under profiles/ is the cross-check). Two-launch path: the
    # residual kernel is bracketed per launch by the library's event pairs in a second, untimed pass.
    roof = None
    if not args.no_roofline and rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
        if fused and T == 1 and os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = json.load(fh).get("hbm_bytes_per_launch")
        if fused and graph is not None:
            # events recorded around hipGraphLaunch calls do not bracket the replayed kernels here (measured:
            # 1.8 us "per launch" against 9.3 us on the host clock), so graph mode reports the host clock
            dur_s, n, how = elapsed / launches, launches, "host clock around the timed region (synchronised) / launches"
            kname = "decode_fused_kernel replayed from a hipGraph (base + residual in one launch)"
        elif fused:
            dur_s, n, how = span_ms / 1e3 / launches, launches, "HIP events on the launch stream around the timed region / launches"
            kname = ("decode_direct_kernel (each workgroup forms its rows' base term; one launch)" if seeddelta.last_decode_kind() == 1
                     else "decode_fused_kernel (base + residual in one launch)") if T == 1 else \
                "decode_fused_nt_kernel (2-4 tokens in one launch; shapes it does not cover run as two kernels)"
        else:
            seeddelta.profile_enable(True)
            for _ in range(max(1, min(launches, 4096) // NP)):
                run_rotation(sp)
            torch.cuda.synchronize()
            ms, n = seeddelta.profile_read()
            seeddelta.profile_enable(False)
            dur_s, how = ms / 1e3 / max(n, 1), "HIP event pair around each launch (adds ~2-3 us of event overhead)"
            kname = "delta_rows_kernel (residual; the base kernel runs before it)"
        achieved = alg_bytes / dur_s / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": kname, "avg_kernel_us": dur_s * 1e6, "launches": n, "timing": how,
                "algorithmic_bytes_per_launch": alg_bytes}
        if T == 1:
            # the device's floor for one launch per token: a kernel that only streams the payload's bytes once, launched
            # back to back over 32 distinct buffers (sdh_probe_stream_floor), and an empty launch of the same shape
            try:
                us_stream, us_empty = seeddelta.probe_stream_floor(int(alg_bytes), n_buffers=NP, reps=32, stream=sp)
                roof["attainable"] = {"us_per_launch": us_stream, "frac": alg_bytes / (us_stream * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                      "empty_launch_us": us_empty,
                                      "note": "a launch that only streams these bytes once (no base term, no gathers), back to back: "
                                              "what one launch per token can reach on this device"}
            except Exception as exc:                        # the probe is a measurement aid, never a reason to fail the bench
                roof["attainable"] = {"error": str(exc)[:120]}

    host_io = None
    if args.host_io and rank == 0:
        xh = [x.cpu().numpy() for x in xs]
        for i in range(NP):
            weights[i].forward_host(xh[i])
        n_h, th = 0, time.perf_counter()
        while time.perf_counter() - th < 2.0:
            for i in range(NP):
                weights[i].forward_host(xh[i])
            n_h += NP
        host_io = {"value": n_h * T / (time.perf_counter() - th), "unit": "tok/s",
                   "note": "SDH_FWD_HOST_IO: pageable host x -> device, kernel, y -> host, synchronous per call"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, T, args.cpu_seconds)

    for w in weights:
        w.close()
    weights.clear()
    torch.cuda.empty_cache()

    ffn = None
    if not args.no_ffn and rank == 0 and world == 1:
        ffn = run_ffn(args, dev)
        torch.cuda.empty_cache()

    stack = None
    if not args.no_stack and rank == 0 and world == 1:
        stack = run_ffn_stack(args, dev)
        torch.cuda.empty_cache()

    prefill = None
    if not args.no_prefill and rank == 0 and world == 1:
        prefill = run_prefill(args, dev)
        torch.cuda.empty_cache()

    scan = None
    if not args.no_scan:
        scan = run_scan(args, rank, world, dev)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            scan["cpu_baseline"] = scan_cpu_baseline(args.cpu_seconds)

    if rank == 0:
        value = world * launches * T / elapsed
        line = {
            "metric": "SeedDelta FFN decode tok/s (Mistral-7B)", "value": value, "unit": "tok/s",
            "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "tokens_per_step": NP * T, "launches_per_step": NP,
                       "step": "one rotation over the distinct payloads", "tokens_per_launch": T,
                       "n_in": n_in, "n_out": n_out, "distinct_payloads": NP, "launch": args.launch,
                       "clock_settle_ms": CLOCK_SETTLE_MS, "parallelism": "replicas only" if world > 1 else "single GPU"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        detail = dict(line)
        for key, leg in (("pcie_inclusive", host_io), ("ffn_block", ffn), ("ffn_stack", stack), ("prefill", prefill), ("scan", scan)):
            if leg is not None:
                detail[key] = leg
        # The long form goes to a file; the printed line stays small (a few KB) so that whoever captures the tail of
        # stdout gets ALL of it: headline fields, roofline, cpu_baseline, and one compact object per leg.
        dpath = os.environ.get("SDH_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
        try:
            with open(dpath, "w") as fh:
                json.dump(detail, fh, indent=1)
            line["detail_file"] = os.path.relpath(dpath, ROOT)
        except OSError:
            line["detail_file"] = None
        r3 = lambda v: None if v is None else float("%.4g" % v)         # noqa: E731
        if roof is not None:
            line["roofline"] = {k: (r3(v) if isinstance(v, float) else v) for k, v in roof.items() if k not in ("timing", "attainable")}
            if isinstance(roof.get("attainable"), dict) and "frac" in roof["attainable"]:
                line["roofline"]["attainable"] = {k: r3(v) for k, v in roof["attainable"].items() if k != "note"}
        if cpu is not None:
            line["cpu_baseline"] = dict(cpu, value=r3(cpu["value"]), sample=cpu["sample"][:120])
        if host_io is not None:
            line["pcie_inclusive"] = {"value": r3(host_io["value"]), "unit": "tok/s"}
        if ffn is not None:
            line["ffn_block"] = {"value": r3(ffn["value"]), "unit": ffn["unit"], "us_per_layer": r3(ffn["us_per_layer"]),
                                 "bound": "hbm", "frac": r3(ffn["roofline"]["frac"]), "timing_consistent": ffn["timing_consistent"]}
            if ffn.get("prefill_layer"):
                pl = ffn["prefill_layer"]
                line["ffn_block"]["prefill_layer"] = {"tokens": pl["tokens"], "ms": r3(pl["ms_per_layer"]["default"]),
                                                      "fp32_ms": r3(pl["ms_per_layer"]["fp32"]),
                                                      "fold_ms": r3(pl["ms_per_layer"]["mfma_fold"]),
                                                      "fold_tok_s": r3(pl["tokens_per_s_fold"]), "fold_mfma_frac": r3(pl["mfma_frac_fold"])}
        if stack is not None:
            line["ffn_stack"] = {"value": r3(stack["value"]), "unit": stack["unit"], "us_per_layer": r3(stack["us_per_layer"]),
                                 "bound": "hbm", "frac": r3(stack["roofline"]["frac"]), "handoff_timeouts": stack["handoff_timeouts"]}
        if prefill is not None:
            legs = {}
            for name, leg in prefill["workloads"].items():
                best = min(("fp32", "w0_mfma", "rows_lds", "mfma_fold"), key=lambda t: leg[t]["ms"] if t in leg else 1e30)
                legs[name] = {"ms": r3(leg["default"]["ms"]), "hbm_frac": r3(leg["default"]["hbm"]["frac"]),
                              "fp32_ms": r3(leg["fp32"]["ms"]),
                              "mfma_ms": r3(leg["w0_mfma"]["ms"]), "mfma_frac": r3(leg["w0_mfma"]["mfma"]["frac"]),
                              "rows_ms": r3(leg["rows_lds"]["ms"]), "fold_ms": r3(leg["mfma_fold"]["ms"]),
                              "fold_mfma_frac": r3(leg["mfma_fold"]["mfma"]["frac"]), "best": best}
            line["prefill"] = {"tokens": 512, "unit": "ms per ubatch and tensor", "workloads": legs}
        if scan is not None:
            sc = {"value": r3(scan["value"]), "unit": scan["unit"], "n_gpus": scan["n_gpus"], "scaling": scan["scaling"],
                  "candidates": scan["candidates"], "seconds": r3(scan["seconds"]), "bound": "hbm",
                  "frac": r3(scan["roofline"]["frac"]), "traffic_per_candidate": scan["roofline"]["traffic"],
                  "upload_inclusive": r3(scan["upload_inclusive"]["value"]), "metrics_only": r3(scan["metrics_only"]["value"]),
                  "host_threads_per_rank": scan["config"]["host_threads_per_rank"]}
            if scan.get("cpu_baseline"):
                cb = scan["cpu_baseline"]
                sc["cpu_baseline"] = {"value": r3(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"]}
            line["scan"] = sc
        out = json.dumps(line)
        assert len(out) < 6000, len(out)
        sys.stdout.flush()
        print(out, flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
