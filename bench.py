#!/usr/bin/env python
"""Headline benchmark: points/sec, forward+backward (+Adam step), DGCNN + MLSP heads, B=32 N=1024 k=20
per GPU, fp32, synthetic clouds (BASELINE.json configs[1]); weak scaling over N GPUs with one RCCL
all-reduce of the flat 18.2 MB gradient bucket per step.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant kernel
(HIP-event timed inside the timed region through the library's profiling hook) and `cpu_baseline`
(the CPU oracle restating the reference's op sequence, timed on this box's host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, NPTS, K_NN = 32, 1024, 20
FLOP_PER_POINT = 28.19e6        # algorithmic fwd+bwd FLOP per point (reference op sequence)
PEAK_FP32_TFLOPS = 157.3        # MI355X_MICROARCH.md: f32 MFMA == f32 vector peak
PEAK_HBM_GBS = 8000.0
PEAK_BF16_TFLOPS = 2500.0       # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_SPLIT_TFLOPS = PEAK_BF16_TFLOPS / 6     # fp32-accurate product = six bf16 piece products (gemm_split_kernel, mode "bf16x6")
PEAK_SPLIT3_TFLOPS = PEAK_BF16_TFLOPS / 3    # ... = three f16 piece products (mode "f16x3"; the dense f16 MFMA peak equals the bf16 one)


def split_peak(half_launches, split_launches):
    """(peak TFLOP/s fp32-equivalent, piece products per fp32 product, note) of the launches that ran on gemm_split_kernel: priced against
    the three-product peak only when EVERY one of them ran on the two-piece f16 products, else against the launch-weighted harmonic mix."""
    if split_launches <= 0 or half_launches <= 0:
        return PEAK_SPLIT_TFLOPS, 6.0, "dense bf16 MFMA peak 2500 TFLOP/s / 6 piece products per fp32 product"
    if half_launches >= split_launches:
        return PEAK_SPLIT3_TFLOPS, 3.0, "dense f16 MFMA peak 2500 TFLOP/s / 3 piece products per fp32 product (every launch on the two-piece f16 split)"
    f = half_launches / split_launches
    pp = 3.0 * f + 6.0 * (1.0 - f)
    return PEAK_BF16_TFLOPS / pp, pp, "dense 16-bit MFMA peak 2500 TFLOP/s / %.2f piece products per fp32 product (%.0f %% of the launches on the two-piece f16 split, the rest on three bf16 pieces)" % (pp, 100 * f)


PMC_WORKLOAD_FILES = {"configs1": r"summary(_v\d+)?\.csv$", "configs4": r"summary_config4(_v\d+)?\.csv$"}


def pmc_summary_file(workload="configs1", root=None):
    """The committed rocprofv3 --pmc summary OF THIS WORKLOAD (profiles/pmc_r<round>/summary[_vN].csv for the headline step,
    summary_config4[_vN].csv for the configs[4] step): newest round, then newest version.  The file is chosen by workload, never by name
    order (round 5's line read the configs[4] profile for the headline because `summary_config4.csv` sorts after `summary.csv`)."""
    import glob, re
    root = root or os.path.dirname(os.path.abspath(__file__))
    pat = re.compile(PMC_WORKLOAD_FILES[workload])
    files = [p for p in glob.glob(os.path.join(root, "profiles", "pmc_r*", "summary*.csv")) if pat.search(os.path.basename(p))]
    if not files:
        return None

    def order(path):
        ver = re.search(r"_v(\d+)\.csv$", path)
        return (int(re.search(r"pmc_r(\d+)", path).group(1)), int(ver.group(1)) if ver else 0)
    return max(files, key=order)


def pmc_gemm_traffic(kernel="gemm_split_kernel", workload="configs1", root=None):
    """(HBM-side bytes per launch of the kernels whose name contains `kernel`, source file, launches counted) from the committed --pmc
    summary of `workload` (tools/prof/run_profiles.sh: FETCH_SIZE / WRITE_SIZE in separate passes): launch-weighted mean of
    2*FETCH_SIZE + WRITE_SIZE (the gfx950 correction of MI355X_MICROARCH.md).  (None, None, 0) when no summary of the workload is
    committed; RuntimeError when the chosen file holds NO row of the priced kernel (a traffic figure of another kernel is worse than none)."""
    import csv
    root = root or os.path.dirname(os.path.abspath(__file__))
    f = pmc_summary_file(workload, root)
    if f is None:
        return None, None, 0
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if kernel in r["kernel"]:
            l = int(r["launches"])
            tot += l * (2 * float(r["fetch_KB_per_launch_raw"]) + float(r["write_KB_per_launch"])) * 1024
            n += l
    if n == 0:
        raise RuntimeError("bench.py: %s holds no row of %s -- the committed counter profile does not cover the priced kernel; re-run "
                           "tools/prof/run_profiles.sh for this workload" % (os.path.relpath(f, root), kernel))
    return tot / n, os.path.relpath(f, root), n


def pmc_split_traffic_by_kind(workload="configs1", root=None):
    """HBM-side bytes per launch (2*FETCH_SIZE + WRITE_SIZE) of gemm_split_kernel by kind -- forward <false, true, ..>, dgrad <false, false, ..>,
    wgrad <true, false, ..> -- from the committed --pmc summary of `workload`: ({kind: (bytes per launch, launches)}, source)."""
    import csv, re
    root = root or os.path.dirname(os.path.abspath(__file__))
    f = pmc_summary_file(workload, root)
    if f is None:
        return {}, None
    acc = {}
    for r in csv.DictReader(open(f)):
        m = re.search(r"gemm_split_kernel<(false|true)[;,] (false|true)", r["kernel"])
        if not m:
            continue
        kind = "wgrad" if m.group(1) == "true" else ("fwd" if m.group(2) == "true" else "dgrad")
        l = int(r["launches"])
        t, n = acc.get(kind, (0.0, 0))
        acc[kind] = (t + l * (2 * float(r["fetch_KB_per_launch_raw"]) + float(r["write_KB_per_launch"])) * 1024, n + l)
    return {k: (t / n, n) for k, (t, n) in acc.items() if n}, os.path.relpath(f, root)


def count_launches(step_fn, nsteps):
    """Device launches of `nsteps` steps COUNTED IN THIS RUN (an untimed block): torch.profiler's device-side kernel records (roctracer /
    rocprofiler-sdk underneath: every kernel of the process, the library's and torch's alike, with its duration) -> dict with launches
    per step, launches under 8 us per step and their device time, memset / memcpy records separately, and the ten most frequent
    kernels under 8 us.  None when the profiler delivers no device records on this box (the line then says so)."""
    try:
        from torch.profiler import profile, ProfilerActivity
        from torch.autograd import DeviceType
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=False) as prof:
            for _ in range(nsteps):
                step_fn()
            torch.cuda.synchronize()
        evs = [e for e in prof.events() if e.device_type == DeviceType.CUDA]
    except Exception as exc:                      # (profiler unavailable: report, do not guess)
        return {"error": "%s: %s" % (type(exc).__name__, exc)}
    if not evs:
        return None
    kern = [e for e in evs if not (e.name.startswith("Memcpy") or e.name.startswith("Memset"))]
    other = len(evs) - len(kern)
    dur = [e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total for e in kern]     # microseconds
    small = [(e.name, d) for e, d in zip(kern, dur) if d < 8.0]
    freq = {}
    for name, d in small:
        key = name.split("(")[0][:80]
        c, t = freq.get(key, (0, 0.0))
        freq[key] = (c + 1, t + d)
    top = sorted(freq.items(), key=lambda kv: -kv[1][0])[:10]
    return {"launches_per_step": len(kern) / nsteps, "launches_under_8us_per_step": len(small) / nsteps,
            "us_under_8us_per_step": sum(d for _, d in small) / nsteps, "kernel_us_per_step": sum(dur) / nsteps,
            "memset_memcpy_records_per_step": other / nsteps, "steps_counted": nsteps,
            "most_frequent_under_8us": [{"kernel": k, "per_step": c / nsteps, "us_per_step": t / nsteps} for k, (c, t) in top]}


def synth_batch(B, N, device, seed=0):
    """x ~ U[-1,1), first-41-points mask, N(0,1) normals, counts U{0..30} -> soft 16-bin label."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, N, generator=g) * 2 - 1
    gold = x + 0.05 * torch.randn(B, 3, N, generator=g)
    mask = torch.zeros(B, 3, N)
    mask[:, :, :41] = 1.0
    normal_gt = torch.randn(B, N, 3, generator=g)
    count = torch.randint(0, 31, (B * N,), generator=g)
    c1, c2 = count // 2, (count + 1) // 2
    eye = torch.eye(16)
    dens_vec = (eye[c1] + eye[c2]) / 2.0
    out = dict(x=x, gold=gold, mask=mask, normal_gt=normal_gt, dens_vec=dens_vec, dens_val=count.float(),
               mask_cord=(mask.permute(0, 2, 1)[:, :, 0] * 26 + 1).contiguous())
    return {k: v.to(device) for k, v in out.items()}


def make_args(cuda=True):
    """The argparse fields the model and the losses read (PointDA/trainer.py:44-111; Models.py:85,175,255,269; model_utils.py:25,98-99,
    133; mlsp.py:228,286,445-452) at the trainer's defaults."""
    return argparse.Namespace(num_class=10, dropout=0.5, model="dgcnn", encoder_type=None, cuda=cuda, density_num_class=16,
                              pergroup=2.0, DefRec_weight=0.5, normal_pred_weight=0.5, Density_weight=0.05, Scan_Rec_weight=0.5)


def make_seg_args():
    """PointSegDA/Models.py:25-28 (gpus), :257-258 (dropout), :353 (density_num_class), :370 (pergroup)."""
    return argparse.Namespace(gpus=[0], dropout=0.5, density_num_class=16, pergroup=2.0, DefRec_weight=0.5, normal_pred_weight=0.5,
                              Density_weight=0.05)


def make_adam(params):
    """optim.Adam(model.parameters(), lr, weight_decay) of PointDA/trainer.py:258-259 -- as mlsp_amd.optim.FlatAdam (the same fused kernel
    over flat parameter / gradient / moment buffers: one launch, bit-identical parameters); MLSP_BENCH_TORCH_ADAM=1: torch's own, for A/B."""
    if os.environ.get("MLSP_BENCH_TORCH_ADAM"):
        return torch.optim.Adam(params, lr=1e-3, weight_decay=5e-5, fused=True)
    from mlsp_amd.optim import FlatAdam
    return FlatAdam(params, lr=1e-3, weight_decay=5e-5)


def gpu_step(model, mlsp, args, batch, opt):
    """zero_grad -> forward (all three heads) -> position + normal + cardinality losses -> backward -> step
    (PointDA/trainer.py:542-571, target branch)."""
    opt.zero_grad()
    logits = model(batch["x"], activate_density_normal_ondef=True)
    loss = mlsp.calc_loss(args, logits, batch["gold"], batch["mask"])
    loss = loss + mlsp.calc_masked_normal_loss(args, logits["Normal"], batch["normal_gt"], batch["mask_cord"])
    kl, mae = mlsp.densityloss(args, logits, batch["dens_val"], batch["dens_vec"], mask=batch["mask_cord"].reshape(-1))
    loss = loss + kl + mae
    loss.backward()
    opt.step()
    return loss


def _cpu_info():
    """(model string, physical cores usable by this process, logical CPUs) from /proc/cpuinfo and the affinity mask."""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    physical = min(len(phys) or max(1, logical // 2), usable)
    return model, max(1, physical), logical


def cpu_baseline(budget_s=60.0):
    """The reference's CPU path, timed on this box's host cores beside the GPU number: stock
    torch modules with the reference's operator sequence (oracle/ref_torch_modules.py: matmul+topk kNN, index gather, cat,
    nn.Conv2d, nn.BatchNorm, the [B,N,N,3] Chamfer tensor), pinned to the reference's golden vectors by
    tests/test_oracle_golden.py.  Same timed region as the GPU step minus the optimizer (zero_grad -> forward with the three heads
    -> three losses -> backward), dropout 0.5, BN train.  Bounded sample: B = 8 and B = 32 clouds of N = 1024 at threads =
    physical cores (and 32 when the box has more: torch's CPU kernels stop scaling long before 128 threads), B = 8 at one thread."""
    from oracle import ref_torch_modules as rtm
    model_name, physical, logical = _cpu_info()
    args = make_args(cuda=False)
    torch.manual_seed(0)
    model = rtm.StockDGCNN(args).train()
    batches = {}

    def step(Bc):
        if Bc not in batches:
            batches[Bc] = synth_batch(Bc, NPTS, torch.device("cpu"))
        b = batches[Bc]
        for p in model.parameters():
            p.grad = None
        rtm.step_loss(args, model(b["x"], activate_density_normal_ondef=True), b).backward()

    # second restatement: the build's own functional oracle (oracle/ref_cpu.py: einsum convolutions, canonical C kNN) -- SURVEY 8d asks for both
    from oracle import ref_cpu, knn_canon
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}

    def step_oracle(Bc):
        if Bc not in batches:
            batches[Bc] = synth_batch(Bc, NPTS, torch.device("cpu"))
        b = batches[Bc]
        for p in params.values():
            p.grad = None
        out, _ = ref_cpu.dgcnn_forward(params, b["x"], training=True, dropout_p=0.5, knn_fn=knn_canon.knn, activate_density_normal_ondef=True)
        rtm.step_loss(args, out, b).backward()

    t_start = time.perf_counter()
    torch.set_num_threads(physical)
    step(2)                                                # untimed: allocator / thread-pool / first-touch warm-up
    # (threads, clouds, timed steps, which restatement); cheapest runs first so that the time budget can only drop the slow tail on a
    # many-core host; >= 3 timed steps for the headline leg
    mt = 32 if physical > 32 else physical
    plan = [(mt, 8, 3, "stock"), (mt, 32, 3, "stock"), (1, 8, 1, "stock"), (mt, 8, 2, "oracle")]
    if physical > 32:
        plan.append((physical, 8, 2, "stock"))
    runs = []
    for threads, Bc, nsteps, which in plan:
        if len(runs) >= 2 and time.perf_counter() - t_start > budget_s:
            break
        torch.set_num_threads(threads)
        fn = step if which == "stock" else step_oracle
        t0 = time.perf_counter()
        for _ in range(nsteps):
            fn(Bc)
        dt = (time.perf_counter() - t0) / nsteps
        runs.append({"threads": threads, "B": Bc, "steps": nsteps, "restatement": "stock torch modules" if which == "stock" else "oracle/ref_cpu.py",
                     "s_per_step": round(dt, 3), "points_per_s": round(Bc * NPTS / dt, 1)})
    best = max((r for r in runs if r["threads"] > 1 and r["restatement"].startswith("stock")), key=lambda r: r["points_per_s"])
    return {"value": best["points_per_s"], "unit": "points/s", "cores": best["threads"], "kind": "port",
            "sample": "stock-torch restatement of the reference's op sequence (oracle/ref_torch_modules.py, golden-pinned): fwd + 3 "
                      "losses + bwd, N=%d k=%d fp32, dropout 0.5; best of the multi-thread runs (B=%d, %d threads, %d timed steps, %.2f s/step); "
                      "`runs` also holds the build's own functional oracle (oracle/ref_cpu.py); "
                      "host: %s, %d physical cores usable, os.cpu_count()=%d" % (NPTS, K_NN, best["B"], best["threads"], best["steps"],
                                                                                 best["s_per_step"], model_name, physical, logical),
            "runs": runs, "cpu_model": model_name, "physical_cores": physical, "logical_cpus": logical}


PROF_CLASSES = 8          # include/mlsp_hip.h MLSP_PROF_CLASSES


def profiled_steps(lib, step_fn, nsteps):
    """`nsteps` untimed steps with the library's HIP-event hook armed (events on the launch stream around every launch of the priced
    kernel families) -> (per-class [ms, launches, work] rows, [gemm ms, launches, flop, algorithmic bytes], seconds per step)."""
    import ctypes
    lib.mlsp_profile_begin()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        step_fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / nsteps
    buf = (ctypes.c_double * 4)()
    lib.mlsp_profile_end(buf)
    cls = (ctypes.c_double * (3 * PROF_CLASSES))()
    lib.mlsp_profile_classes(cls, PROF_CLASSES)
    rows = [list(cls[3 * c:3 * c + 3]) for c in range(PROF_CLASSES)]
    kinds = (ctypes.c_double * 16)()
    lib.mlsp_profile_split_kinds(kinds)
    profiled_steps.split_kinds = {k: list(kinds[4 * i:4 * i + 4]) for i, k in enumerate(("fwd", "dgrad", "wgrad"))}    # [ms, launches, FLOP, bytes]
    profiled_steps.split_half = list(kinds[12:16])                  # the launches of all kinds on the two-piece f16 products
    return rows, list(buf), dt


def _kernel_entry(name, bound, row, nsteps, note):
    """One `roofline_kernels` entry from a class row [ms, launches, work]: achieved = algorithmic work / HIP-event time."""
    ms, n, work = row
    if n <= 0 or ms <= 0:
        return None
    if bound == "hbm":
        ach, peak, unit = work / (ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
    else:
        ach, peak, unit = work / (ms * 1e-3) / 1e12, PEAK_FP32_TFLOPS, "TFLOP/s"
    return {"kernel": name, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
            "launches_per_step": n / nsteps, "avg_us": 1e3 * ms / n, "us_per_step": 1e3 * ms / nsteps, "note": note}


def median_block_ms(step_fn, steps, repeats, warm):
    """Median block (ms per step) of `repeats` timed blocks of `steps` steps, after `warm` steps and untimed settle blocks (until two
    consecutive ones agree to 1 %, at most 4: a fresh model's first blocks run at ramping clocks).  The cyclic garbage collector is
    paused inside a block and run between blocks (what `timeit` does): a generation-2 pass in the middle of a block showed as 5.9-6.5 ms
    blocks among 5.2 ms ones (configs[4]) and as a 10.8 ms settle block of the headline."""
    import gc
    for _ in range(warm):
        step_fn()
    torch.cuda.synchronize()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        prev = None
        for _ in range(4):
            gc.collect()
            t0 = time.perf_counter()
            for _ in range(steps):
                step_fn()
            torch.cuda.synchronize()
            cur = time.perf_counter() - t0
            if prev is not None and abs(cur - prev) <= 0.01 * prev:
                break
            prev = cur
        ts = []
        for _ in range(repeats):
            gc.collect()                               # untimed, between blocks: garbage never accumulates over more than one block
            t0 = time.perf_counter()
            for _ in range(steps):
                step_fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / steps)
    finally:
        if was_enabled:
            gc.enable()
    median_block_ms.last_blocks = [round(1e3 * t, 3) for t in ts]       # for the caller's record
    return 1e3 * sorted(ts)[len(ts) // 2]


def secondary_workloads(lib, dev):
    """BASELINE.json configs[3] and configs[4] on this GPU, AFTER the headline blocks (never inside its timed region): a few steps each,
    median of five blocks (10 steps for configs[3], 20 for configs[4]) after settle blocks, plus the GEMM family's roofline entry from one HIP-event-profiled block."""
    from mlsp_amd import pointnet2 as p2, seg_models, functional as Fh
    out = []
    # configs[3]: PointNet++ set-abstraction encoder (hengshuang_transformer/pointnet_util.py:159-196), B=32 N=2048, fwd + bwd, fp32
    torch.manual_seed(0)
    B, N = 32, 2048
    xyz = (torch.rand(B, N, 3) * 2 - 1).to(dev)
    layers = [p2.PointNetSetAbstraction(512, 0.2, 32, 3, [64, 64, 128], False),
              p2.PointNetSetAbstraction(128, 0.4, 64, 131, [128, 128, 256], False),
              p2.PointNetSetAbstraction(None, None, None, 259, [256, 512, 1024], True)]
    for l in layers:
        l.to(dev)
    params = [p for l in layers for p in l.parameters()]

    def sa_step():
        for p in params:
            p.grad = None
        x, f = xyz, None
        for l in layers:
            x, f = l(x, f)
        f.sum().backward()

    ms = median_block_ms(sa_step, 10, 5, 4)
    sa_blocks = median_block_ms.last_blocks
    rows, g, _ = profiled_steps(lib, sa_step, 2)
    sp = rows[7]                                   # the launches that ran on the bf16-split kernel; the rest ran on the f32 MFMA kernels
    f32_ms, f32_flop = g[0] - sp[0], g[2] - sp[2]
    ach_sp = sp[2] / (sp[0] * 1e-3) / 1e12 if sp[0] > 0 else 0.0
    ach32 = f32_flop / (f32_ms * 1e-3) / 1e12 if f32_ms > 0 else 0.0
    sa_peak, _, sa_peak_note = split_peak(profiled_steps.split_half[1], sp[1])
    roof = ({"kernel": "gemm_split_kernel<*> (SA-MLP contractions, fp32-accurate piece products on the 16-bit matrix cores)", "bound": "mfma", "achieved": ach_sp,
             "peak": sa_peak, "peak_note": sa_peak_note, "unit": "TFLOP/s", "frac": ach_sp / sa_peak, "vs_f32_mfma_peak": ach_sp / PEAK_FP32_TFLOPS,
             "share_of_step": sp[0] / 2 / ms, "launches_per_step": sp[1] / 2,
             "f32_mfma_launches": {"achieved": ach32, "frac": ach32 / PEAK_FP32_TFLOPS, "share_of_step": f32_ms / 2 / ms,
                                   "launches_per_step": (g[1] - sp[1]) / 2}} if sp[1] > 0 else
            {"kernel": "gemm_f32_kernel<*> (SA-MLP contractions)", "bound": "mfma", "achieved": ach32, "peak": PEAK_FP32_TFLOPS,
             "unit": "TFLOP/s", "frac": ach32 / PEAK_FP32_TFLOPS, "share_of_step": g[0] / 2 / ms, "launches_per_step": g[1] / 2})
    out.append({"workload": "PointNet++ SA encoder (3 set-abstraction layers: FPS + ball query + grouping + SA-MLP), fwd+bwd, B=32 N=2048 "
                            "(BASELINE.json configs[3])", "ms_per_step": ms, "points_per_s": B * N / ms * 1e3, "dtype": "f32",
                "blocks_ms_per_step": sa_blocks, "gemm_products": Fh.gemm_precision.current, "roofline": roof})
    del layers, params, xyz

    # configs[4]: PointSegDA DGCNN_DefRec (PointSegDA/Models.py:197-242), N=2048, k=40, all heads, fwd + bwd + Adam, bf16 GEMM operands and
    # bf16 activation storage (fp32 accumulation / statistics / kNN / losses); B=16 per GPU = the reference trainer's batch
    B, N, K = 16, 2048, 40
    torch.manual_seed(0)
    seg = seg_models.DGCNN_DefRec(make_seg_args(), in_size=3, num_classes=8)
    seg.k = seg.shared_layers.k = K
    seg = seg.to(dev).train()
    opt = make_adam(seg.parameters())
    x = torch.rand(B, 3, N, device=dev) * 2 - 1
    w = {k: torch.randn(sh, device=dev) for k, sh in (("seg", (B, N, 8)), ("DefRec", (B, N, 3)), ("Normal", (B, N, 3)),
                                                      ("density", (B * N, 16)), ("density_mse", (B * N,)))}

    def seg_step():
        opt.zero_grad()
        o = seg(x, make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True)
        sum((o[k].float() * w[k]).mean() for k in w).backward()
        opt.step()

    # the same step with the backward SEEDED by fixed output gradients (the gradients the weighted means above produce) instead of
    # ~25 small torch element-wise / reduction launches per step of synthetic loss arithmetic: model forward + backward + Adam only
    # (0.1-0.15 ms of device time at this shape; the step is device-bound: tools/r5/enqueue_c4.py).
    wg = {k: w[k] / w[k].numel() for k in w}

    def seg_step_seeded():
        opt.zero_grad()
        o = seg(x, make_seg=True, activate_DefRec=True, activate_density_normal_ondef=True)
        torch.autograd.backward([o[k] for k in w], [wg[k] for k in w])
        opt.step()

    with Fh.gemm_precision("bf16"), Fh.activation_storage("bf16"):
        ms = median_block_ms(seg_step, 20, 5, 4)
        seg_blocks = median_block_ms.last_blocks
        ms_seeded = median_block_ms(seg_step_seeded, 20, 3, 2)
        rows, g, _ = profiled_steps(lib, seg_step, 2)
    gbs = g[3] / (g[0] * 1e-3) / 1e9 if g[0] > 0 else 0.0
    tfs = g[2] / (g[0] * 1e-3) / 1e12 if g[0] > 0 else 0.0
    try:                # HBM-side bytes of the bf16 GEMM launches from the counter profile OF THIS WORKLOAD (summary_config4*.csv)
        c4_traffic, c4_src, _ = pmc_gemm_traffic("gemm_bf16_kernel", "configs4")
    except RuntimeError as exc:
        c4_traffic, c4_src = None, str(exc)
    out.append({"workload": "PointSegDA DGCNN_DefRec + seg + 3 MLSP heads, fwd+bwd+Adam, B=16 N=2048 k=40 (BASELINE.json configs[4], one GPU)",
                # the path = model forward + backward + Adam: the backward is SEEDED with the output gradients of the synthetic loss (what a
                # trainer's fused loss kernels would hand it); the same step with that loss spelled out in ~25 small torch launches beside it
                "ms_per_step": ms_seeded, "points_per_s": B * N / ms_seeded * 1e3, "dtype": "bf16 GEMM operands + bf16 activation storage, fp32 accumulate",
                "backward": "seeded with the output gradients of the synthetic loss (torch.autograd.backward)",
                "with_synthetic_loss_in_torch_ops": {"ms_per_step": ms, "blocks_ms_per_step": seg_blocks, "points_per_s": B * N / ms * 1e3,
                                                     "note": "same forward / backward / Adam plus sum((out * w).mean()) per output in torch: ~25 small "
                                                             "element-wise / reduction launches of harness, 0.1 ms of device time (the round 2-5 figure)"},
                "seeded_backward": {"ms_per_step": ms_seeded, "points_per_s": B * N / ms_seeded * 1e3, "note": "= ms_per_step (kept for readers of earlier lines)"},
                "roofline": {"kernel": "gemm_bf16_kernel<*> / gemm_f32_kernel<*> (every MFMA GEMM launch)", "bound": "hbm", "achieved": gbs,
                             "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "mfma_tflops": tfs,
                             "mfma_frac_of_bf16_dense_peak": tfs / PEAK_BF16_TFLOPS, "share_of_step": g[0] / 2 / ms,
                             "launches_per_step": g[1] / 2, "traffic": c4_traffic, "traffic_source": c4_src,
                             "algorithmic_bytes_per_launch": g[3] / g[1] if g[1] else None,
                             "note": "achieved = algorithmic operand + result bytes of the launches (A + B + C in their storage types) / "
                                     "HIP-event time: at bf16 the contractions sit under the HBM roof, not the 2.5 PF matrix roof"},
                "knn": [e for e in (_kernel_entry("kNN C=3, k=40, N=2048 (knn6w_kernel + prep + v5 on flagged clouds)", "hbm", rows[1], 2, "compulsory (C+k)*4 B/pt"),
                                    _kernel_entry("kNN C=64, k=40, N=2048", "mfma", rows[2], 2, "2*N*C FLOP per point, one pass")) if e]})
    return out


def trainer_shaped_workload(dev, steps=10, repeats=3):
    """One optimizer step as PointDA/trainer.py runs it (lines 374-571, the default flags of train.sh: Density_normal_viainput,
    Normal_ondef, Density_ondef, PCM), on synthetic loaders, B = 32 clouds per domain, N = 1024, with the model and the MLSP functions
    imported through the drop-in shim paths the unmodified trainer uses: source branch = deform_input -> forward (position head) ->
    masked Chamfer -> backward, PCM.mix_shapes -> forward (classifier) -> mixup CE -> backward; target branch = normals + cardinality
    labels ON DEVICE (the trainer's per-cloud python-pcl loops, trainer.py:525-536) -> deform_input -> forward (three
    heads) -> three losses with the trainer's inline masked normal loss (:551-556) -> backward; Adam.  Per-component device time from
    HIP events on the launch stream; beside it the numpy restatements of the label generators and the corruption (oracle/labels_np.py,
    oracle/ref_corrupt_np.py) timed on this box's host cores: what the reference's CPU loops cost per step."""
    import numpy as np
    import torch.nn.functional as F
    shim = os.path.join(ROOT, "mlsp_amd", "shims")
    sys.path.insert(0, shim)
    try:
        from PointDA.Models import DGCNN                     # trainer.py:14
        from MLSP import PCM, mlsp                           # trainer.py:15
        import pcl                                           # trainer.py:18
        from mlsp_amd import pc_utils
        args = make_args()
        args.radius, args.near, args.DefRec_dist, args.mixup_params = 0.135, 20, "volume_based_voxels", 1.0      # trainer.py:103-119
        torch.manual_seed(0)
        np.random.seed(0)
        model = DGCNN(args).to(dev).train()
        opt = make_adam(model.parameters())
        criterion = torch.nn.CrossEntropyLoss()
        lookup = torch.Tensor(pc_utils.region_mean(3)).to(dev)
        B, N = B_PER_GPU, NPTS
        g = torch.Generator().manual_seed(0)
        # clouds in [-0.66, 0.66]^3: the centre voxel of the 3 x 3 x 3 grid holds ~N/8 >= 40 points (deform_input's minimum)
        src = ((torch.rand(B, N, 3, generator=g) * 2 - 1) * 0.66).to(dev)
        trg = ((torch.rand(B, N, 3, generator=g) * 2 - 1) * 0.66).to(dev)
        src_label = torch.randint(0, 10, (B,), generator=g).to(dev)
        names = ["src deform_input", "src fwd(DefRec)+Chamfer+bwd", "src PCM.mix_shapes (2 x FPS)", "src fwd(cls)+CE+bwd",
                 "trgt normals k=20 (device)", "trgt cal_density (device)", "trgt deform_input", "trgt fwd(3 heads)+losses+bwd", "Adam"]
        ev = None

        def mark(i):
            if ev is not None:
                ev[i].record()

        def step():
            opt.zero_grad()
            mark(0)
            sd = src.permute(0, 2, 1)                         # the trainer's non-contiguous [B,3,N] view
            orig = sd.clone()
            sd, smask = mlsp.deform_input(sd, lookup, args.DefRec_dist, dev)
            mark(1)
            mlsp.calc_loss(args, model(sd, activate_DefRec=True), orig, smask).backward()
            mark(2)
            mixed, vals = PCM.mix_shapes(args, orig.clone(), src_label)
            mark(3)
            PCM.calc_loss(args, model(mixed, activate_DefRec=False), vals, criterion).backward()
            mark(4)
            normal_gt = mlsp.estimate_normals(trg, near=args.near)
            mark(5)
            dl, dml = mlsp.cal_density_gpu(trg, args.radius, args.density_num_class, args.pergroup)
            mark(6)
            td = trg.permute(0, 2, 1)
            torig = td.clone()
            td, mask = mlsp.deform_input(td, lookup, args.DefRec_dist, dev)
            mark(7)
            lp = model(td, activate_density_normal_ondef=True)
            loss = mlsp.calc_loss(args, lp, torig, mask)
            mask_cord = mask.permute(0, 2, 1)[:, :, 0] * 26 + 1
            npred, ngt = F.normalize(lp["Normal"], p=2, dim=-1), F.normalize(normal_gt, p=2, dim=-1)
            loss = loss + args.normal_pred_weight * (-torch.sum(torch.abs(torch.sum(npred * ngt, dim=-1)) * mask_cord) / torch.sum(mask_cord))
            kl, mae = mlsp.densityloss(args, lp, dml.reshape(-1), dl.reshape(-1, args.density_num_class), mask=mask_cord.reshape(-1))
            (loss + kl + mae).backward()
            mark(8)
            opt.step()
            mark(9)

        ms = median_block_ms(step, steps, repeats, 3)
        blocks = median_block_ms.last_blocks
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(10)]
        comp = [0.0] * 9
        nprof = 5
        for _ in range(nprof):
            step()
            torch.cuda.synchronize()
            for i in range(9):
                comp[i] += ev[i].elapsed_time(ev[i + 1]) / nprof
        ev = None
        # the unmodified trainer's label loop through the pcl shim: B host round trips per step (trainer.py:525-531)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(B):
            cloud = pcl.PointCloud()
            cloud.from_array(np.array(trg[i].cpu().numpy(), dtype=np.float32))
            ne = cloud.make_NormalEstimation()
            ne.set_SearchMethod(cloud.make_kdtree())
            ne.set_KSearch(args.near)
            ne.compute().to_array()
        shim_loop_ms = 1e3 * (time.perf_counter() - t0)
        # CPU beside it: numpy restatements on a bounded sample of the same clouds, scaled to B clouds
        from oracle import labels_np, ref_corrupt_np, knn_canon
        nb = 4
        tc = trg[:nb].cpu()
        t0 = time.perf_counter()
        labels_np.cal_density(tc.numpy(), args.radius, args.density_num_class, args.pergroup)
        t_den = (time.perf_counter() - t0) * B / nb
        t0 = time.perf_counter()
        idx = knn_canon.knn_point_major(tc, args.near)
        for i in range(nb):
            labels_np.knn_normals(tc[i].numpy(), idx[i])
        t_nrm = (time.perf_counter() - t0) * B / nb
        Xc = tc.permute(0, 2, 1).contiguous().numpy()
        t0 = time.perf_counter()
        ref_corrupt_np.deform(Xc, lookup.cpu().numpy(), np.random.permutation(27), np.random.randn(nb, 3, N).astype(np.float32))
        t_def = (time.perf_counter() - t0) * B / nb * 2          # two deform_input calls per step
        t0 = time.perf_counter()
        ref_corrupt_np.mix_shapes(Xc, np.random.permutation(nb), 0.5, np.zeros(nb, np.int64), np.zeros(nb, np.int64), np.random.permutation(N))
        t_mix = (time.perf_counter() - t0) * B / nb
        return {"workload": "trainer-shaped optimizer step (PointDA/trainer.py:374-571 through the shim imports): source DefRec + PCM "
                            "branches, target branch with on-device normals / cardinality labels / deform_input, 3 forwards + 3 backwards "
                            "+ Adam, B=%d per domain, N=%d" % (B, N),
                "ms_per_step": ms, "blocks_ms_per_step": blocks, "points_per_s": 2 * B * N / ms * 1e3,
                "points_note": "source + target clouds of one step",
                "components_us": {n: round(1e3 * c, 1) for n, c in zip(names, comp)},
                "components_note": "HIP events on the launch stream, mean of %d steps (host-side gaps inside a component included)" % nprof,
                "unmodified_trainer_label_loop_ms": shim_loop_ms,
                "unmodified_trainer_label_loop_note": "the trainer's own per-cloud pcl loop (trainer.py:525-531) served by mlsp_amd/shims/pcl.py: "
                                                      "%d host round trips; the batched device entry (mlsp.estimate_normals) is what `components_us` times" % B,
                "cpu_numpy_restatements_ms_per_step": {"cal_density (radius count)": round(1e3 * t_den, 1), "normals k=20 (kNN + PCA)": round(1e3 * t_nrm, 1),
                                                       "deform_input x2": round(1e3 * t_def, 1), "PCM.mix_shapes (FPS)": round(1e3 * t_mix, 1)},
                "cpu_note": "oracle/labels_np.py + oracle/ref_corrupt_np.py (python-pcl itself is absent: SURVEY 8c) on %d clouds, scaled to %d; "
                            "host threads as numpy / OpenMP default" % (nb, B)}
    finally:
        sys.path.remove(shim)
        for mod in ("PointDA", "PointDA.Models", "MLSP", "MLSP.mlsp", "MLSP.PCM", "pcl"):
            sys.modules.pop(mod, None)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher around it: THIS process (which has not touched a GPU and never will) starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <argv>` as a child, passes rank 0's JSON line through and
    returns the child's exit code (non-zero if any rank failed or no line came out).  The reference's own mechanism is nn.DataParallel
    inside one process (PointDA/trainer.py:251-252); one process per GPU over RCCL replaces it."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        ln = ln.rstrip("\n")
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)          # anything else the ranks print is not the result line
    rc = proc.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks exited cleanly but printed no result line", file=sys.stderr)
        rc = 1
    if line is not None:
        got = json.loads(line).get("n_gpus")
        if got != n:
            print("bench.py: result line reports n_gpus=%r, asked for %d" % (got, n), file=sys.stderr)
            rc = rc or 1
        print(line)
    return rc


def stub_ranks(a, world, rank):
    """Launcher / rendezvous / result-line check without a GPU (tests/test_bench_launcher_cpu.py): the ranks meet over gloo, a "step" is a
    1 ms sleep, the line has the fields of the real one.  Never a measurement."""
    import torch.distributed as dist
    if os.environ.get("MLSP_BENCH_STUB_FAIL_RANK") == str(rank):
        raise SystemExit(3)                      # (the test of "a failing rank fails the launcher")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(1e-3)
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    if rank == 0:
        print(json.dumps({"metric": "points/sec fwd+bwd, DGCNN+MLSP B=32 N=1024 k=20", "value": B_PER_GPU * NPTS * world * a.steps / dt,
                          "unit": "points/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "stub (no GPU work)",
                          "config": {"workload": "launcher stub", "parallelism": "dp%d" % world}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps; the median block is reported")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling: this many clouds in total, split over the ranks (SURVEY 8d config (2) secondary: 32). "
                         "Default 0 = weak scaling, 32 clouds per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[3] / configs[4] runs after the headline")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the comparison blocks with every GEMM on the f32 MFMA")
    ap.add_argument("--stub", action="store_true", help="launcher self-test: gloo ranks, no GPU, no kernels (never a measurement)")
    a = ap.parse_args()

    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # no launcher around us: become one.  Nothing in this process has initialised the GPU (and nothing will).
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    if a.stub:
        return stub_ranks(a, world, rank)

    import torch.distributed as dist
    # MLSP_BENCH_FORCE_DIST=1: a one-rank RCCL group with the exchange forced on (pack -> all-reduce -> divide at world size 1), so that the
    # whole N > 1 code path of this file -- and its `distributed` self-check record -- runs on a single-GPU box
    force_dist = world == 1 and bool(os.environ.get("MLSP_BENCH_FORCE_DIST"))
    distributed = world > 1 or force_dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # RCCL prints a version banner on STDOUT when its first communicator comes up: keep rank 0's stdout to the ONE JSON line by pointing
        # fd 1 at stderr while the group is created and the first collective runs
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.all_reduce(torch.zeros(1, device=dev))
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    n_gpus = world
    if a.global_batch:
        if a.global_batch % n_gpus:
            raise SystemExit("--global-batch must be a multiple of the number of ranks")
        b_local, scaling = a.global_batch // n_gpus, "strong"
    else:
        b_local, scaling = B_PER_GPU, "weak"

    from mlsp_amd import Models, mlsp, _lib, functional as Fh
    from mlsp_amd.ddp import FlatGradSync
    lib = _lib.load()
    # One process per GPU: autograd's per-device worker threads have nothing to run in parallel, and handing every backward node to
    # another thread costs 0.4 ms of the 3.2 ms the host needs to enqueue a step (tools/r5/host_profile.py: 3.17 -> 2.77 ms on the
    # host-bound probe).  The backward then runs on the calling thread; results are identical.  (recommended with
    # torchrun; nn.DataParallel -- several devices in ONE process -- keeps the default.)
    torch.autograd.set_multithreading_enabled(False)
    args = make_args()
    torch.manual_seed(0)                                   # identical replicas on every rank
    model = Models.DGCNN(args).to(dev).train()
    sync = FlatGradSync(model, align=4, force=force_dist)   # 16-byte aligned gradient views: FlatAdam reads the packed gradients in place
    opt = sync.wrap(make_adam(model.parameters()))                                            # trainer.py:258-259
    batch = synth_batch(b_local, NPTS, dev, seed=1000 + rank)

    def one_step():
        return gpu_step(model, mlsp, args, batch, opt)

    for _ in range(a.warmup):
        one_step()

    import gc

    def timed_block():
        """EXACTLY a.steps steps between two (barrier + synchronize) brackets; max over ranks.  Python's cyclic garbage collector runs
        BEFORE the bracket and is paused inside it (as `timeit` does): a generation-2 pass inside a block is host time of the interpreter,
        not of the step (it showed as a 10.8 ms block among 5.1 ms ones)."""
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        try:
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                last = one_step()
            torch.cuda.synchronize()
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t0
        finally:
            gc.enable()
        if distributed:
            t = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = t.item()
        return dt_, last

    # After --warmup steps the clocks / allocator / caches of a fresh process are still ramping (round 2: first block 10.5 ms against 5.75):
    # untimed SETTLE blocks run until two consecutive ones agree to 1 % (at most 4); only then do the timed blocks start.  Every rank
    # takes the same decision (block times are max-reduced over the ranks).
    settle, prev = [], None
    for _ in range(4):
        dt_s, _ = timed_block()
        settle.append(dt_s)
        if prev is not None and abs(dt_s - prev) <= 0.01 * prev:
            break
        prev = dt_s

    # One block of a.steps steps lasts ~0.1 s: the block is repeated and the MEDIAN block is the headline (a single short
    # sample is at the mercy of clock ramps and of whatever else the host does); min/max go into `blocks_ms_per_step`.
    blocks = []
    for _ in range(max(1, a.repeats)):
        dt_b, loss = timed_block()
        blocks.append(dt_b)
    dt = sorted(blocks)[len(blocks) // 2]
    assert torch.isfinite(loss).item()

    # HIP events around every launch of the priced kernel families: a separate, untimed block AFTER the timing
    prof_steps = 3
    rows, prof, prof_dt = profiled_steps(lib, one_step, prof_steps)

    # launches counted in THIS run (torch.profiler device records; untimed, rank 0 only -- no collectives skipped: every rank runs the steps)
    counted = None
    count_steps = 4
    if rank == 0:
        counted = count_launches(one_step, count_steps)
    else:
        for _ in range(count_steps):
            one_step()
    torch.cuda.synchronize()

    # N > 1: what a SCALE record needs to verify itself -- the world as RCCL sees it, collectives per optimizer step, the gradient
    # all-reduce timed alone (HIP events on the stream the collective is enqueued from), the device of every rank
    dist_info = None
    if distributed:
        steps_so_far = getattr(opt, "flat_steps", None)
        names = [None] * world
        dist.all_gather_object(names, "%s (cuda:%d)" % (torch.cuda.get_device_name(dev), local_rank))
        ar = []
        for _ in range(12):
            torch.cuda.synchronize()
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(sync._bucket, op=dist.ReduceOp.SUM)
            e1.record()
            e1.synchronize()
            ar.append(e0.elapsed_time(e1))
        t_ar = torch.tensor([sorted(ar[2:])[len(ar[2:]) // 2]], dtype=torch.float64, device=dev)
        dist.all_reduce(t_ar, op=dist.ReduceOp.MAX)
        nbytes = sync._bucket.numel() * 4
        dist_info = {"backend": dist.get_backend(), "world_size_seen_by_backend": dist.get_world_size(), "devices": names,
                     "collectives_per_step": sync.collectives / max(1, sync.steps),
                     "collectives": sync.collectives, "optimizer_steps": sync.steps, "optimizer_flat_steps": steps_so_far,
                     "allreduce_bytes": nbytes, "allreduce_alone_ms": t_ar.item(),
                     "allreduce_alone_busbw_GBs": 2.0 * (world - 1) / world * nbytes / (t_ar.item() * 1e-3) / 1e9,
                     "note": "all-reduce of the %d-byte gradient bucket alone: median of 10 (after 2 warm-up), HIP events around the call on "
                             "the enqueueing stream, max over ranks; bus bandwidth = 2 (n - 1) / n x bytes / time" % nbytes}

    # the same step with every GEMM on the f32 MFMA, for the record (not the headline): one settle block + the same number of blocks
    f32_leg, f32_blocks, x6_leg, x6_blocks = None, [], None, []
    if Fh.gemm_precision.current in ("bf16x6", "f16x3") and not a.no_fp32_leg:
        with Fh.gemm_precision("fp32"):
            timed_block()
            f32_blocks = [timed_block()[0] for _ in range(max(1, a.repeats))]
        f32_leg = sorted(f32_blocks)[len(f32_blocks) // 2]
        if Fh.gemm_precision.current == "f16x3":                     # ... and on the six-product bf16 split (the round 3-5 default)
            with Fh.gemm_precision("bf16x6"):
                timed_block()
                x6_blocks = [timed_block()[0] for _ in range(max(1, a.repeats))]
            x6_leg = sorted(x6_blocks)[len(x6_blocks) // 2]

    if rank == 0:
        pts = b_local * NPTS * n_gpus * a.steps
        value = pts / dt
        out = {"metric": "points/sec fwd+bwd, DGCNN+MLSP B=32 N=1024 k=20", "value": value, "unit": "points/s",
               "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
               "blocks_ms_per_step": {"n": len(blocks), "median": 1e3 * dt / a.steps, "min": 1e3 * min(blocks) / a.steps,
                                      "max": 1e3 * max(blocks) / a.steps,
                                      "untimed_settle_blocks": [round(1e3 * t / a.steps, 3) for t in settle]},
               "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "timing_note": "cyclic GC collected before and paused inside every timed block (timeit's convention)",
               "config": {"workload": "DGCNN encoder + 3 MLSP heads + losses, fwd+bwd+Adam, B=%d/GPU N=1024 k=20 fp32 "
                                      "(BASELINE.json configs[1]), dropout 0.5, BN train" % b_local,
                          "global_batch": b_local * n_gpus, "points_per_cloud": NPTS, "k": K_NN,
                          "parallelism": "dp%d" % n_gpus, "grad_allreduce": "1 x flat 18.2 MB fp32 per step (RCCL); no host-side exchange (presence='uniform', verified in-band)",
                          "gemm_products": "%s (f16x3 = fp32 operands, scaled by a per-workgroup power of two, split into two f16 pieces (11 + 11 "
                                           "bits), three piece products, fp32 accumulation; bf16x6 = three bf16 pieces, six piece products.  "
                                           "Either way an fp32 computation: error vs float64 below the f32-MFMA chain's, tests/test_gpu_kernels.py::"
                                           "test_gemm_split_bf16_accuracy holds both to the same bar; MLSP_GEMM_PRECISION=fp32 runs every GEMM on "
                                           "the f32 MFMA, =bf16x6 on the six-product split)" % Fh.gemm_precision.current}}
        if x6_leg is not None:
            out["bf16x6_split"] = {"ms_per_step": 1e3 * x6_leg / a.steps, "value": pts / x6_leg,
                                   "note": "the same step with the GEMM family on the six-product bf16 split (gemm_precision('bf16x6'), the "
                                           "round 3-5 default): median of %d blocks of %d steps, same process" % (len(x6_blocks), a.steps)}
        if f32_leg is not None:
            out["fp32_mfma"] = {"ms_per_step": 1e3 * f32_leg / a.steps, "value": pts / f32_leg,
                                "note": "the same step with every GEMM on the f32 MFMA (gemm_precision('fp32')): median of %d blocks "
                                        "of %d steps, timed after the headline blocks in the same process" % (len(f32_blocks), a.steps)}
        P_ROWS = b_local * NPTS
        if prof and prof[1] > 0:
            # prof = [total ms of the GEMM family, launches, algorithmic FLOP summed over launches, algorithmic bytes];
            # rows[7] = the subset that ran on the bf16-split kernel (default mode "bf16x6"), the rest ran on the f32 MFMA kernels
            sp = rows[7]
            # counter profile OF THIS WORKLOAD and OF THE PRICED KERNEL (raises when the committed file has no row of it)
            traffic, traffic_src, _ = pmc_gemm_traffic("gemm_split_kernel" if sp[1] > 0 else "gemm_f32_kernel", "configs1")
            step_ms = 1e3 * dt / a.steps
            common_note = ("achieved = sum(2*M*N*K of the launches: algorithmic fp32 FLOP) / sum(HIP-event time of the launches), events on "
                           "the launch stream, %d untimed steps after the timed blocks (%.2f ms/step with the events armed)"
                           % (prof_steps, 1e3 * prof_dt))
            f32_ms, f32_n, f32_flop = prof[0] - sp[0], prof[1] - sp[1], prof[2] - sp[2]
            f32_entry = None
            if f32_n > 0 and f32_ms > 0:
                ach32 = f32_flop / (f32_ms * 1e-3) / 1e12
                f32_entry = {"kernel": "gemm_f32_kernel<*> / gemm_f32_n64_kernel (GEMM launches on the f32 MFMA: short K loops, N = 64, "
                                       "ragged tiles)" if sp[1] > 0 else "gemm_f32_kernel<*> (every fp32 MFMA GEMM launch: fwd, dgrad, wgrad)",
                             "bound": "mfma", "achieved": ach32, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach32 / PEAK_FP32_TFLOPS,
                             "launches_per_step": f32_n / prof_steps, "avg_us": 1e3 * f32_ms / f32_n, "us_per_step": 1e3 * f32_ms / prof_steps,
                             "share_of_step": f32_ms / prof_steps / step_ms, "note": common_note}
            if sp[1] > 0 and sp[0] > 0:
                ach = sp[2] / (sp[0] * 1e-3) / 1e12
                # HBM-side bytes (committed --pmc passes) and algorithmic bytes over the SAME launches: the split kernel's, by instantiation
                kinds = getattr(profiled_steps, "split_kinds", {})
                pmc_kind, pmc_src = pmc_split_traffic_by_kind("configs1")
                by_kind, alg_all, tr_all, n_all = {}, 0.0, 0.0, 0.0
                for kname, (kms, kn, kflop, kbytes) in kinds.items():
                    if kn <= 0:
                        continue
                    e = {"launches_per_step": kn / prof_steps, "avg_us": 1e3 * kms / kn, "achieved_tflops": kflop / (kms * 1e-3) / 1e12,
                         "algorithmic_bytes_per_launch": kbytes / kn}
                    if kname in pmc_kind:
                        e["traffic_per_launch"] = pmc_kind[kname][0]
                        e["traffic_over_algorithmic"] = pmc_kind[kname][0] / (kbytes / kn)
                        tr_all += pmc_kind[kname][0] * kn
                    alg_all += kbytes
                    n_all += kn
                    by_kind[kname] = e
                if pmc_kind and n_all:
                    traffic, traffic_src = tr_all / n_all, pmc_src
                half = getattr(profiled_steps, "split_half", [0.0, 0.0, 0.0, 0.0])
                sp_peak, sp_pp, sp_peak_note = split_peak(half[1], sp[1])
                out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": sp_peak, "unit": "TFLOP/s", "frac": ach / sp_peak,
                                   "traffic": traffic, "traffic_source": "committed profile (%s), not measured in this run" % traffic_src,
                                   "algorithmic_bytes_per_launch": alg_all / n_all if n_all else None,
                                   "traffic_note": "HBM-side bytes per launch (2*FETCH_SIZE + WRITE_SIZE, %s) and algorithmic bytes per launch (A + B + C as the "
                                                   "formulation reads them: a gradient formed on the fly is a (d', y) operand PAIR, a beta = 1 launch reads C too; "
                                                   "split-K slabs written once and read once by the reduce) over the SAME launches: the ones that ran on "
                                                   "gemm_split_kernel, weighted by this run's launch mix; `by_kind` has them per instantiation "
                                                   "(fwd / dgrad / wgrad)" % traffic_src,
                                   "by_kind": by_kind,
                                   "kernel": "gemm_split_kernel<*> (fp32-accurate products as 16-bit MFMA piece products: fwd, dgrad, wgrad)",
                                   "peak_note": sp_peak_note, "launches_on_f16_pieces_per_step": half[1] / prof_steps,
                                   "executed_16bit_tflops": sp_pp * ach, "vs_f32_mfma_peak": ach / PEAK_FP32_TFLOPS,
                                   "launches": int(sp[1]), "launches_per_step": sp[1] / prof_steps, "avg_us": 1e3 * sp[0] / sp[1],
                                   "share_of_step": sp[0] / prof_steps / step_ms,
                                   "note": common_note + "; the kernel is clock(DVFS)-limited on real operands: the same launches on zero-filled "
                                           "operands run 1.33x faster (tools/x6/lib_bench, DESIGN.md)"}
            else:
                out["roofline"] = dict(f32_entry, traffic=traffic, launches=int(f32_n), traffic_source="committed profile (%s), not measured in this run" % traffic_src,
                                       traffic_note="HBM-side bytes per launch (2*FETCH_SIZE + WRITE_SIZE), %s; algorithmic A+B+C bytes per "
                                                    "launch: %.0f" % (traffic_src, prof[3] / prof[1]))
                f32_entry = None
            # the other families north_star names: kNN / gather against HBM, kNN distance sweeps and the T-Net stage against the matrix peak
            knn3 = _kernel_entry("kNN C=3 (distance + select kernel + row norms), stages 0-1", "hbm", rows[1], prof_steps,
                                 "compulsory bytes (C+k)*4 per point: brute-force kNN on 3 channels is not HBM-bound; `second_roof` prices the "
                                 "same launches against the vector-ALU distance + select bound of SURVEY 8d")
            if knn3 is not None:
                # N^2 (C FMA + compare) per cloud -> per query N candidates x (C FMAs + the two norm terms + one compare) vector
                # lane-operations; the chip issues PEAK_FP32_TFLOPS / 2 of them per second (an FMA counts as two FLOP)
                lane_ops = 2 * P_ROWS * NPTS * (3 + 3)            # two launches per step (raw and transformed cloud), C = 3
                secs = knn3["us_per_step"] * 1e-6
                peak_ops = PEAK_FP32_TFLOPS / 2 * 1e12
                knn3["second_roof"] = {"bound": "valu", "achieved": lane_ops / secs / 1e12, "peak": peak_ops / 1e12, "unit": "T lane-op/s",
                                       "frac": lane_ops / secs / peak_ops,
                                       "note": "N*(C+3) vector lane-operations per query (C FMAs, two norm terms, one compare) -- the distance + "
                                               "select work a brute-force kNN cannot avoid; k-selection bookkeeping not counted"}
            ks = [knn3,
                  _kernel_entry("EdgeConv neighbour gather-reduce (edge_reduce_wide_kernel), 4 EdgeConv + T-Net conv1", "hbm", rows[4], prof_steps,
                                "compulsory bytes: u half + indices in, msel + s1 + arg slot out (17 B per point and channel)"),
                  _kernel_entry("kNN C=64 (knn6_prep_kernel<64> + knn6_kernel<64>), stages 2-3", "mfma", rows[2], prof_steps,
                                "algorithmic 2*N*C FLOP per point (one distance sweep) against the f32 MFMA peak; the kernel runs two split-bf16 "
                                "sweeps (3 bf16 products each) and resolves only the ambiguous survivors in canonical fp32"),
                  _kernel_entry("kNN C=128 (knn6_prep_kernel<128> + knn6_kernel<128>), stage 4", "mfma", rows[3], prof_steps,
                                "algorithmic 2*N*C FLOP per point (one distance sweep) against the f32 MFMA peak; two split-bf16 sweeps + exact "
                                "resolution of the ambiguous survivors"),
                  _kernel_entry("T-Net per-edge stage forward (tnet_edge_fwd3_kernel<20>: split products on the bf16 cores)", "mfma", rows[5], prof_steps,
                                "2*E*64*128 algorithmic FLOP, priced against the f32 MFMA peak (the kernel executes 6x that on the bf16 cores)"),
                  _kernel_entry("T-Net per-edge stage backward (tnet_edge_bwds_kernel: Gram form, dense split products on the bf16 cores; + prep / slab "
                                "reduce / finish)", "mfma", rows[6], prof_steps,
                                "reference FLOP 4*E*64*128, priced against the f32 MFMA peak (the kernel executes 1,152 bf16 MFMAs per 128-row "
                                "tile: 9x the reference FLOP count as bf16 piece products, 1.5x as six-product fp32 equivalents)")]
            out["roofline_kernels"] = [k for k in [f32_entry] + ks if k]
        else:
            out["roofline"] = {"bound": "mfma", "achieved": value * FLOP_PER_POINT / 1e12, "peak": PEAK_FP32_TFLOPS,
                               "unit": "TFLOP/s", "frac": value * FLOP_PER_POINT / 1e12 / PEAK_FP32_TFLOPS, "traffic": None,
                               "kernel": "whole step (algorithmic 28.19 MFLOP/pt)"}
        # Whole-step figures.  The kernels do NOT execute the reference's 28.19 MFLOP/pt: the folded EdgeConv, the per-cloud
        # head bias and the Gram-matrix backward remove ~60 % of them.  Executed MFMA work per step = the GEMM launches (HIP
        # events above) + the fused T-Net per-edge stage (64x128 MACs per edge, forward + two backward products) + the fp32
        # distance sweeps of the five kNN stages (channel counts padded to the MFMA tile): both passes of the two 3-channel
        # stages, the exact pass B only of the 64 / 64 / 128-channel stages (their pass A runs on the bf16 matrix cores and is
        # not counted as fp32 work).
        P = b_local * NPTS
        split_flop = rows[7][2] / prof_steps if prof and prof[1] > 0 else 0.0      # runs on the bf16 cores (six products each): not f32-MFMA work
        exec_flop = ((prof[2] / prof_steps - split_flop) if prof and prof[1] > 0 else 0.0) + 3 * 2.0 * P * K_NN * 64 * 128 \
            + 2.0 * P * NPTS * (2 * (4 + 4) + (64 + 64 + 128))
        step_s = dt / a.steps
        out["executed_tflops"] = exec_flop / step_s / 1e12
        out["executed_mfma_frac"] = out["executed_tflops"] / PEAK_FP32_TFLOPS
        half_ = getattr(profiled_steps, "split_half", [0.0, 0.0, 0.0, 0.0]) if prof and prof[1] > 0 else [0.0, 0.0, 0.0, 0.0]
        out["executed_16bit_tflops"] = (6 * (split_flop - half_[2] / prof_steps) + 3 * half_[2] / prof_steps) / step_s / 1e12   # the split GEMMs' piece products
        out["reference_flop_equivalent_frac"] = value / n_gpus * FLOP_PER_POINT / 1e12 / PEAK_FP32_TFLOPS
        if counted and "launches_per_step" in counted:
            out["launches_per_step"] = counted["launches_per_step"]
            out["launches_under_8us_per_step"] = counted["launches_under_8us_per_step"]
            out["launches_source"] = "counted"
            out["launches_counted"] = dict(counted, note="device kernel records of torch.profiler over %d untimed steps of THIS run, after the "
                                                         "timed blocks: every kernel of the process (libmlsp_hip.so's and torch's)" % counted["steps_counted"])
        else:
            out["launches_source"] = "not counted: torch.profiler delivered no device records on this box (%r)" % (counted,)
        if distributed:
            out["distributed"] = dist_info
        if n_gpus == 1 and not a.no_secondary:
            del model, opt, sync, batch
            out["secondary"] = secondary_workloads(lib, dev)
            out["secondary"].append(trainer_shaped_workload(dev))
        if n_gpus == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
