#!/usr/bin/env python3
"""bench.py — tree-evaluations/s of the SR fitness hot path on N MI355X (one process per GPU over RCCL).

    python bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json north_star / configs[2]): SymbolicRegression synthetic 10-var, GLOBAL population
1 000 000 trees x 1024 datapoints, max_tree_len 64, funcs + - * /.  It fits one GPU (512 MB of trees), so N = 1 runs all
of it; with N ranks the trees are split into N contiguous shards (rank r generates trees [r P/N, (r+1) P/N) with the
tree-index offset, so the union is the single-device forest) — "scaling": "strong".  A "step" is ONE pass of the hot
path: `tree_SR_fitness` over the rank's shard, inputs resident in HBM; the fitness pass has no data-path collective.
`value` = global_pop x datapoints x steps / max-over-ranks wall time.

With N > 1 and no torch.distributed environment, bench.py launches its N ranks itself (torch.distributed.run, one per
GPU, backend nccl = RCCL) and fails loudly when fewer than N GPUs are visible.

The same JSON line also carries
  configs1      BASELINE configs[1] (pop 100k PER GPU x 1024 datapoints, weak), same protocol: ms_per_step, tree-evals/s,
                generation_ms (fitness + DefaultSelection + DefaultCrossover + DefaultMutation on one shard)
  generation_ms_sharded  one generation of the GLOBAL population: local fitness + all-gather of the fitness values +
                all-gather of the survivor rows (RCCL) + selection + breeding of the local rows
  roofline      the dominant kernel (the threaded-code interpreter sr_tc_kernel): algorithmic bytes / its launch time
                (HIP events on the launch stream) against the HBM peak, and the fraction of the VALU issue rate it uses,
                computed from the handler histogram of the compiled population x the generated interpreter's per-handler
                instruction counts
  cpu_baseline  the CPU oracle (plain-C port of the reference algorithm, OpenMP) on this host's cores: a bounded sample of
                the headline workload, and configs[0] (XOR-3d, pop 5000, max_tree_len 32, 8 datapoints) next to the GPU's
                time for the same call
"""
import argparse
import ctypes
import hashlib
import json
import os
import socket
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

EXTRAS_TIMEOUT_S = 300        # N > 1: the measurements beyond the headline may take this long before every rank leaves
HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
GLOBAL_POP = 1_000_000
POP_PER_GPU = 100_000          # configs[1]
DATAPOINTS = 1024
VAR_LEN = 10
GP_LEN = 64
PREWARM = 40       # untimed passes in front of the warm-up steps: the device's clock ramp (see main)


def self_launch(args):
    """`python bench.py --gpus N` without a torch.distributed environment: become the launcher of N ranks."""
    have = torch.cuda.device_count()
    share = os.environ.get("EVOGP_BENCH_SHARE_GPU", "0") == "1"
    if have < args.gpus and not share:
        sys.exit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node (no oversubscription, no CPU fallback)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(cmd[0], cmd, env)


def sr_inputs(lo, pop, device):
    """SURVEY.md §8d synthetic inputs: trees [lo, lo + pop) of the forest tree_generate(keys=[42,0], max_layer_cnt=6,
    + - * /, consts {-1,0,1}) produces; X ~ U(-5,5) seed 1234; y = x0*x1 + x2*x3 - x4 + 0.5*x5^2."""
    from evogp_amd.tree import Forest, GenerateDescriptor

    desc = GenerateDescriptor(max_tree_len=GP_LEN, input_len=VAR_LEN, output_len=1, using_funcs=["+", "-", "*", "/"],
                              max_layer_cnt=6, const_samples=[-1, 0, 1])
    keys = torch.tensor([42, 0], dtype=torch.uint32, device=device)
    forest = Forest.random_generate(pop, desc, keys=keys, tree_index_offset=lo)
    rng = np.random.default_rng(1234)
    X = rng.uniform(-5, 5, (DATAPOINTS, VAR_LEN)).astype(np.float32)
    y = (X[:, 0] * X[:, 1] + X[:, 2] * X[:, 3] - X[:, 4] + 0.5 * X[:, 5] ** 2).astype(np.float32)[:, None]
    return forest, torch.from_numpy(X).to(device), torch.from_numpy(y).to(device), X, y


def source_sha():
    """Hash of the kernel sources: measurements stored in profiles/pmc_latest.json are only quoted for the code they
    were taken on."""
    h = hashlib.sha1()
    base = os.path.join(ROOT, "evogp_amd", "csrc")
    names = sorted(f for f in os.listdir(base) if f.endswith((".hip", ".hpp"))) + ["gen/gen_tc_asm.py"]
    for n in names:
        h.update(open(os.path.join(base, n), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(forest, X, y, device, budget_s=6.0):
    """The CPU oracle on this host: (a) a bounded sample of the headline workload (first S trees of rank 0's shard x all
    1024 datapoints, S sized from a probe for ~budget_s seconds); (b) BASELINE configs[0] exactly, beside the GPU."""
    from oracle.pyoracle import Oracle, depth2leaf, have_reference, roulette_uniform

    o = Oracle("port", native=True)     # compiled -march=native on this box where a compiler exists (else the x86-64-v3 build)
    n = min(forest.pop_size, 1_000_000)
    v = forest.batch_node_value[:n].cpu().numpy(); t = forest.batch_node_type[:n].cpu().numpy(); s = forest.batch_subtree_size[:n].cpu().numpy()
    # warm-up (thread pool, page faults), then whole passes over the sample until ~budget_s seconds of wall time are spent
    probe = 4096
    o.sr_fitness(v[:probe], t[:probe], s[:probe], X, y, True, 0)
    t0 = time.perf_counter(); o.sr_fitness(v[:probe * 8], t[:probe * 8], s[:probe * 8], X, y, True, 0); dt = time.perf_counter() - t0
    sample = int(min(n, max(probe * 8, probe * 8 * (budget_s / 2) / max(dt, 1e-6))))
    reps, t0 = 0, time.perf_counter()
    while reps == 0 or (time.perf_counter() - t0 < budget_s and reps < 50):
        o.sr_fitness(v[:sample], t[:sample], s[:sample], X, y, True, 0); reps += 1
    dt = (time.perf_counter() - t0) / reps
    out = {
        "value": sample * X.shape[0] / dt,
        "unit": "tree-evals/s",
        "cores": int(o.threads_used),
        "kind": "port",
        "sample": f"first {sample} trees of the rank-0 shard x {X.shape[0]} datapoints, {reps} passes of {dt:.2f} s, plain-C oracle ({o.flags}, "
                  f"IEEE fp32, no FMA contraction), OpenMP over trees: {int(o.threads_used)} threads on a host with "
                  f"{os.cpu_count()} logical cpus ({len(os.sched_getaffinity(0))} usable by this process)",
        "node_evals_per_s": float(s[:sample, 0].astype(np.int64).sum()) * X.shape[0] / dt,
    }
    # one core: the port, and the REFERENCE's own device code compiled for the host (oracle/_ref, serial harness) on the same
    # trees -- anchors "port ~ reference speed" (VERDICT r02 #6).  ~1 s each.
    one = {}
    k = 2000
    try:
        o.sr_fitness(v[:64], t[:64], s[:64], X, y, True, 1)
        t0 = time.perf_counter(); o.sr_fitness(v[:k], t[:k], s[:k], X, y, True, 1); d1 = time.perf_counter() - t0
        one["port_1core"] = {"value": k * X.shape[0] / d1, "unit": "tree-evals/s", "cores": 1, "kind": "port", "sample": f"first {k} trees x {X.shape[0]} datapoints, one pass of {d1:.2f} s"}
        if have_reference():
            r = Oracle("reference")
            r.sr_fitness(v[:64], t[:64], s[:64], X, y, True)
            t0 = time.perf_counter(); fr = r.sr_fitness(v[:k], t[:k], s[:k], X, y, True); d2 = time.perf_counter() - t0
            fp = o.sr_fitness(v[:k], t[:k], s[:k], X, y, True, 1)
            same = bool(np.array_equal(np.where(np.isnan(fr), 0, fr).view(np.uint32), np.where(np.isnan(fp), 0, fp).view(np.uint32)))
            one["reference_1core"] = {"value": k * X.shape[0] / d2, "unit": "tree-evals/s", "cores": 1, "kind": "reference, 1 core",
                                      "sample": f"the same {k} trees; the reference's forward.cu compiled for the host (hipcc --offload-host-only -O2, oracle/build_ref.py), "
                                                f"one pass of {d2:.2f} s; its harness emulates the 1024-thread blocks serially (every 'thread' copies the tree, forward.cu:284-287)",
                                      "fitness_bits_equal_port": same}
    except Exception as exc:
        one["error"] = repr(exc)[:200]
    out.update(one)
    # configs[0]: XOR-3d SymbolicRegression, pop 5000, max_tree_len 32, 8 datapoints (README.md:130-175), CPU-runnable
    pop1 = 5000
    xs = np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], np.float32)
    ys = (xs[:, 0].astype(int) ^ xs[:, 1].astype(int) ^ xs[:, 2].astype(int)).astype(np.float32)[:, None]
    f1 = o.generate(pop1, 32, 3, 1, 0.5, 0.5, [42, 0], depth2leaf(4), roulette_uniform([1, 2, 3, 4]), [-1, 0, 1])
    reps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        o.sr_fitness(*f1, xs, ys, True, 0); reps += 1
    cpu_s = (time.perf_counter() - t0) / reps
    from evogp_amd.tree import Forest

    gf = Forest(3, 1, *(torch.from_numpy(a).to(device) for a in f1))
    gx, gy = torch.from_numpy(xs).to(device), torch.from_numpy(ys).to(device)
    for _ in range(5):
        gf.SR_fitness(gx, gy, True, "auto")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        gf.SR_fitness(gx, gy, True, "auto")
    torch.cuda.synchronize(); gpu_s = (time.perf_counter() - t0) / 50
    out["configs0"] = {
        "workload": "BASELINE configs[0]: XOR-3d SR, pop 5000, max_tree_len 32, 8 datapoints, one fitness pass",
        "cpu_tree_evals_per_s": pop1 * 8 / cpu_s, "cpu_ms": cpu_s * 1e3, "cpu_threads": int(o.threads_used),
        "gpu_tree_evals_per_s": pop1 * 8 / gpu_s, "gpu_ms": gpu_s * 1e3,
        "mean_tree_len": float(f1[2][:, 0].mean()),
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-pop", type=int, default=GLOBAL_POP, help="trees of the headline workload, split over the ranks")
    ap.add_argument("--pop-per-gpu", type=int, default=POP_PER_GPU, help="trees per rank of the configs[1] line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline steps (profiling runs: every launch of the fitness kernels then has the headline's shape)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # does not return

    # (the host driver only supports dmabuf IPC; RCCL's hipIpcGetMemHandle fails without this -- read when the HIP runtime starts)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    # EVOGP_BENCH_SHARE_GPU=1 is a functional check of the multi-rank code path on a box with fewer GPUs than ranks: the
    # ranks share the visible devices and talk over gloo (RCCL refuses two ranks on one device).  Never a measurement.
    share_gpu = os.environ.get("EVOGP_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank %= torch.cuda.device_count()
    assert local_rank < torch.cuda.device_count(), f"rank {rank}: no GPU {local_rank} on this node"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("EVOGP_BENCH_BACKEND", "gloo" if share_gpu else "nccl")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            # nccl = RCCL.  The headline step has no data-path collective (barriers and two scalar reductions around the timed region),
            # so a node whose RCCL does not come up still gets measured: the ranks fall back to gloo for those, and the line says so.
            try:
                dist.init_process_group("nccl", device_id=device)
                probe = torch.ones(1, device=device)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                assert int(probe.item()) == world, f"all_reduce over {world} ranks gave {probe.item()}"
            except Exception as exc:   # noqa: BLE001 -- whatever RCCL raises
                print(f"[bench] rank {rank}: RCCL did not come up ({exc!r:.300}); barriers and reductions go over gloo", file=sys.stderr, flush=True)
                try:
                    dist.destroy_process_group()
                except Exception:   # noqa: BLE001
                    pass
                store = f"/tmp/evogp_bench_store_{os.environ.get('TORCHELASTIC_RUN_ID', 'run')}_{os.environ.get('MASTER_PORT', '0')}"
                dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=world)
                backend = "gloo (RCCL initialisation failed)"
        world = dist.get_world_size()

    import evogp_amd  # noqa: F401
    from evogp_amd import _lib
    from evogp_amd.tree import set_default_device

    set_default_device(device)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        tmax = torch.tensor([x], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    def sum_over_ranks(x):
        if dist is None:
            return x
        tsum = torch.tensor([x], dtype=torch.int64, device=device)
        dist.all_reduce(tsum)
        return int(tsum.item())

    def reference_op(forest, Xd, yd):
        """the reference's OWN operator, as its unchanged Python calls it (/root/reference/src/evogp/tree/forest.py:340-351): no function mask"""
        return torch.ops.evogp_cuda.tree_SR_fitness(forest.pop_size, Xd.shape[0], forest.max_tree_len, forest.input_len, forest.output_len, True,
                                                    forest.batch_node_value, forest.batch_node_type, forest.batch_subtree_size, Xd, yd, 4)

    def timed_steps(forest, Xd, yd, warmup, steps, call=None):
        """W untimed + exactly K timed fitness passes, barrier + synchronize on both sides, max over ranks;
        also the average duration of one call from a HIP event pair on the launch stream (this rank).
        call: the pass (default Forest.SR_fitness, which hands the engine the forest's function mask)"""
        call = call or (lambda f, X_, y_: f.SR_fitness(X_, y_, True, "auto"))
        for _ in range(warmup):
            call(forest, Xd, yd)
        barrier()
        _lib.check(_lib.lib.evogp_hip_timer_begin(stream), "timer_begin")
        t0 = time.perf_counter()
        for _ in range(steps):
            call(forest, Xd, yd)
        ev_ms = ctypes.c_float(0)
        _lib.check(_lib.lib.evogp_hip_timer_end(stream, ctypes.byref(ev_ms)), "timer_end")
        barrier()
        return max_over_ranks(time.perf_counter() - t0), ev_ms.value / steps

    # ---- headline: the global population, split over the ranks ------------------------------------------------------
    P = args.global_pop
    lo, hi = rank * P // world, (rank + 1) * P // world
    pop = hi - lo
    forest, Xd, yd, X, y = sr_inputs(lo, pop, device)
    total_nodes = int(forest.batch_subtree_size[:, 0].to(torch.int64).sum())
    # The device reaches its steady clocks only after ~25 ms of work (profiles/r03_first_calls.log: calls
    # 1-4 of a fresh process take 1.28 ms, 5-9 1.23, 10-19 1.18, every later one 1.14): the W warm-up steps the driver asks for
    # (5) end inside that ramp.  PREWARM untimed passes of the same step precede them, so that W + K measure the state a run of
    # thousands of generations is in; the count is reported in the JSON line (`device_prewarm_calls`).
    # (for the record, the same W + K steps are timed once BEFORE those passes: `cold_start.ms_per_step` in the JSON line is what a
    # process that has done nothing else yet measures)
    cold_elapsed, _ = timed_steps(forest, Xd, yd, args.warmup, args.steps)
    for _ in range(PREWARM):
        forest.SR_fitness(Xd, yd, True, "auto")
    torch.cuda.synchronize()
    elapsed, call_ms = timed_steps(forest, Xd, yd, args.warmup, args.steps)
    all_nodes = sum_over_ranks(total_nodes)
    # the same steps once more with per-stage events inside the call (compiler | interpreter | follow-ups): the duration of the
    # dominant kernel itself.  Kept out of the timed region above.
    _lib.check(_lib.lib.evogp_hip_debug_profile(1), "profile on")
    for _ in range(args.steps):
        forest.SR_fitness(Xd, yd, True, "auto")
    stage = (ctypes.c_float * 3)()
    ncalls = ctypes.c_int(0)
    _lib.check(_lib.lib.evogp_hip_debug_profile_read(stage, ctypes.byref(ncalls)), "profile read")
    _lib.check(_lib.lib.evogp_hip_debug_profile(0), "profile off")
    stage_ms = {"program_compiler": stage[0], "interpreter": stage[1], "follow_ups": stage[2], "calls": ncalls.value}

    # the same steps through torch.ops.evogp_cuda.tree_SR_fitness -- the operator a drop-in under the reference's own Python is called
    # through: same protocol, right behind the headline's
    for _ in range(3):
        reference_op(forest, Xd, yd)
    ref_elapsed, ref_call_ms = timed_steps(forest, Xd, yd, args.warmup, args.steps, call=reference_op)
    torch.cuda.synchronize()
    ref_record_bytes = int(evogp_amd.program_buffer_bytes())
    ref_same = bool(torch.equal(reference_op(forest, Xd, yd).view(torch.int32), forest.SR_fitness(Xd, yd, True, "auto").view(torch.int32)))


    # VALU issue: handler histogram of the compiled shard x instruction counts of the generated interpreter
    valu = None
    try:
        nh = _lib.lib.evogp_hip_debug_tc_nhandlers()
        hist = torch.zeros(2 * nh, dtype=torch.int64, device=device)
        _lib.check(_lib.lib.evogp_hip_debug_tc_histogram(pop, ctypes.c_void_p(hist.data_ptr()), 2 * nh, stream), "tc_histogram")
        hist = hist.cpu().numpy()
        table = json.load(open(os.path.join(ROOT, "evogp_amd", "lib", "tc_handlers.json")))
        mode = evogp_amd.get_sr_division()
        info = table[f"K8_{mode}"]
        assert info["nhandlers"] == nh
        per = np.zeros(nh); names = [None] * nh; clk = np.zeros(nh)
        for name, h in info["handlers"].items():
            per[h["id"]] = h["valu"]; names[h["id"]] = name; clk[h["id"]] = h.get("valu_clk", 4 * h["valu"])
        words = hist[:nh] + hist[nh:]
        tiles = (DATAPOINTS + 64 * info["K"] - 1) // (64 * info["K"])
        wave_insts = float((words * per).sum()) * tiles          # VALU instructions issued per launch (one per 64 lanes)
        props = torch.cuda.get_device_properties(device)
        clock_hz = float(getattr(props, "clock_rate", 2_400_000)) * 1e3
        cus = props.multi_processor_count
        peak = cus * 4 * clock_hz / 4.0                           # 4 SIMDs per CU, one 64-lane VALU instruction per 4 clocks each
        peak_spec = cus * 4 * clock_hz / 2.0                      # MI355X_MICROARCH.md: a SIMD issues a VALU instruction over 2 cycles (157.3 TFLOP/s fp32 vector / 128)
        kernel_s = stage_ms["interpreter"] / 1e3
        top = sorted(((int(w), names[i]) for i, w in enumerate(words) if w), reverse=True)[:8]
        # the SQ counters of the committed PMC run (same sources): how busy the pipe really is
        pipe_busy = clk_per_inst = None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            if pj.get("source_sha") == source_sha() and pj.get("pop_per_launch") == pop:
                c = pj["kernels"]["sr_tc_kernel"]["per_launch"]
                pipe_busy = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (cus * 4) / (kernel_s * clock_hz)
                clk_per_inst = c["SQ_ACTIVE_INST_VALU"] * 4.0 / c["SQ_INSTS_VALU"]
        except Exception:
            pass
        valu = {
            "frac": wave_insts / kernel_s / peak if kernel_s > 0 else None,
            "frac_of_spec": wave_insts / kernel_s / peak_spec if kernel_s > 0 else None,
            "valu_pipe_busy": pipe_busy, "clocks_per_valu_inst": clk_per_inst,
            "valu_per_row_per_word": wave_insts / (float(words.sum()) * tiles * info["K"]) if words.sum() else None,
            "valu_insts_per_launch": wave_insts, "peak_insts_per_s": peak, "peak_insts_per_s_spec": peak_spec, "cus": cus, "clock_hz": clock_hz,
            "program_words": int(words.sum()), "words_per_tree": float(words.sum()) / pop, "passes_per_tree": tiles,
            "top_handlers": [f"{n}:{w}" for w, n in top],
            "handlers": {names[i]: int(w) for i, w in enumerate(words) if w},
            "valu_clocks_model": {"per_simd": float((words * clk).sum()) * tiles / (cus * 4),
                                  "frac_of_kernel": float((words * clk).sum()) * tiles / (cus * 4) / (kernel_s * clock_hz) if kernel_s > 0 else None,
                                  "division_share": float(sum(words[i] * clk[i] for i in range(nh) if names[i] and names[i].startswith("div"))) / max(float((words * clk).sum()), 1.0),
                                  "what": "issue clocks of the VALU by instruction class (2 for a VOP2 add / sub / mul / mov on registers, 8 for a "
                                          "transcendental, 4 for the rest; gen_tc_asm.py count_path) summed over the program words, per SIMD, and as a "
                                          "fraction of the interpreter's clocks"},
            "what": "sum over program words of the handler's VALU instruction count (evogp_amd/lib/tc_handlers.json, from the generator) "
                    "x datapoint tiles, / interpreter launch time.  frac: against one instruction per 4 clocks and SIMD (the rate the VOP3 / 3-source / "
                    "SGPR-source class runs at: scripts/ubench/valu_rates.hip measures 4.1 clocks); frac_of_spec: against the chip's issue rate, one per 2 "
                    "clocks (157.3 TFLOP/s fp32 vector).  valu_pipe_busy = SQ_ACTIVE_INST_VALU x 4 / SIMDs / kernel clocks and clocks_per_valu_inst = "
                    "SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU from the PMC run of the same sources (profiles/pmc_latest.json; null when it is of other "
                    "sources): the pipe is nearly full of slow-class instructions, so the lever is their number and class, not latency",
        }
    except Exception as exc:
        valu = {"frac": None, "error": repr(exc)[:300]}

    def emit(extras, with_cpu_baseline=True):
        """rank 0's ONE JSON line: the headline fields (all known before the extras start) + whatever extras exist"""
        with emit_lock:
            if emitted:
                return
            emitted.append(True)
        evals = float(P) * DATAPOINTS * args.steps
        kernel_s = stage_ms["interpreter"] / 1e3 if stage_ms["calls"] else call_ms / 1e3
        alg_bytes = 6.0 * total_nodes + 2.0 * pop + 4.0 * DATAPOINTS * (VAR_LEN + 1) + 4.0 * pop  # SURVEY.md §8d, this rank's launch
        achieved = alg_bytes / kernel_s / 1e9
        traffic, traffic_call, traffic_note = None, None, "no PMC record for this build"
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                if pj.get("source_sha") == source_sha() and pj.get("pop_per_launch") == pop:
                    traffic = pj.get("sr_tc_kernel_hbm_bytes_per_launch")
                    traffic_call = pj.get("call_hbm_bytes")
                    traffic_note = f"FETCH_SIZE + WRITE_SIZE of the interpreter kernel, rocprofv3 --pmc, measured on this source ({pj.get('source_sha')}): {pj.get('files')}"
                else:
                    traffic_note = (f"profiles/pmc_latest.json was measured on source {pj.get('source_sha')} / {pj.get('pop_per_launch')} trees per launch, "
                                    f"this is {source_sha()} / {pop}: not quoted")
            except Exception:
                pass
        out = {
            "metric": "tree_evals_per_s",
            "value": evals / elapsed,
            "unit": "tree-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1000.0,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not share_gpu else "synthetic; FUNCTIONAL CHECK ONLY: ranks share a GPU over gloo",
            "device_prewarm_calls": PREWARM,
            "cold_start": {"ms_per_step": cold_elapsed / args.steps * 1e3,
                           "what": "the same W warm-up + K timed steps run once before the untimed passes: a fresh process, device clocks still ramping"},
            "config": {
                "workload": f"BASELINE north_star / configs[2] shape: SymbolicRegression synthetic 10-var, GLOBAL pop={P} x 1024 datapoints, "
                            "max_tree_len=64, funcs + - * /, one tree_SR_fitness pass over every rank's shard per step",
                "global_pop": P, "pop_per_gpu": pop, "datapoints": DATAPOINTS, "var_len": VAR_LEN, "max_tree_len": GP_LEN,
                "mean_tree_len": all_nodes / P, "sharding": f"trees x{world} (contiguous shards, tree-index offset), no data-path collective in the step",
                "ranks": world, "backend": backend,
                # (both numbers: the timed K steps follow PREWARM untimed passes -- the device's clock ramp --; the same W + K steps of a
                # fresh process are cold_start.ms_per_step)
                "device_prewarm_calls": PREWARM, "ms_per_step_after_prewarm": elapsed / args.steps * 1000.0, "ms_per_step_fresh_process": cold_elapsed / args.steps * 1e3,
                # the same K steps through the reference's own operator (no function mask): what a drop-in under the reference's unchanged Python runs
                "ms_per_step_reference_op": ref_elapsed / args.steps * 1000.0,
                "reference_op": {"op": "torch.ops.evogp_cuda.tree_SR_fitness", "ms_per_step": ref_elapsed / args.steps * 1000.0, "call_ms_events": ref_call_ms,
                                 "gap_vs_masked_call": ref_elapsed / elapsed - 1.0, "record_buffer_bytes": ref_record_bytes,
                                 "fitness_words_equal_masked_call": ref_same},
            },
            "node_evals_per_s": float(all_nodes) * DATAPOINTS * args.steps / elapsed,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_call": traffic_call, "traffic_note": traffic_note,
                "kernel": "sr_tc_kernel<8,false,2> (threaded-code interpreter, short division), rank 0's launch; algorithmic bytes = SURVEY.md §8d "
                          "(6 B per live node + size + dataset + fitness) x the trees of the launch",
                "kernel_ms": kernel_s * 1e3, "algorithmic_bytes": alg_bytes,
                "call_ms": call_ms, "stage_ms": stage_ms,
                "frac_whole_call": alg_bytes / (call_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                "valu_issue": valu,
                "note": "stack-machine interpreter: ~0.16 algorithmic B per tree-eval at D=1024, so the HBM fraction is small by construction "
                        "(SURVEY.md §7.3-1); the binding resource is VALU issue (valu_issue.frac) plus per-instruction dispatch latency. "
                        "kernel_ms: HIP events around the interpreter launch inside the call (a second pass of the same steps); "
                        "call_ms: event pair around the timed steps (compiler + interpreter + follow-up launches)",
            },
        }
        out.update(extras)
        if with_cpu_baseline and not args.no_cpu_baseline and world == 1 and not args.headline_only:
            try:
                out["cpu_baseline"] = cpu_baseline(forest, X, y, device)
            except Exception as exc:
                out["cpu_baseline"] = {"value": None, "error": repr(exc)[:300]}
        print(json.dumps(out), flush=True)

    # Everything below is extra: the headline is measured.  With several ranks the extras contain collectives of their own; a rank
    # that fails or hangs there must not take the line (or the launcher) with it: after EXTRAS_TIMEOUT_S every rank leaves, rank 0
    # printing the line with the extras it has.
    extras = {}
    emit_lock, emitted = threading.Lock(), []

    def bail():
        if rank == 0:
            emit(dict(extras, extras_aborted=f"the extra measurements did not finish within {EXTRAS_TIMEOUT_S} s"), with_cpu_baseline=False)
        sys.stdout.flush()
        os._exit(0)

    watchdog = None
    if world > 1:
        watchdog = threading.Timer(EXTRAS_TIMEOUT_S, bail)
        watchdog.daemon = True
        watchdog.start()

    if not args.headline_only:
        # the same launch in the other division modes of the fitness path (include/evogp_hip.h), for the record
        div_ms = {}
        default_mode = evogp_amd.get_sr_division()
        for mode in ("ieee", "short", "fast"):
            evogp_amd.set_sr_division(mode)
            for _ in range(2):
                forest.SR_fitness(Xd, yd, True, "auto")
            torch.cuda.synchronize()
            _lib.check(_lib.lib.evogp_hip_timer_begin(stream), "timer_begin")
            for _ in range(5):
                forest.SR_fitness(Xd, yd, True, "auto")
            ms = ctypes.c_float(0)
            _lib.check(_lib.lib.evogp_hip_timer_end(stream, ctypes.byref(ms)), "timer_end")
            div_ms[mode] = ms.value / 5
        evogp_amd.set_sr_division(default_mode)
        extras["division"] = {"mode": default_mode, "call_ms_by_mode": div_ms,
                              "what": "short = IEEE range/special handling with one residual correction (default; bit-identical fitness "
                                      "to ieee on this workload); blocks of rows whose operands all lie in [2^-46, 2^46] skip the range "
                                      "scaling, which does nothing there (same bits).  ieee = correctly rounded always.  fast = as short, "
                                      "but blocks outside that range take rows without range scaling"}

        from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, DefaultSelection, GeneticProgramming
        from evogp_amd.tree import Forest, GenerateDescriptor

        mdesc = GenerateDescriptor(max_tree_len=GP_LEN, input_len=VAR_LEN, output_len=1, using_funcs=["+", "-", "*", "/"],
                                   max_layer_cnt=3, const_samples=[-1, 0, 1])

        # one generation of the GLOBAL population: local fitness, the exchange, identical selection / random words on
        # every rank, every rank breeds its own rows.  Measured for DefaultSelection and for the TournamentSelection BASELINE
        # configs[2] names (the reference's own setting, example/uci_sr.py:73-75), and for each form of the exchange: two
        # all-gathers (fitness, then the rows the selection names; exact or sync-free row count) or ONE packed all-gather.
        from evogp_amd.algorithm.selection import TournamentSelection
        from evogp_amd.parallel import ShardedGeneticProgramming

        def sharded_generation(selection, sel_name, exchange, cap):
            ms, err, sent = [], None, None
            try:
                sg = ShardedGeneticProgramming(forest, 0.2, mdesc, selection, seed=1234, exchange=exchange, cap=cap)
                neg_inf = torch.full((pop,), float("-inf"), dtype=torch.float32, device=device)
                for _ in range(5):
                    barrier(); g0 = time.perf_counter()
                    f = torch.ops.evogp_hip.fitness_scores(sg.forest.SR_fitness(Xd, yd, True, "auto"), True)   # (sign + NaN -> -inf: SymbolicRegression.scores)
                    sg.step(f)
                    barrier(); ms.append(max_over_ranks((time.perf_counter() - g0) * 1000))
                sent = dict(sg.last_exchange)
                del sg
            except Exception as exc:  # the fitness line must survive a failure of the exchange step
                err = repr(exc)[:300]
            return {"selection": sel_name, "exchange": "none (one rank)" if world == 1 else (exchange if exchange == "packed" else f"rows/{cap}"),
                    "median": float(np.median(ms[1:])) if len(ms) > 1 else None, "last_exchange": sent, "error": err}

        default_sel = lambda: DefaultSelection(0.3, elite_rate=0.01)                                             # noqa: E731
        tournament_sel = lambda: TournamentSelection(tournament_size=20, survivor_rate=0.5, elite_rate=0.1)       # noqa: E731
        modes = [("rows", "exact")] if world == 1 else [("rows", "exact"), ("rows", "bound"), ("packed", "exact")]
        runs = [sharded_generation(default_sel(), "DefaultSelection(0.3, elite_rate=0.01)", e, c) for e, c in modes]
        runs += [sharded_generation(tournament_sel(), "TournamentSelection(20, survivor_rate=0.5, elite_rate=0.1)", e, c) for e, c in modes]
        extras["generation_ms_sharded"] = {
            "median": runs[0]["median"], "selection": runs[0]["selection"], "exchange": runs[0]["exchange"], "error": runs[0]["error"],
            "global_pop": P, "ranks": world, "backend": backend, "runs": runs,
            "what": "whole population: local fitness + exchange (rows: all-gather of the fitness values, selection, all-gather of the rows the "
                    "selection names; packed: ONE all-gather of {fitness|value|type|size} per tree, then selection) + breeding pass for the local "
                    "rows, max over ranks, barriers on both sides; DefaultMutation(0.2); last_exchange.bytes_sent = bytes this rank contributes"}

        # What strong scaling can give at best: the fitness pass on the shards N = 8, 4, 2 ranks would own, timed on THIS GPU.  The
        # first SCALE record can be checked against it (a rank cannot be faster than its shard alone).
        if world == 1 and P >= 8:
            sm_trees, sm_ms = [], []
            for nshard in (8, 4, 2, 1):
                n_s = P // nshard
                fs = forest if nshard == 1 else forest[:n_s]   # (a view that keeps the forest's function mask: the call a rank of an N-rank run makes on its shard)
                for _ in range(10):
                    fs.SR_fitness(Xd, yd, True, "auto")
                reps = []
                for _ in range(3):   # (the median of three runs of 20 calls: a single run of 20 x 0.15 ms is 3 ms of wall clock)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(20):
                        fs.SR_fitness(Xd, yd, True, "auto")
                    torch.cuda.synchronize()
                    reps.append((time.perf_counter() - t0) / 20 * 1e3)
                sm_trees.append(n_s); sm_ms.append(float(np.median(reps)))
            extras["shard_model"] = {"trees": sm_trees, "ms": sm_ms, "efficiency_vs_linear": [sm_ms[-1] * t / P / m for t, m in zip(sm_trees, sm_ms)],
                                     "what": "tree_SR_fitness (Forest.SR_fitness, the timed step's call) on the first P/8, P/4, P/2, P trees of the headline population on one GPU: the per-rank time "
                                             "an N-rank strong-scaling run cannot beat; efficiency = (time of P trees x share) / time of the shard"}

        # BASELINE configs[1]: 100k trees per GPU (weak), same protocol
        pop1 = args.pop_per_gpu
        forest1, _, _, _, _ = sr_inputs(rank * pop1, pop1, device)
        nodes1 = int(forest1.batch_subtree_size[:, 0].to(torch.int64).sum())
        e1, call1_ms = timed_steps(forest1, Xd, yd, args.warmup, args.steps)
        for _ in range(3):
            reference_op(forest1, Xd, yd)
        e1_ref, _ = timed_steps(forest1, Xd, yd, args.warmup, args.steps, call=reference_op)
        algo = GeneticProgramming(forest1, DefaultCrossover(), DefaultMutation(0.2, mdesc), DefaultSelection(0.3, elite_rate=0.01))
        gen_ms = []
        neg_inf = torch.full((pop1,), float("-inf"), dtype=torch.float32, device=device)
        for _ in range(6):
            torch.cuda.synchronize(); g0 = time.perf_counter()
            f = torch.ops.evogp_hip.fitness_scores(algo.forest.SR_fitness(Xd, yd, True, "auto"), True)   # (sign + NaN -> -inf in one launch: SymbolicRegression.scores, what StandardPipeline.step calls)
            algo.step(f)
            torch.cuda.synchronize(); gen_ms.append((time.perf_counter() - g0) * 1000)
        extras["configs1"] = {
            "workload": "BASELINE configs[1]: SymbolicRegression synthetic 10-var, pop=100k per GPU, 1024 datapoints, max_tree_len=64, "
                        "funcs + - * /, one tree_SR_fitness pass per step",
            "scaling": "weak", "pop_per_gpu": pop1, "ms_per_step": e1 / args.steps * 1000.0, "call_ms_events": call1_ms,
            "ms_per_step_reference_op": e1_ref / args.steps * 1000.0, "reference_op_gap": e1_ref / e1 - 1.0,
            "tree_evals_per_s": float(pop1) * DATAPOINTS * world * args.steps / e1,
            "mean_tree_len": nodes1 / pop1,
            "generation_ms": {"median": float(np.median(gen_ms[1:])), "first": gen_ms[0],
                              "what": "fitness + DefaultSelection + DefaultCrossover + DefaultMutation(0.2) on one shard"},
        }

    if not args.headline_only and rank == 0:
        # ONE generation on a FIXED workload with a stage split (the step of /root/reference/src/evogp/pipeline/standard.py:38-54,
        # algorithm/genetic_programming.py:105-124): the same forest and the same random words every repetition (the population is put back
        # and the step counter reset), HIP events between the stages -- fitness (the pass + its sign and NaN -> -inf, what
        # SymbolicRegression.evaluate hands the algorithm) | selection | masked donor generation | breeding pass --, and the whole
        # generation once more WITHOUT events in it (wall clock over the repetitions); for DefaultSelection and for the tournament
        # selection configs[2] names, at the headline population and at configs[1].
        def generation_stages(f0, selection, sel_name, reps=10):
            algo = GeneticProgramming(f0, DefaultCrossover(), DefaultMutation(0.2, mdesc), selection)
            neg_inf = torch.full((f0.pop_size,), float("-inf"), dtype=torch.float32, device=device)
            names = ["fitness", "select", "donors", "breeding"]

            def one(events):
                algo.forest = f0
                algo._steps = 0
                k = [0]

                def mark(_name=None):
                    if events is not None:
                        events[k[0]].record()
                        k[0] += 1
                algo.stage_marker = mark if events is not None else None
                mark()
                f = torch.ops.evogp_hip.fitness_scores(f0.SR_fitness(Xd, yd, True, "auto"), True)
                mark()
                algo.step(f)
                return k[0]

            for _ in range(3):
                one(None)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps):
                one(None)
            torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e3
            per = {n: [] for n in names}
            whole = []
            for _ in range(reps):
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
                got = one(evs)
                torch.cuda.synchronize()
                if got != 5:
                    return {"selection": sel_name, "error": f"the fused step recorded {got} of 5 events (composed operators ran)"}
                for i, n in enumerate(names):
                    per[n].append(evs[i].elapsed_time(evs[i + 1]))
                whole.append(evs[0].elapsed_time(evs[4]))
            stage = {n: float(np.median(v)) for n, v in per.items()}
            total = sum(stage.values())
            # (an event between two kernels costs the stream ~4 us: a generation WITH the five events in it is that much longer than the one
            # timed without -- both are reported; the stages add up to the former)
            return {"selection": sel_name, "trees": f0.pop_size, "mean_tree_len": float(f0.batch_subtree_size[:, 0].float().mean()),
                    "generation_ms": wall, "generation_ms_with_events": float(np.median(whole)), "stage_ms": stage, "stage_sum_ms": total,
                    "stage_sum_over_generation": total / wall, "stage_sum_over_generation_with_events": total / float(np.median(whole))}

        try:
            extras["generation_fixed_workload"] = {
                "headline_population": [generation_stages(forest, default_sel(), "DefaultSelection(0.3, elite_rate=0.01)"),
                                        generation_stages(forest, tournament_sel(), "TournamentSelection(20, survivor_rate=0.5, elite_rate=0.1)")],
                "configs1": [generation_stages(forest1, default_sel(), "DefaultSelection(0.3, elite_rate=0.01)"),
                             generation_stages(forest1, tournament_sel(), "TournamentSelection(20, survivor_rate=0.5, elite_rate=0.1)")],
                "what": "one generation of a FIXED forest (population and random words restored every repetition): generation_ms = wall clock per "
                        "repetition without events inside; stage_ms = medians of HIP-event intervals fitness | select | donors | breeding "
                        "(DefaultCrossover + DefaultMutation(0.2, max_layer_cnt 3) in one breeding pass)"}
        except Exception as exc:
            extras["generation_fixed_workload"] = {"error": repr(exc)[:300]}

    if not args.headline_only:
        # BASELINE configs[4] (example/brax_task.py:19-32 shape: policy trees pop 50 000, 17 observations, 6 actions, max_tree_len 256,
        # 1000 steps per generation) across the ranks: every rank rolls out its own 50 000 / N trees (HIP-graph replay of one step: the
        # prepared forward pass + the environment's kernels; no collective inside the episode), then the sharded generation step.  Brax
        # is not in this image: the environment is the batched linear system of evogp_amd/problem/rollout.py (per-tree episodes
        # keyed by the global tree index).
        try:
            from evogp_amd.problem import RolloutProblem
            from evogp_amd.problem.rollout import LinearTrackingEnv

            c5_pop, c5_steps = 50_000, 1000
            n5 = c5_pop // world
            pdesc = GenerateDescriptor(max_tree_len=256, input_len=17, output_len=6, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6,
                                       const_samples=torch.linspace(-1, 1, 100).tolist())
            pf = Forest.random_generate(n5, pdesc, keys=torch.tensor([7, 0], dtype=torch.uint32, device=device), tree_index_offset=rank * n5)
            prob5 = RolloutProblem(LinearTrackingEnv(device=device, randomize=0.1), c5_steps, use_graph=True)
            sg5 = ShardedGeneticProgramming(pf, 0.2, pdesc.update(max_layer_cnt=3), DefaultSelection(0.3, elite_rate=0.01), seed=99)
            roll_ms, gen5_ms = [], []
            for _ in range(3):
                barrier(); g0 = time.perf_counter()
                fit5 = prob5.evaluate(sg5.forest, tree_index_offset=rank * n5)
                barrier(); g1 = time.perf_counter()
                sg5.step(fit5)
                barrier(); g2 = time.perf_counter()
                roll_ms.append(max_over_ranks((g1 - g0) * 1e3)); gen5_ms.append(max_over_ranks((g2 - g0) * 1e3))
            # the engine's share of a step: the prepared forward pass alone (and, for the record, the stack interpreter a forest's FIRST
            # call runs), each as a captured graph replayed like the rollout's step -- the rest of us_per_step is the environment
            def replayed_us(fn, reps=300):
                obs5 = torch.randn(n5, 17, device=device)
                fn(obs5); fn(obs5)
                side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
                gf = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    with torch.cuda.graph(gf, stream=side):
                        fn(obs5)
                torch.cuda.current_stream().wait_stream(side)
                for _ in range(20):
                    gf.replay()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(reps):
                    gf.replay()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps * 1e6

            fwd_us = stack_us = None
            try:
                f5 = sg5.forest
                fwd_us = replayed_us(lambda o: f5.forward(o))
                vt, tt, st5 = f5.batch_node_value, f5.batch_node_type, f5.batch_subtree_size
                stack_us = replayed_us(lambda o: torch.ops.evogp_cuda.tree_evaluate(n5, 256, 17, 6, vt, tt, st5, o))
            except Exception as exc:
                fwd_us = fwd_us if fwd_us is not None else repr(exc)[:200]
            extras["c5_rollout"] = {
                "forward_us_per_step": fwd_us, "stack_interpreter_us_per_step": stack_us,
                "env_us_per_step": (float(np.median(roll_ms[1:])) * 1e3 / c5_steps - fwd_us) if isinstance(fwd_us, float) else None,
                "workload": f"BASELINE configs[4] shape: policy trees pop {c5_pop} over {world} rank(s) ({n5} per rank), 17 observations, 6 actions, "
                            f"max_tree_len 256, {c5_steps} steps per generation; batched linear stand-in environment (no Brax in this image)",
                "rollout_ms": float(np.median(roll_ms[1:])), "us_per_step": float(np.median(roll_ms[1:])) * 1e3 / c5_steps,
                "policy_steps_per_s": c5_pop * c5_steps / (float(np.median(roll_ms[1:])) / 1e3),
                "generation_ms": float(np.median(gen5_ms[1:])), "first_rollout_ms": roll_ms[0],
                "what": "rollout = prepare the forest's operation lists + capture one step + 1000 graph replays (max over ranks); generation = "
                        "rollout + sharded step (fitness all-gather, selection, row exchange, breeding); forward_us_per_step = the prepared forward "
                        "pass alone (Forest.forward on an unchanged forest, replayed graph), stack_interpreter_us_per_step = tree_evaluate itself "
                        "(what a forest's first call runs), env_us_per_step = us_per_step - forward: the torch stand-in environment's kernels"}
            del sg5, prob5, pf
        except Exception as exc:
            extras["c5_rollout"] = {"error": repr(exc)[:300]}

    if not args.headline_only and rank == 0:
        # The reference's own SR script shape (example/uci_sr.py:45-75): max_tree_len 512, max_layer_cnt 9, layer_leaf_prob 0.3, 10 000
        # constants in [-5, 5], functions + - * / sin cos tan, DefaultMutation(0.1, max_layer_cnt 4), TournamentSelection(20, 0.5, 0.1),
        # on the 10-variable dataset of the headline (the UCI tables need the network), pop 100 000 x 1024 rows.  Timed at generation 0
        # (short trees: the generator's depth limits) and after 30 generations of the script's own operators (rows fill up towards
        # 512 nodes: the long-program paths), each with the stage split, the share of trees the threaded code leaves to the register
        # kernels and the HBM roofline of SURVEY.md section 8d.
        try:
            from evogp_amd.algorithm.selection import TournamentSelection as _TS

            torch.manual_seed(42)   # (the constants of the descriptor and the word seed of GeneticProgramming come from torch's generators: the trajectory)
            udesc = GenerateDescriptor(max_tree_len=512, input_len=VAR_LEN, output_len=1, using_funcs=["+", "-", "*", "/", "sin", "cos", "tan"],
                                       max_layer_cnt=9, const_range=[-5, 5], sample_cnt=10000, layer_leaf_prob=0.3)
            upop = 100_000
            ualgo = GeneticProgramming(Forest.random_generate(upop, udesc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=device)),
                                       DefaultCrossover(), DefaultMutation(0.1, udesc.update(max_layer_cnt=4)),
                                       _TS(tournament_size=20, survivor_rate=0.5, elite_rate=0.1))
            uneg = torch.full((upop,), float("-inf"), dtype=torch.float32, device=device)

            def uci_point(f):
                for _ in range(3):
                    f.SR_fitness(Xd, yd, True, "auto")
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10):
                    f.SR_fitness(Xd, yd, True, "auto")
                torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
                _lib.check(_lib.lib.evogp_hip_debug_profile(2), "profile on")   # stage events; the call stops behind the threaded code
                words = f.SR_fitness(Xd, yd, True, "auto").view(torch.int32)
                st = (ctypes.c_float * 3)(); nc = ctypes.c_int(0)
                _lib.check(_lib.lib.evogp_hip_debug_profile_read(st, ctypes.byref(nc)), "profile read")
                _lib.check(_lib.lib.evogp_hip_debug_profile(0), "profile off")
                left = int(((words == 0x7FC0FEED) | (words == 0x7FC0BEEF) | (words == 0x7FC0DEED)).sum())
                lens = f.batch_subtree_size[:, 0].to(torch.int64)
                nodes = int(lens.sum())
                abytes = 6.0 * nodes + 2.0 * upop + 4.0 * DATAPOINTS * (VAR_LEN + 1) + 4.0 * upop
                return {"call_ms": ms, "tree_evals_per_s": upop * DATAPOINTS / (ms / 1e3), "node_evals_per_s": nodes * DATAPOINTS / (ms / 1e3),
                        "mean_tree_len": nodes / upop, "max_tree_len": int(lens.max()), "share_longer_than_64": float((lens > 64).float().mean()),
                        "stage_ms": {"program_compilers": st[0], "interpreter": st[1]},
                        "share_left_to_register_kernels": left / upop,
                        "roofline": {"bound": "hbm", "algorithmic_bytes": abytes, "achieved": abytes / (ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS,
                                     "unit": "GB/s", "frac": abytes / (ms / 1e3) / 1e9 / HBM_PEAK_GBS}}

            upoints = {"generation_0": uci_point(ualgo.forest)}
            ugen = []
            for g_ in range(30):
                torch.cuda.synchronize(); g0 = time.perf_counter()
                ualgo.step(torch.ops.evogp_hip.fitness_scores(ualgo.forest.SR_fitness(Xd, yd, True, "auto"), True))
                torch.cuda.synchronize(); ugen.append((time.perf_counter() - g0) * 1000)
            upoints["generation_30"] = uci_point(ualgo.forest)
            extras["uci_sr_shape"] = {
                "workload": "example/uci_sr.py:45-75 shape: pop 100k x 1024 rows x 10 variables, max_tree_len 512, max_layer_cnt 9, layer_leaf_prob 0.3, "
                            "10 000 constants in [-5, 5], + - * / sin cos tan, DefaultCrossover, DefaultMutation(0.1, max_layer_cnt 4), "
                            "TournamentSelection(20, survivor_rate 0.5, elite_rate 0.1)",
                "points": upoints, "generation_ms": {"first": ugen[0], "median_gen_10_29": float(np.median(ugen[10:])), "last": ugen[-1]}}
            del ualgo
        except Exception as exc:
            extras["uci_sr_shape"] = {"error": repr(exc)[:300]}

    if not args.headline_only and rank == 0:
        # the reference's only published timing: test/vis.ipynb:12-45,171,181 -- XOR-3d, pop 100 000, max_tree_len 128, 8 datapoints,
        # functions + - log sqrt pow / inv, max_layer_cnt 2, default operators: 14-21 ms per generation on an unnamed NVIDIA GPU
        try:
            vdesc = GenerateDescriptor(max_tree_len=128, input_len=3, output_len=1, using_funcs=["+", "-", "log", "sqrt", "pow", "/", "inv"],
                                       max_layer_cnt=2, const_samples=[-1, 0, 1])
            vX = torch.tensor([[a, b, c] for a in (0., 1.) for b in (0., 1.) for c in (0., 1.)], device=device)
            vy = (vX.sum(1) % 2)[:, None].contiguous()
            vf = Forest.random_generate(100_000, vdesc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=device))
            valgo = GeneticProgramming(vf, DefaultCrossover(), DefaultMutation(0.2, vdesc), DefaultSelection(0.3, elite_rate=0.01))
            vms, vlen = [], []
            vneg = torch.full((100_000,), float("-inf"), dtype=torch.float32, device=device)
            for g_ in range(40):
                torch.cuda.synchronize(); g0 = time.perf_counter()
                valgo.step(torch.ops.evogp_hip.fitness_scores(valgo.forest.SR_fitness(vX, vy, True, "auto"), True))
                torch.cuda.synchronize(); vms.append((time.perf_counter() - g0) * 1000)
                if g_ in (0, 39):
                    vlen.append(float(valgo.forest.batch_subtree_size[:, 0].float().mean()))
            extras["vis_ipynb_config"] = {
                "workload": "test/vis.ipynb:171,181: XOR-3d SR, pop 100k, max_tree_len 128, 8 datapoints, funcs + - log sqrt pow / inv, default operators",
                "generation_ms": {"median_gen_10_39": float(np.median(vms[10:])), "first": vms[0], "max_after_warmup": float(max(vms[2:]))},
                "mean_tree_len_first_last": vlen, "reference_published_ms": "14-21 (hardware not stated)"}
        except Exception as exc:
            extras["vis_ipynb_config"] = {"error": repr(exc)[:300]}

        # BASELINE configs[3]: the classifier of example/uci_classifier.py at its shape -- pop 200 000 multi-output trees, output_len 10,
        # max_tree_len 128 -- on sklearn's digits (64 features x 1797 rows, 10 classes: the offline data set classification.py:35-48
        # loads; the example's UCI table needs the network); fitness = fused arg-max count (compiled programs + END_CLS handler,
        # equal to torch's argmax(clip(softmax)) count), then the default generation step
        try:
            from evogp_amd.problem import Classification

            cdesc = GenerateDescriptor(max_tree_len=128, input_len=64, output_len=10, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6,
                                       const_samples=[-1, 0, 1])
            cprob = Classification(dataset="digits")
            cpop = 200_000
            calgo = GeneticProgramming(Forest.random_generate(cpop, cdesc, keys=torch.tensor([7, 0], dtype=torch.uint32, device=device)),
                                       DefaultCrossover(), DefaultMutation(0.2, cdesc.update(max_layer_cnt=3)), DefaultSelection(0.3, elite_rate=0.01))
            for _ in range(2):
                cprob.evaluate(calgo.forest)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                cprob.evaluate(calgo.forest)
            torch.cuda.synchronize(); fit_ms = (time.perf_counter() - t0) / 5 * 1000
            cms = []
            for _ in range(6):
                torch.cuda.synchronize(); g0 = time.perf_counter()
                calgo.step(cprob.evaluate(calgo.forest))
                torch.cuda.synchronize(); cms.append((time.perf_counter() - g0) * 1000)
            extras["configs3"] = {
                "workload": "BASELINE configs[3]: classifier trees, pop 200k, output_len 10, max_tree_len 128, Classification(dataset='digits') = "
                            f"sklearn load_digits, {cprob.datapoints.shape[1]} features x {cprob.datapoints.shape[0]} rows, accuracy fitness + default operators",
                "best_accuracy_after_6_generations": float(cprob.evaluate(calgo.forest).max()),
                "fitness_ms": fit_ms, "tree_evals_per_s": cpop * 1797 / (fit_ms / 1e3), "generation_ms": {"median": float(np.median(cms[1:])), "first": cms[0]}}
            del calgo, cprob
        except Exception as exc:
            extras["configs3"] = {"error": repr(exc)[:300]}

    if not args.headline_only and rank == 0:
        # one generation (selection + crossover + mutation, the fitness handed in) under the reference's OTHER operator
        # sets -- example/brax_task.py:38-45 CombinedMutation[DefaultMutation(0.2), DeleteMutation(0.8)] at configs[4]'s population, a Hoist /
        # Insert list, the point mutations at configs[1]'s -- next to the default step on the same forests: the fused breeding pass plus one
        # launch per further operator (csrc/mutate_ops.hip; round 4's torch programs: 0.7-1.1 ms and 90-120 launches, profiles/r05A_n3_generation.log)
        try:
            from evogp_amd.algorithm import CombinedMutation, DeleteMutation, HoistMutation, InsertMutation, MultiConstMutation, SinglePointMutation

            def gen_ms(forest, mutation, reps=10):
                algo = GeneticProgramming(forest, DefaultCrossover(), mutation, DefaultSelection(survival_rate=0.3, elite_rate=0.01))
                fit = torch.rand(forest.pop_size, device=device)
                for _ in range(3):
                    algo.step(fit)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(reps):
                    algo.step(fit)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps * 1e3

            d6 = GenerateDescriptor(max_tree_len=256, input_len=17, output_len=6, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_range=[-1, 1], sample_cnt=100)
            d1 = GenerateDescriptor(max_tree_len=64, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1])
            f6 = lambda: Forest.random_generate(50_000, d6, keys=torch.tensor([7, 1], dtype=torch.uint32, device=device))
            f1 = lambda: Forest.random_generate(100_000, d1, keys=torch.tensor([42, 0], dtype=torch.uint32, device=device))
            n3 = {"pop50k_L256_out6": {"default_step": gen_ms(f6(), DefaultMutation(0.2, d6.update(max_layer_cnt=3))),
                                       "brax_task_combined_default_delete": gen_ms(f6(), CombinedMutation([DefaultMutation(0.2, d6.update(max_layer_cnt=3)), DeleteMutation(0.8)])),
                                       "combined_hoist_insert": gen_ms(f6(), CombinedMutation([HoistMutation(0.2), InsertMutation(0.2, d6.update(max_layer_cnt=3))]))},
                  "pop100k_L64": {"default_step": gen_ms(f1(), DefaultMutation(0.2, d1.update(max_layer_cnt=3))),
                                  "single_point": gen_ms(f1(), SinglePointMutation(0.2, d1)),
                                  "combined_single_point_multi_const": gen_ms(f1(), CombinedMutation([SinglePointMutation(0.2, d1), MultiConstMutation(0.2, d1)]))}}
            extras["n3_operator_sets"] = {"generation_ms": n3, "what": "selection + DefaultCrossover + the named mutation(s) on a random fitness vector, ms per generation; "
                                          "DefaultSelection(0.3, elite_rate 0.01)"}
        except Exception as exc:
            extras["n3_operator_sets"] = {"error": repr(exc)[:300]}

    if rank == 0:
        emit(extras)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()


if __name__ == "__main__":
    main()
