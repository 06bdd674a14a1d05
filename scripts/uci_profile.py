#!/usr/bin/env python3
"""The example/uci_sr.py shape after GENS generations (bench.py uci_sr_shape's trajectory) for a kernel trace: PASSES fitness calls with
the straight-line long-tree compiler, then PASSES with the general compiler's staged passes (evogp_hip_debug_long_compiler), so that
rocprofv3 --kernel-trace --stats shows the program compilers of both.   ROCPD_BY_GRID=1 rocpd_summary separates the launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import evogp_amd  # noqa: F401
from evogp_amd import _lib
from evogp_amd.algorithm import DefaultCrossover, DefaultMutation, GeneticProgramming
from evogp_amd.algorithm.selection import TournamentSelection
from evogp_amd.tree import Forest, GenerateDescriptor
sys.argv = [sys.argv[0]]
import bench

dev = torch.device("cuda", 0)
GENS, PASSES = int(os.environ.get("GENS", "30")), int(os.environ.get("PASSES", "5"))
_, Xd, yd, _, _ = bench.sr_inputs(0, 1000, dev)
torch.manual_seed(42)
udesc = GenerateDescriptor(max_tree_len=512, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/", "sin", "cos", "tan"], max_layer_cnt=9,
                           const_range=[-5, 5], sample_cnt=10000, layer_leaf_prob=0.3)
upop = 100_000
algo = GeneticProgramming(Forest.random_generate(upop, udesc, keys=torch.tensor([42, 0], dtype=torch.uint32, device=dev)), DefaultCrossover(),
                          DefaultMutation(0.1, udesc.update(max_layer_cnt=4)), TournamentSelection(tournament_size=20, survivor_rate=0.5, elite_rate=0.1))
neg = torch.full((upop,), float("-inf"), dtype=torch.float32, device=dev)
for _ in range(GENS):
    f = -algo.forest.SR_fitness(Xd, yd, True, "auto")
    algo.step(torch.where(torch.isnan(f), neg, f))
f = algo.forest
lens = f.batch_subtree_size[:, 0].float()
print(f"generation {GENS}: mean {float(lens.mean()):.1f} nodes, longest {int(lens.max())}, {float((lens > 64).float().mean()):.1%} beyond 64, "
      f"{float((lens > 128).float().mean()):.1%} beyond 128, {float((lens > 256).float().mean()):.1%} beyond 256")
words = {}
for fast in (1, 0):
    _lib.lib.evogp_hip_debug_long_compiler(fast)
    words[fast] = f.SR_fitness(Xd, yd, True, "auto").view(torch.int32).clone(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(PASSES):
        f.SR_fitness(Xd, yd, True, "auto")
    torch.cuda.synchronize()
    print(f"long compiler {'straight line' if fast else 'general passes'}: {(time.perf_counter() - t0) / PASSES * 1e3:.3f} ms per call")
    # marker launches so that the trace shows where one setting ends (a tiny generate call)
    Forest.random_generate(64 + fast, udesc, keys=torch.tensor([1, 2], dtype=torch.uint32, device=dev))
_lib.lib.evogp_hip_debug_long_compiler(-1)
diff = torch.nonzero(words[0] != words[1]).flatten()
print(f"fitness words: {len(diff)} of {upop} differ between the two long-tree compilers", diff[:10].tolist(),
      words[1][diff[:10]].view(torch.float32).tolist(), words[0][diff[:10]].view(torch.float32).tolist(), lens[diff[:10]].tolist())

if len(diff):
    import collections
    d_ = diff.cpu().numpy(); w1 = words[1].cpu().numpy().view("uint32")[d_]
    print("words of the differing trees:", collections.Counter(hex(int(x)) for x in w1).most_common(5))
    print("index range", d_.min(), d_.max(), "groups mod 1280 histogram (8 bins):", collections.Counter(((d_ // 16) % 1280) // 160), "iteration", collections.Counter((d_ // 16) // 1280),
          "rank in group", collections.Counter(d_ % 16), "lens", collections.Counter((lens[diff].cpu().numpy() // 16 * 16).astype(int)))
if len(diff) and os.environ.get("DUMP", "0") != "0":
    import ctypes, json
    table = json.load(open(os.path.join(ROOT, "evogp_amd", "lib", "tc_handlers.json")))["K8_short"]["handlers"]
    names = {v["id"]: k for k, v in table.items()}
    nh = _lib.lib.evogp_hip_debug_tc_nhandlers()
    NAMES = "if + - * / ldiv pow lpow max min < > <= >= sin cos tan sinh cosh tanh log llog exp inv linv neg abs sqrt lsqrt".split()
    pick = sorted(diff.tolist(), key=lambda r: float(lens[r]))[:2]
    for r in pick:
        n = int(lens[r])
        v, t, s_ = (a[r, :n].cpu().numpy() for a in (f.batch_node_value, f.batch_node_type, f.batch_subtree_size))
        print(f"tree {r} ({n} nodes):", " ".join(f"x{int(v[i])}" if t[i] == 0 else f"{v[i]:.4g}" if t[i] == 1 else f"{NAMES[int(v[i])]}[{int(s_[i])}]" for i in range(n)))
        for fast in (1, 0):   # (the tree's record after a call on the whole population)
            _lib.lib.evogp_hip_debug_long_compiler(fast)
            val = float(f.SR_fitness(Xd, yd, True, "auto")[r])
            buf = (ctypes.c_uint * 4096)()
            k = _lib.lib.evogp_hip_debug_tc_program(int(r), ctypes.cast(buf, ctypes.c_void_p), 2048)
            prog = []
            for i in range(k):
                w0, w1 = buf[2 * i], buf[2 * i + 1]
                h = (w0 & 0xFFFF) // 256
                prog.append(f"{names.get(h % nh, '?')}{'' if h < nh else chr(39)}(v{w0 >> 24},a{(w0 >> 16) & 255},{ctypes.c_float.from_buffer(ctypes.c_uint(w1)).value:.4g}|{w1:#x})")
            print(f"  {'straight line ' if fast else 'general passes'} fitness {val!r}: {k} words:", " ".join(prog))
    _lib.lib.evogp_hip_debug_long_compiler(-1)
