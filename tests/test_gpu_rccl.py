"""GPU, two or more devices: the sharded generation step over RCCL (backend "nccl" on ROCm), one process per GPU — the union of
the shards must equal the single-device population bit for bit.  Skipped on a one-GPU box (there the same code runs over gloo
on the CPU, tests/test_sharded_gloo.py, and the slices of G = 1, 2, 3, 8 ranks are built on one GPU, tests/test_gpu_breed.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
POP, L, GENS = 40_000, 64, 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(rank, world, port, outdir):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import evogp_amd  # noqa: F401
    from evogp_amd.parallel import ShardedGeneticProgramming
    from evogp_amd.tree import Forest, GenerateDescriptor, set_default_device

    set_default_device(dev)
    desc = GenerateDescriptor(max_tree_len=L, input_len=10, output_len=1, using_funcs=["+", "-", "*", "/"], max_layer_cnt=6, const_samples=[-1, 0, 1])
    n_local = POP // world
    keys = torch.tensor([42, 0], dtype=torch.uint32, device=dev)
    local = Forest.random_generate(n_local, desc, keys=keys, tree_index_offset=rank * n_local)
    g = torch.Generator().manual_seed(1234)
    X = (torch.rand(256, 10, generator=g) * 10 - 5).to(dev)
    y = (X[:, 0] * X[:, 1] - X[:, 4])[:, None].contiguous()
    from evogp_amd.algorithm.selection import DefaultSelection, TournamentSelection

    # DefaultSelection over two all-gathers, and BASELINE configs[2]'s tournament selection over ONE packed all-gather
    for name, sel, exchange in (("default", DefaultSelection(0.3, elite_rate=0.01), "rows"),
                                ("tournament", TournamentSelection(20, survivor_rate=0.5, elite_rate=0.1), "packed")):
        start = Forest(local.input_len, local.output_len, local.batch_node_value.clone(), local.batch_node_type.clone(), local.batch_subtree_size.clone())
        gp = ShardedGeneticProgramming(start, 0.2, desc.update(max_layer_cnt=3), selection=sel, seed=123, exchange=exchange)
        for _ in range(GENS):
            fit = -gp.forest.SR_fitness(X, y)
            fit = torch.where(torch.isnan(fit), torch.full_like(fit, float("-inf")), fit)
            gp.step(fit)
        f = gp.forest
        np.savez(os.path.join(outdir, f"w{world}_r{rank}_{name}.npz"), v=f.batch_node_value.cpu().numpy(), t=f.batch_node_type.cpu().numpy(),
                 s=f.batch_subtree_size.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_rccl_shards_equal_the_single_device_population(tmp_path):
    import torch.multiprocessing as mp

    out = str(tmp_path)
    world = 2 if torch.cuda.device_count() < 4 else 4
    mp.spawn(_run, args=(1, _free_port(), out), nprocs=1, join=True)
    mp.spawn(_run, args=(world, _free_port(), out), nprocs=world, join=True)
    for name in ("default", "tournament"):
        one = np.load(os.path.join(out, f"w1_r0_{name}.npz"))
        parts = [np.load(os.path.join(out, f"w{world}_r{r}_{name}.npz")) for r in range(world)]
        for k in ("v", "t", "s"):
            got = np.concatenate([p[k] for p in parts])
            a, b = (got.view(np.uint32), one[k].view(np.uint32)) if k == "v" else (got, one[k])
            assert np.array_equal(a, b), f"{name}, {k}: the union of the {world} RCCL shards differs from the single-device population"
