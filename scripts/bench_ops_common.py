"""helpers shared by the single-operator timing scripts"""
import numpy as np, torch


def depth2leaf(max_layer_cnt, leaf_prob=0.2):
    """f32[10]: leaf probability per depth (descriptor.py:33-38)"""
    return np.array([leaf_prob] * (max_layer_cnt - 1) + [1.0] * (10 - (max_layer_cnt - 1)), np.float32)


def roulette_uniform(funcs):
    """f32[29]: cumulative weights of the functions in use, equal shares (descriptor.py:106-111)"""
    w = np.zeros(29, np.float64)
    w[list(funcs)] = 1.0 / len(funcs)
    return np.cumsum(w.astype(np.float32), dtype=np.float32)


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us
