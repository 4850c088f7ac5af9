"""Object sharding across ranks + the one collective of the path.

Objects are independent units (the reference loops over them serially, eval_tool/eval_classification.cpp:347-356;
Voting::clear() per object, implicit_shape_model.cpp:666), so rank r owns a contiguous block of the object list, holds a
full replica of the codebook, and the only exchange is an all-gather of fixed-size per-object records
{best_class, class_score[C]} (RCCL over xGMI on the GPU box, gloo in the CPU tests). Payload is KB-sized: latency-bound.
"""
import numpy as np


def shard_range(n_objects, rank, world_size):
    """contiguous block partition: rank r gets [r*ceil(n/R), (r+1)*ceil(n/R)) clipped to n"""
    per = (n_objects + world_size - 1) // world_size
    lo = min(n_objects, rank * per)
    return lo, min(n_objects, lo + per)


def shard_ranges_balanced(costs, world_size):
    """contiguous partition balanced by a per-object cost (e.g. point counts, cfg 4): greedy prefix split"""
    costs = np.asarray(costs, np.float64)
    total = costs.sum()
    bounds = [0]
    acc = 0.0
    r = 1
    for i, c in enumerate(costs):
        acc += c
        while r < world_size and acc >= total * r / world_size:
            bounds.append(i + 1)
            r += 1
    while len(bounds) < world_size:
        bounds.append(len(costs))
    bounds.append(len(costs))
    return [(bounds[i], max(bounds[i], bounds[i + 1])) for i in range(world_size)]


def pack_records(obj_index, class_score, pad_to):
    """fixed-size records [pad_to, 2 + C] float32: (object index, best class, scores); rows past the shard are -1"""
    import torch
    n, C = class_score.shape
    rec = torch.full((pad_to, 2 + C), -1.0, dtype=torch.float32, device=class_score.device)
    if n:
        best = torch.where(class_score.max(dim=1).values > 0, class_score.argmax(dim=1), torch.full((n,), -1, device=class_score.device))
        rec[:n, 0] = obj_index.to(torch.float32)
        rec[:n, 1] = best.to(torch.float32)
        rec[:n, 2:] = class_score
    return rec


def all_gather_records(rec, world_size):
    """one all-gather of the per-object records; returns [world_size * pad_to, 2 + C] on every rank"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return rec
    if dist.get_backend() == "gloo" and rec.is_cuda:
        # functional runs of the N > 1 path on ONE GPU (bench.py --share-gpu: RCCL refuses two ranks on one device): the records
        # travel through host memory; the data path proper (RCCL, device tensors) is the branch below
        host = rec.contiguous().cpu()
        out_h = torch.empty((world_size * host.shape[0], host.shape[1]), dtype=host.dtype)
        dist.all_gather_into_tensor(out_h, host)
        return out_h.to(rec.device)
    out = torch.empty((world_size * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec.contiguous())
    return out


def unpack_records(gathered):
    """-> (object indices, best classes, class scores) of the valid rows, sorted by object index"""
    import torch
    valid = gathered[:, 0] >= 0
    g = gathered[valid]
    order = torch.argsort(g[:, 0])
    g = g[order]
    return g[:, 0].to(torch.int64), g[:, 1].to(torch.int64), g[:, 2:]
