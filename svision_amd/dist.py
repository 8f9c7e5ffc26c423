"""Multi-GPU layout of the hot path: one process per GPU, chromosomes sharded across ranks,
one exchange at the end.

The collection and prediction of a chromosome never look at another chromosome
(SVision:264-278, :318-319), so ranks work on disjoint chromosome sets with no data-path
collective.  The only cross-shard dependency of the whole pipeline is the global score
min/max used to rescale QUAL (output.py:336-338) plus the concatenation of the per-chromosome
VCF bodies in task order (output.py:305-331): one all_reduce(MIN) + all_reduce(MAX) on a
scalar and one gather of packed records to rank 0, over RCCL/xGMI (backend "nccl") on GPUs
or gloo on CPU (tests).
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def world_initialized():
    return dist.is_available() and dist.is_initialized()


def env_rank():
    """(rank, world size) from the launcher's environment, without bringing the process group up."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def local_device_index():
    """The GPU of this rank: LOCAL_RANK modulo the visible devices (one process per GPU)."""
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return int(os.environ.get("LOCAL_RANK", "0")) % n if n else 0


def init_from_env(timeout_hours=24.0):
    """Initialise the process group when launched under torchrun (RANK/WORLD_SIZE set).  Ranks meet only once, at the
    end of their shards, which can be hours apart (chr1 vs a rank of small contigs): the collective timeout is long."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    # SVX_FORCE_DIST=1 initialises the group even for a single rank (exercises the RCCL path on one GPU)
    force = os.environ.get("SVX_FORCE_DIST") == "1" and "RANK" in os.environ
    if (ws > 1 or force) and not dist.is_initialized():
        # SVX_DIST_BACKEND=gloo: several ranks sharing one GPU (1-GPU test boxes; RCCL refuses duplicate devices)
        backend = os.environ.get("SVX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_device_index())
        import datetime
        kw = {}
        if backend == "nccl":                  # bind the communicator to this rank's GPU (otherwise guessed from the global rank)
            # one process per GPU: more local ranks than visible devices would put two ranks on one device, which RCCL answers
            # with a hang or an obscure "duplicate GPU" error minutes later.  Said here, at once, on every rank.
            # (only when the launcher says how many ranks THIS node runs: mpirun / srun set RANK and WORLD_SIZE alone, and the global
            # size of a multi-node job says nothing about one node.  ADVICE r5)
            local_ws, n_dev = int(os.environ.get("LOCAL_WORLD_SIZE", "0")), torch.cuda.device_count()
            if local_ws > n_dev and os.environ.get("SVX_DEVICE_CHECK", "fatal") != "off":
                raise RuntimeError("%d ranks on this node but %d visible GPU(s): rank %s would share cuda:%d with another rank under RCCL "
                                   "(one process per GPU; SVX_DIST_BACKEND=gloo runs several ranks on one device for rehearsals)"
                                   % (local_ws, n_dev, os.environ.get("RANK", "?"), local_device_index()))
            kw["device_id"] = torch.device("cuda", local_device_index())
        dist.init_process_group(backend, timeout=datetime.timedelta(hours=timeout_hours), **kw)
        if backend == "nccl":
            assert_one_device_per_rank()
    return world()


def device_identity():
    """(host name, physical identity of this rank's GPU): the device's UUID where the runtime reports one, else its PCI bus id."""
    import socket
    if not torch.cuda.is_available():
        return socket.gethostname(), "cpu"
    i = torch.cuda.current_device()
    props = torch.cuda.get_device_properties(i)
    ident = str(getattr(props, "uuid", "")) or ""
    pci = "pci:%s:%s:%s" % (getattr(props, "pci_domain_id", "?"), getattr(props, "pci_bus_id", i), getattr(props, "pci_device_id", "?"))
    if not ident or set(ident) <= set("0-"):
        ident = pci
    elif "?" not in pci:
        ident = "%s@%s" % (ident, pci)        # a runtime that reports one UUID for every device must not make all ranks look like one
    return socket.gethostname(), ident


def duplicate_devices(identities):
    """[(host, device identity) per rank] -> {(host, identity): [ranks]} of the devices that more than one rank resolved to."""
    seen = {}
    for r, ident in enumerate(identities):
        seen.setdefault(tuple(ident), []).append(r)
    return {k: v for k, v in seen.items() if len(v) > 1}


def gather_identities():
    """Every rank's :func:`device_identity`, on every rank (one small all_gather_object; [own] without a group)."""
    if not world_initialized():
        return [device_identity()]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, device_identity())
    return [tuple(x) for x in out]


def assert_one_device_per_rank():
    """Under RCCL every rank must own a GPU of its own (HIP_VISIBLE_DEVICES masks, a launcher that sets LOCAL_RANK wrongly):
    checked once, right after the group is up, and fatal on every rank."""
    dup = duplicate_devices(gather_identities())
    if dup and os.environ.get("SVX_DEVICE_CHECK", "fatal") in ("warn", "off"):       # SVX_DEVICE_CHECK=warn|off: a runtime whose identities cannot be trusted
        import logging
        logging.warning("ranks seem to share a GPU under RCCL (SVX_DEVICE_CHECK=%s, going on): %s", os.environ["SVX_DEVICE_CHECK"], dup)
        return
    if dup:
        raise RuntimeError("ranks share a GPU under the nccl (RCCL) backend: %s" % "; ".join("%s %s <- ranks %s" % (h, d, r) for (h, d), r in sorted(dup.items())))


def _comm_device():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def shard_chromosomes(chroms, weights, n_ranks):
    """Longest-processing-time-first assignment of chromosomes to ranks.
    -> list (per rank) of chromosome names, each in the original task order."""
    load = [0.0] * n_ranks
    owner = {}
    for name, w in sorted(zip(chroms, weights), key=lambda t: -t[1]):
        r = min(range(n_ranks), key=lambda k: (load[k], k))
        owner[name] = r
        load[r] += w
    return [[c for c in chroms if owner[c] == r] for r in range(n_ranks)]


def exchange_score_range(local_scores):
    """Global (max, min) of the per-record scores; (None, None) when no rank has any."""
    rank, ws = world()
    has = len(local_scores) > 0
    mx = float(np.max(local_scores)) if has else -np.inf
    mn = float(np.min(local_scores)) if has else np.inf
    if dist.is_available() and dist.is_initialized():
        dev = _comm_device()
        t_max = torch.tensor([mx], dtype=torch.float64, device=dev)
        t_min = torch.tensor([mn], dtype=torch.float64, device=dev)
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_min, op=dist.ReduceOp.MIN)
        mx, mn = float(t_max.item()), float(t_min.item())
    if mx == -np.inf:
        return None, None
    return np.float64(mx), np.float64(mn)


def gather_texts(texts, dst=0):
    """{key: str} on every rank -> merged dict on rank ``dst`` (None elsewhere).
    Packed as bytes: all_gather of sizes, then one padded all_gather of the payloads."""
    rank, ws = world()
    if not (dist.is_available() and dist.is_initialized()):
        return dict(texts)
    import json
    payload = json.dumps(texts).encode()
    dev = _comm_device()
    size = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ws)]
    dist.all_gather(sizes, size)
    cap = int(max(int(s.item()) for s in sizes))
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if payload:
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    out = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(ws)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    merged = {}
    for r in range(ws):
        n = int(sizes[r].item())
        merged.update(json.loads(bytes(out[r][:n].cpu().numpy().tobytes()).decode()))
    return merged
