"""
Multi-GPU path: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm).

The hot path shards by contig: every rank sketches its own records of EVERY assembly (the sketch is >95 % of
the work and needs no communication).  Uniqueness and intersection need each assembly's complete hash multiset,
so the path has exactly one exchange step per assembly: an all-gather of the rank-local sketches
(out_hash u64, pos u32, record u32, strand u8 = 17 B per minimizer; <= a few hundred MB even at 20 Gbp).  The four
arrays travel as ONE byte buffer per assembly (one collective, fixed rank order => deterministic
concatenation => results identical for any number of ranks).  Every rank then builds the graph of the union
(replicated: cheaper than a distributed table at these sizes).

gather_sketches() is device-agnostic (tensors in, tensors out) so the same code is exercised on CPU with gloo
in tests/test_dist_cpu.py.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


class _DevArray:
    """Expose a raw HBM pointer to torch (zero-copy) through __cuda_array_interface__."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def wrap_device(ptr, n, typestr, device):
    if n == 0:
        dt = {"<i8": torch.int64, "<i4": torch.int32, "|u1": torch.uint8}[typestr]
        return torch.empty(0, dtype=dt, device=device)
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device)


def gather_sketches(local, n_records_local, group=None):
    """local: {"out_hash": int64[n], "pos": int32[n], "record": int32[n], "forward": uint8[n]} on any device.
    Returns the same dict for the UNION in rank order, record indices shifted by the number of records held by
    lower ranks, plus "n_records" (total) and "record_offset" (this rank's shift)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local["out_hash"].device
    n = int(local["out_hash"].numel())
    meta = torch.tensor([n, int(n_records_local)], dtype=torch.int64, device=dev)
    metas = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas, meta, group=group)
    metas = metas.view(world, 2).cpu()
    counts = metas[:, 0].tolist()
    nrecs = metas[:, 1].tolist()
    nmax = (max(max(counts), 1) + 7) // 8 * 8  # keeps every sub-array 8-byte aligned inside the byte buffer
    # one byte buffer per rank: [hash 8*nmax | pos 4*nmax | record 4*nmax | forward nmax]
    buf = torch.zeros(17 * nmax, dtype=torch.uint8, device=dev)
    if n:
        buf[0:8 * n] = local["out_hash"].contiguous().view(torch.uint8)
        buf[8 * nmax:8 * nmax + 4 * n] = local["pos"].contiguous().view(torch.uint8)
        buf[12 * nmax:12 * nmax + 4 * n] = local["record"].contiguous().view(torch.uint8)
        buf[16 * nmax:16 * nmax + n] = local["forward"].contiguous()
    allbuf = torch.empty(world * 17 * nmax, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(allbuf, buf, group=group)
    allbuf = allbuf.view(world, 17 * nmax)
    hs, ps, rs, fs = [], [], [], []
    rec_off = 0
    my_off = 0
    for r in range(world):
        c = counts[r]
        row = allbuf[r]
        if r == rank:
            my_off = rec_off
        if c:
            hs.append(row[0:8 * c].view(torch.int64))
            ps.append(row[8 * nmax:8 * nmax + 4 * c].view(torch.int32))
            rs.append(row[12 * nmax:12 * nmax + 4 * c].view(torch.int32) + rec_off)
            fs.append(row[16 * nmax:16 * nmax + c])
        rec_off += nrecs[r]
    def cat(parts, dt):
        return torch.cat(parts).contiguous() if parts else torch.empty(0, dtype=dt, device=dev)
    return {"out_hash": cat(hs, torch.int64), "pos": cat(ps, torch.int32), "record": cat(rs, torch.int32),
            "forward": cat(fs, torch.uint8), "n_records": rec_off, "record_offset": my_off, "counts": counts}


def all_gather_strings(strings, dev, group=None):
    """every rank's list of strings -> [list of rank 0, list of rank 1, ...] with two tensor collectives (lengths, then the
    length-prefixed bytes padded to the longest), instead of all_gather_object's pickling + per-object size exchange: record
    ids are the only Python objects this module ever sent, and at 5 x 10^5 contigs per assembly their pickles were the
    largest host-side cost of a first step.  `dev`: where the backend wants its tensors (the GPU for RCCL, the CPU for gloo)."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":
        dev = torch.device("cpu")
    enc = [s.encode("utf-8") for s in strings]
    lens = np.fromiter((len(b) for b in enc), dtype=np.uint32, count=len(enc))
    blob = lens.tobytes() + b"".join(enc)
    head = torch.tensor([len(enc), len(blob)], dtype=torch.int64, device=dev)
    heads = torch.empty((world, 2), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(heads.view(-1), head, group=group)
    heads = heads.cpu().numpy()
    cap = int(heads[:, 1].max())
    mine = torch.zeros(max(cap, 1), dtype=torch.uint8, device=dev)
    if blob:
        mine[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    every = torch.empty((world, max(cap, 1)), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(every.view(-1), mine, group=group)
    every = every.cpu().numpy()
    out = []
    for r in range(world):
        n, nb = int(heads[r, 0]), int(heads[r, 1])
        raw = every[r, :nb].tobytes()
        ln = np.frombuffer(raw[:4 * n], dtype=np.uint32)
        ends = 4 * n + np.cumsum(ln.astype(np.int64))
        starts = ends - ln
        out.append([raw[int(a):int(b)].decode("utf-8") for a, b in zip(starts, ends)])
    return out


def shard_records(lengths, world):
    """Greedy longest-processing-time assignment of records (by base count) to ranks, then restored to input
    order inside each rank: returns a list (per rank) of record indices.  Contigs shard naturally."""
    order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")
    load = np.zeros(world, dtype=np.int64)
    out = [[] for _ in range(world)]
    for i in order.tolist():
        r = int(np.argmin(load))
        out[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(x) for x in out]


def allgather_union_graph(eng, k, w, device, union=None, group=None, stream=None):
    """The exchange step + graph of the union.  Per step: ONE small all-gather with every assembly's (count, records)
    and ONE all-gather per assembly of its packed sketch (16 B per minimizer: out_hash, pos, record); packing and
    unpacking (rank-order concatenation, record-index shift) are library kernels (mxg_pack_sketch_device /
    mxg_set_sketch_gathered).  `union` (an MxEngine holding the union's record tables) is created on first use.
    `stream`: the torch.cuda.Stream that `eng` was created on (MxEngine(stream=stream.cuda_stream)); pack, collectives
    and unpack are then ordered by that stream and the only host sync of the exchange is the size read-back."""
    from .engine import MxEngine
    if stream is not None:
        with torch.cuda.stream(stream):
            return _allgather_union_graph(eng, k, w, device, union, group, stream)
    return _allgather_union_graph(eng, k, w, device, union, group, None)


def _allgather_union_graph(eng, k, w, device, union, group, stream):
    from .engine import MxEngine
    A = eng.n_assemblies
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device("cuda", device)
    if union is not None and getattr(union, "_slots", None) is not None:
        if _exchange_one_gather(eng, union, A, world, dev, group, stream):
            return union
    # first step (or a sketch that outgrew its slot): sizes first, then one all-gather per assembly
    # sizes of what follows: (count, records) per assembly from every rank.  Staged through pinned host tensors kept on
    # the engine (a fresh torch.tensor(list, device=...) alone costs ~50 us per step).
    xm = getattr(eng, "_xmeta", None)
    if xm is None or xm[0].shape[0] != A or xm[2].shape[0] != world:
        xm = (torch.empty((A, 2), dtype=torch.int64).pin_memory(), torch.empty((A, 2), dtype=torch.int64, device=dev),
              torch.empty((world, A, 2), dtype=torch.int64, device=dev), torch.empty((world, A, 2), dtype=torch.int64).pin_memory())
        eng._xmeta = xm
    meta_h, meta, metas_d, metas_h = xm
    mh = meta_h.numpy()
    glob = bool(getattr(eng, "global_records", False))  # every rank registered ALL records (shard / split loads): no index shift
    for a in range(A):
        mh[a, 0], mh[a, 1] = eng.sketch_size(a), (0 if glob else eng.n_records(a))
    meta.copy_(meta_h, non_blocking=True)
    dist.all_gather_into_tensor(metas_d.view(-1), meta.view(-1), group=group)
    metas_h.copy_(metas_d, non_blocking=True)
    torch.cuda.current_stream().synchronize()  # the one host sync of the exchange
    metas = metas_h.numpy().copy()
    if union is None:
        union = MxEngine(k=k, w=w, device=device, timing=True, stream=stream.cuda_stream if stream is not None else None)
        for a in range(A):
            if glob:
                flat = eng.record_ids(a, eng.n_records(a))
            else:
                ids_local = [f"r{rank}:{x}" for x in eng.record_ids(a, eng.n_records(a))]
                all_ids = all_gather_strings(ids_local, dev, group)
                flat = [x for part in all_ids for x in part]
            union.add_minimizers(eng.assembly_name(a), eng.assembly_weight(a), np.zeros(0, np.uint64),
                                 np.zeros(0, np.uint32), np.zeros(0, np.uint32), flat)
        union._xbuf = {}
        union._slots = None
    for a in range(A):
        counts = metas[:, a, 0].astype(np.uint64)
        nrecs = metas[:, a, 1].astype(np.uint64)
        rec_off = np.concatenate([[0], np.cumsum(nrecs)[:-1]]).astype(np.uint64)
        nmax = (max(int(counts.max()), 1) + 7) // 8 * 8
        bufs = union._xbuf.get(a)
        if bufs is None or bufs[0].numel() < 16 * nmax:  # exchange buffers are kept across steps
            cap = 16 * (nmax + nmax // 4 + 8)
            bufs = (torch.empty(cap, dtype=torch.uint8, device=dev), torch.empty(world * cap, dtype=torch.uint8, device=dev))
            union._xbuf[a] = bufs
        send, recv = bufs[0][:16 * nmax], bufs[1][:world * 16 * nmax]
        eng.pack_sketch_device(a, send.data_ptr(), nmax)
        dist.all_gather_into_tensor(recv, send, group=group)
        if stream is None:
            torch.cuda.current_stream().synchronize()  # the union handle works on its own stream
        union.set_sketch_gathered(a, recv.data_ptr(), nmax, counts, rec_off)
    # later steps: ONE all-gather.  Every rank's slot = header (count per assembly) + a fixed-capacity region per
    # assembly, 10 % above the largest sketch seen in this step on any rank (the same on all ranks by construction; the whole
    # slot travels, so the slack is paid on the links: 25 % until round 6);
    # pack, unpack and the graph stage then run with the counts on the device (mxg_xchg_*): one host sync per step.
    pct = int(os.environ.get("MXG_XCHG_SLOT_PCT", "110"))  # test knob: < 100 forces the fallback on every later step
    caps = [((int(metas[:, a, 0].max()) * pct // 100 + (64 if pct >= 100 else 0)) + 7) // 8 * 8 for a in range(A)]
    head = 64 * ((16 * A + 63) // 64)
    slot = head + 16 * sum(caps)
    union._slots = {"caps": caps, "head": head, "slot": slot,
                    "rec_off_flat": np.concatenate([np.concatenate([[0], np.cumsum(metas[:, a, 1])[:-1]]) for a in range(A)]).astype(np.uint64),
                    "send": torch.zeros(slot, dtype=torch.uint8, device=dev),
                    "recv": torch.empty(world * slot, dtype=torch.uint8, device=dev)}
    if overlap_exchange():
        # one buffer per assembly (64-byte header + its region): assembly a's all-gather is issued on a stream of its own as soon
        # as a's sketch is packed, and travels while assembly a + 1 is sketched (sketch_union_graph)
        # (12 bytes per minimizer: hash + position; the record column travels as the first entry of every record, mxg_sketch_pack_parts)
        rcaps = [(max(int(metas[:, a, 1].max()), eng.n_records(a), 1) + 3) // 4 * 4 for a in range(A)]
        union._slots["rcaps"] = rcaps
        union._slots["send_parts"] = [torch.zeros(part_bytes(c, rc), dtype=torch.uint8, device=dev) for c, rc in zip(caps, rcaps)]
        union._slots["recv_parts"] = [torch.empty(world * part_bytes(c, rc), dtype=torch.uint8, device=dev) for c, rc in zip(caps, rcaps)]
        if getattr(union, "_comm", None) is None:
            union._comm = torch.cuda.Stream(device=dev)
    union.build_graph()
    return union


PART_HEAD = 64  # bytes in front of an assembly's region in its own exchange buffer (XCHG_PART_HEAD in the library)


def part_bytes(cap, rcap):
    """an assembly's exchange buffer: header, `cap` hashes and positions, the first entry of each of `rcap` records"""
    return PART_HEAD + 12 * int(cap) + 4 * int(rcap)


def overlap_exchange():
    """MXG_XCHG_OVERLAP=0: the steady-state exchange as ONE all-gather behind every sketch (the round-5 path) instead of one per
    assembly overlapped with the next assembly's sketch"""
    return os.environ.get("MXG_XCHG_OVERLAP", "1") != "0"


def _sketch_union_graph_overlapped(eng, union, group, stream):
    """steady state, exchange overlapped with compute: every assembly's sketch is enqueued on `stream` (the library's two
    streams behind it) and packed into its own buffer right behind its last kernel; the communication stream waits for that
    event only and carries the all-gather of assembly a while assembly a + 1 is still being sketched.  `stream` then waits
    for the collectives, unpacks and builds the graph of the union: one host sync per step, as before."""
    sl = union._slots
    A = eng.n_assemblies
    comm = union._comm
    works = []
    with torch.cuda.stream(stream):
        eng.sketch_pack_parts([t.data_ptr() for t in sl["send_parts"]], sl["caps"], sl["rcaps"])
    for a in range(A):
        eng.part_packed_wait(a, comm.cuda_stream)
        with torch.cuda.stream(comm):
            works.append(dist.all_gather_into_tensor(sl["recv_parts"][a], sl["send_parts"][a], group=group, async_op=True))
    with torch.cuda.stream(stream):
        for wk in works:
            wk.wait()                       # (nccl: `stream` waits for the collective; gloo: the host does)
        stream.wait_stream(comm)
        ok = union.xchg_unpack_graph_parts([t.data_ptr() for t in sl["recv_parts"]], dist.get_world_size(group), sl["caps"],
                                           sl["rcaps"], sl["rec_off_flat"])
    comm.wait_stream(stream)                # (the next step's collectives must not overwrite what this unpack reads)
    eng.sketch_finish()
    return ok


def sketch_union_graph(eng, k, w, device, union=None, group=None, stream=None):
    """One step of the union path: sketch this rank's assemblies, exchange, graph of the union.  In steady state (fixed
    slots known, `eng` and `union` on `stream`) nothing waits for the host between the first sketch kernel and the last
    graph kernel: the sketches are enqueued, the packing kernels follow them and read the counts on the device
    (mxg_sketch_pack), then ONE all-gather, then mxg_xchg_unpack_graph, whose sync is the step's only one."""
    if stream is not None and union is not None and getattr(union, "_slots", None) is not None:
        sl = union._slots
        if "send_parts" in sl:
            if _sketch_union_graph_overlapped(eng, union, group, stream):
                return union
            union._slots = None
            return allgather_union_graph(eng, k, w, device, union, group=group, stream=stream)
        with torch.cuda.stream(stream):
            eng.sketch_pack(sl["send"].data_ptr(), sl["head"], sl["caps"])
            dist.all_gather_into_tensor(sl["recv"], sl["send"], group=group)
            ok = union.xchg_unpack_graph(sl["recv"].data_ptr(), dist.get_world_size(group), sl["slot"], sl["head"], sl["caps"],
                                         sl["rec_off_flat"])
        eng.sketch_finish()
        if ok:
            return union
        union._slots = None              # some rank's sketch travelled as -1: sizes first (the sketches are complete now)
    else:
        eng.sketch(-2)
    return allgather_union_graph(eng, k, w, device, union, group=group, stream=stream)


def _exchange_one_gather(eng, union, A, world, dev, group, stream):
    """steady-state exchange + graph of the union: pack (library, counts written by the kernels) -> ONE all-gather ->
    unpack with the counts read from the headers on the device + graph stage, one host sync in all (mxg_xchg_*).
    Returns False (nothing usable in `union`) when some sketch no longer fits its slot on some rank: every rank sees
    the same headers, so all fall back to the size exchange together."""
    sl = union._slots
    eng.xchg_pack(sl["send"].data_ptr(), sl["head"], sl["caps"])
    dist.all_gather_into_tensor(sl["recv"], sl["send"], group=group)
    if stream is None:
        torch.cuda.current_stream().synchronize()    # the union handle works on its own stream
    ok = union.xchg_unpack_graph(sl["recv"].data_ptr(), world, sl["slot"], sl["head"], sl["caps"], sl["rec_off_flat"])
    if not ok:
        union._slots = None
    return ok


def concat_tsv_parts(parts, continues, out_path):
    """Rank-ordered concatenation of the shards' TSV parts into `out_path`.  continues[r] (mxg_assembly_continues on rank
    r): part r's first line continues the record of the last line written so far (sub-record sharding) -- its entries
    are appended to that line instead of starting a new one."""
    with open(out_path, "wb") as out:
        pending = None                      # last line written so far, without its newline, kept back for a continuation
        for r, path in enumerate(parts):
            with open(path, "rb") as f:
                first = True
                for line in f:
                    line = line.rstrip(b"\n")
                    if first and continues[r] and pending is not None:
                        rest = line.split(b"\t", 1)[1] if b"\t" in line else b""
                        if rest:
                            pending += rest if pending.endswith(b"\t") else b" " + rest
                    else:
                        if pending is not None:
                            out.write(pending + b"\n")
                        pending = line
                    first = False
        if pending is not None:
            out.write(pending + b"\n")


def shard_range(lengths, shard, n_shards):
    """[lo, hi) of the records shard `shard` owns: contiguous, balanced by base count (the library's rule)."""
    import ctypes as C
    from . import capi
    lib = capi.load()
    arr = np.ascontiguousarray(lengths, dtype=np.uint64)
    lo, hi = C.c_uint64(), C.c_uint64()
    rc = lib.mxg_shard_range(arr.ctypes.data_as(C.POINTER(C.c_uint64)), len(arr), int(shard), int(n_shards),
                             C.byref(lo), C.byref(hi))
    if rc != 0:
        raise ValueError("mxg_shard_range: bad arguments")
    return int(lo.value), int(hi.value)


def allgather_inplace(eng, device, group=None):
    """Exchange step for engines whose assemblies were added with add_fasta_shard (global record indices, contiguous
    record ranges per rank): every assembly's rank-local sketch is replaced by the union, in rank order = global
    (record,pos) order.  One small all-gather of counts, one all-gather per assembly."""
    A = eng.n_assemblies
    world = dist.get_world_size(group)
    dev = torch.device("cuda", device)
    meta = torch.tensor([eng.sketch_size(a) for a in range(A)], dtype=torch.int64, device=dev)
    metas = torch.empty((world, A), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(metas.view(-1), meta, group=group)
    metas = metas.cpu().numpy()
    zeros = np.zeros(world, dtype=np.uint64)
    for a in range(A):
        counts = metas[:, a].astype(np.uint64)
        nmax = (max(int(counts.max()), 1) + 7) // 8 * 8
        send = torch.empty(16 * nmax, dtype=torch.uint8, device=dev)
        recv = torch.empty(world * 16 * nmax, dtype=torch.uint8, device=dev)
        eng.pack_sketch_device(a, send.data_ptr(), nmax)
        dist.all_gather_into_tensor(recv, send, group=group)
        torch.cuda.current_stream().synchronize()
        eng.set_sketch_gathered(a, recv.data_ptr(), nmax, counts, zeros)
    return metas


# ---- graph stage distributed by hash range (csrc/dgraph.hip) ------------------------------------------------------
def _u64p(arr):
    import ctypes as C
    return arr.ctypes.data_as(C.POINTER(C.c_uint64))


def _grow(cache, key, n_bytes, dev):
    t = cache.get(key)
    if t is None or t.numel() < n_bytes:
        t = torch.empty(max(int(n_bytes * 5 // 4) + 256, 256), dtype=torch.uint8, device=dev)
        cache[key] = t
    return t


def partitioned_graph(eng, k, w, device, owner=None, group=None, stream=None, sketch=False):
    """The graph stage without replication: every minimizer goes to the rank that owns its hash (all-to-all), the owner
    decides uniqueness / intersection for its hashes and numbers its vertices, the verdicts come back, the adjacency of
    this rank's records goes to the owners of the end points, and every owner emits its edges.  Work and memory per rank
    stay constant as ranks are added (the union path does N times the work on every rank).

    Returns the OWNER engine of this rank (created on first use, reused afterwards).  owner.get_graph() is this rank's
    part: vertex arrays of its own vertices, edges with edge_u = local vertex index, edge_v = GLOBAL vertex id; global
    id = base + local index with base = partitioned_totals(owner)["base"].  eng.get_mx_flags(a) afterwards are the flags
    of this rank's own minimizers.  Per step: 6 collectives (two size exchanges, items, verdicts, vertex counts,
    messages) and 5 host syncs (the split sizes of the two variable all-to-alls must be known on the host)."""
    import ctypes as C
    from .engine import MxEngine
    if stream is not None and torch.cuda.current_stream() != stream:
        with torch.cuda.stream(stream):
            return partitioned_graph(eng, k, w, device, owner, group, stream, sketch)
    # sketch=True: the call sketches `eng`'s assemblies itself -- in steady state (fixed slots, MXG_XCHG_OVERLAP != 0, a stream) every
    # assembly's items are packed on the device right behind its sketch and its all-to-all leaves while the next assembly is sketched
    sketched = not sketch
    overlap = (sketch and stream is not None and overlap_exchange() and owner is not None and getattr(owner, "_slots", None) is not None
               and os.environ.get("MXG_DG_EXACT") != "1" and not getattr(owner, "_no_overlap", False))
    if sketch and not overlap:
        eng.sketch(-2)
        sketched = True
    lib, A = eng._lib, eng.n_assemblies
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device("cuda", device)
    cur = torch.cuda.current_stream()
    if owner is None:
        owner = MxEngine(k=k, w=w, device=device, timing=True, stream=stream.cuda_stream if stream is not None else None)
        owner._rec_off = []
        glob = bool(getattr(eng, "global_records", False))  # every rank registered ALL records: record indices are global already
        for a in range(A):
            if glob:
                owner._rec_off.append(0)
                flat = eng.record_ids(a, eng.n_records(a))
            else:
                ids_local = [f"r{rank}:{x}" for x in eng.record_ids(a, eng.n_records(a))]
                all_ids = all_gather_strings(ids_local, dev, group)
                owner._rec_off.append(sum(len(p) for p in all_ids[:rank]))
                flat = [x for part in all_ids for x in part]
            owner.add_minimizers(eng.assembly_name(a), eng.assembly_weight(a), np.zeros(0, np.uint64), np.zeros(0, np.uint32),
                                 np.zeros(0, np.uint32), flat)
        owner._buf = {}
        # records cut between ranks (sub-record shards)?  Then the last shared minimizer of every rank travels to the ranks
        # after it, so that the adjacency across a cut is not lost (mxg_dg_last_shared / mxg_dg_set_ghosts).  Decided once,
        # together: the extra all-gather must be entered by every rank or by none.
        cut = torch.tensor([1 if (world > 1 and any(eng.assembly_continues(a) for a in range(A))) else 0], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(cut, op=dist.ReduceOp.MAX, group=group)
        owner._cut = bool(int(cut.item()))
        # small fixed-size staging, pinned on the host side: [world][A] size matrices, [world] message sizes, vertex counts
        owner._st = {"c_h": torch.empty((world, A), dtype=torch.int64).pin_memory(), "c_out": torch.empty((world, A), dtype=torch.int64, device=dev),
                     "c_in": torch.empty((world, A), dtype=torch.int64, device=dev), "g_h": torch.empty((world, A), dtype=torch.int64).pin_memory(),
                     "m_h": torch.empty(world, dtype=torch.int64).pin_memory(), "m_out": torch.empty(world, dtype=torch.int64, device=dev),
                     "m_in": torch.empty(world, dtype=torch.int64, device=dev), "mi_h": torch.empty(world, dtype=torch.int64).pin_memory(),
                     "nv": torch.zeros(1, dtype=torch.int64, device=dev), "nvs": torch.empty(world, dtype=torch.int64, device=dev),
                     "bases": torch.zeros(world + 1, dtype=torch.int32, device=dev)}
    buf, st = owner._buf, owner._st
    if getattr(owner, "_slots", None) is not None and os.environ.get("MXG_DG_EXACT") != "1":
        if _partitioned_slots(eng, owner, A, world, rank, dev, group, stream, sketch_inside=not sketched):
            return owner
        owner._slots = None                  # a slot was too small somewhere: this step the exact way, new capacities
        if not sketched:                     # ... or a sketch could not be packed on the device (an assembly that does not take the
            owner._no_overlap = True         # one-batch route would say so every step): from now on behind the sketches
        sketched = True                      # (the sketches are complete: mxg_sketch_finish redid what did not travel)
    prof = getattr(owner, "_prof", None)     # tools: owner._prof = {} collects host seconds per section
    import time as _time
    t_last = [_time.perf_counter()]

    def mark(name):
        if prof is not None:
            now = _time.perf_counter()
            prof[name] = prof.get(name, 0.0) + now - t_last[0]
            t_last[0] = now

    def chk(e, rc):
        if rc < 0:
            e._check(rc)

    # 1. where do my minimizers go: cnt[a][dest]                                                  (host sync 1)
    cnt = np.zeros(A * world, dtype=np.uint64)
    chk(eng, lib.mxg_dg_owner_counts(eng._h, world, _u64p(cnt)))
    cnt = cnt.reshape(A, world).astype(np.int64)
    to_dest = cnt.sum(axis=0)                                  # items per destination
    dest_start = np.concatenate([[0], np.cumsum(to_dest)[:-1]])
    n_send = int(to_dest.sum())
    send = _grow(buf, "send", n_send * 16, dev)
    for a in range(A):                                         # layout of the send buffer: [dest][assembly]
        starts = (dest_start + cnt[:a].sum(axis=0)).astype(np.uint64)
        chk(eng, lib.mxg_dg_pack_items(eng._h, a, world, owner._rec_off[a], _u64p(starts), C.c_void_p(send.data_ptr())))
    mark("1_counts_pack")
    # 2. tell every destination how much of which assembly is coming                             (host sync 2)
    st["c_h"].numpy()[:] = cnt.T
    st["c_out"].copy_(st["c_h"], non_blocking=True)
    dist.all_to_all_single(st["c_in"].view(-1), st["c_out"].view(-1), group=group)
    st["g_h"].copy_(st["c_in"], non_blocking=True)
    cur.synchronize()
    got = st["g_h"].numpy().copy()                             # [src][a]
    from_src = got.sum(axis=1)
    src_start = np.concatenate([[0], np.cumsum(from_src)[:-1]])
    n_recv = int(from_src.sum())
    mark("2_size_exchange")
    # 3. the items
    recv = _grow(buf, "recv", n_recv * 16, dev)
    dist.all_to_all_single(recv[:n_recv * 16].view(n_recv, 16), send[:n_send * 16].view(n_send, 16),
                           output_split_sizes=from_src.tolist(), input_split_sizes=to_dest.tolist(), group=group)
    mark("3_items_a2a")
    # 4. owner: uniqueness, intersection, local vertex ids (the vertex count stays on the device)
    if stream is None:
        cur.synchronize()                                      # ... and the items must have arrived
    secs = []
    for a in range(A):
        sec_start = (src_start + got[:, :a].sum(axis=1)).astype(np.uint64)
        sec_count = got[:, a].astype(np.uint64)
        secs.append((sec_start, sec_count))
        chk(owner, lib.mxg_dg_set_items(owner._h, a, C.c_void_p(recv.data_ptr()), world, _u64p(sec_start), _u64p(sec_count)))
    chk(owner, lib.mxg_dg_vertices(owner._h, C.c_void_p(st["nv"].data_ptr())))
    mark("4_owner_vertices")
    # 5. global vertex ids: rank r's vertices are [bases[r], bases[r + 1]) -- computed on the device
    dist.all_gather_into_tensor(st["nvs"], st["nv"], group=group)
    st["bases"][1:] = torch.cumsum(st["nvs"], 0).to(torch.int32)
    mark("5_bases")
    # 6. the verdicts travel back along the same splits
    ret_out = _grow(buf, "ret_out", n_recv * 8, dev)
    gbase_ptr = st["bases"].data_ptr() + 4 * rank
    for a in range(A):
        chk(owner, lib.mxg_dg_item_results(owner._h, a, C.c_void_p(gbase_ptr), world, _u64p(secs[a][0]), _u64p(secs[a][1]),
                                           C.c_void_p(ret_out.data_ptr())))
    ret_in = _grow(buf, "ret_in", n_send * 8, dev)
    dist.all_to_all_single(ret_in[:n_send * 8].view(n_send, 8), ret_out[:n_recv * 8].view(n_recv, 8),
                           output_split_sizes=to_dest.tolist(), input_split_sizes=from_src.tolist(), group=group)
    mark("6_verdicts_a2a")
    # 7. adjacency of MY records -> messages to the owners of the two end points                  (host sync 3)
    if stream is None:
        cur.synchronize()
    _ghosts(eng, owner, lib, A, world, rank, dev, ret_in, group, stream is None)
    mcnt = np.zeros(A * world, dtype=np.uint64)
    chk(eng, lib.mxg_dg_msg_counts(eng._h, world, C.c_void_p(ret_in.data_ptr()), C.c_void_p(st["bases"].data_ptr()), _u64p(mcnt)))
    mcnt = mcnt.reshape(A, world).astype(np.int64)
    m_to = mcnt.sum(axis=0)
    m_start = np.concatenate([[0], np.cumsum(m_to)[:-1]])
    n_msend = int(m_to.sum())
    msend = _grow(buf, "msend", n_msend * 16, dev)
    for a in range(A):                                         # messages carry their assembly: one bucket per destination
        starts = (m_start + mcnt[:a].sum(axis=0)).astype(np.uint64)
        chk(eng, lib.mxg_dg_pack_msgs(eng._h, a, world, C.c_void_p(st["bases"].data_ptr()), _u64p(starts), C.c_void_p(msend.data_ptr())))
    mark("7_msg_counts_pack")
    st["m_h"].numpy()[:] = m_to
    st["m_out"].copy_(st["m_h"], non_blocking=True)
    dist.all_to_all_single(st["m_in"], st["m_out"], group=group)
    st["mi_h"].copy_(st["m_in"], non_blocking=True)
    cur.synchronize()                                          #                                     (host sync 4)
    m_from = st["mi_h"].numpy().copy()
    n_mrecv = int(m_from.sum())
    mrecv = _grow(buf, "mrecv", n_mrecv * 16, dev)
    dist.all_to_all_single(mrecv[:n_mrecv * 16].view(n_mrecv, 16), msend[:n_msend * 16].view(n_msend, 16),
                           output_split_sizes=m_from.tolist(), input_split_sizes=m_to.tolist(), group=group)
    mark("8_msg_exchange")
    # 8. owner: edges whose first supporter's source vertex is mine                               (host sync 5)
    if stream is None:
        cur.synchronize()                                      # the owner handle runs on its own stream
    nv_l, ne_l = C.c_uint64(), C.c_uint64()
    chk(owner, lib.mxg_dg_edges(owner._h, C.c_void_p(mrecv.data_ptr()), n_mrecv, C.byref(nv_l), C.byref(ne_l)))
    mark("9_owner_edges")
    owner.dg = {"local_vertices": int(nv_l.value), "local_edges": int(ne_l.value), "rank": rank, "world": world}
    # capacities for the steady state (fixed slots, device-side counts): 30 % above the largest bucket any rank saw
    if A <= 8 and os.environ.get("MXG_DG_EXACT") != "1":
        big = torch.tensor(list(cnt.max(axis=1)) + [int(m_to.max()) if world else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(big, op=dist.ReduceOp.MAX, group=group)
        big = big.cpu().numpy()
        pct = int(os.environ.get("MXG_DG_SLOT_PCT", "130"))    # test knob: < 100 makes every slot overflow
        pad = 64 if pct >= 100 else 0
        caps = np.array([((int(x) * pct // 100 + pad) + 7) // 8 * 8 for x in big[:A]], dtype=np.uint32)
        owner._slots = {"caps": caps, "items": int(caps.sum()), "M": ((int(big[A]) * pct // 100 + pad) + 7) // 8 * 8,
                        "flag": torch.zeros(1, dtype=torch.int32, device=dev)}
    return owner


def _ghosts(eng, owner, lib, A, world, rank, dev, ret_in, group, own):
    """records cut between ranks: every rank's last shared minimizer per assembly, all-gathered, becomes the predecessor of the
    first shared minimizer of the ranks after it (device-side; one small collective, no host sync)"""
    import ctypes as C
    if not getattr(owner, "_cut", False):
        return
    st = owner._st
    if "last" not in st:
        st["last"] = torch.empty(A * 2, dtype=torch.int32, device=dev)
        st["lasts"] = torch.empty(world * A * 2, dtype=torch.int32, device=dev)
    rc = lib.mxg_dg_last_shared(eng._h, C.c_void_p(ret_in.data_ptr()), C.c_void_p(st["last"].data_ptr()))
    if rc < 0:
        eng._check(rc)
    dist.all_gather_into_tensor(st["lasts"], st["last"], group=group)
    if own:
        torch.cuda.current_stream().synchronize()   # the handle works on its own stream: the gathered values must be there
    rc = lib.mxg_dg_set_ghosts(eng._h, C.c_void_p(st["lasts"].data_ptr()), world, rank)
    if rc < 0:
        eng._check(rc)


def _partitioned_slots(eng, owner, A, world, rank, dev, group, stream, sketch_inside=False):
    """one step of the partitioned graph stage with fixed-capacity slots: all-to-all / all-gather collectives with equal
    splits, counts read on the device, ONE host sync (the owner's last kernel) + the agreement on overflow.  False: a
    slot overflowed on some rank (all ranks return False together).  The item slots are assembly-major -- assembly a's
    slots for all destinations side by side -- and travel in one all-to-all per assembly.  sketch_inside: `eng` has not been
    sketched yet; its sketches are enqueued here, every assembly's items are packed on the device right behind its last kernel
    (mxg_sketch_dg_pack_slots) and its all-to-all is issued on a communication stream that waits for that event only: the
    reference's items travel while the target is being sketched."""
    import ctypes as C
    lib, sl, st, buf = eng._lib, owner._slots, owner._st, owner._buf
    caps, items, M = sl["caps"], sl["items"], sl["M"]
    capp = caps.ctypes.data_as(C.POINTER(C.c_uint32))
    cur = torch.cuda.current_stream()
    mstride = 64 + 16 * M
    astride = [64 + 16 * int(c) for c in caps[:A]]
    abase = np.concatenate([[0], np.cumsum([world * x for x in astride])]).astype(np.int64)
    total = int(abase[A])

    def chk(e, rc):
        if rc < 0:
            e._check(rc)

    send = _grow(buf, "s_send", total, dev)[:total]
    recv = _grow(buf, "s_recv", total, dev)[:total]
    part = lambda t, a: t[int(abase[a]):int(abase[a + 1])].view(world, astride[a])  # noqa: E731  (assembly a's slots, one per rank)
    own = stream is None                                                      # handles on their own streams: a collective's
    if sketch_inside:
        comm = getattr(owner, "_comm", None)
        if comm is None:
            comm = owner._comm = torch.cuda.Stream(device=dev)
        comm.wait_stream(cur)                                                 # (the last step's owner kernels have read `recv`)
        roff = np.ascontiguousarray(owner._rec_off, dtype=np.uint32)
        chk(eng, lib.mxg_sketch_dg_pack_slots(eng._h, world, A, capp, roff.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_void_p(send.data_ptr())))
        works = []
        for a in range(A):
            chk(eng, lib.mxg_part_packed_wait(eng._h, a, C.c_void_p(comm.cuda_stream)))
            with torch.cuda.stream(comm):
                works.append(dist.all_to_all_single(part(recv, a), part(send, a), group=group, async_op=True))
        for wk in works:
            wk.wait()
        cur.wait_stream(comm)
        # a sender whose sketch could not be packed on the device marked ALL its slots (a count no slot can hold): every rank sees that
        # in what it received, so all leave together -- before this rank's own verdict lookup runs over an index it never wrote
        worst = torch.stack([part(recv, a)[:, :8].contiguous().view(torch.int64).max() for a in range(A)]).max()
        invalid = int(worst.item()) >= (1 << 40)                              # (the sync: the sketches are complete, the items have arrived)
        eng.sketch_finish()
        if invalid:
            return False
    else:
        for a in range(A):
            chk(eng, lib.mxg_dg_pack_slots(eng._h, a, owner._rec_off[a], world, A, capp, C.c_void_p(send.data_ptr())))
        for a in range(A):
            dist.all_to_all_single(part(recv, a), part(send, a), group=group)
    if own:                                                                   # result must be complete before a handle reads it
        cur.synchronize()
    chk(owner, lib.mxg_dg_owner_slots(owner._h, world, A, capp, C.c_void_p(recv.data_ptr()), C.c_void_p(st["nv"].data_ptr())))
    dist.all_gather_into_tensor(st["nvs"], st["nv"], group=group)
    st["bases"][1:] = torch.cumsum(st["nvs"], 0).to(torch.int32)
    if own:
        cur.synchronize()
    ret_out = _grow(buf, "s_ret_out", world * items * 8, dev)[:world * items * 8]
    ret_in = _grow(buf, "s_ret_in", world * items * 8, dev)[:world * items * 8]
    chk(owner, lib.mxg_dg_slot_results(owner._h, world, A, capp, C.c_void_p(recv.data_ptr()),
                                       C.c_void_p(st["bases"].data_ptr() + 4 * rank), C.c_void_p(ret_out.data_ptr())))
    dist.all_to_all_single(ret_in.view(world, items * 8), ret_out.view(world, items * 8), group=group)
    if own:
        cur.synchronize()
    msend = _grow(buf, "s_msend", world * mstride, dev)[:world * mstride]
    mrecv = _grow(buf, "s_mrecv", world * mstride, dev)[:world * mstride]
    _ghosts(eng, owner, lib, A, world, rank, dev, ret_in, group, own)
    chk(eng, lib.mxg_dg_pack_msg_slots(eng._h, world, M, C.c_void_p(ret_in.data_ptr()), C.c_void_p(st["bases"].data_ptr()),
                                       C.c_void_p(msend.data_ptr())))
    dist.all_to_all_single(mrecv.view(world, mstride), msend.view(world, mstride), group=group)
    if own:
        cur.synchronize()
    nv_l, ne_l, ovf = C.c_uint64(), C.c_uint64(), C.c_uint32()
    chk(owner, lib.mxg_dg_edges_slots(owner._h, C.c_void_p(mrecv.data_ptr()), world, M, C.byref(nv_l), C.byref(ne_l), C.byref(ovf)))
    # overflow anywhere means everybody repeats the step the exact way: the ranks have to agree
    sl["flag"].fill_(int(ovf.value != 0))
    dist.all_reduce(sl["flag"], op=dist.ReduceOp.MAX, group=group)
    if int(sl["flag"].item()):
        return False
    owner.dg = {"local_vertices": int(nv_l.value), "local_edges": int(ne_l.value), "rank": rank, "world": world}
    return True


def partitioned_totals(owner, group=None):
    """global figures of the distributed graph (one all-reduce + one all-gather): total vertices / edges, every rank's
    first global vertex id"""
    dg = owner.dg
    dev = owner._st["nv"].device
    loc = torch.tensor([dg["local_vertices"], dg["local_edges"]], dtype=torch.int64, device=dev)
    every = torch.empty((dg["world"], 2), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(every.view(-1), loc, group=group)
    every = every.cpu().numpy()
    bases = np.concatenate([[0], np.cumsum(every[:, 0])])
    dg.update({"bases": bases, "base": int(bases[dg["rank"]]), "vertices": int(every[:, 0].sum()), "edges": int(every[:, 1].sum())})
    return dg
