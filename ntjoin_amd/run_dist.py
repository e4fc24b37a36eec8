#!/usr/bin/env python3
"""
Multi-GPU driver of the hot path, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        -m ntjoin_amd.run_dist -k 32 -w 1000 -p out --target scaf.fa --target_weight 1 \
        --references ref1.fa ref2.fa --reference_weights 2 2

Every rank opens every FASTA, keeps the records of its shard (contiguous record ranges balanced by bases, or with --split
equal base ranges that cut long records into pieces with a halo; record indices are global), sketches them, and writes its part of `<fasta>.k<k>.w<w>.tsv` (rank 0 concatenates the parts in
rank order = input order).  The sketches are exchanged with one RCCL all-gather per assembly; every rank then holds
the full sketches and builds the minimizer graph; every rank writes its 1/N of `<prefix>.mx.dot` at its own offset.  Same file names as ntJoin
(reference ntJoin:26,204 and bin/ntjoin.py:28).
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

from .dist import allgather_inplace, concat_tsv_parts
from .engine import MxEngine, MxError


def main(argv=None):
    ap = argparse.ArgumentParser(description="ntJoin hot path on N GPUs: FASTA -> .tsv sketches + <prefix>.mx.dot")
    ap.add_argument("-k", type=int, default=32)
    ap.add_argument("-w", type=int, default=1000)
    ap.add_argument("-p", "--prefix", default="out")
    ap.add_argument("--target", required=True)
    ap.add_argument("--target_weight", type=float, default=1.0)
    ap.add_argument("--references", nargs="+", required=True)
    ap.add_argument("--reference_weights", nargs="+", type=float, required=True)
    ap.add_argument("--variant", default="v2")
    ap.add_argument("--split", action="store_true",
                    help="shard by equal base ranges, cutting long records into pieces with a halo (default: whole records)")
    args = ap.parse_args(argv)
    if len(args.references) != len(args.reference_weights):
        sys.exit("ERROR: The length of supplied reference weights and number of references must be equal.")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    # test knobs: NTJOIN_DIST_BACKEND=gloo with NTJOIN_DIST_ONE_DEVICE=1 runs several ranks on ONE GPU (RCCL cannot share a
    # device between ranks); production is one rank per GPU over RCCL
    backend = os.environ.get("NTJOIN_DIST_BACKEND", "nccl")
    if os.environ.get("NTJOIN_DIST_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        fastas = list(args.references) + [args.target]          # references first (CLI order), target last
        weights = list(args.reference_weights) + [args.target_weight]
        tsvs = [f"{fa}.k{args.k}.w{args.w}.tsv" for fa in fastas]
        with MxEngine(k=args.k, w=args.w, variant=args.variant, device=local_rank) as eng:
            for fa, wt, tsv in zip(fastas, weights, tsvs):
                (eng.add_fasta_split if args.split else eng.add_fasta_shard)(tsv, wt, fa, rank, world)
            eng.sketch()
            for a, tsv in enumerate(tsvs):                       # this rank's records only
                eng.write_tsv(a, f"{tsv}.part{rank}", with_pos=True, with_strand=False, with_seq=True)
            # one flag per (rank, assembly): "my first record was begun by the rank before me" -- a tensor collective, which also
            # orders the part files before rank 0 reads them
            tdev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
            mine = torch.tensor([int(eng.assembly_continues(a)) for a in range(len(tsvs))], dtype=torch.int32, device=tdev)
            every = torch.empty((world, len(tsvs)), dtype=torch.int32, device=tdev)
            dist.all_gather_into_tensor(every.view(-1), mine)
            cont = every.cpu().numpy().astype(bool).tolist()
            if rank == 0:
                for a, tsv in enumerate(tsvs):
                    parts = [f"{tsv}.part{r}" for r in range(world)]
                    concat_tsv_parts(parts, [cont[r][a] for r in range(world)], tsv)
                    for part in parts:
                        os.remove(part)
            allgather_inplace(eng, local_rank)
            eng.build_graph()
            # every rank holds the whole graph: each formats 1/world of the .mx.dot text and writes it at its own place
            # (the sizes travel in one small all-gather); rank 0 alone took world times as long for the same bytes
            # A rank that fails (out of memory while formatting, a write error) must not leave the others waiting in the next
            # collective, nor a file with holes behind: the sizes travel with a status word (-1 = failed), the writes are followed
            # by an all-reduce of the return codes, and on any failure rank 0 removes the file and every rank exits non-zero.
            # (all ranks write ONE file at their own offsets: the prefix must lie on a file system they share coherently --
            # one host, or a cluster file system; otherwise gather the parts to rank 0.)
            dot = args.prefix + ".mx.dot"
            try:
                vb, eb = eng.dot_part_format(rank, world)
            except (MxError, MemoryError) as exc:
                print(f"ntjoin_amd.run_dist: rank {rank}: {exc}", file=sys.stderr, flush=True)
                vb, eb = -1, -1
            sizes = torch.empty((world, 2), dtype=torch.int64, device=tdev)
            dist.all_gather_into_tensor(sizes.view(-1), torch.tensor([vb, eb], dtype=torch.int64, device=tdev))
            sizes = sizes.cpu().numpy()
            if (sizes < 0).any():
                dist.barrier()
                return 1
            if rank == 0 and os.path.exists(dot):
                os.remove(dot)                                   # (nobody truncates once the writing has begun)
            dist.barrier()
            v_off = 10 + int(sizes[:rank, 0].sum())
            e_off = 10 + int(sizes[:, 0].sum()) + int(sizes[:rank, 1].sum())
            bad = 0
            try:
                eng.dot_part_write(dot, v_off, e_off, rank == 0, rank == world - 1)
            except (MxError, OSError) as exc:
                print(f"ntjoin_amd.run_dist: rank {rank}: {exc}", file=sys.stderr, flush=True)
                bad = 1
            status = torch.tensor([bad], dtype=torch.int32, device=tdev)
            dist.all_reduce(status, op=dist.ReduceOp.MAX)
            if int(status.item()):
                if rank == 0 and os.path.exists(dot):
                    os.remove(dot)
                dist.barrier()
                return 1
            if rank == 0:
                st = eng.stats()
                print(f"ntjoin_amd.run_dist: {world} GPU(s), {st['minimizers']} minimizers, {st['vertices']} vertices, "
                      f"{st['edges']} edges -> {args.prefix}.mx.dot", flush=True)
        dist.barrier()
    finally:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
