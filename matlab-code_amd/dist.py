"""Multi-GPU plumbing (one process per GPU).

The CP tensor is row-sharded along its first mode; factor matrices, Grams and
the whole ADMM inner loop are replicated, so the only data crossing xGMI is the
all-reduce of each MTTKRP output (I_n x R doubles, <= 320 KB at 2000 x 20) --
SURVEY 8e rows C1/C2 collapse into that one collective per mode update.
`row_block` is the same split `Engine::tensor_upload` / `tensor_synth` use in
csrc/solver.hip.
"""
from __future__ import annotations


def row_block(rows: int, world: int, rank: int):
    """(first_row, n_rows) of `rank`'s block: ceil(rows/world) rows per rank, last ranks may be short."""
    per = -(-int(rows) // int(world))
    r0 = min(int(rows), per * int(rank))
    r1 = min(int(rows), r0 + per)
    return r0, r1 - r0


def init_engine_comm(eng, dist_module):
    """Bootstrap the library's RCCL communicator over an existing torch.distributed group
    (any backend: only a 128-byte id is broadcast)."""
    rank = dist_module.get_rank()
    world = dist_module.get_world_size()
    if world == 1:
        return
    ids = [eng.comm_unique_id() if rank == 0 else None]
    dist_module.broadcast_object_list(ids, src=0)
    eng.comm_init_rank(ids[0], rank, world)
