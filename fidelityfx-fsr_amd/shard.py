"""Frame sharding across the GPUs of a node.

FSR 1.0 is stateless per frame ("no history buffer, no motion vectors"), so a batch shards as
independent frames: rank r of R owns one contiguous block.  No image data crosses xGMI; the only
collective of a multi-GPU run is the gather of throughput counters (bench.py).
"""


def frames_for_rank(total_frames, rank, world_size):
    """Contiguous block [begin, end) of frame indices owned by `rank`; blocks differ by at most 1."""
    if world_size <= 0 or not (0 <= rank < world_size) or total_frames < 0:
        raise ValueError("bad shard request: frames=%r rank=%r world=%r" % (total_frames, rank, world_size))
    q, r = divmod(total_frames, world_size)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)
