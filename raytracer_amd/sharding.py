"""Sample-batch sharding over GPUs (one process per GPU).

Every (pixel, sample, lambda-sample) is independent and Philox is keyed by the
GLOBAL sample index, so rank r of G renders its own sample ranges of every
pixel into its own spectral buffer; one all-reduce (sum) combines them and the
resolve divides by the total sample count.  No collective sits on the data path
before that.  The reference's own cross-pass merge is `merged + loop`
(2d/src/librt2d_reverse.rs:463-467, img.rs:257-274)."""


def sample_offset(step, rank, world_size, spp_per_step):
    """First global sample index of `rank`'s pass number `step`: step-major, rank-minor,
    so the union over ranks and steps is the contiguous range [0, steps * world_size * spp)."""
    if not (0 <= rank < world_size) or step < 0 or spp_per_step <= 0:
        raise ValueError("bad shard coordinates")
    return (step * world_size + rank) * spp_per_step


def total_spp(steps, world_size, spp_per_step):
    return steps * world_size * spp_per_step


def strong_range(rank, world_size, total_samples):
    """Strong scaling: the fixed global sample range [0, total_samples) cut into world_size contiguous parts, the
    first total_samples % world_size of them one sample longer.  Returns this rank's [lo, hi)."""
    if not (0 <= rank < world_size) or total_samples < 0:
        raise ValueError("bad shard coordinates")
    base, extra = divmod(total_samples, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
