"""Row sharding of the training set for the sparse models (one rank per GPU)."""


def shard_range(n, rank, world):
    """Half-open row range [r0, r1) of `rank`: contiguous blocks whose sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (n * rank) // world, (n * (rank + 1)) // world
