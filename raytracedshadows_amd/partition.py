"""Row-stripe partition of a frame over GPUs: no exchange, disjoint output."""


def stripe_rows(height, n_ranks, rank, band=32, interleaved=True):
    """Row ranges [(begin, end), ...] owned by `rank`.

    interleaved: bands of `band` rows dealt round-robin (balances sky vs geometry);
    otherwise one contiguous block of rows per rank, rounded to `band`.  The default band (32) is what
    rts_trace_shadow_mask_stripes_device accepts for every kernel (the default packet kernel also takes multiples of 8).
    """
    if n_ranks < 1 or not 0 <= rank < n_ranks or band < 1:
        raise ValueError("bad stripe arguments")
    if interleaved:
        return [(b, min(height, b + band)) for i, b in enumerate(range(0, height, band)) if i % n_ranks == rank]
    nb = (height + band - 1) // band
    b0 = (nb * rank) // n_ranks * band
    b1 = (nb * (rank + 1)) // n_ranks * band
    b0, b1 = min(b0, height), min(b1, height)
    return [(b0, b1)] if b1 > b0 else []
