"""How validate() cuts a ray range into render() chunks (rays are independent: the chunk is a free knob, implicit_surface.py:437-453 uses 256)."""

# rays per render() chunk the kernels are tuned and measured at (bench.py's headline: 307 200 rays = 10 chunks of 30 720);
# a chunk's transient buffers are ~1.3 GB at this size
MAX_VAL_CHUNK = 32768


def balanced_chunk(n_rays, max_chunk=MAX_VAL_CHUNK, unit=256):
    """Rays per render() chunk for a ray range of n_rays: the fewest chunks of at most max_chunk rays, of EQUAL length (a multiple of the
    reference's 256-ray chunk), instead of full chunks + a short tail that runs at a fraction of the chip's occupancy (an eighth of a
    480 x 640 image is 38 400 rays: 2 x 19 200, not 32 768 + 5 632).  The jitter is drawn in 256-ray groups whatever the chunk (JitterStream)."""
    n_chunks = max(1, -(-n_rays // max_chunk))
    per = -(-n_rays // n_chunks)
    return max(unit, min(max_chunk, -(-per // unit) * unit))
