"""Multi-GPU sample-stream split.  The reference is single-device; this is
the build's own sharding rule, kept in one place so bench.py and the gloo CPU tests agree:

* rank r of G renders the GLOBAL frames r, r+G, r+2G, ... (disjoint tea() seed words, and only
  global frame 0 uses the pixel-centre jitter of raytrace.rgen:38);
* every rank accumulates raw per-frame sums (PT_ACCUM_SUM) against a replicated scene/BVH;
* one reduce(sum) of the W*H*4 fp32 image to rank 0 (RCCL over xGMI on GPUs, gloo in the CPU
  tests), then rank 0 divides by the total number of frames.  No collective on the data path.
"""
from __future__ import annotations


def frame_schedule(rank: int, world: int):
    """(first, stride) for pt_set_frame_schedule."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return rank, world


def global_frames(rank: int, world: int, frames_per_rank: int):
    first, stride = frame_schedule(rank, world)
    return [first + k * stride for k in range(frames_per_rank)]


def strong_frame_count(rank: int, world: int, total_frames: int) -> int:
    """Strong scaling over sample streams: a FIXED total of `total_frames` global frames (BASELINE configs[1]: 1024 spp)
    is split across the ranks by the same rule -- rank r renders the global frames r, r+G, ... below the total -- so the
    ranks' counts differ by at most one and their union is exactly 0 .. total_frames-1, whatever `world`."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    if total_frames < 0:
        raise ValueError("negative frame count")
    return len(range(rank, total_frames, world))


def tile_schedule(rank: int, world: int, width: int, height: int):
    """Image-tile split: rank r renders the r-th of `world` column strips,
    (x0, y0, w, h) for pt_set_tile.  Strip widths differ by at most one pixel; every rank renders the SAME
    global frames (schedule first=0, stride=1) into an image that is zero outside its strip, so the strips
    combine with the same reduce(sum) as sample streams and the result does not depend on `world` at all."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    if world > width:
        raise ValueError(f"{world} strips do not fit {width} columns")
    base, extra = divmod(width, world)
    x0 = rank * base + min(rank, extra)
    return x0, 0, base + (1 if rank < extra else 0), height


TILE_BLOCK_COLUMNS = 32


def interleaved_columns(rank: int, world: int, width: int, block_w: int = TILE_BLOCK_COLUMNS):
    """The image columns pt_set_tile_interleaved(rank, world, block_w) renders, in tile order: blocks of `block_w` columns
    dealt round-robin.  A column strip per rank (tile_schedule) leaves the ranks whose strip shows sky idle -- on configs[4]
    the slowest of 8 strips takes 2.1x the mean -- while every rank's share of interleaved blocks looks like the whole image."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    if block_w < 1:
        raise ValueError("block width must be positive")
    return [x for b in range(rank * block_w, width, world * block_w) for x in range(b, min(b + block_w, width))]


def reduce_sum_image(image_tensor, dist, dst: int = 0):
    """In-place reduce(sum) of the per-rank sum images onto `dst` (torch.distributed module passed in
    so the same code runs over nccl(=RCCL) and gloo)."""
    if image_tensor.is_cuda and dist.get_backend() == "gloo":
        # rehearsal on a box with fewer GPUs than ranks (bench.py --dist-backend gloo): through host memory
        host = image_tensor.cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
        if dist.get_rank() == dst:
            image_tensor.copy_(host)
        return image_tensor
    dist.reduce(image_tensor, dst=dst, op=dist.ReduceOp.SUM)
    return image_tensor


def finalize_mean(image, total_frames: int):
    """Rank 0: sums -> mean over all frames of all ranks; alpha back to 1 (raytrace.rgen:92)."""
    image[..., :3] /= float(total_frames)
    image[..., 3] = 1.0
    return image
