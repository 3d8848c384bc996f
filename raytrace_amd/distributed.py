"""Multi-GPU sharding of the sample axis + the film reduce.

The path shards by sample number: sample k of pixel p is addressable directly
(SobolIntervalToIndex, Enigma/SamplerInclude.h:2600-2620; GlobalSampler::SetSampleNumber,
Enigma/Sampler.cpp:184-189), paths never interact and the film is a plain sum
(Enigma/Film.cpp:117-120,296-297).  Every rank holds a full scene replica, renders a disjoint
range of sample numbers for all pixels into its own film, and ONE sum-reduce of the
{r, g, b, weight} accumulators produces the film on rank 0.  No other exchange exists.

On GPUs that reduce is the library's own (enigma_comm_init / enigma_film_reduce: ncclReduce over RCCL/xGMI on the
library's stream; Renderer.comm_init / film_reduce).  The torch helpers below are the cross-check of it and the CPU
stand-in (gloo) of the tests; torch is imported only when they are used.
"""


def shard_range(begin, end, rank, world):
    """Contiguous, disjoint, exhaustive split of sample numbers [begin, end) over ranks."""
    n = end - begin
    lo = begin + (n * rank) // world
    hi = begin + (n * (rank + 1)) // world
    return lo, hi


class _DeviceArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr, n_floats):
        self.__cuda_array_interface__ = {"shape": (int(n_floats),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def film_tensor(renderer, device):
    """torch view (no copy) of the renderer's device film accumulators."""
    import torch
    ptr, n = renderer.film_device_ptr()
    return torch.as_tensor(_DeviceArray(ptr, n), device=device)


def reduce_film(t, dst=0):
    """Sum-reduce a film tensor onto rank dst (in place on dst)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
    return t
