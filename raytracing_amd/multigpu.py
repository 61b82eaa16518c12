"""Sample slicing across GPUs.

The reference is single-GPU; what it defines is the per-pixel running mean over the
samples of a frame (shader/pathtracing.glsl:657-670).  Samples are independent, so G
ranks each hold the whole scene and render a disjoint slice of the global sample
indices 1..SPP; seeds and jitter are functions of the GLOBAL index, so the set of
samples -- and therefore the image up to fp32 summation order -- does not depend on G.
Each rank accumulates plain sums (pt_params.accum_mode = 1); one all-reduce(sum) over the
nine accumulated floats per pixel and a division by SPP finish the frame.  The four
last-write outputs (motion, depth, normal, instanceIndex) are taken from the rank that
rendered the last sample, which is what a single GPU would have left in them.

On GPUs the combine is the library's own (pt_pass_allreduce over RCCL, device.Comm): the pass
keeps each group of outputs in one allocation, so it is one all-reduce and one broadcast with no
staging copy.  combine() below is the same arithmetic over torch.distributed, used where RCCL
cannot run: gloo on CPUs (tests) and rehearsals of several ranks on one GPU.
"""
import numpy as np

ACCUMULATED = ("directLum", "indirectLum", "albedo")
LAST_WRITE = ("motion", "depth", "normal", "instanceIndex")
CHANNELS = {"directLum": 3, "indirectLum": 3, "albedo": 3, "motion": 2, "depth": 1, "normal": 3, "instanceIndex": 1}


def sample_slice(total_spp, rank, world):
    """1-based global sample indices rendered by `rank`: s with (s - 1) % world == rank."""
    return list(range(1 + rank, total_spp + 1, world))


def owner_of_last_sample(total_spp, world):
    return (total_spp - 1) % world


def combine(dist, tensors, total_spp, world, torch, via_cpu=False):
    """tensors: dict name -> flat float32 tensor holding this rank's sums (ACCUMULATED) and
    last-write values (LAST_WRITE).  In place: afterwards every rank holds the frame.
    Two collectives per frame: ONE all-reduce(sum) over the nine accumulated floats per pixel
    (staged through one contiguous tensor; the outputs are separate allocations) and one
    broadcast of the seven last-write floats per pixel from the rank that rendered the last
    sample.  via_cpu stages them through host memory, for backends without device support
    (gloo rehearsals of the GPU path)."""
    owner = owner_of_last_sample(total_spp, world)
    acc = torch.cat([tensors[name] for name in ACCUMULATED])
    buf = acc.cpu() if via_cpu else acc
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    acc = buf.to(acc.device) if via_cpu else buf
    acc.mul_(1.0 / float(total_spp))
    off = 0
    for name in ACCUMULATED:
        n = tensors[name].numel()
        tensors[name].copy_(acc[off:off + n])
        off += n
    last = torch.cat([tensors[name] for name in LAST_WRITE])
    buf = last.cpu() if via_cpu else last
    dist.broadcast(buf, src=owner)
    last = buf.to(last.device) if via_cpu else buf
    off = 0
    for name in LAST_WRITE:
        n = tensors[name].numel()
        tensors[name].copy_(last[off:off + n])
        off += n
    return tensors


def allreduce_outputs(torch, dist, pass_, npix, total_spp, dev_tensor, via_cpu=False):
    """GPU form: wraps the pass's output buffers as torch tensors (no copy) and combines
    them over the default process group (RCCL)."""
    world = dist.get_world_size()
    tensors = {}
    for name, c in CHANNELS.items():
        tensors[name] = dev_tensor(torch, pass_.outputs[name].device_ptr, npix * c)
    combine(dist, tensors, total_spp, world, torch, via_cpu=via_cpu)
    torch.cuda.synchronize()


def make_comm(dist, ctx, rank, world):
    """device.Comm for this rank: rank 0 draws the RCCL unique id, torch.distributed (whatever
    backend the process group has) carries it to the others."""
    from . import device
    box = [None]
    if rank == 0:
        try:
            box[0] = device.Comm.unique_id()
        except Exception as e:  # the others must not wait for an id that will not come
            box[0] = "pt_comm_unique_id failed: %s" % e
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    if not isinstance(box[0], bytes):
        raise RuntimeError(box[0])
    return device.Comm(ctx, box[0], rank, world)


def allreduce_outputs_capi(comm, pass_, total_spp):
    """The C-ABI form: both collectives and the 1 / SPP scaling are queued by the library on the
    context's stream behind the draws; synchronise the context to wait for the frame."""
    comm.allreduce(pass_, total_spp, owner_of_last_sample(total_spp, comm.world))
