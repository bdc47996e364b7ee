"""Work partitioning for one-process-per-GPU runs.

Candidate roots are split across ranks exactly as the reference splits them
across MPI ranks (assign_indicies_by_rank_exhaustive,
/root/reference/src/model.cpp:1867-1911: contiguous chunks, the first
`mod` ranks take one extra).  Site blocks (new here) are
contiguous slices of the pattern axis; their per-block log-likelihoods are
summed with one all-reduce per batch of evaluations (RCCL on GPUs, gloo in
the CPU tests)."""


def chunk(count, rank, num_tasks):
    """[beg, end) of `count` items for `rank` (src/model.cpp:1899-1907)."""
    size, mod = count // num_tasks, count % num_tasks
    beg = size * rank + min(mod, rank)
    end = size * (rank + 1) + min(mod, rank + 1)
    return beg, end


def assign_candidates(root_count, rank, num_tasks, completed=()):
    """Root ids this rank evaluates; `completed` ids (a resumed run) are removed
    before chunking, as the reference does."""
    done = set(completed)
    left = [i for i in range(root_count) if i not in done]
    beg, end = chunk(len(left), rank, num_tasks)
    return left[beg:end]


def site_block(n_sites, rank, num_tasks):
    return chunk(n_sites, rank, num_tasks)


def grid_2d(world_size, site_groups):
    """BASELINE config c5: candidate groups x site shards.  Returns
    (candidate_groups, site_groups) with candidate_groups*site_groups == world."""
    if world_size % site_groups:
        raise ValueError("site_groups must divide the world size")
    return world_size // site_groups, site_groups


def rank_coords(rank, site_groups):
    """-> (candidate group index, site shard index) of a rank; ranks that share
    a candidate group are adjacent so their all-reduce stays on neighbouring
    xGMI links."""
    return rank // site_groups, rank % site_groups


def allreduce_lnl(values, group=None, mode="gather"):
    """Sum per-block log-likelihoods over the ranks of `group` in place.
    `values` is a torch tensor (device tensor with RCCL, CPU tensor with gloo).

    mode "gather" (default): all_gather + ((v0 + v1) + v2) + ... in RANK ORDER -- every rank
    holds the same bits by construction, whatever algorithm the backend picks, and the sum is
    the one the library's own reducer makes (csrc/comm.cpp RDAMD_COMM_SUM_GATHER,
    rdamd_rank_order_sum) and rd_amd's host reducer (tools/rendezvous.hpp).  The optimisers
    above branch on these sums: one ulp of difference between two ranks forks a trajectory.
    mode "allreduce": one all_reduce; the association of the sum is the backend's."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if mode == "allreduce" or world == 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
        return values
    if mode != "gather":
        raise ValueError("allreduce_lnl: mode is 'gather' or 'allreduce'")
    parts = [torch.empty_like(values) for _ in range(world)]
    dist.all_gather(parts, values.contiguous(), group=group)
    acc = parts[0]
    for p in parts[1:]:
        acc = acc + p
    values.copy_(acc)
    return values


def global_frequencies(local_freqs, local_weight, group=None, device=None):
    """Empirical base frequencies of the WHOLE alignment from each rank's
    site-block figures (site-sharded runs: the model must be the same on every
    rank).  rdamd_msa_empirical_frequencies normalises by (pattern-weight total
    x tips), so the global vector is the weight-total-weighted mean of the
    per-block vectors: one all-reduce of K+1 doubles."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([f * local_weight for f in local_freqs] + [float(local_weight)],
                     dtype=torch.float64, device=device or "cpu")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t = t.cpu()
    return [float(v) / float(t[-1]) for v in t[:-1]]
