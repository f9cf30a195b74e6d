"""Worker of tests/test_gpu_group.py::test_group_over_two_processes_maps_the_peer_buffers_through_hipipc: one rank of a group
whose collectives are the caller's (torch.distributed / gloo).  Launched by torch.distributed.run, two ranks, both on cuda:0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import cpprob_amd as cp
    dist.init_process_group("gloo")
    world, rank = dist.get_world_size(), dist.get_rank()
    z = np.load(sys.argv[1])

    def allgather(b):
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return b"".join(bytes(o.numpy().tobytes()) for o in outs)

    ok = True
    for model, key, T, ess, shards in ((cp.MODEL_HMM3, "hmm16", 16, 2.0, [70001, 49999]), (cp.MODEL_HMM3, "hmm128", 40, 2.0, [20000, 50000]),
                                       (cp.MODEL_LINEAR_GAUSSIAN_1D, "lgssm100", 20, 0.5, [60000, 60000])):
        obs = z[key][:T]
        n = int(sum(shards))
        g = cp.Group([0], world=world, first_rank=rank, allgather=allgather)
        g.begin(cp.ALG_SMC, model, obs, n, seed=23, ess_threshold=ess, shard_sizes=shards)
        g.run(0)
        g.run(1)
        stats, s, reruns = g.results()
        tr = g.traffic()
        e = g.context(0)
        e.n = shards[rank]
        paths = e.paths()
        g.close()
        assert tr["transport"] == cp.capi.TRANSPORT_DIRECT and tr["remote_lineages"] == 1 and tr["records"] > 0 and tr["wire_bytes"] == tr["payload_bytes"], tr
        # the per-step collectives went through the mailboxes (two processes spinning on each other's stores), not through gloo
        assert tr["mailbox_collectives"] == 1 and reruns == 0, (tr, reruns)
        # every rank's shard of the traces, gathered on rank 0, against ONE context holding all particles
        gathered = [None] * world
        dist.all_gather_object(gathered, paths)
        if rank == 0:
            eng = cp.Engine(0)
            eng.begin(cp.ALG_SMC, model, obs, n, seed=23, ess_threshold=ess)
            eng.run(1)
            ref_stats, ref_sum, ref_paths = eng.stats(), eng.summary(), eng.paths()
            eng.close()
            allp = np.concatenate(gathered, axis=1)
            differ = int((allp != ref_paths).any(axis=0).sum())
            ok = ok and differ == 0 and s["log_evidence"] == ref_sum["log_evidence"] and np.abs(stats - ref_stats).max() < 1e-13
            print("case", key, T, "differ", differ, "records", tr["records"], "bytes", tr["wire_bytes"], "reruns", reruns, flush=True)
    flag = [ok]
    dist.broadcast_object_list(flag, src=0)
    dist.barrier()
    if rank == 0 and flag[0]:
        print("EXT_GROUP_OK", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if flag[0] else 1)


if __name__ == "__main__":
    main()
