"""Multi-GPU front-end under torchrun: one process per GPU, every rank runs the pbsim command line itself
(pbsim_cli_main, csrc/cli.cpp) with a torch.distributed communicator for the job's collectives:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m pbsim3_amd.run_multi --strategy wgs --method errhmm --errhmm M --genome G.fa --prefix out [...]

Same options, same files and the same stderr report as `pbsim` on one GPU (rank 0 prints and creates the files, every rank
writes its own byte ranges).  `pbsim --devices 0,1,..` is the same job with one host thread per GPU in one process.
Extra options: --backend nccl|gloo (nccl = RCCL; gloo for CPU-side rendezvous), --one-gpu (all ranks on device 0: the
plumbing check of a single-GPU box), --scratch-mb M (wave scratch per batch slot, = PBSIM_SCRATCH_MB), --comm rccl|torch
(rccl, the default with --backend nccl: the library's own communicator, ncclCommInitRank with the id carried by torch's store,
pbsim_rccl_comm_create; torch: the job's collectives as torch.distributed calls behind Python callbacks).
"""
import os
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    one_gpu = "--one-gpu" in argv
    if one_gpu:
        argv.remove("--one-gpu")
    backend, comm_kind = "nccl", "rccl"
    for opt in ("--backend", "--scratch-mb", "--comm"):
        if opt in argv:
            i = argv.index(opt)
            val = argv[i + 1]
            del argv[i:i + 2]
            if opt == "--backend":
                backend = val
            elif opt == "--comm":
                comm_kind = val
            else:
                os.environ["PBSIM_SCRATCH_MB"] = val
    import pbsim3_amd as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    P.bind_host_to_device(local)   # before the first HIP call: this rank's threads and pinned staging on its GPU's NUMA node
    import torch
    import torch.distributed as dist
    comm = native = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            if comm_kind == "rccl" and not one_gpu:
                try:
                    native = P.RcclComm.from_torch(dist, local)
                except P.PbsimError as e:
                    sys.stderr.write("pbsim3_amd.run_multi: rank %d: %s; falling back to torch.distributed callbacks\n"
                                     % (int(os.environ.get("RANK", "0")), e))
                ok = torch.tensor([1 if native else 0], device=torch.device("cuda", local))
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)      # every rank on the same kind of communicator
                if int(ok.item()) == 0 and native is not None:
                    native.close()
                    native = None
            comm = native.ref if native is not None else P.torch_comm(dist, torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
            comm = P.torch_comm(dist, torch.device("cpu"))
    sys.stderr.flush()
    rc = P.cli_main(argv, comm, local)
    if native is not None:
        native.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc & 255)


if __name__ == "__main__":
    main()
