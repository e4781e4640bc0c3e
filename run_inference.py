#!/usr/bin/env python
"""Same command line as the reference's run_inference.py (:356-390), folding on the GPU.

The trX2 network front-end is not part of this package, so --msa / --msa_dir are accepted for compatibility but the
distograms are read from {save_dir}/{name}/pred_npz/{name}_NMR.npz (_Xray.npz) or from --npz_nmr / --npz_xray."""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    p = argparse.ArgumentParser(description="Protein conformation sampling: trRosettaX2-Dynamics fold on MI355X.")
    p.add_argument("--fasta", type=str); p.add_argument("--msa", type=str)
    p.add_argument("--fasta_dir", type=str); p.add_argument("--msa_dir", type=str)
    p.add_argument("--name", type=str); p.add_argument("--name_lst", type=str)
    p.add_argument("--save_dir", type=str, required=True)
    p.add_argument("--init_num", type=int, default=10)
    p.add_argument("--Nmax", type=int, default=300)
    p.add_argument("--angle", action=argparse.BooleanOptionalAction, default=True)
    p.add_argument("--mult_two_models", action=argparse.BooleanOptionalAction, default=True)
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--npz_nmr", type=str, default=None, help="precomputed NMR-model distogram (extension)")
    p.add_argument("--npz_xray", type=str, default=None, help="precomputed X-ray-model distogram (extension)")
    p.add_argument("--seed", type=int, default=None, help="seed of the random start torsions (extension)")
    p.add_argument("--targets_in_flight", type=int, default=None, help="batch mode: targets a rank folds at a time (extension; default keeps sixty-four "
                   "chains in flight, whose single-decoy folds share launches: 32 targets with both models, 64 with one; 1 = one after the other "
                   "as the reference does; the files do not depend on it)")
    p.add_argument("--candidates", type=int, default=1, help="decoys folded and written per feedback iteration, candidate 0 fed back "
                   "(extension; 1 = the reference's chain)")
    p.add_argument("--keep_tmp_npz", action="store_true", help="write tmp_npz/{name}{k}.npz for every iteration as the reference does "
                   "(nothing reads them here; the directory is deleted at the end either way) (extension)")
    a = p.parse_args(argv)
    if a.name_lst:
        if not a.fasta_dir:
            p.error("Batch mode requires --fasta_dir and --name_lst.")
    elif not a.fasta or not a.name:
        p.error("Single mode requires --fasta and --name.")
    dev = int(a.device.split(":")[1]) if ":" in a.device else 0
    pipe = importlib.import_module("trrosettax2-dynamics_amd.pipeline")
    kw = dict(init_num=a.init_num, Nmax=a.Nmax, angle=a.angle, mult_two_models=a.mult_two_models, device=dev, seed=a.seed,
              keep_tmp_npz=a.keep_tmp_npz, candidates=a.candidates)
    if a.name_lst:
        # run_inference.py:343-348, sharded over ranks when launched by torch.distributed.run (one process per GPU):
        # targets are independent; ranks pull the next target from a shared counter (sched.DynamicQueue on a TCPStore) and the only
        # collective is the final gather of the per-rank summaries
        names = [l.strip() for l in open(a.name_lst) if l.strip()]
        rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
        dist = None
        if world > 1:
            import datetime
            import torch
            import torch.distributed as dist
            # no collective runs on the data path; the long timeout covers the final barrier of ranks that finish hours apart
            # (the summary itself is gathered on a gloo group, sched.summary_group)
            long_ = datetime.timedelta(hours=24)
            if torch.cuda.is_available():
                torch.cuda.set_device(local)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=long_)
            else:
                dist.init_process_group("gloo", timeout=long_)
            kw["device"] = local
        dev_ = kw.pop("device")
        res = pipe.run_batch(names, a.fasta_dir, a.save_dir, rank=rank, world=world, dist=dist, device=dev_, targets_in_flight=a.targets_in_flight, **kw)
        if rank == 0:
            print(f"Batch finished: {res['decoys']} structures from {len(names)} targets on {world} rank(s) in {res['seconds']:.1f} s, "
                  f"{res['failed']} failed")
            for e in res["errors"]:
                print("  FAILED", e, file=sys.stderr)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 1 if res["failed"] else 0
    else:
        pipe.run_single(a.name, a.fasta, a.save_dir, npz_nmr=a.npz_nmr, npz_xray=a.npz_xray, **kw)
    return 0


if __name__ == "__main__":
    sys.exit(main())
