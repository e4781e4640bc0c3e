#!/usr/bin/env python
"""Structure clustering: same flags as the reference's cluster.py (plus --device for the GloCon matrix on the GPU)."""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    p = argparse.ArgumentParser(description="Cluster predicted structures based on GloCon or RMSD.")
    p.add_argument("--pdb_dir", "-d", required=True, type=str, help="Directory containing PDB files to cluster.")
    p.add_argument("--mode", "-m", choices=["glocon", "tmscore", "rmsd"], default="glocon")
    p.add_argument("--output_dir", "-o", type=str, default=None, help="default: pdb_dir/clusters_result")
    p.add_argument("--n_clusters", type=int, default=10)
    p.add_argument("--n_files", type=int, default=5)
    p.add_argument("--device", type=int, default=None, help="GPU index for the GloCon matrix (extension; default: numpy)")
    a = p.parse_args(argv)
    cl = importlib.import_module("trrosettax2-dynamics_amd.cluster")
    out = a.output_dir or os.path.join(a.pdb_dir, "clusters_result")
    r = cl.save_cluster_result(a.pdb_dir, n_clusters=a.n_clusters, n_files=a.n_files, output_dir=out, mode=a.mode, device=a.device)
    print("Clustering failed or not possible." if r == "no_cluster" else f"Clustering completed. Results saved in {out}.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
