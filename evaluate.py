#!/usr/bin/env python
"""Compare predicted with native structures: same flags as the reference's evaluate.py (no external TM-score binary)."""
import argparse
import importlib
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    p = argparse.ArgumentParser(description="Evaluate predicted structures by comparing with native structures.")
    p.add_argument("--native_dir", "-n", required=True, type=str, help="Directory with native PDB structures.")
    p.add_argument("--pred_dir", "-p", required=True, type=str, help="Directory with predicted PDB structures.")
    p.add_argument("--output", "-o", type=str, default=None, help="Output summary file or directory (default: predicted directory).")
    p.add_argument("--align", action="store_true", default=False, help="TM-score's -seq option (not implemented here)")
    a = p.parse_args(argv)
    ev = importlib.import_module("trrosettax2-dynamics_amd.evaluate")
    if a.output:   # evaluate.py:27-39
        out_dir, out_file = (os.path.dirname(a.output) or os.getcwd(), os.path.basename(a.output)) if a.output.endswith(".txt") else (a.output, "summary.txt")
        os.makedirs(out_dir, exist_ok=True)
    else:
        out_dir, out_file = a.pred_dir, "summary.txt"
    mn, mx, mr, mt = ev.run_score(a.native_dir, a.pred_dir, align=a.align, save_summary=True, save_dir=out_dir)
    if out_file != "summary.txt":
        shutil.move(os.path.join(out_dir, "summary.txt"), os.path.join(out_dir, out_file))
    print("Evaluation Summary:")
    print(f"  Min RMSD: {round(mn, 3)}")
    print(f"  Max TM-score: {round(mx, 3)}")
    print(f"  Mean RMSD: {round(mr, 3)}")
    print(f"  Mean TM-score: {round(mt, 3)}")
    print(f"Full summary saved to: {os.path.join(out_dir, out_file)}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
