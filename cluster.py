#!/usr/bin/env python
"""Launcher: the command line lives in trrosettax2-dynamics_amd/cluster.py (same flags as the reference's cluster.py)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
main = importlib.import_module("trrosettax2-dynamics_amd.cluster").main

if __name__ == "__main__":
    sys.exit(main())
