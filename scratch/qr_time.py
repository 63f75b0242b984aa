"""Timing of kktsolver='qr' vs 'chol' on the bench workload (config 5 synthetic) and a feasible-start solve."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from smcp_amd import _lib, problems, chordal
from smcp_amd.symbolic import Symbolic
from smcp_amd.cspmatrix import cspmatrix
from smcp_amd.kkt import KKTSystem
import bench
prob = bench.build_workload() if hasattr(bench, "build_workload") else None
print("workload helper", prob is not None)
