"""Same-process A/B of the interpreter switch interval the prefetch producer runs with (data.Prefetch.switch_interval): the HDF5-fed
train_epoch of bench.py's `hdf5_loop_cfg3` leg at several intervals, alternating, against the in-memory loop.  One GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import recommendersystem_amd as ra
from recommendersystem_amd import data, workload as synth
import bench

for rep in range(2):
    for iv in (5e-3, 1e-3, 3e-4, 1e-4, 3e-5):
        data.Prefetch.switch_interval = iv
        r = bench.hdf5_loop_leg(ra, synth, 64)
        print(f"rep {rep} switch interval {iv:g}: {r['ms_per_step']:.3f} ms/step", flush=True)
